/*
 * qldpc_recon.hip -- reconciliation sessions: the engine behind an ecd2 LDPC packet handler.
 *
 * Fills what the reference left as `return 81` (subcomponents/qber_estim.c:337-340,420-423): between
 * QBER estimation and privAmp_sendPrivAmpMsgAndPrivAmp (subcomponents/priv_amp.c:38) one parity
 * message replaces the cascade_biconf exchange (subcomponents/cascade_biconf.c:427-940).  Rate choice
 * follows the harness (BS/src/main.cpp:29,235-266: highest rate <= min_cr(QBER, f)); frame formation
 * follows BS/src/main.cpp:348-362 (channel bits +-ln((1-p)/p), disclosed bits +-23.03).
 * Host logic only; the arithmetic is the batched HIP decoder / encoder behind the same C ABI.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <new>
#include <vector>

#include "../../include/qldpc.h"
#include "qldpc_graph.h"

#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            qldpc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));     \
            return QLDPC_EHIP;                                                                          \
        }                                                                                               \
    } while (0)

struct recon_entry {
    int K, M;
    qldpc_code *code;
    qldpc_encoder *enc;
    qldpc_decoder *dec;
    uint8_t *d_cls;       /* [N] */
    uint32_t *d_bits;     /* [max_blocks][Wn] */
    uint32_t *d_out;      /* [max_blocks][Wn] */
    float *d_mag;         /* [max_blocks] */
    int *d_iters, *d_ok;  /* [max_blocks] */
    int *d_nch;           /* [max_blocks] channel VNs per frame (block length) */
    int cls_key_bits;     /* K once the class mask is built */
};

struct qldpc_recon {
    qldpc_recon_cfg cfg;
    std::list<recon_entry> cache;   /* most recently used first */
};

static const uint32_t *crc_table()
{
    static uint32_t t[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[i] = c;
        }
        init = true;
    }
    return t;
}

/* CRC-32 (IEEE 802.3) over the key bits, fed MSB-first word by word, bits past n_bits masked to 0 */
extern "C" uint32_t qldpc_crc32_words(const uint32_t *w, int n_bits)
{
    const uint32_t *t = crc_table();
    uint32_t c = 0xFFFFFFFFu;
    const int nw = (n_bits + 31) / 32;
    for (int i = 0; i < nw; i++) {
        uint32_t x = w[i];
        if (i == nw - 1 && (n_bits & 31)) x &= 0xFFFFFFFFu << (32 - (n_bits & 31));
        for (int b = 3; b >= 0; b--) c = t[(c ^ (x >> (8 * b))) & 0xFF] ^ (c >> 8);
    }
    return c ^ 0xFFFFFFFFu;
}

extern "C" void qldpc_recon_cfg_default(qldpc_recon_cfg *c)
{
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->device = 0;
    c->efficiency = 1.4f;
    c->n_rates = 4;
    c->rates[0] = 0.5f; c->rates[1] = 0.7f; c->rates[2] = 0.8f; c->rates[3] = 0.9f;
    c->n_ite = 50;
    c->rule = QLDPC_RULE_NMS;
    c->rule_param = 0.75f;
    c->key_quantum = 1024;
    c->max_blocks = 1;
    c->seed = 7;
}

static void entry_free(recon_entry &e)
{
    qldpc_decoder_free(e.dec);
    qldpc_encoder_free(e.enc);
    qldpc_code_free(e.code);
    (void)hipFree(e.d_cls); (void)hipFree(e.d_bits); (void)hipFree(e.d_out); (void)hipFree(e.d_mag); (void)hipFree(e.d_iters); (void)hipFree(e.d_ok); (void)hipFree(e.d_nch);
}

extern "C" void qldpc_recon_free(qldpc_recon *r)
{
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    for (auto &e : r->cache) entry_free(e);
    delete r;
}

extern "C" int qldpc_recon_create(const qldpc_recon_cfg *cfg, qldpc_recon **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = nullptr;
    if (!cfg) return QLDPC_EINVAL;
    if (cfg->n_rates < 1 || cfg->n_rates > 8 || cfg->key_quantum < 32 || (cfg->key_quantum & 31) || cfg->max_blocks < 1 || cfg->n_ite < 1 ||
        !(cfg->efficiency > 0.0f)) {
        qldpc_set_error("recon_create: bad configuration");
        return QLDPC_EINVAL;
    }
    for (int i = 0; i < cfg->n_rates; i++)
        if (!(cfg->rates[i] > 0.0f && cfg->rates[i] < 1.0f) || (i && cfg->rates[i] <= cfg->rates[i - 1])) { qldpc_set_error("recon_create: rate table must be ascending in (0,1)"); return QLDPC_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { qldpc_set_error("no HIP device visible: libqldpc has no CPU fallback"); return QLDPC_ENODEV; }
    if (cfg->device < 0 || cfg->device >= ndev) { qldpc_set_error("device %d out of range", cfg->device); return QLDPC_ENODEV; }
    qldpc_recon *r = new (std::nothrow) qldpc_recon();
    if (!r) return QLDPC_ENOMEM;
    r->cfg = *cfg;
    *out = r;
    return QLDPC_OK;
}

/* rate = largest table entry <= min_cr(QBER, f) = 1 / (1 + f h(QBER))   (BS/src/main.cpp:29,241-266) */
extern "C" int qldpc_recon_plan(const qldpc_recon *r, int key_bits, float qber, qldpc_recon_msg *msg)
{
    if (!r || !msg) return QLDPC_EINVAL;
    if (key_bits < 32) { qldpc_set_error("recon_plan: key_bits=%d", key_bits); return QLDPC_ESIZE; }
    if (!(qber > 0.0f && qber < 0.5f)) { qldpc_set_error("recon_plan: qber=%g not in (0, 0.5)", (double)qber); return QLDPC_EINVAL; }
    const float need = qldpc_min_code_rate(qber, r->cfg.efficiency);
    int idx = -1;
    for (int i = 0; i < r->cfg.n_rates; i++) if (r->cfg.rates[i] <= need) idx = i;
    if (idx < 0) { qldpc_set_error("recon_plan: QBER %.4f needs rate <= %.3f, below the table", (double)qber, (double)need); return QLDPC_EUNSUPPORTED; }
    const int q = r->cfg.key_quantum;
    const int K = (key_bits + q - 1) / q * q;
    const double R = r->cfg.rates[idx];
    int M = (int)llround((double)K * (1.0 - R) / R);
    if (M < 2) M = 2;
    memset(msg, 0, sizeof(*msg));
    msg->rate_index = (uint32_t)idx;
    msg->key_bits = (uint32_t)key_bits;
    msg->code_k = (uint32_t)K;
    msg->code_m = (uint32_t)M;
    return QLDPC_OK;
}

static int get_entry(qldpc_recon *r, int K, int M, recon_entry **out)
{
    for (auto it = r->cache.begin(); it != r->cache.end(); ++it)
        if (it->K == K && it->M == M) { r->cache.splice(r->cache.begin(), r->cache, it); *out = &r->cache.front(); return QLDPC_OK; }
    recon_entry e;
    memset(&e, 0, sizeof(e));
    e.K = K; e.M = M; e.cls_key_bits = -1;
    const int N = K + M, Wn = (N + 31) / 32, B = r->cfg.max_blocks;
    int rc = qldpc_code_ira(N, K, 0.125f, 11, 3, r->cfg.seed, &e.code);
    if (!rc) rc = qldpc_encoder_create(e.code, "IRA", r->cfg.device, &e.enc);
    if (!rc) {
        qldpc_decoder_cfg dc;
        qldpc_decoder_cfg_default(&dc);
        dc.schedule = r->cfg.schedule == QLDPC_SCHED_HLAYERED ? QLDPC_SCHED_HLAYERED : QLDPC_SCHED_FLOODING; dc.rule = r->cfg.rule; dc.rule_param = r->cfg.rule_param; dc.n_ite = r->cfg.n_ite;
        dc.enable_syndrome = 1; dc.syndrome_depth = 1; dc.max_frames = B; dc.device = r->cfg.device;
        rc = qldpc_decoder_create(e.code, K, nullptr, &dc, &e.dec);
    }
    if (!rc && hipMalloc((void **)&e.d_cls, (size_t)N) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipMalloc((void **)&e.d_bits, sizeof(uint32_t) * (size_t)B * Wn) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipMalloc((void **)&e.d_out, sizeof(uint32_t) * (size_t)B * Wn) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipMalloc((void **)&e.d_mag, sizeof(float) * (size_t)B) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipMalloc((void **)&e.d_iters, sizeof(int) * (size_t)B) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipMalloc((void **)&e.d_ok, sizeof(int) * (size_t)B) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipMalloc((void **)&e.d_nch, sizeof(int) * (size_t)B) != hipSuccess) rc = QLDPC_ENOMEM;
    if (rc) { entry_free(e); return rc; }
    while (r->cache.size() >= 6) { entry_free(r->cache.back()); r->cache.pop_back(); }
    r->cache.push_front(e);
    *out = &r->cache.front();
    return QLDPC_OK;
}

static int check_msg(const qldpc_recon *r, const qldpc_recon_msg *m, int key_bits)
{
    const int q = r->cfg.key_quantum;
    if ((int)m->key_bits != key_bits || (int)m->code_k != (key_bits + q - 1) / q * q || m->code_m < 2 || m->rate_index >= (uint32_t)r->cfg.n_rates ||
        m->code_k > (1u << 26) || m->code_m > (1u << 26)) {
        qldpc_set_error("recon: message header does not match the block (key_bits %u vs %d, K %u, M %u)", m->key_bits, key_bits, m->code_k, m->code_m);
        return QLDPC_ESIZE;
    }
    return QLDPC_OK;
}

extern "C" int qldpc_recon_encode(qldpc_recon *r, const uint32_t *key_words, int key_bits, float qber, qldpc_recon_msg *msg, uint32_t *parity_words, int cap)
{
    if (!r || !key_words || !msg || !parity_words) return QLDPC_EINVAL;
    int rc = qldpc_recon_plan(r, key_bits, qber, msg);
    if (rc) return rc;
    const int K = (int)msg->code_k, M = (int)msg->code_m, N = K + M;
    const int Wk = K / 32, Wkey = (key_bits + 31) / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32;
    if (cap < Wm) { qldpc_set_error("recon_encode: parity buffer holds %d words, need %d", cap, Wm); return QLDPC_ESIZE; }
    HIPCHK(hipSetDevice(r->cfg.device));
    recon_entry *e;
    if ((rc = get_entry(r, K, M, &e))) return rc;
    std::vector<uint32_t> info((size_t)Wk, 0u), cw((size_t)Wn);
    memcpy(info.data(), key_words, sizeof(uint32_t) * (size_t)Wkey);
    if (key_bits & 31) info[(size_t)Wkey - 1] &= 0xFFFFFFFFu << (32 - (key_bits & 31));
    HIPCHK(hipMemcpy(e->d_bits, info.data(), sizeof(uint32_t) * (size_t)Wk, hipMemcpyHostToDevice));
    if ((rc = qldpc_encode_packed_dev(e->enc, e->d_bits, e->d_out, 1, nullptr))) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(cw.data(), e->d_out, sizeof(uint32_t) * (size_t)Wn, hipMemcpyDeviceToHost));
    /* parity bits are codeword bits K..N-1; K is word aligned */
    memset(parity_words, 0, sizeof(uint32_t) * (size_t)Wm);
    memcpy(parity_words, cw.data() + Wk, sizeof(uint32_t) * (size_t)(Wn - Wk));
    msg->crc32 = qldpc_crc32_words(key_words, key_bits);
    return QLDPC_OK;
}

/*
 * Alice's side for many blocks at once: plans every block (msgs[i] is filled as by qldpc_recon_encode), groups the blocks by
 * plan and encodes each group in launches of up to max_blocks frames.  parity_words[i] must hold ceil(code_m / 32) words
 * (parity_cap[i] words are available).
 */
extern "C" int qldpc_recon_encode_blocks(qldpc_recon *r, int n, const uint32_t *const *key_words, const int *key_bits, const float *qber,
                                         qldpc_recon_msg *msgs, uint32_t *const *parity_words, const int *parity_cap)
{
    if (!r || !key_words || !key_bits || !qber || !msgs || !parity_words || !parity_cap || n <= 0) return QLDPC_EINVAL;
    int rc;
    for (int i = 0; i < n; i++) {
        if (!key_words[i] || !parity_words[i]) return QLDPC_EINVAL;
        if ((rc = qldpc_recon_plan(r, key_bits[i], qber[i], &msgs[i]))) return rc;
        if (parity_cap[i] < ((int)msgs[i].code_m + 31) / 32) { qldpc_set_error("recon_encode_blocks: parity buffer %d holds %d words, need %d", i, parity_cap[i], ((int)msgs[i].code_m + 31) / 32); return QLDPC_ESIZE; }
    }
    HIPCHK(hipSetDevice(r->cfg.device));
    std::vector<char> taken((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (taken[(size_t)i]) continue;
        std::vector<int> idx;
        for (int j = i; j < n; j++)
            if (!taken[(size_t)j] && msgs[j].code_k == msgs[i].code_k && msgs[j].code_m == msgs[i].code_m) { idx.push_back(j); taken[(size_t)j] = 1; }
        const int K = (int)msgs[i].code_k, M = (int)msgs[i].code_m, N = K + M;
        const int Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32;
        recon_entry *e;
        if ((rc = get_entry(r, K, M, &e))) return rc;
        for (size_t at = 0; at < idx.size(); at += (size_t)r->cfg.max_blocks) {
            const int m = (int)std::min(idx.size() - at, (size_t)r->cfg.max_blocks);
            std::vector<uint32_t> info((size_t)m * Wk, 0u), cw((size_t)m * Wn);
            for (int t = 0; t < m; t++) {
                const int j = idx[at + (size_t)t], kb = key_bits[j], Wkey = (kb + 31) / 32;
                uint32_t *f = info.data() + (size_t)t * Wk;
                memcpy(f, key_words[j], sizeof(uint32_t) * (size_t)Wkey);
                if (kb & 31) f[Wkey - 1] &= 0xFFFFFFFFu << (32 - (kb & 31));
            }
            HIPCHK(hipMemcpy(e->d_bits, info.data(), sizeof(uint32_t) * info.size(), hipMemcpyHostToDevice));
            if ((rc = qldpc_encode_packed_dev(e->enc, e->d_bits, e->d_out, m, nullptr))) return rc;
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(cw.data(), e->d_out, sizeof(uint32_t) * cw.size(), hipMemcpyDeviceToHost));
            for (int t = 0; t < m; t++) {
                const int j = idx[at + (size_t)t];
                memset(parity_words[j], 0, sizeof(uint32_t) * (size_t)Wm);
                memcpy(parity_words[j], cw.data() + (size_t)t * Wn + Wk, sizeof(uint32_t) * (size_t)(Wn - Wk));
                msgs[j].crc32 = qldpc_crc32_words(key_words[j], key_bits[j]);
            }
        }
    }
    return QLDPC_OK;
}

/* blocks of ONE plan (same K, M), possibly of different length: one launch.  key[i] is decoded in place. */
static int decode_group(qldpc_recon *r, int n, uint32_t *const *key, const int *key_bits, const float *qber, const qldpc_recon_msg *const *msgs,
                        const uint32_t *const *parity, int *const *status, int *const *corrected, int *const *iterations)
{
    const int K = (int)msgs[0]->code_k, M = (int)msgs[0]->code_m, N = K + M;
    const int Wk = K / 32, Wn = (N + 31) / 32;
    int rc;
    HIPCHK(hipSetDevice(r->cfg.device));
    recon_entry *e;
    if ((rc = get_entry(r, K, M, &e))) return rc;
    if (e->cls_key_bits != K) {
        std::vector<uint8_t> cls((size_t)N, (uint8_t)QLDPC_VN_PINNED);       /* parity VNs are known; key VNs past a block's length are pinned per frame */
        for (int i = 0; i < K; i++) cls[(size_t)i] = (uint8_t)QLDPC_VN_CHANNEL;
        HIPCHK(hipMemcpy(e->d_cls, cls.data(), (size_t)N, hipMemcpyHostToDevice));
        e->cls_key_bits = K;
    }
    std::vector<uint32_t> frame((size_t)n * Wn, 0u), outw((size_t)n * Wn);
    std::vector<float> mag((size_t)n);
    for (int i = 0; i < n; i++) {
        const int Wkey = (key_bits[i] + 31) / 32;
        uint32_t *f = frame.data() + (size_t)i * Wn;
        memcpy(f, key[i], sizeof(uint32_t) * (size_t)Wkey);
        if (key_bits[i] & 31) f[Wkey - 1] &= 0xFFFFFFFFu << (32 - (key_bits[i] & 31));
        memcpy(f + Wk, parity[i], sizeof(uint32_t) * (size_t)(Wn - Wk));
        mag[(size_t)i] = qldpc_bsc_llr(qber[i]);
    }
    HIPCHK(hipMemcpy(e->d_bits, frame.data(), sizeof(uint32_t) * frame.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_mag, mag.data(), sizeof(float) * mag.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_nch, key_bits, sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    if ((rc = qldpc_load_bits_short_dev(e->dec, e->d_bits, e->d_mag, e->d_cls, e->d_nch, n))) return rc;
    if ((rc = qldpc_run(e->dec))) return rc;
    if ((rc = qldpc_fetch_packed_dev(e->dec, e->d_out))) return rc;
    if ((rc = qldpc_fetch_status_dev(e->dec, e->d_iters, e->d_ok))) return rc;
    if ((rc = qldpc_sync(e->dec))) return rc;
    std::vector<int> it((size_t)n), ok((size_t)n);
    HIPCHK(hipMemcpy(outw.data(), e->d_out, sizeof(uint32_t) * outw.size(), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(it.data(), e->d_iters, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ok.data(), e->d_ok, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        const int kb = key_bits[i], Wkey = (kb + 31) / 32;
        const uint32_t *o = outw.data() + (size_t)i * Wn;
        uint32_t *kw = key[i];
        const bool good = ok[(size_t)i] && qldpc_crc32_words(o, kb) == msgs[i]->crc32;
        *status[i] = good ? QLDPC_OK : QLDPC_EDECODE;
        if (iterations[i]) *iterations[i] = it[(size_t)i];
        int flips = 0;
        if (good) {
            for (int w = 0; w < Wkey; w++) {
                uint32_t nw = o[w];
                if (w == Wkey - 1 && (kb & 31)) nw &= 0xFFFFFFFFu << (32 - (kb & 31));
                uint32_t old = kw[w];
                if (w == Wkey - 1 && (kb & 31)) old &= 0xFFFFFFFFu << (32 - (kb & 31));
                flips += __builtin_popcount(old ^ nw);
                kw[w] = nw;
            }
        }
        if (corrected[i]) *corrected[i] = flips;
    }
    return QLDPC_OK;
}

/*
 * Blocks of any mix of lengths and plans in one call ("let many blocks queue and decode in one
 * launch"): blocks are grouped by plan (code_k, code_m) and every group goes through the decoder in launches of up to
 * max_blocks frames; within a group the blocks may differ in length (their unused key VNs are pinned per frame).
 */
extern "C" int qldpc_recon_decode_blocks(qldpc_recon *r, int n, uint32_t *const *key_words, const int *key_bits, const float *qber,
                                         const qldpc_recon_msg *msgs, const uint32_t *const *parity_words, int *status, int *corrected, int *iterations)
{
    if (!r || !key_words || !key_bits || !qber || !msgs || !parity_words || !status || n <= 0) return QLDPC_EINVAL;
    int rc;
    for (int i = 0; i < n; i++) {
        if (!key_words[i] || !parity_words[i]) return QLDPC_EINVAL;
        if ((rc = check_msg(r, &msgs[i], key_bits[i]))) return rc;
        if (!(qber[i] > 0.0f && qber[i] < 0.5f)) { qldpc_set_error("recon_decode_blocks: qber[%d]=%g", i, (double)qber[i]); return QLDPC_EINVAL; }
        status[i] = QLDPC_EDECODE;
    }
    std::vector<char> taken((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (taken[(size_t)i]) continue;
        std::vector<int> idx;
        for (int j = i; j < n; j++)
            if (!taken[(size_t)j] && msgs[j].code_k == msgs[i].code_k && msgs[j].code_m == msgs[i].code_m) { idx.push_back(j); taken[(size_t)j] = 1; }
        for (size_t at = 0; at < idx.size(); at += (size_t)r->cfg.max_blocks) {
            const int m = (int)std::min(idx.size() - at, (size_t)r->cfg.max_blocks);
            std::vector<uint32_t *> k((size_t)m);
            std::vector<const uint32_t *> p((size_t)m);
            std::vector<const qldpc_recon_msg *> mm((size_t)m);
            std::vector<int> kb((size_t)m);
            std::vector<float> qb((size_t)m);
            std::vector<int *> st((size_t)m), co((size_t)m), itp((size_t)m);
            for (int t = 0; t < m; t++) {
                const int j = idx[at + (size_t)t];
                k[(size_t)t] = key_words[j]; p[(size_t)t] = parity_words[j]; mm[(size_t)t] = &msgs[j]; kb[(size_t)t] = key_bits[j]; qb[(size_t)t] = qber[j];
                st[(size_t)t] = &status[j]; co[(size_t)t] = corrected ? &corrected[j] : nullptr; itp[(size_t)t] = iterations ? &iterations[j] : nullptr;
            }
            if ((rc = decode_group(r, m, k.data(), kb.data(), qb.data(), mm.data(), p.data(), st.data(), co.data(), itp.data()))) return rc;
        }
    }
    return QLDPC_OK;
}

/* n blocks of ONE plan and length, contiguous arrays (the config-3 stream driver's call) */
extern "C" int qldpc_recon_decode_batch(qldpc_recon *r, int n, uint32_t *key_words, int key_bits, const float *qber, const qldpc_recon_msg *msgs,
                                        const uint32_t *parity_words, int *status, int *corrected, int *iterations)
{
    if (!r || !key_words || !qber || !msgs || !parity_words || !status || n <= 0) return QLDPC_EINVAL;
    if (n > r->cfg.max_blocks) { qldpc_set_error("recon_decode_batch: %d blocks > max_blocks %d", n, r->cfg.max_blocks); return QLDPC_ESIZE; }
    for (int i = 0; i < n; i++)
        if (msgs[i].code_k != msgs[0].code_k || msgs[i].code_m != msgs[0].code_m) { qldpc_set_error("recon_decode_batch: block %d has a different plan", i); return QLDPC_ESIZE; }
    const int Wkey = (key_bits + 31) / 32, Wm = ((int)msgs[0].code_m + 31) / 32;
    std::vector<uint32_t *> k((size_t)n);
    std::vector<const uint32_t *> p((size_t)n);
    std::vector<int> kb((size_t)n, key_bits);
    for (int i = 0; i < n; i++) { k[(size_t)i] = key_words + (size_t)i * Wkey; p[(size_t)i] = parity_words + (size_t)i * Wm; }
    return qldpc_recon_decode_blocks(r, n, k.data(), kb.data(), qber, msgs, p.data(), status, corrected, iterations);
}

extern "C" int qldpc_recon_decode(qldpc_recon *r, uint32_t *key_words, int key_bits, float qber, const qldpc_recon_msg *msg, const uint32_t *parity_words,
                                  int *corrected, int *leaked, int *iterations)
{
    if (!r || !msg) return QLDPC_EINVAL;
    int status = QLDPC_EDECODE, corr = 0, it = 0;
    int rc = qldpc_recon_decode_batch(r, 1, key_words, key_bits, &qber, msg, parity_words, &status, &corr, &it);
    if (rc) return rc;
    if (corrected) *corrected = corr;
    if (iterations) *iterations = it;
    if (leaked) *leaked = (int)msg->code_m + 32;      /* disclosed parity bits + the CRC */
    if (status != QLDPC_OK) qldpc_set_error("recon_decode: no verified codeword after %d iterations", it);
    return status;
}
