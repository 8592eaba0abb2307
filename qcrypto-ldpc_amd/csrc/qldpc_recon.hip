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
    uint8_t *d_cls;       /* [N]: key VNs channel, parity VNs pinned */
    uint32_t *d_key;      /* [max_blocks][Wk]  key words as handed in (Bob) / information words (Alice) */
    uint32_t *d_disc;     /* [max_blocks][Wm]  disclosed parity bits, packed */
    uint32_t *d_bits;     /* [max_blocks][Wn]  assembled frames */
    uint32_t *d_erase;    /* [max_blocks][Wn]  per-frame puncturing masks */
    uint32_t *d_out;      /* [max_blocks][Wn] */
    float *d_mag;         /* [max_blocks] */
    int *d_iters, *d_ok;  /* [max_blocks] */
    int *d_nch;           /* [max_blocks] channel VNs per frame (block length) */
    int *d_np;            /* [max_blocks] punctured parity VNs per frame */
    /* one pinned host block and one device block per direction: a decode is ONE copy in and ONE copy out */
    uint32_t *h_in, *d_in;     /* keys [n][Wk] | disclosed parity [n][Wm] | |LLR| [n] | block length [n] | punctured [n] */
    uint32_t *h_res, *d_res;   /* decoded words [n][Wn] | iterations [n] | success [n] */
};

struct qldpc_recon {
    qldpc_recon_cfg cfg;
    float gap;
    std::list<recon_entry> cache;   /* most recently used first */
    size_t keep;                    /* entries the cache may hold (all mother codes when they are preloaded) */
    long created;                   /* entries built so far (tests: nothing is built after a preload) */
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
    c->n_ite = 60;
    c->rule = QLDPC_RULE_SPA;       /* the harness default (BS/src/main.cpp:193); punctured VNs need it: measured FER 0 where NMS fails */
    c->rule_param = 0.0f;
    c->key_quantum = 1024;
    c->max_blocks = 1;
    c->seed = 7;
    c->mother_step = 8192;
    c->mother_max = 65536;
    c->rate_gap = 0.0f;             /* 0 = by rule: 0.03 for SPA, 0.05 for the min-sum family */
    c->puncture = 1;
    c->preload = 0;
}

/*
 * Punctured parity positions of a block: p of the M accumulator bits, evenly spaced (position j is punctured iff
 * floor((j + 1) p / M) > floor(j p / M)), so no check loses both of its parity neighbours while p <= M / 2 and the pattern needs
 * no table or seed: both sides compute it from (M, p).  The reference shuffles the parity positions at random and searches for
 * good patterns (BS/src/main.cpp:305-333); on this accumulator structure random patterns fail where the even one decodes
 * (tools/punct_probe.py: 37 % punctured at QBER 4 %: FER 0.07 random, 0 even; 49 % at 3 %: 1.0 vs 0).
 */
__host__ __device__ static inline int punct_before(int j, int p, int M) { return (int)(((long long)j * p) / M); }      /* punctured positions in [0, j) */
__host__ __device__ static inline bool punct_at(int j, int p, int M) { return punct_before(j + 1, p, M) > punct_before(j, p, M); }

/* Bob: frame words + puncturing mask of every block from its key words and the disclosed parity bits */
__global__ __launch_bounds__(256) void rk_assemble(const uint32_t *__restrict__ key, const uint32_t *__restrict__ disc, const int *__restrict__ key_bits,
                                                   const int *__restrict__ n_punct, uint32_t *__restrict__ bits, uint32_t *__restrict__ erase,
                                                   int Wk, int Wn, int Wm, int M)
{
    const int b = blockIdx.y;
    const int kb = key_bits[b], p = n_punct[b];
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < Wn; w += gridDim.x * blockDim.x) {
        uint32_t word = 0, er = 0;
        if (w < Wk) {
            const int lo = w * 32;
            if (lo < kb) {
                word = key[(size_t)b * Wk + w];
                if (kb - lo < 32) word &= 0xFFFFFFFFu << (32 - (kb - lo));      /* shortened positions are known zeros */
            }
        } else {
            const uint32_t *dw = disc + (size_t)b * Wm;
            for (int t = 0; t < 32; t++) {
                const int j = (w - Wk) * 32 + t;
                if (j >= M) break;
                if (punct_at(j, p, M)) er |= 1u << (31 - t);
                else {
                    const int r = j - punct_before(j, p, M);
                    word |= ((dw[r >> 5] >> (31 - (r & 31))) & 1u) << (31 - t);
                }
            }
        }
        bits[(size_t)b * Wn + w] = word;
        erase[(size_t)b * Wn + w] = er;
    }
}

/* Alice: the disclosed (non-punctured) parity bits of every codeword, packed in position order */
__global__ __launch_bounds__(256) void rk_disclose(const uint32_t *__restrict__ cw, const int *__restrict__ n_punct, uint32_t *__restrict__ disc,
                                                   int Wk, int Wn, int Wm, int M)
{
    const int b = blockIdx.y;
    const int p = n_punct[b], d = M - p;
    const uint32_t *par = cw + (size_t)b * Wn + Wk;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < Wm; w += gridDim.x * blockDim.x) {
        uint32_t word = 0;
        for (int t = 0; t < 32; t++) {
            const int r = w * 32 + t;
            if (r >= d) break;
            /* the r-th disclosed position: smallest j with (j + 1) - punct_before(j + 1) == r + 1 */
            int lo = r, hi = r + p;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (mid + 1 - punct_before(mid + 1, p, M) >= r + 1) hi = mid; else lo = mid + 1;
            }
            word |= ((par[lo >> 5] >> (31 - (lo & 31))) & 1u) << (31 - t);
        }
        disc[(size_t)b * Wm + w] = word;
    }
}

static void entry_free(recon_entry &e)
{
    qldpc_decoder_free(e.dec);
    qldpc_encoder_free(e.enc);
    qldpc_code_free(e.code);
    (void)hipFree(e.d_cls); (void)hipFree(e.d_key); (void)hipFree(e.d_disc); (void)hipFree(e.d_bits); (void)hipFree(e.d_erase); (void)hipFree(e.d_out);
    (void)hipFree(e.d_mag); (void)hipFree(e.d_iters); (void)hipFree(e.d_ok); (void)hipFree(e.d_nch); (void)hipFree(e.d_np);
    (void)hipFree(e.d_in); (void)hipFree(e.d_res);
    if (e.h_in) (void)hipHostFree(e.h_in);
    if (e.h_res) (void)hipHostFree(e.h_res);
}

extern "C" void qldpc_recon_free(qldpc_recon *r)
{
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    for (auto &e : r->cache) entry_free(e);
    delete r;
}

/* code dimensions of a block of key_bits on table rate R: a mother code (K a multiple of mother_step, the block is shortened) up to
 * mother_max bits, otherwise a code of its own size (K = key_bits rounded up to key_quantum) */
static void code_dims(const qldpc_recon_cfg &c, int key_bits, double R, int *K, int *M)
{
    int k;
    if (c.mother_step > 0 && key_bits <= c.mother_max) k = (key_bits + c.mother_step - 1) / c.mother_step * c.mother_step;
    else k = (key_bits + c.key_quantum - 1) / c.key_quantum * c.key_quantum;
    int m = (int)llround((double)k * (1.0 - R) / R);
    if (m < 2) m = 2;
    *K = k; *M = m;
}

static int get_entry(qldpc_recon *r, int K, int M, recon_entry **out);

extern "C" int qldpc_recon_create(const qldpc_recon_cfg *cfg, qldpc_recon **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = nullptr;
    if (!cfg) return QLDPC_EINVAL;
    if (cfg->n_rates < 1 || cfg->n_rates > 8 || cfg->key_quantum < 32 || (cfg->key_quantum & 31) || cfg->max_blocks < 1 || cfg->n_ite < 1 ||
        !(cfg->efficiency > 0.0f) || cfg->mother_step < 0 || (cfg->mother_step & 31) || (cfg->mother_step > 0 && cfg->mother_max < cfg->mother_step) ||
        !(cfg->rate_gap >= 0.0f && cfg->rate_gap < 0.5f) || cfg->reserved[0] || cfg->reserved[1]) {
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
    r->gap = cfg->rate_gap > 0.0f ? cfg->rate_gap : (cfg->rule == QLDPC_RULE_SPA || cfg->rule == QLDPC_RULE_LSPA ? 0.035f : 0.05f);
    const size_t mothers = cfg->mother_step > 0 ? (size_t)(cfg->mother_max / cfg->mother_step) * (size_t)cfg->n_rates : 0;
    r->keep = mothers + 6;
    r->created = 0;
    if (cfg->preload && mothers) {
        /* every (mother size, table rate) pair now: code, encoder, decoder and staging buffers, so that no block of up to mother_max
         * bits builds a code or allocates device memory later (the daemon calls this from ldpc_init) */
        if (hipSetDevice(cfg->device) != hipSuccess) { delete r; return QLDPC_EHIP; }
        for (int k = cfg->mother_step; k <= cfg->mother_max; k += cfg->mother_step)
            for (int i = 0; i < cfg->n_rates; i++) {
                int K, M;
                code_dims(r->cfg, k, cfg->rates[i], &K, &M);
                recon_entry *e;
                const int rc = get_entry(r, K, M, &e);
                if (rc) { qldpc_recon_free(r); return rc; }
            }
    }
    *out = r;
    return QLDPC_OK;
}

extern "C" long qldpc_recon_entries_created(const qldpc_recon *r) { return r ? r->created : -1; }

/* per-kernel profile of Bob's decoders (qldpc_profile_*), summed over the codes of the session */
extern "C" int qldpc_recon_profile_enable(qldpc_recon *r, int on)
{
    if (!r) return QLDPC_EINVAL;
    for (auto &e : r->cache) { qldpc_profile_enable(e.dec, on); if (on) qldpc_profile_clear(e.dec); }
    return QLDPC_OK;
}
extern "C" int qldpc_recon_profile_read(qldpc_recon *r, qldpc_kernel_stat *out, int cap)
{
    if (!r || (!out && cap > 0)) return QLDPC_EINVAL;
    int n = 0;
    for (auto &e : r->cache) {
        qldpc_kernel_stat st[16];
        const int m = qldpc_profile_read(e.dec, st, 16);
        if (m < 0) return m;
        for (int i = 0; i < m; i++) {
            int k = 0;
            while (k < n && strcmp(out[k].name, st[i].name)) k++;
            if (k == n) { if (n >= cap) continue; out[n] = st[i]; n++; continue; }
            out[k].launches += st[i].launches; out[k].total_ms += st[i].total_ms; out[k].alg_bytes += st[i].alg_bytes; out[k].moved_bytes += st[i].moved_bytes;
        }
    }
    return n;
}

/*
 * Rate choice, code dimensions and puncturing of a block (BS/src/main.cpp:29-34,235-311).  The QBER estimate is clamped to
 * [0.001, 0.25] first: the daemon's localError is exactly 0 when the test sample held no error (qber_estim.c:26).
 *   target rate  R* = min( 1 / (1 + f h(q)),  1 - h(q) - rate_gap (65536 / K)^0.4 c(R) )   the harness's min_cr, kept away from capacity;
 *                c(R) = 0.6 for mother rates <= 0.75, 0.9 up to 0.85, 1 above
 *   table rate   R  = largest entry <= R*                                  (BS/src/main.cpp:241-266)
 *   code         K >= key_bits information VNs (mother code, shortened), M = round(K (1 - R) / R) parity VNs
 *   disclosed    d = ceil(key_bits (1 / R* - 1)) parity bits, the other p = M - d are punctured (parity_bits_to_punct with the block's
 *                own length for the information bits), at most 65 % of M
 * leak = d + 32 (CRC).
 */
#define RECON_PUNCT_CAP 0.65
static float clamp_qber(float q) { return !(q > 0.001f) ? 0.001f : (q > 0.25f ? 0.25f : q); }

extern "C" int qldpc_recon_plan(const qldpc_recon *r, int key_bits, float qber, qldpc_recon_msg *msg)
{
    if (!r || !msg) return QLDPC_EINVAL;
    if (key_bits < 32) { qldpc_set_error("recon_plan: key_bits=%d", key_bits); return QLDPC_ESIZE; }
    if (!(qber >= 0.0f && qber < 0.5f)) { qldpc_set_error("recon_plan: qber=%g not in [0, 0.5)", (double)qber); return QLDPC_EINVAL; }
    const float q = clamp_qber(qber);
    const double h = qldpc_binary_entropy(q);
    /* the distance a code needs from capacity grows as it gets shorter: measured clean (FER 0 / 128 per point, QBER 0.3 .. 8 %, SPA) at
     * 0.035 for K = 65 536, 0.042 for 32 768, 0.06 for 16 384 and 8 192 (tools/punct_probe.py) -- rate_gap (65536 / K)^0.4 covers them */
    int K0, M0;
    code_dims(r->cfg, key_bits, 0.5, &K0, &M0);
    const double gap_len = (double)r->gap * pow(65536.0 / (double)K0, 0.4);
    /* ... and with the rate of the mother code: the low-rate mothers decode punctured to within 0.02 of capacity (FER 0 / 128 at 0.015 -
     * 0.02 for rates 0.5 and 0.7), rate 0.8 needs 0.03 and rate 0.9 0.035 (tools/punct_probe.py, gpurun logs punct_c / punct_e): the
     * table entry is the highest rate that fits under ITS OWN target */
    int idx = -1;
    double need = 0.0;
    for (int i = 0; i < r->cfg.n_rates; i++) {
        const double R = r->cfg.rates[i];
        const double gap = gap_len * (R <= 0.75 ? 0.6 : (R <= 0.85 ? 0.9 : 1.0));
        double t = qldpc_min_code_rate(q, r->cfg.efficiency);
        if (1.0 - h - gap < t) t = 1.0 - h - gap;
        if (R <= t) { idx = i; need = t; }
    }
    if (idx < 0) { qldpc_set_error("recon_plan: QBER %.4f is above what the rate table covers", (double)qber); return QLDPC_EUNSUPPORTED; }
    int K, M;
    code_dims(r->cfg, key_bits, r->cfg.rates[idx], &K, &M);
    int p = 0;
    if (r->cfg.puncture == 1) {
        const int d = (int)ceil((double)key_bits * (1.0 / need - 1.0));
        p = M - d;
        if (p > (int)(RECON_PUNCT_CAP * M)) p = (int)(RECON_PUNCT_CAP * M);
        if (p < 0) p = 0;
    }
    memset(msg, 0, sizeof(*msg));
    msg->rate_index = (uint32_t)idx;
    msg->key_bits = (uint32_t)key_bits;
    msg->code_k = (uint32_t)K;
    msg->code_m = (uint32_t)M;
    msg->n_punct = (uint32_t)p;
    return QLDPC_OK;
}

static int get_entry(qldpc_recon *r, int K, int M, recon_entry **out)
{
    for (auto it = r->cache.begin(); it != r->cache.end(); ++it)
        if (it->K == K && it->M == M) { r->cache.splice(r->cache.begin(), r->cache, it); *out = &r->cache.front(); return QLDPC_OK; }
    recon_entry e;
    memset(&e, 0, sizeof(e));
    e.K = K; e.M = M;
    const int N = K + M, Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32, B = r->cfg.max_blocks;
    int rc = qldpc_code_ira(N, K, 0.125f, 11, 3, r->cfg.seed, &e.code);
    if (!rc) rc = qldpc_encoder_create(e.code, "IRA", r->cfg.device, &e.enc);
    if (!rc) {
        qldpc_decoder_cfg dc;
        qldpc_decoder_cfg_default(&dc);
        dc.schedule = r->cfg.schedule == QLDPC_SCHED_HLAYERED ? QLDPC_SCHED_HLAYERED : QLDPC_SCHED_FLOODING; dc.rule = r->cfg.rule; dc.rule_param = r->cfg.rule_param; dc.n_ite = r->cfg.n_ite;
        dc.enable_syndrome = 1; dc.syndrome_depth = 1; dc.max_frames = B; dc.device = r->cfg.device;
        rc = qldpc_decoder_create(e.code, K, nullptr, &dc, &e.dec);
        if (!rc) rc = qldpc_decoder_reserve(e.dec);      /* nothing is allocated per block later */
    }
    auto alloc = [&](void **ptr, size_t bytes) { if (!rc && hipMalloc(ptr, bytes) != hipSuccess) rc = QLDPC_ENOMEM; };
    alloc((void **)&e.d_cls, (size_t)N);
    alloc((void **)&e.d_key, sizeof(uint32_t) * (size_t)B * Wk);
    alloc((void **)&e.d_disc, sizeof(uint32_t) * (size_t)B * Wm);
    alloc((void **)&e.d_bits, sizeof(uint32_t) * (size_t)B * Wn);
    alloc((void **)&e.d_erase, sizeof(uint32_t) * (size_t)B * Wn);
    alloc((void **)&e.d_out, sizeof(uint32_t) * (size_t)B * Wn);
    alloc((void **)&e.d_mag, sizeof(float) * (size_t)B);
    alloc((void **)&e.d_iters, sizeof(int) * (size_t)B);
    alloc((void **)&e.d_ok, sizeof(int) * (size_t)B);
    alloc((void **)&e.d_nch, sizeof(int) * (size_t)B);
    alloc((void **)&e.d_np, sizeof(int) * (size_t)B);
    alloc((void **)&e.d_in, sizeof(uint32_t) * (size_t)B * (Wk + Wm + 3));
    alloc((void **)&e.d_res, sizeof(uint32_t) * (size_t)B * (Wn + 2));
    if (!rc && hipHostMalloc((void **)&e.h_in, sizeof(uint32_t) * (size_t)B * (Wk + Wm + 3)) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc && hipHostMalloc((void **)&e.h_res, sizeof(uint32_t) * (size_t)B * (Wn + 2)) != hipSuccess) rc = QLDPC_ENOMEM;
    if (!rc) {
        std::vector<uint8_t> cls((size_t)N, (uint8_t)QLDPC_VN_PINNED);       /* parity VNs are disclosed; key VNs past a block's length are pinned per frame */
        for (int i = 0; i < K; i++) cls[(size_t)i] = (uint8_t)QLDPC_VN_CHANNEL;
        if (hipMemcpy(e.d_cls, cls.data(), (size_t)N, hipMemcpyHostToDevice) != hipSuccess) rc = QLDPC_EHIP;
    }
    if (rc) { entry_free(e); return rc; }
    while (r->cache.size() >= r->keep) { entry_free(r->cache.back()); r->cache.pop_back(); }
    r->cache.push_front(e);
    r->created++;
    *out = &r->cache.front();
    return QLDPC_OK;
}

/* a message header must describe the block it came with AND be what the local plan gives for its rate index: the peer cannot make
 * this side build a code of its own choosing */
static int check_msg(const qldpc_recon *r, const qldpc_recon_msg *m, int key_bits)
{
    int K = 0, M = 0;
    const bool idx_ok = m->rate_index < (uint32_t)r->cfg.n_rates;
    if (idx_ok && key_bits >= 32) code_dims(r->cfg, key_bits, r->cfg.rates[m->rate_index], &K, &M);
    if (!idx_ok || key_bits < 32 || (int)m->key_bits != key_bits || (int)m->code_k != K || (int)m->code_m != M || m->n_punct > (uint32_t)(RECON_PUNCT_CAP * M) ||
        (r->cfg.puncture != 1 && m->n_punct != 0)) {
        qldpc_set_error("recon: message header does not match the block (key_bits %u vs %d, rate index %u, K %u vs %d, M %u vs %d, punctured %u)",
                        m->key_bits, key_bits, m->rate_index, m->code_k, K, m->code_m, M, m->n_punct);
        return QLDPC_ESIZE;
    }
    return QLDPC_OK;
}

/* QLDPC_OK if the header is what this side's plan gives for its rate index and the block's length (what the decode calls check per
 * message): lets a packet handler refuse a header before it allocates for its payload */
extern "C" int qldpc_recon_check_header(const qldpc_recon *r, const qldpc_recon_msg *msg, int key_bits)
{
    if (!r || !msg) return QLDPC_EINVAL;
    return check_msg(r, msg, key_bits);
}

/* blocks of ONE entry (same K, M): one encoder launch.  parity[i] receives ceil((M - n_punct) / 32) words. */
static int encode_group(qldpc_recon *r, int n, const uint32_t *const *key, const int *key_bits, qldpc_recon_msg *const *msgs, uint32_t *const *parity)
{
    const int K = (int)msgs[0]->code_k, M = (int)msgs[0]->code_m, N = K + M;
    const int Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32;
    int rc;
    recon_entry *e;
    if ((rc = get_entry(r, K, M, &e))) return rc;
    std::vector<uint32_t> info((size_t)n * Wk, 0u), disc((size_t)n * Wm);
    std::vector<int> np((size_t)n);
    for (int t = 0; t < n; t++) {
        const int kb = key_bits[t], Wkey = (kb + 31) / 32;
        uint32_t *f = info.data() + (size_t)t * Wk;
        memcpy(f, key[t], sizeof(uint32_t) * (size_t)Wkey);
        if (kb & 31) f[Wkey - 1] &= 0xFFFFFFFFu << (32 - (kb & 31));
        np[(size_t)t] = (int)msgs[t]->n_punct;
    }
    HIPCHK(hipMemcpy(e->d_key, info.data(), sizeof(uint32_t) * info.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_np, np.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    if ((rc = qldpc_encode_packed_dev(e->enc, e->d_key, e->d_out, n, nullptr))) return rc;
    hipLaunchKernelGGL(rk_disclose, dim3((unsigned)((Wm + 255) / 256), (unsigned)n), dim3(256), 0, 0, e->d_out, e->d_np, e->d_disc, Wk, Wn, Wm, M);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(disc.data(), e->d_disc, sizeof(uint32_t) * disc.size(), hipMemcpyDeviceToHost));
    for (int t = 0; t < n; t++) {
        const int Wd = (M - np[(size_t)t] + 31) / 32;
        memcpy(parity[t], disc.data() + (size_t)t * Wm, sizeof(uint32_t) * (size_t)Wd);
        msgs[t]->crc32 = qldpc_crc32_words(key[t], key_bits[t]);
    }
    return QLDPC_OK;
}

extern "C" int qldpc_recon_parity_words(const qldpc_recon_msg *msg) { return msg ? (int)((msg->code_m - msg->n_punct + 31) / 32) : QLDPC_EINVAL; }
extern "C" int qldpc_recon_leaked_bits(const qldpc_recon_msg *msg) { return msg ? (int)(msg->code_m - msg->n_punct) + 32 : QLDPC_EINVAL; }

/*
 * Alice's side for many blocks at once: plans every block (msgs[i] is filled as by qldpc_recon_encode), groups the blocks by
 * code and encodes each group in launches of up to max_blocks frames.  parity_words[i] receives qldpc_recon_parity_words(&msgs[i])
 * words (parity_cap[i] are available).
 */
extern "C" int qldpc_recon_encode_blocks(qldpc_recon *r, int n, const uint32_t *const *key_words, const int *key_bits, const float *qber,
                                         qldpc_recon_msg *msgs, uint32_t *const *parity_words, const int *parity_cap)
{
    if (!r || !key_words || !key_bits || !qber || !msgs || !parity_words || !parity_cap || n <= 0) return QLDPC_EINVAL;
    int rc;
    for (int i = 0; i < n; i++) {
        if (!key_words[i] || !parity_words[i]) return QLDPC_EINVAL;
        if ((rc = qldpc_recon_plan(r, key_bits[i], qber[i], &msgs[i]))) return rc;
        if (parity_cap[i] < qldpc_recon_parity_words(&msgs[i])) { qldpc_set_error("recon_encode_blocks: parity buffer %d holds %d words, need %d", i, parity_cap[i], qldpc_recon_parity_words(&msgs[i])); return QLDPC_ESIZE; }
    }
    HIPCHK(hipSetDevice(r->cfg.device));
    std::vector<char> taken((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (taken[(size_t)i]) continue;
        std::vector<int> idx;
        for (int j = i; j < n; j++)
            if (!taken[(size_t)j] && msgs[j].code_k == msgs[i].code_k && msgs[j].code_m == msgs[i].code_m) { idx.push_back(j); taken[(size_t)j] = 1; }
        for (size_t at = 0; at < idx.size(); at += (size_t)r->cfg.max_blocks) {
            const int m = (int)std::min(idx.size() - at, (size_t)r->cfg.max_blocks);
            std::vector<const uint32_t *> k((size_t)m);
            std::vector<uint32_t *> p((size_t)m);
            std::vector<qldpc_recon_msg *> mm((size_t)m);
            std::vector<int> kb((size_t)m);
            for (int t = 0; t < m; t++) { const int j = idx[at + (size_t)t]; k[(size_t)t] = key_words[j]; p[(size_t)t] = parity_words[j]; mm[(size_t)t] = &msgs[j]; kb[(size_t)t] = key_bits[j]; }
            if ((rc = encode_group(r, m, k.data(), kb.data(), mm.data(), p.data()))) return rc;
        }
    }
    return QLDPC_OK;
}

extern "C" int qldpc_recon_encode(qldpc_recon *r, const uint32_t *key_words, int key_bits, float qber, qldpc_recon_msg *msg, uint32_t *parity_words, int cap)
{
    if (!r || !key_words || !msg || !parity_words) return QLDPC_EINVAL;
    const uint32_t *k = key_words;
    uint32_t *p = parity_words;
    return qldpc_recon_encode_blocks(r, 1, &k, &key_bits, &qber, msg, &p, &cap);
}

/*
 * Alice, second round: the parity bits of a plan that is already on the table (msg as qldpc_recon_plan / qldpc_recon_encode left it, n_punct
 * possibly lowered by the caller -- 0 = every parity bit of the mother code).  After a failed decode the bits withheld by puncturing are
 * the cheapest thing to send next (incremental redundancy): the same codeword, a lower effective rate.  msg->crc32 is (re)written.
 */
extern "C" int qldpc_recon_encode_planned(qldpc_recon *r, const uint32_t *key_words, int key_bits, qldpc_recon_msg *msg, uint32_t *parity_words, int cap)
{
    if (!r || !key_words || !msg || !parity_words) return QLDPC_EINVAL;
    int rc;
    if ((rc = check_msg(r, msg, key_bits))) return rc;
    if (cap < qldpc_recon_parity_words(msg)) { qldpc_set_error("recon_encode_planned: parity buffer holds %d words, need %d", cap, qldpc_recon_parity_words(msg)); return QLDPC_ESIZE; }
    HIPCHK(hipSetDevice(r->cfg.device));
    return encode_group(r, 1, &key_words, &key_bits, &msg, &parity_words);
}

/* blocks of ONE entry (same K, M), possibly of different length and puncturing: one launch.  key[i] is decoded in place. */
static int decode_group(qldpc_recon *r, int n, uint32_t *const *key, const int *key_bits, const float *qber, const qldpc_recon_msg *const *msgs,
                        const uint32_t *const *parity, int *const *status, int *const *corrected, int *const *iterations)
{
    const int K = (int)msgs[0]->code_k, M = (int)msgs[0]->code_m, N = K + M;
    const int Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32;
    int rc;
    HIPCHK(hipSetDevice(r->cfg.device));
    recon_entry *e;
    if ((rc = get_entry(r, K, M, &e))) return rc;
    /* staging: everything Bob's side needs, packed into the entry's pinned block -> one copy; the results come back the same way */
    uint32_t *h_keys = e->h_in, *h_disc = h_keys + (size_t)n * Wk;
    float *h_mag = reinterpret_cast<float *>(h_disc + (size_t)n * Wm);
    int *h_nch = reinterpret_cast<int *>(h_mag + n), *h_np = h_nch + n;
    uint32_t *d_keys = e->d_in, *d_disc = d_keys + (size_t)n * Wk;
    float *d_mag = reinterpret_cast<float *>(d_disc + (size_t)n * Wm);
    int *d_nch = reinterpret_cast<int *>(d_mag + n), *d_np = d_nch + n;
    memset(h_keys, 0, sizeof(uint32_t) * (size_t)n * (Wk + Wm));
    bool any_punct = false;
    for (int i = 0; i < n; i++) {
        const int Wkey = (key_bits[i] + 31) / 32;
        memcpy(h_keys + (size_t)i * Wk, key[i], sizeof(uint32_t) * (size_t)Wkey);
        h_np[i] = (int)msgs[i]->n_punct;
        h_nch[i] = key_bits[i];
        any_punct = any_punct || h_np[i] != 0;
        memcpy(h_disc + (size_t)i * Wm, parity[i], sizeof(uint32_t) * (size_t)((M - h_np[i] + 31) / 32));
        h_mag[i] = qldpc_bsc_llr(clamp_qber(qber[i]));
    }
    HIPCHK(hipMemcpyAsync(e->d_in, e->h_in, sizeof(uint32_t) * (size_t)n * (Wk + Wm + 3), hipMemcpyHostToDevice, 0));
    hipLaunchKernelGGL(rk_assemble, dim3((unsigned)((Wn + 255) / 256), (unsigned)n), dim3(256), 0, 0, d_keys, d_disc, d_nch, d_np, e->d_bits, e->d_erase, Wk, Wn, Wm, M);
    HIPCHK(hipGetLastError());
    if ((rc = qldpc_load_bits_short_dev(e->dec, e->d_bits, d_mag, e->d_cls, d_nch, n))) return rc;
    if (any_punct && (rc = qldpc_load_erasures_dev(e->dec, e->d_erase, n))) return rc;
    if ((rc = qldpc_run(e->dec))) return rc;
    uint32_t *d_outw = e->d_res;
    int *d_it = reinterpret_cast<int *>(d_outw + (size_t)n * Wn), *d_okf = d_it + n;
    if ((rc = qldpc_fetch_packed_dev(e->dec, d_outw))) return rc;
    if ((rc = qldpc_fetch_status_dev(e->dec, d_it, d_okf))) return rc;
    HIPCHK(hipMemcpyAsync(e->h_res, e->d_res, sizeof(uint32_t) * (size_t)n * (Wn + 2), hipMemcpyDeviceToHost, 0));
    if ((rc = qldpc_sync(e->dec))) return rc;
    const uint32_t *outw_p = e->h_res;
    const int *it = reinterpret_cast<const int *>(outw_p + (size_t)n * Wn), *ok = it + n;
    for (int i = 0; i < n; i++) {
        const int kb = key_bits[i], Wkey = (kb + 31) / 32;
        const uint32_t *o = outw_p + (size_t)i * Wn;
        uint32_t *kw = key[i];
        const bool good = ok[i] && qldpc_crc32_words(o, kb) == msgs[i]->crc32;
        *status[i] = good ? QLDPC_OK : QLDPC_EDECODE;
        if (iterations[i]) *iterations[i] = it[i];
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
 * Blocks of any mix of lengths, rates and puncturing in one call ("let many blocks queue and decode in one
 * launch"): blocks are grouped by code (code_k, code_m) and every group goes through the decoder in launches of up to max_blocks
 * frames; within a group the blocks may differ in length (shortening per frame) and in efficiency (puncturing per frame).
 * Every message is validated on its own: status[i] = QLDPC_OK | QLDPC_EDECODE | QLDPC_ESIZE (header does not match the block /
 * the local plan -- that block is not decoded, the others are).
 */
extern "C" int qldpc_recon_decode_blocks(qldpc_recon *r, int n, uint32_t *const *key_words, const int *key_bits, const float *qber,
                                         const qldpc_recon_msg *msgs, const uint32_t *const *parity_words, int *status, int *corrected, int *iterations)
{
    if (!r || !key_words || !key_bits || !qber || !msgs || !parity_words || !status || n <= 0) return QLDPC_EINVAL;
    int rc;
    std::vector<char> taken((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (!key_words[i] || !parity_words[i]) return QLDPC_EINVAL;
        status[i] = QLDPC_EDECODE;
        if (corrected) corrected[i] = 0;
        if (iterations) iterations[i] = 0;
        if (check_msg(r, &msgs[i], key_bits[i]) || !(qber[i] >= 0.0f && qber[i] < 0.5f)) { status[i] = QLDPC_ESIZE; taken[(size_t)i] = 1; }
    }
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

/* n blocks of ONE length, contiguous arrays (the config-3 stream driver's call); parity_words rows are ceil(code_m / 32) words apart */
extern "C" int qldpc_recon_decode_batch(qldpc_recon *r, int n, uint32_t *key_words, int key_bits, const float *qber, const qldpc_recon_msg *msgs,
                                        const uint32_t *parity_words, int *status, int *corrected, int *iterations)
{
    if (!r || !key_words || !qber || !msgs || !parity_words || !status || n <= 0) return QLDPC_EINVAL;
    for (int i = 0; i < n; i++)
        if (msgs[i].code_k != msgs[0].code_k || msgs[i].code_m != msgs[0].code_m) { qldpc_set_error("recon_decode_batch: block %d has a different code", i); return QLDPC_ESIZE; }
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
    if (leaked) *leaked = qldpc_recon_leaked_bits(msg);      /* disclosed parity bits + the CRC */
    if (status == QLDPC_EDECODE) qldpc_set_error("recon_decode: no verified codeword after %d iterations", it);
    return status;
}
