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
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <new>
#include <string>
#include <thread>
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

/*
 * One of the two staging sets of an entry: while batch i decodes, batch i + 1 is copied into the other set's pinned block, sent to
 * the device and assembled into frames on the lane's copy stream, and batch i - 1's results travel back.
 */
struct stage_set {
    uint32_t *h_in, *d_in;     /* keys [n][Wk] | disclosed parity [n][Wm] | |LLR| [n] | block length [n] | punctured [n] | Alice's CRC [n] */
    uint32_t *h_res, *d_res;   /* verified key words [n][Wk] | status [n] | bits flipped [n] | iterations [n] | CRC found [n] */
    uint32_t *d_bits;          /* [max_blocks][Wn]  assembled frames */
    uint32_t *d_erase;         /* [max_blocks][Wn]  per-frame puncturing masks */
    hipEvent_t ev_in, ev_out, ev_done;
};

struct recon_entry {
    int K, M;
    qldpc_code *code;
    qldpc_encoder *enc;
    qldpc_decoder *dec;
    uint8_t *d_cls;       /* [N]: key VNs channel, parity VNs pinned */
    uint32_t *d_out;      /* [max_blocks][Wn] decoded words / Alice's codewords */
    int *d_iters, *d_ok;  /* [max_blocks] */
    stage_set set[2];
    int next_set;
};

/* a lane = one host worker with a compute stream and a copy stream: the rate groups of a call run side by side, one lane each */
#define RECON_LANES 4
struct recon_lane {
    hipStream_t stream, copy;
};

struct qldpc_recon {
    qldpc_recon_cfg cfg;
    float gap;
    std::list<recon_entry> cache;   /* most recently used first */
    size_t keep;                    /* entries the cache may hold (all mother codes when they are preloaded) */
    long created;                   /* entries built so far (tests: nothing is built after a preload) */
    recon_lane lane[RECON_LANES];
    int profiling;
};

#define CRC_POLY 0xEDB88320u      /* CRC-32 (IEEE 802.3), reflected */

static const uint32_t *crc_table()
{
    /* built once behind the compiler's guard of a function-local static: the lanes' host workers may get here together */
    struct table {
        uint32_t t[256];
        table()
        {
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t c = i;
                for (int k = 0; k < 8; k++) c = (c & 1) ? CRC_POLY ^ (c >> 1) : c >> 1;
                t[i] = c;
            }
        }
    };
    static const table tab;
    return tab.t;
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

/*
 * The same CRC computed in parallel: the register is linear over GF(2) in (start value, message), so with start value 0 the
 * register after A || B is reg(A) x^|B| + reg(B) (polynomials mod the CRC polynomial) and the start value 0xFFFFFFFF adds
 * 0xFFFFFFFF x^|message|.  Every lane runs the byte-wise recurrence over its own chunk (chunks of equal length; zero words in
 * front of the message leave a zero register alone), the partial registers are folded in a tree -- the same jump-ahead idea the
 * privacy-amplification kernel uses for the LFSR -- and the two constants are applied at the end.  In the reflected form bit 31
 * is the coefficient of x^0 and multiplying by x is a right shift.
 */
__host__ __device__ static inline uint32_t gf_mulmod(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : (b >> 1);
    }
    return p;
}
/* x^n mod the CRC polynomial */
__host__ __device__ static inline uint32_t gf_xpow(unsigned long long n)
{
    uint32_t r = 0x80000000u, sq = 0x40000000u;      /* 1, x */
    while (n) {
        if (n & 1ull) r = gf_mulmod(r, sq);
        sq = gf_mulmod(sq, sq);
        n >>= 1;
    }
    return r;
}
__host__ __device__ static inline uint32_t crc_word_step(uint32_t c, uint32_t x, const uint32_t *t)
{
    for (int b = 3; b >= 0; b--) c = t[(c ^ (x >> (8 * b))) & 0xFF] ^ (c >> 8);
    return c;
}
__host__ __device__ static inline uint32_t tail_mask(int i, int nw, int n_bits)
{
    return (i == nw - 1 && (n_bits & 31)) ? 0xFFFFFFFFu << (32 - (n_bits & 31)) : 0xFFFFFFFFu;
}

/* host mirror of the device verification (rk_verify / rk_crc): `lanes` chunks folded in a tree; equals qldpc_crc32_words */
extern "C" uint32_t qldpc_crc32_words_chunked(const uint32_t *w, int n_bits, int lanes)
{
    if (!w || n_bits <= 0 || lanes < 1 || (lanes & (lanes - 1))) return 0;
    const uint32_t *t = crc_table();
    const int nw = (n_bits + 31) / 32;
    const int c = (nw + lanes - 1) / lanes, pad = c * lanes - nw;
    std::vector<uint32_t> reg((size_t)lanes, 0u);
    for (int l = 0; l < lanes; l++)
        for (int v = l * c; v < (l + 1) * c; v++) {
            const int i = v - pad;
            if (i >= 0) reg[(size_t)l] = crc_word_step(reg[(size_t)l], w[i] & tail_mask(i, nw, n_bits), t);
        }
    uint32_t X = gf_xpow(32ull * (unsigned long long)c);
    for (int s = 1; s < lanes; s <<= 1) {
        for (int l = 0; l + s < lanes; l += 2 * s) reg[(size_t)l] = gf_mulmod(reg[(size_t)l], X) ^ reg[(size_t)(l + s)];
        X = gf_mulmod(X, X);
    }
    return reg[0] ^ gf_mulmod(0xFFFFFFFFu, gf_xpow(32ull * (unsigned long long)nw)) ^ 0xFFFFFFFFu;
}

/* one workgroup of 256 lanes per block: CRC-32 of words[0 .. ceil(n_bits / 32)) (tail bits masked) by the chunked fold; with `old` also
 * the number of key bits in which words differs from old, and with `copy_to` the masked words are written out.  Every lane returns the CRC. */
#define RK_LANES 256
__device__ static uint32_t rk_block_crc(const uint32_t *__restrict__ words, int n_bits, const uint32_t *__restrict__ old, uint32_t *__restrict__ copy_to,
                                        int *flips_out, uint32_t *s_tab, uint32_t *s_reg, int *s_cnt)
{
    const int t = (int)threadIdx.x;
    {
        uint32_t c = (uint32_t)t;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? CRC_POLY ^ (c >> 1) : c >> 1;
        s_tab[t] = c;
    }
    __syncthreads();
    const int nw = (n_bits + 31) / 32;
    const int c = (nw + RK_LANES - 1) / RK_LANES, pad = c * RK_LANES - nw;
    uint32_t reg = 0;
    int flips = 0;
    for (int v = t * c; v < (t + 1) * c; v++) {
        const int i = v - pad;
        if (i < 0) continue;
        const uint32_t m = tail_mask(i, nw, n_bits);
        const uint32_t x = words[i] & m;
        reg = crc_word_step(reg, x, s_tab);
        if (old) flips += __popc((old[i] & m) ^ x);
        if (copy_to) copy_to[i] = x;
    }
    s_reg[t] = reg;
    s_cnt[t] = flips;
    __syncthreads();
    uint32_t X = gf_xpow(32ull * (unsigned long long)c);
    for (int s = 1; s < RK_LANES; s <<= 1) {
        if ((t & (2 * s - 1)) == 0) {
            s_reg[t] = gf_mulmod(s_reg[t], X) ^ s_reg[t + s];
            s_cnt[t] += s_cnt[t + s];
        }
        X = gf_mulmod(X, X);
        __syncthreads();
    }
    if (flips_out) *flips_out = s_cnt[0];
    return s_reg[0] ^ gf_mulmod(0xFFFFFFFFu, gf_xpow(32ull * (unsigned long long)nw)) ^ 0xFFFFFFFFu;
}

/*
 * Bob, after the decode: verification on the device.  Per block: CRC-32 of the decoded key bits against Alice's, the number of bits the
 * decode flipped, and the verified key words (tail masked) in a compact [n][Wk] array -- the host receives {status, flips,
 * iterations, crc} and the words, and touches nothing per bit.  A block that failed gets its ORIGINAL words back (key untouched).
 */
__global__ __launch_bounds__(RK_LANES) void rk_verify(const uint32_t *__restrict__ out, const uint32_t *__restrict__ key, const int *__restrict__ key_bits,
                                                      const uint32_t *__restrict__ crc_want, const int *__restrict__ ok, const int *__restrict__ iters,
                                                      uint32_t *__restrict__ res_keys, int *__restrict__ res, int n, int Wk, int Wn)
{
    __shared__ uint32_t s_tab[256], s_reg[RK_LANES];
    __shared__ int s_cnt[RK_LANES];
    const int b = (int)blockIdx.x;
    const int kb = key_bits[b], nw = (kb + 31) / 32;
    int flips = 0;
    const uint32_t crc = rk_block_crc(out + (size_t)b * Wn, kb, key + (size_t)b * Wk, res_keys + (size_t)b * Wk, &flips, s_tab, s_reg, s_cnt);
    const bool good = ok[b] && crc == crc_want[b];
    if (!good)      /* uniform per workgroup */
        for (int i = (int)threadIdx.x; i < nw; i += RK_LANES) res_keys[(size_t)b * Wk + i] = key[(size_t)b * Wk + i];
    if (threadIdx.x == 0) {
        res[b] = good ? QLDPC_OK : QLDPC_EDECODE;
        res[n + b] = good ? flips : 0;
        res[2 * n + b] = iters[b];
        res[3 * n + b] = (int)crc;
    }
}

/* Alice: CRC-32 of every key as it lies on the device (information words [n][Wk]) */
__global__ __launch_bounds__(RK_LANES) void rk_crc(const uint32_t *__restrict__ key, const int *__restrict__ key_bits, uint32_t *__restrict__ crc_out, int Wk)
{
    __shared__ uint32_t s_tab[256], s_reg[RK_LANES];
    __shared__ int s_cnt[RK_LANES];
    const int b = (int)blockIdx.x;
    const uint32_t crc = rk_block_crc(key + (size_t)b * Wk, key_bits[b], nullptr, nullptr, nullptr, s_tab, s_reg, s_cnt);
    if (threadIdx.x == 0) crc_out[b] = crc;
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
    c->schedule = QLDPC_RECON_SCHED_AUTO;
    c->mother_step = 8192;
    c->mother_max = 65536;
    c->rate_gap = 0.0f;             /* 0 = by rule: 0.03 for SPA, 0.05 for the min-sum family */
    c->puncture = 1;
    c->preload = 0;
    c->gap_profile = 0;
    c->peg_depth = 2;               /* mother codes by progressive edge growth, no 4-cycles */
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
    (void)hipFree(e.d_cls); (void)hipFree(e.d_out); (void)hipFree(e.d_iters); (void)hipFree(e.d_ok);
    for (auto &st : e.set) {
        (void)hipFree(st.d_in); (void)hipFree(st.d_res); (void)hipFree(st.d_bits); (void)hipFree(st.d_erase);
        if (st.h_in) (void)hipHostFree(st.h_in);
        if (st.h_res) (void)hipHostFree(st.h_res);
        if (st.ev_in) (void)hipEventDestroy(st.ev_in);
        if (st.ev_out) (void)hipEventDestroy(st.ev_out);
        if (st.ev_done) (void)hipEventDestroy(st.ev_done);
    }
}

extern "C" void qldpc_recon_free(qldpc_recon *r)
{
    if (!r) return;
    (void)hipSetDevice(r->cfg.device);
    for (auto &e : r->cache) entry_free(e);
    for (auto &l : r->lane) {
        if (l.stream) (void)hipStreamDestroy(l.stream);
        if (l.copy) (void)hipStreamDestroy(l.copy);
    }
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

static int get_entry(qldpc_recon *r, int K, int M, recon_entry **out, qldpc_code *prebuilt = nullptr);
static int build_code(const qldpc_recon_cfg &cfg, int K, int M, qldpc_code **out);
static void cache_trim(qldpc_recon *r)
{
    if (r->cache.size() <= r->keep) return;
    (void)hipDeviceSynchronize();
    while (r->cache.size() > r->keep) { entry_free(r->cache.back()); r->cache.pop_back(); }
}
#define RECON_IN_SCALARS 4      /* |LLR|, block length, punctured, Alice's CRC */
#define RECON_RES_SCALARS 4     /* status, bits flipped, iterations, CRC found */

extern "C" int qldpc_recon_create(const qldpc_recon_cfg *cfg, qldpc_recon **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = nullptr;
    if (!cfg) return QLDPC_EINVAL;
    if (cfg->n_rates < 1 || cfg->n_rates > 8 || cfg->key_quantum < 32 || (cfg->key_quantum & 31) || cfg->max_blocks < 1 || cfg->n_ite < 1 ||
        !(cfg->efficiency > 0.0f) || cfg->mother_step < 0 || (cfg->mother_step & 31) || (cfg->mother_step > 0 && cfg->mother_max < cfg->mother_step) ||
        !(cfg->rate_gap >= 0.0f && cfg->rate_gap < 0.5f) || cfg->peg_depth < 0 || cfg->peg_depth > 4 || cfg->gap_profile < 0 || cfg->gap_profile > 1 || cfg->schedule < 0 || cfg->schedule > QLDPC_RECON_SCHED_AUTO) {
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
    r->profiling = 0;
    memset(r->lane, 0, sizeof(r->lane));
    if (hipSetDevice(cfg->device) != hipSuccess) { delete r; return QLDPC_EHIP; }
    /* the session's own streams: nothing of it runs on the null stream, so a second session or other work on the device does not
     * serialise behind it */
    /* the runtime binds a stream to one of its few hardware queues when the stream is created, round robin: the compute streams are
     * created back to back so that the lanes land on different queues (interleaved with the copy streams, four lanes shared two queues
     * and at most two kernels ever ran side by side: rocprofv3 kernel trace, profiles/r03_config3_*) */
    bool ok = true;
    const bool interleave = getenv("QLDPC_RECON_STREAMS_INTERLEAVED") != nullptr;      /* A/B: the old order */
    if (interleave) { for (auto &l : r->lane) ok = ok && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&l.copy, hipStreamNonBlocking) == hipSuccess; }
    else {
        for (auto &l : r->lane) ok = ok && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess;
        for (auto &l : r->lane) ok = ok && hipStreamCreateWithFlags(&l.copy, hipStreamNonBlocking) == hipSuccess;
    }
    if (!ok) {
        qldpc_set_error("recon_create: hipStreamCreate failed");
        qldpc_recon_free(r);
        return QLDPC_EHIP;
    }
    if (cfg->preload && mothers) {
        /* every (mother size, table rate) pair now: code, encoder, decoder and staging buffers, so that no block of up to mother_max
         * bits builds a code or allocates device memory later (the daemon calls this from ldpc_init) */
        struct todo { int K, M; qldpc_code *code; int rc; std::string err; };
        std::vector<todo> list;
        for (int k = cfg->mother_step; k <= cfg->mother_max; k += cfg->mother_step)
            for (int i = 0; i < cfg->n_rates; i++) {
                todo t;
                code_dims(r->cfg, k, cfg->rates[i], &t.K, &t.M);
                t.code = nullptr; t.rc = QLDPC_OK;
                list.push_back(t);
            }
        /* the codes are host work (progressive edge growth: 0.1 - 10 s each unless QLDPC_CODE_CACHE holds them): built side by side on
         * the host's cores, largest first; the device objects follow one by one */
        std::vector<size_t> order(list.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return (double)list[a].K / list[a].M * list[a].K > (double)list[b].K / list[b].M * list[b].K; });
        std::atomic<size_t> next(0);
        auto worker = [&]() {
            for (size_t at = next++; at < order.size(); at = next++) {
                todo &t = list[order[at]];
                t.rc = build_code(r->cfg, t.K, t.M, &t.code);
                if (t.rc) t.err = qldpc_last_error();
            }
        };
        unsigned nthreads = std::thread::hardware_concurrency();
        if (const char *env = getenv("QLDPC_BUILD_THREADS")) nthreads = (unsigned)atoi(env);
        nthreads = std::max(1u, std::min(nthreads, std::min(16u, (unsigned)list.size())));
        if (cfg->peg_depth <= 0) nthreads = 1;      /* the seeded shuffle takes milliseconds */
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthreads; t++) th.emplace_back(worker);
        worker();
        for (auto &t : th) t.join();
        int rc = QLDPC_OK;
        for (auto &t : list) if (t.rc && !rc) { rc = t.rc; qldpc_set_error("%s", t.err.c_str()); }
        for (auto &t : list) {
            if (rc) { qldpc_code_free(t.code); continue; }
            recon_entry *e;
            rc = get_entry(r, t.K, t.M, &e, t.code);      /* takes the code over (also on failure: entry_free) */
        }
        if (rc) { qldpc_recon_free(r); return rc; }
    }
    *out = r;
    return QLDPC_OK;
}

extern "C" long qldpc_recon_entries_created(const qldpc_recon *r) { return r ? r->created : -1; }

/* per-kernel profile of Bob's decoders (qldpc_profile_*), summed over the codes of the session */
extern "C" int qldpc_recon_profile_enable(qldpc_recon *r, int on)
{
    if (!r) return QLDPC_EINVAL;
    r->profiling = on;
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

/*
 * c(R): how much of rate_gap a mother code of rate R needs.  Calibrated on streams of 2 048 epochs x 52 429 bits, QBER ~ U[0.5 %, 6 %], two
 * seeds each (qldpc_stream -e 2048 -P <depth>, QLDPC_RECON_GAP_SCALE; logs: profiles/r03_gap_calibration.txt):
 *   PEG-built mothers (peg_depth > 0)   0.10 (mothers of K >= 32 768; more below) / 0.85 / 1.0   for R <= 0.75 / <= 0.85 / above: 0 - 1 first-round failures in 2 048 epochs, leak 0.2896 of
 *                                       the key (the configured efficiency f = 1.4 alone gives 0.288); the low-rate mothers decode to within
 *                                       0.004 of capacity, the rate-0.8 / 0.9 mothers need their 0.03 (0.025: 2 - 4 % failures)
 *   seeded-shuffle mothers (round 2)    0.60 / 0.90 / 1.0   (0.85 for the middle value: 39 failures of 918 on one seed where PEG has none)
 * QLDPC_RECON_GAP_SCALE="lo,mid,hi" overrides the three values for calibration runs -- both sides must then use the same.
 */
static double gap_scale(const qldpc_recon_cfg &cfg, double R, int K)
{
    struct from_env {      /* read once, behind the guard of a function-local static (sessions may plan on several threads) */
        double v[3] = {-1.0, -1.0, -1.0};
        from_env() { if (const char *e = getenv("QLDPC_RECON_GAP_SCALE")) (void)sscanf(e, "%lf,%lf,%lf", &v[0], &v[1], &v[2]); }
    };
    static const from_env env_scale;
    const double *env = env_scale.v;
    const int k = R <= 0.75 ? 0 : (R <= 0.85 ? 1 : 2);
    if (env[0] >= 0.0 && env[k] >= 0.0) return env[k];
    static const double peg[3] = {0.10, 0.85, 1.0}, shuffle[3] = {0.60, 0.90, 1.0};
    if (cfg.peg_depth <= 0 || cfg.gap_profile == 1) return shuffle[k];
    /* short low-rate mothers need more room: 0.1 from K = 32 768 upwards, 0.25 at 16 384, 0.6 at 8 192 (7 000-bit blocks: 24 failures of
     * 1 203 at rate 0.7 with 0.1 - 0.3, 1 with 0.6; 15 000-bit blocks: 1 - 2 of 1 100 at any value) */
    if (k == 0) return std::min(0.6, std::max(0.10, 0.10 * pow(32768.0 / (double)K, 1.3)));
    return peg[k];
}

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
        const double gap = gap_len * gap_scale(r->cfg, R, K0);
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

/* the code of an entry: DVB-like IRA profile (12.5 % of the information VNs of degree 11, the rest 3), information part by the seeded socket
 * shuffle or by progressive edge growth (peg_depth); both sides derive the same code from (K, M, peg_depth, seed) */
static int build_code(const qldpc_recon_cfg &cfg, int K, int M, qldpc_code **out)
{
    const int N = K + M;
    return cfg.peg_depth > 0 ? qldpc_code_ira_peg(N, K, 0.125f, 11, 3, cfg.peg_depth, cfg.seed, out) : qldpc_code_ira(N, K, 0.125f, 11, 3, cfg.seed, out);
}

static int get_entry(qldpc_recon *r, int K, int M, recon_entry **out, qldpc_code *prebuilt)
{
    for (auto it = r->cache.begin(); it != r->cache.end(); ++it)
        if (it->K == K && it->M == M) { r->cache.splice(r->cache.begin(), r->cache, it); *out = &r->cache.front(); return QLDPC_OK; }
    recon_entry e;
    memset(&e, 0, sizeof(e));
    e.K = K; e.M = M;
    const int N = K + M, Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32, B = r->cfg.max_blocks;
    int rc = QLDPC_OK;
    if (prebuilt) e.code = prebuilt;
    else rc = build_code(r->cfg, K, M, &e.code);
    if (!rc) rc = qldpc_encoder_create(e.code, "IRA", r->cfg.device, &e.enc);
    if (!rc) rc = qldpc_encoder_reserve(e.enc, B);      /* nothing is allocated per block later */
    if (!rc) {
        qldpc_decoder_cfg dc;
        qldpc_decoder_cfg_default(&dc);
        /* batches run the horizontal-layered schedule unless told otherwise: half the iterations for the same bytes per sweep (config-3 stream 16.1 -> 14.4 ms, 0 instead of 0 - 1 first-round
         * failures per 2 048 epochs); a decoder for <= 8 blocks is the edge-parallel engine, which is flooding */
        const bool layered = r->cfg.schedule == QLDPC_SCHED_HLAYERED || (r->cfg.schedule == QLDPC_RECON_SCHED_AUTO && B > 8);
        dc.schedule = layered ? QLDPC_SCHED_HLAYERED : QLDPC_SCHED_FLOODING; dc.rule = r->cfg.rule; dc.rule_param = r->cfg.rule_param; dc.n_ite = r->cfg.n_ite;
        dc.enable_syndrome = 1; dc.syndrome_depth = 1; dc.max_frames = B; dc.device = r->cfg.device;
        dc.layer_chain = 2;      /* the session's decoders run side by side (lanes): persistent one-launch sweeps would fight for the chip (measured 19.0 -> 23.9 ms) */
        rc = qldpc_decoder_create(e.code, K, nullptr, &dc, &e.dec);
        if (!rc) rc = qldpc_decoder_reserve(e.dec);      /* nothing is allocated per block later */
    }
    auto alloc = [&](void **ptr, size_t bytes) { if (!rc && hipMalloc(ptr, bytes) != hipSuccess) rc = QLDPC_ENOMEM; };
    alloc((void **)&e.d_cls, (size_t)N);
    alloc((void **)&e.d_out, sizeof(uint32_t) * (size_t)B * Wn);
    alloc((void **)&e.d_iters, sizeof(int) * (size_t)B);
    alloc((void **)&e.d_ok, sizeof(int) * (size_t)B);
    for (auto &st : e.set) {
        const size_t in_words = (size_t)B * (Wk + Wm + RECON_IN_SCALARS), res_words = (size_t)B * (Wk + RECON_RES_SCALARS);
        alloc((void **)&st.d_in, sizeof(uint32_t) * in_words);
        alloc((void **)&st.d_res, sizeof(uint32_t) * std::max(res_words, (size_t)B * (Wm + 1)));      /* Alice's side: disclosed parity [n][Wm] | CRC [n] */
        alloc((void **)&st.d_bits, sizeof(uint32_t) * (size_t)B * Wn);
        alloc((void **)&st.d_erase, sizeof(uint32_t) * (size_t)B * Wn);
        if (!rc && hipHostMalloc((void **)&st.h_in, sizeof(uint32_t) * in_words) != hipSuccess) rc = QLDPC_ENOMEM;
        if (!rc && hipHostMalloc((void **)&st.h_res, sizeof(uint32_t) * std::max(res_words, (size_t)B * (Wm + 1))) != hipSuccess) rc = QLDPC_ENOMEM;
        if (!rc && (hipEventCreateWithFlags(&st.ev_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&st.ev_out, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&st.ev_done, hipEventDisableTiming) != hipSuccess)) rc = QLDPC_EHIP;
    }
    if (!rc) {
        std::vector<uint8_t> cls((size_t)N, (uint8_t)QLDPC_VN_PINNED);       /* parity VNs are disclosed; key VNs past a block's length are pinned per frame */
        for (int i = 0; i < K; i++) cls[(size_t)i] = (uint8_t)QLDPC_VN_CHANNEL;
        if (hipMemcpy(e.d_cls, cls.data(), (size_t)N, hipMemcpyHostToDevice) != hipSuccess) rc = QLDPC_EHIP;
    }
    if (rc) { entry_free(e); return rc; }
    if (r->profiling) qldpc_profile_enable(e.dec, 1);
    r->cache.push_front(e);      /* the cache is trimmed to `keep` entries at the end of the call (cache_trim): nothing in use is evicted */
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

extern "C" int qldpc_recon_parity_words(const qldpc_recon_msg *msg) { return msg ? (int)((msg->code_m - msg->n_punct + 31) / 32) : QLDPC_EINVAL; }
extern "C" int qldpc_recon_leaked_bits(const qldpc_recon_msg *msg) { return msg ? (int)(msg->code_m - msg->n_punct) + 32 : QLDPC_EINVAL; }

/* ------------------------------------------------------------------ the pipeline ------------------------------------------------
 *
 * A call brings blocks of any mix of codes.  Blocks of one code (entry) form batches of up to max_blocks frames = jobs; the jobs of
 * one entry go to one lane (an entry has one decoder), the entries of a call are spread over up to RECON_LANES lanes, and every
 * lane is a host worker with its own compute and copy stream:
 *
 *     copy stream     H2D(j) assemble(j) | H2D(j+1) assemble(j+1) | D2H(j)            | H2D(j+2) ...
 *     compute stream                     | load(j) BP iterations(j) fetch verify(j)   | load(j+1) ...
 *     host            stage(j) stage(j+1)  ... polls the early-exit mailbox of j ...    results(j-1) stage(j+2)
 *
 * so the rate groups of a stream of epochs decode side by side (their launches fill each other's tails), and within a lane the
 * staging and the copies of the neighbouring batches hide behind the decode.  Nothing runs on the null stream.
 */
struct recon_job {
    recon_entry *e;
    std::vector<int> idx;       /* the job's blocks in the caller's arrays */
    stage_set *st;
    bool any_punct;
};

struct recon_call {
    bool bob;
    int n;
    const int *key_bits;
    /* Bob */
    uint32_t *const *key; const float *qber; const qldpc_recon_msg *msgs; const uint32_t *const *parity; int *status, *corrected, *iterations;
    /* Alice */
    const uint32_t *const *akey; qldpc_recon_msg *amsgs; uint32_t *const *aparity;
};

struct lane_run {
    qldpc_recon *r;
    recon_lane *lane;
    const recon_call *call;
    std::vector<recon_job> jobs;
    int rc;
    std::string err;
};

static int job_stage(lane_run *L, recon_job &j, hipStream_t cs)
{
    recon_entry *e = j.e;
    const recon_call &c = *L->call;
    const int n = (int)j.idx.size(), K = e->K, M = e->M, N = K + M;
    const int Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32;
    stage_set *st = &e->set[e->next_set];
    e->next_set ^= 1;
    j.st = st;
    j.any_punct = false;
    uint32_t *h_keys = st->h_in, *h_disc = h_keys + (size_t)n * Wk;
    float *h_mag = reinterpret_cast<float *>(h_disc + (size_t)n * Wm);
    int *h_nch = reinterpret_cast<int *>(h_mag + n), *h_np = h_nch + n;
    uint32_t *h_crc = reinterpret_cast<uint32_t *>(h_np + n);
    for (int t = 0; t < n; t++) {
        const int i = j.idx[(size_t)t], kb = c.key_bits[i], Wkey = (kb + 31) / 32;
        uint32_t *f = h_keys + (size_t)t * Wk;
        if (c.bob) {
            /* only the words the kernels read are copied: rk_assemble masks the tail and never looks past the block's length */
            memcpy(f, c.key[i], sizeof(uint32_t) * (size_t)Wkey);
            h_np[t] = (int)c.msgs[i].n_punct;
            memcpy(h_disc + (size_t)t * Wm, c.parity[i], sizeof(uint32_t) * (size_t)((M - h_np[t] + 31) / 32));
            h_mag[t] = qldpc_bsc_llr(clamp_qber(c.qber[i]));
            h_crc[t] = c.msgs[i].crc32;
        } else {
            /* information word of the mother code: the key, then zeros (shortened positions) */
            memcpy(f, c.akey[i], sizeof(uint32_t) * (size_t)Wkey);
            if (kb & 31) f[Wkey - 1] &= 0xFFFFFFFFu << (32 - (kb & 31));
            memset(f + Wkey, 0, sizeof(uint32_t) * (size_t)(Wk - Wkey));
            h_np[t] = (int)c.amsgs[i].n_punct;
            h_mag[t] = 0.0f; h_crc[t] = 0;
        }
        h_nch[t] = kb;
        j.any_punct = j.any_punct || h_np[t] != 0;
    }
    const size_t in_words = (size_t)n * (Wk + Wm + RECON_IN_SCALARS);
    HIPCHK(hipMemcpyAsync(st->d_in, st->h_in, sizeof(uint32_t) * in_words, hipMemcpyHostToDevice, cs));
    if (c.bob) {
        uint32_t *d_keys = st->d_in, *d_disc = d_keys + (size_t)n * Wk;
        int *d_nch = reinterpret_cast<int *>(d_disc + (size_t)n * Wm + n), *d_np = d_nch + n;
        hipLaunchKernelGGL(rk_assemble, dim3((unsigned)((Wn + 255) / 256), (unsigned)n), dim3(256), 0, cs, d_keys, d_disc, d_nch, d_np, st->d_bits, st->d_erase, Wk, Wn, Wm, M);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(st->ev_in, cs));
    return QLDPC_OK;
}

static int job_launch(lane_run *L, recon_job &j, hipStream_t cs)
{
    recon_entry *e = j.e;
    stage_set *st = j.st;
    hipStream_t s = L->lane->stream;
    const int n = (int)j.idx.size(), K = e->K, M = e->M, N = K + M;
    const int Wk = K / 32, Wn = (N + 31) / 32, Wm = (M + 31) / 32;
    int rc;
    uint32_t *d_keys = st->d_in, *d_disc = d_keys + (size_t)n * Wk;
    float *d_mag = reinterpret_cast<float *>(d_disc + (size_t)n * Wm);
    int *d_nch = reinterpret_cast<int *>(d_mag + n), *d_np = d_nch + n;
    uint32_t *d_crc = reinterpret_cast<uint32_t *>(d_np + n);
    if (cs != s) HIPCHK(hipStreamWaitEvent(s, st->ev_in, 0));
    size_t res_words;
    if (L->call->bob) {
        if ((rc = qldpc_decoder_set_stream(e->dec, (void *)s))) return rc;
        if ((rc = qldpc_load_bits_short_dev(e->dec, st->d_bits, d_mag, e->d_cls, d_nch, n))) return rc;
        if (j.any_punct && (rc = qldpc_load_erasures_dev(e->dec, st->d_erase, n))) return rc;
        if ((rc = qldpc_run(e->dec))) return rc;      /* the host polls the decoder's early-exit mailbox in here */
        if ((rc = qldpc_fetch_packed_dev(e->dec, e->d_out))) return rc;
        if ((rc = qldpc_fetch_status_dev(e->dec, e->d_iters, e->d_ok))) return rc;
        uint32_t *res_keys = st->d_res;
        int *res = reinterpret_cast<int *>(res_keys + (size_t)n * Wk);
        hipLaunchKernelGGL(rk_verify, dim3((unsigned)n), dim3(RK_LANES), 0, s, e->d_out, d_keys, d_nch, d_crc, e->d_ok, e->d_iters, res_keys, res, n, Wk, Wn);
        HIPCHK(hipGetLastError());
        res_words = (size_t)n * (Wk + RECON_RES_SCALARS);
    } else {
        if ((rc = qldpc_encode_packed_dev(e->enc, d_keys, e->d_out, n, (void *)s))) return rc;
        uint32_t *res_disc = st->d_res, *res_crc = res_disc + (size_t)n * Wm;
        hipLaunchKernelGGL(rk_disclose, dim3((unsigned)((Wm + 255) / 256), (unsigned)n), dim3(256), 0, s, e->d_out, d_np, res_disc, Wk, Wn, Wm, M);
        hipLaunchKernelGGL(rk_crc, dim3((unsigned)n), dim3(RK_LANES), 0, s, d_keys, d_nch, res_crc, Wk);
        HIPCHK(hipGetLastError());
        res_words = (size_t)n * (Wm + 1);
    }
    if (cs != s) {
        HIPCHK(hipEventRecord(st->ev_out, s));
        HIPCHK(hipStreamWaitEvent(cs, st->ev_out, 0));
    }
    HIPCHK(hipMemcpyAsync(st->h_res, st->d_res, sizeof(uint32_t) * res_words, hipMemcpyDeviceToHost, cs));
    HIPCHK(hipEventRecord(st->ev_done, cs));
    return QLDPC_OK;
}

static int job_finish(lane_run *L, recon_job &j)
{
    recon_entry *e = j.e;
    stage_set *st = j.st;
    const recon_call &c = *L->call;
    const int n = (int)j.idx.size(), K = e->K, M = e->M;
    const int Wk = K / 32, Wm = (M + 31) / 32;
    HIPCHK(hipEventSynchronize(st->ev_done));
    if (c.bob) {
        const uint32_t *keys = st->h_res;
        const int *res = reinterpret_cast<const int *>(keys + (size_t)n * Wk);
        for (int t = 0; t < n; t++) {
            const int i = j.idx[(size_t)t], Wkey = (c.key_bits[i] + 31) / 32;
            c.status[i] = res[t];
            if (c.corrected) c.corrected[i] = res[n + t];
            if (c.iterations) c.iterations[i] = res[2 * n + t];
            if (res[t] == QLDPC_OK) memcpy(c.key[i], keys + (size_t)t * Wk, sizeof(uint32_t) * (size_t)Wkey);      /* a failed block keeps its key untouched */
        }
    } else {
        const uint32_t *disc = st->h_res, *crc = disc + (size_t)n * Wm;
        for (int t = 0; t < n; t++) {
            const int i = j.idx[(size_t)t];
            memcpy(c.aparity[i], disc + (size_t)t * Wm, sizeof(uint32_t) * (size_t)qldpc_recon_parity_words(&c.amsgs[i]));
            c.amsgs[i].crc32 = crc[t];
        }
    }
    return QLDPC_OK;
}

static void lane_main(lane_run *L)
{
    int rc = QLDPC_OK;
    const bool dbg = getenv("QLDPC_DEBUG") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    if (hipSetDevice(L->r->cfg.device) != hipSuccess) { L->rc = QLDPC_EHIP; L->err = "hipSetDevice failed"; return; }
    const size_t nj = L->jobs.size();
    /* one job: everything in order on the compute stream (no cross-stream hand-offs on the daemon's one-block path) */
    hipStream_t cs = nj > 1 ? L->lane->copy : L->lane->stream;
    size_t launched = 0, finished = 0;
    rc = job_stage(L, L->jobs[0], cs);
    for (size_t i = 0; i < nj && !rc; i++) {
        if (i + 1 < nj) rc = job_stage(L, L->jobs[i + 1], cs);
        if (!rc) { rc = job_launch(L, L->jobs[i], cs); if (!rc) launched = i + 1; }
        if (!rc && i > 0) { rc = job_finish(L, L->jobs[i - 1]); if (!rc) finished = i; }
    }
    for (size_t i = finished; i < launched && !rc; i++) rc = job_finish(L, L->jobs[i]);
    if (dbg) {
        size_t blocks = 0;
        for (auto &j : L->jobs) blocks += j.idx.size();
        fprintf(stderr, "libqldpc: recon lane %d: %zu job(s), %zu blocks, first code K %d M %d, %.3f ms\n", (int)(L->lane - L->r->lane), nj, blocks, L->jobs[0].e->K, L->jobs[0].e->M,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    }
    if (rc) {
        L->err = qldpc_last_error();
        (void)hipStreamSynchronize(L->lane->stream);
        (void)hipStreamSynchronize(L->lane->copy);
    }
    L->rc = rc;
}

/* groups: blocks by code.  Entries are looked up / built here, in the calling thread; jobs of one entry stay on one lane, the
 * entries are dealt onto the lanes largest first. */
static int run_call(qldpc_recon *r, const recon_call &call, const std::vector<char> &skip)
{
    int rc;
    HIPCHK(hipSetDevice(r->cfg.device));
    struct group { recon_entry *e; std::vector<int> idx; double cost; };
    std::vector<group> groups;
    const int n = call.n;
    std::vector<char> taken(skip);
    for (int i = 0; i < n; i++) {
        if (taken[(size_t)i]) continue;
        const qldpc_recon_msg &mi = call.bob ? call.msgs[i] : call.amsgs[i];
        group g;
        for (int j = i; j < n; j++) {
            const qldpc_recon_msg &mj = call.bob ? call.msgs[j] : call.amsgs[j];
            if (!taken[(size_t)j] && mj.code_k == mi.code_k && mj.code_m == mi.code_m) { g.idx.push_back(j); taken[(size_t)j] = 1; }
        }
        if ((rc = get_entry(r, (int)mi.code_k, (int)mi.code_m, &g.e))) { cache_trim(r); return rc; }
        g.cost = (double)g.idx.size() * (double)(mi.code_k + mi.code_m);
        groups.push_back(std::move(g));
    }
    if (groups.empty()) return QLDPC_OK;
    int n_lanes = RECON_LANES;
    if (const char *env = getenv("QLDPC_RECON_LANES")) n_lanes = std::max(1, std::min(RECON_LANES, atoi(env)));
    n_lanes = std::min(n_lanes, (int)groups.size());
    std::sort(groups.begin(), groups.end(), [](const group &a, const group &b) { return a.cost > b.cost; });
    std::vector<lane_run> runs((size_t)n_lanes);
    std::vector<double> load((size_t)n_lanes, 0.0);
    for (int l = 0; l < n_lanes; l++) { runs[(size_t)l].r = r; runs[(size_t)l].lane = &r->lane[l]; runs[(size_t)l].call = &call; runs[(size_t)l].rc = QLDPC_OK; }
    const size_t B = (size_t)r->cfg.max_blocks;
    for (auto &g : groups) {
        const size_t l = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
        load[l] += g.cost;
        for (size_t at = 0; at < g.idx.size(); at += B) {
            recon_job j;
            j.e = g.e; j.st = nullptr; j.any_punct = false;
            j.idx.assign(g.idx.begin() + (long)at, g.idx.begin() + (long)std::min(g.idx.size(), at + B));
            runs[l].jobs.push_back(std::move(j));
        }
    }
    if (n_lanes == 1) lane_main(&runs[0]);
    else {
        std::vector<std::thread> th;
        for (int l = 1; l < n_lanes; l++) th.emplace_back(lane_main, &runs[(size_t)l]);
        lane_main(&runs[0]);
        for (auto &t : th) t.join();
    }
    rc = QLDPC_OK;
    for (auto &L : runs)
        if (L.rc && !rc) { rc = L.rc; qldpc_set_error("%s", L.err.c_str()); }
    cache_trim(r);
    return rc;
}

/*
 * Alice's side for many blocks at once: plans every block (msgs[i] is filled as by qldpc_recon_encode), groups the blocks by
 * code and encodes each group in launches of up to max_blocks frames.  parity_words[i] receives qldpc_recon_parity_words(&msgs[i])
 * words (parity_cap[i] are available).  The key's CRC-32 is computed on the device as well (rk_crc).
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
    recon_call c;
    memset(&c, 0, sizeof(c));
    c.bob = false; c.n = n; c.key_bits = key_bits; c.akey = key_words; c.amsgs = msgs; c.aparity = parity_words;
    return run_call(r, c, std::vector<char>((size_t)n, 0));
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
    recon_call c;
    memset(&c, 0, sizeof(c));
    c.bob = false; c.n = 1; c.key_bits = &key_bits; c.akey = &key_words; c.amsgs = msg; c.aparity = &parity_words;
    return run_call(r, c, std::vector<char>(1, 0));
}

/*
 * Blocks of any mix of lengths, rates and puncturing in one call ("let many blocks queue and decode in one
 * launch"): blocks are grouped by code (code_k, code_m), every group goes through its decoder in batches of up to max_blocks
 * frames, and the groups of a call run side by side (the pipeline above); within a group the blocks may differ in length
 * (shortening per frame) and in efficiency (puncturing per frame).  keys are decoded in place.
 * Every message is validated on its own: status[i] = QLDPC_OK | QLDPC_EDECODE | QLDPC_ESIZE (header does not match the block /
 * the local plan -- that block is not decoded, the others are).
 */
extern "C" int qldpc_recon_decode_blocks(qldpc_recon *r, int n, uint32_t *const *key_words, const int *key_bits, const float *qber,
                                         const qldpc_recon_msg *msgs, const uint32_t *const *parity_words, int *status, int *corrected, int *iterations)
{
    if (!r || !key_words || !key_bits || !qber || !msgs || !parity_words || !status || n <= 0) return QLDPC_EINVAL;
    std::vector<char> skip((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        if (!key_words[i] || !parity_words[i]) return QLDPC_EINVAL;
        status[i] = QLDPC_EDECODE;
        if (corrected) corrected[i] = 0;
        if (iterations) iterations[i] = 0;
        if (check_msg(r, &msgs[i], key_bits[i]) || !(qber[i] >= 0.0f && qber[i] < 0.5f)) { status[i] = QLDPC_ESIZE; skip[(size_t)i] = 1; }
    }
    recon_call c;
    memset(&c, 0, sizeof(c));
    c.bob = true; c.n = n; c.key_bits = key_bits; c.key = key_words; c.qber = qber; c.msgs = msgs; c.parity = parity_words;
    c.status = status; c.corrected = corrected; c.iterations = iterations;
    return run_call(r, c, skip);
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
