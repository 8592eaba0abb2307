/*
 * qldpc_kernels_edge.h -- gfx950 kernels of the LOW-BATCH ("edge-parallel") BP decoder.
 *
 * The frame-interleaved engine (qldpc_kernels.h) needs >= 64 frames to fill a wavefront.  The ecd2
 * daemon decodes ONE block of <= 65 535 bits at a time, so here the wavefront lanes run across the
 * EDGES of one frame instead:
 *   check nodes : a workgroup owns QE_CPB consecutive checks; their contiguous CN-major edge range is
 *                 staged into LDS with fully coalesced index loads + one gather per edge, each check
 *                 is then folded by a segment of S = 2^k lanes with wavefront shuffles (butterfly
 *                 all-reduce of (min1, min2, sign) -- exact for the min-sum family because min /
 *                 second-min / xor do not depend on association order), results go back through LDS
 *                 and are scattered with all lanes busy.  The syndrome of the previous posterior is
 *                 folded in the same pass (one extra bit per edge).
 *   variable nodes: a workgroup owns QE_THREADS consecutive VNs = one contiguous slot range, staged through
 *                 LDS so global traffic is coalesced; each lane sums its VN's slots in order (the same
 *                 order as AFF3CT's _initialize_var_to_chk) and the hard decisions leave as wave
 *                 ballots, i.e. directly as MSB-first packed words (helpers.h:65-70).
 * Layout: llr [F][N], v2c [F][E], c2v [2][F][E] (VN-major slots; ping-pong by iteration parity),
 * sgn / hard [F][ceil(N/32)] packed words.
 * State lives in L2 / Infinity Cache (2 MB per frame at N = 65 536); no MFMA.
 */
#ifndef QLDPC_KERNELS_EDGE_H
#define QLDPC_KERNELS_EDGE_H

#include <hip/hip_cooperative_groups.h>

#include "qldpc_kernels.h"

#ifndef QE_THREADS
#define QE_THREADS 512               /* threads per workgroup = VNs per variable-node chunk (measured per block, with 32 checks per check chunk:
                                     128: 340 us, 256: 302 us, 512: 284 us, 1024: 313 us) */
#endif
#ifndef QE_CPB
#define QE_CPB 32                 /* checks per workgroup (measured per 65 536-VN block at 256 threads: 64: 335 us, 32: 302 us, 16: 306 us) */
#endif
#define QE_MAX_EDGES (QE_CPB * 64)

template <bool COH> __device__ __forceinline__ float qe_ld(const float *p);
template <bool COH> __device__ __forceinline__ uint32_t qe_ldu(const uint32_t *p);

/* convergence bookkeeping of one frame, all in global memory:
 *   unsat[f][ite]  OR over checks of the syndrome of the bits that existed when CN(ite) ran
 *   done_at[f]     -1, or the iteration count at which AFF3CT's loop would have broken            */
__device__ __forceinline__ bool qe_converged(const int *__restrict__ unsat, int stride, int f, int ite, int depth)
{
    /* bits checked in CN(ite) are the posterior after iteration ite-1; AFF3CT needs `depth`
     * consecutive zero syndromes, the first check happens after iteration 0 (slot 1) */
    if (ite < depth) return false;
    for (int k = 0; k < depth; k++) if (unsat[(size_t)f * stride + ite - k] != 0) return false;
    return true;
}

/* ------------------------------------------------------------------ check nodes -------------- */

/* one chunk of QE_CPB checks of frame f; s_idx / s_val: QE_MAX_EDGES words of LDS each, s_unsat: one LDS int, s_ptr: QE_CPB + 1 LDS ints.
 * Every thread of the workgroup must call it (it synchronises the workgroup).
 * The pass is latency-bound (a chunk is a chain of dependent round trips to the L2 / Infinity Cache, not bytes), so the chain is kept
 * short: the chunk's slice of cn_ptr is staged into LDS with the edges (round 2 re-read it from global memory inside the fold loop:
 * two dependent loads per round), and a thread asks for the indices of BOTH its edges before it waits and then for both gathers --
 * two round trips for up to 2 * QE_THREADS edges where the one-edge-per-trip loop made four (profiles/r03_edge_engine_breakdown.txt). */
template <int S, int FAM, bool COH = false>
__device__ __forceinline__ void qe_cn_chunk(int f, int chunk, int *s_idx, float *s_val, int *s_unsat_p, int *s_ptr,
                                            const float *__restrict__ v2c, float *__restrict__ c2v,
                                            const int *__restrict__ cn_ptr, const int *__restrict__ cn_tr,
                                            const int *__restrict__ cn_var, const uint32_t *__restrict__ sgn,
                                            int M, int E, int W, int *__restrict__ unsat, int unsat_stride, int ite,
                                            qk_rule rule, int syndrome_only, const uint32_t *__restrict__ synd, int Wm)
{
#define s_unsat (*s_unsat_p)
    const int tid = threadIdx.x;
    const int c0 = chunk * QE_CPB;
    const int c1 = min(M, c0 + QE_CPB);
    const int e0 = cn_ptr[c0], nE = cn_ptr[c1] - e0;
    const float *vin = v2c + (size_t)f * E;
    float *cout = c2v + (size_t)f * E;
    const uint32_t *sw = sgn + (size_t)f * W;
    if (tid == 0) s_unsat = 0;
    /* phase A: stage the chunk's edges (index + message + sign bit of the VN's last posterior) and its slice of cn_ptr */
    const int pv = cn_ptr[c0 + (tid <= c1 - c0 ? tid : 0)];      /* asked for with the first indices, written to LDS behind them */
    for (int base = 0; base < nE; base += 2 * QE_THREADS) {
        const int ea = base + tid, eb = ea + QE_THREADS;
        const bool a = ea < nE, b = eb < nE;
        const int ia = e0 + (a ? ea : 0), ib = e0 + (b ? eb : 0);      /* clamped: every load below has a valid address, no branch in between */
        const int sa = cn_tr[ia], va = cn_var[ia], sb = cn_tr[ib], vb = cn_var[ib];
        const uint32_t wa = qe_ldu<COH>(sw + (va >> 5)), wb = qe_ldu<COH>(sw + (vb >> 5));
        const float xa = syndrome_only ? 0.0f : qe_ld<COH>(vin + sa), xb = syndrome_only ? 0.0f : qe_ld<COH>(vin + sb);
        if (a) { s_idx[ea] = sa | (int)(((wa >> (31 - (va & 31))) & 1u) << 31); s_val[ea] = xa; }
        if (b) { s_idx[eb] = sb | (int)(((wb >> (31 - (vb & 31))) & 1u) << 31); s_val[eb] = xb; }
    }
    if (tid <= c1 - c0) s_ptr[tid] = pv - e0;
    __syncthreads();
    /* phase B: one S-lane segment per check, butterfly all-reduce with wavefront shuffles */
    constexpr int PER_ROUND = QE_THREADS / S;
    const int seg = tid / S, s = tid % S;
    uint32_t any_unsat = 0;
    for (int c = c0 + seg; c < c0 + QE_CPB; c += PER_ROUND) {
        const bool live = c < c1;
        const int off = live ? s_ptr[c - c0] : 0;
        const int deg = live ? s_ptr[c - c0 + 1] - off : 0;
        const bool act = s < deg;
        const float x = act ? s_val[off + s] : 0.0f;
        uint32_t par = act ? ((uint32_t)s_idx[off + s] >> 31) : 0u;
        /* syndrome form: lane 0 of the segment carries the target parity s_c into both folds */
        const uint32_t sc = (synd && live && s == 0) ? ((synd[(size_t)f * Wm + (c >> 5)] >> (31 - (c & 31))) & 1u) : 0u;
        par ^= sc;
        if constexpr (FAM == QK_FAM_MS) {
            float m1 = act ? fabsf(x) : 3.402823466e+38f, m2 = 3.402823466e+38f;
            uint32_t sg = (act ? (qk_bits(x) & 0x80000000u) : 0u) ^ (sc << 31);
#pragma unroll
            for (int o = 1; o < S; o <<= 1) {
                const float p1 = __shfl_xor(m1, o), p2 = __shfl_xor(m2, o);
                sg ^= __shfl_xor(sg, o);
                par ^= __shfl_xor(par, o);
                const float lo = qk_min(m1, p1), hi = qk_max(m1, p1);
                m2 = qk_min(hi, qk_min(m2, p2));
                m1 = lo;
            }
            if (act && !syndrome_only) {
                float cst1, cst2;
                if (rule.rule == 0)      { cst1 = qk_max(0.0f, m2);              cst2 = qk_max(0.0f, m1); }
                else if (rule.rule == 1) { cst1 = qk_max(0.0f, m2 - rule.param); cst2 = qk_max(0.0f, m1 - rule.param); }
                else                     { cst1 = m2 * rule.param;               cst2 = m1 * rule.param; }
                s_val[off + s] = qk_withsign((fabsf(x) == m1) ? cst1 : cst2, sg ^ qk_bits(x));      /* sg already folds s_c in */
            }
        } else {   /* SPA: the product is reduced as a tree, so it is tolerance-class (not order-exact) */
            const float t = act ? qk_tanh_half(fabsf(x)) : 1.0f;
            float prod = t;
            uint32_t sg = (act ? (qk_bits(x) & 0x80000000u) : 0u) ^ (sc << 31);
#pragma unroll
            for (int o = 1; o < S; o <<= 1) {
                prod *= __shfl_xor(prod, o);
                sg ^= __shfl_xor(sg, o);
                par ^= __shfl_xor(par, o);
            }
            if (act && !syndrome_only) {
                float r = prod * qk_rcp(t);
                r = (r < 1.0f) ? r : 1.0f - 1.1920928955078125e-07f;
                s_val[off + s] = qk_withsign(qk_2atanh(r), sg ^ qk_bits(x));
            }
        }
        any_unsat |= par;
    }
    if (__any(any_unsat != 0) && (tid & 63) == 0) atomicOr(&s_unsat, 1);
    __syncthreads();
    /* phase C: scatter with every lane busy */
    if (!syndrome_only)
        for (int e = tid; e < nE; e += QE_THREADS) cout[s_idx[e] & 0x7fffffff] = s_val[e];
    if (tid == 0 && s_unsat) atomicOr(&unsat[(size_t)f * unsat_stride + ite], 1);
#undef s_unsat
}

template <int S, int FAM>
__global__ __launch_bounds__(QE_THREADS) void qe_cn(const float *__restrict__ v2c, float *__restrict__ c2v,
                                                    const int *__restrict__ cn_ptr, const int *__restrict__ cn_tr,
                                                    const int *__restrict__ cn_var, const uint32_t *__restrict__ sgn,
                                                    int M, int E, int W, int *__restrict__ unsat, int unsat_stride, int ite,
                                                    const int *__restrict__ done_at, qk_rule rule, int syndrome_only, const uint32_t *__restrict__ synd, int Wm)
{
    __shared__ int s_idx[QE_MAX_EDGES];
    __shared__ float s_val[QE_MAX_EDGES];
    __shared__ int s_unsat;
    __shared__ int s_ptr[QE_CPB + 1];
    const int f = blockIdx.y;
    if (!syndrome_only && done_at[f] >= 0) return;      /* the final success-flag pass covers every frame */
    qe_cn_chunk<S, FAM>(f, blockIdx.x, s_idx, s_val, &s_unsat, s_ptr, v2c, c2v, cn_ptr, cn_tr, cn_var, sgn, M, E, W, unsat, unsat_stride, ite, rule, syndrome_only, synd, Wm);
}

/* ------------------------------------------------------------------ variable nodes ----------- */

/*
 * MODE as in the batch engine: QK_VN_FIRST (chk_to_var == 0), QK_VN_NORMAL, QK_VN_POST.
 * Dynamic LDS: 256 * max_dv floats.  The convergence test of the PREVIOUS check pass is evaluated
 * here by every workgroup (same global words, so the same answer); workgroup 0 records it.
 */
/* one chunk of QE_THREADS VNs of frame f; s_msg: QE_THREADS * max_dv floats of LDS; sel: which chk_to_var buffer to read.  Every thread
 * of the workgroup must call it (it synchronises the workgroup). */
template <int MODE, bool COH = false>
__device__ __forceinline__ void qe_vn_chunk(int f, int chunk, float *s_msg, const float *__restrict__ c2v0, const float *__restrict__ c2v1, int sel,
                                            const float *__restrict__ llr, float *__restrict__ v2c,
                                            uint32_t *__restrict__ sgn, uint32_t *__restrict__ hard, float *__restrict__ post_out,
                                            const int *__restrict__ vn_ptr, int N, int E, int W)
{
    const int tid = threadIdx.x;
    const int v0 = chunk * QE_THREADS;
    const int v1 = min(N, v0 + QE_THREADS);
    const int sl0 = vn_ptr[v0], nS = vn_ptr[v1] - sl0;
    const float *cin = (sel ? c2v1 : c2v0) + (size_t)f * E + sl0;
    float *vout = v2c + (size_t)f * E + sl0;
    /* the lane's own VN: its slot range and channel LLR are asked for before the chunk's messages are staged, so that the three loads
     * travel with the staging loads instead of forming two more dependent round trips behind the barrier */
    const int v = v0 + tid, vc = v < v1 ? v : v0;
    const int pb = vn_ptr[vc], pe = vn_ptr[vc + 1];
    const float y = llr[(size_t)f * N + vc];
    if (MODE != QK_VN_FIRST) {
        for (int k = tid; k < nS; k += QE_THREADS) s_msg[k] = qe_ld<COH>(cin + k);
        __syncthreads();
    }
    float tmp = 0.0f;
    if (v < v1) {
        const int b = pb - sl0, deg = pe - pb;
        float sum = 0.0f;
        if (MODE != QK_VN_FIRST) for (int k = 0; k < deg; k++) sum += s_msg[b + k];
        tmp = y + sum;
        if (MODE == QK_VN_FIRST) { const float o = tmp - 0.0f; for (int k = 0; k < deg; k++) s_msg[b + k] = o; }
        else if (MODE == QK_VN_NORMAL) for (int k = 0; k < deg; k++) s_msg[b + k] = tmp - s_msg[b + k];
        if (MODE == QK_VN_POST && post_out) post_out[(size_t)f * N + v] = tmp;
    }
    /* ballots -> MSB-first words: lane l of the wave is bit (31 - l%32) of word l/32 */
    const u64 sb = __ballot(v < v1 && (qk_bits(tmp) >> 31) != 0);
    const u64 hb = __ballot(v < v1 && !(tmp >= 0.0f));
    const int lane = tid & 63;
    if (lane < 2) {
        const int w = (v0 + (tid & ~63)) / 32 + lane;
        if (w < W) {
            sgn[(size_t)f * W + w] = __brev((uint32_t)(sb >> (32 * lane)));
            hard[(size_t)f * W + w] = __brev((uint32_t)(hb >> (32 * lane)));
        }
    }
    if (MODE != QK_VN_POST) {
        __syncthreads();
        for (int k = tid; k < nS; k += QE_THREADS) vout[k] = s_msg[k];
    }
}

template <int MODE>
__global__ __launch_bounds__(QE_THREADS) void qe_vn(const float *__restrict__ c2v0, const float *__restrict__ c2v1, int sel, int n_ite,
                                                    const float *__restrict__ llr, float *__restrict__ v2c,
                                                    uint32_t *__restrict__ sgn, uint32_t *__restrict__ hard, float *__restrict__ post_out,
                                                    const int *__restrict__ vn_ptr, int N, int E, int W,
                                                    const int *__restrict__ unsat, int unsat_stride, int ite, int depth, int check,
                                                    int *__restrict__ done_at, int force)
{
    extern __shared__ __align__(16) float s_msg[];
    const int f = blockIdx.y;
    if (!force) {
        if (done_at[f] >= 0) return;
        if (check && qe_converged(unsat, unsat_stride, f, ite, depth)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) done_at[f] = ite;     /* iterations executed */
            return;
        }
    }
    /* chk_to_var is double-buffered by iteration parity so that the check pass which DETECTS convergence
     * (it runs one iteration ahead) does not clobber the messages the posterior is made of */
    if (sel < 0) sel = ((done_at[f] >= 0 ? done_at[f] : n_ite) - 1) & 1;
    qe_vn_chunk<MODE>(f, blockIdx.x, s_msg, c2v0, c2v1, sel, llr, v2c, sgn, hard, post_out, vn_ptr, N, E, W);
}

/* ------------------------------------------------------------------ one launch per decode ------------ */

/*
 * The whole decode of up to 8 blocks as ONE launch, ONE XCD PER BLOCK.
 *
 * A flooding iteration is two phases with an all-to-all hand-off in between; as launches that costs ~27 us per iteration for one
 * 65 536-VN block, and a grid barrier across XCDs is no cheaper: their L2s are not coherent with each other, so either every word
 * goes through the fabric (sc1 stores / loads: measured 446 us per block against 327 us with a launch per pass) or every barrier
 * pays an L2 write-back and invalidate.  Inside ONE XCD the L2 IS the coherence point: plain stores land there and loads that bypass
 * the CU's L1 (nt) see them.  So each workgroup reads the XCD it runs on (HW_REG_XCC_ID); the XCDs claim the blocks in order of
 * arrival (block f is decoded by the workgroups of one XCD only -- that is read from the hardware, not assumed from blockIdx), up to
 * `nb` workgroups per block stay resident on its 32 CUs, the rest of the grid exits at once.  The phases are the launches of
 * run_edges() -- VN(first) | CN(0) VN(0) | CN(1) VN(1) ... | syndrome of the hard decisions -- separated by a barrier on a counter in
 * that L2: every wave waits for its stores (s_waitcnt vmcnt(0)), workgroup barrier, one lane adds and polls.  The early exit is
 * decided on the device (every workgroup evaluates the same flags after the barrier).  Every wait is bounded: a workgroup that never
 * arrives sets *fault instead of hanging the grid, and the host then decodes with a launch per pass.
 */
#define QE_PERSIST_MAX_FRAMES 8
#define QE_CTL_XCC 0          /* [8]  block claimed by each XCD (-1 free, -2 being claimed) */
#define QE_CTL_NEXT 8         /*      next block to hand out                                */
#define QE_CTL_RANK 16        /* [8]  workgroups that joined block f                        */
#define QE_CTL_BAR 32         /* [8]  barrier counter of block f                            */
#define QE_CTL_FAULT 48
#define QE_CTL_ITERS 49       /*      max over blocks of the check passes executed          */
#define QE_CTL_WORDS 64
#define QE_SPIN_LIMIT (1u << 21)

/* COH: data handed between workgroups of one XCD inside one launch: loads bypass the CU's L1 (served by the XCD's L2) */
template <bool COH> __device__ __forceinline__ float qe_ld(const float *p)
{
    if constexpr (COH) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool COH> __device__ __forceinline__ uint32_t qe_ldu(const uint32_t *p)
{
    if constexpr (COH) return __builtin_nontemporal_load(p);
    else return *p;
}

__device__ __forceinline__ int qe_atomic_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* barrier among the `nb` workgroups of one block (all on one XCD): returns false if it timed out */
__device__ __forceinline__ void qe_xcd_barrier(int *counter, int target, int *fault)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* every wave: its stores have reached the L2 */
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (qe_atomic_ld(counter) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > QE_SPIN_LIMIT) { __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

template <int S, int FAM>
__global__ __launch_bounds__(QE_THREADS) void qe_xcd(float *__restrict__ v2c, float *__restrict__ c2v0, float *__restrict__ c2v1,
                                                     const float *__restrict__ llr, const int *__restrict__ cn_ptr, const int *__restrict__ cn_tr,
                                                     const int *__restrict__ cn_var, const int *__restrict__ vn_ptr,
                                                     uint32_t *__restrict__ sgn, uint32_t *__restrict__ hard,
                                                     int N, int M, int E, int W, int F, int nb, int n_ite, int enable_syndrome, int depth,
                                                     int *__restrict__ unsat, int unsat_stride, int *__restrict__ done_at, qk_rule rule,
                                                     const uint32_t *__restrict__ synd, int Wm, int *__restrict__ ctl)
{
    extern __shared__ __align__(16) float s_dyn[];      /* max(2 * QE_MAX_EDGES + 16 + QE_CPB words, QE_THREADS * max_dv floats) */
    __shared__ int s_role[2];                           /* block index, rank within the block's workgroups (-1: leave) */
    __shared__ int s_done;
    int *s_idx = reinterpret_cast<int *>(s_dyn);
    float *s_val = s_dyn + QE_MAX_EDGES;
    int *s_unsat = reinterpret_cast<int *>(s_dyn + 2 * QE_MAX_EDGES);
    int *s_ptr = s_unsat + 8;                           /* QE_CPB + 1 ints */
    int *fault = ctl + QE_CTL_FAULT;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        int f = -1;
        int seen = __hip_atomic_load(ctl + QE_CTL_XCC + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == -1) {
            int expect = -1;
            if (__hip_atomic_compare_exchange_strong(ctl + QE_CTL_XCC + xcc, &expect, -2, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                f = __hip_atomic_fetch_add(ctl + QE_CTL_NEXT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* this XCD takes the next block */
                if (f >= F) f = 1 << 20;
                __hip_atomic_store(ctl + QE_CTL_XCC + xcc, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else seen = expect;
        }
        if (f < 0) {
            unsigned spins = 0;
            while ((f = (seen >= 0 ? seen : qe_atomic_ld(ctl + QE_CTL_XCC + xcc))) < 0) { seen = -2; __builtin_amdgcn_s_sleep(1); if (++spins > QE_SPIN_LIMIT) { f = 1 << 20; break; } }
        }
        int rank = -1;
        if (f < F) {
            rank = __hip_atomic_fetch_add(ctl + QE_CTL_RANK + f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rank >= nb) rank = -1;
        }
        s_role[0] = f; s_role[1] = rank;
        s_done = -1;
    }
    __syncthreads();
    const int f = s_role[0], me = s_role[1];
    if (me < 0) return;
    int *bar = ctl + QE_CTL_BAR + f;
    int round = 0;
    const int nCN = (M + QE_CPB - 1) / QE_CPB, nVN = (N + QE_THREADS - 1) / QE_THREADS;
    /* decoder reset + iteration 0's variable-node pass (chk_to_var == 0) */
    for (int i = me * QE_THREADS + threadIdx.x; i < unsat_stride; i += nb * QE_THREADS) __hip_atomic_store(unsat + (size_t)f * unsat_stride + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int it = me; it < nVN; it += nb) { qe_vn_chunk<QK_VN_FIRST, true>(f, it, s_dyn, c2v0, c2v1, 0, llr, v2c, sgn, hard, nullptr, vn_ptr, N, E, W); __syncthreads(); }
    qe_xcd_barrier(bar, ++round * nb, fault);
    int ite = 0;
    for (; ite < n_ite; ite++) {
        for (int it = me; it < nCN; it += nb) {
            qe_cn_chunk<S, FAM, true>(f, it, s_idx, s_val, s_unsat, s_ptr, v2c, (ite & 1) ? c2v1 : c2v0, cn_ptr, cn_tr, cn_var, sgn, M, E, W, unsat, unsat_stride, ite, rule, 0, synd, Wm);
            __syncthreads();
        }
        qe_xcd_barrier(bar, ++round * nb, fault);
        /* AFF3CT's stop rule on the flags the check pass left: the same words for every workgroup of the block */
        if (threadIdx.x == 0 && enable_syndrome && ite >= depth) {
            bool conv = true;
            for (int k = 0; k < depth; k++) conv = conv && qe_atomic_ld(unsat + (size_t)f * unsat_stride + ite - k) == 0;
            if (conv) s_done = ite;      /* iterations executed */
        }
        __syncthreads();
        if (s_done >= 0 || qe_atomic_ld(fault)) break;
        const bool last = ite == n_ite - 1;
        for (int it = me; it < nVN; it += nb) {
            if (last) qe_vn_chunk<QK_VN_POST, true>(f, it, s_dyn, c2v0, c2v1, ite & 1, llr, v2c, sgn, hard, nullptr, vn_ptr, N, E, W);
            else qe_vn_chunk<QK_VN_NORMAL, true>(f, it, s_dyn, c2v0, c2v1, ite & 1, llr, v2c, sgn, hard, nullptr, vn_ptr, N, E, W);
            __syncthreads();
        }
        qe_xcd_barrier(bar, ++round * nb, fault);
    }
    /* success flag: syndrome of the final hard decisions into the last slot; iteration count */
    for (int it = me; it < nCN; it += nb) {
        qe_cn_chunk<S, FAM, true>(f, it, s_idx, s_val, s_unsat, s_ptr, v2c, c2v0, cn_ptr, cn_tr, cn_var, hard, M, E, W, unsat, unsat_stride, n_ite + 1, rule, 1, synd, Wm);
        __syncthreads();
    }
    if (me == 0 && threadIdx.x == 0) {
        done_at[f] = s_done;
        atomicMax(ctl + QE_CTL_ITERS, ite < n_ite ? ite + 1 : n_ite);
    }
}

/* ------------------------------------------------------------------ load / fetch / status ---- */

__global__ void qe_init(int *__restrict__ unsat, int unsat_stride, int *__restrict__ done_at, int F)
{
    const int f = blockIdx.x;
    for (int k = threadIdx.x; k < unsat_stride; k += blockDim.x) unsat[(size_t)f * unsat_stride + k] = 0;
    if (threadIdx.x == 0) done_at[f] = -1;
    (void)F;
}

/* QKD frame formation for frame-major floats: bits[F][W] + |LLR| per frame + class per VN -> llr[F][N] */
__global__ void qe_load_bits(const uint32_t *__restrict__ bits, const float *__restrict__ llr_mag, const uint8_t *__restrict__ vn_class,
                             float *__restrict__ llr, int N, int W, const int *__restrict__ n_channel)
{
    const int f = blockIdx.y;
    const float mag = llr_mag[f];
    const int nch = n_channel ? n_channel[f] : N;      /* channel VNs at v >= nch are known (shortened) bits of this frame */
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        const bool y = (bits[(size_t)f * W + (v >> 5)] >> (31 - (v & 31))) & 1u;
        int cls = vn_class ? vn_class[v] : 0;
        if (cls == 0 && v >= nch) cls = 1;
        const float m = (cls == 0) ? mag : (cls == 1 ? 23.025850929840455f : 0.0f);
        llr[(size_t)f * N + v] = y ? -m : m;
    }
}

/* per-frame puncturing for the frame-major floats: erase[F][W] packed MSB-first, a set bit zeroes that LLR */
__global__ void qe_erase(const uint32_t *__restrict__ erase, float *__restrict__ llr, int N, int W)
{
    const int f = blockIdx.y;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x)
        if ((erase[(size_t)f * W + (v >> 5)] >> (31 - (v & 31))) & 1u) llr[(size_t)f * N + v] = 0.0f;
}

__global__ void qe_fetch_info(const uint32_t *__restrict__ hard, const int *__restrict__ info_pos, int *__restrict__ out, int K, int W)
{
    const int f = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        const int v = info_pos[i];
        out[(size_t)f * K + i] = (int)((hard[(size_t)f * W + (v >> 5)] >> (31 - (v & 31))) & 1u);
    }
}

/* iters / ok per frame: ok = syndrome of the final HARD bits, left in unsat[f][final_slot] by a syndrome-only pass */
__global__ void qe_status_out(const int *__restrict__ unsat, int unsat_stride, int final_slot, const int *__restrict__ done_at, int n_ite,
                              int *__restrict__ out_iters, int *__restrict__ out_ok, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    if (out_iters) out_iters[f] = done_at[f] >= 0 ? done_at[f] : n_ite;
    if (out_ok) out_ok[f] = unsat[(size_t)f * unsat_stride + final_slot] == 0;
}

#endif /* QLDPC_KERNELS_EDGE_H */
