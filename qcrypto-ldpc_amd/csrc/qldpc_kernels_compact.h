/*
 * qldpc_kernels_compact.h -- active-frame compaction of the FRAMES engine's early-exit mode ("converged
 * frames compacted every k iterations so waves stay full").
 *
 * With enable_syndrome a 64*V-frame group runs until its slowest frame has converged (AFF3CT decodes frame by frame and
 * simply stops each one, Decoder_LDPC_BP_flooding::_decode; here a converged frame keeps occupying its lane).  Once few
 * enough frames are still active they are dealt into fewer, full groups -- a new GENERATION of the per-frame state:
 *
 *   plan     : src[d] = slot (frame position g*FG + lane*V + j) in the old generation that moves to slot d of the new one,
 *              active frames in slot order (qk_compact_count / _scan / _scatter / _fill); origin[d] = index of the frame
 *              in the caller's batch; the small per-frame arrays (|LLR|, shortening length, syndrome depth) move with it
 *   ballots  : the received-bit ballots (coded LLRs) and the target-syndrome ballots are re-dealt bit by bit
 *   rows     : an LLR array (float or 8-bit), when the decoder reads one, is gathered into the new layout
 *   messages : NOT copied.  The next check-node pass reads var_to_chk through src[] (REMAP variants of the check kernels:
 *              a per-lane base pointer into the old layout) and writes chk_to_var in the new layout; the variable-node pass
 *              after it writes var_to_chk in the new layout over the old one.  So a compaction costs one check pass that
 *              fetches the rows of all old groups, and nothing else on the message arrays.
 *
 * The old generation keeps its ballots / done bits / iteration counts: frames that converged there are read from there
 * by the fetch calls (origin[] says where they belong in the caller's order).  Decisions, iteration counts and success
 * flags are those of the uncompacted run bit for bit (tests/test_compaction_gpu.py, tests/fuzz_parity.py).
 */
#ifndef QLDPC_KERNELS_COMPACT_H
#define QLDPC_KERNELS_COMPACT_H

#include "qldpc_kernels.h"

/* gcount[g] = frames of group g still active (done is [G][V], bit = lane) */
template <int V>
__global__ __launch_bounds__(256) void qk_compact_count(const u64 *__restrict__ done, int *__restrict__ gcount, int G)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    int n = 0;
#pragma unroll
    for (int j = 0; j < V; j++) n += __popcll(~done[(size_t)g * V + j]);
    gcount[g] = n;
}

/* exclusive prefix sum of gcount into goff (one workgroup; G is a few thousand at most); goff[G] = total */
static __global__ __launch_bounds__(256) void qk_compact_scan(const int *__restrict__ gcount, int *__restrict__ goff, int G)
{
    __shared__ int part[256];
    const int t = threadIdx.x;
    const int per = (G + 255) / 256;
    const int lo = t * per, hi = min(G, lo + per);
    int s = 0;
    for (int g = lo; g < hi; g++) s += gcount[g];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { const int x = part[i]; part[i] = acc; acc += x; }
        goff[G] = acc;
    }
    __syncthreads();
    int acc = part[t];
    for (int g = lo; g < hi; g++) { goff[g] = acc; acc += gcount[g]; }
}

/* one wavefront per old group: every still-active frame learns its new slot (slot order is kept) */
template <int V>
__global__ __launch_bounds__(64) void qk_compact_scatter(const u64 *__restrict__ done, const int *__restrict__ goff, int *__restrict__ src)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.x, lane = threadIdx.x;
    u64 act[V];
    int before = 0;      /* active frames of this group in lanes below mine */
#pragma unroll
    for (int j = 0; j < V; j++) { act[j] = ~done[(size_t)g * V + j]; before += __popcll(act[j] & ((1ull << lane) - 1ull)); }
    int d = goff[g] + before;
#pragma unroll
    for (int j = 0; j < V; j++)
        if ((act[j] >> lane) & 1ull) src[d++] = g * FG + lane * V + j;
}

/*
 * per new slot: padding (d >= total) gets src = -1 and its done bit set; origin / |LLR| / shortening / depth follow the frame;
 * iteration counts start at n_ite (the value of a frame that never converges), unsat words are cleared.
 * origin_old == NULL: the old generation is the caller's order itself (slot f = frame f).
 */
template <int V>
__global__ __launch_bounds__(64) void qk_compact_fill(int *__restrict__ src, const int *__restrict__ goff_total, const int *__restrict__ origin_old,
                                                      int *__restrict__ origin_new, const float *__restrict__ fmag_old, float *__restrict__ fmag_new,
                                                      const int *__restrict__ fnch_old, int *__restrict__ fnch_new, const int *__restrict__ depth_old,
                                                      int *__restrict__ depth_new, int *__restrict__ iters_new, u64 *__restrict__ done_new,
                                                      u64 *__restrict__ unsat_new, int n_ite, int N)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.x, lane = threadIdx.x;
    const int total = *goff_total;
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int d = g * FG + lane * V + j;
        const bool pad = d >= total;
        const int s = pad ? -1 : src[d];
        if (pad) src[d] = -1;
        origin_new[d] = pad ? -1 : (origin_old ? origin_old[s] : s);
        if (fmag_new) { fmag_new[d] = pad ? 1.0f : fmag_old[s]; fnch_new[d] = pad ? N : fnch_old[s]; }
        depth_new[d] = pad ? 0 : depth_old[s];
        iters_new[d] = n_ite;
        const u64 padmask = __ballot(pad);
        if (lane == 0) { done_new[(size_t)g * V + j] = padmask; unsat_new[(size_t)g * V + j] = 0; }
    }
}

/*
 * ballot arrays [G][n][V] (bit = lane): new word (g', i, j') collects, for each lane, the bit of the frame that moved there.
 * One wavefront does 64 consecutive items i of one new group; all its words leave in one store per j'.
 */
template <int V>
__global__ __launch_bounds__(QK_THREADS) void qk_compact_ballots(const u64 *__restrict__ old_b, u64 *__restrict__ new_b, const int *__restrict__ src, int n)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = (blockIdx.x * QK_WAVES + wave) * 64;
    if (i0 >= n) return;
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int s = src[(size_t)g * FG + lane * V + j];
        const bool live = s >= 0;
        const int sg = live ? s / FG : 0, sl = live ? (s % FG) / V : 0, sj = live ? s % V : 0;
        const u64 *from = old_b + (size_t)sg * n * V + sj;
        u64 mine = 0;
        for (int t = 0; t < 64; t++) {
            const int i = i0 + t;
            if (i >= n) break;      /* wave-uniform */
            const u64 w = from[(size_t)i * V];
            const u64 b = __ballot(live && ((w >> sl) & 1ull));
            if (lane == t) mine = b;
        }
        if (i0 + lane < n) new_b[((size_t)g * n + i0 + lane) * V + j] = mine;
    }
}

/* element rows [G][n][FG] of T (channel LLRs: float, or the 8-bit quantised form with T = uint8_t) gathered into the new layout */
template <int V, typename T>
__global__ __launch_bounds__(QK_THREADS) void qk_compact_rows(const T *__restrict__ old_r, T *__restrict__ new_r, const int *__restrict__ src, int n, T pad)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T *from[V];
    bool live[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int s = src[(size_t)g * FG + lane * V + j];
        live[j] = s >= 0;
        from[j] = old_r + (live[j] ? (size_t)(s / FG) * n * FG + (s % FG) : 0);
    }
    for (int i = blockIdx.x * QK_WAVES + wave; i < n; i += gridDim.x * QK_WAVES) {
        T *to = new_r + ((size_t)g * n + i) * FG + lane * V;
#pragma unroll
        for (int j = 0; j < V; j++) to[j] = live[j] ? from[j][(size_t)i * FG] : pad;
    }
}

#endif /* QLDPC_KERNELS_COMPACT_H */
