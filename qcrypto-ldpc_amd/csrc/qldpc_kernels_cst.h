/*
 * qldpc_kernels_cst.h -- horizontal-layered min-sum sweeps on a COMPRESSED check state (fp32, 64-frame groups).
 *
 * In the layered schedule (Decoder_LDPC_BP_horizontal_layered::_decode_single_ite; ML/BPSK_nrldpc_sim.m:29-69 is the same
 * recursion) a check's dc messages are private to that check: written at the end of its update, read back at the start of its
 * next one, by nobody else.  For the min-sum family (MS / OMS / NMS: tools::Update_rule_{MS,OMS,NMS}) and the AMS rules
 * (tools::Update_rule_AMS: |contribution_k| == min ? delta_min : delta) those dc floats per frame take only two magnitudes,
 *     messages[k] = +-( |contribution_k| == min1 ? cst1 : cst2 ),
 * so the check keeps, per frame, {cst1, cst2} and two dc-bit masks -- bit k of `took1`: edge k took cst1; bit k of `neg`: the
 * message of edge k is negative -- i.e. FOUR 256-byte rows per check and 64-frame group instead of dc, and rebuilds the very
 * float it would have read: same magnitude bits, same sign bit (a -0.0 stays a -0.0).  contribution = var_nodes - message, the
 * fold and var_nodes = contribution + message are untouched, so the sweep is bit-identical to the one that stores every message
 * (and to the oracle); on a tie min2 == min1 both forms give cst1.
 *
 * HBM bytes of a sweep per 64-frame group: 2 E rows (posteriors read + written) + 8 M rows (state read + written) against 4 E
 * rows with explicit messages: 0.61 x on the N = 10^6 code of BASELINE configs[4] (dc = 18).  Unlike the flooding form of this
 * idea (the VN pass has to GATHER the state rows of its checks, which is fabric-bound), nothing but the
 * owner ever touches a check's state here: its four rows are one contiguous kilobyte.
 * (Masks per FRAME rather than ballots per EDGE: a lane needs bit `lane` of dc wave-uniform words for the ballot form, i.e. a
 * 64 x dc bit transpose through v_readlane / v_writelane on the way in and out; the per-frame masks need none and cost 224 bytes
 * more per check.)
 *
 * The state lives in the message array the decoder owns anyway ([G][E][64] floats per group): rows of check c at c * 256
 * floats -- {cst1, cst2, neg, took1}[64]; M * 1024 <= E * 256 bytes is checked by the host, check degree <= 32.
 * Sweep 0 takes the state as zero (messages +0.0) without reading it, as qk_cn_layer does.  freeze_messages = 0 only.
 */
#ifndef QLDPC_KERNELS_CST_H
#define QLDPC_KERNELS_CST_H

#include "qldpc_kernels.h"

#define QK_CST_ROWS 4      /* {cst1, cst2, neg, took1} */

/* the message of edge k for this lane's frame, rebuilt from the check's state */
__device__ __forceinline__ float qk_cst_msg(uint32_t neg, uint32_t took1, int k, float c1, float c2)
{
    return qk_withsign(((took1 >> k) & 1u) ? c1 : c2, (neg >> k) << 31);
}

/* the two magnitudes of a finished check and the test that tells which one an edge took, per rule family */
template <int FAM> struct qk_cst_of;
template <> struct qk_cst_of<QK_FAM_MS> {
    static __device__ __forceinline__ float c1(const qk_acc<QK_FAM_MS> &a) { return a.cst1; }
    static __device__ __forceinline__ float c2(const qk_acc<QK_FAM_MS> &a) { return a.cst2; }
    static __device__ __forceinline__ bool took1(const qk_acc<QK_FAM_MS> &a, float x) { return fabsf(x) == a.min1; }
};
template <> struct qk_cst_of<QK_FAM_AMS> {
    static __device__ __forceinline__ float c1(const qk_acc<QK_FAM_AMS> &a) { return a.delta_min; }
    static __device__ __forceinline__ float c2(const qk_acc<QK_FAM_AMS> &a) { return a.delta; }
    static __device__ __forceinline__ bool took1(const qk_acc<QK_FAM_AMS> &a, float x) { return fabsf(x) == a.mn; }
};

template <int DCMAX, int FAM>
__global__ __launch_bounds__(QK_THREADS) void qk_cn_layer_cst(float *__restrict__ post, float *__restrict__ st,
                                                              const int *__restrict__ list, int n_list,
                                                              const int *__restrict__ cn_ptr, const int *__restrict__ cn_var,
                                                              int N, size_t group_stride, const u64 *__restrict__ done, qk_rule rule, const u64 *__restrict__ synd, int M,
                                                              int first, const int *__restrict__ rec, int rec_stride)
{
    static_assert(DCMAX > 0 && DCMAX <= 32, "one mask bit per edge");
    static_assert(FAM == QK_FAM_MS || FAM == QK_FAM_AMS, "rules whose messages take two magnitudes per check");
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * QK_WAVES + wave;
    if (i >= n_list) return;
    float *pg = post + (size_t)g * N * 64 + lane;

    /* what this wave works on: one aligned record per list entry (bucket::d_rec) asked for together with the group's done word -- one
     * scalar round trip in front of the row loads instead of done -> list -> cn_ptr -> cn_var */
    int c, deg, vn[DCMAX];
    if (rec) {
        const int *r = static_cast<const int *>(__builtin_assume_aligned(rec + (size_t)i * rec_stride, 16));
        c = r[0]; deg = r[2];
#pragma unroll
        for (int k = 0; k < DCMAX; k++) vn[k] = r[QK_REC_HDR + k];
        const u64 dn = done[g];
        /* keeps the record loads above the early return (the compiler otherwise sinks them behind the done test: two round trips again) */
        asm volatile("" ::"s"(c), "s"(deg), "s"(vn[0]), "s"(vn[DCMAX / 2]), "s"(vn[DCMAX - 1]), "s"(dn));
        if (dn == ~0ull) return;
    } else {
        if (qk_group_done<1>(done, g)) return;
        c = list[i];
        const int b = cn_ptr[c];
        deg = cn_ptr[c + 1] - b;
#pragma unroll
        for (int k = 0; k < DCMAX; k++) vn[k] = cn_var[b + k];
    }
    float *crow = st + (size_t)g * group_stride + (size_t)c * (64 * QK_CST_ROWS) + lane;
    float x[DCMAX];
#pragma unroll
    for (int k = 0; k < DCMAX; k++)
        if (k < deg) x[k] = pg[(size_t)vn[k] * 64];
    float c1 = 0.0f, c2 = 0.0f;
    uint32_t neg = 0u, took1 = 0u;
    if (!first) {
        c1 = qk_ldm1(crow); c2 = qk_ldm1(crow + 64);
        neg = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(crow + 128)); took1 = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(crow + 192));
    }
    qk_acc<FAM> acc;
    acc.begin();
    if (synd) acc.sign = (uint32_t)((synd[(size_t)g * M + c] >> lane) & 1ull) << 31;
#pragma unroll
    for (int k = 0; k < DCMAX; k++)
        if (k < deg) { x[k] = x[k] - qk_cst_msg(neg, took1, k, c1, c2); qk_acc_in<FAM>(acc, x[k], rule); }
    acc.finish(rule);
    neg = 0u; took1 = 0u;
#pragma unroll
    for (int k = 0; k < DCMAX; k++)
        if (k < deg) {
            const float o = acc.out(x[k], rule);
            pg[(size_t)vn[k] * 64] = x[k] + o;      /* posteriors are re-read by later layers: cached */
            neg |= (qk_bits(o) >> 31) << k;
            took1 |= (qk_cst_of<FAM>::took1(acc, x[k]) ? 1u : 0u) << k;
        }
    __builtin_nontemporal_store(qk_cst_of<FAM>::c1(acc), crow);
    __builtin_nontemporal_store(qk_cst_of<FAM>::c2(acc), crow + 64);
    __builtin_nontemporal_store(neg, reinterpret_cast<uint32_t *>(crow + 128));
    __builtin_nontemporal_store(took1, reinterpret_cast<uint32_t *>(crow + 192));
}

#endif /* QLDPC_KERNELS_CST_H */
