/*
 * qldpc_kernels_h16.h -- packed check-node kernel for the binary16 message variant (msg_dtype = 1, two frames per lane).
 *
 * The generic kernel (qk_cn_flood<2, ., QK_FAM_MS, __half>) widens every message to fp32 and is VALU-bound at 4.1 TB/s
 * of the halved bytes.  For the min-sum family nothing in the fold needs arithmetic on the values: |x|, min1 / min2 and
 * "is this the minimum" are order comparisons, and non-negative binary16 numbers order like their bit patterns.  So
 * the two halves of a lane's dword stay packed and the fold runs on v_pk_{min,max,sub,mad}_u16 (11 VALU instructions
 * per edge for two frames); only the rule (min * factor, min - offset) is evaluated in fp32 and rounded to nearest
 * even, once per check, exactly as the generic kernel does per edge.  Bit-identical results
 * (tests/test_parity_gpu.py::test_fp16_*, which compare against the rounding oracle).
 */
#ifndef QLDPC_KERNELS_H16_H
#define QLDPC_KERNELS_H16_H

#include "qldpc_kernels_i8.h"

/* rule on a non-negative binary16 bit pattern: fp32 arithmetic, RNE back to binary16 (update rules MS / OMS / NMS) */
__device__ __forceinline__ unsigned qh_rule1(unsigned mbits, const qk_rule &r)
{
    const float m = (mbits == 0x7c00u) ? 3.402823466e+38f : __half2float(__ushort_as_half((unsigned short)mbits));   /* the fold starts at FLT_MAX */
    float c;
    if (r.rule == 0) c = qk_max(0.0f, m);
    else if (r.rule == 1) c = qk_max(0.0f, m - r.param);
    else c = m * r.param;
    return (unsigned)__half_as_ushort(__float2half_rn(c));
}
__device__ __forceinline__ qi_u2 qh_rule(qi_u2 m, const qk_rule &r)
{
    qi_u2 o;
    o.x = (unsigned short)qh_rule1(m.x, r);
    o.y = (unsigned short)qh_rule1(m.y, r);
    return o;
}

template <int DCMAX>
__global__ __launch_bounds__(QK_THREADS) void qh_cn_flood(const uint32_t *__restrict__ v2c, uint32_t *__restrict__ c2v,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ cn_ptr, const int *__restrict__ cn_tr,
                                                          size_t group_stride /* dwords */, const u64 *__restrict__ done, qk_rule rule,
                                                          const u64 *__restrict__ synd, int M)
{
    static_assert(DCMAX > 0, "register-resident degrees only");
    const int g = blockIdx.y;
    if (qk_group_done<2>(done, g)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * QK_WAVES + wave;
    if (i >= n_list) return;
    const uint32_t *vin = v2c + (size_t)g * group_stride + lane;
    uint32_t *cout = c2v + (size_t)g * group_stride + lane;
    const int c = list[i];
    const int b = cn_ptr[c];
    const int deg = cn_ptr[c + 1] - b;
    uint32_t sg = 0;
    if (synd) sg = ((uint32_t)((synd[((size_t)g * M + c) * 2 + 0] >> lane) & 1ull) << 15) | ((uint32_t)((synd[((size_t)g * M + c) * 2 + 1] >> lane) & 1ull) << 31);
    qi_u2 m1 = qi_as_u2(qi_splat(0x7c00)), m2 = m1;
    int slot[DCMAX];
#pragma unroll
    for (int k = 0; k < DCMAX; k++) slot[k] = cn_tr[b + k];
    uint32_t x[DCMAX];
#pragma unroll
    for (int k = 0; k < DCMAX; k++)
        if (k < deg) x[k] = qi_ldm(vin + (size_t)slot[k] * 64);
#pragma unroll
    for (int k = 0; k < DCMAX; k++)
        if (k < deg) {
            const qi_u2 a = __builtin_bit_cast(qi_u2, x[k] & 0x7fff7fffu);
            sg ^= x[k];
            const qi_u2 t = __builtin_elementwise_min(a, m2);
            m2 = __builtin_elementwise_max(t, m1);
            m1 = __builtin_elementwise_min(t, m1);
        }
    /* cst2 goes to every edge, cst1 to the edge(s) holding the minimum (Update_rule_*::compute_chk_node_out) */
    const qi_u2 n1 = qh_rule(m1, rule);
    const qi_u2 dn = qh_rule(m2, rule) - n1;
    const qi_u2 one = qi_as_u2(qi_splat(1));
#pragma unroll
    for (int k = 0; k < DCMAX; k++)
        if (k < deg) {
            const qi_u2 a = __builtin_bit_cast(qi_u2, x[k] & 0x7fff7fffu);
            const qi_u2 ind = __builtin_elementwise_min(__builtin_elementwise_sub_sat(m2, a), one);      /* 1 iff a < min2, i.e. a == min1 < min2 */
            const uint32_t mag = __builtin_bit_cast(uint32_t, (qi_u2)(n1 + (qi_u2)(ind * dn)));
            qi_stm(cout + (size_t)(b + k) * 64, mag | ((sg ^ x[k]) & 0x80008000u));      /* chk_to_var is CN-major */
        }
}

#endif /* QLDPC_KERNELS_H16_H */
