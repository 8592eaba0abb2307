/*
 * qldpc_kernels_i8.h -- 8-bit fixed-point variant of the flooding min-sum kernels (qldpc_decoder_cfg.msg_dtype = 2).
 *
 * Same frame-interleaved layout as qldpc_kernels.h with FOUR frames per wavefront lane (V = 4, FG = 256): a lane's
 * four messages of one edge are the four bytes of one dword, so a row is still 256 bytes and every HBM byte carries a
 * message.  A quarter of the fp32 bytes per iteration ("int8 quarters A_iter").
 *
 * Arithmetic (integer, therefore order-free and bit-exact against oracle/qldpc_oracle.c:decode_flooding_i8):
 *   Yq        = clamp(rint(LLR * quant_scale), -127, 127)                      (qi_quant_llr)
 *   tmp       = Yq + sum of chk_to_var over the VN's slots                     (16-bit, cannot overflow for dv <= 256)
 *   var_to_chk= clamp(tmp - chk_to_var, -127, 127)
 *   check fold: min1 / min2 of |var_to_chk|, sign = xor of (var_to_chk < 0) [xor target syndrome];
 *               MS: n = m; OMS: n = max(0, m - rint(offset * quant_scale)); NMS: n = (m * rint(factor * 128)) >> 7
 *   chk_to_var= +-(|var_to_chk| == min1 ? n(min2) : n(min1))
 *   decision  = tmp < 0 (an integer has no -0, so signbit(post) and !(post >= 0) coincide)
 * The structure follows the saturating fixed-point min-sum the reference keeps in MATLAB
 * (ldpc_examples/.../BPSK_nrldpc_sim_RM_FP.m:37-98: quantise, subtract, saturate, min1/min2/parity, offset, saturate),
 * carried over to AFF3CT's flooding schedule (and, as qi_cn_layer, to its horizontal-layered one with the posterior
 * itself kept in 8 bits, as the MATLAB decoder does).  FER-tolerance class against the float decoder.
 *
 * The four bytes are widened to two packed-int16 registers (v_perm_b32 with its sign-extension selectors) and all
 * arithmetic is v_pk_*_{i16,u16}: two frames per VALU lane-operation, which keeps the kernel HBM-bound.
 */
#ifndef QLDPC_KERNELS_I8_H
#define QLDPC_KERNELS_I8_H

#include "qldpc_kernels.h"

typedef short qi_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short qi_u2 __attribute__((ext_vector_type(2)));

#define QI_V 4
#define QI_FG 256

__device__ __forceinline__ qi_s2 qi_as_s2(uint32_t w) { return __builtin_bit_cast(qi_s2, w); }
__device__ __forceinline__ qi_u2 qi_as_u2(qi_s2 x) { return __builtin_bit_cast(qi_u2, x); }
__device__ __forceinline__ qi_s2 qi_as_s2(qi_u2 x) { return __builtin_bit_cast(qi_s2, x); }
__device__ __forceinline__ uint32_t qi_as_u32(qi_s2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ qi_s2 qi_splat(int v) { qi_s2 r; r.x = (short)v; r.y = (short)v; return r; }

/* bytes {b0,b1,b2,b3} of w (signed) -> lo = {b0,b1}, hi = {b2,b3} as packed int16.  v_perm_b32 picks bytes out of
 * {S0 = w << 8 (bytes 4..7), S1 = w (bytes 0..3)}; selectors 8..11 give the sign of bytes 1, 3, 5, 7 replicated. */
__device__ __forceinline__ void qi_unpack(uint32_t w, qi_s2 &lo, qi_s2 &hi)
{
    const uint32_t s = w << 8;
    lo = qi_as_s2(__builtin_amdgcn_perm(s, w, 0x08010a00u));      /* b0, sign(b0 = byte 5), b1, sign(b1 = byte 1) */
    hi = qi_as_s2(__builtin_amdgcn_perm(s, w, 0x09030b02u));      /* b2, sign(b2 = byte 7), b3, sign(b3 = byte 3) */
}
__device__ __forceinline__ uint32_t qi_pack(qi_s2 lo, qi_s2 hi)
{
    return __builtin_amdgcn_perm(qi_as_u32(hi), qi_as_u32(lo), 0x06040200u);
}
__device__ __forceinline__ uint32_t qi_ldm(const uint32_t *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void qi_stm(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ qi_s2 qi_abs(qi_s2 x) { return __builtin_elementwise_max(x, (qi_s2)(-x)); }
__device__ __forceinline__ qi_s2 qi_clamp127(qi_s2 x) { return __builtin_elementwise_max(__builtin_elementwise_min(x, qi_splat(127)), qi_splat(-127)); }

struct qi_rule {
    int rule;       /* 0 MS, 1 OMS, 2 NMS */
    int param;      /* OMS: offset in quantiser steps; NMS: factor * 128 */
};

/* magnitudes after the rule, two frames at once (0 <= m <= 127) */
__device__ __forceinline__ qi_s2 qi_norm(qi_s2 m, const qi_rule &r)
{
    if (r.rule == 0) return m;
    const qi_u2 p = qi_as_u2(qi_splat(r.param));
    if (r.rule == 1) return qi_as_s2(__builtin_elementwise_sub_sat(qi_as_u2(m), p));
    return qi_as_s2((qi_u2)((qi_u2)(qi_as_u2(m) * p) >> 7));
}

/* one pair of frames of one check */
struct qi_acc {
    qi_s2 sg, m1, m2, n1, dn;
    __device__ __forceinline__ void begin(unsigned s0, unsigned s1, int mag_max = 127)
    {
        sg.x = (short)(s0 << 15); sg.y = (short)(s1 << 15);
        m1 = qi_splat(mag_max); m2 = qi_splat(mag_max);
    }
    __device__ __forceinline__ void in(qi_s2 x)
    {
        const qi_s2 a = qi_abs(x);
        sg ^= x;
        const qi_s2 t = __builtin_elementwise_min(a, m2);
        m2 = __builtin_elementwise_max(t, m1);
        m1 = __builtin_elementwise_min(t, m1);
    }
    __device__ __forceinline__ void finish(const qi_rule &r)
    {
        n1 = qi_norm(m1, r);
        dn = qi_norm(m2, r) - n1;
    }
    __device__ __forceinline__ qi_s2 out(qi_s2 x) const
    {
        const qi_s2 a = qi_abs(x);
        /* ind = 1 iff a < m2, i.e. this edge holds the unique minimum; a is either == m1 or >= m2 */
        const qi_u2 ind = __builtin_elementwise_min(__builtin_elementwise_sub_sat(qi_as_u2(m2), qi_as_u2(a)), qi_as_u2(qi_splat(1)));
        const qi_s2 mag = n1 + qi_as_s2((qi_u2)(ind * qi_as_u2(dn)));
        const qi_s2 s = (qi_s2)(sg ^ x) >> 15;                 /* 0 or -1 */
        return (qi_s2)(mag ^ s) - s;
    }
};

/* ------------------------------------------------------------------ check nodes -------------- */

/* FIRST: the check pass of iteration 0 reads the quantised channel LLRs of its VNs (what the first variable-node pass would
 * have copied into var_to_chk: |Yq| <= 127 needs no clamp), so that pass is not run */
/* REMAP: the check pass right after a compaction (see qk_cn_flood): each of the lane's four frames reads its byte from the slot
 * of the old layout it came from */
template <int DCMAX, bool FIRST = false, bool REMAP = false>
__global__ __launch_bounds__(QK_THREADS) void qi_cn_flood(const uint32_t *__restrict__ v2c, uint32_t *__restrict__ c2v,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ cn_ptr, const int *__restrict__ cn_tr,
                                                          size_t group_stride /* dwords */, const u64 *__restrict__ done, qi_rule rule,
                                                          const u64 *__restrict__ synd, int M,
                                                          const uint32_t *__restrict__ llr8 = nullptr, const int *__restrict__ cn_var = nullptr, int N = 0,
                                                          const int *__restrict__ remap_src = nullptr)
{
    static_assert(!(FIRST && REMAP), "no compaction before the first check pass");
    const int g = blockIdx.y;
    if (qk_group_done<QI_V>(done, g)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * QK_WAVES + wave;
    if (i >= n_list) return;
    const uint32_t *vin = v2c + (size_t)g * group_stride + lane;
    uint32_t *cout = c2v + (size_t)g * group_stride + lane;
    const uint8_t *vin_r[QI_V];      /* REMAP only */
    if constexpr (REMAP) {
#pragma unroll
        for (int j = 0; j < QI_V; j++) {
            int s = remap_src[(size_t)g * QI_FG + lane * QI_V + j];
            s = s < 0 ? 0 : s;
            vin_r[j] = reinterpret_cast<const uint8_t *>(v2c + (size_t)(s / QI_FG) * group_stride) + (s % QI_FG);
        }
    }
    auto remap_word = [&](size_t slot) {      /* the four bytes of this lane's dword, each from its own source row */
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < QI_V; j++) w |= (uint32_t)__builtin_nontemporal_load(vin_r[j] + slot * QI_FG) << (8 * j);
        return w;
    };
    const int c = list[i];
    const int b = cn_ptr[c];
    const int deg = cn_ptr[c + 1] - b;
    qi_acc lo, hi;
    {
        unsigned s[QI_V] = {0, 0, 0, 0};
        if (synd) {
#pragma unroll
            for (int j = 0; j < QI_V; j++) s[j] = (unsigned)((synd[((size_t)g * M + c) * QI_V + j] >> lane) & 1ull);
        }
        lo.begin(s[0], s[1]); hi.begin(s[2], s[3]);
    }
    if constexpr (DCMAX > 0) {
        int slot[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++) slot[k] = cn_tr[b + k];
        uint32_t w[DCMAX];
        if constexpr (FIRST) {
            int vid[DCMAX];
#pragma unroll
            for (int k = 0; k < DCMAX; k++) vid[k] = cn_var[b + k];
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) w[k] = llr8[((size_t)g * N + vid[k]) * 64 + lane];
        } else if constexpr (REMAP) {
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) w[k] = remap_word((size_t)slot[k]);
        } else {
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) w[k] = qi_ldm(vin + (size_t)slot[k] * 64);
        }
        qi_s2 xl[DCMAX], xh[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) { qi_unpack(w[k], xl[k], xh[k]); lo.in(xl[k]); hi.in(xh[k]); }
        lo.finish(rule); hi.finish(rule);
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) qi_stm(cout + (size_t)(b + k) * 64, qi_pack(lo.out(xl[k]), hi.out(xh[k])));      /* chk_to_var is CN-major */
    } else {
        for (int k = 0; k < deg; k++) {
            qi_s2 xl, xh;
            qi_unpack(FIRST ? llr8[((size_t)g * N + cn_var[b + k]) * 64 + lane] : (REMAP ? remap_word((size_t)cn_tr[b + k]) : vin[(size_t)cn_tr[b + k] * 64]), xl, xh);
            lo.in(xl); hi.in(xh);
        }
        lo.finish(rule); hi.finish(rule);
        for (int k = 0; k < deg; k++) {
            const size_t off = (size_t)cn_tr[b + k] * 64;
            qi_s2 xl, xh;
            qi_unpack(FIRST ? llr8[((size_t)g * N + cn_var[b + k]) * 64 + lane] : (REMAP ? remap_word((size_t)cn_tr[b + k]) : vin[off]), xl, xh);
            cout[(size_t)(b + k) * 64] = qi_pack(lo.out(xl), hi.out(xh));
        }
    }
}

/* ------------------------------------------------------------------ variable nodes ----------- */

__device__ __forceinline__ int qi_quant1(float llr, float scale)
{
    const float t = llr * scale;
    return !(t < 127.0f) ? 127 : (t < -127.0f ? -127 : __float2int_rn(t));
}

/* CODED: the quantised channel LLRs are rebuilt from the received bits (qk_coded_llr, qldpc_kernels.h) instead of read from llr8 */
template <int DVMAX, int UN, int MODE, bool CODED = false>
__global__ __launch_bounds__(QK_THREADS) void qi_vn_flood(const uint32_t *__restrict__ c2v, const uint32_t *__restrict__ llr8,
                                                          uint32_t *__restrict__ v2c, u64 *__restrict__ sgn, u64 *__restrict__ hard,
                                                          float *__restrict__ post_out,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ vn_ptr, int N, size_t group_stride /* dwords */,
                                                          const u64 *__restrict__ done, qk_coded_llr coded = qk_coded_llr{}, float scale = 0.0f, int want_ballots = 1,
                                                          const int *__restrict__ vn_tr = nullptr)
{
    const int g = blockIdx.y;
    if ((MODE != QK_VN_POST || (want_ballots & 2)) && qk_group_done<QI_V>(done, g)) return;      /* bit 1: see qk_vn_flood */
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t *cin = c2v + (size_t)g * group_stride + lane;
    uint32_t *vout = v2c + (size_t)g * group_stride + lane;
    const uint32_t *yin = llr8 + (size_t)g * N * 64 + lane;
    const int i0 = (blockIdx.x * QK_WAVES + wave) * UN;
    if (i0 >= n_list) return;

    int vv[UN], bb[UN], dd[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const int i = (i0 + u < n_list) ? i0 + u : i0;      /* the tail repeats entry i0 (idempotent) */
        vv[u] = list[i];
    }
#pragma unroll
    for (int u = 0; u < UN; u++) { bb[u] = vn_ptr[vv[u]]; dd[u] = vn_ptr[vv[u] + 1] - bb[u]; }
    uint32_t y[UN];
    if constexpr (CODED) {
        int qm[QI_V], nc[QI_V];
        const int qpin = qi_quant1(23.025850929840455f, scale);
#pragma unroll
        for (int j = 0; j < QI_V; j++) { qm[j] = qi_quant1(coded.fmag[(size_t)g * QI_FG + lane * QI_V + j], scale); nc[j] = coded.fnch[(size_t)g * QI_FG + lane * QI_V + j]; }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int cls = (int)((reinterpret_cast<const uint32_t *>(coded.vcls)[vv[u] >> 2] >> ((vv[u] & 3) * 8)) & 0xffu);
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < QI_V; j++) {
                const bool bit = (coded.ybits[((size_t)g * N + vv[u]) * QI_V + j] >> lane) & 1ull;
                int m = (cls == 0) ? (vv[u] < nc[j] ? qm[j] : qpin) : (cls == 1 ? qpin : 0);
                if (coded.ebits && ((coded.ebits[((size_t)g * N + vv[u]) * QI_V + j] >> lane) & 1ull)) m = 0;
                w |= (uint32_t)((bit ? -m : m) & 0xff) << (8 * j);
            }
            y[u] = w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < UN; u++) y[u] = qi_ldm(yin + (size_t)vv[u] * 64);
    }
    qi_s2 tl[UN], th[UN];

    if constexpr (MODE == QK_VN_FIRST) {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            qi_unpack(y[u], tl[u], th[u]);
            for (int k = 0; k < dd[u]; k++) qi_stm(vout + (size_t)(bb[u] + k) * 64, y[u]);     /* |Yq| <= 127 already */
        }
    } else if constexpr (DVMAX > 0) {
        int tr[UN][DVMAX];      /* chk_to_var is CN-major: slot s is row vn_tr[s] */
#pragma unroll
        for (int u = 0; u < UN; u++) {
#pragma unroll
            for (int k = 0; k < DVMAX; k++) tr[u][k] = vn_tr[bb[u] + k];
        }
        uint32_t m[UN][DVMAX];
#pragma unroll
        for (int u = 0; u < UN; u++) {
#pragma unroll
            for (int k = 0; k < DVMAX; k++)
                if (k < dd[u]) m[u][k] = qi_ldm(cin + (size_t)tr[u][k] * 64);
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            qi_s2 ml[DVMAX], mh[DVMAX];
            qi_unpack(y[u], tl[u], th[u]);
#pragma unroll
            for (int k = 0; k < DVMAX; k++)
                if (k < dd[u]) { qi_unpack(m[u][k], ml[k], mh[k]); tl[u] += ml[k]; th[u] += mh[k]; }
            if constexpr (MODE == QK_VN_NORMAL) {
#pragma unroll
                for (int k = 0; k < DVMAX; k++)
                    if (k < dd[u]) qi_stm(vout + (size_t)(bb[u] + k) * 64, qi_pack(qi_clamp127(tl[u] - ml[k]), qi_clamp127(th[u] - mh[k])));
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            qi_unpack(y[u], tl[u], th[u]);
            for (int k = 0; k < dd[u]; k++) {
                qi_s2 ml, mh;
                qi_unpack(cin[(size_t)vn_tr[bb[u] + k] * 64], ml, mh);
                tl[u] += ml; th[u] += mh;
            }
            if constexpr (MODE == QK_VN_NORMAL) {
                for (int k = 0; k < dd[u]; k++) {
                    qi_s2 ml, mh;
                    qi_unpack(cin[(size_t)vn_tr[bb[u] + k] * 64], ml, mh);
                    qi_stm(vout + (size_t)(bb[u] + k) * 64, qi_pack(qi_clamp127(tl[u] - ml), qi_clamp127(th[u] - mh)));
                }
            }
        }
    }
    /* all UN x 4 ballot words of the wavefront leave in one store instruction (lane 4u + j carries ballot j of VN u);
     * hard == NULL: the decoder aliases it to sgn, an integer posterior has one sign */
    uint32_t mlo = 0, mhi = 0;      /* v_writelane puts each (wave-uniform) ballot word into its lane: see qk_vn_flood */
#pragma unroll
    for (int u = 0; u < UN; u++) {
        const short t[QI_V] = {tl[u].x, tl[u].y, th[u].x, th[u].y};
        const size_t b0 = ((size_t)g * N + vv[u]) * QI_V;
        if (want_ballots) {      /* wave-uniform; see qk_vn_flood */
#pragma unroll
            for (int j = 0; j < QI_V; j++) {
                u64 s = __ballot(t[j] < 0);
                const u64 dm = (MODE == QK_VN_FIRST) ? 0ull : done[(size_t)g * QI_V + j];
                if (dm) s = (s & ~dm) | (sgn[b0 + j] & dm);       /* converged frames keep the ballots they converged with */
                mlo = qk_wlane(mlo, (uint32_t)s, u * QI_V + j);
                mhi = qk_wlane(mhi, (uint32_t)(s >> 32), u * QI_V + j);
            }
        }
        if constexpr (MODE == QK_VN_POST) {
            if (post_out) {
                qk_f32x4 p;
                p.x = (float)t[0]; p.y = (float)t[1]; p.z = (float)t[2]; p.w = (float)t[3];
                *reinterpret_cast<qk_f32x4 *>(post_out + ((size_t)g * N + vv[u]) * QI_FG + lane * QI_V) = p;
            }
        }
    }
    if (want_ballots && lane < UN * QI_V) {
        const int ul = lane / QI_V, il = (i0 + ul < n_list) ? i0 + ul : i0;
        const size_t bl = ((size_t)g * N + list[il]) * QI_V + (lane % QI_V);
        const u64 mine = ((u64)mhi << 32) | mlo;
        sgn[bl] = mine;
        if (hard) hard[bl] = mine;
    }
}

/* ------------------------------------------------------------------ horizontal layered ------- */

/*
 * One layer (VN-disjoint checks) of the fixed-point layered sweep -- the recursion of the reference's MATLAB decoder
 * (BPSK_nrldpc_sim_RM_FP.m:50-93: L = L - R; saturate; min1/min2/parity; offset; R = new; L = saturate(L + R)) with
 * one check row per "layer row", on the 8-bit containers:
 *   contr = post[v] - msg[k]  (16-bit);  x = clamp(contr, +-31) enters the fold;  msg[k] = out (|out| <= 31);
 *   post[v] = clamp(contr + out, +-127)
 * Messages get two bits less than posteriors, as there (maxqr = 31, maxqL = 127): with equal ranges a saturated
 * posterior minus a saturated message is 0 and a converged word falls apart again (measured: FER 1 after 50 sweeps).
 * post8 is [G][N][256] int8, msg8 is CN-major [G][E][256] int8.  One wavefront per check.
 */
#define QI_LAYER_MSG_MAX 31
__device__ __forceinline__ qi_s2 qi_clamp_msg(qi_s2 x) { return __builtin_elementwise_max(__builtin_elementwise_min(x, qi_splat(QI_LAYER_MSG_MAX)), qi_splat(-QI_LAYER_MSG_MAX)); }

template <int DCMAX>
__global__ __launch_bounds__(QK_THREADS) void qi_cn_layer(uint32_t *__restrict__ post8, uint32_t *__restrict__ msg8,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ cn_ptr, const int *__restrict__ cn_var,
                                                          int N, size_t group_stride /* dwords */, const u64 *__restrict__ done, qi_rule rule,
                                                          const u64 *__restrict__ synd, int M)
{
    const int g = blockIdx.y;
    if (qk_group_done<QI_V>(done, g)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * QK_WAVES + wave;
    if (i >= n_list) return;
    uint32_t *pg = post8 + (size_t)g * N * 64 + lane;
    uint32_t *mg = msg8 + (size_t)g * group_stride + lane;
    const int c = list[i];
    const int b = cn_ptr[c];
    const int deg = cn_ptr[c + 1] - b;
    qi_acc lo, hi;
    {
        unsigned s[QI_V] = {0, 0, 0, 0};
        if (synd) {
#pragma unroll
            for (int j = 0; j < QI_V; j++) s[j] = (unsigned)((synd[((size_t)g * M + c) * QI_V + j] >> lane) & 1ull);
        }
        lo.begin(s[0], s[1], QI_LAYER_MSG_MAX); hi.begin(s[2], s[3], QI_LAYER_MSG_MAX);
    }
    if constexpr (DCMAX > 0) {
        int vn[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++) vn[k] = cn_var[b + k];
        uint32_t pw[DCMAX], mw[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
                pw[k] = pg[(size_t)vn[k] * 64];                   /* posteriors are re-read by later layers: cached */
                mw[k] = qi_ldm(mg + (size_t)(b + k) * 64);        /* the check's own messages: read once, written once per sweep */
            }
        qi_s2 cl[DCMAX], ch[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
                qi_s2 pl, ph, ml, mh;
                qi_unpack(pw[k], pl, ph); qi_unpack(mw[k], ml, mh);
                cl[k] = pl - ml; ch[k] = ph - mh;
                lo.in(qi_clamp_msg(cl[k])); hi.in(qi_clamp_msg(ch[k]));
            }
        lo.finish(rule); hi.finish(rule);
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
                const qi_s2 ol = lo.out(qi_clamp_msg(cl[k])), oh = hi.out(qi_clamp_msg(ch[k]));
                qi_stm(mg + (size_t)(b + k) * 64, qi_pack(ol, oh));
                pg[(size_t)vn[k] * 64] = qi_pack(qi_clamp127(cl[k] + ol), qi_clamp127(ch[k] + oh));
            }
    } else {
        for (int k = 0; k < deg; k++) {
            qi_s2 pl, ph, ml, mh;
            qi_unpack(pg[(size_t)cn_var[b + k] * 64], pl, ph); qi_unpack(mg[(size_t)(b + k) * 64], ml, mh);
            lo.in(qi_clamp_msg(pl - ml)); hi.in(qi_clamp_msg(ph - mh));
        }
        lo.finish(rule); hi.finish(rule);
        for (int k = 0; k < deg; k++) {
            qi_s2 pl, ph, ml, mh;
            const size_t po = (size_t)cn_var[b + k] * 64, mo = (size_t)(b + k) * 64;
            qi_unpack(pg[po], pl, ph); qi_unpack(mg[mo], ml, mh);
            const qi_s2 cl = pl - ml, ch = ph - mh;
            const qi_s2 ol = lo.out(qi_clamp_msg(cl)), oh = hi.out(qi_clamp_msg(ch));
            mg[mo] = qi_pack(ol, oh);
            pg[po] = qi_pack(qi_clamp127(cl + ol), qi_clamp127(ch + oh));
        }
    }
}

/* ballots of the 8-bit posterior array: sgn = hard = (post < 0); converged frames keep theirs */
static __global__ __launch_bounds__(QK_THREADS) void qi_post_ballots(const uint32_t *__restrict__ post8, u64 *__restrict__ sgn, u64 *__restrict__ hard,
                                                              int N, const u64 *__restrict__ done)
{
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int v = blockIdx.x * QK_WAVES + wave; v < N; v += gridDim.x * QK_WAVES) {
        const uint32_t w = post8[((size_t)g * N + v) * 64 + lane];
#pragma unroll
        for (int j = 0; j < QI_V; j++) {
            u64 s = __ballot((w >> (8 * j + 7)) & 1u);
            const size_t bi = ((size_t)g * N + v) * QI_V + j;
            const u64 dm = done[(size_t)g * QI_V + j];
            if (dm) s = (s & ~dm) | (sgn[bi] & dm);
            if (lane == 0) { sgn[bi] = s; if (hard) hard[bi] = s; }
        }
    }
}

/* [G][N][256] int8 -> [G][N][256] f32 (posterior read-back) */
static __global__ __launch_bounds__(256) void qi_post_to_f32(const uint32_t *__restrict__ post8, float *__restrict__ dst, size_t n_dwords)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = post8[i];
        qk_f32x4 p;
        p.x = (float)(signed char)(w & 0xff); p.y = (float)(signed char)((w >> 8) & 0xff);
        p.z = (float)(signed char)((w >> 16) & 0xff); p.w = (float)(signed char)(w >> 24);
        *reinterpret_cast<qk_f32x4 *>(dst + i * 4) = p;
    }
}

/*
 * QKD frame formation straight into the quantised form (qk_load_bits followed by qi_quant_llr, without the fp32 array in
 * between): packed sifted-key words + per-frame |LLR| -> llr8[G][N][256], Yq = +-quant(|LLR|) at channel VNs, +-quant(23.03) at
 * pinned VNs (and at channel VNs past the frame's shortening length), 0 at punctured VNs; padding frames get quant(1).
 */
static __global__ __launch_bounds__(QK_THREADS) void qi_load_bits(const uint32_t *__restrict__ bits, const float *__restrict__ llr_mag,
                                                           const uint8_t *__restrict__ vn_class, uint32_t *__restrict__ llr8,
                                                           int N, int W, int n_frames, const int *__restrict__ n_channel, float scale)
{
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int qm[QI_V], nch[QI_V];
    bool live[QI_V];
    const int qpin = qi_quant1(23.025850929840455f, scale), qpad = qi_quant1(1.0f, scale);
#pragma unroll
    for (int j = 0; j < QI_V; j++) {
        const int f = g * QI_FG + lane * QI_V + j;
        live[j] = f < n_frames;
        qm[j] = live[j] ? qi_quant1(llr_mag[f], scale) : 0;
        nch[j] = (n_channel && live[j]) ? n_channel[f] : N;
    }
    for (int w = blockIdx.x * QK_WAVES + wave; w < W; w += gridDim.x * QK_WAVES) {
        uint32_t word[QI_V];
#pragma unroll
        for (int j = 0; j < QI_V; j++) word[j] = live[j] ? bits[(size_t)(g * QI_FG + lane * QI_V + j) * W + w] : 0u;
        for (int b = 0; b < 32; b++) {
            const int v = w * 32 + b;
            if (v >= N) break;
            const int cls = vn_class ? vn_class[v] : 0;
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < QI_V; j++) {
                const bool y = (word[j] >> (31 - b)) & 1u;
                const int m = (cls == 0) ? (v < nch[j] ? qm[j] : qpin) : (cls == 1 ? qpin : 0);
                const int q = live[j] ? (y ? -m : m) : qpad;
                o |= (uint32_t)(q & 0xff) << (8 * j);
            }
            llr8[((size_t)g * N + v) * 64 + lane] = o;
        }
    }
}

/* channel LLRs [G][N][256] f32 -> [G][N][256] int8, four frames of a lane per dword */
static __global__ __launch_bounds__(256) void qi_quant_llr(const float *__restrict__ llr, uint32_t *__restrict__ llr8, size_t n_dwords, float scale)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_dwords; i += (size_t)gridDim.x * blockDim.x) {
        const qk_f32x4 v = *reinterpret_cast<const qk_f32x4 *>(llr + i * 4);
        const float f[4] = {v.x, v.y, v.z, v.w};
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float t = f[j] * scale;
            const int q = !(t < 127.0f) ? 127 : (t < -127.0f ? -127 : __float2int_rn(t));
            w |= (uint32_t)(q & 0xff) << (8 * j);
        }
        llr8[i] = w;
    }
}

#endif /* QLDPC_KERNELS_I8_H */
