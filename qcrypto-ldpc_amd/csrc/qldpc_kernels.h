/*
 * qldpc_kernels.h -- gfx950 (CDNA4) device kernels of the batched BP decoder.
 *
 * Layout ("frame-interleaved"): frames are packed V per wavefront lane, FG = 64*V frames per group.
 *   llr  [G][N][FG] f32      channel LLRs
 *   v2c  [G][E][FG] f32      variable->check messages, VN-major slot order
 *   c2v  [G][E][FG] f32      check->variable messages, CN-major edge order (every pass WRITES contiguous rows and gathers its reads:
 *                            measured +4.5 % on the check pass against scattering into VN-major slots; gathered 256-byte row
 *                            reads run at the rate of streamed ones, tools/tform_bench.hip form D)
 *   sgn / hard [G][N][V] u64 per-VN ballots over the 64 lanes (bit = lane), one per frame-in-lane
 * One wavefront works on ONE graph node for FG frames at a time, so every graph index is
 * wave-uniform (scalar loads) and every message access is a contiguous FG*4-byte row.  The
 * arithmetic per frame is the scalar sequence of AFF3CT's Decoder_LDPC_BP_flooding /
 * _horizontal_layered (call sites: BS/src/main.cpp:193,365; VAR/main.cpp (alist-v1.0.1):203-237),
 * in the same operation order, so fp32 results are bit-identical to a scalar CPU decoder for the
 * min-sum family.  No MFMA: this is gather / min / xor / add, HBM-bound.
 */
#ifndef QLDPC_KERNELS_H
#define QLDPC_KERNELS_H

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

typedef unsigned long long u64;

#ifndef QK_WAVES
#define QK_WAVES 4            /* wavefronts per workgroup (2 and 8 measured: see DESIGN.md) */
#endif
#define QK_THREADS (QK_WAVES * 64)
#define QK_IDX_PAD 64         /* ints of padding after cn_tr / cn_var so unconditional index reads stay in bounds */

/* rule families (template parameter); the member of the family is a wave-uniform runtime value */
#define QK_FAM_MS 0           /* MS / OMS / NMS                 */
#define QK_REC_HDR 4      /* header words of a layer record {check, first edge, degree, 0, vn[...]} (bucket::d_rec, qldpc_engine_int.h) */
#define QK_FAM_SPA 1
#define QK_FAM_LSPA 2
#define QK_FAM_AMS 3          /* AMS<min | min_star_linear2 | min_star> */

struct qk_rule {
    int rule;      /* qldpc_rule */
    float param;
};

/* ------------------------------------------------------------------ small helpers ------------ */

typedef float qk_f32x2 __attribute__((ext_vector_type(2)));
typedef float qk_f32x4 __attribute__((ext_vector_type(4)));
template <int V> struct qk_vec;
template <> struct qk_vec<1> { typedef float t; };
template <> struct qk_vec<2> { typedef qk_f32x2 t; };
template <> struct qk_vec<4> { typedef qk_f32x4 t; };

template <int V> __device__ __forceinline__ void qk_load(float (&d)[V], const float *p)
{
    typename qk_vec<V>::t t = *reinterpret_cast<const typename qk_vec<V>::t *>(p);
    const float *s = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int j = 0; j < V; j++) d[j] = s[j];
}
template <int V> __device__ __forceinline__ void qk_store(float *p, const float (&d)[V])
{
    typename qk_vec<V>::t t;
    float *s = reinterpret_cast<float *>(&t);
#pragma unroll
    for (int j = 0; j < V; j++) s[j] = d[j];
    *reinterpret_cast<typename qk_vec<V>::t *>(p) = t;
}
/*
 * Streaming forms for the flooding message arrays: every row is read once and written once per pass and is
 * not needed again for a whole iteration (7.7 GB later), so both directions carry the non-temporal hint.
 * Measured on MI355X (4 096 frames): stores alone +4.5 %, loads alone -2 %, both +10.6 % (check-node kernel
 * 5.33 -> 5.71 TB/s, variable-node passes 5.5 -> 6.25 TB/s).
 */
template <int V> __device__ __forceinline__ void qk_ldm(float (&d)[V], const float *p)
{
    typename qk_vec<V>::t t = __builtin_nontemporal_load(reinterpret_cast<const typename qk_vec<V>::t *>(p));
    const float *s = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int j = 0; j < V; j++) d[j] = s[j];
}
template <int V> __device__ __forceinline__ void qk_stm(float *p, const float (&d)[V])
{
    typename qk_vec<V>::t t;
    float *s = reinterpret_cast<float *>(&t);
#pragma unroll
    for (int j = 0; j < V; j++) s[j] = d[j];
    __builtin_nontemporal_store(t, reinterpret_cast<typename qk_vec<V>::t *>(p));
}
/* store only the frames whose bit in keep[j] is set for this lane (frozen = converged frames) */
template <int V> __device__ __forceinline__ void qk_store_masked(float *p, const float (&d)[V], const bool (&frozen)[V], bool any_frozen)
{
    if (!any_frozen) { qk_store<V>(p, d); return; }
#pragma unroll
    for (int j = 0; j < V; j++) if (!frozen[j]) p[j] = d[j];
}

/*
 * fp16 message storage (qldpc_decoder_cfg.msg_dtype = 1): messages are ROUNDED TO NEAREST EVEN to binary16 when
 * stored and widened exactly when loaded; every sum / min / product is still fp32.  Halves the HBM bytes per
 * iteration.  Not the AFF3CT float build any more (FER-tolerance class against it), but bit-exact against the
 * oracle run with the same rounding.
 */
template <int V> __device__ __forceinline__ void qk_load(float (&d)[V], const __half *p)
{
    if constexpr (V == 1) d[0] = __half2float(p[0]);
    else if constexpr (V == 2) { const __half2 t = *reinterpret_cast<const __half2 *>(p); d[0] = __low2float(t); d[1] = __high2float(t); }
    else {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        const __half2 a = *reinterpret_cast<const __half2 *>(&t.x), b = *reinterpret_cast<const __half2 *>(&t.y);
        d[0] = __low2float(a); d[1] = __high2float(a); d[2] = __low2float(b); d[3] = __high2float(b);
    }
}
/* two floats -> one dword of two binary16 values, round to nearest even: v_cvt_pk_f16_f32 (one instruction on gfx950; the scalar
 * form is v_cvt_f16_f32 + v_cvt_f16_f32_sdwa + a merge) */
typedef float qk_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 qk_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t qk_pack_h2(float lo, float hi)
{
    const qk_f32x2 f = {lo, hi};
    const qk_f16x2 h = __builtin_convertvector(f, qk_f16x2);
    return *reinterpret_cast<const uint32_t *>(&h);
}
template <int V> __device__ __forceinline__ void qk_store(__half *p, const float (&d)[V])
{
    if constexpr (V == 1) p[0] = __float2half_rn(d[0]);
    else if constexpr (V == 2) *reinterpret_cast<uint32_t *>(p) = qk_pack_h2(d[0], d[1]);
    else {
        uint2 t;
        t.x = qk_pack_h2(d[0], d[1]); t.y = qk_pack_h2(d[2], d[3]);
        *reinterpret_cast<uint2 *>(p) = t;
    }
}
template <int V> __device__ __forceinline__ void qk_store_masked(__half *p, const float (&d)[V], const bool (&frozen)[V], bool any_frozen)
{
    if (!any_frozen) { qk_store<V>(p, d); return; }
#pragma unroll
    for (int j = 0; j < V; j++) if (!frozen[j]) p[j] = __float2half_rn(d[j]);
}
typedef unsigned qk_u32x2 __attribute__((ext_vector_type(2)));
template <int V> __device__ __forceinline__ void qk_ldm(float (&d)[V], const __half *p)
{
    if constexpr (V == 1) { const unsigned short r = __builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p)); d[0] = __half2float(__ushort_as_half(r)); }
    else if constexpr (V == 2) {
        const unsigned r = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(p));
        const __half2 t = *reinterpret_cast<const __half2 *>(&r);
        d[0] = __low2float(t); d[1] = __high2float(t);
    } else {
        const qk_u32x2 r = __builtin_nontemporal_load(reinterpret_cast<const qk_u32x2 *>(p));
        const unsigned r0 = r.x, r1 = r.y;
        const __half2 a = *reinterpret_cast<const __half2 *>(&r0), b = *reinterpret_cast<const __half2 *>(&r1);
        d[0] = __low2float(a); d[1] = __high2float(a); d[2] = __low2float(b); d[3] = __high2float(b);
    }
}
template <int V> __device__ __forceinline__ void qk_stm(__half *p, const float (&d)[V])
{
    if constexpr (V == 1) __builtin_nontemporal_store(__half_as_ushort(__float2half_rn(d[0])), reinterpret_cast<unsigned short *>(p));
    else if constexpr (V == 2) __builtin_nontemporal_store(qk_pack_h2(d[0], d[1]), reinterpret_cast<unsigned *>(p));
    else {
        qk_u32x2 t;
        t.x = qk_pack_h2(d[0], d[1]); t.y = qk_pack_h2(d[2], d[3]);
        __builtin_nontemporal_store(t, reinterpret_cast<qk_u32x2 *>(p));
    }
}
/* one element (REMAP check passes: a lane's V frames come from V different rows of the old layout) */
/* v_writelane_b32: a wave-uniform word into ONE lane of a VGPR (lane is a compile-time constant after unrolling) */
__device__ __forceinline__ uint32_t qk_wlane(uint32_t old, uint32_t val, int lane)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(lane));
    return old;
}
__device__ __forceinline__ u64 qk_uniform64(u64 v)
{
    return ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ __forceinline__ float qk_ldm1(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float qk_ldm1(const __half *p) { return __half2float(__ushort_as_half(__builtin_nontemporal_load(reinterpret_cast<const unsigned short *>(p)))); }
__device__ __forceinline__ void qk_put(float *p, float v) { *p = v; }
__device__ __forceinline__ void qk_put(__half *p, float v) { *p = __float2half_rn(v); }

__device__ __forceinline__ uint32_t qk_bits(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float qk_withsign(float mag, uint32_t signbit) { return __uint_as_float((__float_as_uint(mag) & 0x7fffffffu) | (signbit & 0x80000000u)); }
__device__ __forceinline__ float qk_min(float a, float b) { return (b < a) ? b : a; }   /* std::min */
__device__ __forceinline__ float qk_max(float a, float b) { return (a < b) ? b : a; }   /* std::max */

__device__ __forceinline__ float qk_corr_l2(float x) { float t = 0.6f - 0.24f * fabsf(x); return t > 0.0f ? t : 0.0f; }
__device__ __forceinline__ float qk_ams_min(int rule, float a, float b)
{
    const float m = qk_min(a, b);
    if (rule == 5) return m;
    if (rule == 7) return m + __logf(1.0f + __expf(-(a + b))) - __logf(1.0f + __expf(-fabsf(a - b)));   /* tolerance class: hardware exp/log */
    const float r = m + qk_corr_l2(a + b) - qk_corr_l2(a - b);
    return r > 0.0f ? r : 0.0f;
}

/* Per-check accumulator of one frame.  in() is compute_chk_node_in, finish() is end_chk_node_in,
 * out() is compute_chk_node_out of tools::Update_rule_*. */
template <int FAM> struct qk_acc;

template <> struct qk_acc<QK_FAM_MS> {
    uint32_t sign; float min1, min2, cst1, cst2;
    __device__ __forceinline__ void begin() { sign = 0; min1 = 3.402823466e+38f; min2 = 3.402823466e+38f; }
    __device__ __forceinline__ void in(float x)
    {
        const float a = fabsf(x);
        sign ^= qk_bits(x);
        min2 = qk_min(min2, qk_max(a, min1));
        min1 = qk_min(min1, a);
    }
    __device__ __forceinline__ void finish(const qk_rule &r)
    {
        if (r.rule == 0)      { cst1 = qk_max(0.0f, min2);           cst2 = qk_max(0.0f, min1); }
        else if (r.rule == 1) { cst1 = qk_max(0.0f, min2 - r.param); cst2 = qk_max(0.0f, min1 - r.param); }
        else                  { cst1 = min2 * r.param;               cst2 = min1 * r.param; }
    }
    __device__ __forceinline__ float out(float x, const qk_rule &) const
    {
        const float a = fabsf(x);
        return qk_withsign((a == min1) ? cst1 : cst2, sign ^ qk_bits(x));
    }
};

/*
 * SPA transcendentals.  The float-LLR SPA variant is tolerance-class (FER / converged-word agreement, not
 * bit-exactness: device and glibc tanh/atanh already differ in the last ulp), so the hardware exp / log /
 * rcp units are used: tanh(a/2) = (1 - e^-a) / (1 + e^-a), 2 atanh(r) = ln((1 + r) / (1 - r)).  With the libm
 * forms the check-node kernel is compute-bound at 3x the min-sum time; with these it is HBM-bound again.
 */
__device__ __forceinline__ float qk_rcp(float x) { return __builtin_amdgcn_rcpf(x); }      /* v_rcp_f32, 1 ulp; __fdividef expands to the 10-instruction IEEE sequence */
__device__ __forceinline__ float qk_tanh_half(float a) { const float e = __expf(-a); return (1.0f - e) * qk_rcp(1.0f + e); }
__device__ __forceinline__ float qk_2atanh(float r) { return __logf((1.0f + r) * qk_rcp(1.0f - r)); }

/*
 * SPA on exponentials (three transcendentals per edge instead of five).  With e_i = exp(-|x_i|), tanh(|x_i|/2) =
 * (1 - e_i) / (1 + e_i), so the product over the other edges is Nn / D with
 *   Nn = P_num * (1 + e_i),  D = P_den * (1 - e_i),   P_num = prod (1 - e_j),  P_den = prod (1 + e_j)   (all j)
 * and 2 atanh(Nn / D) = ln((D + Nn) / (D - Nn)): one exp on the way in, one rcp and one log on the way out, no division
 * by tanh_i.  AFF3CT's clamp of the product to 1 - FLT_EPSILON becomes a clamp of the ratio to (2 - eps) / eps; the
 * comparison is written so that NaN (an erased edge: 0 / 0, which AFF3CT's `v < 1 ? v : 1 - eps` also sends to the
 * clamp) takes it too.  Messages enter the fold as sign-carrying e_i so the exponential is computed once per edge.
 * P_den <= 2^dc stays finite in fp32 for check degrees below 128.
 */
#define QK_SPA_RMAX 16777215.0f      /* (2 - eps) / eps with eps = 2^-23 */
template <int FAM> __device__ __forceinline__ float qk_prep(float x) { return x; }
/* The two transcendentals of the fold as the bare hardware instructions (v_exp_f32 / v_log_f32 work in base 2).  __expf / __logf wrap them in
 * range handling this rule does not need: exp(-|x|) may flush to 0 below 2^-126 (1 +- e is 1 either way), and the logarithm's argument lies
 * in [1, QK_SPA_RMAX], never denormal -- 7 VALU instructions fewer per edge of the ~30 the pass spent (ISA: the cmp / cndmask / ldexp / two-term
 * ln 2 sequences are gone). */
__device__ __forceinline__ float qk_spa_exp_neg(float a) { return __builtin_amdgcn_exp2f(a * -1.4426950408889634f); }      /* exp(-a), a >= 0 */
__device__ __forceinline__ float qk_spa_ln(float r) { return __builtin_amdgcn_logf(r) * 0.6931471805599453f; }              /* ln r, 1 <= r < 2^24 */
template <> __device__ __forceinline__ float qk_prep<QK_FAM_SPA>(float x) { return qk_withsign(qk_spa_exp_neg(fabsf(x)), qk_bits(x)); }

template <> struct qk_acc<QK_FAM_SPA> {
    uint32_t sign; float pnum, pden;
    __device__ __forceinline__ void begin() { sign = 0; pnum = 1.0f; pden = 1.0f; }
    __device__ __forceinline__ void in(float xp) { const float e = fabsf(xp); sign ^= qk_bits(xp); pnum *= 1.0f - e; pden *= 1.0f + e; }      /* xp = qk_prep(x) */
    __device__ __forceinline__ void finish(const qk_rule &) {}
    __device__ __forceinline__ float out(float xp, const qk_rule &) const
    {
        const float e = fabsf(xp);
        const float nn = pnum * (1.0f + e), d = pden * (1.0f - e);
        /* Nn <= D holds for the exact products; when the other edges' e_j are ~2^-23 the rounded ones can come out an ulp the wrong way round
         * (the true ratio is then beyond the clamp anyway): |D - Nn| keeps r positive, so the logarithm never sees a negative argument */
        float r = (d + nn) * qk_rcp(fabsf(d - nn));
        r = (r < QK_SPA_RMAX) ? r : QK_SPA_RMAX;
        return qk_withsign(qk_spa_ln(r), sign ^ qk_bits(xp));
    }
};

template <> struct qk_acc<QK_FAM_LSPA> {
    uint32_t sign; float sum;
    __device__ __forceinline__ static float term(float x)
    {
        const float t = qk_tanh_half(fabsf(x));
        return (t != 0.0f) ? __logf(t) : 1.175494351e-38f;
    }
    __device__ __forceinline__ void begin() { sign = 0; sum = 0.0f; }
    __device__ __forceinline__ void in(float x) { sign ^= qk_bits(x); sum += term(x); }
    __device__ __forceinline__ void finish(const qk_rule &) {}
    __device__ __forceinline__ float out(float x, const qk_rule &) const
    {
        float t = sum - term(x);
        t = (t != 0.0f) ? __expf(t) : 1.0f - 1.1920928955078125e-07f;
        t = (t < 1.0f) ? t : 1.0f - 1.1920928955078125e-07f;      /* keep ln((1+t)/(1-t)) finite when rounding gives 1 */
        return qk_withsign(qk_2atanh(t), sign ^ qk_bits(x));
    }
};

template <> struct qk_acc<QK_FAM_AMS> {
    uint32_t sign; float mn, delta_min, delta; int rule;
    __device__ __forceinline__ void begin() { sign = 0; mn = 3.402823466e+38f; delta_min = 3.402823466e+38f; }
    __device__ __forceinline__ void in_r(float x, int r)
    {
        const float a = fabsf(x);
        sign ^= qk_bits(x);
        float other;
        if (a < mn) { other = mn; mn = a; } else other = a;
        delta_min = qk_ams_min(r, delta_min, other);
    }
    __device__ __forceinline__ void finish(const qk_rule &r)
    {
        delta = qk_max(0.0f, qk_ams_min(r.rule, delta_min, mn));
        delta_min = qk_max(0.0f, delta_min);
    }
    __device__ __forceinline__ float out(float x, const qk_rule &) const
    {
        const float a = fabsf(x);
        return qk_withsign((a == mn) ? delta_min : delta, sign ^ qk_bits(x));
    }
};

template <int FAM> __device__ __forceinline__ void qk_acc_in(qk_acc<FAM> &a, float x, const qk_rule &) { a.in(x); }
template <> __device__ __forceinline__ void qk_acc_in<QK_FAM_AMS>(qk_acc<QK_FAM_AMS> &a, float x, const qk_rule &r) { a.in_r(x, r.rule); }

/* wave-uniform "all frames of this group have converged" test; done is [G][V] */
template <int V> __device__ __forceinline__ bool qk_group_done(const u64 *__restrict__ done, int g)
{
    bool all = true;
#pragma unroll
    for (int j = 0; j < V; j++) all = all && (done[(size_t)g * V + j] == ~0ull);
    return all;
}
template <int V> __device__ __forceinline__ bool qk_frozen(const u64 *__restrict__ done, int g, int lane, bool (&frozen)[V])
{
    bool any = false;
#pragma unroll
    for (int j = 0; j < V; j++) {
        const u64 d = done[(size_t)g * V + j];
        frozen[j] = (d >> lane) & 1ull;
        any = any || (d != 0ull);
    }
    return any;   /* wave-uniform */
}

/*
 * Channel LLRs of a QKD frame take three magnitudes (BS/src/main.cpp:348-362: the frame's ln((1-p)/p) at channel VNs, 23.03 at
 * pinned VNs, 0 at punctured ones) and the sign is the received bit.  After qldpc_load_bits_* the flooding passes therefore do
 * not read an LLR array at all (CODED = true): the received bits sit as one ballot word per VN (bit = lane, like sgn / hard),
 * the class is a byte per VN, the magnitude and the shortening length a value per frame; Y is rebuilt in registers, the same
 * float the array would hold.  N/8 bytes per frame and pass instead of 4 N, and the loads are scalar.
 */
struct qk_coded_llr {
    const u64 *ybits;          /* [G][N][V] received bits as ballots                         */
    const float *fmag;         /* [G*FG] |LLR| of a channel bit of each frame (padding: 1)  */
    const int *fnch;           /* [G*FG] class-0 VNs at index >= fnch[f] are pinned (shortening) */
    const uint8_t *vcls;       /* [N] VN class (allocated to a multiple of 4 bytes: read as dwords) */
    const u64 *ebits;          /* [G][N][V] per-frame erasures as ballots (qldpc_load_erasures_dev), or NULL */
};


/* Y of VN v for the V frames of this lane, rebuilt from the coded form (the float the LLR array would hold) */
template <int V> __device__ __forceinline__ void qk_coded_y(float (&y)[V], const qk_coded_llr &c, int g, int v, int N, int lane, const float (&mg)[V], const int (&nc)[V])
{
    /* v is wave-uniform: the class byte comes out of a scalar dword load (a byte load is a vector load with a round trip and per-lane compares) */
    const int cls = (int)((reinterpret_cast<const uint32_t *>(c.vcls)[v >> 2] >> ((v & 3) * 8)) & 0xffu);
#pragma unroll
    for (int j = 0; j < V; j++) {
        const bool bit = (c.ybits[((size_t)g * N + v) * V + j] >> lane) & 1ull;
        float m = (cls == 0) ? (v < nc[j] ? mg[j] : 23.025850929840455f) : (cls == 1 ? 23.025850929840455f : 0.0f);
        if (c.ebits && ((c.ebits[((size_t)g * N + v) * V + j] >> lane) & 1ull)) m = 0.0f;      /* punctured for this frame: LLR 0 (BS/src/main.cpp:359-362) */
        y[j] = bit ? -m : m;
    }
}
/* what _initialize_var_to_chk leaves in var_to_chk before the first check pass: (Y + 0) - 0, through the message type */
__device__ __forceinline__ float qk_first_v2c(float y, const float *) { return (y + 0.0f) - 0.0f; }
__device__ __forceinline__ float qk_first_v2c(float y, const __half *) { return __half2float(__float2half_rn((y + 0.0f) - 0.0f)); }

/* ------------------------------------------------------------------ flooding: check nodes ---- */

/*
 * _decode_single_ite of Decoder_LDPC_BP_flooding: for each check gather var_to_chk through
 * `transpose`, fold, emit chk_to_var to the same slots.  ONE wavefront per check, FG frames wide,
 * one check per wavefront per launch: all graph indices are fetched with back-to-back scalar loads
 * (cn_tr is padded by QK_IDX_PAD ints so the reads past `deg` stay in bounds), then every message
 * row load is in flight before the first use; the hardware overlaps checks across wavefronts.
 * DCMAX > 0: messages stay in registers (checks in `list` have degree <= DCMAX).
 * DCMAX == 0: any degree, second pass re-reads the rows (they are L2-hot).
 */
/* FIRST: the check pass of iteration 0 in coded-LLR mode.  var_to_chk would hold (Y + 0) - 0 on every edge, so it is not read
 * (and the variable-node pass that would have written it is not run): the inputs are rebuilt from the coded LLRs of cn_var. */
/* REMAP: the check pass right after a compaction (qldpc_kernels_compact.h).  var_to_chk is still laid out for the old generation:
 * frame (lane, j) of this group reads slot remap_src[g * FG + lane * V + j] of it (a per-lane base pointer; padding lanes read slot 0,
 * their results are never looked at); chk_to_var is written in the new layout. */
template <int V, int DCMAX, int FAM, typename MT, bool FIRST = false, bool REMAP = false>
__global__ __launch_bounds__(QK_THREADS) void qk_cn_flood(const MT *__restrict__ v2c, MT *__restrict__ c2v,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ cn_ptr, const int *__restrict__ cn_tr,
                                                          size_t group_stride, const u64 *__restrict__ done, qk_rule rule, int freeze, const u64 *__restrict__ synd, int M,
                                                          const int *__restrict__ cn_var = nullptr, int N = 0, qk_coded_llr coded = qk_coded_llr{},
                                                          const int *__restrict__ remap_src = nullptr)
{
    static_assert(!(FIRST && REMAP), "no compaction before the first check pass");
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    if (qk_group_done<V>(done, g)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * QK_WAVES + wave;
    if (i >= n_list) return;
    bool frozen[V];
    const bool any_frozen = qk_frozen<V>(done, g, lane, frozen) && freeze;
    const MT *vin = v2c + (size_t)g * group_stride + lane * V;
    MT *cout = c2v + (size_t)g * group_stride + lane * V;
    const MT *vin_r[V];      /* REMAP only */
    if constexpr (REMAP) {
#pragma unroll
        for (int j = 0; j < V; j++) {
            int s = remap_src[(size_t)g * FG + lane * V + j];
            s = s < 0 ? 0 : s;
            vin_r[j] = v2c + (size_t)(s / FG) * group_stride + (s % FG);
        }
    }

    const int c = list[i];
    const int b = cn_ptr[c];
    const int deg = cn_ptr[c + 1] - b;
    qk_acc<FAM> acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        acc[j].begin();
        /* syndrome form: the check must come out with parity s_c, i.e. its sign product starts at (-1)^s_c */
        if (synd) acc[j].sign = (uint32_t)((synd[((size_t)g * M + c) * V + j] >> lane) & 1ull) << 31;
    }
    if constexpr (DCMAX > 0) {
        int slot[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++) slot[k] = cn_tr[b + k];
        float x[DCMAX][V];
        if constexpr (FIRST) {
            float mg[V];
            int nc[V];
#pragma unroll
            for (int j = 0; j < V; j++) { mg[j] = coded.fmag[(size_t)g * FG + lane * V + j]; nc[j] = coded.fnch[(size_t)g * FG + lane * V + j]; }
            int vid[DCMAX];
#pragma unroll
            for (int k = 0; k < DCMAX; k++) vid[k] = cn_var[b + k];
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) {
                    qk_coded_y<V>(x[k], coded, g, vid[k], N, lane, mg, nc);
#pragma unroll
                    for (int j = 0; j < V; j++) x[k][j] = qk_first_v2c(x[k][j], vin);
                }
        } else if constexpr (REMAP) {
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) {
#pragma unroll
                    for (int j = 0; j < V; j++) x[k][j] = qk_ldm1(vin_r[j] + (size_t)slot[k] * FG);
                }
        } else {
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) qk_ldm<V>(x[k], vin + (size_t)slot[k] * FG);
        }
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
#pragma unroll
                for (int j = 0; j < V; j++) { x[k][j] = qk_prep<FAM>(x[k][j]); qk_acc_in<FAM>(acc[j], x[k][j], rule); }
            }
#pragma unroll
        for (int j = 0; j < V; j++) acc[j].finish(rule);
        if (!any_frozen) {
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) {
                    float o[V];
#pragma unroll
                    for (int j = 0; j < V; j++) o[j] = acc[j].out(x[k][j], rule);
                    qk_stm<V>(cout + (size_t)(b + k) * FG, o);      /* chk_to_var is CN-major: a check's rows are contiguous */
                }
        } else {
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) {
#pragma unroll
                    for (int j = 0; j < V; j++)
                        if (!frozen[j]) qk_put(&cout[(size_t)(b + k) * FG + j], acc[j].out(x[k][j], rule));
                }
        }
    } else {
        float mg[V];
        int nc[V];
        if constexpr (FIRST) {
#pragma unroll
            for (int j = 0; j < V; j++) { mg[j] = coded.fmag[(size_t)g * FG + lane * V + j]; nc[j] = coded.fnch[(size_t)g * FG + lane * V + j]; }
        }
        for (int k = 0; k < deg; k++) {
            float x[V];
            if constexpr (FIRST) {
                qk_coded_y<V>(x, coded, g, cn_var[b + k], N, lane, mg, nc);
#pragma unroll
                for (int j = 0; j < V; j++) x[j] = qk_first_v2c(x[j], vin);
            } else if constexpr (REMAP) {
#pragma unroll
                for (int j = 0; j < V; j++) x[j] = qk_ldm1(vin_r[j] + (size_t)cn_tr[b + k] * FG);
            } else qk_load<V>(x, vin + (size_t)cn_tr[b + k] * FG);
#pragma unroll
            for (int j = 0; j < V; j++) qk_acc_in<FAM>(acc[j], qk_prep<FAM>(x[j]), rule);
        }
#pragma unroll
        for (int j = 0; j < V; j++) acc[j].finish(rule);
        for (int k = 0; k < deg; k++) {
            float x[V], o[V];
            const size_t off = (size_t)cn_tr[b + k] * FG;
            if constexpr (FIRST) {
                qk_coded_y<V>(x, coded, g, cn_var[b + k], N, lane, mg, nc);
#pragma unroll
                for (int j = 0; j < V; j++) x[j] = qk_first_v2c(x[j], vin);
            } else if constexpr (REMAP) {
#pragma unroll
                for (int j = 0; j < V; j++) x[j] = qk_ldm1(vin_r[j] + off);
            } else qk_load<V>(x, vin + off);
#pragma unroll
            for (int j = 0; j < V; j++) o[j] = acc[j].out(qk_prep<FAM>(x[j]), rule);
            qk_store_masked<V>(cout + (size_t)(b + k) * FG, o, frozen, any_frozen);
        }
    }
}

/* ------------------------------------------------------------------ flooding: variable nodes - */

#define QK_VN_FIRST 0     /* iteration 0: chk_to_var is all zero (decoder after reset()), not read */
#define QK_VN_NORMAL 1    /* _initialize_var_to_chk                                               */
#define QK_VN_POST 2      /* _compute_post only: ballots, optional post store, no var_to_chk       */

/*
 * _initialize_var_to_chk / _compute_post: sum = 0; sum += chk_to_var[slot] in slot order;
 * tmp = Y[v] + sum; var_to_chk[slot] = tmp - chk_to_var[slot].  A VN's slots are contiguous rows,
 * so this kernel streams.  Also emits the per-VN ballots: sgn = signbit(tmp) (what
 * check_syndrome_soft tests) and hard = !(tmp >= 0) (what decode_siho outputs).
 * One wavefront handles UN list entries with every row load issued before the first use.
 */
template <int V, int DVMAX, int UN, int MODE, typename MT, bool CODED = false>
__global__ __launch_bounds__(QK_THREADS) void qk_vn_flood(const MT *__restrict__ c2v, const float *__restrict__ llr,
                                                          MT *__restrict__ v2c, u64 *__restrict__ sgn, u64 *__restrict__ hard,
                                                          float *__restrict__ post_out,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ vn_ptr, int N, size_t group_stride,
                                                          const u64 *__restrict__ done, qk_coded_llr coded = qk_coded_llr{}, int want_ballots = 1,
                                                          const int *__restrict__ vn_tr = nullptr)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    /* want_ballots bit 1: this _compute_post closes an early-exit run -- a group whose frames have all converged already holds its
     * final (frozen) ballots and is skipped; without it (posterior read-back) every group is recomputed */
    if ((MODE != QK_VN_POST || (want_ballots & 2)) && qk_group_done<V>(done, g)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const MT *cin = c2v + (size_t)g * group_stride + lane * V;
    MT *vout = v2c + (size_t)g * group_stride + lane * V;
    const float *yin = llr + (size_t)g * N * FG + lane * V;
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
    float y[UN][V], tmp[UN][V];
    /* the channel values: rebuilt from ballots (CODED) or read as rows.  In the unrolled path this runs AFTER the message rows have been
     * asked for, so that its own small loads (frame constants, ballot words, class) travel with them instead of in front of them */
    auto channel = [&]() {
        if constexpr (CODED) {
            float mg[V];
            int nc[V];
#pragma unroll
            for (int j = 0; j < V; j++) { mg[j] = coded.fmag[(size_t)g * FG + lane * V + j]; nc[j] = coded.fnch[(size_t)g * FG + lane * V + j]; }
#pragma unroll
            for (int u = 0; u < UN; u++) qk_coded_y<V>(y[u], coded, g, vv[u], N, lane, mg, nc);
        } else {
#pragma unroll
            for (int u = 0; u < UN; u++) qk_ldm<V>(y[u], yin + (size_t)vv[u] * FG);
        }
    };
    if constexpr (MODE == QK_VN_FIRST || DVMAX <= 0) channel();

    if constexpr (MODE == QK_VN_FIRST) {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            float o[V];
#pragma unroll
            for (int j = 0; j < V; j++) { tmp[u][j] = y[u][j] + 0.0f; o[j] = tmp[u][j] - 0.0f; }
            for (int k = 0; k < dd[u]; k++) qk_stm<V>(vout + (size_t)(bb[u] + k) * FG, o);
        }
    } else if constexpr (DVMAX > 0) {
        /* chk_to_var is CN-major (the check pass streams it out): slot s of this VN-major order is row vn_tr[s] (padded array) */
        int tr[UN][DVMAX];
#pragma unroll
        for (int u = 0; u < UN; u++) {
#pragma unroll
            for (int k = 0; k < DVMAX; k++) tr[u][k] = vn_tr[bb[u] + k];
        }
        float m[UN][DVMAX][V];
#pragma unroll
        for (int u = 0; u < UN; u++) {
#pragma unroll
            for (int k = 0; k < DVMAX; k++)
                if (k < dd[u]) qk_ldm<V>(m[u][k], cin + (size_t)tr[u][k] * FG);
        }
        channel();
#pragma unroll
        for (int u = 0; u < UN; u++) {
            float sum[V];
#pragma unroll
            for (int j = 0; j < V; j++) sum[j] = 0.0f;
#pragma unroll
            for (int k = 0; k < DVMAX; k++)
                if (k < dd[u]) {
#pragma unroll
                    for (int j = 0; j < V; j++) sum[j] += m[u][k][j];
                }
#pragma unroll
            for (int j = 0; j < V; j++) tmp[u][j] = y[u][j] + sum[j];
            if constexpr (MODE == QK_VN_NORMAL) {
#pragma unroll
                for (int k = 0; k < DVMAX; k++)
                    if (k < dd[u]) {
                        float o[V];
#pragma unroll
                        for (int j = 0; j < V; j++) o[j] = tmp[u][j] - m[u][k][j];
                        qk_stm<V>(vout + (size_t)(bb[u] + k) * FG, o);
                    }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < UN; u++) {
            float sum[V];
#pragma unroll
            for (int j = 0; j < V; j++) sum[j] = 0.0f;
            for (int k = 0; k < dd[u]; k++) {
                float m[V];
                qk_load<V>(m, cin + (size_t)vn_tr[bb[u] + k] * FG);
#pragma unroll
                for (int j = 0; j < V; j++) sum[j] += m[j];
            }
#pragma unroll
            for (int j = 0; j < V; j++) tmp[u][j] = y[u][j] + sum[j];
            if constexpr (MODE == QK_VN_NORMAL) {
                for (int k = 0; k < dd[u]; k++) {
                    float m[V], o[V];
                    qk_load<V>(m, cin + (size_t)vn_tr[bb[u] + k] * FG);
#pragma unroll
                    for (int j = 0; j < V; j++) o[j] = tmp[u][j] - m[j];
                    qk_stm<V>(vout + (size_t)(bb[u] + k) * FG, o);
                }
            }
        }
    }
    /* all 2 * UN * V ballot words of this wavefront leave in ONE store instruction: lane 2k carries the k-th sgn word,
     * lane 2k + 1 the k-th hard word, each with its own address (lane-0-only stores cost 2 * UN * V instructions and
     * showed up in the profile: +16 % on the 8-bit VN pass, where V = 4).  The words are wave-uniform (SGPR pairs): v_writelane
     * drops each half into its lane, and the lane works out its own address from the list entry it reloads (selecting words and
     * addresses with per-lane compares cost 8 v_cndmask per VN and frame set: a third of the VALU work of the binary16 pass) */
    if (want_ballots) {      /* wave-uniform: without the syndrome test nobody reads the ballots of the in-between passes, only those of _compute_post */
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int u = 0; u < UN; u++) {
#pragma unroll
        for (int j = 0; j < V; j++) {
            /* converged frames keep the ballots they converged with (their messages may go on evolving when
             * the decoder is not in freeze mode: full-row stores are much cheaper than lane-masked ones) */
            u64 s = __ballot((qk_bits(tmp[u][j]) >> 31) != 0);
            u64 h = __ballot(!(tmp[u][j] >= 0.0f));
            const size_t bi = ((size_t)g * N + vv[u]) * V + j;
            const u64 dm = (MODE == QK_VN_FIRST) ? 0ull : done[(size_t)g * V + j];
            if (dm) { s = (s & ~dm) | (sgn[bi] & dm); h = (h & ~dm) | (hard[bi] & dm); }
            const int slot = (u * V + j) * 2;
            mlo = qk_wlane(mlo, (uint32_t)s, slot);
            mhi = qk_wlane(mhi, (uint32_t)(s >> 32), slot);
            mlo = qk_wlane(mlo, (uint32_t)h, slot + 1);
            mhi = qk_wlane(mhi, (uint32_t)(h >> 32), slot + 1);
        }
    }
    if (lane < 2 * UN * V) {      /* a repeated tail entry writes the same value to the same address */
        const int t = lane >> 1, ul = t / V, jl = t % V;
        const int il = (i0 + ul < n_list) ? i0 + ul : i0;
        const size_t bl = ((size_t)g * N + list[il]) * V + jl;
        ((lane & 1) ? hard : sgn)[bl] = ((u64)mhi << 32) | mlo;
    }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
        if constexpr (MODE == QK_VN_POST) {
            if (post_out) qk_store<V>(post_out + ((size_t)g * N + vv[u]) * FG + lane * V, tmp[u]);
        }
    }
}

/* ------------------------------------------------------------------ horizontal layered ------- */

/*
 * One layer (VN-disjoint checks) of Decoder_LDPC_BP_horizontal_layered::_decode_single_ite:
 *   contributions[i] = var_nodes[v_i] - messages[k]; fold; messages[k] = out_i;
 *   var_nodes[v_i] = contributions[i] + messages[k]
 * msg is CN-major [G][E][FG]; post is [G][N][FG].  One wavefront per check.
 */
template <int V, int DCMAX, int FAM>
__global__ __launch_bounds__(QK_THREADS) void qk_cn_layer(float *__restrict__ post, float *__restrict__ msg,
                                                          const int *__restrict__ list, int n_list,
                                                          const int *__restrict__ cn_ptr, const int *__restrict__ cn_var,
                                                          int N, size_t group_stride, const u64 *__restrict__ done, qk_rule rule, int freeze, const u64 *__restrict__ synd, int M,
                                                          int first /* sweep 0: messages are all zero -- they are not read (and the host has not cleared the array) */,
                                                          const int *__restrict__ rec = nullptr, int rec_stride = 0 /* bucket::d_rec (DCMAX > 0 only) */)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * QK_WAVES + wave;
    if (i >= n_list) return;
    /* with records the wave's check, first edge, degree and VNs arrive in one scalar round trip, asked for together with the done words */
    int c, b, deg;
    [[maybe_unused]] int rvn[DCMAX > 0 ? DCMAX : 1];
    bool have_rec = false;
    if constexpr (DCMAX > 0) {
        if (rec) {
            const int *r = static_cast<const int *>(__builtin_assume_aligned(rec + (size_t)i * rec_stride, 16));
            c = r[0]; b = r[1]; deg = r[2];
#pragma unroll
            for (int k = 0; k < DCMAX; k++) rvn[k] = r[QK_REC_HDR + k];
            have_rec = true;
            /* keeps the record loads above the early return below (one round trip for the record and the done words together) */
            u64 dn[V];
#pragma unroll
            for (int j = 0; j < V; j++) dn[j] = done[(size_t)g * V + j];
            asm volatile("" ::"s"(c), "s"(b), "s"(deg), "s"(rvn[0]), "s"(rvn[DCMAX / 2]), "s"(rvn[DCMAX - 1]), "s"(dn[0]), "s"(dn[V - 1]));
            bool all = true;
#pragma unroll
            for (int j = 0; j < V; j++) all = all && dn[j] == ~0ull;
            if (all) return;
        }
    }
    if (!have_rec) {
        if (qk_group_done<V>(done, g)) return;
        c = list[i];
        b = cn_ptr[c];
        deg = cn_ptr[c + 1] - b;
    }
    bool frozen[V];
    const bool any_frozen = qk_frozen<V>(done, g, lane, frozen) && freeze;
    float *pg = post + (size_t)g * N * FG + lane * V;
    float *mg = msg + (size_t)g * group_stride + lane * V;

    qk_acc<FAM> acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        acc[j].begin();
        /* syndrome form: the check must come out with parity s_c, i.e. its sign product starts at (-1)^s_c */
        if (synd) acc[j].sign = (uint32_t)((synd[((size_t)g * M + c) * V + j] >> lane) & 1ull) << 31;
    }
    if constexpr (DCMAX > 0) {
        int vn[DCMAX];
#pragma unroll
        for (int k = 0; k < DCMAX; k++) vn[k] = have_rec ? rvn[k] : cn_var[b + k];
        float x[DCMAX][V], m[DCMAX][V];
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
                qk_load<V>(x[k], pg + (size_t)vn[k] * FG);
                if (!first) qk_ldm<V>(m[k], mg + (size_t)(b + k) * FG);        /* the check's own messages: read once, written once per sweep */
                else {
#pragma unroll
                    for (int j = 0; j < V; j++) m[k][j] = 0.0f;
                }
            }
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
#pragma unroll
                for (int j = 0; j < V; j++) { x[k][j] = x[k][j] - m[k][j]; qk_acc_in<FAM>(acc[j], qk_prep<FAM>(x[k][j]), rule); }
            }
#pragma unroll
        for (int j = 0; j < V; j++) acc[j].finish(rule);
#pragma unroll
        for (int k = 0; k < DCMAX; k++)
            if (k < deg) {
                float o[V], p[V];
#pragma unroll
                for (int j = 0; j < V; j++) { o[j] = acc[j].out(qk_prep<FAM>(x[k][j]), rule); p[j] = x[k][j] + o[j]; }
                if (!any_frozen) qk_stm<V>(mg + (size_t)(b + k) * FG, o);
                else qk_store_masked<V>(mg + (size_t)(b + k) * FG, o, frozen, any_frozen);
                qk_store_masked<V>(pg + (size_t)vn[k] * FG, p, frozen, any_frozen);      /* posteriors are re-read by later layers: cached */
            }
    } else {
        for (int k = 0; k < deg; k++) {
            float p[V], m[V];
            qk_load<V>(p, pg + (size_t)cn_var[b + k] * FG);
            if (!first) qk_load<V>(m, mg + (size_t)(b + k) * FG);
            else {
#pragma unroll
                for (int j = 0; j < V; j++) m[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < V; j++) qk_acc_in<FAM>(acc[j], qk_prep<FAM>(p[j] - m[j]), rule);
        }
#pragma unroll
        for (int j = 0; j < V; j++) acc[j].finish(rule);
        for (int k = 0; k < deg; k++) {
            float p[V], m[V], o[V];
            qk_load<V>(p, pg + (size_t)cn_var[b + k] * FG);
            if (!first) qk_load<V>(m, mg + (size_t)(b + k) * FG);
            else {
#pragma unroll
                for (int j = 0; j < V; j++) m[j] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < V; j++) { const float x = p[j] - m[j]; o[j] = acc[j].out(qk_prep<FAM>(x), rule); p[j] = x + o[j]; }
            qk_store_masked<V>(mg + (size_t)(b + k) * FG, o, frozen, any_frozen);
            qk_store_masked<V>(pg + (size_t)cn_var[b + k] * FG, p, frozen, any_frozen);
        }
    }
}

/* ballots of an explicit posterior array (layered schedule): sgn = signbit, hard = !(p >= 0) */
template <int V>
__global__ __launch_bounds__(QK_THREADS) void qk_post_ballots(const float *__restrict__ post, u64 *__restrict__ sgn,
                                                              u64 *__restrict__ hard, int N, const u64 *__restrict__ done)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    /* one wavefront takes 32 consecutive VNs per trip; their 32 sgn + 32 hard ballot words (of frame slot j) leave in ONE store
     * instruction -- lane 2k carries the sgn word of VN k, lane 2k + 1 its hard word (lane-0 stores were one instruction per word) */
    for (int v0 = (blockIdx.x * QK_WAVES + wave) * 32; v0 < N; v0 += gridDim.x * QK_WAVES * 32) {
        uint32_t mlo[V], mhi[V];
#pragma unroll
        for (int j = 0; j < V; j++) { mlo[j] = 0; mhi[j] = 0; }
        /* eight rows asked for before the first is looked at (one row per trip left a wave with 256 bytes in flight: the 256 MB of
         * config 5's posteriors took as long as a third of a sweep); the ballot words reach their lanes by v_writelane */
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            float p[8][V];
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (v0 + k0 + t < N) qk_load<V>(p[t], post + ((size_t)g * N + v0 + k0 + t) * FG + lane * V);      /* wave-uniform */
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const int v = v0 + k0 + t;
                if (v >= N) break;      /* wave-uniform */
#pragma unroll
                for (int j = 0; j < V; j++) {
                    u64 s = __ballot((qk_bits(p[t][j]) >> 31) != 0);
                    u64 h = __ballot(!(p[t][j] >= 0.0f));
                    const size_t bi = ((size_t)g * N + v) * V + j;
                    const u64 dm = done[(size_t)g * V + j];      /* converged frames keep the ballots they converged with */
                    if (dm) {      /* (the old words come through a vector load of a wave-uniform address: readfirstlane keeps s and h in SGPRs for v_writelane) */
                        const u64 os = sgn[bi], oh = hard[bi];
                        s = (s & ~dm) | (qk_uniform64(os) & dm); h = (h & ~dm) | (qk_uniform64(oh) & dm);
                    }
                    mlo[j] = qk_wlane(mlo[j], (uint32_t)s, 2 * (k0 + t)); mhi[j] = qk_wlane(mhi[j], (uint32_t)(s >> 32), 2 * (k0 + t));
                    mlo[j] = qk_wlane(mlo[j], (uint32_t)h, 2 * (k0 + t) + 1); mhi[j] = qk_wlane(mhi[j], (uint32_t)(h >> 32), 2 * (k0 + t) + 1);
                }
            }
        }
        const int v = v0 + (lane >> 1);
        if (v < N) {
#pragma unroll
            for (int j = 0; j < V; j++) ((lane & 1) ? hard : sgn)[((size_t)g * N + v) * V + j] = ((u64)mhi[j] << 32) | mlo[j];
        }
    }
}

/* ------------------------------------------------------------------ syndrome + status -------- */

/*
 * check_syndrome_soft for FG frames at once: XOR the VN ballots of each check, OR over checks.
 * 1-D grid of 8 * ceil(G / 8) * bx workgroups.  Workgroups are dealt round-robin to the 8 XCDs, whose L2s are not coherent with
 * each other: the ballots the variable-node pass just wrote come in over the fabric, so ALL workgroups of a group are placed on ONE
 * XCD (g = id % 8 + 8 * (id / 8 / bx)) and its 8 N bytes of ballots cross the fabric once instead of eight times (measured on the
 * config-2 batch: 68 -> 58 us per pass; what remains is the tag-lookup rate of 64-address gathers, not bytes.  A VN-major form with
 * the syndrome words of a group in LDS (one 1024-thread workgroup per group, ds_xor_b64 per edge) was measured at 84 us: slower).
 */
template <int V>
__global__ __launch_bounds__(256) void qk_syndrome(const u64 *__restrict__ mask, const int *__restrict__ cn_var_t, int max_dc, int M, int N,
                                                   u64 *__restrict__ unsat, const u64 *__restrict__ done, int skip_done, const u64 *__restrict__ synd, int G, int bx,
                                                   int c_lo, int c_hi, int gated)
{
    const int id = blockIdx.x;
    /* fewer than 8 groups: one XCD per group would leave the others idle (config 5's single group ran on 32 of 256 CUs: 94 us per pass);
     * the host then launches G * bx workgroups and a group's chunks go round all XCDs -- its ballots are a few megabytes */
    const int g = G < 8 ? id % G : (id & 7) + 8 * ((id >> 3) / bx);
    const int chunk = G < 8 ? id / G : (id >> 3) % bx;
    if (g >= G) return;
    if (skip_done && qk_group_done<V>(done, g)) return;
    if (gated) {
        /* the pass runs in two launches: checks [0, M/8) first, the rest here.  If every frame of the group that is still running already
         * shows an unsatisfied check, the remaining checks cannot change any verdict: in the iterations before the first convergence
         * (nine of ~fourteen at QBER 2 %) this launch returns at once */
        bool settled = true;
#pragma unroll
        for (int j = 0; j < V; j++) settled = settled && ((__hip_atomic_load(&unsat[(size_t)g * V + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | done[(size_t)g * V + j]) == ~0ull);
        if (settled) return;
    }
    const u64 *mg = mask + (size_t)g * N * V;
    u64 acc[V];
#pragma unroll
    for (int j = 0; j < V; j++) acc[j] = 0;
    for (int c = c_lo + chunk * blockDim.x + threadIdx.x; c < c_hi; c += bx * blockDim.x) {
        u64 s[V];
#pragma unroll
        for (int j = 0; j < V; j++) s[j] = synd ? synd[((size_t)g * M + c) * V + j] : 0ull;      /* H x must equal the target syndrome */
        /* cn_var_t is the check -> VN table transposed to [edge position][check] (-1 past a check's degree): consecutive lanes read
         * consecutive ints, ten independent index loads and then ten independent ballot gathers per trip */
        constexpr int B = 10;
        for (int t0 = 0; t0 < max_dc; t0 += B) {
            int v[B];
#pragma unroll
            for (int t = 0; t < B; t++) v[t] = (t0 + t < max_dc) ? cn_var_t[(size_t)(t0 + t) * M + c] : -1;
#pragma unroll
            for (int t = 0; t < B; t++)
                if (v[t] >= 0) {
                    const u64 *p = mg + (size_t)v[t] * V;
#pragma unroll
                    for (int j = 0; j < V; j++) s[j] ^= p[j];
                }
        }
#pragma unroll
        for (int j = 0; j < V; j++) acc[j] |= s[j];
    }
#pragma unroll
    for (int j = 0; j < V; j++) {
        u64 a = acc[j];
        for (int o = 32; o > 0; o >>= 1) a |= __shfl_xor(a, o);
        if ((threadIdx.x & 63) == 0 && a) atomicOr(&unsat[(size_t)g * V + j], a);
    }
}

/* per-frame bookkeeping after a syndrome pass: AFF3CT's cur_syndrome_depth logic, done bits,
 * iteration counts; clears unsat for the next pass; counts groups still active. */
template <int V>
__global__ void qk_status(u64 *__restrict__ unsat, u64 *__restrict__ done, int *__restrict__ depth, int *__restrict__ iters,
                          int G, int syndrome_depth, int ite_done /* iterations executed so far */, int *__restrict__ active_groups,
                          unsigned long long *__restrict__ work, volatile int *__restrict__ host_report, int seq)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.x;
    const int lane = threadIdx.x;   /* 64 threads */
    bool all = true, was_all = true;
    int left = 0;
#pragma unroll
    for (int j = 0; j < V; j++) {
        const u64 u = unsat[(size_t)g * V + j];
        const u64 d = done[(size_t)g * V + j];
        was_all = was_all && (d == ~0ull);
        const int f = g * FG + lane * V + j;
        const bool was_done = (d >> lane) & 1ull;
        const bool zero = !((u >> lane) & 1ull);
        bool now = false;
        if (!was_done) {
            int cur = depth[f];
            cur = zero ? (cur + 1) % syndrome_depth : 0;
            depth[f] = cur;
            now = zero && cur == 0;
            if (now) iters[f] = ite_done;
        }
        const u64 nd = d | __ballot(now);
        all = all && (nd == ~0ull);
        left += __popcll(~nd);
        if (lane == 0) { done[(size_t)g * V + j] = nd; unsat[(size_t)g * V + j] = 0; }
    }
    /* active_groups[0] = groups with unconverged frames, [1] = those frames; work += 1 per group that ran this iteration */
    if (lane == 0 && !all) { atomicAdd(active_groups, 1); atomicAdd(active_groups + 1, left); }
    if (lane == 0 && !was_all) atomicAdd(work, 1ull);
    /* the workgroup that finishes last hands the two counts to the host through mapped pinned memory (the host spins on the sequence
     * word: no copy, no stream synchronisation in the early-exit loop) and clears them for the next pass */
    if (lane == 0) {
        __threadfence();
        if (atomicAdd(active_groups + 2, 1) == G - 1) {
            const int a0 = atomicExch(active_groups, 0), a1 = atomicExch(active_groups + 1, 0);
            atomicExch(active_groups + 2, 0);
            host_report[0] = a0; host_report[1] = a1;
            __threadfence_system();
            __hip_atomic_store(const_cast<int *>(host_report) + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int V>
__global__ void qk_status_init(u64 *__restrict__ unsat, u64 *__restrict__ done, int *__restrict__ depth, int *__restrict__ iters,
                               int n_frames, int n_ite, unsigned long long *__restrict__ work, int *__restrict__ active_groups)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { *work = 0ull; active_groups[0] = 0; active_groups[1] = 0; active_groups[2] = 0; }
    constexpr int FG = 64 * V;
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int f = g * FG + lane * V + j;
        depth[f] = 0;
        iters[f] = n_ite;
        const u64 pad = __ballot(f >= n_frames);   /* padding frames never run */
        if (lane == 0) { done[(size_t)g * V + j] = pad; unsat[(size_t)g * V + j] = 0; }
    }
}

/* ok[f] = !(unsat bit) after a syndrome pass over the hard ballots; iters copied out */
/*
 * Where slot f of a generation belongs in the caller's batch, or -1 if this generation does not hold the frame's result:
 * origin == NULL: slot f is frame f (generation 0); final_mask (the generation's done words) != NULL: only frames that
 * converged here count, the others moved on to the next generation at a compaction.
 */
template <int V> __device__ __forceinline__ int qk_result_frame(int f, int n_slots, const int *__restrict__ origin, const u64 *__restrict__ final_mask)
{
    constexpr int FG = 64 * V;
    if (f >= n_slots) return -1;
    const int g = f / FG, r = f % FG, lane = r / V, j = r % V;
    if (final_mask && !((final_mask[(size_t)g * V + j] >> lane) & 1ull)) return -1;
    return origin ? origin[f] : f;
}

template <int V>
__global__ void qk_status_out(const u64 *__restrict__ unsat, const int *__restrict__ iters, int *__restrict__ out_iters,
                              int *__restrict__ out_ok, int n_slots, const int *__restrict__ origin, const u64 *__restrict__ final_mask)
{
    constexpr int FG = 64 * V;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int o = qk_result_frame<V>(f, n_slots, origin, final_mask);
    if (o < 0) return;
    const int g = f / FG, r = f % FG, lane = r / V, j = r % V;
    if (out_iters) out_iters[o] = iters[f];
    if (out_ok) out_ok[o] = !((unsat[(size_t)g * V + j] >> lane) & 1ull);
}

/* ------------------------------------------------------------------ load / fetch ------------- */

/* [n_frames][N] frame-major floats -> [G][N][FG]; padding frames get +1 (all-zero word, benign) */
template <int V>
__global__ __launch_bounds__(256) void qk_load_llr(const float *__restrict__ src, float *__restrict__ dst, int N, int n_frames)
{
    constexpr int FG = 64 * V;
    __shared__ float tile[64][65];
    const int g = blockIdx.y, v0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   /* 4 rows at a time */
    for (int r0 = 0; r0 < FG; r0 += 64) {
        for (int r = ty; r < 64; r += 4) {
            const int f = g * FG + r0 + r, v = v0 + tx;
            tile[r][tx] = (f < n_frames && v < N) ? src[(size_t)f * N + v] : 1.0f;
        }
        __syncthreads();
        for (int vv = ty; vv < 64; vv += 4) {
            const int v = v0 + vv;
            if (v < N) dst[((size_t)g * N + v) * FG + r0 + tx] = tile[tx][vv];
        }
        __syncthreads();
    }
}

/* [G][N][FG] -> [n_frames][N] (posterior read-back for tests) */
template <int V>
__global__ __launch_bounds__(256) void qk_unload_f32(const float *__restrict__ src, float *__restrict__ dst, int N, int n_frames)
{
    constexpr int FG = 64 * V;
    __shared__ float tile[64][65];
    const int g = blockIdx.y, v0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r0 = 0; r0 < FG; r0 += 64) {
        for (int vv = ty; vv < 64; vv += 4) {
            const int v = v0 + vv;
            tile[vv][tx] = (v < N) ? src[((size_t)g * N + v) * FG + r0 + tx] : 0.0f;
        }
        __syncthreads();
        for (int r = ty; r < 64; r += 4) {
            const int f = g * FG + r0 + r, v = v0 + tx;
            if (f < n_frames && v < N) dst[(size_t)f * N + v] = tile[tx][r];
        }
        __syncthreads();
    }
}

/*
 * QKD frame formation (BS/src/main.cpp:348-362): packed sifted-key words + per-frame |LLR| (the
 * host computes ln((1-p)/p) from the estimated QBER, so no device log is involved) ->
 * LLR = (1 - 2y) * |LLR| at channel VNs, +-23.02585 at pinned VNs, 0 at punctured VNs.
 * bits[n_frames][W] MSB-first (helpers.h:65-70).  One wavefront = 64 lanes x V frames, 32 VNs per word.
 */
template <int V>
__global__ __launch_bounds__(QK_THREADS) void qk_load_bits(const uint32_t *__restrict__ bits, const float *__restrict__ llr_mag,
                                                           const uint8_t *__restrict__ vn_class, float *__restrict__ dst,
                                                           int N, int W, int n_frames, const int *__restrict__ n_channel)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float mag[V];
    bool live[V];
    int nch[V];                  /* per frame: channel VNs at v >= nch are known (shortened) bits, i.e. pinned */
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int f = g * FG + lane * V + j;
        live[j] = f < n_frames;
        mag[j] = live[j] ? llr_mag[f] : 1.0f;
        nch[j] = (n_channel && live[j]) ? n_channel[f] : N;
    }
    for (int w = blockIdx.x * QK_WAVES + wave; w < W; w += gridDim.x * QK_WAVES) {
        uint32_t word[V];
#pragma unroll
        for (int j = 0; j < V; j++) word[j] = live[j] ? bits[(size_t)(g * FG + lane * V + j) * W + w] : 0u;
        for (int b = 0; b < 32; b++) {
            const int v = w * 32 + b;
            if (v >= N) break;
            const int cls = vn_class ? vn_class[v] : 0;
            float o[V];
#pragma unroll
            for (int j = 0; j < V; j++) {
                const bool y = (word[j] >> (31 - b)) & 1u;
                const float m = (cls == 0) ? (v < nch[j] ? mag[j] : 23.025850929840455f) : (cls == 1 ? 23.025850929840455f : 0.0f);
                o[j] = live[j] ? (y ? -m : m) : 1.0f;
            }
            qk_store<V>(dst + ((size_t)g * N + v) * FG + lane * V, o);
        }
    }
}

/*
 * Per-frame puncturing (BS/src/main.cpp:359-362: LLR[pattern[i]] = 0) for decoders that read an LLR array: erase[n_frames][W]
 * packed MSB-first, a set bit zeroes that VN's channel LLR of that frame.  T = float ([G][N][FG]) or uint8_t (the 8-bit form).
 */
template <int V, typename T>
__global__ __launch_bounds__(QK_THREADS) void qk_erase_rows(const uint32_t *__restrict__ erase, T *__restrict__ llr, int N, int W, int n_frames)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int w = blockIdx.x * QK_WAVES + wave; w < W; w += gridDim.x * QK_WAVES) {
        uint32_t word[V];
        bool any = false;
#pragma unroll
        for (int j = 0; j < V; j++) { const int f = g * FG + lane * V + j; word[j] = f < n_frames ? erase[(size_t)f * W + w] : 0u; any = any || word[j] != 0u; }
        if (!__any(any)) continue;
        for (int b = 0; b < 32; b++) {
            const int v = w * 32 + b;
            if (v >= N) break;
#pragma unroll
            for (int j = 0; j < V; j++)
                if ((word[j] >> (31 - b)) & 1u) llr[((size_t)g * N + v) * FG + lane * V + j] = (T)0;
        }
    }
}

/* per-frame constants of the coded-LLR form: fmag[f] = |LLR| (1 for padding frames), fnch[f] = shortening length (N = none) */
static __global__ __launch_bounds__(256) void qk_load_frame_consts(const float *__restrict__ llr_mag, const int *__restrict__ n_channel,
                                                            float *__restrict__ fmag, int *__restrict__ fnch, int n_frames, int total, int N)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total) return;
    fmag[f] = f < n_frames ? llr_mag[f] : 1.0f;
    fnch[f] = (f < n_frames && n_channel) ? n_channel[f] : N;
}

/* packed MSB-first syndrome words synd_bits[n_frames][Wm] -> per-check ballots synd[G][M][V] (bit = lane) */
template <int V>
__global__ __launch_bounds__(QK_THREADS) void qk_load_syndrome(const uint32_t *__restrict__ bits, u64 *__restrict__ synd, int M, int Wm, int n_frames)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int w = blockIdx.x * QK_WAVES + wave; w < Wm; w += gridDim.x * QK_WAVES) {
        uint32_t word[V];
#pragma unroll
        for (int j = 0; j < V; j++) { const int f = g * FG + lane * V + j; word[j] = f < n_frames ? bits[(size_t)f * Wm + w] : 0u; }
        /* the 32 ballot words of this input word leave in one store per frame slot: lane b carries the word of item 32 w + b */
        u64 mine[V];
#pragma unroll
        for (int j = 0; j < V; j++) mine[j] = 0;
        for (int b = 0; b < 32; b++) {
            if (w * 32 + b >= M) break;
#pragma unroll
            for (int j = 0; j < V; j++) {
                const u64 m = __ballot((word[j] >> (31 - b)) & 1u);
                if (lane == b) mine[j] = m;
            }
        }
        const int c = w * 32 + lane;
        if (lane < 32 && c < M) {
#pragma unroll
            for (int j = 0; j < V; j++) synd[((size_t)g * M + c) * V + j] = mine[j];
        }
    }
}

/* s = H x for packed words x[n_frames][Wn] -> s[n_frames][Wm] (Alice's side of the syndrome form) */
static __global__ __launch_bounds__(256) void qk_syndrome_of_bits(const uint32_t *__restrict__ x, const int *__restrict__ cn_ptr, const int *__restrict__ cn_var,
                                                           uint32_t *__restrict__ s, int M, int Wn, int Wm)
{
    const int f = blockIdx.y;
    const uint32_t *xf = x + (size_t)f * Wn;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < Wm; w += gridDim.x * blockDim.x) {
        uint32_t word = 0;
        for (int b = 0; b < 32; b++) {
            const int c = w * 32 + b;
            if (c >= M) break;
            uint32_t p = 0;
            for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) { const int v = cn_var[k]; p ^= (xf[v >> 5] >> (31 - (v & 31))) & 1u; }
            word |= p << (31 - b);
        }
        s[(size_t)f * Wm + w] = word;
    }
}

/* hard ballots -> packed MSB-first words out[n_frames][W] */
template <int V>
__global__ __launch_bounds__(QK_THREADS) void qk_fetch_packed(const u64 *__restrict__ hard, uint32_t *__restrict__ out,
                                                              int N, int W, int n_slots, const int *__restrict__ origin, const u64 *__restrict__ final_mask)
{
    constexpr int FG = 64 * V;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int dest[V];
#pragma unroll
    for (int j = 0; j < V; j++) dest[j] = qk_result_frame<V>(g * FG + lane * V + j, n_slots, origin, final_mask);
    for (int w = blockIdx.x * QK_WAVES + wave; w < W; w += gridDim.x * QK_WAVES) {
        uint32_t word[V];
#pragma unroll
        for (int j = 0; j < V; j++) word[j] = 0;
        for (int b = 0; b < 32; b++) {
            const int v = w * 32 + b;
            if (v >= N) break;
#pragma unroll
            for (int j = 0; j < V; j++) word[j] |= (uint32_t)((hard[((size_t)g * N + v) * V + j] >> lane) & 1ull) << (31 - b);
        }
#pragma unroll
        for (int j = 0; j < V; j++)
            if (dest[j] >= 0) out[(size_t)dest[j] * W + w] = word[j];
    }
}

/* hard ballots -> V_K[n_frames][K] ints at info_bits_pos (decode_siho's output layout) */
template <int V>
__global__ __launch_bounds__(256) void qk_fetch_info(const u64 *__restrict__ hard, const int *__restrict__ info_pos,
                                                     int *__restrict__ out, int N, int K, int n_slots, const int *__restrict__ origin, const u64 *__restrict__ final_mask)
{
    constexpr int FG = 64 * V;
    const int f = blockIdx.y;
    const int o = qk_result_frame<V>(f, n_slots, origin, final_mask);
    if (o < 0) return;
    const int g = f / FG, r = f % FG, lane = r / V, j = r % V;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < K; i += gridDim.x * blockDim.x) {
        const int v = info_pos[i];
        out[(size_t)o * K + i] = (int)((hard[((size_t)g * N + v) * V + j] >> lane) & 1ull);
    }
}

#endif /* QLDPC_KERNELS_H */
