/*
 * qldpc_privamp.hip -- privacy amplification (the step right after reconciliation) on the GPU.
 *
 * Replaces the hash loop of privAmp_doPrivAmp (subcomponents/priv_amp.c:213-218):
 *     for i < finalKeyBits: m = 0; for j < numwords: m ^= key[j] & rnd_getPrngValue2_32(&state);
 *                           finalkey bit i = parity(m)
 * where rnd_getPrngValue2_32 (subcomponents/rnd.c:118-127) steps a 32-bit LFSR 32 times
 * (state = (state << 1) + parity(state & 0xe0000200), subcomponents/rnd.h:46).  On the CPU this is
 * finalKeyBits x numwords x 32 serial bit steps (~8 s for a 57 kbit block).
 * The LFSR is linear over GF(2): output bit i starts from R^i(seed) with R = A^(32 numwords), reached by
 * 17 precomputed jump matrices, and 10 LFSR steps are taken per word-parallel operation (the taps
 * 31,30,29,9 only read bits that exist before the chunk).  Pure integer: bit-exact with the reference.
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
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

#define PA_FEEDBACK 0xe0000200u
#define PA_JUMPS 17                /* output bit index < 2^17 */

/* c <= 10 LFSR steps at once: new bits b_k = s[31-k]^s[30-k]^s[29-k]^s[9-k], b_0 ends at bit c-1 */
__host__ __device__ static inline uint32_t lfsr_chunk(uint32_t s, int c)
{
    const uint32_t nb = ((s >> (32 - c)) ^ (s >> (31 - c)) ^ (s >> (30 - c)) ^ (s >> (10 - c))) & ((1u << c) - 1u);
    return (s << c) | nb;
}
__host__ __device__ static inline uint32_t lfsr_step32(uint32_t s)
{
    s = lfsr_chunk(s, 10);
    s = lfsr_chunk(s, 10);
    s = lfsr_chunk(s, 10);
    return lfsr_chunk(s, 2);
}

struct pa_jumps { uint32_t col[PA_JUMPS][32]; };   /* col[k][b] = R^(2^k) applied to basis bit b */

__device__ static inline uint32_t pa_apply(const uint32_t *col, uint32_t s)
{
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) r ^= ((s >> b) & 1u) ? col[b] : 0u;
    return r;
}

__global__ __launch_bounds__(256) void qp_privamp(const uint32_t *__restrict__ key, int numwords, uint32_t seed, int final_bits,
                                                  pa_jumps J, uint32_t *__restrict__ out)
{
    extern __shared__ uint32_t s_key[];
    __shared__ uint32_t s_col[PA_JUMPS][32];
    for (int j = threadIdx.x; j < numwords; j += blockDim.x) s_key[j] = key[j];
    for (int t = threadIdx.x; t < PA_JUMPS * 32; t += blockDim.x) s_col[t / 32][t % 32] = J.col[t / 32][t % 32];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = seed;
    for (int k = 0; k < PA_JUMPS; k++) if ((i >> k) & 1) s = pa_apply(s_col[k], s);
    uint32_t m = 0;
    for (int j = 0; j < numwords; j++) { s = lfsr_step32(s); m ^= s_key[j] & s; }
    const bool bit = (i < final_bits) && (__popc(m) & 1);
    const unsigned long long b = __ballot(bit);
    const int lane = threadIdx.x & 63;
    if (lane < 2) {
        const int w = (blockIdx.x * blockDim.x + (threadIdx.x & ~63)) / 32 + lane;
        if (w < (final_bits + 31) / 32) out[w] = __brev((uint32_t)(b >> (32 * lane)));
    }
}

/* linear maps over GF(2)^32 as images of the 32 basis bits */
static uint32_t map_apply(const uint32_t *col, uint32_t s)
{
    uint32_t r = 0;
    for (int b = 0; b < 32; b++) if ((s >> b) & 1u) r ^= col[b];
    return r;
}
static void map_compose(const uint32_t *outer, const uint32_t *inner, uint32_t *res)   /* res = outer o inner */
{
    uint32_t t[32];
    for (int b = 0; b < 32; b++) t[b] = map_apply(outer, inner[b]);
    memcpy(res, t, sizeof(t));
}
static void map_pow(const uint32_t *base, uint64_t e, uint32_t *res)
{
    uint32_t acc[32], sq[32];
    for (int b = 0; b < 32; b++) acc[b] = 1u << b;
    memcpy(sq, base, sizeof(sq));
    while (e) {
        if (e & 1) map_compose(sq, acc, acc);
        map_compose(sq, sq, sq);
        e >>= 1;
    }
    memcpy(res, acc, sizeof(acc));
}

static int build_jumps(int numwords, pa_jumps *J)
{
    uint32_t a32[32];
    for (int b = 0; b < 32; b++) a32[b] = lfsr_step32(1u << b);     /* one word = 32 steps (linear, so basis images suffice) */
    uint32_t R[32];
    map_pow(a32, (uint64_t)numwords, R);                            /* one output bit = numwords words */
    memcpy(J->col[0], R, sizeof(R));
    for (int k = 1; k < PA_JUMPS; k++) map_compose(J->col[k - 1], J->col[k - 1], J->col[k]);
    return QLDPC_OK;
}

extern "C" int qldpc_privamp_dev(const uint32_t *d_key_words, int workbits, uint32_t seed, int final_bits, uint32_t *d_final_words, void *hip_stream)
{
    if (!d_key_words || !d_final_words) return QLDPC_EINVAL;
    if (workbits <= 0 || final_bits < 0 || final_bits >= (1 << PA_JUMPS)) { qldpc_set_error("privamp: workbits=%d final_bits=%d", workbits, final_bits); return QLDPC_ESIZE; }
    if (final_bits == 0) return QLDPC_OK;
    const int numwords = (workbits + 31) / 32;
    if ((size_t)numwords * 4 > 60 * 1024) { qldpc_set_error("privamp: key of %d words does not fit LDS", numwords); return QLDPC_ESIZE; }
    pa_jumps J;
    build_jumps(numwords, &J);
    const int blocks = (final_bits + 255) / 256;
    hipLaunchKernelGGL(qp_privamp, dim3((unsigned)blocks), dim3(256), (size_t)numwords * 4, (hipStream_t)hip_stream, d_key_words, numwords, seed, final_bits, J, d_final_words);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { qldpc_set_error("privamp launch: %s", hipGetErrorString(e)); return QLDPC_EHIP; }
    return QLDPC_OK;
}

/* host buffers, as privAmp_doPrivAmp holds them: key = mainBufPtr (tail bits past workbits are masked like
 * priv_amp.c:196-198), out = ceil(final_bits/32) words, MSB-first */
extern "C" int qldpc_privamp(int device, const uint32_t *key_words, int workbits, uint32_t seed, int final_bits, uint32_t *final_words)
{
    if (!key_words || !final_words) return QLDPC_EINVAL;
    if (workbits <= 0 || final_bits < 0) return QLDPC_ESIZE;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { qldpc_set_error("no HIP device visible: libqldpc has no CPU fallback"); return QLDPC_ENODEV; }
    if (device < 0 || device >= ndev) return QLDPC_ENODEV;
    HIPCHK(hipSetDevice(device));
    const int numwords = (workbits + 31) / 32, outwords = (final_bits + 31) / 32;
    if (final_bits == 0) return QLDPC_OK;
    std::vector<uint32_t> key(key_words, key_words + numwords);
    if (workbits & 31) key[(size_t)numwords - 1] &= 0xFFFFFFFFu << (32 - (workbits & 31));
    uint32_t *d_key = nullptr, *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_key, sizeof(uint32_t) * (size_t)numwords));
    if (hipMalloc((void **)&d_out, sizeof(uint32_t) * (size_t)outwords) != hipSuccess) { (void)hipFree(d_key); return QLDPC_ENOMEM; }
    int rc = QLDPC_OK;
    if (hipMemcpy(d_key, key.data(), sizeof(uint32_t) * (size_t)numwords, hipMemcpyHostToDevice) != hipSuccess) rc = QLDPC_EHIP;
    if (!rc) rc = qldpc_privamp_dev(d_key, workbits, seed, final_bits, d_out, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = QLDPC_EHIP;
    if (!rc && hipMemcpy(final_words, d_out, sizeof(uint32_t) * (size_t)outwords, hipMemcpyDeviceToHost) != hipSuccess) rc = QLDPC_EHIP;
    (void)hipFree(d_key); (void)hipFree(d_out);
    return rc;
}
