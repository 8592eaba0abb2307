/*
 * qldpc_encoder.hip -- Alice-side encoder: the parity bits that make H x = 0.
 *
 * Replaces m.encoder->encode(ref_bits, enc_bits) of the harness (BS/src/main.cpp:341):
 *   "IRA"      Encoder_LDPC_DVBS2 / Encoder_LDPC_from_IRA (BS/src/main.cpp:190): dual-diagonal
 *              accumulator, p_0 = s_0, p_c = p_{c-1} ^ s_c with s = H_info u.
 *   "IDENTITY" Encoder_LDPC_from_H(K, N, H, "IDENTITY", ...) (VAR/main.cpp (alist-v1.0.1):142-145):
 *              systematic form by GF(2) elimination, parity = A u; get_info_bits_pos() (:147-159).
 *   "LU_DEC"   the harness' other G_method (:135): the same elimination with the pivots searched from the last column down, so the
 *              parity bits sit at the end of the codeword wherever H allows.  AFF3CT's own column permutation has no vector in the
 *              reference (parity unpinned): take the positions from qldpc_encoder_info_bits_pos, as the harness does (:147-159).
 *   "QC"       Encoder_LDPC_from_QC (VAR/main.cpp (qc):145): x = [u | H2^-1 H1 u], info bits = the first N - M positions;
 *              refused when the last M columns of H are singular (AFF3CT throws there too).
 * Bits are packed MSB-first in 32-bit words like ProcessBlock.mainBufPtr (helpers.h:65-70).
 */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
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

struct qldpc_encoder {
    int N, M, K, R;          /* R = number of parity positions (rank of H) */
    int ira;
    int device;
    std::vector<int> info_pos, parity_pos;
    int *d_map;              /* [N] >= 0: info index, < 0: -1 - parity index */
    /* IRA */
    int *d_cn_ptr, *d_cn_var;
    /* IDENTITY */
    uint32_t *d_A; int wpr;  /* [R][wpr] MSB-first rows over the info index */
    uint8_t *d_par; size_t par_frames;   /* workspace: parity bytes [frames][R] of the call in flight (one encoder is not re-entrant) */
};

/* the workspace holds n_frames frames: grown (never shrunk) with plain hipMalloc -- after a device-wide synchronisation, so nothing in
 * flight still uses the old block -- and kept; qldpc_encoder_reserve sizes it up front so that no call allocates later */
static int ensure_workspace(qldpc_encoder *e, int n_frames)
{
    if ((size_t)n_frames <= e->par_frames) return QLDPC_OK;
    HIPCHK(hipDeviceSynchronize());
    if (e->d_par) (void)hipFree(e->d_par);
    e->d_par = nullptr; e->par_frames = 0;
    if (hipMalloc((void **)&e->d_par, (size_t)n_frames * (size_t)e->R) != hipSuccess) { qldpc_set_error("encoder workspace: hipMalloc(%zu bytes) failed", (size_t)n_frames * (size_t)e->R); return QLDPC_ENOMEM; }
    e->par_frames = (size_t)n_frames;
    return QLDPC_OK;
}

__device__ __forceinline__ uint32_t getbit(const uint32_t *w, int i) { return (w[i >> 5] >> (31 - (i & 31))) & 1u; }

/* s[f][c] = XOR of the info bits on check c (info VN v < K is info index v) */
__global__ void qe_ira_syndrome(const uint32_t *__restrict__ info, const int *__restrict__ cn_ptr, const int *__restrict__ cn_var,
                                uint8_t *__restrict__ s, int M, int K, int Wk)
{
    const int f = blockIdx.y;
    const uint32_t *u = info + (size_t)f * Wk;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < M; c += gridDim.x * blockDim.x) {
        uint32_t p = 0;
        for (int k = cn_ptr[c]; k < cn_ptr[c + 1]; k++) { const int v = cn_var[k]; if (v < K) p ^= getbit(u, v); }
        s[(size_t)f * M + c] = (uint8_t)p;
    }
}

/* in-place prefix XOR over s[f][0..M): one wavefront per frame, each lane owns a contiguous chunk */
__global__ __launch_bounds__(64) void qe_prefix_xor(uint8_t *__restrict__ s, int M)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    uint8_t *p = s + (size_t)f * M;
    const int chunk = (M + 63) / 64, lo = lane * chunk, hi = min(M, lo + chunk);
    uint32_t t = 0;
    for (int c = lo; c < hi; c++) t ^= p[c];
    uint32_t incl = t;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if (lane >= o) incl ^= y; }
    uint32_t run = incl ^ t;   /* exclusive prefix of this lane's chunk */
    for (int c = lo; c < hi; c++) { run ^= p[c]; p[c] = (uint8_t)run; }
}

/* p[f][j] = <A[j], u_f> over GF(2) */
__global__ void qe_dense_parity(const uint32_t *__restrict__ info, const uint32_t *__restrict__ A, uint8_t *__restrict__ p, int R, int wpr)
{
    const int f = blockIdx.y;
    const uint32_t *u = info + (size_t)f * wpr;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < R; j += gridDim.x * blockDim.x) {
        uint32_t acc = 0;
        for (int w = 0; w < wpr; w++) acc ^= A[(size_t)j * wpr + w] & u[w];
        p[(size_t)f * R + j] = (uint8_t)(__popc(acc) & 1);
    }
}

/* codeword words from info words + parity bytes through the position map */
__global__ void qe_assemble(const uint32_t *__restrict__ info, const uint8_t *__restrict__ par, const int *__restrict__ map,
                            uint32_t *__restrict__ cw, int N, int Wn, int Wk, int R)
{
    const int f = blockIdx.y;
    for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < Wn; w += gridDim.x * blockDim.x) {
        uint32_t word = 0;
        for (int b = 0; b < 32; b++) {
            const int n = w * 32 + b;
            if (n >= N) break;
            const int m = map[n];
            const uint32_t bit = m >= 0 ? getbit(info + (size_t)f * Wk, m) : par[(size_t)f * R + (-1 - m)];
            word |= bit << (31 - b);
        }
        cw[(size_t)f * Wn + w] = word;
    }
}

extern "C" void qldpc_encoder_free(qldpc_encoder *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipFree(e->d_map); (void)hipFree(e->d_cn_ptr); (void)hipFree(e->d_cn_var); (void)hipFree(e->d_A); (void)hipFree(e->d_par);
    delete e;
}

static int enc_create(const qldpc_code *code, const char *method, int device, qldpc_encoder *e)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { qldpc_set_error("no HIP device visible: libqldpc has no CPU fallback"); return QLDPC_ENODEV; }
    if (device < 0 || device >= ndev) { qldpc_set_error("device %d out of range", device); return QLDPC_ENODEV; }
    HIPCHK(hipSetDevice(device));
    e->device = device; e->N = code->N; e->M = code->M;
    std::vector<int> map((size_t)code->N);
    if (!strcmp(method, "IRA")) {
        if (code->ira_K <= 0) { qldpc_set_error("encoder IRA: H has no dual-diagonal parity part"); return QLDPC_EUNSUPPORTED; }
        e->ira = 1; e->K = code->ira_K; e->R = code->M;
        for (int n = 0; n < e->N; n++) { if (n < e->K) { e->info_pos.push_back(n); map[(size_t)n] = n; } else { e->parity_pos.push_back(n); map[(size_t)n] = -1 - (n - e->K); } }
        HIPCHK(hipMalloc((void **)&e->d_cn_ptr, sizeof(int) * ((size_t)e->M + 1)));
        HIPCHK(hipMalloc((void **)&e->d_cn_var, sizeof(int) * (size_t)code->E));
        HIPCHK(hipMemcpy(e->d_cn_ptr, code->cn_ptr, sizeof(int) * ((size_t)e->M + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_cn_var, code->cn_var, sizeof(int) * (size_t)code->E, hipMemcpyHostToDevice));
    } else if (!strcmp(method, "IDENTITY") || !strcmp(method, "LU_DEC") || !strcmp(method, "QC")) {
        int *piv = nullptr, *fr = nullptr; uint64_t *A = nullptr; int wpr64 = 0;
        const int r = qldpc_gf2_systematic_ord(code, method[0] == 'I' ? 0 : method[0] == 'L' ? 1 : 2, &piv, &fr, &A, &wpr64);
        if (r < 0) return r;
        e->ira = 0; e->R = r; e->K = e->N - r; e->wpr = (e->K + 31) / 32;
        for (int j = 0; j < r; j++) { e->parity_pos.push_back(piv[j]); map[(size_t)piv[j]] = -1 - j; }
        for (int i = 0; i < e->K; i++) { e->info_pos.push_back(fr[i]); map[(size_t)fr[i]] = i; }
        std::vector<uint32_t> A32((size_t)r * (size_t)(e->wpr > 0 ? e->wpr : 1), 0u);
        for (int j = 0; j < r; j++)
            for (int i = 0; i < e->K; i++)
                if (A[(size_t)j * wpr64 + i / 64] >> (i % 64) & 1) A32[(size_t)j * e->wpr + i / 32] |= 1u << (31 - i % 32);
        free(piv); free(fr); free(A);
        HIPCHK(hipMalloc((void **)&e->d_A, sizeof(uint32_t) * A32.size()));
        HIPCHK(hipMemcpy(e->d_A, A32.data(), sizeof(uint32_t) * A32.size(), hipMemcpyHostToDevice));
    } else {
        qldpc_set_error("encoder: unknown method '%s' (IRA | IDENTITY | LU_DEC | QC)", method);
        return QLDPC_EINVAL;
    }
    HIPCHK(hipMalloc((void **)&e->d_map, sizeof(int) * (size_t)e->N));
    HIPCHK(hipMemcpy(e->d_map, map.data(), sizeof(int) * (size_t)e->N, hipMemcpyHostToDevice));
    return QLDPC_OK;
}

extern "C" int qldpc_encoder_create(const qldpc_code *code, const char *method, int device, qldpc_encoder **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = nullptr;
    if (!code || !method) return QLDPC_EINVAL;
    qldpc_encoder *e = new (std::nothrow) qldpc_encoder();
    if (!e) return QLDPC_ENOMEM;
    e->d_map = nullptr; e->d_cn_ptr = nullptr; e->d_cn_var = nullptr; e->d_A = nullptr; e->wpr = 0; e->d_par = nullptr; e->par_frames = 0;
    int rc = enc_create(code, method, device, e);
    if (rc != QLDPC_OK) { qldpc_encoder_free(e); return rc; }
    *out = e;
    return QLDPC_OK;
}

extern "C" int qldpc_encoder_reserve(qldpc_encoder *e, int max_frames)
{
    if (!e || max_frames <= 0) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    return ensure_workspace(e, max_frames);
}

extern "C" int qldpc_encoder_k(const qldpc_encoder *e) { return e ? e->K : QLDPC_EINVAL; }

extern "C" int qldpc_encoder_info_bits_pos(const qldpc_encoder *e, int *pos)
{
    if (!e || !pos) return QLDPC_EINVAL;
    memcpy(pos, e->info_pos.data(), sizeof(int) * (size_t)e->K);
    return QLDPC_OK;
}

extern "C" int qldpc_encode_packed_dev(qldpc_encoder *e, const uint32_t *d_info, uint32_t *d_cw, int n_frames, void *hip_stream)
{
    if (!e || !d_info || !d_cw || n_frames <= 0) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int Wk = (e->K + 31) / 32, Wn = (e->N + 31) / 32;
    int rc = ensure_workspace(e, n_frames);
    if (rc) return rc;
    uint8_t *par = e->d_par;
    if (e->ira) {
        hipLaunchKernelGGL(qe_ira_syndrome, dim3((unsigned)((e->M + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, d_info, e->d_cn_ptr, e->d_cn_var, par, e->M, e->K, Wk);
        hipLaunchKernelGGL(qe_prefix_xor, dim3((unsigned)n_frames), dim3(64), 0, st, par, e->M);
    } else {
        hipLaunchKernelGGL(qe_dense_parity, dim3((unsigned)((e->R + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, d_info, e->d_A, par, e->R, e->wpr);
    }
    hipLaunchKernelGGL(qe_assemble, dim3((unsigned)((Wn + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, d_info, par, e->d_map, d_cw, e->N, Wn, Wk, e->R);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) { qldpc_set_error("encode: kernel launch -> %s", hipGetErrorString(le)); return QLDPC_EHIP; }
    return QLDPC_OK;
}

/* host mirror: ints in, ints out (the AFF3CT buffers are std::vector<int>, BS/src/main.cpp:117-118) */
extern "C" int qldpc_encode(qldpc_encoder *e, const int *U_K, int *X_N, int n_frames)
{
    if (!e || !U_K || !X_N || n_frames <= 0) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    const int Wk = (e->K + 31) / 32, Wn = (e->N + 31) / 32;
    std::vector<uint32_t> info((size_t)n_frames * Wk, 0u), cw((size_t)n_frames * Wn);
    for (int f = 0; f < n_frames; f++)
        for (int i = 0; i < e->K; i++)
            if (U_K[(size_t)f * e->K + i] & 1) info[(size_t)f * Wk + i / 32] |= 1u << (31 - i % 32);
    uint32_t *d_info = nullptr, *d_cw = nullptr;
    HIPCHK(hipMalloc((void **)&d_info, sizeof(uint32_t) * info.size()));
    if (hipMalloc((void **)&d_cw, sizeof(uint32_t) * cw.size()) != hipSuccess) { (void)hipFree(d_info); return QLDPC_ENOMEM; }
    int rc = QLDPC_OK;
    if (hipMemcpy(d_info, info.data(), sizeof(uint32_t) * info.size(), hipMemcpyHostToDevice) != hipSuccess) rc = QLDPC_EHIP;
    if (!rc) rc = qldpc_encode_packed_dev(e, d_info, d_cw, n_frames, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = QLDPC_EHIP;
    if (!rc && hipMemcpy(cw.data(), d_cw, sizeof(uint32_t) * cw.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = QLDPC_EHIP;
    (void)hipFree(d_info); (void)hipFree(d_cw);
    if (rc) return rc;
    for (int f = 0; f < n_frames; f++)
        for (int n = 0; n < e->N; n++) X_N[(size_t)f * e->N + n] = (int)((cw[(size_t)f * Wn + n / 32] >> (31 - n % 32)) & 1u);
    return QLDPC_OK;
}
