// tform_bench.hip -- developer experiment: would a "posterior-form" flooding pass move fewer HBM bytes?
//
//   form A (what the engine does): CN pass gathers var_to_chk rows and scatters chk_to_var rows (VN-major
//          slots), VN pass reads / writes contiguous rows:           4E + N rows per iteration
//   form B: messages live CN-major and only chk_to_var is kept; the CN pass reads its contiguous rows, gathers
//          the VN totals T[v] = Y + sum (AFF3CT's `tmp`), forms v2c = T - c2v_old (the same float operation
//          AFF3CT does), and overwrites the rows in place; the VN pass gathers c2v rows and writes T:
//          3E + 2N rows if every T row is fetched from HBM once, 4E + 2N if the gather never hits a cache.
//
// Synthetic IRA-shaped graph of config 2 (M = 13107 checks of 16 random information VNs + 2 accumulator VNs).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/tform_bench tools/tform_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int DC = 18;
typedef float f1 __attribute__((ext_vector_type(1)));

__device__ __forceinline__ float ldnt(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(float* p, float v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ void fold(const float (&v)[DC], float (&o)[DC])
{
    float m1 = 3e38f, m2 = 3e38f; unsigned sg = 0;
#pragma unroll
    for (int k = 0; k < DC; k++) { float a = fabsf(v[k]); sg ^= __float_as_uint(v[k]) & 0x80000000u; float t = fminf(a, m2); m2 = fmaxf(t, m1); m1 = fminf(t, m1); }
#pragma unroll
    for (int k = 0; k < DC; k++) { float a = fabsf(v[k]); float r = (a == m1 ? m2 : m1) * 0.75f; o[k] = __uint_as_float(__float_as_uint(r) | (sg ^ (__float_as_uint(v[k]) & 0x80000000u))); }
}

// A: gather v2c rows by slot, scatter c2v rows by slot. slot[] is [M][DC].
__global__ __launch_bounds__(256) void a_cn(const float* __restrict__ v2c, float* __restrict__ c2v, const int* __restrict__ slot, int M, size_t E)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bpg = (M + 3) / 4;
    const int g = blockIdx.x / bpg, c = (blockIdx.x % bpg) * 4 + wave;
    if (c >= M) return;
    int s[DC];
#pragma unroll
    for (int k = 0; k < DC; k++) s[k] = slot[c * DC + k];
    const float* in = v2c + (size_t)g * E * 64 + lane; float* out = c2v + (size_t)g * E * 64 + lane;
    float v[DC], o[DC];
#pragma unroll
    for (int k = 0; k < DC; k++) v[k] = ldnt(in + (size_t)s[k] * 64);
    fold(v, o);
#pragma unroll
    for (int k = 0; k < DC; k++) stnt(out + (size_t)s[k] * 64, o[k]);
}
// A: VN pass, contiguous slots vptr[v]..vptr[v+1]
__global__ __launch_bounds__(256) void a_vn(const float* __restrict__ c2v, const float* __restrict__ llr, float* __restrict__ v2c, const int* __restrict__ vptr, int N, size_t E)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bpg = (N + 3) / 4;
    const int g = blockIdx.x / bpg, v = (blockIdx.x % bpg) * 4 + wave;
    if (v >= N) return;
    const int s0 = vptr[v], s1 = vptr[v + 1];
    const float* in = c2v + ((size_t)g * E + s0) * 64 + lane; float* out = v2c + ((size_t)g * E + s0) * 64 + lane;
    float m[16]; float sum = 0.f; const int d = min(s1 - s0, 16);
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) m[k] = ldnt(in + (size_t)k * 64);
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) sum += m[k];
    const float t = ldnt(llr + ((size_t)g * N + v) * 64 + lane) + sum;
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) stnt(out + (size_t)k * 64, t - m[k]);
}
// B: CN-major rows, T gathered by var[] = [M][DC]
template <bool NT_T>
__global__ __launch_bounds__(256) void b_cn(float* __restrict__ c2v, const float* __restrict__ T, const int* __restrict__ var, int M, int N, size_t E)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bpg = (M + 3) / 4;
    const int g = blockIdx.x / bpg, c = (blockIdx.x % bpg) * 4 + wave;
    if (c >= M) return;
    int s[DC];
#pragma unroll
    for (int k = 0; k < DC; k++) s[k] = var[c * DC + k];
    float* row = c2v + ((size_t)g * E + (size_t)c * DC) * 64 + lane; const float* t = T + (size_t)g * N * 64 + lane;
    float v[DC], o[DC], tv[DC];
#pragma unroll
    for (int k = 0; k < DC; k++) v[k] = ldnt(row + (size_t)k * 64);
#pragma unroll
    for (int k = 0; k < DC; k++) tv[k] = NT_T ? ldnt(t + (size_t)s[k] * 64) : t[(size_t)s[k] * 64];
#pragma unroll
    for (int k = 0; k < DC; k++) v[k] = tv[k] - v[k];
    fold(v, o);
#pragma unroll
    for (int k = 0; k < DC; k++) stnt(row + (size_t)k * 64, o[k]);
}
// B: VN pass gathers c2v rows (cnslot[] in VN order), writes T
__global__ __launch_bounds__(256) void b_vn(const float* __restrict__ c2v, const float* __restrict__ llr, float* __restrict__ T, const int* __restrict__ vptr, const int* __restrict__ cnslot, int N, size_t E)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bpg = (N + 3) / 4;
    const int g = blockIdx.x / bpg, v = (blockIdx.x % bpg) * 4 + wave;
    if (v >= N) return;
    const int s0 = vptr[v], s1 = vptr[v + 1];
    const int d = min(s1 - s0, 16);
    int s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = k < d ? cnslot[s0 + k] : 0;
    const float* in = c2v + (size_t)g * E * 64 + lane;
    float m[16]; float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) m[k] = ldnt(in + (size_t)s[k] * 64);
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) sum += m[k];
    T[((size_t)g * N + v) * 64 + lane] = ldnt(llr + ((size_t)g * N + v) * 64 + lane) + sum;
}

// B': VN-major slots as in form A; the CN pass gathers c2v_old and T, scatters c2v_new in place; the VN pass is contiguous
__global__ __launch_bounds__(256) void b2_cn(float* __restrict__ c2v, const float* __restrict__ T, const int* __restrict__ slot, const int* __restrict__ var, int M, int N, size_t E)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bpg = (M + 3) / 4;
    const int g = blockIdx.x / bpg, c = (blockIdx.x % bpg) * 4 + wave;
    if (c >= M) return;
    int s[DC], w[DC];
#pragma unroll
    for (int k = 0; k < DC; k++) { s[k] = slot[c * DC + k]; w[k] = var[c * DC + k]; }
    float* base = c2v + (size_t)g * E * 64 + lane; const float* t = T + (size_t)g * N * 64 + lane;
    float v[DC], o[DC], tv[DC];
#pragma unroll
    for (int k = 0; k < DC; k++) v[k] = ldnt(base + (size_t)s[k] * 64);
#pragma unroll
    for (int k = 0; k < DC; k++) tv[k] = t[(size_t)w[k] * 64];
#pragma unroll
    for (int k = 0; k < DC; k++) v[k] = tv[k] - v[k];
    fold(v, o);
#pragma unroll
    for (int k = 0; k < DC; k++) stnt(base + (size_t)s[k] * 64, o[k]);
}
__global__ __launch_bounds__(256) void b2_vn(const float* __restrict__ c2v, const float* __restrict__ llr, float* __restrict__ T, const int* __restrict__ vptr, int N, size_t E)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bpg = (N + 3) / 4;
    const int g = blockIdx.x / bpg, v = (blockIdx.x % bpg) * 4 + wave;
    if (v >= N) return;
    const int s0 = vptr[v], s1 = vptr[v + 1];
    const float* in = c2v + ((size_t)g * E + s0) * 64 + lane;
    float m[16]; float sum = 0.f; const int d = min(s1 - s0, 16);
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) m[k] = ldnt(in + (size_t)k * 64);
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < d) sum += m[k];
    T[((size_t)g * N + v) * 64 + lane] = ldnt(llr + ((size_t)g * N + v) * 64 + lane) + sum;
}

int main(int argc, char** argv)
{
    const int M = 13107, K = 52429, N = M + K, G = argc > 1 ? atoi(argv[1]) : 64;
    const size_t E = (size_t)M * DC;
    std::mt19937 rng(7);
    std::vector<int> var(E);
    for (int c = 0; c < M; c++) {
        int* r = &var[(size_t)c * DC];
        for (int k = 0; k < DC - 2;) { int v = rng() % K; bool dup = false; for (int j = 0; j < k; j++) dup |= r[j] == v; if (!dup) r[k++] = v; }
        std::sort(r, r + DC - 2);
        r[DC - 2] = K + (c == 0 ? M - 1 : c - 1); r[DC - 1] = K + c;      // accumulator chain (wrap only to keep dc fixed)
    }
    // VN-major slots
    std::vector<int> vptr(N + 1, 0);
    for (size_t e = 0; e < E; e++) vptr[var[e] + 1]++;
    int maxd = 0; for (int v = 0; v < N; v++) { maxd = std::max(maxd, vptr[v + 1]); vptr[v + 1] += vptr[v]; }
    std::vector<int> fill(vptr.begin(), vptr.end() - 1), slot(E), cnslot(E);
    for (size_t e = 0; e < E; e++) { int s = fill[var[e]]++; slot[e] = s; cnslot[s] = (int)e; }
    printf("M %d N %d E %zu groups %d (frames %d) max dv %d\n", M, N, E, G, G * 64, maxd);
    if (maxd > 16) { printf("dv > 16: regenerate\n"); return 1; }

    const size_t msg = (size_t)G * E * 64 * 4, nb = (size_t)G * N * 64 * 4;
    float *v2c, *c2v, *llr, *T; int *d_var, *d_slot, *d_vptr, *d_cnslot;
    CK(hipMalloc(&v2c, msg)); CK(hipMalloc(&c2v, msg)); CK(hipMalloc(&llr, nb)); CK(hipMalloc(&T, nb));
    CK(hipMemset(v2c, 0x3c, msg)); CK(hipMemset(c2v, 0x3c, msg)); CK(hipMemset(llr, 0x3c, nb)); CK(hipMemset(T, 0x3c, nb));
    CK(hipMalloc(&d_var, E * 4)); CK(hipMalloc(&d_slot, E * 4)); CK(hipMalloc(&d_vptr, (N + 1) * 4)); CK(hipMalloc(&d_cnslot, E * 4));
    CK(hipMemcpy(d_var, var.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_slot, slot.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vptr, vptr.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cnslot, cnslot.data(), E * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double rows, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        float tot = 0, best = 1e9; const int reps = 10;
        for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = std::min(best, ms); }
        printf("%-34s avg %.3f ms best %.3f ms   %.0f GB/s of %.2f GB\n", name, tot / reps, best, rows * 256.0 * G / (tot / reps) / 1e6, rows * 256.0 * G / 1e9);
        return tot / reps;
    };
    const int gcn = G * ((M + 3) / 4), gvn = G * ((N + 3) / 4);
    float acn = timeit("A cn  gather+scatter   (2E)", 2.0 * E, [&] { a_cn<<<gcn, 256>>>(v2c, c2v, d_slot, M, E); });
    float avn = timeit("A vn  contiguous       (2E+N)", 2.0 * E + N, [&] { a_vn<<<gvn, 256>>>(c2v, llr, v2c, d_vptr, N, E); });
    float bcn = timeit("B cn  in place + T     (2E+N..3E)", 2.0 * E + N, [&] { b_cn<false><<<gcn, 256>>>(c2v, T, d_var, M, N, E); });
    float bcn2 = timeit("B cn  in place + T(nt) (2E+N..3E)", 2.0 * E + N, [&] { b_cn<true><<<gcn, 256>>>(c2v, T, d_var, M, N, E); });
    float bvn = timeit("B vn  gather, write T  (E+2N)", 1.0 * E + 2.0 * N, [&] { b_vn<<<gvn, 256>>>(c2v, llr, T, d_vptr, d_cnslot, N, E); });
    float b2cn = timeit("B' cn gather c2v+T, scatter (2E+N..3E)", 2.0 * E + N, [&] { b2_cn<<<gcn, 256>>>(c2v, T, d_slot, d_var, M, N, E); });
    float b2vn = timeit("B' vn contiguous, write T  (E+2N)", 1.0 * E + 2.0 * N, [&] { b2_vn<<<gvn, 256>>>(c2v, llr, T, d_vptr, N, E); });
    printf("iteration: B' %.3f ms (%.1f %%)\n", b2cn + b2vn, 100.0 * (b2cn + b2vn) / (acn + avn) - 100.0);
    printf("iteration: A %.3f ms   B %.3f ms (%.1f %%)   B(nt) %.3f ms (%.1f %%)\n", acn + avn, bcn + bvn, 100.0 * (bcn + bvn) / (acn + avn) - 100.0,
           bcn2 + bvn, 100.0 * (bcn2 + bvn) / (acn + avn) - 100.0);
    return 0;
}
