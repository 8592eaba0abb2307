// sc1_rows.hip -- developer microbenchmark (DESIGN section 8 #6c): what do agent-coherent (sc1) accesses cost for the decoder's 256-byte rows
// (one dword per lane)?  One wave gathers R random rows and scatters R rows, as a layer-kernel wave does, with
//   mode 0: plain loads / plain stores     mode 1: plain loads / sc1 stores (__hip_atomic_store relaxed, agent)
//   mode 2: sc1 loads / sc1 stores          mode 3: sc1 loads / plain stores
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/sc1_rows tools/sc1_rows.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int R, int MODE>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ a, float* __restrict__ b, const int* __restrict__ idx, int n_items)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int it = blockIdx.x * 4 + wave;
    if (it >= n_items) return;
    int s[R];
#pragma unroll
    for (int k = 0; k < R; k++) s[k] = idx[it * R + k];
    float x[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        const float *p = a + (size_t)s[k] * 64 + lane;
        if (MODE == 2 || MODE == 3) x[k] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else x[k] = *p;
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        float *p = b + (size_t)s[k] * 64 + lane;
        if (MODE == 1 || MODE == 2) __hip_atomic_store(p, x[k] + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = x[k] + 1.0f;
    }
}

template <int MODE> static double run(const float *a, float *b, const int *idx, int n_items, int reps)
{
    constexpr int R = 18;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rows<R, MODE>), dim3((n_items + 3) / 4), dim3(256), 0, 0, a, b, idx, n_items);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_rows<R, MODE>), dim3((n_items + 3) / 4), dim3(256), 0, 0, a, b, idx, n_items);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * R * 256.0 * n_items * reps / (ms * 1e-3) / 1e9;
}

int main()
{
    const size_t rows = (size_t)1 << 22;      // 1 GiB of 256-byte rows
    float *a, *b; int *idx;
    CK(hipMalloc(&a, rows * 256)); CK(hipMalloc(&b, rows * 256));
    CK(hipMemset(a, 0, rows * 256));
    for (int n_items : {6667, 26668, 200000}) {
        std::vector<int> h((size_t)n_items * 18);
        std::mt19937 g(1);
        for (auto &v : h) v = (int)(g() % rows);
        CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const int reps = n_items > 100000 ? 10 : 200;
        printf("%6d waves x 18 rows gathered + 18 scattered: plain/plain %7.0f GB/s | plain loads, sc1 stores %7.0f | sc1/sc1 %7.0f | sc1 loads, plain stores %7.0f\n", n_items,
               run<0>(a, b, idx, n_items, reps), run<1>(a, b, idx, n_items, reps), run<2>(a, b, idx, n_items, reps), run<3>(a, b, idx, n_items, reps));
        CK(hipFree(idx));
    }
    return 0;
}
