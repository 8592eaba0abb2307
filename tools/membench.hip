// membench.hip -- developer microbenchmark: what HBM rate do the decoder's access patterns allow on this GPU?
// build: hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// 1. grid-stride float4 copy
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 2. read-only (sum) / write-only
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* out, size_t n)
{
    float s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 1234.5f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = make_float4(1, 2, 3, 4);
}
// 3. one wave copies R rows (1 KiB each, float4 per lane) given by an index list: all loads first, then stores
template <int R>
__global__ __launch_bounds__(256) void k_rows(const float4* __restrict__ a, float4* __restrict__ b, const int* __restrict__ idx, int n_items)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int it = blockIdx.x * 4 + wave;
    if (it >= n_items) return;
    int s[R];
#pragma unroll
    for (int k = 0; k < R; k++) s[k] = idx[it * R + k];
    float4 x[R];
#pragma unroll
    for (int k = 0; k < R; k++) x[k] = a[(size_t)s[k] * 64 + lane];
#pragma unroll
    for (int k = 0; k < R; k++) { float4 v = x[k]; v.x += 1.0f; b[(size_t)s[k] * 64 + lane] = v; }
}

int main(int argc, char** argv)
{
    const size_t rows = 3774800;            // ~ E * 16 groups: 3.87 GB per array at 1 KiB rows
    const size_t n4 = rows * 64;
    float4 *a, *b; float* out; int* idx;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 1, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
    std::vector<int> h(rows);
    for (size_t i = 0; i < rows; i++) h[i] = (int)i;
    CK(hipMalloc(&idx, rows * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        float best = 1e9, tot = 0; const int reps = 5;
        for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); tot += ms; }
        printf("%-44s best %.3f ms  %.0f GB/s   avg %.0f GB/s\n", name, best, bytes / best / 1e6, bytes / (tot / reps) / 1e6);
    };
    for (int blocks : {2048, 4096, 8192, 16384})
        { char nm[64]; snprintf(nm, 64, "copy float4 grid %d", blocks); timeit(nm, 2.0 * n4 * 16, [&] { k_copy<<<blocks, 256>>>(a, b, n4); }); }
    timeit("read-only float4 grid 8192", 1.0 * n4 * 16, [&] { k_read<<<8192, 256>>>(a, out, n4); });
    timeit("write-only float4 grid 8192", 1.0 * n4 * 16, [&] { k_write<<<8192, 256>>>(b, n4); });
    // sequential rows, R per wave
    CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
    timeit("rows sequential R=4", 2.0 * n4 * 16, [&] { int items = rows / 4; k_rows<4><<<(items + 3) / 4, 256>>>(a, b, idx, items); });
    timeit("rows sequential R=18", 2.0 * (rows / 18 * 18) * 1024.0, [&] { int items = rows / 18; k_rows<18><<<(items + 3) / 4, 256>>>(a, b, idx, items); });
    // random rows within 16 groups of rows/16 (like the CN gather: each wave's rows come from one group)
    std::mt19937 rng(1);
    const size_t G = 16, per = rows / G;
    for (size_t g = 0; g < G; g++) std::shuffle(h.begin() + g * per, h.begin() + (g + 1) * per, rng);
    CK(hipMemcpy(idx, h.data(), rows * 4, hipMemcpyHostToDevice));
    timeit("rows random-in-group R=18", 2.0 * (rows / 18 * 18) * 1024.0, [&] { int items = rows / 18; k_rows<18><<<(items + 3) / 4, 256>>>(a, b, idx, items); });
    timeit("rows random-in-group R=4", 2.0 * n4 * 16, [&] { int items = rows / 4; k_rows<4><<<(items + 3) / 4, 256>>>(a, b, idx, items); });
    timeit("rows random-in-group R=9", 2.0 * (rows / 9 * 9) * 1024.0, [&] { int items = rows / 9; k_rows<9><<<(items + 3) / 4, 256>>>(a, b, idx, items); });
    return 0;
}
