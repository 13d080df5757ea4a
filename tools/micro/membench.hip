// membench.hip — calibrates achievable HBM copy bandwidth for the access shapes the
// FftFilter kernel uses (8 B/lane vs 16 B/lane, 1-wave workgroups, 8 waves/CU).
// hipcc --offload-arch=gfx950 -O3 tools/micro/membench.hip -o /tmp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class V, int UNROLL>
__global__ void copy_tiles(const V* __restrict__ in, V* __restrict__ out, long ntiles, int tile_elems) {
    // one wave copies one contiguous tile of tile_elems V's per iteration (UNROLL loads in flight)
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const V* p = in + tile * tile_elems + threadIdx.x;
        V* q = out + tile * tile_elems + threadIdx.x;
        V r[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; i++) r[i] = p[i * blockDim.x];
#pragma unroll
        for (int i = 0; i < UNROLL; i++) q[i * blockDim.x] = r[i];
    }
}

template <class V, int UNROLL>
double run(const void* in, void* out, size_t bytes, int threads, int grid, int iters) {
    const int tile_elems = threads * UNROLL;
    const long ntiles = bytes / (sizeof(V) * tile_elems);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((copy_tiles<V, UNROLL>), dim3(grid), dim3(threads), 0, 0, (const V*)in, (V*)out, ntiles, tile_elems);
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((copy_tiles<V, UNROLL>), dim3(grid), dim3(threads), 0, 0, (const V*)in, (V*)out, ntiles, tile_elems);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 2.0 * ntiles * tile_elems * sizeof(V) * iters / (ms * 1e-3) / 1e12;
}

int main() {
    const size_t bytes = 800ull << 20;
    void *in, *out;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes));
    CK(hipMemset(in, 1, bytes));
    printf("copy TB/s (read+write), 800 MiB each way\n");
    for (int grid : {2048, 4096, 8192, 65536}) {
        printf("grid %6d x 64 thr : float2 x16 %.2f | float4 x8 %.2f | float4 x16 %.2f\n", grid,
               run<float2, 16>(in, out, bytes, 64, grid, 10), run<float4, 8>(in, out, bytes, 64, grid, 10),
               run<float4, 16>(in, out, bytes, 64, grid, 10));
    }
    for (int grid : {512, 1024, 2048, 16384}) {
        printf("grid %6d x 256 thr: float2 x16 %.2f | float4 x8 %.2f | float4 x4 %.2f\n", grid,
               run<float2, 16>(in, out, bytes, 256, grid, 10), run<float4, 8>(in, out, bytes, 256, grid, 10),
               run<float4, 4>(in, out, bytes, 256, grid, 10));
    }
    return 0;
}
