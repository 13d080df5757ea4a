// l1bench.hip — what one CU's vector-memory path delivers from an L2-resident table, by access width and by the number
// of waves asking (the response stream of k_fm_multi_poly: every wave reads 48 KB per channel and tile, 512 B or 1 KB per
// wave-instruction, each line once).  One workgroup per CU (the LDS request forces that), W waves each streaming its own
// region of a 1.5 MB table (or all waves the SAME region: L1 hits) with `depth` loads in flight.
// hipcc --offload-arch=gfx950 -O3 tools/micro/l1bench.hip -o tools/micro/l1bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class V, int DEPTH>
__global__ __launch_bounds__(512) void stream(const V* __restrict__ tab, long tab_elems, int reps, int same, float* sink, long long* clocks) {
    extern __shared__ unsigned char smem[];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    // wave w of workgroup b reads the slice [start, start + per) of the table, DEPTH loads of 64 lanes each per step
    const long per = tab_elems / 8;
    const long start = same ? 0 : ((w + blockIdx.x) % 8) * per;
    (void)nw;
    V acc = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; r++) {
        for (long i = 0; i + 64 * DEPTH <= per; i += 64 * DEPTH) {
            V v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) v[d] = tab[start + i + d * 64 + l];
#pragma unroll
            for (int d = 0; d < DEPTH; d++) acc += v[d];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (acc.x == 123.456f) sink[0] = acc.x;
    if (threadIdx.x == 0 && smem[0] == 77) sink[1] = 1;
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

template <class V, int DEPTH> int run(const void* tab, size_t bytes, int waves, int same, float* sink, long long* clocks) {
    const int grid = 256, reps = 40;
    const long elems = bytes / sizeof(V);
    CK(hipFuncSetAttribute((const void*)stream<V, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((stream<V, DEPTH>), dim3(grid), dim3(64 * waves), 100 * 1024, 0, (const V*)tab, elems, 2, same, sink, clocks);
    hipEventRecord(a);
    hipLaunchKernelGGL((stream<V, DEPTH>), dim3(grid), dim3(64 * waves), 100 * 1024, 0, (const V*)tab, elems, reps, same, sink, clocks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)(elems / 8 / (64 * DEPTH) * (64 * DEPTH)) * sizeof(V) * reps * waves;
    printf("  %2zu B/lane depth %2d waves %d %s: %7.1f B/ns/CU  (%6.1f TB/s chip)\n", sizeof(V), DEPTH, waves, same ? "same slice " : "own slices ",
           bytes_per_cu / (ms * 1e6), bytes_per_cu * grid / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    const size_t bytes = 1536 * 1024;
    void* tab; float* sink; long long* clocks;
    CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clocks, 256 * 8));
    printf("L2-resident 1.5 MB table, 256 workgroups (one per CU), per-CU delivered rate (divide by the clock in GHz for B/clk)\n");
    for (int same = 0; same < 2; same++)
        for (int waves : {1, 2, 4, 8}) {
            run<float2, 16>(tab, bytes, waves, same, sink, clocks);
            run<float2, 32>(tab, bytes, waves, same, sink, clocks);
            run<float4, 8>(tab, bytes, waves, same, sink, clocks);
            run<float4, 16>(tab, bytes, waves, same, sink, clocks);
        }
    return 0;
}
