// readbench.hip — read-only streaming of a 192 MB window in 48 KB tiles, the shapes of the decimate-first chain kernel's
// input: (0) lane-consecutive 16-byte chunks, (1) every third 16-byte chunk per wave (48-byte lane stride: wave w of a
// 3-wave workgroup takes the chunks j = w mod 3, 24 cache lines per wave-instruction), WG workgroups resident per CU,
// 16 loads in flight per wave.  How many TB/s does the memory system deliver for each shape, with nothing else going on?
// hipcc --offload-arch=gfx950 -O3 tools/micro/readbench.hip -o tools/micro/readbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(192) void rd(const f4* __restrict__ in, long ntiles, float* sink) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    f4 acc = {0, 0, 0, 0};
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const f4* base = in + tile * 3072;                 // 48 KB = 3072 chunks of 16 B
        f4 v[16];
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = MODE == 0 ? base[(3 * n + w) * 64 + l] : base[(n * 64 + l) * 3 + w];
#pragma unroll
        for (int n = 0; n < 16; n++) acc += v[n];
    }
    if (acc.x == 123.456f) sink[0] = acc.x;
}
template <int MODE> void run(const f4* in, long ntiles, int per_cu, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * per_cu;
    hipLaunchKernelGGL(rd<MODE>, dim3(grid), dim3(192), 0, 0, in, ntiles, sink);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(rd<MODE>, dim3(grid), dim3(192), 0, 0, in, ntiles, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  %s, %d workgroups of 3 waves per CU: %6.2f TB/s  (%.1f us per 192 MB)\n", MODE ? "48-byte lane stride " : "lane-consecutive    ", per_cu,
           ntiles * 49152.0 * 20 / (ms * 1e-3) / 1e12, ms / 20 * 1e3);
}
int main() {
    const long ntiles = 4096;
    f4* in; float* sink;
    hipMalloc(&in, ntiles * 49152); hipMemset(in, 0, ntiles * 49152); hipMalloc(&sink, 64);
    for (int per_cu : {1, 2, 4, 8}) { run<0>(in, ntiles, per_cu, sink); run<1>(in, ntiles, per_cu, sink); }
    return 0;
}
