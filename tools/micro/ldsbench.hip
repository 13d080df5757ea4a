// ldsbench.hip — LDS instruction throughput on gfx950 (per CU): ds_read_b64 vs ds_read2_b64 vs
// ds_read_b128, ds_write_b64 vs ds_write2_b64 vs ds_write_b128.  One workgroup of 512 threads
// per CU (2 waves/SIMD), unit-stride (conflict-free) addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP 64
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane_addr8 = threadIdx.x * 8;      // 8 B per lane
    const unsigned lane_addr16 = threadIdx.x * 16;    // 16 B per lane
    f2 a = {(float)threadIdx.x, 1.0f}, b = a;
    f4 q = {1, 2, 3, 4};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (MODE == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(lane_addr8));
            if (MODE == 1) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(q) : "v"(lane_addr8));
            if (MODE == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(lane_addr16));
            if (MODE == 3) asm volatile("ds_write_b64 %0, %1" :: "v"(lane_addr8), "v"(a));
            if (MODE == 4) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:64" :: "v"(lane_addr8), "v"(a), "v"(b));
            if (MODE == 5) asm volatile("ds_write_b128 %0, %1" :: "v"(lane_addr16), "v"(q));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (a.x == 123456.f || q.x == 654321.f) out[0] = a.x + q.x;
}
template <int MODE> void run(const char* name, int bytes_per_lane) {
    const int iters = 2000, grid = 256;
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 32768, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 32768, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_cu = (double)iters * REP * 8;                 // wave-instructions per CU
    const double bytes = insts_per_cu * 64 * bytes_per_lane * grid;
    printf("%-14s %7.2f ns/wave-instr/CU   %6.1f TB/s chip  (~%.1f clk @2.1GHz per wave-instr)\n", name,
           ms * 1e6 / insts_per_cu, bytes / (ms * 1e-3) / 1e12, ms * 1e6 / insts_per_cu * 2.1);
}
int main() {
    run<0>("ds_read_b64", 8); run<1>("ds_read2_b64", 16); run<2>("ds_read_b128", 16);
    run<3>("ds_write_b64", 8); run<4>("ds_write2_b64", 16); run<5>("ds_write_b128", 16);
    return 0;
}
