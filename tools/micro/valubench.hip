// valubench.hip — VALU issue rate on gfx950 per SIMD: v_fma_f32 vs v_pk_fma_f32 (VGPR operands, SGPR
// operand with op_sel broadcast) vs v_pk_mul/add.  W waves per SIMD (workgroup = 256*W threads, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 64
template <int MODE>
__global__ void k(float* out, int iters, float sv) {
    f2 acc[8];
    for (int j = 0; j < 8; j++) acc[j] = f2{(float)threadIdx.x + j, 1.0f};
    f2 x = {1.0001f, 0.9999f};
    f2 s2 = {sv, sv * 0.5f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j].x) : "v"(x.x), "v"(x.y));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x), "v"(x));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "s"(s2), "v"(x));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc[j]) : "s"(s2), "v"(x));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(x));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[j]) : "v"(x));
                if (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[j].x) : "s"(sv), "v"(x.y));
                if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[j]) : "v"(x), "v"(x));
            }
        }
    }
    float t = 0;
    for (int j = 0; j < 8; j++) t += acc[j].x + acc[j].y;
    if (t == 123456.f) out[0] = t;
}
template <int MODE> void run(const char* name, int W) {
    const int iters = 4000, grid = 256;
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256 * W), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256 * W), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * REP * W;
    printf("%-28s W=%d  %6.2f ns/wave-instr/SIMD (~%.2f clk @2.4GHz)\n", name, W, ms * 1e6 / insts_per_simd,
           ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    for (int W = 1; W <= 4; W *= 2) {
        run<0>("v_fma_f32 vgpr", W); run<6>("v_fmac_f32 sgpr", W); run<1>("v_pk_fma_f32 vgpr", W);
        run<7>("v_pk_fma_f32 vgpr op_sel_hi", W);
        run<2>("v_pk_fma_f32 sgpr bcast lo", W); run<3>("v_pk_fma_f32 sgpr bcast hi", W);
        run<4>("v_pk_mul_f32", W); run<5>("v_pk_add_f32", W);
    }
    return 0;
}
