// bankbench.hip — does the VGPR bank of the 64-bit operands matter for v_pk_fma_f32 issue rate on gfx950?
// acc pairs are hard-wired to v[100..115]; the multiplicand pair to a chosen register.  4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(W) \
    "v_pk_fma_f32 v[100:101], s[4:5], " W ", v[100:101] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[102:103], s[4:5], " W ", v[102:103] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[104:105], s[4:5], " W ", v[104:105] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[106:107], s[4:5], " W ", v[106:107] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[108:109], s[4:5], " W ", v[108:109] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[110:111], s[4:5], " W ", v[110:111] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[112:113], s[4:5], " W ", v[112:113] op_sel_hi:[0,1,1]\n" \
    "v_pk_fma_f32 v[114:115], s[4:5], " W ", v[114:115] op_sel_hi:[0,1,1]\n"
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
             "v120","v121","v122","v123","v124","v125","v126","v127","s4","s5"
template <int MODE> __global__ void k(float* out, int iters) {
    asm volatile("v_mov_b32 v120, 1.0\nv_mov_b32 v121, 1.0\nv_mov_b32 v122, 1.0\nv_mov_b32 v123, 1.0\n"
                 "v_mov_b32 v124, 1.0\nv_mov_b32 v125, 1.0\nv_mov_b32 v126, 1.0\nv_mov_b32 v127, 1.0\ns_mov_b32 s4, 0\ns_mov_b32 s5, 0" ::: CLOB);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(REP8("v[120:121]") REP8("v[120:121]") ::: CLOB);          // same banks as acc 100,104,108,112
        if (MODE == 1) asm volatile(REP8("v[122:123]") REP8("v[122:123]") ::: CLOB);          // same banks as acc 102,106,...
        if (MODE == 2) asm volatile(REP8("v[120:121]") REP8("v[122:123]") ::: CLOB);
        // rotating window like the FIR loop: 8 different multiplicand pairs
        if (MODE == 3) asm volatile(
            "v_pk_fma_f32 v[100:101], s[4:5], v[120:121], v[100:101] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[102:103], s[4:5], v[122:123], v[102:103] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[104:105], s[4:5], v[124:125], v[104:105] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[106:107], s[4:5], v[126:127], v[106:107] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[108:109], s[4:5], v[120:121], v[108:109] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[110:111], s[4:5], v[122:123], v[110:111] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[112:113], s[4:5], v[124:125], v[112:113] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[114:115], s[4:5], v[126:127], v[114:115] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[100:101], s[4:5], v[122:123], v[100:101] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[102:103], s[4:5], v[124:125], v[102:103] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[104:105], s[4:5], v[126:127], v[104:105] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[106:107], s[4:5], v[120:121], v[106:107] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[108:109], s[4:5], v[122:123], v[108:109] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[110:111], s[4:5], v[124:125], v[110:111] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[112:113], s[4:5], v[126:127], v[112:113] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[114:115], s[4:5], v[120:121], v[114:115] op_sel_hi:[0,1,1]\n" ::: CLOB);
    }
    float r; asm volatile("v_mov_b32 %0, v100" : "=v"(r));
    if (r == 123456.f) out[0] = r;
}
template <int MODE> void run(const char* name, int W, int iters) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * W), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * W), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s W=%d iters=%-7d %6.2f ns/pk_fma/SIMD  (total %.2f ms)\n", name, W, iters, ms * 1e6 / ((double)iters * 16 * W), ms);
}
int main() {
    for (int iters : {2000, 20000, 200000}) {
        run<0>("multiplicand banks {0,1}", 4, iters);
        run<1>("multiplicand banks {2,3}", 4, iters);
        run<2>("alternating", 4, iters);
        run<3>("rotating window (FIR-like)", 4, iters);
    }
    return 0;
}
