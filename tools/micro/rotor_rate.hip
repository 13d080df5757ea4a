// GPU box: how fast ONE lane walks FirFilter::translate's rotator chain (fir.rs:464-473: phase *= step, four rounded
// products and two rounded sums per step) — the three-packed-instruction step of k_rotor_replay under a few issue
// conditions.  hipcc --offload-arch=gfx950 -O3 tools/micro/rotor_rate.hip -o tools/micro/rotor_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float creg __attribute__((ext_vector_type(2)));
__device__ __forceinline__ creg step3(creg z, creg st) {
    creg p, q, r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p) : "v"(z), "s"(st));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(z), "s"(st));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));
    return r;
}
__device__ __forceinline__ creg step6(creg z, creg st) {          // scalar form: 4 multiplies, 1 subtract, 1 add
    float a, b, c, d, x, y;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(z.x), "s"(st.x));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(z.y), "s"(st.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(c) : "v"(z.x), "s"(st.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(z.y), "s"(st.x));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x) : "v"(a), "v"(b));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(y) : "v"(c), "v"(d));
    creg r; r.x = x; r.y = y; return r;
}
// the three instructions of a step as ONE asm statement: the compiler puts no s_nop between them (it guards every
// separately written packed instruction that reads the previous one's result with a wait state)
__device__ __forceinline__ creg step3_fused(creg z, creg st) {
    creg r, p, q;
    asm volatile("v_pk_mul_f32 %1, %3, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
                 "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
                 "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]"
                 : "=v"(r), "=&v"(p), "=&v"(q) : "v"(z), "s"(st));
    return r;
}
template <int MODE> __global__ void k(float2* out, float sx, float sy, long n) {
    if (MODE == 1) __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x != 0 && MODE != 3) return;                    // MODE 3: all 64 lanes walk (the same chain)
    creg st; st.x = sx; st.y = sy;
    creg z; z.x = 1.0f; z.y = 0.0f;
    for (long i = 0; i < n; i += 16) {
#pragma unroll
        for (int k2 = 0; k2 < 16; k2++) z = MODE == 2 ? step6(z, st) : MODE == 4 ? step3_fused(z, st) : step3(z, st);
    }
    if (threadIdx.x == 0) out[blockIdx.x] = make_float2(z.x, z.y);
}
template <int MODE> static void run(const char* name, int blocks) {
    float2* d; CK(hipMalloc(&d, 64 * sizeof(float2)));
    const long n = 4000000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0.99f, 0.1f, 1000L);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 0.99f, 0.1f, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    float2 h; CK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("%-52s %6.2f ns per step   (end phase %.9g %.9g)\n", name, ms * 1e6 / n, h.x, h.y);
    CK(hipFree(d));
}
int main() {
    run<0>("3 packed instructions, one lane, one wave", 1);
    run<1>("... with s_setprio 3", 1);
    run<2>("6 scalar instructions", 1);
    run<3>("3 packed, all 64 lanes active", 1);
    run<0>("3 packed, 8 such waves on 8 CUs (independent chains)", 8);
    run<4>("3 packed as ONE asm statement (no s_nop in between)", 1);
    return 0;
}
