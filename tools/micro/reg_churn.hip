// reg_churn.hip — a reproducer independent of the library: does a kernel that works IN PLACE on hipHostRegister'd memory
// always reach the pages behind it when the same virtual addresses are registered again and again with new pages?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/reg_churn.hip -o /tmp/reg_churn && /tmp/reg_churn [rounds] [mode] [flags]
//   mode 0: fresh mmap per round (addresses recycled)      mode 1: one mapping, registered / unregistered per round
//   mode 2: one mapping, registered once                   mode 3: fresh mmap per round, never unmapped (addresses never recycled)
//   mode 4: malloc from the HEAP per round (unaligned, the two arrays and their neighbours share pages), freed after each round
//   mode 5: like 4 with glibc's default trimming AND malloc_trim(0) after the frees: the freed heap pages go back to the kernel
//           (brk shrinks / MADV_DONTNEED), so the next round's arrays sit at the SAME virtual addresses on NEW physical pages —
//           the page-locked -> released -> page-locked-again history the library's zero-copy admission rule excludes (round 5)
//   mode 6: like 5, and a neighbour array that shares the first / last page stays allocated and is written by the host
//   flags : hipHostRegister flags (0 default, 1 portable, 2 mapped, ...)
//   [temps]   : 1 = like a numpy caller, every call builds its input in two large temporaries (calloc'ed f64 + f32 arrays,
//               mmap'ed and unmapped again by the allocator) before copying it into the window with memcpy
//   [pause_us]: host work of a random 0 .. pause_us microseconds after every unregister / before every register (a Python
//               caller has such gaps; a tight loop serialises behind whatever the driver still has to do for the old mapping)
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <unistd.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)
__global__ void k_half(const float* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = 0.5f * x[i];
}
static float* map_fresh(size_t bytes) {
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) { perror("mmap"); exit(2); }
    return static_cast<float*>(p);
}
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000, mode = argc > 2 ? atoi(argv[2]) : 0;
    const unsigned flags = argc > 3 ? (unsigned)atoi(argv[3]) : 0;
    const unsigned pause_us = argc > 4 ? (unsigned)atoi(argv[4]) : 0;
    const int temps = argc > 5 ? atoi(argv[5]) : 0;
    if (mode == 4) { mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); }
    if (mode >= 5) mallopt(M_MMAP_THRESHOLD, 1 << 30);          // heap (brk) arrays, default trim threshold
    char* neighbour = nullptr;
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned lcg = 12345;
    auto rnd = [&] { lcg = lcg * 1664525u + 1013904223u; return lcg; };
    const size_t maxn = 600000, maxb = (maxn + 64) * 4;
    float *xin = nullptr, *yout = nullptr;
    if (mode == 1 || mode == 2) { xin = map_fresh(maxb); yout = map_fresh(maxb); }
    if (mode == 2) { CK(hipHostRegister(xin, maxb, flags)); CK(hipHostRegister(yout, maxb, flags)); }
    long bad_calls = 0, calls = 0;
    std::vector<float> ref(maxn);
    for (int r = 0; r < rounds; r++) {
        const size_t n = 50000 + rnd() % 550000, bytes = (n + 64) * 4;
        if (mode == 0 || mode == 3) { xin = map_fresh(bytes); yout = map_fresh(bytes); }
        if (mode >= 4) { xin = static_cast<float*>(malloc(bytes)); yout = static_cast<float*>(malloc(bytes)); memset(xin, 0, bytes); memset(yout, 0, bytes); }
        if (mode == 6) { neighbour = static_cast<char*>(malloc(3000)); memset(neighbour, r, 3000); }
        if (mode != 2) { CK(hipHostRegister(xin, mode == 1 ? maxb : bytes, flags)); CK(hipHostRegister(yout, mode == 1 ? maxb : bytes, flags)); }
        if (r < 3 && mode >= 4) printf("round %d: x at %p, y at %p\n", r, (void*)xin, (void*)yout);
        float *dx, *dy;
        CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dx), xin, 0));
        CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dy), yout, 0));
        const int ncall = 1 + rnd() % 5;
        for (int c = 0; c < ncall; c++) {
            if (temps) {
                double* t64 = static_cast<double*>(calloc(n, 8));
                float* t32 = static_cast<float*>(calloc(n, 4));
                for (size_t i = 0; i < n; i++) { t64[i] = (double)(rnd() >> 8) * (1.0 / 8388608.0) - 1.0; t32[i] = (float)t64[i]; ref[i] = 0.5f * t32[i]; }
                memcpy(xin + 3, t32, n * 4);                              // (glibc: non-temporal stores beyond its threshold)
                for (size_t i = 0; i < n + 59; i++) yout[i] = -7.0f;
                free(t64); free(t32);
            } else
            for (size_t i = 0; i < n; i++) { xin[3 + i] = (float)(rnd() >> 8) * (1.0f / 8388608.0f) - 1.0f; ref[i] = 0.5f * xin[3 + i]; yout[5 + i] = -7.0f; }
            hipLaunchKernelGGL(k_half, dim3(1024), dim3(256), 0, s, dx + 3, dy + 5, (long)n);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            long d = 0, never = 0, first = -1;
            for (size_t i = 0; i < n; i++) if (yout[5 + i] != ref[i]) { d++; if (first < 0) first = (long)i; if (yout[5 + i] == -7.0f) never++; }
            calls++;
            if (d) { bad_calls++; if (bad_calls <= 6) printf("round %d call %d: %ld of %zu outputs differ (first %ld), %ld never written\n", r, c, d, n, first, never); }
        }
        if (mode != 2) { CK(hipHostUnregister(xin)); CK(hipHostUnregister(yout)); }
        if (mode == 0) { munmap(xin, bytes); munmap(yout, bytes); }
        if (mode >= 4) { free(xin); free(yout); }
        if (mode == 6) { neighbour[r % 3000] ^= 1; free(neighbour); }
        if (mode >= 5) malloc_trim(0);
        if (pause_us) usleep(rnd() % pause_us);
    }
    printf("pause <= %u us, mode %d flags %u: %d rounds, %ld calls, %ld with mismatches\n", pause_us, mode, flags, rounds, calls, bad_calls);
    return bad_calls ? 1 : 0;
}
