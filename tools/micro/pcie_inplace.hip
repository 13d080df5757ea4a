// GPU box: what a KERNEL gets out of page-locked host memory over PCIe (the drop-in path's in-place windows):
// read-only (host -> HBM), write-only (HBM -> host) and both at once, 4,096,000-byte windows, by load width and grid.
// hipcc --offload-arch=gfx950 -O3 tools/micro/pcie_inplace.hip -o tools/micro/pcie_inplace.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <class V> __global__ void k_copy(const V* __restrict__ in, V* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
template <class V> static float run(const void* in, void* out, long bytes, int grid, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const long n = bytes / sizeof(V);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_copy<V>, dim3(grid), dim3(256), 0, 0, (const V*)in, (V*)out, n);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_copy<V>, dim3(grid), dim3(256), 0, 0, (const V*)in, (V*)out, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}
int main() {
    const long B = 4096000;
    void *h1 = mmap(nullptr, B, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0), *h2 = mmap(nullptr, B, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(h1, 1, B); memset(h2, 2, B);                 // (touched before they are page-locked, as a ring in use is)
    CK(hipHostRegister(h1, B, hipHostRegisterDefault)); CK(hipHostRegister(h2, B, hipHostRegisterDefault));
    void *d1, *d2, *g1, *g2; CK(hipHostGetDevicePointer(&d1, h1, 0)); CK(hipHostGetDevicePointer(&d2, h2, 0));
    CK(hipMalloc(&g1, B)); CK(hipMalloc(&g2, B));
    printf("%-28s %8s %8s %8s %8s   us per 4,096,000-byte window (GB/s)\n", "direction / width", "g=256", "g=1024", "g=4096", "g=16000");
    const int grids[4] = {256, 1024, 4096, 16000};
    struct { const char* name; const void* in; void* out; } dirs[3] = {{"read host -> HBM", d1, g2}, {"write HBM -> host", g1, d2}, {"host -> host (both ways)", d1, d2}};
    for (auto& d : dirs) {
        for (int w = 0; w < 3; w++) {
            printf("%-22s %2d B ", d.name, w == 0 ? 4 : w == 1 ? 8 : 16);
            for (int g : grids) {
                const float us = w == 0 ? run<float>(d.in, d.out, B, g, 50) : w == 1 ? run<float2>(d.in, d.out, B, g, 50) : run<float4>(d.in, d.out, B, g, 50);
                printf(" %6.1f(%4.1f)", us, B / us * 1e-3);
            }
            printf("\n");
        }
    }
    // DMA for comparison
    hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int dir = 0; dir < 2; dir++) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 50; i++) CK(dir ? hipMemcpyAsync(h2, g1, B, hipMemcpyDeviceToHost, s) : hipMemcpyAsync(g2, h1, B, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("hipMemcpyAsync %s: %.1f us (%.1f GB/s)\n", dir ? "D2H" : "H2D", ms / 50 * 1e3, B / (ms / 50 * 1e3) * 1e-3);
    }
    // a kernel READING host memory on one stream while the DMA engine downloads on another: does the pair beat the 64 GB/s
    // that one kernel gets reading and writing host memory at once?
    {
        hipStream_t s2; CK(hipStreamCreate(&s2));
        hipEvent_t a2, b2, c2; CK(hipEventCreate(&a2)); CK(hipEventCreate(&b2)); CK(hipEventCreate(&c2));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a2, s));
        CK(hipStreamWaitEvent(s2, a2, 0));
        for (int i = 0; i < 50; i++) {
            hipLaunchKernelGGL(k_copy<float2>, dim3(1024), dim3(256), 0, s, (const float2*)d1, (float2*)g2, B / 8);
            CK(hipMemcpyAsync(h2, g1, B, hipMemcpyDeviceToHost, s2));
        }
        CK(hipEventRecord(b2, s)); CK(hipEventRecord(c2, s2));
        CK(hipEventSynchronize(b2)); CK(hipEventSynchronize(c2));
        float m1, m2; CK(hipEventElapsedTime(&m1, a2, b2)); CK(hipEventElapsedTime(&m2, a2, c2));
        const float ms = m1 > m2 ? m1 : m2;
        printf("kernel read host + DMA download, concurrently: %.1f us per window pair (%.1f GB/s each way)\n", ms / 50 * 1e3, B / (ms / 50 * 1e3) * 1e-3);
    }
    return 0;
}
