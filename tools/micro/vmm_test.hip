// vmm_test.hip — can HIP map one physical allocation at two consecutive virtual ranges (the reference's
// double-mapped ring, src/nowasm/circular_buffer.rs:98-128, in HBM)?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned* p, size_t n) { for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i; }
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu\n", gran);
    const size_t P = ((4096000 + gran - 1) / gran) * gran;
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, P, &prop, 0));
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, 2 * P, 0, nullptr, 0));
    CK(hipMemMap(va, P, 0, h, 0));
    CK(hipMemMap((char*)va + P, P, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, 2 * P, &acc, 1));
    unsigned* p = (unsigned*)va;
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, p, P / 4);
    CK(hipDeviceSynchronize());
    unsigned a[4], b[4];
    CK(hipMemcpy(a, p + 1000, 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b, p + P / 4 + 1000, 16, hipMemcpyDeviceToHost));
    printf("first mapping %u %u, second mapping %u %u -> %s\n", a[0], a[1], b[0], b[1], (a[0] == 1000 && b[0] == 1000) ? "DOUBLE MAPPING WORKS" : "mismatch");
    // a window that straddles the seam, written through the second mapping, read through the first
    hipLaunchKernelGGL(fill, dim3(1), dim3(64), 0, 0, p + P / 4 - 8, (size_t)16);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(a, p + P / 4 - 2, 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b, p, 8, hipMemcpyDeviceToHost));
    printf("seam: %u %u | %u %u   wrapped into the start: %u %u\n", a[0], a[1], a[2], a[3], b[0], b[1]);
    CK(hipMemUnmap(va, P)); CK(hipMemUnmap((char*)va + P, P)); CK(hipMemAddressFree(va, 2 * P)); CK(hipMemRelease(h));
    printf("ok\n");
    return 0;
}
