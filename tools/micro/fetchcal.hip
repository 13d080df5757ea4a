// fetchcal.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, for the access
// shapes the tile kernels use.  /opt/skills/guides/MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of
// a wide coalesced streaming read (16 B/lane) ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
// byte count in your own access pattern" (k_fftfilt_os reads 8 B/lane, the decimate-first
// kernels 8 / 16 B at a D-sample lane stride, the real-stream kernels 4 B/lane, the RTL-SDR ones 2 B/lane).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fetchcal.hip -o tools/micro/fetchcal.bin
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o p -- tools/micro/fetchcal.bin
//   rocprofv3 --pmc WRITE_SIZE ...                     (separate pass)      -> tools/fetchcal_summary.py <dir>...
//
// Every kernel touches exactly `bytes` (printed per kernel, and encoded in the kernel name's order) of a 1 GiB buffer —
// four times the 256 MiB Infinity Cache — once, with the tile-strided lane-consecutive pattern of load_tile16
// (kernels_fft.hip): thread t of a 64- or 128-thread workgroup reads element n * T + t of its tile, n < 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class V> __device__ __forceinline__ float fold(V v);
template <> __device__ __forceinline__ float fold(float v) { return v; }
template <> __device__ __forceinline__ float fold(float2 v) { return v.x + v.y; }
template <> __device__ __forceinline__ float fold(float4 v) { return v.x + v.y + v.z + v.w; }
template <> __device__ __forceinline__ float fold(unsigned short v) { return (float)v; }

// read-only: 16 lane-consecutive loads per thread and tile, the sum goes out once per thread (negligible write traffic)
template <class V, int T>
__global__ __launch_bounds__(T) void rd_tiles(const V* __restrict__ in, float* __restrict__ sink, long ntiles) {
    float acc = 0.0f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const V* p = in + tile * (16 * T) + threadIdx.x;
        V r[16];
#pragma unroll
        for (int n = 0; n < 16; n++) r[n] = p[n * T];
#pragma unroll
        for (int n = 0; n < 16; n++) acc += fold(r[n]);
    }
    if (acc == 1234.5f) sink[threadIdx.x] = acc;
}
// the decimate-first kernels' pattern: lane t reads the 8-byte sample at D * (64 n + t) - p of its tile for phase p < D
// (every phase touches every cache line of the tile; the tile is read once in total)
template <int D>
__global__ __launch_bounds__(64) void rd_phases(const float2* __restrict__ in, float* __restrict__ sink, long ntiles) {
    float acc = 0.0f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float2* base = in + tile * (1024L * D) + (D - 1) + (long)D * threadIdx.x;
        for (int p = 0; p < D; p++) {
            float2 r[16];
#pragma unroll
            for (int n = 0; n < 16; n++) r[n] = base[(long)D * 64 * n - p];
#pragma unroll
            for (int n = 0; n < 16; n++) acc += r[n].x + r[n].y;
        }
    }
    if (acc == 1234.5f) sink[threadIdx.x] = acc;
}
// write-only: 16 lane-consecutive stores per thread and tile
template <class V, int T>
__global__ __launch_bounds__(T) void wr_tiles(V* __restrict__ out, long ntiles, V val) {
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        V* p = out + tile * (16 * T) + threadIdx.x;
#pragma unroll
        for (int n = 0; n < 16; n++) p[n * T] = val;
    }
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 8;
    const int reps = 3;
    printf("every kernel below moves %zu bytes once (launched %d times)\n", bytes, reps);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL((rd_tiles<float4, 64>), dim3(grid), dim3(64), 0, 0, (const float4*)buf, sink, (long)(bytes / (16 * 64 * 16)));
        hipLaunchKernelGGL((rd_tiles<float2, 64>), dim3(grid), dim3(64), 0, 0, (const float2*)buf, sink, (long)(bytes / (8 * 64 * 16)));
        hipLaunchKernelGGL((rd_tiles<float2, 128>), dim3(grid), dim3(128), 0, 0, (const float2*)buf, sink, (long)(bytes / (8 * 128 * 16)));
        hipLaunchKernelGGL((rd_tiles<float, 128>), dim3(grid), dim3(128), 0, 0, (const float*)buf, sink, (long)(bytes / (4 * 128 * 16)));
        hipLaunchKernelGGL((rd_tiles<unsigned short, 128>), dim3(grid), dim3(128), 0, 0, (const unsigned short*)buf, sink, (long)(bytes / (2 * 128 * 16)));
        hipLaunchKernelGGL((rd_phases<6>), dim3(grid), dim3(64), 0, 0, (const float2*)buf, sink, (long)(bytes / (8 * 1024 * 6)) - 1);
        hipLaunchKernelGGL((rd_phases<4>), dim3(grid), dim3(64), 0, 0, (const float2*)buf, sink, (long)(bytes / (8 * 1024 * 4)) - 1);
        hipLaunchKernelGGL((wr_tiles<float4, 64>), dim3(grid), dim3(64), 0, 0, (float4*)buf, (long)(bytes / (16 * 64 * 16)), make_float4(1, 2, 3, 4));
        hipLaunchKernelGGL((wr_tiles<float2, 128>), dim3(grid), dim3(128), 0, 0, (float2*)buf, (long)(bytes / (8 * 128 * 16)), make_float2(1, 2));
        hipLaunchKernelGGL((wr_tiles<float, 128>), dim3(grid), dim3(128), 0, 0, (float*)buf, (long)(bytes / (4 * 128 * 16)), 1.0f);
        CK(hipDeviceSynchronize());
    }
    // known bytes per kernel, for tools/fetchcal_summary.py
    printf("KNOWN rd_tiles<float4,64> %zu\nKNOWN rd_tiles<float2,64> %zu\nKNOWN rd_tiles<float2,128> %zu\nKNOWN rd_tiles<float,128> %zu\n"
           "KNOWN rd_tiles<unsigned short,128> %zu\nKNOWN rd_phases<6> %zu\nKNOWN rd_phases<4> %zu\n"
           "KNOWN wr_tiles<float4,64> %zu\nKNOWN wr_tiles<float2,128> %zu\nKNOWN wr_tiles<float,128> %zu\n",
           bytes, bytes, bytes, bytes, bytes, (bytes / (8 * 1024 * 6) - 1) * 8 * 1024 * 6, (bytes / (8 * 1024 * 4) - 1) * 8 * 1024 * 4,
           bytes, bytes, bytes);
    return 0;
}
