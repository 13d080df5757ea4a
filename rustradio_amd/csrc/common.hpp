// common.hpp — error plumbing and small RAII helpers shared by the host side of the library.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "fft_core.hpp"

namespace rr {
// every kernel launch of the library goes through hipLaunchKernelGGL: counted for rr_debug_kernel_launches (a test asserts
// that a clean FftFilter work() is ONE launch)
extern std::atomic<unsigned long long> g_kernel_launches;
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                                                     \
    do {                                                                                        \
        ::rr::g_kernel_launches.fetch_add(1, std::memory_order_relaxed);                        \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);                                  \
    } while (0)

namespace rr {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
// Thrown by the constructor of a FUSED block for a shape its kernels do not cover (tap count beyond the tile, ratio beyond
// the index arithmetic): the factories in compose.cpp catch exactly this and build the unfused composition instead.  Every
// other failure — a bad argument the reference rejects too, a HIP error, out of memory — stays an Error and propagates.
struct NotFusedShape : Error {
    using Error::Error;
};

void set_last_error(const std::string& m);

// Path-selection overrides of the block being constructed (include/rustradio_amd.h rr_build_opts): set by
// rr_next_create_options() for the next rr_*_create call of this thread and cleared when that call returns.
// All-zero = every choice automatic.  There are no environment switches in the product.
struct BuildOpts {
    int fir_path = 0;          // RR_PATH_AUTO / RR_PATH_DIRECT / RR_PATH_FFT
    int fir_prune = 0;         // 0 auto, > 0 on, < 0 off
    int fir_half = 0;          // 0 auto, < 0 off
    int fir_cfg = -1;          // >= 0: direct-form tile shape
    int fft_log2f = 0;         // 10..14: forced overlap-save tile
    int fft_no_split = 0;
    int fftfloat_complex = 0;
    int fm_full = 0;
    int fm_poly = 0;           // 0 auto, < 0 off: polyphase (decimate-first) tiles of the fused chains
    int dstream_no_vmm = 0;
    int fft_nonfinite_tiles = 0;   // FftFilter / FftFilterFloat: no reference-block pass for non-finite samples
    int host_in_staged = 0;    // page-locked input windows: > 0 staged through a copy kernel, < 0 read in place, 0 the block's default
    int fir_poly = 0;          // 0 auto, > 0 on wherever supported, < 0 off: decimating FirFilter<Complex> on decimate-first tiles
};
const BuildOpts& build_opts();
void set_build_opts(const BuildOpts* o);   // nullptr = reset

#define RR_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            throw ::rr::Error(std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

void stage_upload_sync(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);      // stage.cpp
void stage_download_sync(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s);

// Device buffer that only grows.
template <class T> struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) RR_HIP(hipFree(p));
        p = nullptr; cap = 0;
        RR_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
        cap = n;
    }
    // one-time setup tables: blocking, so the host vector may go out of scope right after the call
    void upload(const T* h, size_t n, hipStream_t s) {
        reserve(n);
        if (n) stage_upload_sync(p, h, n * sizeof(T), s);   // (stage.hpp: through the library's own pinned chunks, never a DMA out of `h`)
    }
};

// A stream window that may be preceded by a small handle-owned prefix (carry state):
// virtual index v < plen -> prefix[v]; else in[v - plen] (zero beyond in_len).
template <class T> struct VSrc {
    const T* prefix;
    long plen;
    const T* in;
    long in_len;
#if defined(__HIPCC__)
    __device__ __forceinline__ T load(long v) const {
        if (v < plen) return prefix[v];
        long i = v - plen;
        if (i < in_len) return in[i];
        T z{};
        return z;
    }
#endif
};

// Carry-state update folded into the main kernel of a work() call: dst[i] = src.load(v0 + i), i < n (n == 0: nothing).
// The carry only READS this call's virtual stream (prefix[cur] ++ window) and writes the OTHER prefix buffer, which no
// workgroup of the launch reads, so any workgroup may do it at any time: every thread of the grid moves at most one
// element (round 2 spent a separate k_vcopy launch per call on it: 5-8 % of the GPU time of the fused chains).
struct CarryOut {
    void* dst = nullptr;
    long v0 = 0, n = 0;
};
#if defined(__HIPCC__)
template <class T, class SRC> __device__ __forceinline__ void carry_store(const SRC& src, const CarryOut& c) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < c.n; i += (long)gridDim.x * blockDim.x)
        static_cast<T*>(c.dst)[i] = src.load(c.v0 + i);
}
#endif

#if defined(__HIPCC__)
// Window pointers as GLOBAL-address-space pointers.  The windows reach the kernels inside by-value structs (VSrc); whether
// the compiler then proves them global (global_load, vmcnt only) or leaves them generic (flat_load: also counted on
// lgkmcnt, and ordered with the LDS exchanges of the tile) turned out to depend on unrelated code of the same translation
// unit — round 3: rewriting k_fm_chain's epilogue turned the 16 tile loads of EVERY load_tile16 kernel, the headline's
// included, into flat loads and made two kernels spill.  The hot loads therefore say it themselves.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> using gptr = const T __attribute__((address_space(1)))*;
template <class T> __device__ __forceinline__ gptr<T> as_global(const T* p) { return (gptr<T>)p; }
#else           // (host pass of the same translation unit: plain pointers)
template <class T> using gptr = const T*;
template <class T> __host__ __device__ inline gptr<T> as_global(const T* p) { return p; }
#endif
#endif

#if defined(__HIPCC__)
// Exactly-rounded single operations that the optimiser cannot fuse.  The kernels are compiled with
// -ffp-contract=fast and HIP's __fmul_rn/__fadd_rn/__fsub_rn are plain operators — which it DOES
// contract into FMAs (found by the bit-exact MultiplyConst<Complex> test).  Used wherever the
// reference's un-fused f32 arithmetic has to be reproduced bit for bit.
__device__ __forceinline__ float mul_rn(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sub_rn(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#endif

// The same with the window still in the RTL-SDR wire format (u8 I/Q pairs, rtlsdr_decode.rs:35-42):
// the carried prefix is already Complex, window samples are decoded on load —
// (Float::from(b) - 127.0) * 0.008, bit-identical to the RtlSdrDecode block.
struct iq8 { unsigned char i, q; };
struct VSrcIQ8 {
    const cf* prefix;
    long plen;
    const iq8* in;       // 2-byte aligned
    long in_len;         // samples (byte pairs)
#if defined(__HIPCC__)
    static __device__ __forceinline__ float cvt(unsigned b) { return mul_rn(sub_rn((float)b, 127.0f), 0.008f); }
    static __device__ __forceinline__ cf decode(unsigned short w) {
        cf r; r.x = cvt(w & 0xffu); r.y = cvt(w >> 8); return r;
    }
    __device__ __forceinline__ cf load(long v) const {
        if (v < plen) return prefix[v];
        long i = v - plen;
        if (i < in_len) return decode(reinterpret_cast<const unsigned short*>(in)[i]);
        cf z{};
        return z;
    }
#endif
};

}  // namespace rr
