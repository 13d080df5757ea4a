// tile_common.hpp — device helpers shared by the tile kernels (kernels_fft.hip, kernels_poly.hip).
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "kernels.hpp"

namespace rr {

unsigned long long* fft_stamp_buffer();      // measurement builds only (kernels_fft.hip), nullptr otherwise

// Exchange synchronisation.  __syncthreads() also drains vmcnt (it is a fence), which would
// serialise outstanding global traffic behind every LDS exchange.  A one-wave workgroup needs
// no barrier at all (a wave's LDS operations execute in order); larger tiles wait for their
// own LDS writes only and then meet at a raw s_barrier.
template <int T> __device__ __forceinline__ void tile_sync() {
    if constexpr (T > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("" ::: "memory");
}

// Workgroup b -> tile sequence.  The dispatcher places workgroup b on XCD b % 8 (observed,
// used for speed only): each XCD gets one contiguous eighth of the tiles and its
// workgroups sweep it together, so the L-1 samples two neighbouring tiles share are
// fetched from HBM once and re-read from that XCD's L2.
struct TileIter {
    long tile, end, step;
    __device__ __forceinline__ TileIter(long ntiles) {
        const int b = blockIdx.x, g = gridDim.x;
        const int nx = g < 8 ? g : 8;                  // partitions (XCDs that have a workgroup)
        const int xcd = b % nx, slot = b / nx;
        const int gx = (g - xcd + nx - 1) / nx;        // workgroups in this partition
        const long lo = ntiles * xcd / nx, hi = ntiles * (xcd + 1) / nx;
        tile = lo + slot; end = hi; step = gx;
    }
};

// fast-math 0.1.1 atan2 restated from its published algorithm (crate not vendored: parity-unpinned
// flavour); same code as k_quaddemod in kernels_misc.hip.
__device__ __forceinline__ float fmc_flip_sign(float v, float s) {
    return __uint_as_float(__float_as_uint(v) ^ (__float_as_uint(s) & 0x80000000u));
}
__device__ __forceinline__ float fmc_atan_raw(float x) {
    return mul_rn(sub_rn(add_rn(0.78539816339744830962f, 0.273f), mul_rn(0.273f, fabsf(x))), x);
}
__device__ __forceinline__ float fmc_atan2(float y, float x) {
    if (fabsf(y) < fabsf(x)) {
        const float bias = x > 0.0f ? 0.0f : 3.14159265358979323846f;
        return add_rn(fmc_flip_sign(bias, y), fmc_atan_raw(__fdiv_rn(y, x)));
    } else if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return fmc_flip_sign(1.57079632679489661923f, y);
    }
    return sub_rn(fmc_flip_sign(1.57079632679489661923f, y), fmc_atan_raw(__fdiv_rn(x, y)));
}

// f32::atan2 (quadrature_demod.rs:106-108) for the fused epilogues: |y| / |x| folded into [0, 1] with one v_rcp_f32
// (1 ulp), an odd polynomial atan(t) = t P(t^2), P(0) = 1 (degree 7 in t^2, minimax fit with the constant term pinned so
// that tiny angles stay exact to rounding; 1.2e-7 rad worst case evaluated in f32 — round 5: one term fewer than the
// degree-8 fit it replaces, 9e-8, both at the level of f32 rounding), then the octant fix-ups.  Worst case 3e-7 rad against
// libm (test_quaddemod_exact_atan2_accuracy asks for 1e-6);
// atan2(+-0, +x) = +-0 and atan2(+-0, -x) = +-pi exactly (quad_nulls), inf / inf = pi/4 multiples, NaN propagates.
// ~23 VALU instructions; the library atan2f is 2-3x that, and the epilogue is the largest part of a decimated chain's work.
__device__ __forceinline__ float atan_poly01(float t) {
    const float s = t * t;
    float p = -0.004355404991656542f;
    p = fmaf(p, s, 0.023040134459733963f);
    p = fmaf(p, s, -0.05777358636260033f);
    p = fmaf(p, s, 0.09794234484434128f);
    p = fmaf(p, s, -0.13976581394672394f);
    p = fmaf(p, s, 0.19962704181671143f);
    p = fmaf(p, s, -0.3333165943622589f);
    p = fmaf(p, s, 1.0f);
    return t * p;
}
// TAME = the caller has established that x and y are finite (see tile_tame in kernels_poly.hip): the inf / inf and NaN
// fix-ups — two compares and two selects per value — are left out; everything else is the same instruction for instruction.
template <bool TAME = false>
__device__ __forceinline__ float atan2_poly(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1.17549435e-38f);
#if defined(__HIP_DEVICE_COMPILE__)
    // min(|x|, |y|) as ONE instruction: the source modifiers take the absolute values.  (fminf(fabsf, fabsf) compiles to two
    // v_max_f32 |a|, |a| canonicalisations + the minimum: 4 of the 79 instructions of a pair of demodulated samples.)
    float mn;
    asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
#else
    const float mn = fminf(ax, ay);
#endif
    float t = mn * __builtin_amdgcn_rcpf(mx);
    if constexpr (!TAME) { if (mn == __builtin_inff()) t = 1.0f; }   // inf / inf
    float r = atan_poly01(t);
    if (ay > ax) r = 1.57079632679489661923f - r;
    if (__float_as_uint(x) >> 31) r = 3.14159265358979323846f - r;
    if constexpr (!TAME) { if (__builtin_isunordered(x, y)) r = __builtin_nanf(""); }   // (fmaxf / fminf drop a NaN operand)
    return __uint_as_float(__float_as_uint(r) | (__float_as_uint(y) & 0x80000000u));
}

// gain * atan2(conj(rl) * ru) in num-complex order, un-contracted (quadrature_demod.rs:72-109)
template <bool POLY> __device__ __forceinline__ float demod_pair(creg rl, creg ru, float gain, int mode) {
    const float na = -rl.y;
    const float re = sub_rn(mul_rn(rl.x, ru.x), mul_rn(na, ru.y));
    const float im = add_rn(mul_rn(rl.x, ru.y), mul_rn(na, ru.x));
    const float ang = mode == 0 ? (POLY ? atan2_poly(im, re) : atan2f(im, re)) : fmc_atan2(im, re);
    return mul_rn(gain, ang);
}

template <class KFn> static long grid_for_tiles(KFn kfn, int T, size_t smem, long ntiles) {
    // launch setup (shared-memory attribute, occupancy) is per kernel AND per device: a process may drive several
    // GPUs (rr_set_device); first launches may come from several host threads at once
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> per_cu_of;
    int dev = 0;
    RR_HIP(hipGetDevice(&dev));
    const std::pair<const void*, int> key(reinterpret_cast<const void*>(kfn), dev);
    int per_cu;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = per_cu_of.find(key);
        if (it == per_cu_of.end()) {
            RR_HIP(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            int n = 0;
            RR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kfn, T, smem));
            if (n < 1) n = 1;
#ifdef RR_MEASURE_KNOBS
            if (const char* e = getenv("RR_FFT_PERCU")) n = atoi(e) > 0 ? atoi(e) : n;   // measurement builds only (tools/fft_percu.sh)
#endif
            it = per_cu_of.emplace(key, n).first;
        }
        per_cu = it->second;
    }
    long grid = (long)device_cu_count() * per_cu;
    return grid > ntiles ? ntiles : grid;
}

}  // namespace rr
