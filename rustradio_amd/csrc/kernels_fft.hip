// kernels_fft.hip — FftFilter on gfx950: overlap-save tiles, one workgroup of F/16
// threads per F-point tile, 16 complex values per thread in VGPRs, in-place
// digit-reversed DIF forward / DIT inverse (fft_core.hpp), LDS only for the radix
// regrouping between passes.  Replaces RustFftEngine::run + the overlap-add loop of
// FftFilter::work (/root/reference/src/fft_filter.rs:172-176, 290-354); results are
// the same linear convolution, computed tile-independently.
#include <cstdlib>

#include "kernels.hpp"

// Phase-ablation switches for measurement builds (-DRR_FFT_ABLATE_BUILD): RR_FFT_ABLATE=bits
// 1: no input loads, 2: no output stores, 4: no LDS exchanges, 8: no butterflies,
// 16: inputs re-read from an L2-resident window, 32: outputs written to an L2-resident window.
#ifdef RR_FFT_ABLATE_BUILD
#define RR_ABLATE(bit) (ablate & (bit))
#else
#define RR_ABLATE(bit) false
#endif
// keep the scheduler from overlapping the live ranges of neighbouring phases
#define RR_PHASE() __builtin_amdgcn_sched_barrier(0)

namespace rr {

// Register policy (VAR).  Every tile uses the same per-thread twiddles and H values.
//   VAR 0 (F <= 4096, <= 256 threads): the 30 twiddles of the two twiddled passes and the
//          thread's 16 H values stay in VGPRs for the whole kernel; 2 waves/SIMD.
//   VAR 3 (F >= 8192, 512/1024 threads, <= 128 VGPRs): twiddles and H are re-read per tile
//          from the L1/L2-resident tables.
// Measured alternatives that lost and were removed (DESIGN.md "FftFilter tuning log"):
// H/twiddles re-read from L1 at 3 waves/SIMD (2.1x slower: TA-bound), H in LDS shared by
// several tiles per workgroup at 3-4 waves/SIMD (1.3-1.7x slower: LDS-bound), register
// prefetch of the next tile (no gain).
template <int LOG2F, int VAR> struct KCfg {
    static constexpr int T = 1 << (LOG2F - 4);
    static constexpr bool TW0_REG = VAR == 0;
    static constexpr bool TW1_REG = VAR == 0;
    static constexpr bool H_REG = VAR == 0;
    static constexpr bool H_LDS = false;
    static constexpr bool PREFETCH = false;
    static constexpr int WAVES_PER_SIMD = VAR == 0 ? 2 : ((T / 64 + 3) / 4 < 2 ? 2 : (T / 64 + 3) / 4);
};

template <int LOG2F, int I, bool PERSIST>
__device__ __forceinline__ void get_tw(creg* dst, const creg* persist, int t, const cf* __restrict__ tw) {
    if constexpr (pass_has_twiddles<LOG2F, I>()) {
        if constexpr (PERSIST) {
#pragma unroll
            for (int k = 0; k < 15; k++) dst[k] = persist[k];
        } else {
            load_twiddles<LOG2F, I>(dst, t, tw);
        }
    }
}

// Exchange synchronisation.  __syncthreads() also drains vmcnt (it is a fence), which would
// serialise the prefetched global loads of the next tile behind every LDS exchange.  A
// one-wave workgroup needs no barrier at all (a wave's LDS operations execute in order);
// larger tiles wait for their own LDS writes only and then meet at a raw s_barrier.
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

template <int LOG2F, int VAR>
__global__ __launch_bounds__((KCfg<LOG2F, VAR>::T), (KCfg<LOG2F, VAR>::WAVES_PER_SIMD))
void k_fftfilt_os(VSrc<cf> src, cf* __restrict__ out, long n_out, int L, long ntiles,
                  const cf* __restrict__ tw, const cf* __restrict__ hpos, int ablate) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    constexpr int NP = Plan<LOG2F>::NP;
    using K = KCfg<LOG2F, VAR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const long S = F - L + 1;
    const int first = L - 1;                 // first valid position of a tile

    // per-thread constants kept in registers for the whole kernel (per K)
    creg tw0[15], tw1[15], hreg[16];
    if constexpr (K::TW0_REG) load_twiddles<LOG2F, 0>(tw0, t, tw);
    if constexpr (K::TW1_REG) load_twiddles<LOG2F, 1>(tw1, t, tw);
    if constexpr (K::H_REG) load_h<LOG2F, NP - 1>(hreg, t, hpos);
    creg* hlds = lds + lds_elems(F);
    if constexpr (K::H_LDS) {
        for (int p = t; p < F; p += T) hlds[lds_pad(p)] = to_reg(hpos[p]);
        __syncthreads();
    }
    const creg* in_reg = reinterpret_cast<const creg*>(src.in);
    creg* out_reg = reinterpret_cast<creg*>(out);

    auto load_tile = [&](long tile, creg* dst) {
        if (RR_ABLATE(16)) tile = 8 + (tile & 63);      // measurement only: inputs from an L2-resident window
        const long v0 = tile * S;            // virtual index of the tile's first sample
        if (v0 >= src.plen && v0 - src.plen + F <= src.in_len) {       // interior tile: plain coalesced loads
            const creg* p = in_reg + (v0 - src.plen) + t;
#pragma unroll
            for (int n = 0; n < 16; n++) dst[n] = p[n * T];
        } else {
#pragma unroll
            for (int n = 0; n < 16; n++) dst[n] = to_reg(src.load(v0 + n * T + t));
        }
    };
    TileIter it(ntiles);
    creg nxt[16];
    if constexpr (K::PREFETCH) {
        if (it.tile < it.end) load_tile(it.tile, nxt);
    }
    for (; it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        if constexpr (VAR != 0) asm volatile("" ::: "memory");  // keep per-tile table loads inside the loop
        creg v[16];
        if constexpr (K::PREFETCH) {
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = nxt[n];
            if (tile + it.step < it.end) load_tile(tile + it.step, nxt);
        } else {
            if (RR_ABLATE(1)) {      // measurement only: no input traffic
#pragma unroll
                for (int n = 0; n < 16; n++) v[n] = mk((float)(t + n), (float)tile);
            } else load_tile(tile, v);
        }
        RR_PHASE();
        creg twl[15];
        const bool do_lds = !RR_ABLATE(4), do_math = !RR_ABLATE(8);
#define lds_store if (do_lds) lds_store
#define lds_load if (do_lds) lds_load
#define fwd_pass if (do_math) fwd_pass
#define inv_pass if (do_math) inv_pass

        // ---- forward ----
        get_tw<LOG2F, 0, K::TW0_REG>(twl, tw0, t, tw);
        fwd_pass<LOG2F, 0>(v, twl);
        lds_store<LOG2F, 0>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 1>(v, t, lds);
        get_tw<LOG2F, 1, K::TW1_REG>(twl, tw1, t, tw);
        fwd_pass<LOG2F, 1>(v, twl);
        lds_store<LOG2F, 1>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 2>(v, t, lds);
        get_tw<LOG2F, 2, false>(twl, tw1, t, tw);
        fwd_pass<LOG2F, 2>(v, twl);
        if constexpr (NP == 4) {
            lds_store<LOG2F, 2>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 3>(v, t, lds);
            fwd_pass<LOG2F, 3>(v, twl);
        }
        // ---- frequency response, then the mirror ----
        if constexpr (K::H_REG) {
            apply_h(v, hreg);
        } else if constexpr (K::H_LDS) {
            creg h[16];
            lds_load<LOG2F, NP - 1>(h, t, hlds);
            apply_h(v, h);
        } else {
            creg h[16];
            load_h<LOG2F, NP - 1>(h, t, hpos);
            apply_h(v, h);
        }
        if constexpr (NP == 4) {
            inv_pass<LOG2F, 3>(v, twl);
            lds_store<LOG2F, 3>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 2>(v, t, lds);
            get_tw<LOG2F, 2, false>(twl, tw1, t, tw);
        }
        inv_pass<LOG2F, 2>(v, twl);
        lds_store<LOG2F, 2>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 1>(v, t, lds);
        get_tw<LOG2F, 1, K::TW1_REG>(twl, tw1, t, tw);
        inv_pass<LOG2F, 1>(v, twl);
        lds_store<LOG2F, 1>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 0>(v, t, lds);
        get_tw<LOG2F, 0, K::TW0_REG>(twl, tw0, t, tw);
        inv_pass<LOG2F, 0>(v, twl);

#undef lds_store
#undef lds_load
#undef fwd_pass
#undef inv_pass
        if (RR_ABLATE(2)) {          // measurement only: no output traffic (keeps v live)
            bool odd = false;
#pragma unroll
            for (int n = 0; n < 16; n++) odd |= (v[n].x == 12345.678f);
            if (!odd) continue;
        }
        // tile positions [L-1, F) are valid linear-convolution outputs
        const long o0 = (RR_ABLATE(32) ? 8 + (tile & 63) : tile) * S - first;   // 32: outputs to an L2-resident window
        creg* po = out_reg + o0 + t;
        if (o0 + F <= n_out) {                                  // whole tile inside the output window
#pragma unroll
            for (int n = 0; n < 16; n++) {
                if (n * T >= first) po[n * T] = v[n];           // wave-uniform
                else if ((n + 1) * T > first) { if (n * T + t >= first) po[n * T] = v[n]; }
            }
        } else {
#pragma unroll
            for (int n = 0; n < 16; n++) {
                const int idx = n * T + t;
                if (idx >= first && o0 + idx < n_out) po[n * T] = v[n];
            }
        }
        // next tile's first lds_store touches exactly the slots this thread just read
    }
}

bool fftfilt_supported(int log2f) { return log2f >= 10 && log2f <= 14; }

int device_cu_count() {
    int dev = 0, n = 0;
    RR_HIP(hipGetDevice(&dev));
    RR_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    return n;
}

template <int LOG2F, int VAR>
static void launch_one(VSrc<cf> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos,
                       hipStream_t s) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const long S = F - L + 1;
    const long ntiles = (n_out + S - 1) / S;
    if (ntiles <= 0) return;
    const size_t smem = sizeof(cf) * lds_elems(F) * (KCfg<LOG2F, VAR>::H_LDS ? 2 : 1);
    static bool attr_set = false;
    static int per_cu = 0;
    if (!attr_set) {
        RR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftfilt_os<LOG2F, VAR>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        RR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fftfilt_os<LOG2F, VAR>, T, smem));
        if (per_cu < 1) per_cu = 1;
        attr_set = true;
    }
    long grid = (long)device_cu_count() * per_cu;
    if (grid > ntiles) grid = ntiles;
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("RR_FFT_ABLATE"); ablate = e ? atoi(e) : 0; }   // measurement knob
    hipLaunchKernelGGL((k_fftfilt_os<LOG2F, VAR>), dim3((unsigned)grid), dim3(T), smem, s, src, out, n_out, L,
                       ntiles, tw, hpos, ablate);
    RR_HIP(hipGetLastError());
}

void launch_fftfilt_os(int log2f, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw,
                       const cf* hpos, hipStream_t s) {
    switch (log2f) {
    case 10: launch_one<10, 0>(src, out, n_out, L, tw, hpos, s); break;
    case 11: launch_one<11, 0>(src, out, n_out, L, tw, hpos, s); break;
    case 12: launch_one<12, 0>(src, out, n_out, L, tw, hpos, s); break;
    case 13: launch_one<13, 3>(src, out, n_out, L, tw, hpos, s); break;
    case 14: launch_one<14, 3>(src, out, n_out, L, tw, hpos, s); break;
    default: throw Error("fftfilt: unsupported tile size");
    }
}

}  // namespace rr
