// kernels_fft.hip — FftFilter on gfx950: overlap-save tiles, one workgroup of F/16
// threads per F-point tile, 16 complex values per thread in VGPRs, in-place
// digit-reversed DIF forward / DIT inverse (fft_core.hpp), LDS only for the radix
// regrouping between passes.  Replaces RustFftEngine::run + the overlap-add loop of
// FftFilter::work (/root/reference/src/fft_filter.rs:172-176, 290-354); results are
// the same linear convolution, computed tile-independently.
#include "kernels.hpp"

namespace rr {

// Register policy.  F <= 4096 (<= 256 threads): the per-thread twiddles of the two
// twiddled passes (30 complex) and the thread's 16 H values stay in VGPRs for the whole
// kernel (every tile uses the same ones), and the next tile's 16 inputs are prefetched
// while the current tile is transformed.  F >= 8192 (512/1024 threads, <= 128 VGPRs):
// twiddles and H are re-read from L1/L2 per tile.
template <int LOG2F> struct KCfg {
    static constexpr bool PERSIST = LOG2F <= 12;
    static constexpr int T = 1 << (LOG2F - 4);
    static constexpr int WAVES_PER_SIMD = PERSIST ? WPS : (T / 64 + 3) / 4;
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

template <int LOG2F>
__global__ __launch_bounds__(KCfg<LOG2F>::T, KCfg<LOG2F>::WAVES_PER_SIMD)
void k_fftfilt_os(VSrc<cf> src, cf* __restrict__ out, long n_out, int L, long ntiles,
                  const cf* __restrict__ tw, const cf* __restrict__ hpos) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    constexpr int NP = Plan<LOG2F>::NP;
    constexpr bool PERSIST = KCfg<LOG2F>::PERSIST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const long S = F - L + 1;
    const int first = L - 1;                 // first valid position of a tile

    // persistent per-thread constants (PERSIST only)
    creg tw0[15], tw1[15], tw2[15], hreg[16];
    if constexpr (PERSIST) {
        load_twiddles<LOG2F, 0>(tw0, t, tw);
        load_twiddles<LOG2F, 1>(tw1, t, tw);
        load_twiddles<LOG2F, 2>(tw2, t, tw);
        load_h<LOG2F, NP - 1>(hreg, t, hpos);
    }
    const creg* in_reg = reinterpret_cast<const creg*>(src.in);
    creg* out_reg = reinterpret_cast<creg*>(out);

    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        const long v0 = tile * S;            // virtual index of the tile's first sample
        if constexpr (!PERSIST) asm volatile("" ::: "memory");  // keep per-tile table loads inside the loop
        creg v[16];
        if (v0 >= src.plen && v0 - src.plen + F <= src.in_len) {       // interior tile: plain coalesced loads
            const creg* p = in_reg + (v0 - src.plen) + t;
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = p[n * T];
        } else {
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = to_reg(src.load(v0 + n * T + t));
        }
        creg twl[15];

        // ---- forward ----
        get_tw<LOG2F, 0, PERSIST>(twl, tw0, t, tw);
        fwd_pass<LOG2F, 0>(v, twl);
        lds_store<LOG2F, 0>(v, t, lds);
        __syncthreads();
        lds_load<LOG2F, 1>(v, t, lds);
        get_tw<LOG2F, 1, PERSIST>(twl, tw1, t, tw);
        fwd_pass<LOG2F, 1>(v, twl);
        lds_store<LOG2F, 1>(v, t, lds);
        __syncthreads();
        lds_load<LOG2F, 2>(v, t, lds);
        get_tw<LOG2F, 2, PERSIST>(twl, tw2, t, tw);
        fwd_pass<LOG2F, 2>(v, twl);
        if constexpr (NP == 4) {
            lds_store<LOG2F, 2>(v, t, lds);
            __syncthreads();
            lds_load<LOG2F, 3>(v, t, lds);
            fwd_pass<LOG2F, 3>(v, twl);
        }
        // ---- frequency response, then the mirror ----
        if constexpr (PERSIST) {
            apply_h(v, hreg);
        } else {
            creg h[16];
            load_h<LOG2F, NP - 1>(h, t, hpos);
            apply_h(v, h);
        }
        if constexpr (NP == 4) {
            inv_pass<LOG2F, 3>(v, twl);
            lds_store<LOG2F, 3>(v, t, lds);
            __syncthreads();
            lds_load<LOG2F, 2>(v, t, lds);
            get_tw<LOG2F, 2, PERSIST>(twl, tw2, t, tw);
        }
        inv_pass<LOG2F, 2>(v, twl);
        lds_store<LOG2F, 2>(v, t, lds);
        __syncthreads();
        lds_load<LOG2F, 1>(v, t, lds);
        get_tw<LOG2F, 1, PERSIST>(twl, tw1, t, tw);
        inv_pass<LOG2F, 1>(v, twl);
        lds_store<LOG2F, 1>(v, t, lds);
        __syncthreads();
        lds_load<LOG2F, 0>(v, t, lds);
        get_tw<LOG2F, 0, PERSIST>(twl, tw0, t, tw);
        inv_pass<LOG2F, 0>(v, twl);

        // tile positions [L-1, F) are valid linear-convolution outputs
        const long o0 = tile * S - first;
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

template <int LOG2F>
static void launch_one(VSrc<cf> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos,
                       hipStream_t s) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const long S = F - L + 1;
    const long ntiles = (n_out + S - 1) / S;
    if (ntiles <= 0) return;
    const size_t smem = sizeof(cf) * lds_elems(F);
    static bool attr_set = false;
    if (!attr_set) {
        RR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fftfilt_os<LOG2F>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    int per_cu = 0;
    RR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fftfilt_os<LOG2F>, T, smem));
    if (per_cu < 1) per_cu = 1;
    long grid = (long)device_cu_count() * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(k_fftfilt_os<LOG2F>, dim3((unsigned)grid), dim3(T), smem, s, src, out, n_out, L,
                       ntiles, tw, hpos);
    RR_HIP(hipGetLastError());
}

void launch_fftfilt_os(int log2f, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw,
                       const cf* hpos, hipStream_t s) {
    switch (log2f) {
    case 10: launch_one<10>(src, out, n_out, L, tw, hpos, s); break;
    case 11: launch_one<11>(src, out, n_out, L, tw, hpos, s); break;
    case 12: launch_one<12>(src, out, n_out, L, tw, hpos, s); break;
    case 13: launch_one<13>(src, out, n_out, L, tw, hpos, s); break;
    case 14: launch_one<14>(src, out, n_out, L, tw, hpos, s); break;
    default: throw Error("fftfilt: unsupported tile size");
    }
}

}  // namespace rr
