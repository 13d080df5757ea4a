// kernels_fft.hip — FftFilter on gfx950: overlap-save tiles, one workgroup of F/16
// threads per F-point tile, 16 complex values per thread in VGPRs, in-place
// digit-reversed DIF forward / DIT inverse (fft_core.hpp), LDS only for the radix
// regrouping between passes.  Replaces RustFftEngine::run + the overlap-add loop of
// FftFilter::work (/root/reference/src/fft_filter.rs:172-176, 290-354); results are
// the same linear convolution, computed tile-independently.
//
// Two kernels share the transform body:
//   k_fftfilt_os : the FftFilter block (tile in -> filtered tile out)
//   k_fm_chain   : FftFilter -> RationalResampler -> QuadratureDemod fused: the filtered tile
//                  never leaves the CU; the resampler pick (src/rational_resampler.rs:183-198)
//                  and the conj-multiply + atan2 (src/quadrature_demod.rs:65-109) run as an
//                  LDS epilogue and only the demodulated f32 stream is written to HBM.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "kernels.hpp"
#include "tile_common.hpp"
#include "nan_fix.hpp"

// Measurement builds (never shipped; make ABLATE=<bits> / TIMING=1 OUT=../lib_ablate):
//   -DRR_FFT_ABLATE_BITS=<bits>  compile-time phase ablation: 1 no input loads, 2 no output stores,
//        4 no LDS exchanges, 8 no butterflies, 16 inputs from an L2-resident window, 32 outputs to
//        an L2-resident window; in k_fftfilt_prune's real-stream path also 64 no second response, 128 no batched tail;
//        in k_fm_chain_split 512 no inverse transforms / output butterfly, 1024 no demodulation.
//        (Compile-time so that the ablated kernel keeps the production
//        register allocation; a runtime flag version spilled 208 B/lane and skewed every number.)
//   -DRR_FFT_TIMING_BUILD  workgroup 0 / thread 0 stamps s_memtime at every phase boundary of its
//        3rd tile (env RR_FFT_STAMPS=1, read back with rr_debug_fft_stamps).
#ifndef RR_FFT_ABLATE_BITS
#define RR_FFT_ABLATE_BITS 0
#endif
#define RR_ABLATE(bit) (((RR_FFT_ABLATE_BITS) & (bit)) != 0)
#ifdef RR_FFT_TIMING_BUILD
#define RR_STAMP(i) do { if (stamps) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define RR_LDSWAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")   // attribute LDS latency to its own interval
#else
#define RR_STAMP(i) do { } while (0)
#define RR_LDSWAIT() do { } while (0)
#endif
// keep the scheduler from overlapping the live ranges of neighbouring phases
#define RR_PHASE() __builtin_amdgcn_sched_barrier(0)

namespace rr {

// Register policy (VAR).  Every tile uses the same per-thread twiddles and H values.
//   VAR 0 (F <= 4096, <= 256 threads): the 30 twiddles of the two twiddled passes and the
//          thread's 16 H values stay in VGPRs for the whole kernel; 2 waves/SIMD.
//   VAR 3 (F >= 8192, 512/1024 threads, <= 128 VGPRs): twiddles and H are re-read per tile
//          from the L1/L2-resident tables.
// Measured alternatives that lost and were removed (profiles/TUNING_LOG.md "FftFilter tuning log"):
// H/twiddles re-read from L1 at 3 waves/SIMD (2.1x slower: TA-bound), H in LDS shared by
// several tiles per workgroup at 3-4 waves/SIMD (1.3-1.7x slower: LDS-bound), register
// prefetch of the next tile (no gain), next tile fetched by LDS-DMA (global_load_lds_dwordx4,
// counted vmcnt; no gain), two tiles per wave interleaved phase by phase (spills; 10 % slower).
template <int LOG2F, int VAR> struct KCfg {
    static constexpr int T = 1 << (LOG2F - 4);
    static constexpr bool REG = VAR == 0;
    static constexpr int WAVES_PER_SIMD = VAR == 0 ? 2 : ((T / 64 + 3) / 4 < 2 ? 2 : (T / 64 + 3) / 4);
};



// Per-thread constants of a kernel instance + the transform of one tile held in v[16]:
// forward FFT, multiply by H, inverse FFT.  On entry v[n] = x[n*T + t]; on exit
// v[n] = y[n*T + t] (circular convolution of the tile with the taps, times 1: H carries 1/F).
template <int LOG2F, int VAR> struct TileXform {
    static constexpr int F = 1 << LOG2F;
    static constexpr int T = F / 16;
    static constexpr int NP = Plan<LOG2F>::NP;
    static constexpr bool REG = KCfg<LOG2F, VAR>::REG;
    creg tw0[15], tw1[15], hreg[16];
    const cf* __restrict__ tw;
    const cf* __restrict__ hpos;
    int t;

    __device__ __forceinline__ void init(int t_, const cf* tw_, const cf* hpos_) {
        t = t_; tw = tw_; hpos = hpos_;
        if constexpr (REG) {
            load_twiddles<LOG2F, 0>(tw0, t, tw);
            load_twiddles<LOG2F, 1>(tw1, t, tw);
            load_h<LOG2F, NP - 1>(hreg, t, hpos);
        }
    }
    // twiddles only: H comes per channel (multi-channel kernel)
    __device__ __forceinline__ void init_no_h(int t_, const cf* tw_) {
        t = t_; tw = tw_; hpos = nullptr;
        if constexpr (REG) {
            load_twiddles<LOG2F, 0>(tw0, t, tw);
            load_twiddles<LOG2F, 1>(tw1, t, tw);
        }
    }
    template <int I> __device__ __forceinline__ void get_tw(creg* dst, const creg* persist) const {
        if constexpr (pass_has_twiddles<LOG2F, I>()) {
            if constexpr (REG && I < 2) {
#pragma unroll
                for (int k = 0; k < 15; k++) dst[k] = persist[k];
            } else {
                load_twiddles<LOG2F, I>(dst, t, tw);
            }
        }
    }
    // forward half: on exit v holds the spectrum in the (digit-reversed) pass-(NP-1) layout
    // (s_setprio 2 around the transform halves: waves in their arithmetic / LDS phases issue ahead of waves that are only
    //  queueing memory operations — same-box A/B at the sustained clocks: FftFilter 0.3337 -> 0.3310 ms, FirFilter 127
    //  taps 0.3202 -> 0.3135; priority 3 is the same)
    __device__ __forceinline__ void forward(creg* v, creg* lds, unsigned long long* stamps = nullptr) const {
        (void)stamps;
        __builtin_amdgcn_s_setprio(2);
        if constexpr (!REG) asm volatile("" ::: "memory");   // keep per-tile table loads inside the tile loop
        creg twl[15];
        const bool do_lds = !RR_ABLATE(4), do_math = !RR_ABLATE(8);
#define lds_store if (do_lds) lds_store
#define lds_load if (do_lds) lds_load
#define fwd_pass if (do_math) fwd_pass
        get_tw<0>(twl, tw0);
        fwd_pass<LOG2F, 0>(v, twl);
        RR_PHASE(); RR_STAMP(2);
        lds_store<LOG2F, 0>(v, t, lds);
        tile_sync<T>();
        RR_PHASE(); RR_STAMP(3);
        lds_load<LOG2F, 1>(v, t, lds);
        RR_LDSWAIT();
        RR_PHASE(); RR_STAMP(4);
        get_tw<1>(twl, tw1);
        fwd_pass<LOG2F, 1>(v, twl);
        RR_PHASE(); RR_STAMP(5);
        lds_store<LOG2F, 1>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 2>(v, t, lds);
        RR_LDSWAIT();
        RR_PHASE(); RR_STAMP(6);
        get_tw<2>(twl, tw1);
        fwd_pass<LOG2F, 2>(v, twl);
        if constexpr (NP == 4) {
            lds_store<LOG2F, 2>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 3>(v, t, lds);
            fwd_pass<LOG2F, 3>(v, twl);
        }
#undef lds_store
#undef lds_load
#undef fwd_pass
        __builtin_amdgcn_s_setprio(0);
    }
    // inverse half (the mirror): spectrum (already multiplied by H) -> v[n] = y[n*T + t]
    __device__ __forceinline__ void inverse(creg* v, creg* lds, unsigned long long* stamps = nullptr) const {
        (void)stamps;
        __builtin_amdgcn_s_setprio(2);
        creg twl[15];
        const bool do_lds = !RR_ABLATE(4), do_math = !RR_ABLATE(8);
#define lds_store if (do_lds) lds_store
#define lds_load if (do_lds) lds_load
#define inv_pass if (do_math) inv_pass
        if constexpr (NP == 4) {
            inv_pass<LOG2F, 3>(v, twl);
            lds_store<LOG2F, 3>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 2>(v, t, lds);
        }
        get_tw<2>(twl, tw1);
        inv_pass<LOG2F, 2>(v, twl);
        RR_PHASE(); RR_STAMP(7);
        lds_store<LOG2F, 2>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 1>(v, t, lds);
        RR_LDSWAIT();
        RR_PHASE(); RR_STAMP(8);
        get_tw<1>(twl, tw1);
        inv_pass<LOG2F, 1>(v, twl);
        RR_PHASE(); RR_STAMP(9);
        lds_store<LOG2F, 1>(v, t, lds);
        tile_sync<T>();
        RR_PHASE();
        lds_load<LOG2F, 0>(v, t, lds);
        RR_LDSWAIT();
        RR_PHASE(); RR_STAMP(10);
        get_tw<0>(twl, tw0);
        inv_pass<LOG2F, 0>(v, twl);
        RR_PHASE(); RR_STAMP(11);
#undef lds_store
#undef lds_load
#undef inv_pass
        __builtin_amdgcn_s_setprio(0);
    }
    __device__ __forceinline__ void run(creg* v, creg* lds, int ablate, unsigned long long* stamps = nullptr) const {
        (void)ablate;
        forward(v, lds, stamps);
        if constexpr (REG) {
            apply_h(v, hreg);
        } else {
            creg h[16];
            load_h<LOG2F, NP - 1>(h, t, hpos);
            apply_h(v, h);
        }
        inverse(v, lds, stamps);
    }
};

// Boundary tiles (touching the carry prefix, the start of the stream or the end of the window)
// are staged through LDS by an out-of-line routine so that their index arithmetic does not
// cost the steady-state path any registers.
template <int T>
__device__ __attribute__((noinline)) void stage_tile_slow(creg* lds, const cf* prefix, long plen, const cf* in,
                                                          long in_len, long v0, int t) {
    for (int n = 0; n < 16; n++) {
        const int p = n * T + t;
        const long vi = v0 + p;
        cf x = mkcf(0.0f, 0.0f);
        if (vi >= 0) {
            if (vi < plen) x = prefix[vi];
            else if (vi - plen < in_len) x = in[vi - plen];
        }
        lds[lds_pad(p)] = to_reg(x);
    }
}

// v[n] = xx[v0 + n*T + t]; interior tiles use plain lane-consecutive loads.
template <int LOG2F> __device__ __forceinline__ void load_tile16(creg* v, const VSrc<cf>& src, long v0, int t, creg* lds) {
    constexpr int T = 1 << (LOG2F - 4);
    if (v0 >= src.plen && v0 - src.plen + 16 * T <= src.in_len) {
        const gptr<creg> p = as_global(reinterpret_cast<const creg*>(src.in) + (v0 - src.plen) + t);
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = p[n * T];
    } else {
        stage_tile_slow<T>(lds, src.prefix, src.plen, src.in, src.in_len, v0, t);
        tile_sync<T>();          // (own slots only: each thread reads back what it wrote)
        lds_load<LOG2F, 0>(v, t, lds);
    }
}

// The same for a window in RTL-SDR wire format (2-byte loads, decoded in registers).
template <int T>
__device__ __attribute__((noinline)) void stage_tile_slow_iq8(creg* lds, VSrcIQ8 src, long v0, int t) {
    for (int n = 0; n < 16; n++) {
        const int p = n * T + t;
        const long vi = v0 + p;
        cf x = mkcf(0.0f, 0.0f);
        if (vi >= 0) x = src.load(vi);
        lds[lds_pad(p)] = to_reg(x);
    }
}
template <int LOG2F> __device__ __forceinline__ void load_tile16(creg* v, const VSrcIQ8& src, long v0, int t, creg* lds) {
    constexpr int T = 1 << (LOG2F - 4);
    if (v0 >= src.plen && v0 - src.plen + 16 * T <= src.in_len) {
        const gptr<unsigned short> p = as_global(reinterpret_cast<const unsigned short*>(src.in) + (v0 - src.plen) + t);
        unsigned short w[16];
#pragma unroll
        for (int n = 0; n < 16; n++) w[n] = p[n * T];
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = to_reg(VSrcIQ8::decode(w[n]));
    } else {
        stage_tile_slow_iq8<T>(lds, src, v0, t);
        tile_sync<T>();
        lds_load<LOG2F, 0>(v, t, lds);
    }
}

// outputs are written once and never re-read by this kernel: streaming (non-temporal) store
__device__ __forceinline__ void store_stream(creg* p, creg v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---- FftFilter -----------------------------------------------------------------------------------
// (FIX 1: FirFilter on these tiles — nan_fix.hpp nf_finish; FIX 2: the FftFilter BLOCK — the reference's blocks, rb_finish;
//  FIX 0: the chains and measurement builds carry none of it)
template <int LOG2F, int VAR, int FIX = 0>
__global__ __launch_bounds__((KCfg<LOG2F, VAR>::T), (KCfg<LOG2F, VAR>::WAVES_PER_SIMD))
void k_fftfilt_os(NanFixCtx nfx, VSrc<cf> src, cf* __restrict__ out, long n_out, int L, long ntiles,
                  const cf* __restrict__ tw, const cf* __restrict__ hpos, int ablate,
                  unsigned long long* __restrict__ dbg, long tile_base, CarryOut carry) {
    (void)nfx;                                   // (nan_fix.hpp: read from the argument segment by nf_finish, never by the body)
    carry_store<cf>(src, carry);
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const long S = F - L + 1;
    if constexpr (FIX) nf_init();
    const int first = L - 1;                 // first valid position of a tile
    TileXform<LOG2F, VAR> X;
    X.init(t, tw, hpos);
    creg* out_reg = reinterpret_cast<creg*>(out);

    int iter = 0;
    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step, iter++) {
        const long tile = tile_base + it.tile;           // (tile_base: a sub-range of the tiles, see launch_fftfilt_t32)
#ifdef RR_FFT_TIMING_BUILD
        unsigned long long* stamps = (dbg && blockIdx.x == 0 && t == 0 && iter == 2) ? dbg : nullptr;
#else
        (void)dbg; (void)iter;
        unsigned long long* stamps = nullptr;
#endif
        RR_STAMP(0);
        creg v[16];
        if (RR_ABLATE(1)) {      // measurement only: no input traffic
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = mk((float)(t + n), (float)tile);
        } else {
            load_tile16<LOG2F>(v, src, (RR_ABLATE(16) ? 8 + (tile & 63) : tile) * S, t, lds);
        }
#ifdef RR_FFT_TIMING_BUILD
        if (stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // attribute the load latency to interval 0-1
#endif
        RR_PHASE(); RR_STAMP(1);
        X.run(v, lds, ablate, stamps);
        if constexpr (FIX) nf_mark(nf_bad(v[15]));   // (a non-finite input makes every output of the tile non-finite)

        if (RR_ABLATE(2)) {          // measurement only: no output traffic (keeps v live)
            bool odd = false;
#pragma unroll
            for (int n = 0; n < 16; n++) odd |= (v[n].x == 12345.678f);
            if (!odd) continue;
        }
        // tile positions [L-1, F) are valid linear-convolution outputs
        const long o0 = (RR_ABLATE(32) ? 8 + (tile & 63) : tile) * S - first;
        creg* po = out_reg + o0 + t;
        if (o0 + F <= n_out) {                                  // whole tile inside the output window
#pragma unroll
            for (int n = 0; n < 16; n++) {
                if (n * T >= first) store_stream(&po[n * T], v[n]);           // wave-uniform
                else if ((n + 1) * T > first) { if (n * T + t >= first) store_stream(&po[n * T], v[n]); }
            }
        } else {
#pragma unroll
            for (int n = 0; n < 16; n++) {
                const int idx = n * T + t;
                if (idx >= first && o0 + idx < n_out) po[n * T] = v[n];
            }
        }
        RR_PHASE(); RR_STAMP(12);
        // next tile's first lds_store touches exactly the slots this thread just read
    }
    if constexpr (FIX == 1) nf_finish<cf, cf>();  // FirFilter on these tiles: the reference's locality for non-finite samples
    if constexpr (FIX == 2) rb_finish<cf>();      // FftFilter: the reference's blocks
}

// ---- decimating FirFilter on the same tiles ----------------------------------------------------------
// y[m] = z[m d], z = the full-rate filter output of k_fftfilt_os (fir.rs:181-189 evaluates only every d-th window;
// here every d-th sample of the filtered tile is kept).  The lanes of one store instruction hold 64 consecutive z,
// so the 64/d samples they keep are contiguous in `out`.  Index arithmetic: g = tile S - first + idx is the z index
// of tile position idx; with gb = tile S + (K d - first), K = ceil(first / d), the position keeps iff
// (gb + idx) % d == 0 and lands at (gb + idx) / d - K.  One 64-bit division per tile, then exact float quotients
// (x < d + F <= 8192 here, so floor((x + 0.5) / d) in f32 is exact).
template <int LOG2F, int VAR>
__global__ __launch_bounds__((KCfg<LOG2F, VAR>::T), (KCfg<LOG2F, VAR>::WAVES_PER_SIMD))
void k_fftfilt_deci(NanFixCtx nfx, VSrc<cf> src, cf* __restrict__ out, long n_out, int L, int d, long ntiles,
                    const cf* __restrict__ tw, const cf* __restrict__ hpos) {
    (void)nfx;
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const long S = F - L + 1;
    const int first = L - 1;
    const long K = (first + d - 1) / d;
    nf_init();
    const float inv_d = 1.0f / (float)d;
    TileXform<LOG2F, VAR> X;
    X.init(t, tw, hpos);
    creg* out_reg = reinterpret_cast<creg*>(out);
    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        creg v[16];
        load_tile16<LOG2F>(v, src, tile * S, t, lds);
        RR_PHASE();
        X.run(v, lds, 0, nullptr);
        nf_mark(nf_bad(v[15]));
        const long gb = tile * S + (K * d - first);
        const long qb = gb / d;
        const int rb = (int)(gb - qb * d) + t;             // (gb + t) - qb d, in [0, d + T)
        creg* po = out_reg + (qb - K);
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int x = rb + n * T;
            const int q = (int)(((float)x + 0.5f) * inv_d);
            const long m = (qb - K) + q;
            if (q * d == x && n * T + t >= first && m < n_out) po[q] = v[n];
        }
        RR_PHASE();
    }
    nf_finish<cf, cf>();
}

// ---- real streams: two segments per Complex tile -------------------------------------------------------
// Real taps commute with taking real / imaginary parts: filtering z = a + i b gives (t * a) + i (t * b).  One
// F-point Complex tile therefore carries TWO consecutive overlap-save segments of a real stream (segment 2j in
// the real, 2j + 1 in the imaginary lane): half the transform work and 8 B of traffic per real sample, against
// 40 B for f32 -> Complex(x, 0) -> FftFilter -> .re (what FftFilterFloat does in the reference,
// fft_filter.rs:365-491) — and the long FirFilter<Float> (fir.rs:113-147) on the same kernel.
// DECI keeps every d-th filtered sample (see k_fftfilt_deci), n_out counts kept samples.
template <int T>
__device__ __attribute__((noinline)) void stage_pair_slow(creg* lds, VSrc<float> src, long va, long vb, int t) {
    for (int n = 0; n < 16; n++) {
        const int p = n * T + t;
        lds[lds_pad(p)] = mk(src.load(va + p), src.load(vb + p));
    }
}
// HILB (round 4): the Hilbert block on these tiles (hilbert.rs:113-116) — taps = the transformer, and the stored sample is
// Complex: out[k] = (xp[k + L/2], y[k]).  The real part is the input delayed by half the filter: re-read from the window
// (this workgroup loaded the line a moment ago) at tile position idx - L/2.
template <int T>
__device__ __attribute__((noinline)) void hilbert_store_slow(const creg* lds, VSrc<float> src, creg* out, long va, long S, int first, long n_out, int t) {
    const long oa = va - first, vb = va + S;
    const int half = first / 2;
    for (int n = 0; n < 16; n++) {
        const int idx = n * T + t;
        const creg v = lds[lds_pad(idx)];
        if (idx >= first && oa + idx < n_out) out[oa + idx] = mk(src.load(va + idx - half), v.x);
        if (idx >= first && oa + S + idx < n_out) out[oa + S + idx] = mk(src.load(vb + idx - half), v.y);
    }
}
template <int LOG2F, bool DECI, bool HILB = false>
__global__ __launch_bounds__((KCfg<LOG2F, 0>::T), (KCfg<LOG2F, 0>::WAVES_PER_SIMD))
void k_fftfilt_real(NanFixCtx nfx, VSrc<float> src, float* __restrict__ out, long n_out, int L, int d, long ntiles,
                    const cf* __restrict__ tw, const cf* __restrict__ hpos, CarryOut carry) {
    (void)nfx;
    static_assert(!(DECI && HILB), "the Hilbert form does not decimate");
    carry_store<float>(src, carry);
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const long S = F - L + 1;
    // (nan_fix.hpp; not in the Hilbert form: it sits at 255 of 256 registers and anything more spills into its tile loop —
    //  a pass behind the kernel gives a Hilbert transformer on these tiles the reference's locality: kernels_misc.hip
    //  k_hilbert_refold_nonfinite, round 5)
    if constexpr (!HILB) nf_init();
    const int first = L - 1;
    const long K = DECI ? (first + d - 1) / d : 0;
    const float inv_d = DECI ? 1.0f / (float)d : 0.0f;
    TileXform<LOG2F, 0> X;
    X.init(t, tw, hpos);
    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long va = 2 * it.tile * S, vb = va + S;       // virtual index of position 0 of the two segments
        creg v[16];
        if (va >= src.plen && vb - src.plen + F <= src.in_len) {
            const float* pa = src.in + (va - src.plen) + t;
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = mk(pa[n * T], pa[S + n * T]);
        } else {
            stage_pair_slow<T>(lds, src, va, vb, t);
            tile_sync<T>();
            lds_load<LOG2F, 0>(v, t, lds);
        }
        RR_PHASE();
        X.run(v, lds, 0, nullptr);
        if constexpr (!HILB) nf_mark(nf_bad(v[15]));         // (either segment: the transform mixes the two)
        if constexpr (HILB) {
            // The kernel sits at 243 of 256 registers without this epilogue, and a spilled register is poison here (a scratch
            // access waits on the whole in-order vmcnt queue: 0.44 instead of ~0.25 ms per 1e8 samples with 18 spilled).  So the
            // thread parks its 16 values in its OWN slots of the idle exchange area (no barrier: it reads back only what it
            // wrote), and the epilogue walks them four rows at a time: read back, re-read the delayed input, store.
#pragma unroll
            for (int n = 0; n < 16; n++) lds[lds_pad(n * T + t)] = v[n];
            __builtin_amdgcn_sched_barrier(0);
            const long oa = va - first;
            const int half = first / 2;
            if (va >= src.plen && vb - src.plen + F <= src.in_len && oa + S + F <= n_out) {
                creg* pc = reinterpret_cast<creg*>(out) + oa + t;
                const gptr<float> px = as_global(src.in + (va - src.plen) - half + t);   // (only positions >= first are read: >= window start)
                constexpr int HG = LOG2F == 11 ? 2 : 4;      // rows per group (2048-point tiles: 4 spill two registers)
#pragma unroll
                for (int g = 0; g < 16 / HG; g++) {
                    float xa[HG], xb[HG];
                    creg y[HG];
#pragma unroll
                    for (int i = 0; i < HG; i++) {
                        const int n = HG * g + i;
                        const bool ok = n * T + t >= first;
                        xa[i] = ok ? px[n * T] : 0.0f;
                        xb[i] = ok ? px[S + n * T] : 0.0f;
                        y[i] = lds[lds_pad(n * T + t)];
                    }
#pragma unroll
                    for (int i = 0; i < HG; i++) {
                        const int n = HG * g + i;
                        if (n * T + t >= first) { pc[n * T] = mk(xa[i], y[i].x); pc[S + n * T] = mk(xb[i], y[i].y); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                hilbert_store_slow<T>(lds, src, reinterpret_cast<creg*>(out), va, S, first, n_out, t);   // boundary tiles, out of line
            }
            tile_sync<T>();                                  // (the next tile's first exchange rewrites those slots)
        } else if constexpr (!DECI) {
            const long oa = va - first;                      // output index of position 0 of segment A
            float* po = out + oa + t;
            if (oa + S + F <= n_out) {
#pragma unroll
                for (int n = 0; n < 16; n++)
                    if (n * T + t >= first) { po[n * T] = v[n].x; po[S + n * T] = v[n].y; }
            } else {
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const int idx = n * T + t;
                    if (idx >= first && oa + idx < n_out) po[n * T] = v[n].x;
                    if (idx >= first && oa + S + idx < n_out) po[S + n * T] = v[n].y;
                }
            }
        } else {
#pragma unroll
            for (int seg = 0; seg < 2; seg++) {
                const long gb = (seg ? vb : va) + (K * d - first);
                const long qb = gb / d;
                const int rb = (int)(gb - qb * d) + t;
                float* po = out + (qb - K);
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const int x = rb + n * T;
                    const int q = (int)(((float)x + 0.5f) * inv_d);
                    if (q * d == x && n * T + t >= first && (qb - K) + q < n_out) po[q] = seg ? v[n].y : v[n].x;
                }
            }
        }
        RR_PHASE();
    }
    if constexpr (!HILB) nf_finish<float, float>();
}

// ---- decimation by the last radix of the tile plan: pruned inverse transform ---------------------------------
// F = 16 * 16 * D (D = 4 / 8 / 16 for 1024 / 2048 / 4096 points).  Time index n = n1 T + n2 D + n3, bin
// k = k1 + 16 k2 + 256 k3.  A decimating filter keeps y[n] only for n3 = c (c = (L - 1) % D when tiles advance
// by a multiple of D), and
//     y[n1, n2, c] = sum_k1 w16^(-n1 k1) w_F^(-k1 (n2 D + c))  sum_k2 w16^(-n2 k2)  z[k1, k2],
//     z[k1, k2]    = sum_k3 Y[k1, k2, k3] w_D^(-c k3) w_16D^(-c k2).
// The two constant factors of z are folded into the frequency response (table hpos2), so after the forward
// transform and the product the last inverse butterfly collapses to a thread-local SUM of the D registers of a
// group, and what remains is a 256-point inverse per tile.  D tiles are parked in LDS and finished together as
// one full-width batch (T = 16 D threads x 16 values): DFT-16 over k2, exchange, twiddle, DFT-16 over k1 — per
// tile 1 forward + 1/D inverse transforms instead of 1 + 1.  Lanes (tile b, n2) hold 16 consecutive kept samples.
//   REAL2 = false: Complex stream (decimating FirFilter<Complex>, deci == D).
//   REAL2 = true : real stream, Complex taps t = Gr + i Gi (the fused Hilbert -> FirFilter): two overlap-save
//                  segments a, b ride in the re / im lanes (k_fftfilt_real); the real tap sets act on them as
//                  Gr*a + i Gr*b and Gi*a + i Gi*b (two products, two sums; a batch is D/2 tiles x 2 responses, so
//                  one park area and one tail), and the outputs are y_a = (Gr*a) + i (Gi*a), y_b = (Gr*b) + i (Gi*b).
template <int T>
__device__ __forceinline__ void prune_tail(creg* v, creg* park, int t, const creg* __restrict__ twb) {
    creg* own = park + 17 * t;                          // lds_pad(16 t + k) = 17 t + k
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = own[k];
    Dft<16, true>::run(v);                              // over k2 -> n2
#pragma unroll
    for (int k = 0; k < 16; k++) own[k] = v[k];
    tile_sync<T>();
    const creg* col = park + 272 * (t >> 4) + (t & 15); // lds_pad(256 b + 16 k1 + n2) = 272 b + 17 k1 + n2
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = cmul(col[17 * k], twb[k]);   // twb: this thread's row of the L1-resident table
    Dft<16, true>::run(v);                              // over k1 -> n1
}

//   MODE 0 = Complex stream, 1 = real stream x Complex taps (REAL2 above), 2 = real stream x real taps (decimating
//   FirFilter<Float>): one response, one tail, y_a / y_b are its real / imaginary parts, f32 output.
template <int LOG2F, int MODE>
__global__ __launch_bounds__((KCfg<LOG2F, 0>::T), (KCfg<LOG2F, 0>::WAVES_PER_SIMD))
void k_fftfilt_prune(NanFixCtx nfx, VSrc<cf> csrc, VSrc<float> rsrc, cf* __restrict__ out, long n_out, int L, long S, long ntiles,
                     const cf* __restrict__ tw, const cf* __restrict__ hpos2, const cf* __restrict__ hpos2b,
                     const cf* __restrict__ twb_tab, CarryOut carry, int sub) {
    (void)nfx;
    nf_init();
    // sub > 1 (round 4): a decimation d = D * sub.  The tail's kept samples are the stream's y[D m]; of those only m = sub q
    // are stored, at out[q] (n_out counts the D-decimated samples).  Per thread the 16 candidates are m0 + 16 n1: m0 is split
    // once into sub * base + r0, and r0 + 16 n1 < sub + 256 is small enough for an exact float reciprocal.
    const float inv_sub = 1.0f / (float)sub;
    auto sub_split = [&](long m0, long& base, int& r0) {
        long r = m0 % sub;
        if (r < 0) r += sub;
        r0 = (int)r;
        base = (m0 - r) / sub;
    };
    auto sub_hit = [&](int r0, int n1, int& q) {
        const int x = r0 + 16 * n1;
        q = (int)(((float)x + 0.5f) * inv_sub);
        return q * sub == x;
    };
    if constexpr (MODE == 0) carry_store<cf>(csrc, carry); else carry_store<float>(rsrc, carry);
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    constexpr int D = F / 256;
    constexpr int U = 16 / D;
    constexpr int PARK = 256 * D + 16 * D;              // lds_elems(256 D)
    constexpr bool REAL2 = MODE != 0, TWO = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* park = lds + lds_elems(F);                    // D slots of 256 values: D tiles, or (MODE 1) D/2 tiles x 2 responses
    constexpr int BT = TWO ? D / 2 : D;                 // tiles per batch
    const int t = threadIdx.x;
    const int first = L - 1;
    const long fq = first / D;                          // kept samples in front of a tile's first valid one
    const long Sd = S / D;                              // kept samples per tile (segment)
    TileXform<LOG2F, 0> X;
    // Complex stream: twiddles and the folded response stay in registers like k_fftfilt_os's.  The real-stream
    // variant has two responses: the first stays in registers, and to make room beside the tail's registers the
    // pass-1 twiddles w_16D^(k (t % D)) — 16 D distinct values — are read per tile from an LDS table; the second
    // response is re-read per tile (both sets of twiddles plus one response spill 80-110 B/lane, the LDS table
    // plus both responses 150 B/lane — and a spill waits on the whole in-order vmcnt queue).
    creg* tw1tab = park + PARK;                         // real streams only: w_16D^j, j < 16 D
    creg hrA[16], hrB[16], tw0p[15];
    const int lo1 = PassGeom<LOG2F, 1>::lo(t);
    if constexpr (REAL2) {
        static_assert(PassGeom<LOG2F, 1>::R * PassGeom<LOG2F, 1>::P == 16 * D, "pass-1 twiddles are powers of w_16D");
        load_twiddles<LOG2F, 0>(tw0p, t, tw);
        load_h<LOG2F, Plan<LOG2F>::NP - 1>(hrA, t, hpos2);
        for (int j = t; j < 16 * D; j += T) tw1tab[j] = to_reg(tw[j * (F / (16 * D))]);
        tile_sync<T>();
    } else {
        X.init(t, tw, hpos2);
    }
    creg* out_reg = reinterpret_cast<creg*>(out);
    const long nbatch = (ntiles + BT - 1) / BT;
    for (TileIter it(nbatch); it.tile < it.end; it.tile += it.step) {
        const long tile0 = it.tile * BT;
#pragma unroll 1
        for (int b = 0; b < BT; b++) {
            const long tile = tile0 + b;
            creg v[16];
            if constexpr (REAL2) {
                const long va = 2 * tile * S, vb = va + S;
                if (RR_ABLATE(256) && va >= rsrc.plen && vb - rsrc.plen + F <= rsrc.in_len) {   // measurement only: the tile's 16 KB as 8 dense 16-byte loads per lane (wrong data)
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    const f4* pa = reinterpret_cast<const f4*>(rsrc.in + ((va - rsrc.plen) & ~3L)) + t;
#pragma unroll
                    for (int n = 0; n < 4; n++) {
                        const f4 qa = pa[n * T], qb = pa[(S & ~3L) / 4 + n * T];
                        v[4 * n] = mk(qa.x, qb.x); v[4 * n + 1] = mk(qa.y, qb.y); v[4 * n + 2] = mk(qa.z, qb.z); v[4 * n + 3] = mk(qa.w, qb.w);
                    }
                } else if (RR_ABLATE(1)) {   // measurement only: no input traffic
#pragma unroll
                    for (int n = 0; n < 16; n++) v[n] = mk((float)(t + n) * 1e-3f, (float)(tile & 255) * 1e-3f);
                } else if (va >= rsrc.plen && vb - rsrc.plen + F <= rsrc.in_len) {
                    const float* pa = rsrc.in + (va - rsrc.plen) + t;
#pragma unroll
                    for (int n = 0; n < 16; n++) v[n] = mk(pa[n * T], pa[S + n * T]);
                } else {
                    stage_pair_slow<T>(lds, rsrc, va, vb, t);
                    tile_sync<T>();
                    lds_load<LOG2F, 0>(v, t, lds);
                }
            } else {
                load_tile16<LOG2F>(v, csrc, tile * S, t, lds);
            }
            RR_PHASE();
            if constexpr (REAL2) {                       // TileXform::forward with the pass-1 twiddles from the LDS table
                creg twl[15];                            // (no s_setprio here: it costs this kernel 4 %)
                fwd_pass<LOG2F, 0>(v, tw0p);
                RR_PHASE();
                lds_store<LOG2F, 0>(v, t, lds);
                tile_sync<T>();
                lds_load<LOG2F, 1>(v, t, lds);
#pragma unroll
                for (int k = 1; k < 16; k++) twl[k - 1] = tw1tab[k * lo1];
                RR_PHASE();
                fwd_pass<LOG2F, 1>(v, twl);
                RR_PHASE();
                lds_store<LOG2F, 1>(v, t, lds);
                tile_sync<T>();
                lds_load<LOG2F, 2>(v, t, lds);
                RR_PHASE();
                fwd_pass<LOG2F, 2>(v, twl);              // (P == 1: no twiddles)
            } else {
                X.forward(v, lds);
            }
            tile_sync<T>();      // the next tile's first exchange overwrites slots other waves read in this one's last
            // product with the folded response(s), then the collapsed last inverse butterfly: a plain sum
#pragma unroll
            for (int u = 0; u < U; u++) {
                const creg* h1 = REAL2 ? hrA : X.hreg;
                creg z = cmul(v[u * D], h1[u * D]);
#pragma unroll
                for (int k = 1; k < D; k++) z = cadd(z, cmul(v[u * D + k], h1[u * D + k]));
                park[lds_pad(256 * (TWO ? 2 * b : b) + t + T * u)] = z;
            }
            if constexpr (TWO && !RR_ABLATE(64)) {
                RR_PHASE();
                load_h<LOG2F, Plan<LOG2F>::NP - 1>(hrB, t, hpos2b);   // the second response is re-read per tile (L1 / L2)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    creg zb = cmul(v[u * D], hrB[u * D]);
#pragma unroll
                    for (int k = 1; k < D; k++) zb = cadd(zb, cmul(v[u * D + k], hrB[u * D + k]));
                    park[lds_pad(256 * (2 * b + 1) + t + T * u)] = zb;
                }
            }
            RR_PHASE();
        }
        tile_sync<T>();
        // ---- the batch's 256-point inverses: thread (b, n2) ends with the kept samples 16 n1 + n2 of tile b
        const creg* twb = reinterpret_cast<const creg*>(twb_tab) + 16 * (t & 15);
        const int b = t >> 4, n2 = t & 15;
        const int lo = (int)fq - n2, hi = lo + (int)Sd;              // valid kept samples: lo <= 16 n1 < hi
        if constexpr (MODE == 2) {                       // real taps: (t*a) + i (t*b), f32 output
            creg p[16];
            prune_tail<T>(p, park, t, twb);
            nf_mark(nf_bad(p[15]));                      // (nan_fix.hpp: this thread's tile: non-finite iff its input was)
            const long ma = 2 * (tile0 + b) * Sd - fq + n2, mb = ma + Sd;
            const long ra = n_out - ma, rb = n_out - mb;
            const int lima = ra < hi ? (int)ra : hi, limb = rb < hi ? (int)rb : hi;
            float* pa = reinterpret_cast<float*>(out) + ma;
            float* pb = reinterpret_cast<float*>(out) + mb;
            if (sub == 1) {
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++) {
                    if (16 * n1 >= lo && 16 * n1 < lima) pa[16 * n1] = p[n1].x;
                    if (16 * n1 >= lo && 16 * n1 < limb) pb[16 * n1] = p[n1].y;
                }
            } else {
                long ba, bb; int ra, rb2;
                sub_split(ma, ba, ra); sub_split(mb, bb, rb2);
                float* of = reinterpret_cast<float*>(out);
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++) {
                    int q;
                    if (16 * n1 >= lo && 16 * n1 < lima && sub_hit(ra, n1, q)) of[ba + q] = p[n1].x;
                    if (16 * n1 >= lo && 16 * n1 < limb && sub_hit(rb2, n1, q)) of[bb + q] = p[n1].y;
                }
            }
        } else if constexpr (!REAL2) {
            creg p[16];
            prune_tail<T>(p, park, t, twb);
            nf_mark(nf_bad(p[15]));
            const long m0 = (tile0 + b) * Sd - fq + n2;
            const long room = n_out - m0;
            const int lim = room < hi ? (int)room : hi;
            creg* po = out_reg + m0;
            if (sub == 1) {
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++)
                    if (16 * n1 >= lo && 16 * n1 < lim) po[16 * n1] = p[n1];
            } else {
                long bs; int rs;
                sub_split(m0, bs, rs);
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++) {
                    int q;
                    if (16 * n1 >= lo && 16 * n1 < lim && sub_hit(rs, n1, q)) out_reg[bs + q] = p[n1];
                }
            }
        } else {
            // slot = (tile b2, response c): c = 0 holds (Gr*a) + i (Gr*b), the real parts of y_a (segment 2 tile) and
            // y_b (segment 2 tile + 1); c = 1 holds (Gi*a) + i (Gi*b), their imaginary parts.  The two threads of a
            // pair (t ^ 16) swap results through the (now idle) exchange area; thread c then stores whole Complex
            // samples of segment 2 tile + c (storing the parts with 4-byte strided stores costs 1.34x the write traffic).
            const int b2 = b >> 1, c = b & 1;
            creg p[16];
            if (RR_ABLATE(128)) {
#pragma unroll
                for (int k = 0; k < 16; k++) p[k] = park[17 * t + k];
            } else {
                prune_tail<T>(p, park, t, twb);
            }
            nf_mark(nf_bad(p[15]));
            creg* stash = lds + 17 * t;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) stash[n1] = p[n1];
            tile_sync<T>();
            const creg* other = lds + 17 * (t ^ 16);
            const long m0 = (2 * (tile0 + b2) + c) * Sd - fq + n2;
            const long room = n_out - m0;
            const int lim = room < hi ? (int)room : hi;
            creg* po = out_reg + m0;
            long bs = 0; int rs = 0;
            if (sub != 1) sub_split(m0, bs, rs);
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) {
                const creg q = other[n1];
                if (RR_ABLATE(2)) { if (q.x == 1234.5678f) po[16 * n1] = q; }
                else if (16 * n1 >= lo && 16 * n1 < lim) {
                    const creg val = c ? mk(q.y, p[n1].y) : mk(p[n1].x, q.x);
                    int qi;
                    if (sub == 1) po[16 * n1] = val;
                    else if (sub_hit(rs, n1, qi)) out_reg[bs + qi] = val;
                }
            }
        }
        tile_sync<T>();                                  // the next batch parks into the slots just read
    }
    if constexpr (MODE == 0) nf_finish<cf, cf>();
    else if constexpr (MODE == 1) nf_finish<float, cf>();
    else nf_finish<float, float>();
}

// ---- FftFilter tiles of 8192 / 16384 points as NSUB = 2 / 4 sub-transforms of 4096 points ---------------------
// A 512/1024-thread tile fits one workgroup per CU and cannot keep its tables in registers (k_fftfilt_os<13|14, 3>
// costs 6x / 22x a 4096-point tile).  Split in frequency instead (M = 4096, F = NSUB M, n < M):
//     u_r[n]  = w_F^(r n) sum_s w_NSUB^(r s) x[n + s M]                     input butterfly + twiddle
//     X[NSUB k + r] = FFT_M(u_r)[k],   Y = X H,   z_r = IFFT_M(Y[NSUB k + r])   the 4096-point filter body, per r
//     y[n + s M] = sum_r w_NSUB^(-r s) conj(w_F^(r n)) z_r[n]                output twiddle + butterfly
// Both butterflies act on the NSUB quarters of the tile at the same n, and a thread's n are the same for all
// quarters, so they are thread-local: the u_r / z_r wait in the thread's own natural-order LDS slots, nothing is
// exchanged beyond the sub-transforms' own passes, global loads and stores stay lane-consecutive, and each
// sub-transform uses its parking area as its exchange area (NSUB x 34.8 KB: 2 workgroups per CU for NSUB = 2).
// Tables: wk[t] = w_F^t (w_F^(n T + t) = wk[t] * constant);  hs[r][p] = H[NSUB bin(p) + r] / F in the 4096-point position order.
// w_F^(n T), n < 16, for F / T = 32 (two sub-transforms) and 64 (four): the time-domain twiddle of a thread's n-th
// element is w_F^(n T + t) = w_F^t * w_F^(n T) — one persistent register and these constants instead of a table read
__device__ const float kStep32[16][2] = {{1.000000000e+00f, -0.000000000e+00f}, {9.807852804e-01f, -1.950903220e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {6.123233996e-17f, -1.000000000e+00f}, {-1.950903220e-01f, -9.807852804e-01f}, {-3.826834324e-01f, -9.238795325e-01f}, {-5.555702330e-01f, -8.314696123e-01f}, {-7.071067812e-01f, -7.071067812e-01f}, {-8.314696123e-01f, -5.555702330e-01f}, {-9.238795325e-01f, -3.826834324e-01f}, {-9.807852804e-01f, -1.950903220e-01f}};
__device__ const float kStep64[16][2] = {{1.000000000e+00f, -0.000000000e+00f}, {9.951847267e-01f, -9.801714033e-02f}, {9.807852804e-01f, -1.950903220e-01f}, {9.569403357e-01f, -2.902846773e-01f}, {9.238795325e-01f, -3.826834324e-01f}, {8.819212643e-01f, -4.713967368e-01f}, {8.314696123e-01f, -5.555702330e-01f}, {7.730104534e-01f, -6.343932842e-01f}, {7.071067812e-01f, -7.071067812e-01f}, {6.343932842e-01f, -7.730104534e-01f}, {5.555702330e-01f, -8.314696123e-01f}, {4.713967368e-01f, -8.819212643e-01f}, {3.826834324e-01f, -9.238795325e-01f}, {2.902846773e-01f, -9.569403357e-01f}, {1.950903220e-01f, -9.807852804e-01f}, {9.801714033e-02f, -9.951847267e-01f}};
template <int NSUB> __device__ __forceinline__ creg split_step(int n) {
    return NSUB == 2 ? mk(kStep32[n][0], kStep32[n][1]) : mk(kStep64[n][0], kStep64[n][1]);
}

template <int T, class SRC>
__device__ __attribute__((noinline)) void stage_tile_slow_at(creg* lds, SRC src, long v0, int t) {
    for (int n = 0; n < 16; n++) {
        const long vi = v0 + n * T + t;                 // negative in front of the first tile of a fused chain
        lds[lds_pad(n * T + t)] = vi >= 0 ? to_reg(src.load(vi)) : mk(0.0f, 0.0f);
    }
}
// sample i of the caller's window (interior tiles only)
__device__ __forceinline__ creg window_at(const VSrc<cf>& src, long i) { return as_global(reinterpret_cast<const creg*>(src.in))[i]; }
__device__ __forceinline__ creg window_at(const VSrcIQ8& src, long i) {
    return to_reg(VSrcIQ8::decode(as_global(reinterpret_cast<const unsigned short*>(src.in))[i]));
}

// DECI: keep every d-th filtered sample (out[m] = y[m d], n_out counts kept samples) — the index arithmetic of
// k_fftfilt_deci; the decimating FirFilter with more taps than a 4096-point tile takes.
template <int NSUB, bool DECI, bool FIX = false>
__global__ __launch_bounds__(256, 2)
void k_fftfilt_split(NanFixCtx nfx, VSrc<cf> src, cf* __restrict__ out, long n_out, int L, int d, long ntiles, const cf* __restrict__ tw,
                     const cf* __restrict__ hs, const cf* __restrict__ wk, CarryOut carry) {
    (void)nfx;
    carry_store<cf>(src, carry);
    constexpr int LOG2M = 12, M = 1 << LOG2M, T = M / 16, F = NSUB * M;
    if constexpr (FIX) nf_init();
    constexpr int NP = Plan<LOG2M>::NP;
    constexpr int LE = lds_elems(M);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* area = reinterpret_cast<creg*>(smem_raw);     // NSUB areas of LE slots: u_r, the r-th transform's exchanges, z_r
    const int t = threadIdx.x;
    const long S = F - L + 1;
    const int first = L - 1;
    TileXform<LOG2M, 0> X;
    X.init_no_h(t, tw);
    const creg wbase = to_reg(wk[t]);                   // w_F^t
    creg* out_reg = reinterpret_cast<creg*>(out);

    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long v0 = it.tile * S;                    // virtual index of tile position 0
        const bool interior = v0 >= src.plen && v0 - src.plen + F <= src.in_len;
        if (!interior) {                                // boundary tiles: quarters staged out of line, then as below
#pragma unroll
            for (int s = 0; s < NSUB; s++) stage_tile_slow_at<T>(area + s * LE, src, v0 + (long)s * M, t);
        }
        // ---- input butterfly across the quarters (thread-local), u_r -> own natural slots of area r.
        //      All loads of a batch are issued before the first use (one memory latency per batch, not per n).
        {
            const creg* p = reinterpret_cast<const creg*>(src.in) + (v0 - src.plen) + t;
            constexpr int NB = 16 / NSUB;                // positions per batch: 16 values + NB twiddles in flight
#pragma unroll 1
            for (int n0 = 0; n0 < 16; n0 += NB) {
                creg xin[NSUB][NB], wkr[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) wkr[k] = cmul(wbase, split_step<NSUB>(n0 + k));
                if (interior) {
#pragma unroll
                    for (int s = 0; s < NSUB; s++)
#pragma unroll
                        for (int k = 0; k < NB; k++) xin[s][k] = p[(long)s * M + (n0 + k) * T];
                } else {
#pragma unroll
                    for (int s = 0; s < NSUB; s++)
#pragma unroll
                        for (int k = 0; k < NB; k++) xin[s][k] = area[s * LE + lds_pad((n0 + k) * T + t)];
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    const int n = n0 + k;
                    creg e[NSUB];
#pragma unroll
                    for (int s = 0; s < NSUB; s++) e[s] = xin[s][k];
                    Dft<NSUB, false>::run(e);           // across s -> r
                    const creg w1 = wkr[k];
                    e[1] = cmul(e[1], w1);
                    if constexpr (NSUB == 4) { const creg w2 = cmul(w1, w1); e[2] = cmul(e[2], w2); e[3] = cmul(e[3], cmul(w2, w1)); }
#pragma unroll
                    for (int r = 0; r < NSUB; r++) area[r * LE + lds_pad(n * T + t)] = e[r];
                }
            }
        }
        // ---- the 4096-point filter body per r, exchanges in the r-th area, z_r back to the own natural slots
#pragma unroll 1
        for (int r = 0; r < NSUB; r++) {
            creg* lds = area + r * LE;
            creg v[16];
            creg h[16];
            load_h<LOG2M, NP - 1>(h, t, hs + (long)r * M);     // in flight during the forward transform
            lds_load<LOG2M, 0>(v, t, lds);
            X.forward(v, lds);
            apply_h(v, h);
            X.inverse(v, lds);
            lds_store<LOG2M, 0>(v, t, lds);
        }
        if constexpr (FIX) nf_mark(nf_bad(area[lds_pad(t)]));   // (z_0 of this thread: non-finite iff the tile's input was)
        // ---- output twiddle + butterfly, lane-consecutive stores
        const long o0 = it.tile * S - first;            // output index of tile position 0
        creg* po = out_reg + o0 + t;
        const bool whole = o0 + F <= n_out;
        long qbK = 0;                                   // DECI: see k_fftfilt_deci
        int rb = 0;
        float inv_d = 0.0f;
        if constexpr (DECI) {
            const long K = (first + d - 1) / d;
            const long gb = it.tile * S + (K * d - first), qb = gb / d;
            rb = (int)(gb - qb * d) + t;
            qbK = qb - K;
            inv_d = 1.0f / (float)d;
        }
#pragma unroll 1
        for (int n0 = 0; n0 < 16; n0 += 8) {
            creg wko[8];
#pragma unroll
            for (int k = 0; k < 8; k++) wko[k] = cmul(wbase, split_step<NSUB>(n0 + k));
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int n = n0 + k;
                const int slot = lds_pad(n * T + t);
                creg e[NSUB];
#pragma unroll
                for (int r = 0; r < NSUB; r++) e[r] = area[r * LE + slot];
                const creg w1 = wko[k];
                e[1] = cmulc(e[1], w1);
                if constexpr (NSUB == 4) { const creg w2 = cmul(w1, w1); e[2] = cmulc(e[2], w2); e[3] = cmulc(e[3], cmul(w2, w1)); }
                Dft<NSUB, true>::run(e);                // across r -> s
#pragma unroll
                for (int s = 0; s < NSUB; s++) {
                    const long pos = (long)s * M + n * T + t;
                    if constexpr (DECI) {
                        const int x = rb + s * M + n * T;            // < d + F <= 20480: exact in f32
                        const int q = (int)(((float)x + 0.5f) * inv_d);
                        if (q * d == x && pos >= first && qbK + q < n_out) out_reg[qbK + q] = e[s];
                    } else {
                        if (pos >= first && (whole || o0 + pos < n_out)) po[(long)s * M + n * T] = e[s];
                    }
                }
            }
        }
        // (the next tile's first writes go to the own natural slots this thread just read)
    }
    if constexpr (FIX) nf_finish<cf, cf>();
}

int fftfilt_split_bin(int p) { return bin_of_pos<12>(p); }

template <int NSUB, bool DECI>
static void launch_split_one(VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw, const cf* hs, const cf* wk, hipStream_t s,
                             CarryOut carry = {}, NanFix fx = {}) {
    constexpr int M = 4096, F = NSUB * M;
    const long S = F - L + 1;
    if (n_out <= 0) { launch_carry(src, carry, s); return; }
    const long n_full = DECI ? (n_out - 1) * (long)d + 1 : n_out;
    const long ntiles = (n_full + S - 1) / S;
    const size_t smem = sizeof(cf) * lds_elems(M) * NSUB;
    const NanFixCtx nfx = nanfix_ctx(fx, src, out, S, DECI ? d : 1, n_out, ntiles);
    if (fx.rev) {
        const long grid = grid_for_tiles(k_fftfilt_split<NSUB, DECI, true>, 256, smem, ntiles);
        hipLaunchKernelGGL((k_fftfilt_split<NSUB, DECI, true>), dim3((unsigned)grid), dim3(256), smem, s, nfx, src, out, n_out, L, d, ntiles, tw, hs, wk, carry);
    } else {
        const long grid = grid_for_tiles(k_fftfilt_split<NSUB, DECI, false>, 256, smem, ntiles);
        hipLaunchKernelGGL((k_fftfilt_split<NSUB, DECI, false>), dim3((unsigned)grid), dim3(256), smem, s, nfx, src, out, n_out, L, d, ntiles, tw, hs, wk, carry);
    }
    RR_HIP(hipGetLastError());
}
void launch_fftfilt_split(int nsub, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw4096, const cf* hs, const cf* wk,
                          hipStream_t s, CarryOut carry, NanFix fx) {
    if (nsub == 2) launch_split_one<2, false>(src, out, n_out, L, 1, tw4096, hs, wk, s, carry, fx);
    else if (nsub == 4) launch_split_one<4, false>(src, out, n_out, L, 1, tw4096, hs, wk, s, carry, fx);
    else throw Error("fftfilt_split: 2 or 4 sub-transforms");
}
void launch_fftfilt_split_deci(int nsub, VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw4096, const cf* hs,
                               const cf* wk, hipStream_t s, NanFix fx) {
    if (d < 1 || d > 4096) throw Error("fftfilt_split_deci: decimation out of range");
    if (nsub == 2) launch_split_one<2, true>(src, out, n_out, L, d, tw4096, hs, wk, s, CarryOut{}, fx);
    else if (nsub == 4) launch_split_one<4, true>(src, out, n_out, L, d, tw4096, hs, wk, s, CarryOut{}, fx);
    else throw Error("fftfilt_split: 2 or 4 sub-transforms");
}

// ---- FftStream (src/fft_stream.rs:71-117): forward FFT of consecutive frames --------------------------
// Frames of 1024..16384 points are one tile each: the forward half of the filter transform, then one
// more LDS exchange that undoes the digit reversal (position p holds bin bin_of_pos(p)) so that the
// stores are lane-consecutive in natural bin order.  16 B per sample.
template <int LOG2F, int VAR>
__global__ __launch_bounds__((KCfg<LOG2F, VAR>::T), (KCfg<LOG2F, VAR>::WAVES_PER_SIMD))
void k_fft_frames(const cf* __restrict__ in, cf* __restrict__ out, long nframes, const cf* __restrict__ tw) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    constexpr int NP = Plan<LOG2F>::NP;
    using G = PassGeom<LOG2F, NP - 1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    TileXform<LOG2F, VAR> X;
    X.init_no_h(t, tw);
    for (TileIter it(nframes); it.tile < it.end; it.tile += it.step) {
        const creg* p = reinterpret_cast<const creg*>(in) + it.tile * F + t;
        creg v[16];
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = p[n * T];
        X.forward(v, lds);
        tile_sync<T>();                      // the last exchange's reads are done everywhere
#pragma unroll
        for (int u = 0; u < G::U; u++)
#pragma unroll
            for (int n = 0; n < G::R; n++)
                lds[lds_pad(bin_of_pos<LOG2F>(G::pos(t + G::T * u, n)))] = v[u * G::R + n];
        tile_sync<T>();
        creg* po = reinterpret_cast<creg*>(out) + it.tile * F + t;
#pragma unroll
        for (int n = 0; n < 16; n++) po[n * T] = lds[lds_pad(n * T + t)];
        tile_sync<T>();                      // before the next frame's first exchange overwrites the slots
    }
}

// Frames of any other size N <= F/2 + 1 (the reference plans every size with rustfft, fft_stream.rs:43-44): Bluestein's
// chirp-z identity n k = (n^2 + k^2 - (k - n)^2) / 2 turns the N-point transform into a circular convolution of F
// points, which is exactly the filter tile:  X[k] = c[k] * IFFT_F( FFT_F(x c zero-padded) * B )[k],
// c[n] = exp(-i pi n^2 / N), B = FFT_F(b) / F with b[m] = conj(c[|m|]) wrapped.  chirp = c, hpos = B in position order.
template <int LOG2F>
__global__ __launch_bounds__((KCfg<LOG2F, 0>::T), (KCfg<LOG2F, 0>::WAVES_PER_SIMD))
void k_fft_bluestein(const cf* __restrict__ in, cf* __restrict__ out, long nframes, int N, const cf* __restrict__ tw,
                     const cf* __restrict__ hpos, const cf* __restrict__ chirp) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    TileXform<LOG2F, 0> X;
    X.init(t, tw, hpos);
    const creg* ch = reinterpret_cast<const creg*>(chirp);
    for (TileIter it(nframes); it.tile < it.end; it.tile += it.step) {
        const creg* p = reinterpret_cast<const creg*>(in) + it.tile * N + t;
        creg v[16];
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int idx = n * T + t;
            v[n] = idx < N ? cmul(p[n * T], ch[idx]) : mk(0.0f, 0.0f);
        }
        RR_PHASE();
        X.run(v, lds, 0, nullptr);
        creg* po = reinterpret_cast<creg*>(out) + it.tile * N + t;
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int idx = n * T + t;
            if (idx < N) po[n * T] = cmul(v[n], ch[idx]);
        }
        RR_PHASE();
    }
}

// Frames of 2..512 points: radix-2 Stockham autosort in LDS, one butterfly per thread and stage, 512
// points (512/N frames) per 256-thread workgroup.  Small transforms are not the hot path.
__global__ __launch_bounds__(256) void k_fft_small(const cf* __restrict__ in, cf* __restrict__ out, long nframes,
                                                   int log2n, const cf* __restrict__ tw) {
    __shared__ cf bufA[512], bufB[512];
    const int N = 1 << log2n, half = N >> 1;
    const int fpw = 512 / N;                               // frames per workgroup pass
    const int t = threadIdx.x;
    const int fl = t / half, i = t % half;                 // local frame, butterfly index
    for (long f0 = (long)blockIdx.x * fpw; f0 < nframes; f0 += (long)gridDim.x * fpw) {
        const long rem = nframes - f0;
        const int nf = rem < fpw ? (int)rem : fpw;
        for (int e = t; e < nf * N; e += 256) bufA[e] = in[f0 * N + e];
        __syncthreads();
        cf* x = bufA + fl * N;
        cf* y = bufB + fl * N;
        int n = N, s = 1;
        for (int st = 0; st < log2n; st++) {
            const int m = n >> 1;
            if (fl < nf) {
                const int pp = i / s, q = i % s;
                const cf a = x[q + s * pp], b = x[q + s * (pp + m)];
                const cf w = tw[pp * s];                       // w_n^p = w_N^(p N/n), N/n = s
                const cf d = mkcf(a.x - b.x, a.y - b.y);
                y[q + s * (2 * pp)] = mkcf(a.x + b.x, a.y + b.y);
                y[q + s * (2 * pp + 1)] = mkcf(d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x);
            }
            __syncthreads();
            cf* tmp = x; x = y; y = tmp;
            n = m; s <<= 1;
        }
        const cf* res = (log2n & 1) ? bufB : bufA;
        for (int e = t; e < nf * N; e += 256) out[f0 * N + e] = res[e];
        __syncthreads();
    }
}

// Frames of 8192 / 16384 points: the frequency split of k_fftfilt_split, forward half only.  Sub-transform r yields
// the bins NSUB k + r; each sub-spectrum is scattered into its area at its bin index k and the frame is then
// written lane-consecutively in natural order (bin j = NSUB k + r sits in area r, slot k).
// twF = w_F^i (i < F), tw4096 = w_4096^k.
template <int NSUB>
__global__ __launch_bounds__(256, 2)
void k_fft_frames_split(const cf* __restrict__ in, cf* __restrict__ out, long nframes, const cf* __restrict__ tw4096,
                        const cf* __restrict__ twF) {
    constexpr int LOG2M = 12, M = 1 << LOG2M, T = M / 16, F = NSUB * M;
    constexpr int NP = Plan<LOG2M>::NP;
    constexpr int LE = lds_elems(M);
    using G = PassGeom<LOG2M, NP - 1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* area = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    TileXform<LOG2M, 0> X;
    X.init_no_h(t, tw4096);
    const creg wbase = to_reg(twF[t]);                  // w_F^t
    for (TileIter it(nframes); it.tile < it.end; it.tile += it.step) {
        const creg* p = reinterpret_cast<const creg*>(in) + it.tile * F + t;
        constexpr int NB = 16 / NSUB;
#pragma unroll 1
        for (int n0 = 0; n0 < 16; n0 += NB) {
            creg xin[NSUB][NB], wkr[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) wkr[k] = cmul(wbase, split_step<NSUB>(n0 + k));
#pragma unroll
            for (int s = 0; s < NSUB; s++)
#pragma unroll
                for (int k = 0; k < NB; k++) xin[s][k] = p[(long)s * M + (n0 + k) * T];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                creg e[NSUB];
#pragma unroll
                for (int s = 0; s < NSUB; s++) e[s] = xin[s][k];
                Dft<NSUB, false>::run(e);
                const creg w1 = wkr[k];
                e[1] = cmul(e[1], w1);
                if constexpr (NSUB == 4) { const creg w2 = cmul(w1, w1); e[2] = cmul(e[2], w2); e[3] = cmul(e[3], cmul(w2, w1)); }
#pragma unroll
                for (int r = 0; r < NSUB; r++) area[r * LE + lds_pad((n0 + k) * T + t)] = e[r];
            }
        }
#pragma unroll 1
        for (int r = 0; r < NSUB; r++) {
            creg* lds = area + r * LE;
            creg v[16];
            lds_load<LOG2M, 0>(v, t, lds);
            X.forward(v, lds);
            tile_sync<T>();                              // the last exchange is read everywhere: the area can be rewritten
#pragma unroll
            for (int n = 0; n < 16; n++) lds[lds_pad(bin_of_pos<LOG2M>(G::pos(t, n)))] = v[n];
        }
        tile_sync<T>();
        creg* po = reinterpret_cast<creg*>(out) + it.tile * F + t;
#pragma unroll 8
        for (int c = 0; c < 16 * NSUB; c++) {
            const int j = c * T + t;                     // natural bin, lane-consecutive
            po[c * T] = area[(j % NSUB) * LE + lds_pad(j / NSUB)];
        }
        tile_sync<T>();                                  // before the next frame overwrites the areas
    }
}
template <int NSUB>
static void launch_frames_split(const cf* in, cf* out, long nframes, const cf* tw4096, const cf* twF, hipStream_t s) {
    const size_t smem = sizeof(cf) * lds_elems(4096) * NSUB;
    const long grid = grid_for_tiles(k_fft_frames_split<NSUB>, 256, smem, nframes);
    hipLaunchKernelGGL((k_fft_frames_split<NSUB>), dim3((unsigned)grid), dim3(256), smem, s, in, out, nframes, tw4096, twF);
    RR_HIP(hipGetLastError());
}

template <int LOG2F, int VAR>
static void launch_frames_one(const cf* in, cf* out, long nframes, const cf* tw, hipStream_t s) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fft_frames<LOG2F, VAR>, T, smem, nframes);
    hipLaunchKernelGGL((k_fft_frames<LOG2F, VAR>), dim3((unsigned)grid), dim3(T), smem, s, in, out, nframes, tw);
    RR_HIP(hipGetLastError());
}
template <int LOG2F>
static void launch_bluestein_one(const cf* in, cf* out, long nframes, int N, const cf* tw, const cf* hpos, const cf* chirp,
                                 hipStream_t s) {
    constexpr int F = 1 << LOG2F, T = F / 16;
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fft_bluestein<LOG2F>, T, smem, nframes);
    hipLaunchKernelGGL((k_fft_bluestein<LOG2F>), dim3((unsigned)grid), dim3(T), smem, s, in, out, nframes, N, tw, hpos, chirp);
    RR_HIP(hipGetLastError());
}
void launch_fft_bluestein(int log2m, const cf* in, cf* out, long nframes, int N, const cf* tw, const cf* hpos,
                          const cf* chirp, hipStream_t s) {
    if (nframes <= 0) return;
    switch (log2m) {
    case 10: launch_bluestein_one<10>(in, out, nframes, N, tw, hpos, chirp, s); break;
    case 11: launch_bluestein_one<11>(in, out, nframes, N, tw, hpos, chirp, s); break;
    case 12: launch_bluestein_one<12>(in, out, nframes, N, tw, hpos, chirp, s); break;
    default: throw Error("fft_bluestein: unsupported tile size");
    }
}

void launch_fft_frames(int log2n, const cf* in, cf* out, long nframes, const cf* tw, const cf* tw4096, hipStream_t s) {
    if (nframes <= 0) return;
    if (log2n == 13 && tw4096) { launch_frames_split<2>(in, out, nframes, tw4096, tw, s); return; }
    if (log2n == 14 && tw4096) { launch_frames_split<4>(in, out, nframes, tw4096, tw, s); return; }
    switch (log2n) {
    case 10: launch_frames_one<10, 0>(in, out, nframes, tw, s); return;
    case 11: launch_frames_one<11, 0>(in, out, nframes, tw, s); return;
    case 12: launch_frames_one<12, 0>(in, out, nframes, tw, s); return;
    case 13: launch_frames_one<13, 3>(in, out, nframes, tw, s); return;
    case 14: launch_frames_one<14, 3>(in, out, nframes, tw, s); return;
    default: break;
    }
    if (log2n < 1 || log2n > 9) throw Error("fft_frames: size must be a power of two in 2..16384");
    const long per_wg = 512 >> log2n;
    long grid = (nframes + per_wg - 1) / per_wg;
    const long cap = (long)device_cu_count() * 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(k_fft_small, dim3((unsigned)grid), dim3(256), 0, s, in, out, nframes, log2n, tw);
    RR_HIP(hipGetLastError());
}

// ---- fused FftFilter -> RationalResampler -> QuadratureDemod ------------------------------------------

struct FmArgs {
    long A;          // filtered samples emitted before this call (global index of y at tile-space 0)
    long n_y;        // filtered samples of this call: y[A .. A + n_y)
    long r_lo, r_hi; // resampled samples whose source lies in this call: r[r_lo .. r_hi)
    long o_base;     // demod outputs emitted before this call (o index of out[0])
    long I, D;       // reduced interp / deci: r[m] = y[floor(m*D/I)]
    int G;           // max distance between the sources of r[m] and r[m+1]
    float gain;
    int mode;        // RR_ATAN2_*
    CarryOut carry;  // the block's new carry prefix, written by this launch (common.hpp)
};

// Tile advance of the fused chains.  For interp 1 a tile owns exactly Sp / D demodulated samples when D | Sp; rounding
// Sp down to a multiple of lanes * D makes that a whole number of epilogue rounds (463 taps, 1:6, 2048-point tiles:
// 263 outputs = 3 rounds of 128 lanes -> 256 = 2 rounds) for a few percent more tiles.  Same formula on host and device.
__host__ __device__ inline long fm_advance(long smax, long I, long D, int lanes) {
    if (I == 1 && smax > 0) {
        const long q = (long)lanes * D, al = smax / q * q;
        if (al > 0 && al * 10 >= smax * 9) return al;
    }
    return smax;
}

// Sources floor(u*D/I) of the resampled samples r[u], u = u0, u0 + T, u0 + 2T, ...: one 64-bit division
// per thread and TILE instead of two per output (a software division is ~80 VALU instructions; the
// epilogue used to cost more than the inverse FFT).
struct SrcWalk {
    long q, r;                                           // u*D = q*I + r, 0 <= r < I
    __device__ __forceinline__ void init(long u, long I, long D) {
        const long p = u * D;                            // u = -1 occurs (the lower partner of r[0]): floor division
        if (I == 1) { q = p; r = 0; }
        else { q = p / I; r = p - q * I; if (r < 0) { r += I; q--; } }
    }
    __device__ __forceinline__ void step(long qs, long rs, long I) {
        q += qs; r += rs;
        if (r >= I) { q++; r -= I; }
    }
};

// Resample + demodulate the part of one filtered tile (natural order in LDS, `at(p)` = tile position p) that the tile owns:
// the resampled samples u = u_lo + t, u_lo + t + T, ... < u_hi whose source lies in the tile's own Sp outputs
// (rational_resampler.rs:183-198, quadrature_demod.rs:65-109).  Round 3: everything per OUTPUT is 32-bit and tile-local —
// positions pu / pl of r[u] / r[u - 1] advance by (qs, rs) with one conditional carry, the lower partner's start follows
// from the upper one's ((u - 1) D = u D - D: no second division), the three special samples of a call (r[0] has no
// partner, the first pair takes its lower sample from the previous call, the last r is carried) are tested on k == 0 /
// k == n - 1 only.  The first version walked 64-bit stream indices with two 64-bit divisions per thread and tile and cost
// ~150 instructions per output on the 8192-point tiles (4.4 outputs per thread and tile at 25:128): 19 % of the rtl_fm
// front end (tools/rtl_fm_ablate.sh: 0.142 ms with, 0.115 ms without it).
template <int T> struct TileWalk {
    int n = 0;                       // outputs of this thread in the tile
    int pu, pl, qs;                  // tile positions of r[u] / r[u - 1]; step of T outputs
    unsigned ru, rl, rs, I32;        // u D mod I, (u - 1) D mod I; remainder step
    bool first_here, carry_here, no_partner;
    long o_off;                      // index of this thread's first output in the call's output window
    __device__ __forceinline__ void init(long tile, long Sp, long ys, int first, const FmArgs& a, int t) {
        const long y_lo = tile * Sp, y_hi = min((tile + 1) * Sp, a.n_y);
        long u_lo = ((a.A + y_lo) * a.I + a.D - 1) / a.D;             // u >= ceil((A + y) I / D): wave-uniform
        long u_hi = ((a.A + y_hi) * a.I + a.D - 1) / a.D;
        if (u_lo < a.r_lo) u_lo = a.r_lo;
        if (u_hi > a.r_hi) u_hi = a.r_hi;
        const long u0 = u_lo + t;
        n = u0 < u_hi ? (int)((u_hi - u0 + T - 1) / T) : 0;           // (T is a power of two)
        if (n == 0) return;
        SrcWalk wu;
        wu.init(u0, a.I, a.D);
        I32 = (unsigned)a.I;                                          // (I <= 2^31: FmChain's constructor)
        const int qd = (int)(a.D / a.I);
        const unsigned rd = (unsigned)(a.D % a.I);
        qs = (int)(((long)T * a.D) / a.I);
        rs = (unsigned)(((long)T * a.D) % a.I);
        pu = (int)(wu.q - a.A - ys) + first;
        ru = (unsigned)wu.r;
        pl = pu - qd - (ru < rd ? 1 : 0);
        rl = ru < rd ? ru + (I32 - rd) : ru - rd;
        first_here = u0 == a.r_lo;                                    // this thread starts on the call's first sample
        no_partner = first_here && u0 == 0;                           // ... which is r[0] of the stream
        carry_here = u0 + (long)(n - 1) * T == a.r_hi - 1;            // ... ends on the call's last one
        o_off = (u0 - 1) - a.o_base;
    }
    template <class AT>
    __device__ __forceinline__ void run(AT at, const FmArgs& a, float* __restrict__ out, const cf* __restrict__ last_in,
                                        cf* __restrict__ last_out) const {
        int pu_ = pu, pl_ = pl;
        unsigned ru_ = ru, rl_ = rl;
        float* o = out + o_off;
        for (int k = 0; k < n; k++, o += T) {
            const creg xu = at(pu_);
            if (carry_here && k == n - 1) last_out[0] = from_reg(xu); // carry for the next call
            if (!(no_partner && k == 0)) {                            // r[0] has no lower partner
                const creg xl = (first_here && k == 0) ? to_reg(last_in[0]) : at(pl_);   // lower sample from the previous call
                // conj(xl) * xu in num-complex order, un-contracted (quadrature_demod.rs:72)
                const float na = -xl.y;
                const float re = sub_rn(mul_rn(xl.x, xu.x), mul_rn(na, xu.y));
                const float im = add_rn(mul_rn(xl.x, xu.y), mul_rn(na, xu.x));
                const float ang = a.mode == 0 ? atan2_poly(im, re) : fmc_atan2(im, re);
                *o = mul_rn(a.gain, ang);
            }
            pu_ += qs; ru_ += rs;
            if (ru_ >= I32) { ru_ -= I32; pu_++; }
            pl_ += qs; rl_ += rs;
            if (rl_ >= I32) { rl_ -= I32; pl_++; }
        }
    }
};
template <int T, class AT>
__device__ __forceinline__ void fm_epilogue(AT at, long tile, long Sp, long ys, int first, const FmArgs& a, int t,
                                            float* __restrict__ out, const cf* __restrict__ last_in, cf* __restrict__ last_out) {
    TileWalk<T> w;
    w.init(tile, Sp, ys, first, a, t);
    w.run(at, a, out, last_in, last_out);
}

// ---- FftFilterFloat -> RationalResampler -> MultiplyConst fused (the rtl_fm audio stage, examples/rtl_fm.rs:398-418) ----
// The real-stream tile above (two overlap-save segments per Complex tile); instead of storing the filtered segments
// the tile parks them in LDS in stream order and every thread picks resampled samples out[m] = scale * y[floor(m D / I)]
// (rational_resampler.rs:183-198; multiply_const.rs:6-23: one f32 multiply), lane-consecutive in m: 4 B in and
// 4 I / D B out per sample instead of 8 + (4 + 4 I / D) + 8 I / D through three kernels.
struct AudioArgs {
    long A;            // filtered samples emitted before this call (stream index of this call's y[0])
    long n_y;          // filtered samples of this call
    long r_lo, r_hi;   // resampled samples whose source lies in this call
    long I, D;         // reduced interp / deci
    float scale;
    CarryOut carry;    // the block's new carry prefix, written by this launch
};
template <int LOG2F>
__global__ __launch_bounds__((KCfg<LOG2F, 0>::T), (KCfg<LOG2F, 0>::WAVES_PER_SIMD))
void k_audio_chain(VSrc<float> src, float* __restrict__ out, int L, long ntiles, const cf* __restrict__ tw,
                   const cf* __restrict__ hpos, AudioArgs a) {
    carry_store<float>(src, a.carry);
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    float* ybuf = reinterpret_cast<float*>(smem_raw);    // 2 S filtered samples in stream order (after the transform)
    const int t = threadIdx.x;
    const long S = F - L + 1;
    const int first = L - 1;
    const long qs = ((long)T * a.D) / a.I, rs = ((long)T * a.D) % a.I;      // SrcWalk step of T outputs
    TileXform<LOG2F, 0> X;
    X.init(t, tw, hpos);
    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long va = 2 * it.tile * S, vb = va + S;       // virtual index of position 0 of the two segments
        creg v[16];
        if (va >= src.plen && vb - src.plen + F <= src.in_len) {
            const float* pa = src.in + (va - src.plen) + t;
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = mk(pa[n * T], pa[S + n * T]);
        } else {
            stage_pair_slow<T>(lds, src, va, vb, t);
            tile_sync<T>();
            lds_load<LOG2F, 0>(v, t, lds);
        }
        RR_PHASE();
        X.run(v, lds, 0, nullptr);
        tile_sync<T>();                                      // the last exchange is read everywhere
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int p = n * T + t - first;                 // y_rel[va + p] (segment a), y_rel[vb + p] (segment b)
            if (p >= 0) { ybuf[p] = v[n].x; ybuf[S + p] = v[n].y; }
        }
        tile_sync<T>();
        // resampled samples with their source in [va, min(va + 2 S, n_y)) (relative to A)
        const long y_lo = va, y_hi = min(va + 2 * S, a.n_y);
        long u_lo = ((a.A + y_lo) * a.I + a.D - 1) / a.D;
        long u_hi = ((a.A + y_hi) * a.I + a.D - 1) / a.D;
        if (u_lo < a.r_lo) u_lo = a.r_lo;
        if (u_hi > a.r_hi) u_hi = a.r_hi;
        SrcWalk wu;
        wu.init(u_lo + t, a.I, a.D);
        for (long u = u_lo + t; u < u_hi; u += T, wu.step(qs, rs, a.I))
            out[u - a.r_lo] = mul_rn(a.scale, ybuf[wu.q - a.A - va]);
        tile_sync<T>();                                      // before the next tile's exchanges overwrite ybuf
    }
}
template <int LOG2F>
static void launch_audio_one(VSrc<float> src, float* out, int L, const cf* tw, const cf* hpos, const AudioChainArgs& h, hipStream_t s) {
    constexpr int F = 1 << LOG2F, T = F / 16;
    const long S = F - L + 1;
    if (h.n_y <= 0) { launch_carry(src, h.carry, s); return; }
    const long ntiles = (h.n_y + 2 * S - 1) / (2 * S);
    AudioArgs a{h.A, h.n_y, h.r_lo, h.r_hi, h.I, h.D, h.scale, h.carry};
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_audio_chain<LOG2F>, T, smem, ntiles);
    hipLaunchKernelGGL((k_audio_chain<LOG2F>), dim3((unsigned)grid), dim3(T), smem, s, src, out, L, ntiles, tw, hpos, a);
    RR_HIP(hipGetLastError());
}
void launch_audio_chain(int log2f, VSrc<float> src, float* out, int L, const cf* tw, const cf* hpos, const AudioChainArgs& a,
                        hipStream_t s) {
    switch (log2f) {
    case 10: launch_audio_one<10>(src, out, L, tw, hpos, a, s); break;
    case 11: launch_audio_one<11>(src, out, L, tw, hpos, a, s); break;
    case 12: launch_audio_one<12>(src, out, L, tw, hpos, a, s); break;
    default: throw Error("audio_chain: unsupported tile size");
    }
}

// Tile j transforms y[A + j*Sp - G .. + S') and owns every demod output o[u-1] whose UPPER
// sample r[u] has its source in [A + j*Sp, A + (j+1)*Sp), Sp = S' - G; the lower sample r[u-1]
// then lies in the same tile — or is the last r of the previous call (`last_r`).
template <int LOG2F, int VAR, class SRC>
__global__ __launch_bounds__((KCfg<LOG2F, VAR>::T), (KCfg<LOG2F, VAR>::WAVES_PER_SIMD))
void k_fm_chain(SRC src, float* __restrict__ out, int L, long ntiles, const cf* __restrict__ tw,
                const cf* __restrict__ hpos, FmArgs a, const cf* __restrict__ last_r_in,
                cf* __restrict__ last_r_out) {
    carry_store<cf>(src, a.carry);
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const int first = L - 1;
    const long Sp = fm_advance((F - L + 1) - a.G, a.I, a.D, T);
    TileXform<LOG2F, VAR> X;
    X.init(t, tw, hpos);

    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        // tile-space position p holds y[ys + p - first] with ys = tile*Sp - G (relative to A);
        // virtual input index of position 0 = ys (the prefix starts L-1 samples before y[A])
        const long ys = tile * Sp - a.G;
        creg v[16];
        load_tile16<LOG2F>(v, src, ys, t, lds);
        RR_PHASE();
        X.run(v, lds, 0);
        lds_store<LOG2F, 0>(v, t, lds);          // natural order: lds[pad(p)] = tile position p
        tile_sync<T>();

        // upper samples u with source in [tile*Sp, min((tile+1)*Sp, n_y))  (relative to A)
        fm_epilogue<T>([&](int p) -> creg { return lds[lds_pad(p)]; }, tile, Sp, ys, first, a, t, out, last_r_in, last_r_out);
        tile_sync<T>();        // epilogue reads done before the next tile's first exchange
    }
}

// ---- N FM channels on one shared IQ source (BASELINE configs[3]) --------------------------------------------
// Every channel of a channelised receiver filters the SAME input with its own band-pass taps.  The
// reference needs a Tee tree and one FftFilter per channel (src/tee.rs:10-24); here the forward
// FFT of a tile is computed once, parked in LDS, and each channel only pays H_c * X, the inverse
// FFT and the resample/demod epilogue.  On-GPU fan-out is free: the input tile is read from HBM
// once per workgroup, not once per channel.
template <int LOG2F, class SRC>
__global__ __launch_bounds__((1 << (LOG2F - 4)), 2)
void k_fm_multi(SRC src, float* __restrict__ out, long out_stride, int L, long ntiles,
                const cf* __restrict__ tw, const cf* __restrict__ hpos_all, int nchan, FmArgs a,
                const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out) {
    carry_store<cf>(src, a.carry);
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    constexpr int NP = Plan<LOG2F>::NP;
    static_assert(NP == 3, "multi-channel kernel: 3-pass plans");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* ldsX = lds + lds_elems(F);                     // the tile's spectrum, pass-2 layout (own slots)
    const int t = threadIdx.x;
    const int first = L - 1;
    const long Sp = fm_advance((F - L + 1) - a.G, a.I, a.D, T);
    const long qs = ((long)T * a.D) / a.I, rs = ((long)T * a.D) % a.I;      // SrcWalk step of T outputs
    TileXform<LOG2F, 0> X;
    X.init_no_h(t, tw);

    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        const long ys = tile * Sp - a.G;
        {
            creg v[16];
            load_tile16<LOG2F>(v, src, ys, t, lds);
            RR_PHASE();
            X.forward(v, lds);
            lds_store<LOG2F, NP - 1>(v, t, ldsX);
        }
        const long y_lo = tile * Sp, y_hi = min((tile + 1) * Sp, a.n_y);
        long u_lo = ((a.A + y_lo) * a.I + a.D - 1) / a.D;
        long u_hi = ((a.A + y_hi) * a.I + a.D - 1) / a.D;
        if (u_lo < a.r_lo) u_lo = a.r_lo;
        if (u_hi > a.r_hi) u_hi = a.r_hi;
        SrcWalk wu0, wl0;                                // the same sources for every channel
        wu0.init(u_lo + t, a.I, a.D);
        wl0.init(u_lo + t - 1, a.I, a.D);
        // channel c+1's frequency response is fetched (L2) while channel c is inverse-transformed
        creg h[16];
        load_h<LOG2F, NP - 1>(h, t, hpos_all);
        for (int c = 0; c < nchan; c++) {
            RR_PHASE();
            creg w[16];
            lds_load<LOG2F, NP - 1>(w, t, ldsX);
            apply_h(w, h);
            if (c + 1 < nchan) load_h<LOG2F, NP - 1>(h, t, hpos_all + (long)(c + 1) * F);
            X.inverse(w, lds);
            lds_store<LOG2F, 0>(w, t, lds);
            tile_sync<T>();
            float* oc = out + (long)c * out_stride;
            SrcWalk wu = wu0, wl = wl0;
            for (long u = u_lo + t; u < u_hi; u += T, wu.step(qs, rs, a.I), wl.step(qs, rs, a.I)) {
                const long gu = wu.q - a.A;
                const creg ru = lds[lds_pad((int)(gu - ys) + first)];
                if (u == a.r_hi - 1) last_r_out[c] = from_reg(ru);
                if (u != 0) {
                    creg rl;
                    if (u == a.r_lo) rl = to_reg(last_r_in[c]);
                    else rl = lds[lds_pad((int)(wl.q - a.A - ys) + first)];
                    const float na = -rl.y;
                    const float re = sub_rn(mul_rn(rl.x, ru.x), mul_rn(na, ru.y));
                    const float im = add_rn(mul_rn(rl.x, ru.y), mul_rn(na, ru.x));
                    const float ang = a.mode == 0 ? atan2_poly(im, re) : fmc_atan2(im, re);
                    oc[(u - 1) - a.o_base] = mul_rn(a.gain, ang);
                }
            }
            tile_sync<T>();
        }
    }
}

// ---- the same for an even integer decimation (interp 1): half-size inverse transforms ------------------------
// The resampler then keeps only samples y[u D] (rational_resampler.rs:183-198 with interp 1), all of one parity, so
// with the tile start shifted by at most one sample they sit at EVEN tile positions p = 2 n', and
//     y[2 n'] = sum_{k < F/2} (Y[k] + Y[k + F/2]) w_(F/2)^(-n' k)
// (decimation in time = aliasing in frequency).  k + F/2 is k3 + D/2 in the last digit of the tile plan, i.e. the
// partner sits in the same thread's group: after H_c X the spectrum is folded by D/2 thread-local additions and the
// inverse is an F/2-point transform.  For F = 2048 that is a 1024-point tile of ONE wave: each of the two waves of
// the workgroup takes every other channel and runs product, fold, inverse and the resample / demod epilogue without
// a single barrier (the full-size version needs four per channel), on half the arithmetic.
template <int LOG2F, class SRC>
__global__ __launch_bounds__((1 << (LOG2F - 4)), 2)
void k_fm_multi_half(SRC src, float* __restrict__ out, long out_stride, int L, long ntiles, long Sp,
                     const cf* __restrict__ tw, const cf* __restrict__ tw_half, const cf* __restrict__ hpos_all,
                     int nchan, FmArgs a, const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out) {
    carry_store<cf>(src, a.carry);
    constexpr int F = 1 << LOG2F, T = F / 16, LH = LOG2F - 1, FH = F / 2, TH = T / 2;
    constexpr int NP = Plan<LOG2F>::NP;
    constexpr int D3 = F / 256, DH = D3 / 2;             // last radix of the full / the half plan
    constexpr int XROW = 256 + 8;                        // parked spectrum: D3 rows of 256 groups
    static_assert(NP == 3, "3-pass plans");
    static_assert(TH == 64, "one wave per half-size tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* ldsX = lds + lds_elems(F);                     // the tile's spectrum, pass-2 layout
    const int t = threadIdx.x, hg = t / TH, th = t % TH;
    creg* ldsH = lds + hg * lds_elems(FH);               // this wave's exchange / epilogue area (free once the forward is done)
    creg* tw64 = ldsX + lds_elems(F);                    // w_64^j, j < 64: the half plan's pass-1 twiddles (see below)
    const int first = L - 1;
    TileXform<LOG2F, 3> XF;                              // forward: once per tile, tables re-read (L1)
    XF.init_no_h(t, tw);
    // inverse: once per channel and tile.  Its pass-0 twiddles w_(F/2)^(k th) stay in registers; the pass-1 set
    // w_64^(k (th % 4)) is read from a 64-entry LDS table per use: with both sets resident the kernel spills
    // ~120 B/lane, and every scratch access waits on the whole in-order vmcnt queue (the channel's H loads).
    creg tw0h[15];
    load_twiddles<LH, 0>(tw0h, th, tw_half);
    static_assert(PassGeom<LH, 1>::R * PassGeom<LH, 1>::P == 64, "pass-1 twiddles of the half plan are powers of w_64");
    if (t < 64) tw64[t] = to_reg(tw_half[t * (FH / 64)]);
    const int lo1 = PassGeom<LH, 1>::lo(th);

    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        const long base = tile * Sp - a.G;
        const long ys = base - ((a.A + base - first) & 1);   // needed samples (even global index) at even tile positions
        {
            creg v[16];
            load_tile16<LOG2F>(v, src, ys, t, lds);
            RR_PHASE();
            XF.forward(v, lds);
            // parked k3-major (rows of 256 groups + 8 slots of padding): both this store and the waves' reads below are
            // lane-consecutive in the group index, i.e. conflict-free
#pragma unroll
            for (int u = 0; u < 16 / D3; u++)
#pragma unroll
                for (int k = 0; k < D3; k++) ldsX[k * XROW + t + T * u] = v[u * D3 + k];
        }
        tile_sync<T>();                                  // the spectrum is read by both waves
        const long y_lo = tile * Sp, y_hi = min((tile + 1) * Sp, a.n_y);
        long u_lo = (a.A + y_lo + a.D - 1) / a.D;        // interp 1: r[u] = y[u D]
        long u_hi = (a.A + y_hi + a.D - 1) / a.D;
        if (u_lo < a.r_lo) u_lo = a.r_lo;
        if (u_hi > a.r_hi) u_hi = a.r_hi;
        const int hD = (int)(a.D >> 1);
#pragma unroll 1
        for (int c = hg; c < nchan; c += 2) {
            const creg* hc = reinterpret_cast<const creg*>(hpos_all) + (long)c * F;
            creg w[16];
            {
                // the channel's response at this thread's 32 bins, half of them in flight at a time
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    creg h[16];
#pragma unroll
                    for (int uu = 0; uu < 8 / DH; uu++)
#pragma unroll
                        for (int k = 0; k < D3; k++) h[uu * D3 + k] = hc[(th + TH * (half * (8 / DH) + uu)) * D3 + k];
                    RR_PHASE();
#pragma unroll
                    for (int uu = 0; uu < 8 / DH; uu++) {
                        const int u = half * (8 / DH) + uu;
                        const int g = th + TH * u;           // group 16 k1 + k2, the same in both plans
                        const creg* xx = ldsX + g;
#pragma unroll
                        for (int k = 0; k < DH; k++)
                            w[u * DH + k] = cadd(cmul(xx[k * XROW], h[uu * D3 + k]), cmul(xx[(k + DH) * XROW], h[uu * D3 + k + DH]));
                    }
                    RR_PHASE();
                }
            }
            {   // TileXform<LH>::inverse for one wave (no barriers), pass-1 twiddles from the LDS table
                creg twl[15];
                inv_pass<LH, 2>(w, twl);                 // (P == 1: no twiddles)
                RR_PHASE();
                lds_store<LH, 2>(w, th, ldsH);
                asm volatile("" ::: "memory");
                lds_load<LH, 1>(w, th, ldsH);
#pragma unroll
                for (int k = 1; k < 16; k++) twl[k - 1] = tw64[k * lo1];
                RR_PHASE();
                inv_pass<LH, 1>(w, twl);
                RR_PHASE();
                lds_store<LH, 1>(w, th, ldsH);
                asm volatile("" ::: "memory");
                lds_load<LH, 0>(w, th, ldsH);
                RR_PHASE();
                inv_pass<LH, 0>(w, tw0h);
                RR_PHASE();
            }
            lds_store<LH, 0>(w, th, ldsH);               // natural order: ldsH[pad(n')] = y[2 n'] of the tile
            asm volatile("" ::: "memory");
            float* oc = out + (long)c * out_stride;
            for (long u = u_lo + th; u < u_hi; u += TH) {
                const int p2 = (int)((u * a.D - a.A - ys + first) >> 1);      // half the (even) tile position
                const creg ru = ldsH[lds_pad(p2)];
                if (u == a.r_hi - 1) last_r_out[c] = from_reg(ru);
                if (u != 0) {
                    creg rl;
                    if (u == a.r_lo) rl = to_reg(last_r_in[c]);
                    else rl = ldsH[lds_pad(p2 - hD)];
                    const float na = -rl.y;
                    const float re = sub_rn(mul_rn(rl.x, ru.x), mul_rn(na, ru.y));
                    const float im = add_rn(mul_rn(rl.x, ru.y), mul_rn(na, ru.x));
                    const float ang = a.mode == 0 ? atan2_poly(im, re) : fmc_atan2(im, re);
                    oc[(u - 1) - a.o_base] = mul_rn(a.gain, ang);
                }
            }
            asm volatile("" ::: "memory");               // (one wave: its LDS operations execute in order)
        }
        tile_sync<T>();                                  // both waves done before the next forward reuses the areas
    }
}

// ---- the single fused chain with the half-size inverse (interp 1, even decimation, 2048-point tiles) ----------
// Two tiles per workgroup iteration: all T threads run forward transform, H product and the fold of each (the folded
// spectrum of a tile is F/2 values, parked in LDS), then each of the two waves finishes ONE of the tiles — F/2-point
// inverse and the resample / demod epilogue — without barriers.  See k_fm_multi_half for the arithmetic.
// Registers: the pass-0 twiddles of both plans stay resident (60 VGPRs); both pass-1 sets (powers of w_128 and w_64)
// come from LDS tables, H is re-read per tile (in flight during the transform), and the thread index is made opaque
// per tile so that the tile body's LDS / table addresses are not hoisted into persistent registers.  A first version
// without these spilled 130-350 B/lane and ran 1.7x SLOWER than k_fm_chain (a spill waits on the whole vmcnt queue).
template <int LOG2F, class SRC>
__global__ __launch_bounds__((1 << (LOG2F - 4)), 2)
void k_fm_chain_half(SRC src, float* __restrict__ out, int L, long ntiles, long Sp, const cf* __restrict__ tw,
                     const cf* __restrict__ tw_half, const cf* __restrict__ hpos, FmArgs a,
                     const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out) {
    carry_store<cf>(src, a.carry);
    constexpr int F = 1 << LOG2F, T = F / 16, LH = LOG2F - 1, FH = F / 2, TH = T / 2;
    constexpr int NP = Plan<LOG2F>::NP;
    constexpr int D3 = F / 256, DH = D3 / 2, U = 16 / D3;
    constexpr int YROW = 256 + 8;                        // parked folded spectrum: DH rows of 256 groups (DH * YROW <= lds_elems(FH))
    constexpr int N1 = PassGeom<LOG2F, 1>::R * PassGeom<LOG2F, 1>::P;      // 128: pass-1 twiddles of the full plan
    constexpr int N1H = PassGeom<LH, 1>::R * PassGeom<LH, 1>::P;          // 64: ... of the half plan
    static_assert(NP == 3 && TH == 64, "one wave per half-size tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);       // forward exchanges; afterwards the two waves' own areas
    creg* ldsY = lds + lds_elems(F);                     // folded spectra of the two tiles, half-plan pass-2 layout
    creg* tab1 = ldsY + 2 * lds_elems(FH);               // w_128^j
    creg* tab1h = tab1 + N1;                             // w_64^j
    const int t_ = threadIdx.x, hg = t_ / TH, th_ = t_ % TH;
    const int t = t_, th = th_;
    const int first = L - 1;
    creg tw0[15], tw0h[15];
    load_twiddles<LOG2F, 0>(tw0, t, tw);
    load_twiddles<LH, 0>(tw0h, th, tw_half);
    for (int j = t; j < N1; j += T) tab1[j] = to_reg(tw[j * (F / N1)]);
    for (int j = t; j < N1H; j += T) tab1h[j] = to_reg(tw_half[j * (FH / N1H)]);
    const int hD = (int)(a.D >> 1);
    const long npairs = (ntiles + 1) / 2;
    tile_sync<T>();

    for (TileIter it(npairs); it.tile < it.end; it.tile += it.step) {
        long ys_mine = 0;
#pragma unroll 1
        for (int b = 0; b < 2; b++) {
            const long tile = 2 * it.tile + b;
            const long base = tile * Sp - a.G;
            const long ys = base - ((a.A + base - first) & 1);
            if (b == hg) ys_mine = ys;
            if (tile >= ntiles) break;                   // (workgroup-uniform)
            int t = t_;                                  // opaque per tile: keeps the ~60 LDS / table addresses of the tile body
            asm volatile("" : "+v"(t));                  // from being hoisted out of the loop into persistent VGPRs
            creg v[16], twl[15], hreg[16];
            load_tile16<LOG2F>(v, src, ys, t, lds);
            load_h<LOG2F, NP - 1>(hreg, t, hpos);            // in flight during the transform
            RR_PHASE();
            fwd_pass<LOG2F, 0>(v, tw0);
            RR_PHASE();
            lds_store<LOG2F, 0>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 1>(v, t, lds);
#pragma unroll
            for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<LOG2F, 1>::lo(t)];
            RR_PHASE();
            fwd_pass<LOG2F, 1>(v, twl);
            RR_PHASE();
            lds_store<LOG2F, 1>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 2>(v, t, lds);
            RR_PHASE();
            fwd_pass<LOG2F, 2>(v, twl);                  // (P == 1: no twiddles)
            tile_sync<T>();                              // the next forward's first exchange overwrites slots read in this one's last
            creg* py = ldsY + b * lds_elems(FH);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int g = t + T * u;                 // group 16 k1 + k2; its folded values sit at g * DH + k in the half plan
#pragma unroll
                for (int k = 0; k < DH; k++)         // parked k3-major: lane-consecutive (conflict-free) here and when read back
                    py[k * YROW + g] = cadd(cmul(v[u * D3 + k], hreg[u * D3 + k]), cmul(v[u * D3 + k + DH], hreg[u * D3 + k + DH]));
            }
            RR_PHASE();
        }
        tile_sync<T>();
        const long tile = 2 * it.tile + hg;
        if (tile < ntiles) {
            const long ys = ys_mine;
            int th = th_;
            asm volatile("" : "+v"(th));
            creg* ldsH = lds + hg * lds_elems(FH);
            creg w[16], twl[15];
            {
                const creg* py = ldsY + hg * lds_elems(FH) + th;
#pragma unroll
                for (int u = 0; u < 16 / DH; u++)
#pragma unroll
                    for (int k = 0; k < DH; k++) w[u * DH + k] = py[k * YROW + TH * u];
            }
            inv_pass<LH, 2>(w, twl);                     // (P == 1: no twiddles)
            RR_PHASE();
            lds_store<LH, 2>(w, th, ldsH);
            asm volatile("" ::: "memory");
            lds_load<LH, 1>(w, th, ldsH);
#pragma unroll
            for (int k = 1; k < 16; k++) twl[k - 1] = tab1h[k * PassGeom<LH, 1>::lo(th)];
            RR_PHASE();
            inv_pass<LH, 1>(w, twl);
            RR_PHASE();
            lds_store<LH, 1>(w, th, ldsH);
            asm volatile("" ::: "memory");
            lds_load<LH, 0>(w, th, ldsH);
            RR_PHASE();
            inv_pass<LH, 0>(w, tw0h);
            RR_PHASE();
            lds_store<LH, 0>(w, th, ldsH);               // natural order: ldsH[pad(n')] = y[2 n'] of the tile
            asm volatile("" ::: "memory");
            const long y_lo = tile * Sp, y_hi = min((tile + 1) * Sp, a.n_y);
            long u_lo = (a.A + y_lo + a.D - 1) / a.D;
            long u_hi = (a.A + y_hi + a.D - 1) / a.D;
            if (u_lo < a.r_lo) u_lo = a.r_lo;
            if (u_hi > a.r_hi) u_hi = a.r_hi;
            for (long u = u_lo + th; u < u_hi; u += TH) {
                const int p2 = (int)((u * a.D - a.A - ys + first) >> 1);
                const creg ru = ldsH[lds_pad(p2)];
                if (u == a.r_hi - 1) last_r_out[0] = from_reg(ru);
                if (u != 0) {
                    creg rl;
                    if (u == a.r_lo) rl = to_reg(last_r_in[0]);
                    else rl = ldsH[lds_pad(p2 - hD)];
                    const float na = -rl.y;
                    const float re = sub_rn(mul_rn(rl.x, ru.x), mul_rn(na, ru.y));
                    const float im = add_rn(mul_rn(rl.x, ru.y), mul_rn(na, ru.x));
                    const float ang = a.mode == 0 ? atan2_poly(im, re) : fmc_atan2(im, re);
                    out[(u - 1) - a.o_base] = mul_rn(a.gain, ang);
                }
            }
        }
        tile_sync<T>();
    }
}

// ---- decimating FirFilter<Complex> with an even decimation on the half-size inverse -------------------------------
// out[m] = y[m d], d = 2 d2: the kept samples have even full-rate indices, so (tile start shifted by L - 1's parity,
// even tile advance) they sit at even tile positions and the inverse is the folded F/2-point transform of
// k_fm_chain_half — two tiles per iteration, each wave finishes one and stores its kept samples (every d2-th of the
// half-rate sequence; lane-consecutive for d = 2).  Decimations 4 / 8 / 16 have k_fftfilt_prune.
template <int LOG2F>
__global__ __launch_bounds__((1 << (LOG2F - 4)), 2)
void k_fftfilt_half(NanFixCtx nfx, VSrc<cf> src, cf* __restrict__ out, long n_out, int L, int d2, long ntiles, long S,
                    const cf* __restrict__ tw, const cf* __restrict__ tw_half, const cf* __restrict__ hpos) {
    (void)nfx;
    constexpr int F = 1 << LOG2F, T = F / 16, LH = LOG2F - 1, FH = F / 2, TH = T / 2;
    nf_init();
    constexpr int NP = Plan<LOG2F>::NP;
    constexpr int D3 = F / 256, DH = D3 / 2, U = 16 / D3;
    constexpr int N1 = PassGeom<LOG2F, 1>::R * PassGeom<LOG2F, 1>::P;
    constexpr int N1H = PassGeom<LH, 1>::R * PassGeom<LH, 1>::P;
    static_assert(NP == 3 && TH == 64, "one wave per half-size tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* ldsY = lds + lds_elems(F);
    creg* tab1 = ldsY + 2 * lds_elems(FH);
    creg* tab1h = tab1 + N1;
    const int t_ = threadIdx.x, hg = t_ / TH, th_ = t_ % TH;
    const int first = L - 1;
    const int delta = first & 1;                         // S even: (tile S - delta - first) is even for every tile
    creg tw0[15], tw0h[15];
    load_twiddles<LOG2F, 0>(tw0, t_, tw);
    load_twiddles<LH, 0>(tw0h, th_, tw_half);
    for (int j = t_; j < N1; j += T) tab1[j] = to_reg(tw[j * (F / N1)]);
    for (int j = t_; j < N1H; j += T) tab1h[j] = to_reg(tw_half[j * (FH / N1H)]);
    const float inv_d2 = 1.0f / (float)d2;
    const long npairs = (ntiles + 1) / 2;
    creg* out_reg = reinterpret_cast<creg*>(out);
    tile_sync<T>();

    for (TileIter it(npairs); it.tile < it.end; it.tile += it.step) {
#pragma unroll 1
        for (int b = 0; b < 2; b++) {
            const long tile = 2 * it.tile + b;
            if (tile >= ntiles) break;                   // (workgroup-uniform)
            int t = t_;                                  // opaque per tile (see k_fm_chain_half)
            asm volatile("" : "+v"(t));
            creg v[16], twl[15], hreg[16];
            load_tile16<LOG2F>(v, src, tile * S - delta, t, lds);
            load_h<LOG2F, NP - 1>(hreg, t, hpos);        // in flight during the transform
            RR_PHASE();
            fwd_pass<LOG2F, 0>(v, tw0);
            RR_PHASE();
            lds_store<LOG2F, 0>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 1>(v, t, lds);
#pragma unroll
            for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<LOG2F, 1>::lo(t)];
            RR_PHASE();
            fwd_pass<LOG2F, 1>(v, twl);
            RR_PHASE();
            lds_store<LOG2F, 1>(v, t, lds);
            tile_sync<T>();
            lds_load<LOG2F, 2>(v, t, lds);
            RR_PHASE();
            fwd_pass<LOG2F, 2>(v, twl);
            tile_sync<T>();
            creg* py = ldsY + b * lds_elems(FH);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int g = t + T * u;
#pragma unroll
                for (int k = 0; k < DH; k++)
                    py[lds_pad(g * DH + k)] = cadd(cmul(v[u * D3 + k], hreg[u * D3 + k]), cmul(v[u * D3 + k + DH], hreg[u * D3 + k + DH]));
            }
            RR_PHASE();
        }
        tile_sync<T>();
        const long tile = 2 * it.tile + hg;
        if (tile < ntiles) {
            int th = th_;
            asm volatile("" : "+v"(th));
            creg* ldsH = lds + hg * lds_elems(FH);
            creg w[16], twl[15];
            lds_load<LH, 2>(w, th, ldsY + hg * lds_elems(FH));
            inv_pass<LH, 2>(w, twl);
            RR_PHASE();
            lds_store<LH, 2>(w, th, ldsH);
            asm volatile("" ::: "memory");
            lds_load<LH, 1>(w, th, ldsH);
#pragma unroll
            for (int k = 1; k < 16; k++) twl[k - 1] = tab1h[k * PassGeom<LH, 1>::lo(th)];
            RR_PHASE();
            inv_pass<LH, 1>(w, twl);
            RR_PHASE();
            lds_store<LH, 1>(w, th, ldsH);
            asm volatile("" ::: "memory");
            lds_load<LH, 0>(w, th, ldsH);
            RR_PHASE();
            inv_pass<LH, 0>(w, tw0h);                    // w[n] = y[2 (n 64 + th)] of the tile
            nf_mark(nf_bad(w[15].x));
            RR_PHASE();
            // half-rate index of tile position 2 n': hb + n', hb = (tile S - delta - first) / 2; this tile owns the
            // full-rate indices [tile S, (tile + 1) S), i.e. n' in [n_lo, n_hi)
            const long hb = (tile * S - delta - first) / 2;
            const int n_lo = (first + delta) / 2, n_hi = n_lo + (int)(S / 2);
            if (d2 == 1) {
                creg* po = out_reg + hb + th;
                const long room = n_out - hb - th;
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const int np = n * TH + th;
                    if (np >= n_lo && np < n_hi && n * TH < room) po[n * TH] = w[n];
                }
            } else {
                // keep (hb + n') % d2 == 0 -> out[(hb + n') / d2]   (exact f32 quotients as in k_fftfilt_deci)
                const long K = (-hb + d2 - 1) / d2 > 0 ? (-hb + d2 - 1) / d2 : 0;   // hb may be negative in tile 0
                const long gb = hb + K * d2, qb = gb / d2;
                const int rb = (int)(gb - qb * d2) + th;
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const int np = n * TH + th, x = rb + n * TH;
                    const int q = (int)(((float)x + 0.5f) * inv_d2);
                    const long m = qb - K + q;
                    if (q * d2 == x && np >= n_lo && np < n_hi && m < n_out) out_reg[m] = w[n];
                }
            }
        }
        tile_sync<T>();
    }
    nf_finish<cf, cf>();
}

// measurement builds (make TIMING=1): 32 s_memtime stamps of one tile (see RR_STAMP; the two-wave kernels of
// kernels_poly.hip use 16 per wave); nullptr otherwise
unsigned long long* fft_stamp_buffer() {
#ifdef RR_FFT_TIMING_BUILD
    static unsigned long long* p = nullptr;
    if (!p) {
        RR_HIP(hipMalloc(reinterpret_cast<void**>(&p), 32 * sizeof(unsigned long long)));
        RR_HIP(hipMemset(p, 0, 32 * sizeof(unsigned long long)));
    }
    return p;
#else
    return nullptr;
#endif
}
int fft_read_stamps(unsigned long long* host16) {
    unsigned long long* p = fft_stamp_buffer();
    if (!p) return 0;
    RR_HIP(hipDeviceSynchronize());
    RR_HIP(hipMemcpy(host16, p, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 32;
}

bool fftfilt_supported(int log2f) { return log2f >= 10 && log2f <= 14; }

int device_cu_count() {
    // per device: one process may drive several GPUs (rr_set_device), first calls may race
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    RR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = cus.find(dev);
    if (it == cus.end()) {
        int n = 0;
        RR_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        it = cus.emplace(dev, n).first;
    }
    return it->second;
}


template <int LOG2F, int VAR>
static void launch_one(VSrc<cf> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos, hipStream_t s, CarryOut carry, NanFix fx = {}, long tile_lo = 0, long tile_hi = -1) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const long S = F - L + 1;
    if (tile_hi < 0) tile_hi = (n_out + S - 1) / S;     // all tiles
    const long ntiles = tile_hi - tile_lo;
    if (ntiles <= 0) { launch_carry(src, carry, s); return; }
    const size_t smem = sizeof(cf) * lds_elems(F);
    const int ablate = 0;
    const NanFixCtx nfx = nanfix_ctx(fx, src, out, S, 1, n_out, ntiles, 0, tile_lo);
    if (fx.rb_work) {
        const long grid = grid_for_tiles(k_fftfilt_os<LOG2F, VAR, 2>, T, smem, ntiles);
        hipLaunchKernelGGL((k_fftfilt_os<LOG2F, VAR, 2>), dim3((unsigned)grid), dim3(T), smem, s, nfx, src, out, n_out, L,
                           ntiles, tw, hpos, ablate, fft_stamp_buffer(), tile_lo, carry);
    } else if (fx.rev) {
        const long grid = grid_for_tiles(k_fftfilt_os<LOG2F, VAR, 1>, T, smem, ntiles);
        hipLaunchKernelGGL((k_fftfilt_os<LOG2F, VAR, 1>), dim3((unsigned)grid), dim3(T), smem, s, nfx, src, out, n_out, L,
                           ntiles, tw, hpos, ablate, fft_stamp_buffer(), tile_lo, carry);
    } else {
        const long grid = grid_for_tiles(k_fftfilt_os<LOG2F, VAR, 0>, T, smem, ntiles);
        hipLaunchKernelGGL((k_fftfilt_os<LOG2F, VAR, 0>), dim3((unsigned)grid), dim3(T), smem, s, nfx, src, out, n_out, L,
                           ntiles, tw, hpos, ablate, fft_stamp_buffer(), tile_lo, carry);
    }
    RR_HIP(hipGetLastError());
}

void launch_fftfilt_os(int log2f, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw,
                       const cf* hpos, hipStream_t s, CarryOut carry, NanFix fx) {
    switch (log2f) {
    case 10: launch_one<10, 0>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    case 11: launch_one<11, 0>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    case 12: launch_one<12, 0>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    case 13: launch_one<13, 3>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    case 14: launch_one<14, 3>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    default: throw Error("fftfilt: unsupported tile size");
    }
}

template <int LOG2F, int VAR>
static void launch_deci_one(VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw, const cf* hpos, hipStream_t s, NanFix fx) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const long S = F - L + 1;
    if (n_out <= 0) return;
    const long n_full = (n_out - 1) * (long)d + 1;         // full-rate samples up to the last one kept
    const long ntiles = (n_full + S - 1) / S;
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fftfilt_deci<LOG2F, VAR>, T, smem, ntiles);
    // (nan_fix.hpp: the tile keeps the outputs m with m d in [tile S, tile S + S))
    hipLaunchKernelGGL((k_fftfilt_deci<LOG2F, VAR>), dim3((unsigned)grid), dim3(T), smem, s, nanfix_ctx(fx, src, out, S, d, n_out, ntiles),
                       src, out, n_out, L, d, ntiles, tw, hpos);
    RR_HIP(hipGetLastError());
}

void launch_fftfilt_deci(int log2f, VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw, const cf* hpos,
                         hipStream_t s, NanFix fx) {
    if (d < 1 || d > 4096) throw Error("fftfilt_deci: decimation out of range");
    switch (log2f) {
    case 10: launch_deci_one<10, 0>(src, out, n_out, L, d, tw, hpos, s, fx); break;
    case 11: launch_deci_one<11, 0>(src, out, n_out, L, d, tw, hpos, s, fx); break;
    case 12: launch_deci_one<12, 0>(src, out, n_out, L, d, tw, hpos, s, fx); break;
    default: throw Error("fftfilt_deci: unsupported tile size");
    }
}

template <int LOG2F, bool DECI>
static void launch_real_one(VSrc<float> src, float* out, long n_out, int L, int d, const cf* tw, const cf* hpos, hipStream_t s, CarryOut carry, NanFix fx) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const long S = F - L + 1;
    if (n_out <= 0) { launch_carry(src, carry, s); return; }
    const long n_full = DECI ? (n_out - 1) * (long)d + 1 : n_out;
    const long nseg = (n_full + S - 1) / S;
    const long ntiles = (nseg + 1) / 2;
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fftfilt_real<LOG2F, DECI>, T, smem, ntiles);
    // (nan_fix.hpp: a tile = the segments 2 k, 2 k + 1: full-rate outputs [2 k S, 2 k S + 2 S))
    hipLaunchKernelGGL((k_fftfilt_real<LOG2F, DECI>), dim3((unsigned)grid), dim3(T), smem, s, nanfix_ctx(fx, src, out, 2 * S, DECI ? d : 1, n_out, ntiles),
                       src, out, n_out, L, d, ntiles, tw, hpos, carry);
    RR_HIP(hipGetLastError());
}
template <int LOG2F>
static void launch_real_hilbert_one(VSrc<float> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos, hipStream_t s, CarryOut carry, NanFix fx) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    const long S = F - L + 1;
    if (n_out <= 0) { launch_carry(src, carry, s); return; }
    const long ntiles = ((n_out + S - 1) / S + 1) / 2;
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fftfilt_real<LOG2F, false, true>, T, smem, ntiles);
    hipLaunchKernelGGL((k_fftfilt_real<LOG2F, false, true>), dim3((unsigned)grid), dim3(T), smem, s, nanfix_ctx(fx, src, out, 2 * S, 1, n_out, ntiles),
                       src, reinterpret_cast<float*>(out), n_out, L, 1, ntiles, tw, hpos, carry);
    RR_HIP(hipGetLastError());
}
// Hilbert on real-stream tiles: out[k] = (xp[k + L/2], sum_j rev[j] xp[k + j]), k < n_out (L odd)
void launch_fftfilt_real_hilbert(int log2f, VSrc<float> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos, hipStream_t s,
                                 CarryOut carry, NanFix fx) {
    if (2L * ((1L << log2f) - L + 1) <= 0 || !(L & 1)) throw Error("fftfilt_real_hilbert: bad filter length for the tile");
    switch (log2f) {
    case 10: launch_real_hilbert_one<10>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    case 11: launch_real_hilbert_one<11>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    case 12: launch_real_hilbert_one<12>(src, out, n_out, L, tw, hpos, s, carry, fx); break;
    default: throw Error("fftfilt_real_hilbert: unsupported tile size");
    }
}
void launch_fftfilt_real(int log2f, VSrc<float> src, float* out, long n_out, int L, int d, const cf* tw, const cf* hpos,
                         hipStream_t s, CarryOut carry, NanFix fx) {
    if (d < 1 || d > 4096) throw Error("fftfilt_real: decimation out of range");
    if (2L * ((1L << log2f) - L + 1) <= 0) throw Error("fftfilt_real: tile too small");
    switch (log2f * 2 + (d > 1)) {
    case 20: launch_real_one<10, false>(src, out, n_out, L, d, tw, hpos, s, carry, fx); break;
    case 21: launch_real_one<10, true>(src, out, n_out, L, d, tw, hpos, s, carry, fx); break;
    case 22: launch_real_one<11, false>(src, out, n_out, L, d, tw, hpos, s, carry, fx); break;
    case 23: launch_real_one<11, true>(src, out, n_out, L, d, tw, hpos, s, carry, fx); break;
    case 24: launch_real_one<12, false>(src, out, n_out, L, d, tw, hpos, s, carry, fx); break;
    case 25: launch_real_one<12, true>(src, out, n_out, L, d, tw, hpos, s, carry, fx); break;
    default: throw Error("fftfilt_real: unsupported tile size");
    }
}

template <int LOG2F, int MODE>
static void launch_prune_one(VSrc<cf> csrc, VSrc<float> rsrc, cf* out, long n_final, int L, const cf* tw, const cf* hpos2,
                             const cf* hpos2b, const cf* twb, hipStream_t s, int sub, CarryOut carry = {}, NanFix fx = {}) {
    if (sub < 1 || sub > 64) throw Error("fftfilt_prune: sub-decimation out of range");
    const long n_out = n_final > 0 ? (n_final - 1) * (long)sub + 1 : 0;        // in units of the tile's own decimation
    constexpr int F = 1 << LOG2F, T = F / 16, D = F / 256;
    const long S = (F - L + 1) / D * D;                  // tiles advance by a multiple of D: one phase c for all tiles
    if (S <= 0) throw Error("fftfilt_prune: filter too long for the tile");
    if (n_out <= 0) { if (MODE == 0) launch_carry(csrc, carry, s); else launch_carry(rsrc, carry, s); return; }
    const long Sd = S / D;
    const long nseg = (n_out + Sd - 1) / Sd;
    const long ntiles = MODE ? (nseg + 1) / 2 : nseg;
    const size_t smem = sizeof(cf) * (lds_elems(F) + lds_elems(256 * D) + (MODE ? 16 * D : 0));
    constexpr int BT = MODE == 1 ? D / 2 : D;             // tiles per batch (MODE 1: two responses per tile)
    const long nbatch = (ntiles + BT - 1) / BT;
    const long grid = grid_for_tiles(k_fftfilt_prune<LOG2F, MODE>, T, smem, nbatch);
    // (nan_fix.hpp: a batch = BT tiles (real streams: 2 BT segments) of Sd kept samples y[D m] each; stored are the m = sub q, at out[q])
    const NanFixCtx nfx = MODE == 0 ? nanfix_ctx(fx, csrc, out, (long)BT * Sd, sub, n_final, nbatch)
                                    : nanfix_ctx(fx, rsrc, out, 2L * BT * Sd, sub, n_final, nbatch);
    hipLaunchKernelGGL((k_fftfilt_prune<LOG2F, MODE>), dim3((unsigned)grid), dim3(T), smem, s, nfx, csrc, rsrc, out, n_out, L, S,
                       ntiles, tw, hpos2, hpos2b, twb, carry, sub);
    RR_HIP(hipGetLastError());
}
int prune_log2f_for_deci(int d) { return d == 4 ? 10 : d == 8 ? 11 : d == 16 ? 12 : 0; }
// d = D * sub with D the largest of 16 / 8 / 4 that divides d (sub <= 64): the pruned tile of D, every sub-th kept sample stored
bool prune_split(size_t d, size_t& D, size_t& sub) {
    for (size_t c : {(size_t)16, (size_t)8, (size_t)4})
        if (d >= c && d % c == 0 && d / c <= 64) { D = c; sub = d / c; return true; }
    return false;
}
void launch_fftfilt_prune_c32(int log2f, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos2,
                              const cf* twb, hipStream_t s, int sub, NanFix fx) {
    VSrc<float> none{nullptr, 0, nullptr, 0};
    switch (log2f) {
    case 10: launch_prune_one<10, 0>(src, none, out, n_out, L, tw, hpos2, nullptr, twb, s, sub, CarryOut{}, fx); break;
    case 11: launch_prune_one<11, 0>(src, none, out, n_out, L, tw, hpos2, nullptr, twb, s, sub, CarryOut{}, fx); break;
    case 12: launch_prune_one<12, 0>(src, none, out, n_out, L, tw, hpos2, nullptr, twb, s, sub, CarryOut{}, fx); break;
    default: throw Error("fftfilt_prune: unsupported tile size");
    }
}
void launch_fftfilt_prune_f32(int log2f, VSrc<float> src, float* out, long n_out, int L, const cf* tw, const cf* hpos2,
                              const cf* twb, hipStream_t s, int sub, NanFix fx) {
    VSrc<cf> none{nullptr, 0, nullptr, 0};
    cf* o = reinterpret_cast<cf*>(out);
    switch (log2f) {
    case 10: launch_prune_one<10, 2>(none, src, o, n_out, L, tw, hpos2, nullptr, twb, s, sub, CarryOut{}, fx); break;
    case 11: launch_prune_one<11, 2>(none, src, o, n_out, L, tw, hpos2, nullptr, twb, s, sub, CarryOut{}, fx); break;
    case 12: launch_prune_one<12, 2>(none, src, o, n_out, L, tw, hpos2, nullptr, twb, s, sub, CarryOut{}, fx); break;
    default: throw Error("fftfilt_prune: unsupported tile size");
    }
}
void launch_fftfilt_prune_real(int log2f, VSrc<float> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos2r,
                               const cf* hpos2i, const cf* twb, hipStream_t s, CarryOut carry, int sub, NanFix fx) {
    VSrc<cf> none{nullptr, 0, nullptr, 0};
    switch (log2f) {
    case 10: launch_prune_one<10, 1>(none, src, out, n_out, L, tw, hpos2r, hpos2i, twb, s, sub, carry, fx); break;
    case 11: launch_prune_one<11, 1>(none, src, out, n_out, L, tw, hpos2r, hpos2i, twb, s, sub, carry, fx); break;
    case 12: launch_prune_one<12, 1>(none, src, out, n_out, L, tw, hpos2r, hpos2i, twb, s, sub, carry, fx); break;
    default: throw Error("fftfilt_prune: unsupported tile size");
    }
}

template <int LOG2F, int VAR, class SRC>
static void launch_fm_one(SRC src, float* out, int L, const cf* tw, const cf* hpos, const FmChainArgs& h,
                          const cf* last_in, cf* last_out, hipStream_t s) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    FmArgs a;
    a.A = h.A; a.n_y = h.n_y; a.r_lo = h.r_lo; a.r_hi = h.r_hi; a.o_base = h.o_base;
    a.I = h.I; a.D = h.D; a.G = (int)((h.D + h.I - 1) / h.I); a.gain = h.gain; a.mode = h.mode; a.carry = h.carry;
    const long Sp = fm_advance((F - L + 1) - a.G, a.I, a.D, T);
    if (Sp <= 0) throw Error("fm_chain: decimation too large for the tile");
    const long ntiles = (h.n_y + Sp - 1) / Sp;
    if (ntiles <= 0) { launch_carry(src, h.carry, s); return; }
    const size_t smem = sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fm_chain<LOG2F, VAR, SRC>, T, smem, ntiles);
    hipLaunchKernelGGL((k_fm_chain<LOG2F, VAR, SRC>), dim3((unsigned)grid), dim3(T), smem, s, src, out, L, ntiles, tw,
                       hpos, a, last_in, last_out);
    RR_HIP(hipGetLastError());
}

void launch_fm_chain(int log2f, VSrc<cf> src, float* out, int L, const cf* tw, const cf* hpos,
                     const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    switch (log2f) {
    case 10: launch_fm_one<10, 0>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 11: launch_fm_one<11, 0>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 12: launch_fm_one<12, 0>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 13: launch_fm_one<13, 3>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 14: launch_fm_one<14, 3>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    default: throw Error("fm_chain: unsupported tile size");
    }
}
void launch_fm_chain_iq8(int log2f, VSrcIQ8 src, float* out, int L, const cf* tw, const cf* hpos,
                         const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    switch (log2f) {
    case 10: launch_fm_one<10, 0>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 11: launch_fm_one<11, 0>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 12: launch_fm_one<12, 0>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 13: launch_fm_one<13, 3>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    case 14: launch_fm_one<14, 3>(src, out, L, tw, hpos, h, last_in, last_out, s); break;
    default: throw Error("fm_chain: unsupported tile size");
    }
}

// ---- the fused FM chain on 8192 / 16384-point tiles: k_fftfilt_split's filter + k_fm_chain's epilogue ---------------
// After the output butterfly the filtered tile is written to the thread's own natural slots (quarter s in area s),
// so tile position p is area[p / M][p % M] for the resample / demod stage.
template <int NSUB, class SRC>
__global__ __launch_bounds__(256, 2)
void k_fm_chain_split(SRC src, float* __restrict__ out, int L, long ntiles, const cf* __restrict__ tw,
                      const cf* __restrict__ hs, const cf* __restrict__ wk, FmArgs a,
                      const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out) {
    carry_store<cf>(src, a.carry);
    constexpr int LOG2M = 12, M = 1 << LOG2M, T = M / 16, F = NSUB * M;
    constexpr int NP = Plan<LOG2M>::NP;
    constexpr int LE = lds_elems(M);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* area = reinterpret_cast<creg*>(smem_raw);
    const int t = threadIdx.x;
    const int first = L - 1;
    const long Sp = fm_advance((F - L + 1) - a.G, a.I, a.D, 256);
    TileXform<LOG2M, 0> X;
    X.init_no_h(t, tw);
    const creg wbase = to_reg(wk[t]);                   // w_F^t

    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        const long tile = it.tile;
        const long ys = tile * Sp - a.G;                // virtual input index of tile position 0 (see k_fm_chain)
        const bool interior = ys >= src.plen && ys - src.plen + F <= src.in_len;
        if (!interior) {
#pragma unroll
            for (int s = 0; s < NSUB; s++) stage_tile_slow_at<T>(area + s * LE, src, ys + (long)s * M, t);
        }
        {
            const long i0 = ys - src.plen + t;
            constexpr int NB = 16 / NSUB;
#pragma unroll 1
            for (int n0 = 0; n0 < 16; n0 += NB) {
                creg xin[NSUB][NB], wkr[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) wkr[k] = cmul(wbase, split_step<NSUB>(n0 + k));
                if (interior) {
#pragma unroll
                    for (int s = 0; s < NSUB; s++)
#pragma unroll
                        for (int k = 0; k < NB; k++) xin[s][k] = window_at(src, i0 + (long)s * M + (n0 + k) * T);
                } else {
#pragma unroll
                    for (int s = 0; s < NSUB; s++)
#pragma unroll
                        for (int k = 0; k < NB; k++) xin[s][k] = area[s * LE + lds_pad((n0 + k) * T + t)];
                }
#pragma unroll
                for (int k = 0; k < NB; k++) {
                    creg e[NSUB];
#pragma unroll
                    for (int s = 0; s < NSUB; s++) e[s] = xin[s][k];
                    Dft<NSUB, false>::run(e);
                    const creg w1 = wkr[k];
                    e[1] = cmul(e[1], w1);
                    if constexpr (NSUB == 4) { const creg w2 = cmul(w1, w1); e[2] = cmul(e[2], w2); e[3] = cmul(e[3], cmul(w2, w1)); }
#pragma unroll
                    for (int r = 0; r < NSUB; r++) area[r * LE + lds_pad((n0 + k) * T + t)] = e[r];
                }
            }
        }
#pragma unroll 1
        for (int r = 0; r < NSUB; r++) {
            creg* lds = area + r * LE;
            creg v[16];
            creg h[16];
            load_h<LOG2M, NP - 1>(h, t, hs + (long)r * M);     // in flight during the forward transform
            lds_load<LOG2M, 0>(v, t, lds);
            X.forward(v, lds);
            apply_h(v, h);
            if (!RR_ABLATE(512)) X.inverse(v, lds);      // (measurement builds: 512 = no inverse transforms / output butterfly —
            lds_store<LOG2M, 0>(v, t, lds);              //  the most ANY pruning of the inverse side could save)
        }
        // output butterfly: y[n + s M] into the own natural slots of area s
#pragma unroll 1
        for (int n0 = RR_ABLATE(512) ? 16 : 0; n0 < 16; n0 += 8) {
            creg wko[8];
#pragma unroll
            for (int k = 0; k < 8; k++) wko[k] = cmul(wbase, split_step<NSUB>(n0 + k));
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int slot = lds_pad((n0 + k) * T + t);
                creg e[NSUB];
#pragma unroll
                for (int r = 0; r < NSUB; r++) e[r] = area[r * LE + slot];
                const creg w1 = wko[k];
                e[1] = cmulc(e[1], w1);
                if constexpr (NSUB == 4) { const creg w2 = cmul(w1, w1); e[2] = cmulc(e[2], w2); e[3] = cmulc(e[3], cmul(w2, w1)); }
                Dft<NSUB, true>::run(e);
#pragma unroll
                for (int s = 0; s < NSUB; s++) area[s * LE + slot] = e[s];
            }
        }
        tile_sync<T>();
        // resample + demodulate (as k_fm_chain)
        if (!RR_ABLATE(1024))                            // (measurement builds: 1024 = no demodulation)
            fm_epilogue<T>([&](int p) -> creg { return area[(p >> LOG2M) * LE + lds_pad(p & (M - 1))]; }, tile, Sp, ys, first, a, t,
                           out, last_r_in, last_r_out);
        tile_sync<T>();        // epilogue reads done before the next tile rewrites the areas
    }
}

template <int NSUB, class SRC>
static void launch_fm_split_one(SRC src, float* out, int L, const cf* tw, const cf* hs, const cf* wk, const FmChainArgs& h,
                                const cf* last_in, cf* last_out, hipStream_t s) {
    constexpr int F = NSUB * 4096;
    FmArgs a;
    a.A = h.A; a.n_y = h.n_y; a.r_lo = h.r_lo; a.r_hi = h.r_hi; a.o_base = h.o_base;
    a.I = h.I; a.D = h.D; a.G = (int)((h.D + h.I - 1) / h.I); a.gain = h.gain; a.mode = h.mode; a.carry = h.carry;
    const long Sp = fm_advance((F - L + 1) - a.G, a.I, a.D, 256);
    if (Sp <= 0) throw Error("fm_chain: decimation too large for the tile");
    const long ntiles = (h.n_y + Sp - 1) / Sp;
    if (ntiles <= 0) { launch_carry(src, h.carry, s); return; }
    const size_t smem = sizeof(cf) * lds_elems(4096) * NSUB;
    const long grid = grid_for_tiles(k_fm_chain_split<NSUB, SRC>, 256, smem, ntiles);
    hipLaunchKernelGGL((k_fm_chain_split<NSUB, SRC>), dim3((unsigned)grid), dim3(256), smem, s, src, out, L, ntiles, tw, hs, wk,
                       a, last_in, last_out);
    RR_HIP(hipGetLastError());
}
void launch_fm_chain_split(int nsub, VSrc<cf> src, float* out, int L, const cf* tw4096, const cf* hs, const cf* wk,
                           const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    if (nsub == 2) launch_fm_split_one<2>(src, out, L, tw4096, hs, wk, h, last_in, last_out, s);
    else launch_fm_split_one<4>(src, out, L, tw4096, hs, wk, h, last_in, last_out, s);
}
void launch_fm_chain_split_iq8(int nsub, VSrcIQ8 src, float* out, int L, const cf* tw4096, const cf* hs, const cf* wk,
                               const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    if (nsub == 2) launch_fm_split_one<2>(src, out, L, tw4096, hs, wk, h, last_in, last_out, s);
    else launch_fm_split_one<4>(src, out, L, tw4096, hs, wk, h, last_in, last_out, s);
}

template <int LOG2F, class SRC>
static void launch_fm_multi_one(SRC src, float* out, long out_stride, int L, const cf* tw, const cf* hpos_all,
                                int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    constexpr int F = 1 << LOG2F;
    constexpr int T = F / 16;
    FmArgs a;
    a.A = h.A; a.n_y = h.n_y; a.r_lo = h.r_lo; a.r_hi = h.r_hi; a.o_base = h.o_base;
    a.I = h.I; a.D = h.D; a.G = (int)((h.D + h.I - 1) / h.I); a.gain = h.gain; a.mode = h.mode; a.carry = h.carry;
    const long Sp = fm_advance((F - L + 1) - a.G, a.I, a.D, T);
    if (Sp <= 0) throw Error("fm_multi: decimation too large for the tile");
    const long ntiles = (h.n_y + Sp - 1) / Sp;
    if (ntiles <= 0) { launch_carry(src, h.carry, s); return; }
    const size_t smem = 2 * sizeof(cf) * lds_elems(F);
    const long grid = grid_for_tiles(k_fm_multi<LOG2F, SRC>, T, smem, ntiles);
    hipLaunchKernelGGL((k_fm_multi<LOG2F, SRC>), dim3((unsigned)grid), dim3(T), smem, s, src, out, out_stride, L, ntiles,
                       tw, hpos_all, nchan, a, last_in, last_out);
    RR_HIP(hipGetLastError());
}

bool fm_multi_half_supported(int log2f, long I, long D, int L) {
    if (log2f != 11 || I != 1 || D < 2 || (D & 1)) return false;
    return (((1L << log2f) - L + 1) - D - 1) / 2 > 0;
}
template <class SRC>
static void launch_fm_multi_half_t(int log2f, SRC src, float* out, long out_stride, int L, const cf* tw, const cf* tw_half,
                                   const cf* hpos_all, int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    constexpr int LOG2F = 11, F = 1 << LOG2F, T = F / 16;
    if (!fm_multi_half_supported(log2f, h.I, h.D, L)) throw Error("fm_multi_half: unsupported shape");
    FmArgs a;
    a.A = h.A; a.n_y = h.n_y; a.r_lo = h.r_lo; a.r_hi = h.r_hi; a.o_base = h.o_base;
    a.I = h.I; a.D = h.D; a.G = (int)h.D; a.gain = h.gain; a.mode = h.mode; a.carry = h.carry;
    // even advance: one parity shift per call; room for the shift (a multiple of 64 D is even)
    const long Sp = fm_advance(((F - L + 1) - a.G - 1) & ~1L, 1, a.D, 64);
    const long ntiles = (h.n_y + Sp - 1) / Sp;
    if (ntiles <= 0) { launch_carry(src, h.carry, s); return; }
    const size_t smem = sizeof(cf) * (2 * lds_elems(F) + 64);
    const long grid = grid_for_tiles(k_fm_multi_half<LOG2F, SRC>, T, smem, ntiles);
    hipLaunchKernelGGL((k_fm_multi_half<LOG2F, SRC>), dim3((unsigned)grid), dim3(T), smem, s, src, out, out_stride, L, ntiles, Sp,
                       tw, tw_half, hpos_all, nchan, a, last_in, last_out);
    RR_HIP(hipGetLastError());
}
void launch_fm_multi_half(int log2f, VSrc<cf> src, float* out, long out_stride, int L, const cf* tw, const cf* tw_half,
                          const cf* hpos_all, int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_fm_multi_half_t(log2f, src, out, out_stride, L, tw, tw_half, hpos_all, nchan, h, last_in, last_out, s);
}
void launch_fm_multi_half_iq8(int log2f, VSrcIQ8 src, float* out, long out_stride, int L, const cf* tw, const cf* tw_half,
                              const cf* hpos_all, int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_fm_multi_half_t(log2f, src, out, out_stride, L, tw, tw_half, hpos_all, nchan, h, last_in, last_out, s);
}

template <class SRC>
static void launch_fm_chain_half_t(SRC src, float* out, int L, const cf* tw, const cf* tw_half, const cf* hpos,
                                   const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    constexpr int LOG2F = 11, F = 1 << LOG2F, T = F / 16;
    FmArgs a;
    a.A = h.A; a.n_y = h.n_y; a.r_lo = h.r_lo; a.r_hi = h.r_hi; a.o_base = h.o_base;
    a.I = h.I; a.D = h.D; a.G = (int)h.D; a.gain = h.gain; a.mode = h.mode; a.carry = h.carry;
    const long Sp = fm_advance(((F - L + 1) - a.G - 1) & ~1L, 1, a.D, 64);
    const long ntiles = (h.n_y + Sp - 1) / Sp;
    if (ntiles <= 0) { launch_carry(src, h.carry, s); return; }
    const size_t smem = sizeof(cf) * (lds_elems(F) + 2 * lds_elems(F / 2) + 128 + 64);
    const long grid = grid_for_tiles(k_fm_chain_half<LOG2F, SRC>, T, smem, (ntiles + 1) / 2);
    hipLaunchKernelGGL((k_fm_chain_half<LOG2F, SRC>), dim3((unsigned)grid), dim3(T), smem, s, src, out, L, ntiles, Sp, tw,
                       tw_half, hpos, a, last_in, last_out);
    RR_HIP(hipGetLastError());
}
void launch_fm_chain_half(VSrc<cf> src, float* out, int L, const cf* tw, const cf* tw_half, const cf* hpos,
                          const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_fm_chain_half_t(src, out, L, tw, tw_half, hpos, a, last_in, last_out, s);
}
void launch_fm_chain_half_iq8(VSrcIQ8 src, float* out, int L, const cf* tw, const cf* tw_half, const cf* hpos,
                              const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_fm_chain_half_t(src, out, L, tw, tw_half, hpos, a, last_in, last_out, s);
}

// (beyond ~600 taps the 4096-point tiles with a decimating store win again: 1000 taps /2 0.43 vs 0.50 ms)
bool fftfilt_half_supported(int L, long d) { return d >= 2 && (d & 1) == 0 && d / 2 <= 2048 && L >= 1 && L <= 600; }
void launch_fftfilt_half(VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw, const cf* tw_half, const cf* hpos,
                         hipStream_t s, NanFix fx) {
    constexpr int LOG2F = 11, F = 1 << LOG2F, T = F / 16;
    if (!fftfilt_half_supported(L, d)) throw Error("fftfilt_half: unsupported shape");
    if (n_out <= 0) return;
    const long S = ((F - L + 1) - 1) & ~1L;              // even advance, room for the parity shift
    const long n_full = (n_out - 1) * (long)d + 1;
    const long ntiles = (n_full + S - 1) / S;
    const size_t smem = sizeof(cf) * (lds_elems(F) + 2 * lds_elems(F / 2) + 128 + 64);
    const long grid = grid_for_tiles(k_fftfilt_half<LOG2F>, T, smem, (ntiles + 1) / 2);
    // (nan_fix.hpp: tile k keeps the outputs m with m d in [k S, k S + S); tiles go in pairs)
    hipLaunchKernelGGL((k_fftfilt_half<LOG2F>), dim3((unsigned)grid), dim3(T), smem, s, nanfix_ctx(fx, src, out, 2 * S, d, n_out, (ntiles + 1) / 2),
                       src, out, n_out, L, d / 2, ntiles, S, tw, tw_half, hpos);
    RR_HIP(hipGetLastError());
}

bool fm_multi_supported(int log2f) { return log2f >= 10 && log2f <= 12; }

template <class SRC>
static void launch_fm_multi_t(int log2f, SRC src, float* out, long out_stride, int L, const cf* tw, const cf* hpos_all,
                              int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    switch (log2f) {
    case 10: launch_fm_multi_one<10>(src, out, out_stride, L, tw, hpos_all, nchan, h, last_in, last_out, s); break;
    case 11: launch_fm_multi_one<11>(src, out, out_stride, L, tw, hpos_all, nchan, h, last_in, last_out, s); break;
    case 12: launch_fm_multi_one<12>(src, out, out_stride, L, tw, hpos_all, nchan, h, last_in, last_out, s); break;
    default: throw Error("fm_multi: unsupported tile size");
    }
}
void launch_fm_multi(int log2f, VSrc<cf> src, float* out, long out_stride, int L, const cf* tw, const cf* hpos_all,
                     int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_fm_multi_t(log2f, src, out, out_stride, L, tw, hpos_all, nchan, h, last_in, last_out, s);
}
void launch_fm_multi_iq8(int log2f, VSrcIQ8 src, float* out, long out_stride, int L, const cf* tw, const cf* hpos_all,
                         int nchan, const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_fm_multi_t(log2f, src, out, out_stride, L, tw, hpos_all, nchan, h, last_in, last_out, s);
}

}  // namespace rr
