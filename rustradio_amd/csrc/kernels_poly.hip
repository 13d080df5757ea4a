// kernels_poly.hip — fused FM receive chains with an INTEGER decimation, decimate-first ("polyphase") tiles.
//
// FftFilter -> RationalResampler(1, D) keeps r[u] = y[u D] (rational_resampler.rs:183-198 with interp 1): of every D
// filtered samples D - 1 are thrown away.  Splitting the taps and the input into their D phases,
//     r[u] = sum_{p < D} (h_p * x_p)[u],    h_p[j] = t[D j + p],    x_p[u] = x[D u - p],
// turns the L-tap filter at the input rate into D filters of ceil(L / D) taps at the OUTPUT rate, and in the frequency
// domain the sum over the phases is taken BEFORE the inverse transform:
//     R = IFFT_F( sum_p H_p . FFT_F(x_p) ),   H_p = FFT_F(h_p) / F
// — per F low-rate positions D forward transforms and ONE inverse of F points, instead of a forward transform over D F
// input samples, the product, and an inverse whose output is then decimated.  For BASELINE configs[2] / [3] (463 taps,
// 1:6; 78 taps per phase) a 1024-point tile yields 946 demodulated samples from 5676 inputs with 7 transforms of 1024
// points; the 2048-point tiles of kernels_fft.hip spend a 2048- plus a 1024-point transform on 1536 inputs.  (A radix-3
// tile plan, F = 3 * 2^k, would prune the inverse by the same 6 but still pay the full-rate forward transform, and 16
// values per thread do not divide into radix-3 / 6 / 12 groups; the phase transforms are all powers of two.)
//
// One 1024-point transform = ONE WAVE (64 lanes x 16 values, Plan<10> = 16 x 16 x 4; LDS exchanges inside a wave need
// no barrier).  The phases of a tile interleave in memory (x_p[u] = x[D u - p]): lane-consecutive loads of one phase are
// strided by D samples, so the waves that share a tile load their phases at the same time and the cache lines they share
// are fetched once (a wave working through all D phases alone would come back to every line D times, tile-sized reuse
// distances apart).
//   k_fm_chain_poly: one chain.  A workgroup = 2 waves = one tile: wave 0 transforms phases [0, ceil(D/2)), wave 1 the
//     rest, each accumulating its share of sum_p H_p X_p in registers; wave 1 hands its partial sum over through LDS,
//     wave 0 runs the inverse, both demodulate (alternate rounds of 64 outputs).
//   k_fm_multi_poly: N channels on one input (configs[3]).  A workgroup = 8 waves: waves 0 .. D-1 transform one phase
//     each and park the D spectra in LDS (the forward work is shared by all channels, as in k_fm_multi); then every wave
//     takes every 8th channel: sum_p H_{c,p} X_p from the parked spectra, inverse, demodulation.
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "kernels.hpp"
#include "tile_common.hpp"
#include "nan_fix.hpp"

namespace rr {

constexpr int PLG = 10, PF = 1 << PLG, PT = PF / 16;      // tile: 1024 low-rate positions on 64 lanes
constexpr int PLE = lds_elems(PF);

struct PolyArgs {
    long off;            // virtual-stream index of global input sample i is i + off  (off = L - 1 - A)
    long r_lo, r_hi;     // resampled samples of this call: r[u] = y[u D], u in [r_lo, r_hi)
    long o_base;         // demodulated samples emitted before this call
    int Ls;              // taps per phase, ceil(L / D)
    float gain;
    int mode;            // RR_ATAN2_*
    CarryOut carry;      // the block's new carry prefix, written by this launch (common.hpp)
};

// The multi-channel kernel's work split, passed by value (kernel-argument segment: no table in HBM, nothing to upload or to keep
// alive): workgroup b runs the channel rounds [start[b], start[b + 1]) of the launch.
constexpr int POLY_PART_MAX = 320;
struct PolyPart {
    int rounds;                          // channel rounds per tile, ceil(nchan / waves per workgroup)
    int cost;                            // (host) the largest run cost in the units of poly_partition's weights
    int start[POLY_PART_MAX + 1];
};

// tuning: phases loaded per batch and waves per SIMD of the single-chain kernel (measured on MI355X, profiles/TUNING_LOG.md)
#ifndef RR_POLY_NB
#define RR_POLY_NB 3
#endif
#ifndef RR_POLY_WAVES
#define RR_POLY_WAVES 2
#endif
#ifndef RR_POLY_OVERSUB
#define RR_POLY_OVERSUB 3
#endif
#ifndef RR_POLY_WIDE
#define RR_POLY_WIDE 1
#endif
// measurement builds only (make EXTRA=-DRR_POLY_ABLATE=<bits>, wrong results): 1 no input loads, 2 no atan2, 4 no output
// stores, 8 no H loads, 16 no LDS exchanges inside the transforms
#ifndef RR_POLY_CHAIN_W3
#define RR_POLY_CHAIN_W3 1
#endif
#ifndef RR_POLY_PIPE
#define RR_POLY_PIPE 1
#endif
#ifndef RR_POLY_PIPE_SPLIT
#define RR_POLY_PIPE_SPLIT 8
#endif
#ifndef RR_POLY_H16
#define RR_POLY_H16 1
#endif
#ifndef RR_POLY_CHAIN_TWLDS
#define RR_POLY_CHAIN_TWLDS 1
#endif
#ifndef RR_POLY_ABLATE
#define RR_POLY_ABLATE 0
#endif
typedef float creg2 __attribute__((ext_vector_type(4)));
typedef creg2 creg2u __attribute__((aligned(8)));                        // two adjacent Complex samples, 8-byte aligned
#ifndef RR_STAMP_WAVE_A
#define RR_STAMP_WAVE_A 0
#define RR_STAMP_WAVE_B 1
#endif
#ifdef RR_FFT_TIMING_BUILD
#define PSTAMP(i) do { if (stamps) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamps[i] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void wave_fence() { asm volatile("" ::: "memory"); }   // a wave's LDS operations execute in order
// phase markers for tools/isa_phase_table.py (a comment line in the assembly; measurement builds only: -DRR_ISA_MARKS)
#ifdef RR_ISA_MARKS
#define RR_MARK(name) asm volatile("; RRMARK " name ::: "memory")
#else
#define RR_MARK(name) do { } while (0)
#endif

// acc + a * w (complex): the product's two halves as packed FMAs (see cmul in fft_core.hpp)
__device__ __forceinline__ creg cmac(creg acc, creg a, creg w) {
#if defined(__HIP_DEVICE_COMPILE__)
    creg t = a * __builtin_shufflevector(w, w, 0, 0) + acc, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
#else
    return cadd(acc, cmul(a, w));
#endif
}

// two of them issued interleaved (fma, fma, fma, fma): the second half of a product reads what the first half wrote right
// before it, which costs a wait state (an s_nop per product: 96 of a channel-tile's ~1400 issue slots); with the partner's
// first half in between there is nothing to wait for (cmul2 in fft_core.hpp does the same inside the transforms)
__device__ __forceinline__ void cmac2(creg& acc0, creg a0, creg w0, creg& acc1, creg a1, creg w1) {
#if defined(__HIP_DEVICE_COMPILE__)
    creg t0 = a0 * __builtin_shufflevector(w0, w0, 0, 0) + acc0;
    creg t1 = a1 * __builtin_shufflevector(w1, w1, 0, 0) + acc1;
    creg r0, r1;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r0) : "v"(a0), "v"(w0), "v"(t0));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r1) : "v"(a1), "v"(w1), "v"(t1));
    acc0 = r0; acc1 = r1;
#else
    acc0 = cadd(acc0, cmul(a0, w0)); acc1 = cadd(acc1, cmul(a1, w1));
#endif
}

__device__ __forceinline__ creg poly_elem(const VSrc<cf>& s, long i) { return as_global(reinterpret_cast<const creg*>(s.in))[i]; }
__device__ __forceinline__ creg poly_elem(const VSrcIQ8& s, long i) {
    return to_reg(VSrcIQ8::decode(as_global(reinterpret_cast<const unsigned short*>(s.in))[i]));
}

// Boundary tiles (touching the carry prefix, the start of the stream or the end of the window) are staged through
// the wave's exchange area by an out-of-line routine, so that their index arithmetic costs the steady-state path nothing.
template <class SRC>
__device__ __attribute__((noinline)) void poly_stage_slow(creg* ex, SRC src, long vbase, int D, int p, int t) {
    for (int n = 0; n < 16; n++) {
        const long vi = vbase + (long)D * (PT * n + t) - p;
        ex[lds_pad(PT * n + t)] = vi >= 0 ? to_reg(src.load(vi)) : mk(0.0f, 0.0f);
    }
}
// v[n] = V[vbase + D (64 n + t) - p]: phase p of the tile whose position 0 is the low-rate sample at vbase / D.
// interior (wave-uniform): the whole tile, all phases, lies inside the caller's window.
template <int D, class SRC>
__device__ __forceinline__ void poly_load(creg* v, const SRC& src, long vbase, int p, int t, bool interior, creg* ex) {
    if (interior) {
        const long i0 = vbase - src.plen - p + (long)D * t;
#pragma unroll
        for (int n = 0; n < 16; n++) v[n] = poly_elem(src, i0 + (long)D * PT * n);
    } else {
        poly_stage_slow(ex, src, vbase, D, p, t);
        wave_fence();
        lds_load<PLG, 0>(v, t, ex);                      // (own slots: each lane reads back what it wrote)
        wave_fence();
    }
}

// forward 1024-point transform of one wave, exchanges in `ex`; pass-1 twiddles w_64^(k (t % 4)) from the LDS table
__device__ __forceinline__ void poly_forward(creg* v, int t, creg* ex, const creg* tw0, const creg* tab1) {
    creg twl[15];
    fwd_pass<PLG, 0>(v, tw0);
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 0>(v, t, ex);
    wave_fence();
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 1>(v, t, ex);
#pragma unroll
    for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<PLG, 1>::lo(t)];
    fwd_pass<PLG, 1>(v, twl);
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 1>(v, t, ex);
    wave_fence();
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 2>(v, t, ex);
    fwd_pass<PLG, 2>(v, twl);                        // (P == 1: no twiddles)
}
// inverse: spectrum in the pass-2 layout -> v[n] = tile position 64 n + t
__device__ __forceinline__ void poly_inverse(creg* v, int t, creg* ex, const creg* tw0, const creg* tab1) {
    creg twl[15];
    inv_pass<PLG, 2>(v, twl);
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 2>(v, t, ex);
    wave_fence();
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 1>(v, t, ex);
#pragma unroll
    for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<PLG, 1>::lo(t)];
    inv_pass<PLG, 1>(v, twl);
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 1>(v, t, ex);
    wave_fence();
    if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 0>(v, t, ex);
    inv_pass<PLG, 0>(v, tw0);
}

// forward transform with the pass-0 twiddles read from an LDS table [15][64] (see poly_inverse_tab)
__device__ __forceinline__ void poly_forward_tab(creg* v, int t, creg* ex, const creg* tw0tab, const creg* tab1) {
    creg twl[15];
#pragma unroll
    for (int k = 0; k < 15; k++) twl[k] = tw0tab[k * PT + t];
    fwd_pass<PLG, 0>(v, twl);
    lds_store<PLG, 0>(v, t, ex);
    wave_fence();
    lds_load<PLG, 1>(v, t, ex);
#pragma unroll
    for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<PLG, 1>::lo(t)];
    fwd_pass<PLG, 1>(v, twl);
    lds_store<PLG, 1>(v, t, ex);
    wave_fence();
    lds_load<PLG, 2>(v, t, ex);
    fwd_pass<PLG, 2>(v, twl);
}
// the same with the pass-0 twiddles read from an LDS table [15][64] right before the last pass
__device__ __forceinline__ void poly_inverse_tab(creg* v, int t, creg* ex, const creg* tw0tab, const creg* tab1) {
    creg twl[15];
    inv_pass<PLG, 2>(v, twl);
    lds_store<PLG, 2>(v, t, ex);
    wave_fence();
    lds_load<PLG, 1>(v, t, ex);
#pragma unroll
    for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<PLG, 1>::lo(t)];
    inv_pass<PLG, 1>(v, twl);
    lds_store<PLG, 1>(v, t, ex);
    wave_fence();
    lds_load<PLG, 0>(v, t, ex);
#pragma unroll
    for (int k = 0; k < 15; k++) twl[k] = tw0tab[k * PT + t];
    inv_pass<PLG, 0>(v, twl);
}

// Phases [P0, P0 + NPH) of one tile on this wave: z += sum_p H_p X_p.  Loaded in batches of up to 3 phases — the lines
// a batch shares are then touched by back-to-back loads (one fetch per line and wave) and there is one memory latency
// per batch; hreg = this channel's responses, register-major [p][16][64].  Register budget (256 at 2 waves / SIMD): a
// batch 96 + z 32 + pass-0 twiddles 30 + one pass's temporaries; H is fetched during the last (twiddle-free) pass.
template <int D, int NPH, class SRC>
__device__ __forceinline__ void poly_phases(creg* z, const SRC& src, long vbase, bool interior, int t, creg* ex,
                                            const creg* tw0, const creg* tab1, const creg* __restrict__ hreg, int P0) {
    constexpr int NB = RR_POLY_NB < NPH ? RR_POLY_NB : NPH;
#pragma unroll
    for (int pb = 0; pb < NPH; pb += NB) {
        creg v[NB][16];
        if (interior) {
            // position-major: the NB loads of one n touch the same cache lines back to back (phase-major order would
            // come back to every line of the 48 KB tile NB times, a whole tile apart: the L1 does not hold a tile)
            const long i0 = vbase - src.plen - (P0 + pb) + (long)D * t;
            const int nbv = NPH - pb < NB ? NPH - pb : NB;            // phases of this batch (a constant once unrolled)
            if constexpr ((RR_POLY_ABLATE & 1) != 0) {
#pragma unroll
                for (int n = 0; n < 16; n++)
#pragma unroll
                    for (int i = 0; i < NB; i++)
                        if (pb + i < NPH) v[i][n] = mk((float)(t + n) * 1e-3f, (float)(i0 & 255) * 1e-3f);
            } else if constexpr (std::is_same<SRC, VSrc<cf>>::value && RR_POLY_WIDE) {
                // the batch's phases are NBV ADJACENT samples per lane: fetched as 16-byte pairs (+ one 8-byte rest) — every
                // load instruction of a 48-byte lane stride looks up the same 24 cache lines whatever its width, so two
                // loads per position instead of three are a third fewer L1 look-ups
                // (RR_POLY_ABLATE & 128, timing only, wrong data: the same bytes as lane-consecutive 16-byte chunks)
                const gptr<creg> base = (RR_POLY_ABLATE & 128) ? as_global(reinterpret_cast<const creg*>(src.in) + (i0 - (long)D * t + 2 * t + 2))
                                                               : as_global(reinterpret_cast<const creg*>(src.in) + i0);
#pragma unroll
                for (int n = 0; n < 16; n++) {
                    const gptr<creg> q = base + ((RR_POLY_ABLATE & 128) ? 128L * (3 * n + (P0 >> 1)) : (long)D * PT * n);
#pragma unroll
                    for (int k = 0; k < (NB + 1) / 2; k++) {
                        const int i = nbv - 1 - 2 * k;
                        if (i >= 1) {
                            const creg2u w = *reinterpret_cast<gptr<creg2u>>(q - i);
                            v[i][n] = mk(w.x, w.y);
                            v[i - 1][n] = mk(w.z, w.w);
                        } else if (i == 0) {
                            v[0][n] = q[0];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int n = 0; n < 16; n++)
#pragma unroll
                    for (int i = 0; i < NB; i++)
                        if (pb + i < NPH) v[i][n] = poly_elem(src, i0 - i + (long)D * PT * n);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; i++)
                if (pb + i < NPH) poly_load<D>(v[i], src, vbase, P0 + pb + i, t, false, ex);
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            if (pb + i < NPH) {
                __builtin_amdgcn_sched_barrier(0);
                {
                    creg twl[15];
#if RR_POLY_CHAIN_TWLDS
#pragma unroll
                    for (int k = 0; k < 15; k++) twl[k] = tw0[k * PT + t];          // (tw0 = the LDS table [15][64])
                    fwd_pass<PLG, 0>(v[i], twl);
#else
                    fwd_pass<PLG, 0>(v[i], tw0);
#endif
                    if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 0>(v[i], t, ex);
                    wave_fence();
                    if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 1>(v[i], t, ex);
#pragma unroll
                    for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<PLG, 1>::lo(t)];
                    fwd_pass<PLG, 1>(v[i], twl);
                    if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 1>(v[i], t, ex);
                    wave_fence();
                    if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 2>(v[i], t, ex);
                }
                __builtin_amdgcn_sched_barrier(0);
                creg h[16];
#if RR_POLY_H16
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const gptr<f32x4> hq = as_global(reinterpret_cast<const f32x4*>(hreg + (long)(P0 + pb + i) * 16 * PT) + t);
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    if constexpr ((RR_POLY_ABLATE & 8) != 0) { h[2 * jj] = mk(1.0f + jj, (float)t); h[2 * jj + 1] = mk(2.0f + jj, (float)t); }
                    else { const f32x4 q = hq[jj * PT]; h[2 * jj] = mk(q.x, q.y); h[2 * jj + 1] = mk(q.z, q.w); }
                }
#else
                const creg* hp = hreg + (long)(P0 + pb + i) * 16 * PT + t;
#pragma unroll
                for (int j = 0; j < 16; j++) h[j] = (RR_POLY_ABLATE & 8) ? mk(1.0f + j, (float)t) : hp[j * PT];
#endif
                fwd_pass<PLG, 2>(v[i], nullptr);         // (P == 1: no twiddles)
#pragma unroll
                for (int j = 0; j < 16; j += 2) cmac2(z[j], v[i][j], h[j], z[j + 1], v[i][j + 1], h[j + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// The two halves of poly_phases for kernels that issue a tile's loads ahead of its arithmetic (k_fm_chain_polyw; one batch of
// NPH <= 2 phases per wave).  poly_issue: interior tiles only — the NPH phases of position t + 64 n are adjacent samples.
template <int D, int NPH, int N0 = 0, int N1 = 16, class SRC>
__device__ __forceinline__ void poly_issue(creg (*v)[16], const SRC& src, long vbase, int t, int P0) {
    static_assert(NPH == 1 || NPH == 2, "one batch of at most two phases");
    const long i0 = vbase - src.plen - P0 + (long)D * t;
    if constexpr ((RR_POLY_ABLATE & 1) != 0) {
#pragma unroll
        for (int n = N0; n < N1; n++)
#pragma unroll
            for (int i = 0; i < NPH; i++) v[i][n] = mk((float)(t + n) * 1e-3f, (float)(i0 & 255) * 1e-3f);
    } else if constexpr (std::is_same<SRC, VSrc<cf>>::value && NPH == 2) {
        const gptr<creg> base = as_global(reinterpret_cast<const creg*>(src.in) + i0);
#pragma unroll
        for (int n = N0; n < N1; n++) {
            const creg2u w = *reinterpret_cast<gptr<creg2u>>(base + (long)D * PT * n - 1);
            v[1][n] = mk(w.x, w.y);
            v[0][n] = mk(w.z, w.w);
        }
    } else {
#pragma unroll
        for (int n = N0; n < N1; n++)
#pragma unroll
            for (int i = 0; i < NPH; i++) v[i][n] = poly_elem(src, i0 - i + (long)D * PT * n);
    }
}
// the same for NPH == 2 on Complex streams, rows [N0, N1), into RAW 16-byte registers (phase P0 + 1 in .xy, phase P0 in .zw):
// nothing touches the loaded values until poly_unpack, so nothing waits for them where they are issued
template <int D, int N0, int N1>
__device__ __forceinline__ void poly_issue_raw(creg2* raw, const creg* lane_base, int t) {
    const gptr<creg> base = as_global(lane_base);        // the sample of phase P0 + 1 at position t
#pragma unroll
    for (int n = N0; n < N1; n++) {
        if constexpr ((RR_POLY_ABLATE & 1) != 0) raw[n] = creg2{(float)(t + n) * 1e-3f, 1e-3f, 1e-3f, 2e-3f};
        else raw[n] = *reinterpret_cast<gptr<creg2u>>(base + (long)D * PT * n);
    }
}
// forward transform of one phase (pass-0 twiddles from the LDS table tw0tab), z += H_phase X
// (PAIR: products issued two by two, cmac2 — its two temporaries spill in the FirFilter instantiations, which sit at 168 VGPRs)
template <bool PAIR = true>
__device__ __forceinline__ void poly_xform_mac(creg* z, creg* vi, int t, creg* ex, const creg* tw0tab, const creg* tab1,
                                               const creg* __restrict__ hreg, int phase) {
    __builtin_amdgcn_sched_barrier(0);
    {
        creg twl[15];
#pragma unroll
        for (int k = 0; k < 15; k++) twl[k] = tw0tab[k * PT + t];
        fwd_pass<PLG, 0>(vi, twl);
        if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 0>(vi, t, ex);
        wave_fence();
        if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 1>(vi, t, ex);
#pragma unroll
        for (int k = 1; k < 16; k++) twl[k - 1] = tab1[k * PassGeom<PLG, 1>::lo(t)];
        fwd_pass<PLG, 1>(vi, twl);
        if constexpr (!(RR_POLY_ABLATE & 16)) lds_store<PLG, 1>(vi, t, ex);
        wave_fence();
        if constexpr (!(RR_POLY_ABLATE & 16)) lds_load<PLG, 2>(vi, t, ex);
    }
    __builtin_amdgcn_sched_barrier(0);
    creg h[16];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const gptr<f32x4> hq = as_global(reinterpret_cast<const f32x4*>(hreg + (long)phase * 16 * PT) + t);
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        if constexpr ((RR_POLY_ABLATE & 8) != 0) { h[2 * jj] = mk(1.0f + jj, (float)t); h[2 * jj + 1] = mk(2.0f + jj, (float)t); }
        else { const f32x4 q = hq[jj * PT]; h[2 * jj] = mk(q.x, q.y); h[2 * jj + 1] = mk(q.z, q.w); }
    }
    fwd_pass<PLG, 2>(vi, nullptr);                   // (P == 1: no twiddles)
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        if constexpr (PAIR) cmac2(z[j], vi[j], h[j], z[j + 1], vi[j + 1], h[j + 1]);
        else { z[j] = cmac(z[j], vi[j], h[j]); z[j + 1] = cmac(z[j + 1], vi[j + 1], h[j + 1]); }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The finished tile in natural order, UNPADDED (position p in slot p): the demodulation below reads 64 consecutive slots from
// an arbitrary start (Ls + i), and 32 lanes x 8 B from ANY 8-byte-aligned start cover the 64 banks exactly once, while the
// padded exchange layout puts two or three pad slots inside such a run and made every one of these reads a 2-way conflict
// (round 2 counters: SQ_LDS_BANK_CONFLICT 7.8e5 per fm_multi launch = 406 k demodulation reads x 2 extra cycles).
__device__ __forceinline__ void nat_store(const creg* z, int t, creg* area) {
#pragma unroll
    for (int n = 0; n < 16; n++) area[n * PT + t] = z[n];
}

// Demodulation of one tile: tile position Ls + i (natural order, nat_store) holds r[u0 + i]; this lane takes i = lane0,
// lane0 + stride, ...  The loop body is two LDS reads, the
// conj-multiply as one packed multiply + one packed FMA, atan2 and one store; the special samples — r[0] has no partner,
// the first pair of a call takes its lower sample from the previous call, the last r of a call is carried — sit in the
// first / last tile of a call only and are handled under wave-uniform tests outside the steady-state loop.
// (conj(rl) * ru here is ru * conj(rl) with FMA contraction: within 1 ulp of the reference's un-fused form,
//  quadrature_demod.rs:72; zeros stay exact zeros, so atan2(0, 0) * gain == 0 still holds.)
template <int MODE, bool TAME = false>
__device__ __forceinline__ float poly_angle(creg rl, creg ru, float gain) {
    const creg zz = cmulc(ru, rl);
    if constexpr ((RR_POLY_ABLATE & 2) != 0) return gain * (zz.x + zz.y);
    return gain * (MODE == 0 ? atan2_poly<TAME>(zz.y, zz.x) : fmc_atan2(zz.y, zz.x));
}
// Whether a finished tile (the 16 values of every lane of the wave that ran its inverse transform) is TAME: every value
// finite and below 1e18 in magnitude, so that conj(a) * b of any two of them is finite too and the demodulation may leave
// the inf / NaN fix-ups of atan2 out (4 of its ~31 instructions per output; the demodulation is 47 % of the multi-channel
// kernel's VALU instructions: profiles/r05_fm_multi_phase_table.txt).  Sum of squares through packed FMAs: a NaN or an inf
// anywhere propagates into the sum (a maximum would drop NaNs), and a square overflows where the products could.  16 packed
// FMAs + a ballot per tile against 4 instructions x 15 outputs per lane.  Wave-uniform result.
__device__ __forceinline__ bool tile_tame(const creg* z) {
#if defined(__HIP_DEVICE_COMPILE__)
    creg a0 = z[0] * z[0], a1 = z[1] * z[1];
#pragma unroll
    for (int n = 2; n < 16; n += 2) { a0 = z[n] * z[n] + a0; a1 = z[n + 1] * z[n + 1] + a1; }
    const creg a = a0 + a1;
    const float q = a.x + a.y;
    return __builtin_amdgcn_ballot_w64(!(q < 1e36f)) == 0;
#else
    (void)z;
    return false;
#endif
}
template <int MODE, bool TAME = false>
__device__ __forceinline__ void poly_demod_tile(const creg* ldsR, int lane0, int stride, long u0, int Sa, const PolyArgs& a,
                                                float* __restrict__ out, const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out) {
    const long left = a.r_hi - u0;
    const int nv = left < Sa ? (int)left : Sa;                          // valid samples of this tile
    float* o = out + ((u0 - 1) - a.o_base) + lane0;                      // o[i] = demodulated pair (r[u0 + i - 1], r[u0 + i])
    const creg* pu = ldsR + a.Ls + lane0;
    const creg* pl = pu - 1;
    const int inc = stride;
    int i = lane0;
    if (u0 == a.r_lo) {                                                  // (wave-uniform) first tile of the call
        if (i < nv) {
            const creg ru = *pu;
            const creg rl = i == 0 ? to_reg(last_r_in[0]) : *pl;       // lower sample from the previous call (any value: full atan2)
            if (!(i == 0 && u0 == 0)) *o = poly_angle<MODE>(rl, ru, a.gain);   // r[0] has no lower partner
        }
        i += stride; pu += inc; pl += inc; o += stride;
    }
    // two outputs per iteration: the two angle computations are independent, so their instructions interleave and fill the
    // wait states a single dependent chain leaves on gfx950 (a packed product feeding the next operation, v_cmp -> v_cndmask
    // through VCC: 6 of the 45 issue slots of a round were s_nop) and the loop bookkeeping is paid once per pair
    for (; i + stride < nv; i += 2 * stride, pu += 2 * inc, pl += 2 * inc, o += 2 * stride) {
        const creg l0 = pl[0], u0v = pu[0], l1 = pl[inc], u1v = pu[inc];
        // (round 3: the two angles' reductions, polynomials and gains as packed FP32 — 62 instead of 77 instructions per pair —
        //  measured 0.0770 against 0.0765 ms: a v_pk_fma_f32 holds the SIMD 4.5 clocks against 2.6 for a v_fma_f32
        //  (tools/micro/valubench.hip), so packing two independent scalar chains buys 13 % of their issue time and the
        //  dependent packed chain pays it back in wait states.  Removed.)
        const float y0 = poly_angle<MODE, TAME>(l0, u0v, a.gain);
        const float y1 = poly_angle<MODE, TAME>(l1, u1v, a.gain);
        if constexpr ((RR_POLY_ABLATE & 4) != 0) { if (y0 == 1234.5678f) { o[0] = y0; o[stride] = y1; } } else { o[0] = y0; o[stride] = y1; }
    }
    for (; i < nv; i += stride, pu += inc, pl += inc, o += stride) {
        const float y = poly_angle<MODE, TAME>(*pl, *pu, a.gain);
        if constexpr ((RR_POLY_ABLATE & 4) != 0) { if (y == 1234.5678f) *o = y; } else { *o = y; }
    }
    if (left <= Sa) {                                                    // (wave-uniform) last tile: carry r[r_hi - 1]
        if (nv - 1 >= lane0 && ((nv - 1 - lane0) % stride) == 0) last_r_out[0] = from_reg(ldsR[a.Ls + nv - 1]);
    }
}

// MODE 2 of the chain kernel (decimating FirFilter<Complex>): the tile's samples r[u0 + i] go out as they are
__device__ __forceinline__ void poly_store_tile(const creg* ldsR, int lane0, int stride, long u0, int Sa, const PolyArgs& a,
                                                creg* __restrict__ out) {
    const long left = a.r_hi - u0;
    const int nv = left < Sa ? (int)left : Sa;
    creg* o = out + (u0 - a.o_base) + lane0;
    const creg* pu = ldsR + a.Ls + lane0;
    for (int i = lane0; i < nv; i += stride, pu += stride, o += stride) *o = *pu;
}

// mode 2 (decimating FirFilter): the reference's locality for non-finite samples (nan_fix.hpp; the context is the kernels'
// first argument, built by launch_chain_poly_d)

// (FIR = the FirFilter instantiation, launch_fir_poly: the chains' own carry none of the repair's code)
template <int D, class SRC, bool FIR = false>
__global__ __launch_bounds__(128, RR_POLY_WAVES)
void k_fm_chain_poly(NanFixCtx nfx, SRC src, float* __restrict__ out, long ntiles, const cf* __restrict__ tw, const cf* __restrict__ hreg,
                     PolyArgs a, const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out, unsigned long long* __restrict__ dbg) {
    carry_store<cf>(src, a.carry);
    (void)nfx;
    if constexpr (FIR) nf_init();
    constexpr int PHA = (D + 1) / 2, PHB = D - PHA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* exB = lds + PLE;                               // wave 1's exchanges, then its partial sum (register-major)
    creg* ldsR = lds + 2 * PLE;                          // the tile's resampled samples, natural order
    creg* tab1 = lds + 3 * PLE;                          // w_64^j
    const int w = threadIdx.x >> 6, t_ = threadIdx.x & 63;
    creg* ex = lds + w * PLE;
#if RR_POLY_CHAIN_TWLDS
    // pass-0 twiddles of lane t in an LDS table [15][64] (read right before the passes that use them) instead of 30 VGPRs
    // held for the whole kernel: the registers go to the 16-byte response loads
    creg* tw0 = tab1 + 64;
    int* tame_flag = reinterpret_cast<int*>(tw0 + 15 * PT);     // wave 0's tile_tame() verdict for the tile being demodulated
    if (w == 0) {
        creg twr[15];
        load_twiddles<PLG, 0>(twr, t_, tw);
#pragma unroll
        for (int k = 0; k < 15; k++) tw0[k * PT + t_] = twr[k];
    }
#else
    creg tw0[15];
    load_twiddles<PLG, 0>(tw0, t_, tw);
    int* tame_flag = reinterpret_cast<int*>(tab1 + 64);
#endif
    if (w == 0) tab1[t_] = to_reg(tw[t_ * (PF / 64)]);
    tile_sync<128>();
    const int Sa = PF - a.Ls;                            // demodulated samples per tile
    const creg* hr = reinterpret_cast<const creg*>(hreg);

    int iter = 0;
    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step, iter++) {
#ifdef RR_FFT_TIMING_BUILD
        unsigned long long* stamps = (dbg && blockIdx.x == 0 && t_ == 0 && iter == 2) ? dbg + 16 * w : nullptr;
#else
        (void)dbg; (void)iter;
#endif
        PSTAMP(0);
        const long u0 = a.r_lo + it.tile * Sa;           // tile position Ls holds r[u0]
        const long vbase = (u0 - a.Ls) * D + a.off;
        const bool interior = vbase - (D - 1) >= src.plen && vbase + (long)D * (PF - 1) - src.plen < src.in_len;
        creg z[16];
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = mk(0.0f, 0.0f);
        int t = t_;                                      // opaque per tile: keeps the tile body's LDS / table addresses from
        asm volatile("" : "+v"(t));                      // being hoisted out of the loop into persistent VGPRs
        if (w == 0) {
            poly_phases<D, PHA>(z, src, vbase, interior, t, ex, tw0, tab1, hr, 0);
        } else {
            if constexpr (PHB > 0) poly_phases<D, PHB>(z, src, vbase, interior, t, ex, tw0, tab1, hr, PHA);
#pragma unroll
            for (int j = 0; j < 16; j++) exB[j * PT + t] = z[j];
        }
        PSTAMP(1);
        tile_sync<128>();
        PSTAMP(2);
        if (w == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) z[j] = cadd(z[j], exB[j * PT + t]);
#if RR_POLY_CHAIN_TWLDS
            poly_inverse_tab(z, t, ex, tw0, tab1);
#else
            poly_inverse(z, t, ex, tw0, tab1);
#endif
            if constexpr (FIR) nf_mark(nf_bad(z[15].x));
            nat_store(z, t, ldsR);
            if constexpr (!FIR) { const bool tame = tile_tame(z); if (t == 0) tame_flag[0] = tame; }
        }
        PSTAMP(3);
        tile_sync<128>();
        PSTAMP(4);
        if constexpr (FIR) {                                 // (launch_fir_poly: mode 2 only)
            poly_store_tile(ldsR, w * PT + t, 2 * PT, u0, Sa, a, reinterpret_cast<creg*>(out));
        } else if (a.mode == 0) {
            if (__builtin_amdgcn_readfirstlane(tame_flag[0])) poly_demod_tile<0, true>(ldsR, w * PT + t, 2 * PT, u0, Sa, a, out, last_r_in, last_r_out);
            else poly_demod_tile<0>(ldsR, w * PT + t, 2 * PT, u0, Sa, a, out, last_r_in, last_r_out);
        } else if (a.mode == 1) poly_demod_tile<1>(ldsR, w * PT + t, 2 * PT, u0, Sa, a, out, last_r_in, last_r_out);
        else poly_store_tile(ldsR, w * PT + t, 2 * PT, u0, Sa, a, reinterpret_cast<creg*>(out));
        PSTAMP(5);
        // (the next tile rewrites exB / ldsR only after its first barrier, which both waves reach after these reads)
    }
    if constexpr (FIR) nf_finish<cf, cf>();
}

// ---- the same with three waves per SIMD: NW waves per workgroup, D / NW phases each ---------------------------------------
// (1:6 -> 3 waves x 2 phases, 1:4 -> 2 x 2, 1:8 -> 4 x 2, 1:3 -> 3 x 1, 1:2 -> 2 x 1: at most two phases per wave, pass-0
// twiddles in the LDS table, 16-byte response loads = 168 VGPRs.)  The natural-order tile shares wave 0's exchange area —
// one more barrier per tile, 8.7 KB less per workgroup — so that 12 waves' workgroups fit the CU's LDS.
template <int D, int NW, class SRC, bool FIR = false>
__global__ __launch_bounds__(64 * NW, 3)
void k_fm_chain_polyw(NanFixCtx nfx, SRC src, float* __restrict__ out, long ntiles, const cf* __restrict__ tw, const cf* __restrict__ hreg,
                      PolyArgs a, const cf* __restrict__ last_r_in, cf* __restrict__ last_r_out) {
    carry_store<cf>(src, a.carry);
    (void)nfx;
    if constexpr (FIR) nf_init();
    static_assert((D + NW - 1) / NW <= 2, "at most two phases per wave");
    constexpr int PPW = (D + NW - 1) / NW;               // (odd decimations, round 4: the last wave has one phase; its loads still
                                                         //  fetch the pair — the sample in front of phase D - 1 is the neighbouring
                                                         //  position's phase 0 — and tile_geom keeps that sample inside the window)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* ldsR = lds;                                    // the tile's resampled samples, natural order (= wave 0's area)
    creg* tab1 = lds + NW * PLE;                         // w_64^j
    creg* tw0 = tab1 + 64;                               // pass-0 twiddles of lane t, [15][64]
    int* tame_flag = reinterpret_cast<int*>(tw0 + 15 * PT);   // wave 0's tile_tame() verdict for the tile being demodulated
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t_ = threadIdx.x & 63;
    creg* ex = lds + w * PLE;
    if (w == 0) {
        creg twr[15];
        load_twiddles<PLG, 0>(twr, t_, tw);
#pragma unroll
        for (int k = 0; k < 15; k++) tw0[k * PT + t_] = twr[k];
        tab1[t_] = to_reg(tw[t_ * (PF / 64)]);
    }
    tile_sync<64 * NW>();
    const int Sa = PF - a.Ls;
    const creg* hr = reinterpret_cast<const creg*>(hreg);
    // Software-pipelined over tiles: the NEXT tile's loads are issued right after this tile's last product — half of them
    // before the inverse transform, half after it (all 16 before it spill) — and fly during the partial sums, the inverse and
    // the demodulation, a quarter of a tile's time and about one memory latency, in the registers the transforms have just
    // released.  (Steady state, tools/chain_ablate.sh: 14.5 ns per tile, 10.4 without the input loads, 14.0 with them
    // lane-consecutive — what the loads cost is a wave standing still for their latency.  Pipelined: 12.9 ns per tile at
    // 9.6e7 samples, the metric's fused chain 0.2455 -> 0.2325 ms per 1e8; no change where a workgroup gets one or two tiles.)
    auto tile_geom = [&](long tile, long& u0, long& vbase, bool& interior) {
        u0 = a.r_lo + tile * Sa;
        vbase = (u0 - a.Ls) * D + a.off;
        interior = vbase - (NW * PPW - 1) >= src.plen && vbase + (long)D * (PF - 1) - src.plen < src.in_len;
    };
    // Pipelined for two-phase waves on Complex streams (the byte stream's samples are converted where they are loaded).  The
    // prefetch is issued on EVERY iteration — for a boundary or missing next tile from the response table, 48 KB that are
    // always there, and thrown away — so that the raw registers are dead between their unpacking and the next issue; a
    // conditional definition would keep all 64 of them alive through the transforms.
    constexpr bool PIPE = RR_POLY_PIPE && PPW == 2 && std::is_same<SRC, VSrc<cf>>::value;
    creg2 raw[PIPE ? 16 : 1];
    bool have = false;                                   // raw holds, or is receiving, the coming tile's samples
    auto lane_base = [&](bool ok, long vb, int tt) -> const creg* {
        const creg* p = hr;                              // (safe dummy: D * 1024 samples of response)
        if constexpr (std::is_same<SRC, VSrc<cf>>::value) {
            if (ok) p = reinterpret_cast<const creg*>(src.in) + (vb - src.plen - w * PPW - 1);
        }
        return p + (long)D * tt;
    };
    if constexpr (PIPE) {
        TileIter it0(ntiles);
        long u0, vbase = 0; bool interior = false;
        if (it0.tile < it0.end) tile_geom(it0.tile, u0, vbase, interior);
        poly_issue_raw<D, 0, 16>(raw, lane_base(interior, vbase, t_), t_);
        have = interior;
    }
    for (TileIter it(ntiles); it.tile < it.end; it.tile += it.step) {
        long u0, vbase; bool interior;
        tile_geom(it.tile, u0, vbase, interior);
        creg z[16], v[PPW][16];
#pragma unroll
        for (int j = 0; j < 16; j++) z[j] = mk(0.0f, 0.0f);
        int t = t_;
        asm volatile("" : "+v"(t));
        if constexpr (PIPE) {
#pragma unroll
            for (int n = 0; n < 16; n++) { v[1][n] = mk(raw[n].x, raw[n].y); v[0][n] = mk(raw[n].z, raw[n].w); }
        }
        if (!have) {
            if (interior) poly_issue<D, PPW>(v, src, vbase, t, w * PPW);
            else {
#pragma unroll
                for (int i = 0; i < PPW; i++)
                    if (D % PPW == 0 || w * PPW + i < D) poly_load<D>(v[i], src, vbase, w * PPW + i, t, false, ex);
            }
        }
#pragma unroll
        for (int i = 0; i < PPW; i++)
            if (D % PPW == 0 || w * PPW + i < D) poly_xform_mac<!FIR>(z, v[i], t, ex, tw0, tab1, hr, w * PPW + i);
        long vbn = 0; bool intn = false;
        if constexpr (PIPE) {
            if (it.tile + it.step < it.end) { long u0n; tile_geom(it.tile + it.step, u0n, vbn, intn); }
            have = intn;
        }
        const creg* nb = lane_base(intn, vbn, t);
        // (the next tile's rows [0, SPLIT) before the inverse transform, the rest after it)
        if constexpr (PIPE) poly_issue_raw<D, 0, RR_POLY_PIPE_SPLIT>(raw, nb, t);
        if (w != 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) ex[j * PT + t] = z[j];      // partial sum, register-major, in the wave's own area
        }
        tile_sync<64 * NW>();
        if (w == 0) {
#pragma unroll
            for (int k = 1; k < NW; k++) {
#pragma unroll
                for (int j = 0; j < 16; j++) z[j] = cadd(z[j], lds[k * PLE + j * PT + t]);
            }
            poly_inverse_tab(z, t, ex, tw0, tab1);
            if constexpr (FIR) nf_mark(nf_bad(z[15].x));
            nat_store(z, t, ldsR);
            if constexpr (!FIR) { const bool tame = tile_tame(z); if (t == 0) tame_flag[0] = tame; }
        }
        if constexpr (PIPE) poly_issue_raw<D, RR_POLY_PIPE_SPLIT, 16>(raw, nb, t);
        tile_sync<64 * NW>();
        if constexpr (FIR) {                                 // (launch_fir_poly: mode 2 only)
            poly_store_tile(ldsR, w * PT + t, NW * PT, u0, Sa, a, reinterpret_cast<creg*>(out));
        } else if (a.mode == 0) {
            if (__builtin_amdgcn_readfirstlane(tame_flag[0])) poly_demod_tile<0, true>(ldsR, w * PT + t, NW * PT, u0, Sa, a, out, last_r_in, last_r_out);
            else poly_demod_tile<0>(ldsR, w * PT + t, NW * PT, u0, Sa, a, out, last_r_in, last_r_out);
        } else if (a.mode == 1) poly_demod_tile<1>(ldsR, w * PT + t, NW * PT, u0, Sa, a, out, last_r_in, last_r_out);
        else poly_store_tile(ldsR, w * PT + t, NW * PT, u0, Sa, a, reinterpret_cast<creg*>(out));
        tile_sync<64 * NW>();                            // wave 0's next transforms rewrite the area the others read here
    }
    if constexpr (FIR) nf_finish<cf, cf>();
}

// ---- N channels on one input -------------------------------------------------------------------------------------
template <int D, class SRC>
__global__ __launch_bounds__(512, 1)
void k_fm_multi_poly(SRC src, float* __restrict__ out, long out_stride, long ntiles, const cf* __restrict__ tw,
                     const cf* __restrict__ hreg, int nchan, PolyArgs a, const cf* __restrict__ last_r_in,
                     cf* __restrict__ last_r_out, unsigned long long* __restrict__ dbg, PolyPart part) {
    carry_store<cf>(src, a.carry);
    static_assert(D <= 11, "the D parked spectra + eight exchange areas must fit the CU's LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* park = lds + 8 * PLE;                          // D spectra, register-major [p][16][64]
    creg* tab1 = park + D * PF;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t = threadIdx.x & 63;
    creg* ex = lds + w * PLE;
    creg tw0[15];
    load_twiddles<PLG, 0>(tw0, t, tw);
    if (w == 0) tab1[t] = to_reg(tw[t * (PF / 64)]);
    tile_sync<512>();
    const int Sa = PF - a.Ls;
    const creg* hr = reinterpret_cast<const creg*>(hreg);

    // A unit of work = a channel round: 8 channels of one tile, one per wave.  Workgroup b takes the rounds
    // [part.start[b], part.start[b + 1]) of the launch's ntiles * R (R = ceil(nchan / 8), tile-major) and transforms a tile's
    // input once per run of rounds it holds of that tile (launch_multi_poly_d balances rounds + transforms over the CUs).
    (void)ntiles;
    const int R = part.rounds;
    int iter = 0;
    for (long cr = part.start[blockIdx.x], cr_end = part.start[blockIdx.x + 1]; cr < cr_end; iter++) {
#ifdef RR_FFT_TIMING_BUILD
        // (timing builds: two waves of workgroup 0 are stamped — RR_STAMP_WAVE_A / _B, default 0 and 1)
        unsigned long long* stamps = (dbg && blockIdx.x == 0 && t == 0 && iter == 1 && (w == RR_STAMP_WAVE_A || w == RR_STAMP_WAVE_B))
                                         ? dbg + 16 * (w == RR_STAMP_WAVE_A ? 0 : 1) : nullptr;
#else
        (void)dbg; (void)iter;
#endif
        PSTAMP(0);
        const long tile_ = cr / R;
        const int r0 = (int)(cr - tile_ * R);
        const int r1 = (int)(cr_end - cr < (long)(R - r0) ? r0 + (cr_end - cr) : R);
        cr += r1 - r0;
        const long u0 = a.r_lo + tile_ * Sa;
        const long vbase = (u0 - a.Ls) * D + a.off;
        const bool interior = vbase - (D - 1) >= src.plen && vbase + (long)D * (PF - 1) - src.plen < src.in_len;
        // (one phase per wave up to 1:8; 1:9 ... 1:11 — round 4 — give waves 0 ... D - 9 a second one)
#pragma unroll 1
        for (int ph = w; ph < D; ph += 8) {
            creg v[16];
            poly_load<D>(v, src, vbase, ph, t, interior, ex);
            poly_forward(v, t, ex, tw0, tab1);
#pragma unroll
            for (int j = 0; j < 16; j++) park[(ph * 16 + j) * PT + t] = v[j];
        }
        PSTAMP(1);
        tile_sync<512>();
        PSTAMP(2);
        // (round 3, tools/poly_stamps_multi.py on every wave pair: the two waves of a SIMD do not share it evenly — the older
        //  one, waves 0-3, issues first and runs a channel in 14.7 k clocks, the younger one, waves 4-7, takes 25 k for its
        //  first channel and ~16 k for the others — so waves 0-3 wait 17 k of a tile's 79 k clocks at the final barrier.
        //  Handing the channels out dynamically (a queue in LDS, one ds_add_rtn per channel: the fast waves take more) evens
        //  the waves out and changes nothing: 0.0795 against 0.0786 ms.  Removed.  Starting waves 4-7 2 k / 4 k / 8 k clocks
        //  late (s_sleep) so that the waves' response streams do not coincide costs exactly the delay: +3 / +5.5 / +10 %.)
#pragma unroll 1
        for (int c = 8 * r0 + w; c < nchan && c < 8 * r1; c += 8) {
            const int hsel = (RR_POLY_ABLATE & 32) ? (c & ~4) : (RR_POLY_ABLATE & 64) ? (c & ~7) : c;   // (32 / 64: timing only, waves share a response)
            // Two response buffers: phase p + 1 in flight during phase p.  (Round 3: three and four buffers — with the pass-0
            // twiddles moved from 30 persistent VGPRs to an LDS table, 218 / 250 VGPRs, no spills — measured 0.0638 / worse
            // against 0.0618 ms: the stream is bound by bytes per clock through the vector-memory path, not by its latency.)
            constexpr int NB = 2;
            creg z[16], h[NB][16], x[16];
            // 16 bytes per lane and load: registers 2 jj, 2 jj + 1 of a lane sit side by side in the multi-channel table
            // (PolyTables::build) — the CU's vector-memory path moves 64 lanes x 8 B in the 16 clocks it needs for
            // 64 lanes x 16 B (tools/micro/l1bench.hip: 74 against 135-145 B/ns per CU from an L2-resident table)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const gptr<f32x4> hq = as_global(reinterpret_cast<const f32x4*>(hr + (long)hsel * D * 16 * PT) + t);
            auto load_h = [&](creg* dst, int p, int seed) {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    if constexpr ((RR_POLY_ABLATE & 8) != 0) {
                        dst[2 * jj] = mk(1.0f + jj + seed, (float)(t + c)); dst[2 * jj + 1] = mk(2.0f + jj, (float)(t - c));
                    } else {
                        const f32x4 q = hq[(p * 8 + jj) * PT];
                        dst[2 * jj] = mk(q.x, q.y); dst[2 * jj + 1] = mk(q.z, q.w);
                    }
                }
            };
#pragma unroll
            for (int j = 0; j < 16; j++) z[j] = mk(0.0f, 0.0f);
#pragma unroll
            for (int q = 0; q < NB - 1 && q < D; q++) load_h(h[q], q, 0);
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = park[j * PT + t];
#pragma unroll
            for (int p = 0; p < D; p++) {
                if (p + NB - 1 < D) load_h(h[(p + NB - 1) % NB], p + NB - 1, p);
                // All 16 parked values of the phase are requested before its first product: left to itself the compiler
                // keeps two ds_read2st64_b64 in flight and waits for each pair (0.0778 -> 0.0765 ms, same box; requesting
                // phase p + 1's values too — a second buffer of 32 VGPRs — spills 18 and measures 0.0833; rolling the 16 registers,
                // half a phase ahead, 0.0786).
                if (p > 0) {
#pragma unroll
                    for (int j = 0; j < 16; j++) x[j] = park[(p * 16 + j) * PT + t];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 16; j += 2) cmac2(z[j], x[j], h[p % NB][j], z[j + 1], x[j + 1], h[p % NB][j + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c == 8 * r0 + w) PSTAMP(3);
            poly_inverse(z, t, ex, tw0, tab1);
            wave_fence();
            nat_store(z, t, ex);                         // natural order in the wave's own area
            wave_fence();
            if (c == 8 * r0 + w) PSTAMP(4);
            float* oc = out + (long)c * out_stride;
            if (a.mode == 0) {
                if (tile_tame(z)) poly_demod_tile<0, true>(ex, t, PT, u0, Sa, a, oc, last_r_in + c, last_r_out + c);
                else poly_demod_tile<0>(ex, t, PT, u0, Sa, a, oc, last_r_in + c, last_r_out + c);
            } else poly_demod_tile<1>(ex, t, PT, u0, Sa, a, oc, last_r_in + c, last_r_out + c);
            wave_fence();
            if (c == 8 * r0 + w) PSTAMP(5);
        }
        PSTAMP(6);
        tile_sync<512>();                                // every wave is done with the parked spectra
        PSTAMP(7);
    }
}

// ---- the same with three waves per SIMD (decimations up to 6) -------------------------------------------------------
// 12 waves per workgroup = 12 channels per round.  What pays for the third wave: the pass-0 twiddles in an LDS table instead
// of 30 VGPRs, the responses and the parked spectra in HALF-phase buffers (8 registers each, the next half in flight), so
// that the kernel fits 168 VGPRs; LDS: 12 exchange areas + D parked spectra + the tables = 161.8 KB at D = 6.
constexpr int MW = 12;
template <int D, class SRC>
__global__ __launch_bounds__(64 * MW, 1)
void k_fm_multi_poly12(SRC src, float* __restrict__ out, long out_stride, long ntiles, const cf* __restrict__ tw,
                       const cf* __restrict__ hreg, int nchan, PolyArgs a, const cf* __restrict__ last_r_in,
                       cf* __restrict__ last_r_out, PolyPart part) {
    carry_store<cf>(src, a.carry);
    static_assert(D <= 6, "LDS: 12 exchange areas + D parked spectra");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    creg* lds = reinterpret_cast<creg*>(smem_raw);
    creg* park = lds + MW * PLE;                         // D spectra, register-major [p][16][64]
    creg* tab1 = park + D * PF;
    creg* tw0tab = tab1 + 64;                            // pass-0 twiddles of lane t, [15][64]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), t = threadIdx.x & 63;   // (w in an SGPR: wave-uniform index arithmetic)
    creg* ex = lds + w * PLE;
    if (w == 0) {
        creg twr[15];
        load_twiddles<PLG, 0>(twr, t, tw);
#pragma unroll
        for (int k = 0; k < 15; k++) tw0tab[k * PT + t] = twr[k];
        tab1[t] = to_reg(tw[t * (PF / 64)]);
    }
    tile_sync<64 * MW>();
    const int Sa = PF - a.Ls;
    const creg* hr = reinterpret_cast<const creg*>(hreg);
    (void)ntiles;
    const int R = part.rounds;
    for (long cr = part.start[blockIdx.x], cr_end = part.start[blockIdx.x + 1]; cr < cr_end;) {
        const long tile_ = cr / R;
        const int r0 = (int)(cr - tile_ * R);
        const int r1 = (int)(cr_end - cr < (long)(R - r0) ? r0 + (cr_end - cr) : R);
        cr += r1 - r0;
        const long u0 = a.r_lo + tile_ * Sa;
        const long vbase = (u0 - a.Ls) * D + a.off;
        const bool interior = vbase - (D - 1) >= src.plen && vbase + (long)D * (PF - 1) - src.plen < src.in_len;
        // (requesting the NEXT run's tile before the demodulation of a wave's last channel — the k_fm_chain_polyw scheme, two
        //  instances of the channel body — puts the kernel over its 168 registers: 7-14 spilled, 0.0588 against 0.0562 ms)
        RR_MARK("forward");
        if (w < D) {
            creg v[16];
            poly_load<D>(v, src, vbase, w, t, interior, ex);
            poly_forward_tab(v, t, ex, tw0tab, tab1);
#pragma unroll
            for (int j = 0; j < 16; j++) park[(w * 16 + j) * PT + t] = v[j];
        }
        tile_sync<64 * MW>();
        RR_MARK("channels");
#pragma unroll 1
        for (int c = MW * r0 + w; c < nchan && c < MW * r1; c += MW) {
            RR_MARK("products");
            creg z[16], h[2][8], x[8];
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const gptr<f32x4> hq = as_global(reinterpret_cast<const f32x4*>(hr + (long)c * D * 16 * PT) + t);
            auto load_h = [&](creg* dst, int k) {              // half-phase k: registers 8 (k & 1) .. + 7 of phase k / 2
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const f32x4 q = hq[(k * 4 + jj) * PT];
                    dst[2 * jj] = mk(q.x, q.y); dst[2 * jj + 1] = mk(q.z, q.w);
                }
            };
#pragma unroll
            for (int j = 0; j < 16; j++) z[j] = mk(0.0f, 0.0f);
            load_h(h[0], 0);
#pragma unroll
            for (int k = 0; k < 2 * D; k++) {
                if (k + 1 < 2 * D) load_h(h[(k + 1) & 1], k + 1);
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = park[(k * 8 + j) * PT + t];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; j += 2)
                    cmac2(z[8 * (k & 1) + j], x[j], h[k & 1][j], z[8 * (k & 1) + j + 1], x[j + 1], h[k & 1][j + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            RR_MARK("inverse");
            poly_inverse_tab(z, t, ex, tw0tab, tab1);
            wave_fence();
            nat_store(z, t, ex);
            wave_fence();
            RR_MARK("demod");
            float* oc = out + (long)c * out_stride;
            if (a.mode == 0) {
                if (tile_tame(z)) poly_demod_tile<0, true>(ex, t, PT, u0, Sa, a, oc, last_r_in + c, last_r_out + c);
                else poly_demod_tile<0>(ex, t, PT, u0, Sa, a, oc, last_r_in + c, last_r_out + c);
            } else poly_demod_tile<1>(ex, t, PT, u0, Sa, a, oc, last_r_in + c, last_r_out + c);
            wave_fence();
            RR_MARK("channel_end");
        }
        RR_MARK("tile_end");
        tile_sync<64 * MW>();                            // every wave is done with the parked spectra
    }
}

// ---- launchers -----------------------------------------------------------------------------------------------------
bool fm_poly_supported(long I, long D, int L, bool multi) {
    if (I != 1) return false;
    const bool dok = multi ? (D >= 2 && D <= 11) : (D >= 2 && D <= 16);
    if (!dok || L < 1) return false;
    const long Ls = (L + D - 1) / D;
    // taps per phase: a tile yields 1024 - Ls outputs.  One chain: up to 768 (beyond, the 8192-point split tiles win:
    // tools/poly_long_probe.py); N channels: up to 928 — what a long filter falls back to there is one chain per channel
    // (0.95 ms per 2.4e6 samples x 32 channels at 5000 taps against ~0.3 for these tiles at 96 outputs each)
    return Ls <= (multi ? 928 : 768);
}
int fm_poly_bin(int j, int t) {                          // frequency bin held by register j of lane t after the forward transform
    using G = PassGeom<PLG, 2>;
    return bin_of_pos<PLG>(G::pos(t + G::T * (j / G::R), j % G::R));
}

static PolyArgs poly_args(const FmChainArgs& h, int L) {
    PolyArgs a;
    a.off = (long)(L - 1) - h.A;
    a.r_lo = h.r_lo; a.r_hi = h.r_hi; a.o_base = h.o_base;
    a.Ls = (int)((L + h.D - 1) / h.D);
    a.gain = h.gain; a.mode = h.mode; a.carry = h.carry;
    return a;
}

// (1:9 ... 1:12 on five / six waves per workgroup were tried: two workgroups per CU, 0.092-0.133 against 0.081-0.096 ms on the two-wave kernel)
template <int D> struct ChainWaves { static constexpr int NW = D == 6 ? 3 : D == 4 ? 2 : D == 8 ? 4 : D == 3 ? 3 : D == 2 ? 2 : D == 5 ? 3 : D == 7 ? 4 : 0; };
template <int D, class SRC>
static void launch_chain_poly_d(SRC src, float* out, int L, const cf* tw, const cf* hreg, const FmChainArgs& h, const cf* last_in,
                                cf* last_out, hipStream_t s) {
    const PolyArgs a = poly_args(h, L);
    const long Sa = PF - a.Ls, nr = a.r_hi - a.r_lo;
    if (nr <= 0) { launch_carry(src, h.carry, s); return; }
    const long ntiles = (nr + Sa - 1) / Sa;
    // (nan_fix.hpp, mode 2 = launch_fir_poly: r_lo = o_base = 0, tile k owns the outputs [k Sa, k Sa + Sa))
    NanFixCtx nfx{};
    if constexpr (std::is_same<SRC, VSrc<cf>>::value) nfx = nanfix_ctx(h.mode == 2 ? h.fx : NanFix{}, src, out, Sa, 1, a.r_hi - a.o_base, ntiles);
#if RR_POLY_CHAIN_W3
    // (1:5 and 1:7 from RTL-SDR bytes keep the two-wave kernel: their wave-dependent phase count spills 48-76 registers there)
    constexpr int NWsel = (std::is_same<SRC, VSrcIQ8>::value && (D % 2 == 1) && D >= 5) ? 0 : ChainWaves<D>::NW;
    if constexpr (NWsel != 0) {
        constexpr int NW = NWsel;
        const size_t smemw = sizeof(cf) * (NW * PLE + 64 + 15 * PT + 1);
        long gridw = grid_for_tiles(k_fm_chain_polyw<D, NW, SRC>, 64 * NW, smemw, ntiles);
        if (ntiles > gridw && ntiles < 12 * gridw) gridw = std::min(ntiles, (long)RR_POLY_OVERSUB * gridw);
        if constexpr (std::is_same<SRC, VSrc<cf>>::value) {
            if (h.mode == 2) {
                hipLaunchKernelGGL((k_fm_chain_polyw<D, NW, SRC, true>), dim3((unsigned)gridw), dim3(64 * NW), smemw, s, nfx, src, out, ntiles, tw, hreg, a,
                                   last_in, last_out);
                RR_HIP(hipGetLastError());
                return;
            }
        }
        hipLaunchKernelGGL((k_fm_chain_polyw<D, NW, SRC>), dim3((unsigned)gridw), dim3(64 * NW), smemw, s, nfx, src, out, ntiles, tw, hreg, a,
                           last_in, last_out);
        RR_HIP(hipGetLastError());
        return;
    }
#endif
    const size_t smem = sizeof(cf) * (3 * PLE + 64 + (RR_POLY_CHAIN_TWLDS ? 15 * PT : 0) + 1);
    long grid = grid_for_tiles(k_fm_chain_poly<D, SRC>, 128, smem, ntiles);
    // A few tiles per resident workgroup (configs[2]: 4228 tiles on 1024 slots = 4.1) end in a round where most of the
    // chip waits for the workgroups with one tile more.  Launching 3x the resident workgroups lets the hardware dispatcher
    // hand the tiles out as slots free up (tools/percu_sweep.sh: 0.0685 -> 0.063 ms; 1 tile per workgroup 0.0645; with 17
    // tiles per slot, full_chain_fused, the persistent grid stays ahead: 0.319 vs 0.330).
    if (ntiles > grid && ntiles < 12 * grid) grid = std::min(ntiles, (long)RR_POLY_OVERSUB * grid);
    if constexpr (std::is_same<SRC, VSrc<cf>>::value) {
        if (h.mode == 2) {
            hipLaunchKernelGGL((k_fm_chain_poly<D, SRC, true>), dim3((unsigned)grid), dim3(128), smem, s, nfx, src, out, ntiles, tw, hreg, a, last_in, last_out,
                               fft_stamp_buffer());
            RR_HIP(hipGetLastError());
            return;
        }
    }
    hipLaunchKernelGGL((k_fm_chain_poly<D, SRC>), dim3((unsigned)grid), dim3(128), smem, s, nfx, src, out, ntiles, tw, hreg, a, last_in, last_out,
                       fft_stamp_buffer());
    RR_HIP(hipGetLastError());
}
template <class SRC>
static void launch_chain_poly_t(SRC src, float* out, int L, const cf* tw, const cf* hreg, const FmChainArgs& h, const cf* last_in,
                                cf* last_out, hipStream_t s) {
    switch (h.D) {
#define RR_POLY_CASE(DV) case DV: launch_chain_poly_d<DV>(src, out, L, tw, hreg, h, last_in, last_out, s); break
    RR_POLY_CASE(2); RR_POLY_CASE(3); RR_POLY_CASE(4); RR_POLY_CASE(5); RR_POLY_CASE(6); RR_POLY_CASE(7); RR_POLY_CASE(8);
    RR_POLY_CASE(9); RR_POLY_CASE(10); RR_POLY_CASE(11); RR_POLY_CASE(12); RR_POLY_CASE(13); RR_POLY_CASE(14); RR_POLY_CASE(15); RR_POLY_CASE(16);
#undef RR_POLY_CASE
    default: throw Error("fm_chain_poly: unsupported decimation");
    }
}
void launch_fm_chain_poly(VSrc<cf> src, float* out, int L, const cf* tw, const cf* hreg, const FmChainArgs& a, const cf* last_in,
                          cf* last_out, hipStream_t s) {
    launch_chain_poly_t(src, out, L, tw, hreg, a, last_in, last_out, s);
}
void launch_fm_chain_poly_iq8(VSrcIQ8 src, float* out, int L, const cf* tw, const cf* hreg, const FmChainArgs& a, const cf* last_in,
                              cf* last_out, hipStream_t s) {
    launch_chain_poly_t(src, out, L, tw, hreg, a, last_in, last_out, s);
}
// Decimating FirFilter<Complex> on the same tiles: out[m] = sum_k t[k] x[m D + L - 1 - k], m < n_out (Fir::filter_n, fir.rs:181-189;
// "valid" mode: the window itself holds the L - 1 samples of history) = the chain's r[u] with the stream origin at V[L - 1]
// and the samples stored instead of demodulated.
void launch_fir_poly(VSrc<cf> src, cf* out, long n_out, int L, int D, const cf* tw, const cf* hreg, hipStream_t s, NanFix fx) {
    FmChainArgs h{};
    h.A = 0; h.n_y = n_out * (long)D; h.r_lo = 0; h.r_hi = n_out; h.o_base = 0; h.I = 1; h.D = D; h.gain = 1.0f; h.mode = 2;
    h.fx = fx;
    launch_chain_poly_t(src, reinterpret_cast<float*>(out), L, tw, hreg, h, nullptr, nullptr, s);
}

// One workgroup per CU (the kernel's shared memory allows no more), each with a CONTIGUOUS run of channel rounds: a run costs
// wC per round + wF per tile it touches (that tile's input is loaded and transformed again).  Whole tiles handed out one by
// one leave most of the chip idle in the last wave of tiles — configs[3]'s 423 tiles on 256 CUs take two tile times, 2 x 78.5 k
// clocks, for 1.65 tiles' worth of work per CU — so the runs are cut where the LARGEST cost is smallest: binary search on the
// bound, greedy fill under it (optimal for contiguous runs; never worse than whole tiles; for fewer tiles than CUs it is the
// split of a tile's channels over several workgroups that ring-sized windows need: 90 tiles -> 180 runs of two rounds).
static PolyPart poly_partition(long ntiles, int R, int G, int wC, int wF) {
    static std::mutex mu;
    static std::map<std::tuple<long, int, int, int, int>, PolyPart> cache;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(ntiles, R, G, wC, wF);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (cache.size() >= 64) cache.clear();
    const long total = ntiles * R;
    const long tile_cost = wF + (long)R * wC;
    // fills workgroups greedily under the bound T; returns the rounds placed (== total when T is feasible)
    auto fill = [&](long T, int* start) {
        long cr = 0;
        for (int b = 0; b < G; b++) {
            if (start) start[b] = (int)cr;
            long left = T;
            while (cr < total) {
                const long o = cr % R;
                if (o == 0 && left >= tile_cost) {             // whole tiles
                    const long m = std::min(left / tile_cost, (total - cr) / R);
                    if (m > 0) { cr += m * R; left -= m * tile_cost; continue; }
                }
                const long k = std::min<long>(std::min<long>(R - o, total - cr), (left - wF) / wC);
                if (k <= 0) break;
                cr += k; left -= wF + k * wC;
            }
        }
        if (start) start[G] = (int)cr;
        return cr;
    };
    const long m = (total + G - 1) / G;
    long lo = wF + wC, hi = m * wC + ((m + R - 2) / R + 1) * wF;   // hi: every window of m rounds fits, so the fill succeeds
    while (lo < hi) {
        const long mid = (lo + hi) / 2;
        if (fill(mid, nullptr) >= total) hi = mid; else lo = mid + 1;
    }
    PolyPart p{};
    p.rounds = R;
    p.cost = (int)std::min<long>(hi, 0x7fffffffL);
    if (fill(hi, p.start) != total) throw Error("fm_multi_poly: internal error (work partition)");
    for (int b = G + 1; b <= POLY_PART_MAX; b++) p.start[b] = (int)total;
    return cache.emplace(key, p).first->second;
}

template <int D, class SRC>
static void launch_multi_poly_d(SRC src, float* out, long out_stride, int L, const cf* tw, const cf* hreg, int nchan,
                                const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    const PolyArgs a = poly_args(h, L);
    const long Sa = PF - a.Ls, nr = a.r_hi - a.r_lo;
    if (nr <= 0) { launch_carry(src, h.carry, s); return; }
    const long ntiles = (nr + Sa - 1) / Sa;
    const size_t smem = sizeof(cf) * (8 * PLE + D * PF + 64);
    // Two kernels: 8 waves per workgroup (any decimation up to 8) and 12 (three per SIMD, decimations up to 6).  A round of
    // 12 channels costs 1.28x a round of 8 (24 M samples, 32 channels: 567 against 589 us per call) and cuts a launch into
    // coarser pieces — ring-sized windows run faster on 8 waves (512 k samples: 24.5 against 29.6 us), one-second windows on
    // 12 (2.4 M: 0.0568 against 0.0622 ms).  The choice is the smaller predicted cost of the launch's largest run.
    int wC = 16 + 5 * D / 2, wF = 14;
#ifdef RR_MEASURE_KNOBS
    if (const char* e = getenv("RR_POLY_WF")) wF = atoi(e);                          // measurement builds only
    if (const char* e = getenv("RR_POLY_WC")) wC = atoi(e);
#endif
    const int R = std::max(1, (nchan + 7) / 8);
    if (ntiles * R > 0x7fffffffL) throw Error("fm_multi_poly: window too long");
    const long G = std::min<long>(std::min<long>(device_cu_count(), POLY_PART_MAX), ntiles * R);
    const PolyPart part = poly_partition(ntiles, R, (int)G, wC, wF);
    if constexpr (D <= 6) {
        const int R12 = (nchan + MW - 1) / MW;
        const long G12 = std::min<long>(std::min<long>(device_cu_count(), POLY_PART_MAX), ntiles * R12);
        const int wC12 = (wC * 39 + 16) / 32;        // (1.22x: fitted to the per-call times of both kernels, 1 M ... 24 M samples)
        const PolyPart part12 = poly_partition(ntiles, R12, (int)G12, wC12, wF);
        const bool use12 = h.multi_waves == 12 || (h.multi_waves == 0 && nchan > 8 && part12.cost < part.cost);
        if (use12) {
            const size_t smem12 = sizeof(cf) * (MW * PLE + D * PF + 64 + 15 * PT);
            (void)grid_for_tiles(k_fm_multi_poly12<D, SRC>, 64 * MW, smem12, G12);               // (sets the shared-memory attribute once)
            hipLaunchKernelGGL((k_fm_multi_poly12<D, SRC>), dim3((unsigned)G12), dim3(64 * MW), smem12, s, src, out, out_stride, ntiles, tw,
                               hreg, nchan, a, last_in, last_out, part12);
            RR_HIP(hipGetLastError());
            return;
        }
    }
    (void)grid_for_tiles(k_fm_multi_poly<D, SRC>, 512, smem, G);                     // (sets the shared-memory attribute once)
    hipLaunchKernelGGL((k_fm_multi_poly<D, SRC>), dim3((unsigned)G), dim3(512), smem, s, src, out, out_stride, ntiles, tw, hreg,
                       nchan, a, last_in, last_out, fft_stamp_buffer(), part);
    RR_HIP(hipGetLastError());
}
template <class SRC>
static void launch_multi_poly_t(SRC src, float* out, long out_stride, int L, const cf* tw, const cf* hreg, int nchan,
                                const FmChainArgs& h, const cf* last_in, cf* last_out, hipStream_t s) {
    switch (h.D) {
#define RR_POLY_CASE(DV) case DV: launch_multi_poly_d<DV>(src, out, out_stride, L, tw, hreg, nchan, h, last_in, last_out, s); break
    RR_POLY_CASE(2); RR_POLY_CASE(3); RR_POLY_CASE(4); RR_POLY_CASE(5); RR_POLY_CASE(6); RR_POLY_CASE(7); RR_POLY_CASE(8);
    RR_POLY_CASE(9); RR_POLY_CASE(10); RR_POLY_CASE(11);
#undef RR_POLY_CASE
    default: throw Error("fm_multi_poly: unsupported decimation");
    }
}
void launch_fm_multi_poly(VSrc<cf> src, float* out, long out_stride, int L, const cf* tw, const cf* hreg, int nchan,
                          const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_multi_poly_t(src, out, out_stride, L, tw, hreg, nchan, a, last_in, last_out, s);
}
void launch_fm_multi_poly_iq8(VSrcIQ8 src, float* out, long out_stride, int L, const cf* tw, const cf* hreg, int nchan,
                              const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s) {
    launch_multi_poly_t(src, out, out_stride, L, tw, hreg, nchan, a, last_in, last_out, s);
}

}  // namespace rr
