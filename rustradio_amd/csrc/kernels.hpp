// kernels.hpp — host-callable launchers of the HIP kernels (one TU per kernel family).
#pragma once
#include "common.hpp"
#include "nan_fix.hpp"

namespace rr {

// ---- kernels_fft.hip ---------------------------------------------------------------
// Overlap-save FFT filter.  xx = virtual stream whose index 0 is the first of the
// (L-1) history samples.  Writes y[n] = sum_k t[k] * x[n - k], n < n_out, where x[i] = xx[i + L - 1].
//   log2f      : internal tile size F = 2^log2f (10..14), F >= 2*(L-1) recommended
//   tw         : device table of w_F^k, k < F
//   hpos       : device table of H (scaled by 1/F) in digit-reversed position order
bool fftfilt_supported(int log2f);
int fft_read_stamps(unsigned long long* host16);   // measurement builds only (else returns 0)
// (carry: the caller's carry-state update, done by this launch — every launcher below that takes one falls back to a
//  separate copy kernel when it has nothing to launch)
// (fx: FirFilter on these tiles — non-finite input samples get the reference's locality, nan_fix.hpp; default: none)
void launch_fftfilt_os(int log2f, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw,
                       const cf* hpos, hipStream_t s, CarryOut carry = {}, NanFix fx = {});

// The same filter keeping every d-th output: out[m] = y[m d], m < n_out (tiles of 1024..4096 points, d <= 4096) —
// the decimating FirFilter (fir.rs:181-189) on overlap-save tiles.
void launch_fftfilt_deci(int log2f, VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw, const cf* hpos,
                         hipStream_t s, NanFix fx = {});

// Real stream, real taps (hpos from Complex(t, 0)): out[m] = y[m d], y[n] = sum_k t[k] xx[n + L - 1 - k]; two
// overlap-save segments ride in the real / imaginary lanes of one Complex tile (tiles of 1024..4096 points).
void launch_fftfilt_real_hilbert(int log2f, VSrc<float> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos, hipStream_t s,
                                 CarryOut carry, NanFix fx = {});
void launch_fftfilt_real(int log2f, VSrc<float> src, float* out, long n_out, int L, int d, const cf* tw, const cf* hpos,
                         hipStream_t s, CarryOut carry = {}, NanFix fx = {});

// FftFilterFloat -> RationalResampler(I:D) -> MultiplyConst fused on the real-stream tiles: out[m - r_lo] = scale *
// y[floor(m D / I)] for the resampled samples m in [r_lo, r_hi) whose source lies in this call's y[A .. A + n_y).
struct AudioChainArgs {
    long A, n_y, r_lo, r_hi, I, D;
    float scale;
    CarryOut carry;    // the block's carry-state update, written by this launch (common.hpp)
};
void launch_audio_chain(int log2f, VSrc<float> src, float* out, int L, const cf* tw, const cf* hpos, const AudioChainArgs& a,
                        hipStream_t s);

// Decimation by D = F / 256 (4 / 8 / 16 on tiles of 1024 / 2048 / 4096 points) with a pruned inverse transform
// (k_fftfilt_prune): out[m] = y[m D], m < n_out.  Tables (see the kernel): hpos2 = H in position order with the
// factors w_D^(-c k3) w_16D^(-c k2), c = (L - 1) % D, folded in; twb[n2 * 16 + k1] = exp(+2 pi i k1 (n2 D + c) / F).
int prune_log2f_for_deci(int d);                 // 0 when d is not 4 / 8 / 16
bool prune_split(size_t d, size_t& D, size_t& sub);   // d = D * sub, D in {16, 8, 4} (the pruned tile), sub <= 64
void launch_fftfilt_prune_c32(int log2f, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos2,
                              const cf* twb, hipStream_t s, int sub = 1, NanFix fx = {});
// real stream, real taps, f32 output (decimating FirFilter<Float>)
void launch_fftfilt_prune_f32(int log2f, VSrc<float> src, float* out, long n_out, int L, const cf* tw, const cf* hpos2,
                              const cf* twb, hipStream_t s, int sub = 1, NanFix fx = {});
// real stream, Complex taps t = Gr + i Gi: hpos2r / hpos2i from the real tap sets Gr / Gi; Complex output
void launch_fftfilt_prune_real(int log2f, VSrc<float> src, cf* out, long n_out, int L, const cf* tw, const cf* hpos2r,
                               const cf* hpos2i, const cf* twb, hipStream_t s, CarryOut carry = {}, int sub = 1, NanFix fx = {});

// Even decimations on 2048-point tiles with the half-size inverse (k_fftfilt_half): out[m] = y[m d], m < n_out.
// tw = w_2048^k, tw_half = w_1024^k, hpos = H / F in the 2048-point position order.
bool fftfilt_half_supported(int L, long d);
void launch_fftfilt_half(VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw, const cf* tw_half, const cf* hpos,
                         hipStream_t s, NanFix fx = {});

// The same filter with an 8192 / 16384-point tile built from nsub = 2 / 4 sub-transforms of 4096 points
// (k_fftfilt_split).  tw4096: w_4096^k;  wk[t] = w_F^t, t < 256;  hs[r][p] = H[nsub bin(p) + r] / F with
// bin(p) = fftfilt_split_bin(p), the bin at position p of the 4096-point spectrum layout.
void launch_fftfilt_split(int nsub, VSrc<cf> src, cf* out, long n_out, int L, const cf* tw4096, const cf* hs, const cf* wk,
                          hipStream_t s, CarryOut carry = {}, NanFix fx = {});
void launch_fftfilt_split_deci(int nsub, VSrc<cf> src, cf* out, long n_out, int L, int d, const cf* tw4096, const cf* hs,
                               const cf* wk, hipStream_t s, NanFix fx = {});   // out[m] = y[m d], m < n_out
int fftfilt_split_bin(int p);

// FftStream: out = forward unnormalised FFT of each of `nframes` consecutive 2^log2n-point frames
// (log2n 1..14), natural bin order; tw = device table of w_N^k, k < N.
// (tw4096 = w_4096^k: when given, 8192 / 16384-point frames run as 2 / 4 sub-transforms of 4096 points)
void launch_fft_frames(int log2n, const cf* in, cf* out, long nframes, const cf* tw, const cf* tw4096, hipStream_t s);

// ... of any other size N, 2 N - 1 <= 2^log2m <= 4096, by Bluestein's chirp-z convolution on the filter tile:
// chirp[n] = exp(-i pi n^2 / N), n < N;  hpos = FFT_M(b) / M in position order, b[m] = conj(chirp[|m|]) wrapped mod M.
void launch_fft_bluestein(int log2m, const cf* in, cf* out, long nframes, int N, const cf* tw, const cf* hpos,
                          const cf* chirp, hipStream_t s);

// Fused FftFilter -> RationalResampler -> QuadratureDemod over the same virtual stream.
struct FmChainArgs {
    long A;            // filtered samples emitted before this call
    long n_y;          // filtered samples covered by this call (multiple of the reference nsamples)
    long r_lo, r_hi;   // resampled samples r[m] = y[floor(m*D/I)] whose source lies in [A, A+n_y)
    long o_base;       // demodulated samples emitted before this call
    long I, D;         // reduced interp / deci
    float gain;
    int mode;          // RR_ATAN2_*
    CarryOut carry;    // the block's carry-state update (new prefix), written by this launch (common.hpp)
    int multi_waves = 0;   // multi-channel decimate-first kernel: 0 = by predicted cost, 8 / 12 = that many waves per workgroup
    NanFix fx;             // mode 2 only (launch_fir_poly): nan_fix.hpp
};
// out[(u-1) - o_base] = gain * atan2(conj(r[u-1]) r[u]) for u in [max(r_lo,1), r_hi); r[r_lo-1] is
// *last_in (previous call), r[r_hi-1] is written to *last_out.
void launch_fm_chain(int log2f, VSrc<cf> src, float* out, int L, const cf* tw, const cf* hpos,
                     const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// ... with the RtlSdrDecode conversion fused in front (window = u8 I/Q pairs, carried prefix Complex).
void launch_fm_chain_iq8(int log2f, VSrcIQ8 src, float* out, int L, const cf* tw, const cf* hpos,
                         const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// ... on 8192 / 16384-point tiles built from nsub = 2 / 4 sub-transforms (tables as launch_fftfilt_split)
void launch_fm_chain_split(int nsub, VSrc<cf> src, float* out, int L, const cf* tw4096, const cf* hs, const cf* wk,
                           const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);
void launch_fm_chain_split_iq8(int nsub, VSrcIQ8 src, float* out, int L, const cf* tw4096, const cf* hs, const cf* wk,
                               const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// The same for `nchan` channels that share the input: hpos_all = [nchan][F] frequency responses,
// channel c writes out + c*out_stride and carries last_in[c] / last_out[c].  3-pass tiles (F <= 4096).
bool fm_multi_supported(int log2f);
void launch_fm_multi(int log2f, VSrc<cf> src, float* out, long out_stride, int L, const cf* tw, const cf* hpos_all,
                     int nchan, const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// ... with the RtlSdrDecode conversion fused in front (window = u8 I/Q pairs)
void launch_fm_multi_iq8(int log2f, VSrcIQ8 src, float* out, long out_stride, int L, const cf* tw, const cf* hpos_all,
                         int nchan, const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);
void launch_fm_multi_half_iq8(int log2f, VSrcIQ8 src, float* out, long out_stride, int L, const cf* tw, const cf* tw_half,
                              const cf* hpos_all, int nchan, const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);
// interp 1 and an even decimation on 2048-point tiles: per channel a folded 1024-point inverse on one wave
// (k_fm_multi_half); tw_half = w_(F/2)^k.
bool fm_multi_half_supported(int log2f, long I, long D, int L);
void launch_fm_multi_half(int log2f, VSrc<cf> src, float* out, long out_stride, int L, const cf* tw, const cf* tw_half,
                          const cf* hpos_all, int nchan, const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// the single fused chain on the same half-size inverses (2048-point tiles; fm_multi_half_supported(11, I, D, L))
void launch_fm_chain_half(VSrc<cf> src, float* out, int L, const cf* tw, const cf* tw_half, const cf* hpos,
                          const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);
void launch_fm_chain_half_iq8(VSrcIQ8 src, float* out, int L, const cf* tw, const cf* tw_half, const cf* hpos,
                              const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// ---- kernels_poly.hip: fused FM chains with an integer decimation on decimate-first (polyphase) tiles ----------------
// tw = w_1024^k; hreg = per channel and phase p the response FFT_1024(t[D j + p]) / 1024, register-major:
// hreg[((c D + p) 16 + j) 64 + lane] = H_{c,p}[fm_poly_bin(j, lane)].  a.I must be 1.
bool fm_poly_supported(long I, long D, int L, bool multi);
int fm_poly_bin(int j, int lane);
void launch_fm_chain_poly(VSrc<cf> src, float* out, int L, const cf* tw, const cf* hreg, const FmChainArgs& a, const cf* last_in,
                          cf* last_out, hipStream_t s);
void launch_fm_chain_poly_iq8(VSrcIQ8 src, float* out, int L, const cf* tw, const cf* hreg, const FmChainArgs& a, const cf* last_in,
                              cf* last_out, hipStream_t s);
// decimating FirFilter<Complex> (deci = D) on the decimate-first tiles: out[m] = sum_k t[k] x[m D + L - 1 - k]
void launch_fir_poly(VSrc<cf> src, cf* out, long n_out, int L, int D, const cf* tw, const cf* hreg, hipStream_t s, NanFix fx = {});
void launch_fm_multi_poly(VSrc<cf> src, float* out, long out_stride, int L, const cf* tw, const cf* hreg, int nchan,
                          const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);
void launch_fm_multi_poly_iq8(VSrcIQ8 src, float* out, long out_stride, int L, const cf* tw, const cf* hreg, int nchan,
                              const FmChainArgs& a, const cf* last_in, cf* last_out, hipStream_t s);

// ---- kernels_fir.hip ---------------------------------------------------------------
struct FirPlan {             // host-prepared polyphase tap table
    int L = 0, d = 1;
    int qpad = 0;            // taps per phase, padded to a multiple of 8
    bool complex_taps = false;
    int cfg = -1;            // >= 0: force tile shape cfg (rr_build_opts.fir_cfg, parity tests of every shape)
};
// y[m] = sum_k rev[k] * x[m*d + k], m < n_out  (rev = reversed taps), x = virtual stream.
// tp = device polyphase table [d][qpad] (float if real taps else cf), rev = device reversed taps.
bool fir_direct_has_tile(const FirPlan& pl, size_t es_in, size_t es_out);   // false: only the slow one-thread-per-output fallback
void launch_fir_c32(const FirPlan& pl, const void* tp, const void* rev, VSrc<cf> src, cf* out,
                    long n_out, hipStream_t s, NanFix fx = {});
void launch_fir_f32(const FirPlan& pl, const float* tp, const float* rev, VSrc<float> src,
                    float* out, long n_out, hipStream_t s, NanFix fx = {});
// Hilbert: out[k] = (xp[k + L/2], sum_j rev[j] xp[k + j]), xp = virtual stream (history ++ in).
// real samples, Complex taps, Complex output (the composite Hilbert -> FirFilter filter)
void launch_fir_f32c(const FirPlan& pl, const cf* tp, const cf* rev, VSrc<float> src, cf* out, long n_out,
                     hipStream_t s, NanFix fx = {});
void launch_hilbert(const FirPlan& pl, const float* tp, const float* rev, VSrc<float> src, cf* out,
                    long n_out, hipStream_t s, NanFix fx = {});
// Hilbert with the zero taps skipped: hq[q] = rev[2q + par] (Q entries, padded to a multiple of 8).
// Returns false if the shape is not covered (use launch_hilbert).
constexpr int HILBERT_MAX_GRID = 8192;        // workgroups k_hilbert is launched with at most (NanFix::wgflags has one word each)
bool launch_hilbert_skip(int L, int par, int Q, const float* hq, VSrc<float> src, cf* out, long n_out, hipStream_t s, NanFix fx = {});
// y[m] *= phase0 * step^(m0 + m) evaluated in f64 (RR_ROT_MODEL)
void launch_rotate_model(cf* y, long n, double p0x, double p0y, double sx, double sy, long m0,
                         hipStream_t s);
// RR_ROT_REPLAY: phase_{i+1} = phase_i * step in f32 (the reference's recurrence) from *state: nskip steps without a store,
// then ring[(pos0 + i) & mask] = phase_i for i < n (mask = ring capacity - 1, a power of two); *state <- the phase after them
void launch_rotor_replay(cf* state, float stx, float sty, cf* ring, long pos0, long mask, long nskip, long n, hipStream_t s);
// y[m] *= ring[(pos0 + m) & mask]
void launch_rotate_table(cf* y, long n, const cf* ring, long pos0, long mask, hipStream_t s);

// ---- kernels_misc.hip --------------------------------------------------------------
// out[r + m] = in[floor((m*D - c0) / I)], m < n_gather; out[0..r) = *pending
void launch_resample(const void* in, void* out, size_t es, long r, const void* pending,
                     long n_gather, long I, long D, long c0, hipStream_t s);
// out[n] = gain * atan2(Im z, Re z), z = conj(x[n]) x[n+1], n < n_out
void launch_quaddemod(const cf* in, float* out, long n_out, float gain, int mode, hipStream_t s);
void launch_mulconst_f32(const float* in, float* out, long n, float v, hipStream_t s);
void launch_mulconst_c32(const cf* in, cf* out, long n, float vr, float vi, hipStream_t s);
// FastFM over src = (q2, q1) || window: n outputs
void launch_fastfm(VSrc<cf> src, float* out, long n, hipStream_t s);
// RtlSdrDecode: out[i] = ((in[2i] - 127) * 0.008, (in[2i+1] - 127) * 0.008)
void launch_rtlsdr_decode(const unsigned char* in, cf* out, long n_out, hipStream_t s);
// dst[i] = src.load(v0 + i), i < n   (carry-state update)
void launch_vcopy_c32(VSrc<cf> src, long v0, cf* dst, long n, hipStream_t s);
void launch_vcopy_iq8(VSrcIQ8 src, long v0, cf* dst, long n, hipStream_t s);   // decoding copy
void launch_vcopy_f32(VSrc<float> src, long v0, float* dst, long n, hipStream_t s);
// Hilbert on transform tiles: every non-finite output of a poisoned tile recomputed with the reference's fold (kernels_misc.hip)
void launch_hilbert_refold_nonfinite(VSrc<float> src, cf* out, long n_out, long P, int L, const float* rev, hipStream_t s);
// FftFilter / FftFilterFloat: non-finite samples poison the REFERENCE's blocks, not the GPU's tiles (kernels_misc.hip)
void launch_ref_blocks_nonfinite(VSrc<cf> src, cf* out, long n_out, long S, long P, long hist, int L, int front, const cf* rev, int* tail, int seq, bool force0, hipStream_t s);
void launch_ref_blocks_nonfinite(VSrc<float> src, float* out, long n_out, long S, long P, long hist, int L, int front, const float* rev, int* tail, int seq, bool force0, hipStream_t s);
// the fused FM / audio chains: the same through the resampler's index map and the demodulator's pair (kernels_misc.hip); slots = int[6]
void launch_chain_blocks_nonfinite(VSrc<cf> src, float* out, long out_stride, int nchan, const FmChainArgs& a, long S, long hist, long P,
                                   int L, int front, const cf* rev, long rev_stride, const cf* last_in, cf* last_out, int* slots, int seq, int force,
                                   hipStream_t s);
void launch_chain_blocks_nonfinite(VSrc<float> src, float* out, const AudioChainArgs& a, long S, long hist, long P, int L,
                                   const float* rev, int* slots, int seq, hipStream_t s);
// a CarryOut as its own launch (calls without a main kernel; launchers with nothing to launch)
void launch_carry(VSrc<cf> src, const CarryOut& c, hipStream_t s);
void launch_carry(VSrcIQ8 src, const CarryOut& c, hipStream_t s);
void launch_carry(VSrc<float> src, const CarryOut& c, hipStream_t s);
void launch_f32_to_c32(const float* in, cf* out, long n, hipStream_t s);
void launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t s);   // device <-> device view of registered host memory
void launch_c32_re(const cf* in, float* out, long n, hipStream_t s);

// ---- head fix of the fused FirFilter -> FftFilter blocks (stream start only, a few hundred samples) ----------------
// z[m] = sum_k t1[k] V[voff + m + L1 - 1 - k], m < n: the front FirFilter's first outputs (fir.rs:166-177) from the virtual stream
void launch_head_z(VSrc<cf> V, long voff, const cf* t1, int L1, cf* z, long n, hipStream_t s);
// y[i] = sum_{j <= min(i, L2 - 1)} t2[j] z[i - j], i < n: FftFilter's zero-history head (fft_filter.rs:332-348)
void launch_head_y(const cf* z, const cf* t2, int L2, cf* y, long n, hipStream_t s);
// the same head resampled (r[u] = y[floor(u D / I)]) and demodulated: rewrites out[u - 1] for every u < r_hi whose pair
// touches y[n], n < L2 - 1, and *last_r when r[r_hi - 1] does (nz = valid entries of z)
void launch_head_demod(const cf* z, long nz, const cf* t2, int L2, long I, long D, float gain, int mode, long r_hi, float* out,
                       cf* last_r, hipStream_t s);

// ---- glue of the any-size transforms (kernels_misc.hip; see AnyFft in blocks.hpp) --------------------------------------
void launch_transpose_tw(const cf* in, cf* out, int rows, int cols, long nframes, const cf* tw, hipStream_t s);
void launch_chirp_pre(const cf* in, cf* a, long N, long M, long nframes, const cf* chirp, hipStream_t s);
void launch_mul_conj(cf* a, const cf* b, long M, long nframes, hipStream_t s);
void launch_chirp_post(const cf* y, cf* out, long N, long M, long nframes, const cf* chirp, hipStream_t s);
void launch_ols_gather(VSrc<cf> src, cf* frames, long S, long M, long f0, long nframes, hipStream_t s);
void launch_ols_scatter(const cf* frames, cf* out, long S, long M, long L, long f0, long nframes, long n_out, hipStream_t s);

int device_cu_count();

}  // namespace rr
