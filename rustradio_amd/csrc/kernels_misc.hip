// kernels_misc.hip — RationalResampler gather, QuadratureDemod, and the small
// carry-state copies.  All are pure HBM-streaming kernels.
#include "kernels.hpp"

namespace rr {

static inline unsigned grid_for(long n, int block, int per_thread = 1) {
    long g = (n + (long)block * per_thread - 1) / ((long)block * per_thread);
    const long cap = (long)device_cu_count() * 8;  // grid-stride above ~8 blocks/CU
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// ---- RationalResampler (src/rational_resampler.rs:183-198 in closed form) -----------
// The reference's counter loop emits input k while counter > 0; with counter c0 at the
// start of the window, output m comes from input floor((m*D - c0) / I)  (SURVEY F3/A.6).
template <class E>
__global__ __launch_bounds__(256) void k_resample(const E* __restrict__ in, E* __restrict__ out, long r,
                                                  const E* __restrict__ pending, long n_gather,
                                                  long I, long D, long c0) {
    const long total = r + n_gather;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        if (o < r) {
            out[o] = pending[0];
        } else {
            const long m = o - r;
            const long k = (I == 1) ? (m * D - c0) : (m * D - c0) / I;
            out[o] = in[k];
        }
    }
}

struct alignas(16) e16 { unsigned int a, b, c, d; };

void launch_resample(const void* in, void* out, size_t es, long r, const void* pending, long n_gather,
                     long I, long D, long c0, hipStream_t s) {
    const long total = r + n_gather;
    if (total <= 0) return;
    const unsigned g = grid_for(total, 256);
#define RR_RS(E)                                                                                          \
    hipLaunchKernelGGL(k_resample<E>, dim3(g), dim3(256), 0, s, (const E*)in, (E*)out, r, (const E*)pending, \
                       n_gather, I, D, c0)
    switch (es) {
    case 1: RR_RS(unsigned char); break;
    case 2: RR_RS(unsigned short); break;
    case 4: RR_RS(unsigned int); break;
    case 8: RR_RS(unsigned long long); break;
    case 16: RR_RS(e16); break;
    default: throw Error("resampler: unsupported element size");
    }
#undef RR_RS
    RR_HIP(hipGetLastError());
}

// ---- QuadratureDemod (src/quadrature_demod.rs:65-109) ---------------------------------
// fast-math 0.1.1 atan2 restated from its published algorithm (crate not vendored: parity-unpinned flavour).
__device__ __forceinline__ float flip_sign(float v, float s) {
    return __uint_as_float(__float_as_uint(v) ^ (__float_as_uint(s) & 0x80000000u));
}
__device__ __forceinline__ float fm_atan_raw(float x) {
    return mul_rn(sub_rn(add_rn(0.78539816339744830962f, 0.273f), mul_rn(0.273f, fabsf(x))), x);
}
__device__ __forceinline__ float fm_atan2(float y, float x) {
    if (fabsf(y) < fabsf(x)) {
        const float bias = x > 0.0f ? 0.0f : 3.14159265358979323846f;
        return add_rn(flip_sign(bias, y), fm_atan_raw(__fdiv_rn(y, x)));
    } else if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return flip_sign(1.57079632679489661923f, y);
    }
    return sub_rn(flip_sign(1.57079632679489661923f, y), fm_atan_raw(__fdiv_rn(x, y)));
}

template <int MODE>
__global__ __launch_bounds__(256) void k_quaddemod(const cf* __restrict__ in, float* __restrict__ out,
                                                   long n_out, float gain) {
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += (long)gridDim.x * blockDim.x) {
        const cf a = in[n], b = in[n + 1];
        // conj(a) * b in num-complex order, un-contracted so that signed zeros behave as on the CPU
        const float na = -a.y;
        const float re = sub_rn(mul_rn(a.x, b.x), mul_rn(na, b.y));
        const float im = add_rn(mul_rn(a.x, b.y), mul_rn(na, b.x));
        const float ang = MODE == 0 ? atan2f(im, re) : fm_atan2(im, re);
        out[n] = mul_rn(gain, ang);
    }
}

void launch_quaddemod(const cf* in, float* out, long n_out, float gain, int mode, hipStream_t s) {
    if (n_out <= 0) return;
    const unsigned g = grid_for(n_out, 256);
    if (mode == 0) hipLaunchKernelGGL(k_quaddemod<0>, dim3(g), dim3(256), 0, s, in, out, n_out, gain);
    else hipLaunchKernelGGL(k_quaddemod<1>, dim3(g), dim3(256), 0, s, in, out, n_out, gain);
    RR_HIP(hipGetLastError());
}

// ---- RtlSdrDecode (src/rtlsdr_decode.rs:35-42): u8 I/Q pairs -> Complex ---------------------
// (Float::from(b) - 127.0) * 0.008, two roundings at most (the subtraction is exact): bit-exact.
// 10 B of traffic per sample; a thread converts 2 samples (4-byte load, 16-byte store), both
// lane-consecutive.
__device__ __forceinline__ float rtl_cvt(unsigned b) { return mul_rn(sub_rn((float)b, 127.0f), 0.008f); }
__global__ __launch_bounds__(256) void k_rtlsdr_decode2(const unsigned* __restrict__ in, float4* __restrict__ out,
                                                        long npairs) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (long)gridDim.x * blockDim.x) {
        const unsigned w = in[i];
        out[i] = make_float4(rtl_cvt(w & 0xffu), rtl_cvt((w >> 8) & 0xffu), rtl_cvt((w >> 16) & 0xffu), rtl_cvt(w >> 24));
    }
}
__global__ __launch_bounds__(256) void k_rtlsdr_decode1(const unsigned char* __restrict__ in, cf* __restrict__ out,
                                                        long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = mkcf(rtl_cvt(in[2 * i]), rtl_cvt(in[2 * i + 1]));
}
void launch_rtlsdr_decode(const unsigned char* in, cf* out, long n_out, hipStream_t s) {
    if (n_out <= 0) return;
    const bool aligned = ((uintptr_t)in & 3) == 0 && ((uintptr_t)out & 15) == 0;
    const long npairs = aligned ? n_out / 2 : 0;
    if (npairs > 0) {
        hipLaunchKernelGGL(k_rtlsdr_decode2, dim3(grid_for(npairs, 256)), dim3(256), 0, s,
                           reinterpret_cast<const unsigned*>(in), reinterpret_cast<float4*>(out), npairs);
        RR_HIP(hipGetLastError());
    }
    const long done = 2 * npairs;                        // unaligned windows and the odd last sample
    if (done < n_out) {
        hipLaunchKernelGGL(k_rtlsdr_decode1, dim3(grid_for(n_out - done, 256)), dim3(256), 0, s, in + 2 * done,
                           out + done, n_out - done);
        RR_HIP(hipGetLastError());
    }
}

// ---- MultiplyConst (src/multiply_const.rs:20-22) and FastFM (src/quadrature_demod.rs:158-164) -----------
// bit-exact: one rounding per multiply / subtract, no contraction (num-complex Mul order for Complex)
__global__ __launch_bounds__(256) void k_mulconst_f32(const float* __restrict__ in, float* __restrict__ out, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = mul_rn(in[i], v);
}
__global__ __launch_bounds__(256) void k_mulconst_c32(const cf* __restrict__ in, cf* __restrict__ out, long n, float vr, float vi) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const cf a = in[i];
        out[i] = mkcf(sub_rn(mul_rn(a.x, vr), mul_rn(a.y, vi)), add_rn(mul_rn(a.x, vi), mul_rn(a.y, vr)));
    }
}
void launch_mulconst_f32(const float* in, float* out, long n, float v, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_mulconst_f32, dim3(grid_for(n, 256)), dim3(256), 0, s, in, out, n, v);
    RR_HIP(hipGetLastError());
}
void launch_mulconst_c32(const cf* in, cf* out, long n, float vr, float vi, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_mulconst_c32, dim3(grid_for(n, 256)), dim3(256), 0, s, in, out, n, vr, vi);
    RR_HIP(hipGetLastError());
}
// out[n] = (s[n].im - s[n-2].im) * s[n-1].re - (s[n].re - s[n-2].re) * s[n-1].im over the virtual stream
// src = (q2, q1) || window: the sequential q1/q2 update of the reference is just a 2-sample history.
__global__ __launch_bounds__(256) void k_fastfm(VSrc<cf> src, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const cf q2 = src.load(i), q1 = src.load(i + 1), s = src.load(i + 2);
        const float top = mul_rn(sub_rn(s.y, q2.y), q1.x);
        const float bottom = mul_rn(sub_rn(s.x, q2.x), q1.y);
        out[i] = sub_rn(top, bottom);
    }
}
void launch_fastfm(VSrc<cf> src, float* out, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fastfm, dim3(grid_for(n, 256)), dim3(256), 0, s, src, out, n);
    RR_HIP(hipGetLastError());
}

// ---- carry-state copies -----------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_vcopy(VSrc<T> src, long v0, T* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src.load(v0 + i);
}
void launch_vcopy_c32(VSrc<cf> src, long v0, cf* dst, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_vcopy<cf>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, v0, dst, n);
    RR_HIP(hipGetLastError());
}
__global__ __launch_bounds__(256) void k_vcopy_iq8(VSrcIQ8 src, long v0, cf* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = src.load(v0 + i);
}
void launch_vcopy_iq8(VSrcIQ8 src, long v0, cf* dst, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_vcopy_iq8, dim3(grid_for(n, 256)), dim3(256), 0, s, src, v0, dst, n);
    RR_HIP(hipGetLastError());
}
void launch_vcopy_f32(VSrc<float> src, long v0, float* dst, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_vcopy<float>, dim3(grid_for(n, 256)), dim3(256), 0, s, src, v0, dst, n);
    RR_HIP(hipGetLastError());
}
void launch_carry(VSrc<cf> src, const CarryOut& c, hipStream_t s) {
    if (c.n > 0) launch_vcopy_c32(src, c.v0, static_cast<cf*>(c.dst), c.n, s);
}
void launch_carry(VSrcIQ8 src, const CarryOut& c, hipStream_t s) {
    if (c.n > 0) launch_vcopy_iq8(src, c.v0, static_cast<cf*>(c.dst), c.n, s);
}
void launch_carry(VSrc<float> src, const CarryOut& c, hipStream_t s) {
    if (c.n > 0) launch_vcopy_f32(src, c.v0, static_cast<float*>(c.dst), c.n, s);
}

// ---- FftFilter / FftFilterFloat: non-finite samples on the REFERENCE's blocks (round 5) ------------------
// The reference runs one fft_size-point transform per `nsamples` input samples and adds the last `ntaps` points of it to
// the next block (fft_filter.rs:326-347): one NaN / Inf input sample of block b makes ALL of the block's transform
// non-finite, i.e. the outputs [b S, b S + S + L) (S = nsamples, L = ntaps; S >= L by fft_filter.rs:36-42) — and nothing
// else.  The GPU tiles are of another size and on another grid, so the same sample poisons the outputs of ITS tile.  This
// pass runs behind the tile kernels of such a block and puts the reference's set in place:
//   * probe: a tile that read a non-finite sample has NO finite output, so one output every P (P <= the smallest tile
//     advance of the call) and the ends of a range tell whether any tile over the range did — n_out / P loads in the steady
//     state, then every workgroup returns;
//   * a workgroup whose range (its blocks and the one before) holds such a tile scans the INPUT of those blocks, and per
//     block: bad[b] or (bad[b - 1] and m - b S < L) -> out[m] = NaN; otherwise an output the tile left non-finite is the
//     reference's own left fold (nan_fix.hpp nf_direct), whose window lies in clean blocks.
// The block before the call's first is gone with its input: its verdict is carried in tail[(seq - 1) & 1] == seq - 1
// (written by the call that saw it; a call whose probes were clean writes nothing and leaves a stale sequence number).
__host__ __device__ inline long ref_blocks_sub(long S) { const long q = S / 256; return q < 1 ? 1 : q > 64 ? 64 : q; }
struct RefBlocksCtx {
    const void* prefix; long plen; const void* in; long in_len;     // the call's virtual stream (VSrc)
    void* out; long n_out;                                         // k * S outputs
    long S, P, hist;                                               // block b reads the virtual samples [hist + b S, hist + (b + 1) S + front)
    int L, front;                                                  // front: a FirFilter of front + 1 taps fused in front (rr_fir_fftfilter_create) —
                                                                   // L = the composite's taps, the FftFilter stage's are L - front, and the sample z1[m]
                                                                   // the reference's FftFilter reads is not finite iff one of x[m .. m + front] is not
    const void* rev;                                               // the taps reversed (cf / float)
    int* tail; int seq;
    long nb, bpw, sub, qpp;                                        // blocks of the call, blocks per workgroup, sub-ranges per block and per row
    int force0;                                                    // look at block 0 whatever the probes say (the fused block's head fix has
                                                                   // overwritten the first outputs of a possibly poisoned tile with its own values)
};
template <class T>
__global__ __launch_bounds__(256) void k_ref_blocks_nonfinite(RefBlocksCtx c) {
    const VSrc<T> src{static_cast<const T*>(c.prefix), c.plen, static_cast<const T*>(c.in), c.in_len};
    T* out = static_cast<T*>(c.out);
    // (no division on the way to the probes: the launcher passes the block count and the blocks per workgroup — five 64-bit
    //  divisions were 2 us of this kernel's 5 in the steady state, where it runs behind every call)
    const long nb = c.nb;
    const long b0 = (long)blockIdx.x * c.bpw, b1 = b0 + c.bpw < nb ? b0 + c.bpw : nb;
    if (b0 >= b1) return;
    // work items = (block, sub-range of its outputs): long blocks are cut so that the fold over what a tile smeared (up to S
    // outputs x L taps for ONE bad sample) is shared by several waves — sub-ranges of >= 256 outputs, at most 64 per block,
    // qpp of them per workgroup row (blockIdx.y)
    const long sub = c.sub;
    const long q0 = (long)blockIdx.y * c.qpp, q1 = q0 + c.qpp < sub ? q0 + c.qpp : sub, nq = q1 - q0;
    if (nq <= 0) return;
    const long w0 = 0, w1 = (b1 - b0) * nq;                        // this workgroup's items
    const bool tail_bad = c.tail[(c.seq - 1) & 1] == c.seq - 1;    // (uniform)
    const int t = (int)threadIdx.x;
    // any non-finite output at lo, lo + P, ..., or at hi - 1?  A tile is >= P outputs long, so one that overlaps [lo, hi)
    // holds one of these points.  (Ranges are probed block-aligned and the caller's own blocks apart from the block before
    // them: that one is being repaired by another wave meanwhile — a bad block only ever turns all-NaN, but a merely
    // smeared stretch of it turns finite, and a lattice that started there could step over what the same tile left HERE.)
    auto probe = [&](long lo, long hi, int lane, int lanes) {
        bool bad = false;
        for (long m = lo + (long)lane * c.P; m < hi; m += (long)lanes * c.P) bad |= nf_bad(out[m]);
        if (lane == 0) bad |= nf_bad(out[hi - 1]);
        return bad;
    };
    {
        bool bad = probe(b0 * c.S, b1 * c.S, t, (int)blockDim.x);
        if (b0 > 0) bad |= probe((b0 - 1) * c.S, b0 * c.S, t, (int)blockDim.x);
        // (force0: the head fix cut the poisoned tile's run of non-finite outputs short of P, so neither block 0's lattice nor —
        //  for the tail the block before it leaves — block 1's may find it: both look at their input.  Found by the soak, seed 10074.)
        if (!__syncthreads_or((int)bad | (int)(b0 == 0 && tail_bad) | (int)(b0 <= 1 && c.force0))) return;
    }
    const int lane = t & 63, wave = t >> 6, nw = (int)(blockDim.x >> 6);
    auto any64 = [](bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; };
    auto scan = [&](long b) {                                       // a non-finite INPUT sample in block b?
        bool bad = false;
        const long v0 = c.hist + b * c.S;
        const long len = c.S + c.front;
        for (long i0 = 0; i0 < len; i0 += 512) {                    // (uniform trip count; eight loads in flight per lane)
            T x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = src.load(v0 + i0 + 64 * k + lane);   // (past the block's end: still the stream, or its zero padding — masked below)
#pragma unroll
            for (int k = 0; k < 8; k++) bad |= i0 + 64 * k + lane < len && nf_bad(x[k]);
            if (any64(bad)) break;
        }
        return any64(bad);
    };
    T nanv;
    if constexpr (std::is_same<T, float>::value) nanv = __builtin_nanf(""); else nanv = mkcf(__builtin_nanf(""), __builtin_nanf(""));
    for (long w = w0 + wave; w < w1; w += nw) {                     // one wave per item
        const long bq = w / nq, q = q0 + (w - bq * nq), b = b0 + bq;
        const long i0 = c.S * q / sub, i1 = c.S * (q + 1) / sub;    // its outputs of block b
        const bool first = b == 0;
        // the block's own lattice (a bad sample anywhere in it poisons ALL of it: the tile that read the sample may end before
        // this sub-range), the previous block's (its tail), and this sub-range's (what a tile smeared here; the other
        // sub-ranges of the block are being repaired by other waves)
        bool hit = probe(b * c.S, (b + 1) * c.S, lane, 64);
        if (sub > 1) hit |= probe(b * c.S + i0, b * c.S + i1, lane, 64);
        if (!first) hit |= probe((b - 1) * c.S, b * c.S, lane, 64);
        if (!any64(hit) && !(first && tail_bad) && !(b <= 1 && c.force0)) continue;
        const bool bad_prev = first ? tail_bad : scan(b - 1);
        const bool bad_cur = scan(b);
        for (long i = i0 + lane; i < i1; i += 64) {
            const long m = b * c.S + i;
            if (bad_cur || (bad_prev && i < (long)(c.L - c.front))) out[m] = nanv;
            else if (nf_bad(out[m])) {
                if constexpr (std::is_same<T, float>::value) out[m] = nf_fold_ff(src, static_cast<const float*>(c.rev), c.L, m);
                else out[m] = nf_fold_cc(src, static_cast<const cf*>(c.rev), c.L, m);   // (y[m] reads the virtual samples [m, m + L): hist + front = L - 1)
            }
        }
        if (b == nb - 1 && q == sub - 1 && lane == 0) c.tail[c.seq & 1] = bad_cur ? c.seq : -1;
    }
}
template <class T>
static void launch_ref_blocks(VSrc<T> src, T* out, long n_out, long S, long P, long hist, int L, int front, const void* rev, int* tail, int seq, bool force0, hipStream_t s) {
    if (n_out <= 0) return;
    RefBlocksCtx c{src.prefix, src.plen, src.in, src.in_len, out, n_out, S, P < 1 ? 1 : P, hist, L, front, rev, tail, seq, 0, 0, 0, 0, force0 ? 1 : 0};
    c.nb = n_out / S; c.sub = ref_blocks_sub(S);
    const long rows = std::min<long>(16, (c.sub + 3) / 4);          // (a workgroup has four waves)
    c.qpp = (c.sub + rows - 1) / rows;
    c.bpw = std::max<long>(1, (c.nb * rows + 255) / 256);
    const long gx = (c.nb + c.bpw - 1) / c.bpw;
    hipLaunchKernelGGL(k_ref_blocks_nonfinite<T>, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, s, c);
    RR_HIP(hipGetLastError());
}
void launch_ref_blocks_nonfinite(VSrc<cf> src, cf* out, long n_out, long S, long P, long hist, int L, int front, const cf* rev, int* tail, int seq, bool force0, hipStream_t s) {
    launch_ref_blocks<cf>(src, out, n_out, S, P, hist, L, front, rev, tail, seq, force0, s);
}
void launch_ref_blocks_nonfinite(VSrc<float> src, float* out, long n_out, long S, long P, long hist, int L, int front, const float* rev, int* tail, int seq, bool force0, hipStream_t s) {
    launch_ref_blocks<float>(src, out, n_out, S, P, hist, L, front, rev, tail, seq, force0, s);
}

// ---- fused FM / audio chains: non-finite samples on the REFERENCE's blocks (round 6) ------------------------------------
// FftFilter -> RationalResampler -> QuadratureDemod (and FftFilterFloat -> RationalResampler -> MultiplyConst) in the
// reference: the FftFilter stage poisons the filtered samples y[b S, (b + 1) S + Lf) of a block that holds a non-finite input
// (fft_filter.rs:326-347), the resampler picks r[u] = y[floor(u D / I)] (rational_resampler.rs:183-198), the demodulator
// pairs them, out[u - 1] = gain atan2(conj(r[u - 1]) r[u]) (quadrature_demod.rs:65-109): an output is NaN iff one of the (two)
// filtered samples it reads lies in a poisoned stretch — and nothing else is.  The fused kernels work on tiles of another size
// on another grid: a tile that read a NaN has no finite output.  This pass, behind the chain kernel of a call, puts the
// reference's set in place exactly like k_ref_blocks_nonfinite does for the FftFilter block:
//   * probe: one output every P (<= the outputs of the smallest tile of the call) over a workgroup's blocks and the two before
//     them — clean: every workgroup returns (the steady state);
//   * otherwise, per block: scan the INPUT of the block and the two before it, write NaN where the reference does, and
//     recompute what only the tile had smeared: y at the one or two positions an output reads by the reference-order fold
//     (nan_fix.hpp nf_fold_cc / nf_fold_ff), then the reference's own conj-multiply and atan2 (k_quaddemod's code).
// The verdicts a later call needs (was the last / second-to-last block poisoned, was the carried r) sit in three sequence-
// numbered slots like k_ref_blocks_nonfinite's tail.  NaN only: what an Inf turns into inside rustfft (Inf or NaN, bin by
// bin) is not defined by anything this repository holds, and atan2 of two infinities is finite — stated in DESIGN.md.
// RTL-SDR byte sources cannot carry a non-finite sample ((b - 127) * 0.008): those chains never launch this.
struct ChainBlocksCtx {
    const void* prefix; long plen; const void* in; long in_len;     // the call's virtual stream: y[m] reads [m, m + L) of it
    float* out; long out_stride; int nchan;                          // channel c writes out + c * out_stride
    long A, n_y, r_lo, r_hi, o_base, I, D;                           // FmChainArgs / AudioChainArgs (o_base = r_lo for the audio chain)
    long S, hist, P;                                                 // block b reads [hist + b S, hist + (b + 1) S + front); probe stride
    int L, front;
    const void* rev; long rev_stride;                                // reversed taps per channel (cf / float)
    float gain; int mode;                                            // demodulator: RR_ATAN2_*; audio: gain = scale
    const cf* last_in;                                               // [nchan] r[r_lo - 1] (demodulator only)
    cf* last_out;                                                    // [nchan] r[r_hi - 1] as the chain kernel left it for the next call
    int* slots; int seq;                                             // slots[2 k + (q & 1)] == q: k = 0 last block, 1 second-to-last, 2 carried r — of call q
    long nb, bpw;                                                    // blocks of the call, blocks per workgroup
    int force;                                                       // 1: block 0 regardless (head fix), 2: every block (a call without outputs)
    int sparse;                                                      // ceil(D / I) > S: at most one output per reference block
};
__device__ __forceinline__ long chain_n2(long y, long I, long D) { return (long)(((__int128)y * I + D - 1) / D); }   // first r index whose source is >= y
template <class T, bool DEMOD, bool WIDE>
__global__ __launch_bounds__(256) void k_chain_blocks_nonfinite(ChainBlocksCtx c) {
    const VSrc<T> src{static_cast<const T*>(c.prefix), c.plen, static_cast<const T*>(c.in), c.in_len};
    const int ch = (int)blockIdx.y;
    float* out = c.out + (long)ch * c.out_stride;
    const long b0 = (long)blockIdx.x * c.bpw, b1 = b0 + c.bpw < c.nb ? b0 + c.bpw : c.nb;
    if (b0 >= b1) return;
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6, nw = (int)(blockDim.x >> 6);
    auto slot = [&](int k) { return c.slots[2 * k + ((c.seq - 1) & 1)] == c.seq - 1; };
    const bool tail1 = slot(0), tail2 = slot(1), lastr = slot(2);
    // outputs: index o <-> u = o + o_base + (DEMOD ? 1 : 0), the newest filtered sample it reads is y[floor(u D / I)]
    // (WIDE: stream positions times the ratio beyond 2^62 — 128-bit products; otherwise plain 64-bit divisions: the 128-bit
    //  ones are a few hundred instructions each, and six of them per thread were 11 of this pass's 14 us behind a 24e6-sample call)
    const long u_min = DEMOD ? (c.r_lo > 1 ? c.r_lo : 1) : c.r_lo;
    // (a software 64-bit division is ~150 instructions and four of them stood in front of every workgroup's probes; below 2^52
    //  the quotient comes out of one f64 division and a fix-up of at most one either way)
    auto qdiv = [](long x, long d) {
        if (x < ((long)1 << 52)) {
            long q = (long)((double)x / (double)d);
            long r = x - q * d;
            if (r < 0) { q--; r += d; }
            if (r >= d) q++;
            return q;
        }
        return x / d;
    };
    auto n2 = [&](long y) { return WIDE ? chain_n2(y, c.I, c.D) : qdiv(y * c.I + c.D - 1, c.D); };   // first r index whose source is >= y
    auto src_of = [&](long u) { return WIDE ? (long)(((__int128)u * c.D) / c.I) : qdiv(u * c.D, c.I); };
    auto u_of_block = [&](long b) { long u = n2(c.A + b * c.S); return u < u_min ? u_min : u; };   // first u of block b (b may be nb)
    auto o_of = [&](long u) { return u - c.o_base - (DEMOD ? 1 : 0); };
    auto probe = [&](long ulo, long uhi, int ln, int lanes) {
        bool bad = false;
        if (uhi <= ulo) return false;
        for (long u = ulo + (long)ln * c.P; u < uhi; u += (long)lanes * c.P) bad |= nf_bad(out[o_of(u)]);
        if (ln == 0) bad |= nf_bad(out[o_of(uhi - 1)]);
        return bad;
    };
    // Where the evidence of a poisoned block b lies: the tile that read the sample has no finite output, and the outputs nearest
    // to the sample belong to block b, to b - 1 (the one just before it) or — when no output of block b reads a filtered sample
    // at or behind it — to block b + 1 (found by the soak: a 17-tap filter under 1:17, the sample three from its block's end).
    // Nearest to a sample at the very END of a call there may be no output at all yet: the call's last block always looks at
    // its input (one wave, S samples).
    auto any64 = [](bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; };
    auto scan = [&](long b) {                                       // a non-finite INPUT sample in block b?  (wave-cooperative)
        if (b == -1) return tail1;
        if (b < -1) return tail2;
        bool bad = false;
        const long v0 = c.hist + b * c.S, len = c.S + c.front;
        for (long i0 = 0; i0 < len; i0 += 1024) {                   // (sixteen loads in flight: one round trip for a block of up to 1024 samples)
            T x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = src.load(v0 + i0 + 64 * k + lane);
#pragma unroll
            for (int k = 0; k < 16; k++) bad |= i0 + 64 * k + lane < len && nf_bad(x[k]);
            if (any64(bad)) break;
        }
        return any64(bad);
    };
    // A bad sample within ceil(D / I) of the call's END has no output at or behind it yet, and the tile that read it may hold none
    // in front of it either: no evidence anywhere.  So the wave that owns the call's last block always looks at those last few
    // input samples itself — ONE load per lane, issued here and looked at after the workgroup's probes (a scan of the whole block
    // in front of them was 5 of this pass's 11 us in the steady state; behind them, still 3).
    const bool owns_last = b1 == c.nb && (int)((c.nb - 1 - b0) % nw) == wave;
    T tail_x[4];
    long tail_n = 0;
    if (owns_last) {
        const long G = (c.D + c.I - 1) / c.I;
        tail_n = (G < c.S ? G : c.S) + c.front;                      // samples [hist + n_y - min(G, S), hist + n_y + front)
        if (tail_n > 256) tail_n = -1;                               // (a ratio beyond 256: the whole block, below)
        const long v0 = c.hist + c.n_y + c.front - tail_n;
#pragma unroll
        for (int k = 0; k < 4; k++) tail_x[k] = tail_n > 0 && 64 * k + lane < tail_n ? src.load(v0 + 64 * k + lane) : T{};
    }
    bool wg_hit;
    {   // (own blocks, the two before them and the one after as separate lattices: the others' are being rewritten by another
        //  workgroup meanwhile, and a lattice that started there could step over what the same tile left here)
        const long bp = b0 >= 2 ? b0 - 2 : 0, bn = b1 + 1 < c.nb ? b1 + 1 : c.nb;
        const long u0 = u_of_block(b0), u1 = u_of_block(b1);
        const bool hit = c.force == 2 || probe(u0, u1, t, (int)blockDim.x) ||
                         (b0 > 0 && probe(u_of_block(bp), u0, t, (int)blockDim.x)) || (b1 < c.nb && probe(u1, u_of_block(bn), t, (int)blockDim.x));
        wg_hit = __syncthreads_or((int)hit | (int)(b0 <= 1 && (tail1 || tail2 || lastr)) | (int)(b0 <= 1 && c.force)) != 0;
        if (!wg_hit && b1 != c.nb) return;
    }
    const long Lf = (long)(c.L - c.front);
    const float nanv = __builtin_nanf("");
    // Nothing on any lattice of this workgroup: only the call's last block goes on, and only if its input holds a bad sample.
    // (No loop over the other blocks then: the compiler hoists the index arithmetic of the loop body above an early `continue`,
    //  and 67 iterations of it per wave were 12 us behind a 32-channel call.)
    bool last_bad = false;
    if (owns_last) {
        if (tail_n < 0) last_bad = scan(c.nb - 1);
        else {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 4; k++) bad |= nf_bad(tail_x[k]);
            last_bad = any64(bad);
        }
    }
    if (!wg_hit && !last_bad) return;
    for (long b = wg_hit ? b0 + wave : c.nb - 1; b < b1; b += nw) {   // one wave per block
        const bool last = b == c.nb - 1;
        const long ulo = u_of_block(b), uhi = u_of_block(b + 1);
        bool hit = c.force == 2 || probe(ulo, uhi, lane, 64);
        if (b >= 1) hit |= probe(u_of_block(b - 1), ulo, lane, 64);
        if (b >= 2) hit |= probe(u_of_block(b - 2), u_of_block(b - 1), lane, 64);
        if (b + 1 < c.nb) hit |= probe(uhi, u_of_block(b + 2 < c.nb ? b + 2 : c.nb), lane, 64);
        if (!any64(hit) && !(b <= 1 && (tail1 || tail2 || lastr)) && !(b <= 1 && c.force) && !last) continue;
        const bool bad0 = scan(b), bad1 = scan(b - 1), bad2 = scan(b - 2);
        auto poisoned = [&](long yl) {                               // filtered sample A + yl, in block b or b - 1 (or carried: yl < 0)
            if (yl < 0) return lastr;
            const long bb = yl / c.S, off = yl - bb * c.S;
            const bool cur = bb == b ? bad0 : bad1, prv = bb == b ? bad1 : bad2;
            return cur || (prv && off < Lf);
        };
        // (more than one reference block per output — ceil(D / I) > S, tiny filters under huge decimations: the lower sample of a
        //  pair may lie any number of blocks back.  Then a block has at most one output: verdicts by uniform scans, lane 0 writes.)
        const bool sparse = c.sparse != 0;
        auto poisoned_any = [&](long yl) {                           // wave-uniform yl
            if (yl < 0) return lastr;
            const long bb = yl / c.S, off = yl - bb * c.S;
            return scan(bb) || (off < Lf && scan(bb - 1));
        };
        for (long u = ulo; sparse && u < uhi; u++) {
            const long yb = src_of(u) - c.A, o = o_of(u);
            if constexpr (DEMOD) {
                const long ya = src_of(u - 1) - c.A;
                const bool p = poisoned_any(ya) || poisoned_any(yb);
                if (lane != 0) continue;
                if (p) { out[o] = nanv; continue; }
                if (!nf_bad(out[o])) continue;
                const cf* rev = static_cast<const cf*>(c.rev) + (long)ch * c.rev_stride;
                const cf a = ya < 0 ? c.last_in[ch] : nf_fold_cc(src, rev, c.L, ya);
                const cf bq = nf_fold_cc(src, rev, c.L, yb);
                const float na = -a.y;
                const float re = sub_rn(mul_rn(a.x, bq.x), mul_rn(na, bq.y));
                const float im = add_rn(mul_rn(a.x, bq.y), mul_rn(na, bq.x));
                out[o] = mul_rn(c.gain, c.mode == 0 ? atan2f(im, re) : fm_atan2(im, re));
            } else {
                const bool p = poisoned_any(yb);
                if (lane != 0) continue;
                if (p) { out[o] = nanv; continue; }
                if (!nf_bad(out[o])) continue;
                out[o] = mul_rn(nf_fold_ff(src, static_cast<const float*>(c.rev) + (long)ch * c.rev_stride, c.L, yb), c.gain);
            }
        }
        for (long u = ulo + lane; !sparse && u < uhi; u += 64) {
            const long yb = src_of(u) - c.A;
            const long o = o_of(u);
            if constexpr (DEMOD) {
                const long ya = src_of(u - 1) - c.A;
                if (poisoned(ya) || poisoned(yb)) { out[o] = nanv; continue; }
                if (!nf_bad(out[o])) continue;
                const cf* rev = static_cast<const cf*>(c.rev) + (long)ch * c.rev_stride;
                const cf a = ya < 0 ? c.last_in[ch] : nf_fold_cc(src, rev, c.L, ya);
                const cf bq = nf_fold_cc(src, rev, c.L, yb);
                const float na = -a.y;                               // conj(a) * b, num-complex order, un-contracted (k_quaddemod)
                const float re = sub_rn(mul_rn(a.x, bq.x), mul_rn(na, bq.y));
                const float im = add_rn(mul_rn(a.x, bq.y), mul_rn(na, bq.x));
                out[o] = mul_rn(c.gain, c.mode == 0 ? atan2f(im, re) : fm_atan2(im, re));
            } else {
                if (poisoned(yb)) { out[o] = nanv; continue; }
                if (!nf_bad(out[o])) continue;
                const float* rev = static_cast<const float*>(c.rev) + (long)ch * c.rev_stride;
                out[o] = mul_rn(nf_fold_ff(src, rev, c.L, yb), c.gain);      // MultiplyConst: sample * val (multiply_const.rs)
            }
        }
        bool carried_bad = false;                                    // (sparse: a wave-uniform verdict on r[r_hi - 1], all lanes take part)
        if (sparse && last && c.r_hi > c.r_lo) carried_bad = poisoned_any(src_of(c.r_hi - 1) - c.A);
        if constexpr (DEMOD) {
            // the sample the chain kernel carried to the next call comes out of the same tile: poisoned -> NaN, smeared -> refolded
            if (last && lane == 0 && c.r_hi > c.r_lo) {
                const long yl = src_of(c.r_hi - 1) - c.A;
                cf* lo = c.last_out + ch;
                if (sparse ? carried_bad : poisoned(yl)) *lo = mkcf(nanv, nanv);
                else if (nf_bad(*lo)) *lo = nf_fold_cc(src, static_cast<const cf*>(c.rev) + (long)ch * c.rev_stride, c.L, yl);
            }
        }
        if (last && ch == 0 && lane == 0) {                          // what the next call needs (every channel would write the same)
            c.slots[0 + (c.seq & 1)] = bad0 ? c.seq : -1;
            c.slots[2 + (c.seq & 1)] = bad1 ? c.seq : -1;
            long ylast = -1;                                         // the filtered sample the carried r[r_hi - 1] is
            if (c.r_hi > c.r_lo) ylast = src_of(c.r_hi - 1) - c.A;
            c.slots[4 + (c.seq & 1)] = (c.r_hi > c.r_lo ? (sparse ? carried_bad : poisoned(ylast)) : lastr) ? c.seq : -1;
        }
    }
}
template <class T, bool DEMOD>
static void launch_chain_blocks(ChainBlocksCtx c, hipStream_t s) {
    if (c.n_y <= 0 || c.nchan <= 0) return;
    c.nb = c.n_y / c.S;
    if (c.P < 1) c.P = 1;
    if (c.r_hi <= (DEMOD ? (c.r_lo > 1 ? c.r_lo : 1) : c.r_lo)) c.force = 2;      // no output to probe: the verdicts come from the input
    c.sparse = (c.D + c.I - 1) / c.I > c.S ? 1 : 0;
    if (c.sparse) c.force = 2;          // (a tile's NaN run may be shorter than a block's span in outputs: every block looks at its input)
    // ~1 workgroup per CU over all channel rows (a workgroup has four waves: at least four blocks each; 128 / 256 / 512 workgroups
    // measure 5.96 / 6.2 / 6.43 us in the steady state, an EMPTY kernel of this shape 4.8)
    const long per_row = std::max<long>(1, 256 / c.nchan);
    c.bpw = std::max<long>(4, (c.nb + per_row - 1) / per_row);
    const long gx = (c.nb + c.bpw - 1) / c.bpw;
    const __int128 lim = (__int128)1 << 62;
    const bool wide = (__int128)(c.A + c.n_y + 1) * c.I >= lim || (__int128)(c.r_hi + 1) * c.D >= lim;
    if (wide) hipLaunchKernelGGL((k_chain_blocks_nonfinite<T, DEMOD, true>), dim3((unsigned)gx, (unsigned)c.nchan), dim3(256), 0, s, c);
    else hipLaunchKernelGGL((k_chain_blocks_nonfinite<T, DEMOD, false>), dim3((unsigned)gx, (unsigned)c.nchan), dim3(256), 0, s, c);
    RR_HIP(hipGetLastError());
}
void launch_chain_blocks_nonfinite(VSrc<cf> src, float* out, long out_stride, int nchan, const FmChainArgs& a, long S, long hist, long P,
                                   int L, int front, const cf* rev, long rev_stride, const cf* last_in, cf* last_out, int* slots, int seq, int force,
                                   hipStream_t s) {
    ChainBlocksCtx c{src.prefix, src.plen, src.in, src.in_len, out, out_stride, nchan, a.A, a.n_y, a.r_lo, a.r_hi, a.o_base, a.I, a.D,
                     S, hist, P, L, front, rev, rev_stride, a.gain, a.mode, last_in, last_out, slots, seq, 0, 0, force, 0};
    launch_chain_blocks<cf, true>(c, s);
}
void launch_chain_blocks_nonfinite(VSrc<float> src, float* out, const AudioChainArgs& a, long S, long hist, long P, int L,
                                   const float* rev, int* slots, int seq, hipStream_t s) {
    ChainBlocksCtx c{src.prefix, src.plen, src.in, src.in_len, out, 0, 1, a.A, a.n_y, a.r_lo, a.r_hi, a.r_lo, a.I, a.D,
                     S, hist, P, L, 0, rev, 0, a.scale, 0, nullptr, nullptr, slots, seq, 0, 0, 0, 0};
    launch_chain_blocks<float, false>(c, s);
}

// ---- Hilbert on transform tiles: the reference's locality for non-finite samples (round 5) -------------------
// k_fftfilt_real<.., HILB> runs at its register limit and carries no nan_fix.hpp hooks (a transformer of 200 ... 3584 taps
// on large windows): one non-finite input sample makes the imaginary part of its tile's 2 S outputs non-finite, where the
// reference's per-output fold (hilbert.rs:113-116) reaches the ntaps outputs whose window holds it.  Same scheme as above:
// probe one output every P <= S; a workgroup whose range holds a poisoned tile replaces every non-finite output of the range
// by the reference's own fold (nf_direct: non-finite exactly where the reference's is — its window holds the sample, zero
// taps included; a clean tile's windows are clean, so nothing is missed).
struct RefoldCtx {
    const void* prefix; long plen; const void* in; long in_len;
    void* out; long n_out, P, span;                                // span: outputs per workgroup
    int L; const void* rev;
};
__global__ __launch_bounds__(256) void k_hilbert_refold_nonfinite(RefoldCtx c) {
    const VSrc<float> src{static_cast<const float*>(c.prefix), c.plen, static_cast<const float*>(c.in), c.in_len};
    cf* out = static_cast<cf*>(c.out);
    const long lo = (long)blockIdx.x * c.span, hi = lo + c.span < c.n_out ? lo + c.span : c.n_out;
    if (lo >= hi) return;
    const int t = (int)threadIdx.x;
    bool bad = t == 0 && nf_bad(out[hi - 1]);
    for (long m = lo + (long)t * c.P; m < hi; m += (long)blockDim.x * c.P) bad |= nf_bad(out[m]);
    if (!__syncthreads_or((int)bad)) return;
    for (long m = lo + t; m < hi; m += blockDim.x)
        if (nf_bad(nf_peek(out + m))) out[m] = mkcf(src.load(m + c.L / 2), nf_fold_ff(src, static_cast<const float*>(c.rev), c.L, m));   // hilbert.rs:113-116
}
void launch_hilbert_refold_nonfinite(VSrc<float> src, cf* out, long n_out, long P, int L, const float* rev, hipStream_t s) {
    if (n_out <= 0) return;
    RefoldCtx c{src.prefix, src.plen, src.in, src.in_len, out, n_out, P < 1 ? 1 : P, 0, L, rev};
    c.span = std::max<long>(4 * c.P, (n_out + 255) / 256);
    const long grid = (n_out + c.span - 1) / c.span;
    hipLaunchKernelGGL(k_hilbert_refold_nonfinite, dim3((unsigned)grid), dim3(256), 0, s, c);
    RR_HIP(hipGetLastError());
}

// Plain byte copy between a device range and a page-locked host range seen through its device address (rr_dstream_copy_in /
// _copy_out on windows of a ring registered with rr_host_register): a KERNEL moves a reference-sized window over PCIe at
// 55 GB/s either way, hipMemcpyAsync from / to the same registered range at 16 / 50 (tools/micro/pcie_inplace.hip).  8 bytes
// per lane where the alignment allows (16-byte reads of host memory measured slower: 41 GB/s), any alignment and length.
template <class V>
__global__ __launch_bounds__(256) void k_copy_v(const V* __restrict__ src, V* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_bytes(const void* src, void* dst, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)bytes;
    if ((a & 7) == 0) {
        const long n = (long)(bytes / 8);
        hipLaunchKernelGGL(k_copy_v<unsigned long long>, dim3(grid_for(n, 256)), dim3(256), 0, s, static_cast<const unsigned long long*>(src),
                           static_cast<unsigned long long*>(dst), n);
    } else if ((a & 3) == 0) {
        const long n = (long)(bytes / 4);
        hipLaunchKernelGGL(k_copy_v<unsigned>, dim3(grid_for(n, 256)), dim3(256), 0, s, static_cast<const unsigned*>(src), static_cast<unsigned*>(dst), n);
    } else {
        hipLaunchKernelGGL(k_copy_v<unsigned char>, dim3(grid_for((long)bytes, 256)), dim3(256), 0, s, static_cast<const unsigned char*>(src),
                           static_cast<unsigned char*>(dst), (long)bytes);
    }
    RR_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_f32_to_c32(const float* __restrict__ in, cf* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = mkcf(in[i], 0.0f);
}
__global__ __launch_bounds__(256) void k_c32_re(const cf* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = in[i].x;
}
void launch_f32_to_c32(const float* in, cf* out, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_f32_to_c32, dim3(grid_for(n, 256)), dim3(256), 0, s, in, out, n);
    RR_HIP(hipGetLastError());
}
void launch_c32_re(const cf* in, float* out, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_c32_re, dim3(grid_for(n, 256)), dim3(256), 0, s, in, out, n);
    RR_HIP(hipGetLastError());
}

// ---- head fix of the fused FirFilter -> FftFilter blocks ---------------------------------------------------------
// Runs once per stream on a few hundred samples: plain one-thread-per-output sums.
__global__ __launch_bounds__(256) void k_head_z(VSrc<cf> V, long voff, const cf* __restrict__ t1, int L1, cf* __restrict__ z, long n) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    float ar = 0.0f, ai = 0.0f;
    for (int k = 0; k < L1; k++) {
        const cf x = V.load(voff + m + L1 - 1 - k), t = t1[k];
        ar += t.x * x.x - t.y * x.y;
        ai += t.x * x.y + t.y * x.x;
    }
    z[m] = mkcf(ar, ai);
}
__device__ __forceinline__ cf head_y_at(const cf* __restrict__ z, const cf* __restrict__ t2, int L2, long i) {
    float ar = 0.0f, ai = 0.0f;
    const long jmax = i < L2 - 1 ? i : L2 - 1;
    for (long j = 0; j <= jmax; j++) {
        const cf x = z[i - j], t = t2[j];
        ar += t.x * x.x - t.y * x.y;
        ai += t.x * x.y + t.y * x.x;
    }
    return mkcf(ar, ai);
}
__global__ __launch_bounds__(256) void k_head_y(const cf* __restrict__ z, const cf* __restrict__ t2, int L2, cf* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = head_y_at(z, t2, L2, i);
}
__global__ __launch_bounds__(256) void k_head_demod(const cf* __restrict__ z, long nz, const cf* __restrict__ t2, int L2, long I, long D,
                                                    float gain, int mode, long r_hi, float* __restrict__ out, cf* __restrict__ last_r) {
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;        // resampled sample r[u] = y[floor(u D / I)]
    if (u >= r_hi) return;
    const long su = (long)(((__int128)u * D) / I);
    if (su >= nz) return;
    if (u == r_hi - 1 && su < L2 - 1) last_r[0] = head_y_at(z, t2, L2, su);
    if (u == 0) return;
    const long sl = (long)(((__int128)(u - 1) * D) / I);
    if (sl >= L2 - 1) return;                                        // neither sample touches the head
    const cf rl = head_y_at(z, t2, L2, sl), ru = head_y_at(z, t2, L2, su);
    const float na = -rl.y;
    const float re = sub_rn(mul_rn(rl.x, ru.x), mul_rn(na, ru.y));
    const float im = add_rn(mul_rn(rl.x, ru.y), mul_rn(na, ru.x));
    const float ang = mode == 0 ? atan2f(im, re) : fm_atan2(im, re);
    out[u - 1] = mul_rn(gain, ang);
}
void launch_head_z(VSrc<cf> V, long voff, const cf* t1, int L1, cf* z, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_head_z, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, V, voff, t1, L1, z, n);
    RR_HIP(hipGetLastError());
}
void launch_head_y(const cf* z, const cf* t2, int L2, cf* y, long n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_head_y, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, t2, L2, y, n);
    RR_HIP(hipGetLastError());
}
void launch_head_demod(const cf* z, long nz, const cf* t2, int L2, long I, long D, float gain, int mode, long r_hi, float* out,
                       cf* last_r, hipStream_t s) {
    // u with floor((u - 1) D / I) < L2 - 1  <=>  u - 1 < ceil((L2 - 1) I / D)
    long n = (long)((((__int128)(L2 - 1)) * I + D - 1) / D) + 2;
    if (n > r_hi) n = r_hi;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_head_demod, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, nz, t2, L2, I, D, gain, mode, r_hi, out, last_r);
    RR_HIP(hipGetLastError());
}

// ---- transforms of ANY size (FftStream / Fft beyond one LDS tile, FftFilter beyond 16383 taps) ---------------------------
// Sizes up to 16384 = 2^14 (and every size up to 2048) are one tile kernel (kernels_fft.hip).  Beyond that:
//   power of two N = N1 N2: the four-step transform — transpose, N2 x FFT_N1, twiddle w_N^(n2 k1) fused into the second
//       transpose, N1 x FFT_N2, last transpose to natural order (every pass a plain HBM stream of 16 B per sample);
//   any other N: Bluestein's chirp-z on a power-of-two M >= 2 N - 1 (n k = (n^2 + k^2 - (k - n)^2) / 2), the inverse
//       transform as conj(FFT(conj(.))).
// These are the small glue kernels; rustfft plans any size (fft_stream.rs:43-44, fft.rs:27-32) and so does the block now.

// out[f][c][r] = in[f][r][c] * (tw ? tw[(r * c) % (rows * cols)] : 1)      (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void k_transpose_tw(const cf* __restrict__ in, cf* __restrict__ out, int rows, int cols,
                                                      const cf* __restrict__ tw) {
    __shared__ cf tile[32][33];
    const long f = blockIdx.z;
    const cf* pin = in + f * (long)rows * cols;
    cf* pout = out + f * (long)rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) {
            cf v = pin[(long)r * cols + c];
            if (tw) {
                const cf w = tw[((long)r * c) % ((long)rows * cols)];
                v = mkcf(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            }
            tile[j][tx] = v;
        }
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) pout[(long)c * rows + r] = tile[tx][j];
    }
}
void launch_transpose_tw(const cf* in, cf* out, int rows, int cols, long nframes, const cf* tw, hipStream_t s) {
    if (nframes <= 0) return;
    for (long f0 = 0; f0 < nframes; f0 += 65535) {                   // gridDim.z limit
        const long nf = nframes - f0 < 65535 ? nframes - f0 : 65535;
        dim3 grid((cols + 31) / 32, (rows + 31) / 32, (unsigned)nf);
        hipLaunchKernelGGL(k_transpose_tw, grid, dim3(256), 0, s, in + f0 * (long)rows * cols, out + f0 * (long)rows * cols, rows, cols, tw);
    }
    RR_HIP(hipGetLastError());
}
// Bluestein glue (frames of N samples in, M-point work frames)
__global__ __launch_bounds__(256) void k_chirp_pre(const cf* __restrict__ in, cf* __restrict__ a, long N, long M, long total,
                                                   const cf* __restrict__ chirp) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / M, n = i - f * M;
        cf v = mkcf(0.0f, 0.0f);
        if (n < N) {
            const cf x = in[f * N + n], c = chirp[n];
            v = mkcf(x.x * c.x - x.y * c.y, x.x * c.y + x.y * c.x);
        }
        a[i] = v;
    }
}
// a[f][k] = conj(a[f][k] * b[k])   (b: M values; the conjugate sets up the inverse transform as a forward one)
__global__ __launch_bounds__(256) void k_mul_conj(cf* __restrict__ a, const cf* __restrict__ b, long M, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const cf x = a[i], w = b[i % M];
        a[i] = mkcf(x.x * w.x - x.y * w.y, -(x.x * w.y + x.y * w.x));
    }
}
__global__ __launch_bounds__(256) void k_chirp_post(const cf* __restrict__ y, cf* __restrict__ out, long N, long M, long total,
                                                    const cf* __restrict__ chirp) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / N, k = i - f * N;
        const cf v = y[f * M + k], c = chirp[k];
        out[i] = mkcf(v.x * c.x + v.y * c.y, v.x * c.y - v.y * c.x);       // conj(v) * c
    }
}
void launch_chirp_pre(const cf* in, cf* a, long N, long M, long nframes, const cf* chirp, hipStream_t s) {
    const long total = nframes * M;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_chirp_pre, dim3(grid_for(total, 256)), dim3(256), 0, s, in, a, N, M, total, chirp);
    RR_HIP(hipGetLastError());
}
void launch_mul_conj(cf* a, const cf* b, long M, long nframes, hipStream_t s) {
    const long total = nframes * M;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_mul_conj, dim3(grid_for(total, 256)), dim3(256), 0, s, a, b, M, total);
    RR_HIP(hipGetLastError());
}
void launch_chirp_post(const cf* y, cf* out, long N, long M, long nframes, const cf* chirp, hipStream_t s) {
    const long total = nframes * N;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_chirp_post, dim3(grid_for(total, 256)), dim3(256), 0, s, y, out, N, M, total, chirp);
    RR_HIP(hipGetLastError());
}
// overlap-save frames of a long filter: frames[f][i] = V[f S + i], i < M   (V = carry prefix ++ window, zero beyond)
__global__ __launch_bounds__(256) void k_ols_gather(VSrc<cf> src, cf* __restrict__ frames, long S, long M, long f0, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / M, n = i - f * M;
        frames[i] = src.load((f0 + f) * S + n);
    }
}
// out[(f0 + f) S + j] = conj(frames[f][L - 1 + j]), j < S, below n_out
__global__ __launch_bounds__(256) void k_ols_scatter(const cf* __restrict__ frames, cf* __restrict__ out, long S, long M, long L, long f0,
                                                     long n_out, long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / S, j = i - f * S;
        const long o = (f0 + f) * S + j;
        if (o < n_out) {
            const cf v = frames[f * M + L - 1 + j];
            out[o] = mkcf(v.x, -v.y);
        }
    }
}
void launch_ols_gather(VSrc<cf> src, cf* frames, long S, long M, long f0, long nframes, hipStream_t s) {
    const long total = nframes * M;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_ols_gather, dim3(grid_for(total, 256)), dim3(256), 0, s, src, frames, S, M, f0, total);
    RR_HIP(hipGetLastError());
}
void launch_ols_scatter(const cf* frames, cf* out, long S, long M, long L, long f0, long nframes, long n_out, hipStream_t s) {
    const long total = nframes * S;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_ols_scatter, dim3(grid_for(total, 256)), dim3(256), 0, s, frames, out, S, M, L, f0, n_out, total);
    RR_HIP(hipGetLastError());
}

}  // namespace rr
