// fft_core.hpp — register-resident radix-2/4/8/16 butterflies and the in-place
// mixed-radix pass geometry used by the FftFilter kernels (kernels_fft.hip).
//
// Design (MI355X-first, not a translation of rustfft): one workgroup of F/16
// threads owns one F-point tile; every thread keeps 16 complex values in VGPRs.
// The forward transform is an in-place decimation-in-frequency mixed-radix FFT
// whose output is left in digit-reversed order; the frequency response H is
// stored in that same order, and the inverse transform is the exact mirror
// (decimation-in-time, digit-reversed in -> natural out).  That removes every
// reordering pass: data only crosses LDS where the next radix group needs
// values held by other lanes.
//
// Everything here is __host__ __device__ so tests/cpu_emulate_fft.cpp can run the
// same pass functions thread-by-thread on the CPU.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RR_HD __host__ __device__ __forceinline__
#else
#define RR_HD inline
#endif

namespace rr {

struct cf { float x, y; };

RR_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
RR_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
RR_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
RR_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
RR_HD cf cmulc(cf a, cf b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
RR_HD cf mul_mj(cf a) { return mk(a.y, -a.x); }   // -j * a
RR_HD cf mul_pj(cf a) { return mk(-a.y, a.x); }   // +j * a
template <bool INV> RR_HD cf mul_w4(cf a) { return INV ? mul_pj(a) : mul_mj(a); }  // w4^1 (or conj)

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kCos8 = 0.92387953251128675613f;   // cos(pi/8)
constexpr float kSin8 = 0.38268343236508977173f;   // sin(pi/8)

// multiply by w16^M = exp(-2 pi i M / 16) (forward) or its conjugate (INV)
template <int M, bool INV> RR_HD cf mul_w16(cf a) {
    constexpr int m = ((M % 16) + 16) % 16;
    if constexpr (m == 0) return a;
    else if constexpr (m == 4) return mul_w4<INV>(a);
    else if constexpr (m == 8) return mk(-a.x, -a.y);
    else if constexpr (m == 12) return mul_w4<!INV>(a);
    else if constexpr (m == 2) {  // (1 - j)/sqrt2 fwd
        return INV ? mk((a.x - a.y) * kSqrtHalf, (a.x + a.y) * kSqrtHalf)
                   : mk((a.x + a.y) * kSqrtHalf, (a.y - a.x) * kSqrtHalf);
    } else if constexpr (m == 6) {  // (-1 - j)/sqrt2 fwd
        return INV ? mk((-a.x - a.y) * kSqrtHalf, (a.x - a.y) * kSqrtHalf)
                   : mk((a.y - a.x) * kSqrtHalf, (-a.x - a.y) * kSqrtHalf);
    } else if constexpr (m == 10) { cf t = mul_w16<2, INV>(a); return mk(-t.x, -t.y); }
    else if constexpr (m == 14) { cf t = mul_w16<6, INV>(a); return mk(-t.x, -t.y); }
    else {
        // generic: w = (c, -s) forward
        constexpr float c = (m == 1 || m == 15) ? kCos8 : (m == 3 || m == 13) ? kSin8
                          : (m == 5 || m == 11) ? -kSin8 : -kCos8;            // m == 7, 9
        constexpr float s = (m == 1 || m == 7) ? kSin8 : (m == 3 || m == 5) ? kCos8
                          : (m == 9 || m == 15) ? -kSin8 : -kCos8;            // m == 11, 13
        const cf w = mk(c, INV ? s : -s);
        return cmul(a, w);
    }
}

template <bool INV> RR_HD void bfly2(cf& a, cf& b) {
    cf t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// natural-order 4-point DFT in place
template <bool INV> RR_HD void bfly4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_w4<INV>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

// R-point DFT of v[0..R) (stride 1 in the register array), natural order in and out.
template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<2, INV> {
    static RR_HD void run(cf* v) { bfly2<INV>(v[0], v[1]); }
};
template <bool INV> struct Dft<4, INV> {
    static RR_HD void run(cf* v) { bfly4<INV>(v[0], v[1], v[2], v[3]); }
};
template <bool INV> struct Dft<8, INV> {
    // n = 2 n1 + n2 (N1 = 4, N2 = 2), k = k1 + 4 k2
    static RR_HD void run(cf* v) {
        bfly4<INV>(v[0], v[2], v[4], v[6]);
        bfly4<INV>(v[1], v[3], v[5], v[7]);
        v[3] = mul_w16<2, INV>(v[3]);   // w8^1
        v[5] = mul_w16<4, INV>(v[5]);   // w8^2
        v[7] = mul_w16<6, INV>(v[7]);   // w8^3
        bfly2<INV>(v[0], v[1]); bfly2<INV>(v[2], v[3]); bfly2<INV>(v[4], v[5]); bfly2<INV>(v[6], v[7]);
        // v[2 k1 + k2] = X[k1 + 4 k2]  ->  natural order
        cf t1 = v[1], t2 = v[2], t3 = v[3], t4 = v[4], t5 = v[5], t6 = v[6];
        v[1] = t2; v[2] = t4; v[3] = t6; v[4] = t1; v[5] = t3; v[6] = t5;
    }
};
template <bool INV> struct Dft<16, INV> {
    // n = 4 n1 + n2, k = k1 + 4 k2
    static RR_HD void run(cf* v) {
        bfly4<INV>(v[0], v[4], v[8], v[12]);
        bfly4<INV>(v[1], v[5], v[9], v[13]);
        bfly4<INV>(v[2], v[6], v[10], v[14]);
        bfly4<INV>(v[3], v[7], v[11], v[15]);
        // v[4 k1 + n2] *= w16^(n2 k1)
        v[5] = mul_w16<1, INV>(v[5]);   v[6] = mul_w16<2, INV>(v[6]);   v[7] = mul_w16<3, INV>(v[7]);
        v[9] = mul_w16<2, INV>(v[9]);   v[10] = mul_w16<4, INV>(v[10]); v[11] = mul_w16<6, INV>(v[11]);
        v[13] = mul_w16<3, INV>(v[13]); v[14] = mul_w16<6, INV>(v[14]); v[15] = mul_w16<9, INV>(v[15]);
        bfly4<INV>(v[0], v[1], v[2], v[3]);
        bfly4<INV>(v[4], v[5], v[6], v[7]);
        bfly4<INV>(v[8], v[9], v[10], v[11]);
        bfly4<INV>(v[12], v[13], v[14], v[15]);
        // v[4 k1 + k2] = X[k1 + 4 k2]  -> transpose 4x4 to natural order
        cf t;
        t = v[1]; v[1] = v[4]; v[4] = t;
        t = v[2]; v[2] = v[8]; v[8] = t;
        t = v[3]; v[3] = v[12]; v[12] = t;
        t = v[6]; v[6] = v[9]; v[9] = t;
        t = v[7]; v[7] = v[13]; v[13] = t;
        t = v[11]; v[11] = v[14]; v[14] = t;
    }
};

// ---- pass geometry ----------------------------------------------------------------
// F = R1 * R2 * ... * Rm.  Index n = sum_i n_i * P_i with P_i = R_{i+1} ... R_m.
// Pass i transforms digit i in place (n_i -> k_i); output bin k = k1 + R1 k2 + R1 R2 k3 ...
// sits at position sum_i k_i P_i ("digit reversed").
template <int LOG2F> struct Plan;
template <> struct Plan<10> { static constexpr int NP = 3; static constexpr int R[4] = {16, 16, 4, 1}; };
template <> struct Plan<11> { static constexpr int NP = 3; static constexpr int R[4] = {16, 16, 8, 1}; };
template <> struct Plan<12> { static constexpr int NP = 3; static constexpr int R[4] = {16, 16, 16, 1}; };
template <> struct Plan<13> { static constexpr int NP = 4; static constexpr int R[4] = {16, 16, 16, 2}; };
template <> struct Plan<14> { static constexpr int NP = 4; static constexpr int R[4] = {16, 16, 16, 4}; };

template <int LOG2F, int I> struct PassGeom {
    using P_ = Plan<LOG2F>;
    static constexpr int F = 1 << LOG2F;
    static constexpr int T = F / 16;                 // threads per tile
    static constexpr int R = P_::R[I];
    static constexpr int U = 16 / R;                 // groups per thread
    static constexpr int prodAfter() { int p = 1; for (int l = I + 1; l < P_::NP; l++) p *= P_::R[l]; return p; }
    static constexpr int P = prodAfter();            // stride of digit I
    static constexpr int TWSTRIDE = F / (R * P);     // twiddle table step: w_{R P}^1 = w_F^TWSTRIDE
    // in-place position of element n of group g
    static RR_HD int pos(int g, int n) { return (g / P) * (R * P) + (g % P) + n * P; }
    static RR_HD int lo(int g) { return g % P; }
};

// position -> frequency bin for the digit-reversed layout
template <int LOG2F> RR_HD int bin_of_pos(int pos) {
    using P_ = Plan<LOG2F>;
    int Pi = 1 << LOG2F, k = 0, mul = 1;
    for (int i = 0; i < P_::NP; i++) {
        Pi /= P_::R[i];
        int d = (pos / Pi) % P_::R[i];
        k += d * mul;
        mul *= P_::R[i];
    }
    return k;
}

// LDS padding: one 8-byte slot per 16 elements (keeps stride-P and stride-R accesses
// conflict-free for ds_read_b64/ds_write_b64).
RR_HD int lds_pad(int a) { return a + (a >> 4); }
constexpr int lds_elems(int F) { return F + (F >> 4); }

// Per-thread twiddles of pass I: twl[k-1] = w_{R P}^{k lo} = tw[k * lo * TWSTRIDE], k = 1..15.
// Only radix-16 passes (one group per thread) carry twiddles in every Plan above.
template <int LOG2F, int I> RR_HD constexpr bool pass_has_twiddles() { return PassGeom<LOG2F, I>::P > 1; }
template <int LOG2F, int I> RR_HD void load_twiddles(cf* twl, int t, const cf* __restrict__ tw) {
    using G = PassGeom<LOG2F, I>;
    static_assert(G::P == 1 || G::U == 1, "twiddled passes must be radix 16");
    if constexpr (G::P > 1) {
        const int lo = G::lo(t);
#pragma unroll
        for (int k = 1; k < 16; k++) twl[k - 1] = tw[k * lo * G::TWSTRIDE];
    }
}
// One forward pass on the 16 registers of a thread.  `v[u*R + n]` = element n of group
// (t + T u); twl = this thread's twiddles for the pass (unused when P == 1).
template <int LOG2F, int I> RR_HD void fwd_pass(cf* v, const cf* twl) {
    using G = PassGeom<LOG2F, I>;
#pragma unroll
    for (int u = 0; u < G::U; u++) Dft<G::R, false>::run(v + u * G::R);
    if constexpr (G::P > 1) {
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmul(v[k], twl[k - 1]);
    }
}
// Mirror: conj twiddle first, then inverse DFT of the digit.
template <int LOG2F, int I> RR_HD void inv_pass(cf* v, const cf* twl) {
    using G = PassGeom<LOG2F, I>;
    if constexpr (G::P > 1) {
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], twl[k - 1]);
    }
#pragma unroll
    for (int u = 0; u < G::U; u++) Dft<G::R, true>::run(v + u * G::R);
}
// LDS addressing of a pass layout, split into ONE per-thread base and compile-time
// offsets so that every ds_read/ds_write uses base VGPR + immediate:
//   lds_pad(pos(t + T u, n)) == lds_base<I>(t) + lds_off<I>(u, n)
// (holds because every twiddled pass has U == 1 and every multi-group pass has P == 1,
//  and n*P never carries into bit 4 together with the group's low part).
template <int LOG2F, int I> RR_HD int lds_base(int t) {
    using G = PassGeom<LOG2F, I>;
    static_assert(G::U == 1 || G::P == 1, "plan shape");
    if constexpr (G::P == 1) return G::R * t + ((G::R * t) >> 4);
    else return lds_pad((t / G::P) * (G::R * G::P) + (t % G::P));
}
template <int LOG2F, int I> RR_HD constexpr int lds_off(int u, int n) {
    using G = PassGeom<LOG2F, I>;
    if (G::P == 1) return n + u * G::R * G::T + ((u * G::R * G::T) >> 4);
    return n * G::P + ((n * G::P) >> 4);
}
template <int LOG2F, int I> RR_HD void lds_store(const cf* v, int t, cf* lds) {
    using G = PassGeom<LOG2F, I>;
    cf* b = lds + lds_base<LOG2F, I>(t);
#pragma unroll
    for (int u = 0; u < G::U; u++)
#pragma unroll
        for (int n = 0; n < G::R; n++) b[lds_off<LOG2F, I>(u, n)] = v[u * G::R + n];
}
template <int LOG2F, int I> RR_HD void lds_load(cf* v, int t, const cf* lds) {
    using G = PassGeom<LOG2F, I>;
    const cf* b = lds + lds_base<LOG2F, I>(t);
#pragma unroll
    for (int u = 0; u < G::U; u++)
#pragma unroll
        for (int n = 0; n < G::R; n++) v[u * G::R + n] = b[lds_off<LOG2F, I>(u, n)];
}
// H in position order: this thread's 16 values for the layout of pass I
template <int LOG2F, int I> RR_HD void load_h(cf* h, int t, const cf* __restrict__ hpos) {
    using G = PassGeom<LOG2F, I>;
#pragma unroll
    for (int u = 0; u < G::U; u++)
#pragma unroll
        for (int n = 0; n < G::R; n++) h[u * G::R + n] = hpos[G::pos(t + G::T * u, n)];
}
RR_HD void apply_h(cf* v, const cf* h) {
#pragma unroll
    for (int n = 0; n < 16; n++) v[n] = cmul(v[n], h[n]);
}

}  // namespace rr
