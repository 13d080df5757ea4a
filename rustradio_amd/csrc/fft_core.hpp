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

struct alignas(8) cf { float x, y; };   // Complex<f32>, interleaved (src/lib.rs:268-271)
RR_HD cf mkcf(float x, float y) { cf r; r.x = x; r.y = y; return r; }

// Register type of the transform.  On the device a complex value lives in an aligned VGPR
// pair (ext_vector float2) so that complex add/sub are single v_pk_add_f32 and a complex
// multiply is v_pk_mul_f32 + v_pk_fma_f32: the +-j rotations and the (-im, re) swizzle of
// the multiply ride on the VOP3P op_sel / neg modifiers instead of costing instructions.
// hipcc only emits those modifiers for splats and negation, so the swizzled forms are
// written as (non-volatile, register-only) inline asm.  On the host (CPU emulation test)
// the same functions are plain C++.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float creg __attribute__((ext_vector_type(2)));
RR_HD creg mk(float x, float y) { creg r; r.x = x; r.y = y; return r; }
RR_HD creg cadd(creg a, creg b) { return a + b; }
RR_HD creg csub(creg a, creg b) { return a - b; }
// a + (-j) b = (a.x + b.y, a.y - b.x)
RR_HD creg add_mj(creg a, creg b) {
    creg r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+j) b = (a.x - b.y, a.y + b.x)
RR_HD creg add_pj(creg a, creg b) {
    creg r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * w = a * w.xx + (-a.y, a.x) * w.yy
RR_HD creg cmul(creg a, creg w) {
    creg t = a * __builtin_shufflevector(w, w, 0, 0), r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w) = a * w.xx + (a.y, -a.x) * w.yy
RR_HD creg cmulc(creg a, creg w) {
    creg t = a * __builtin_shufflevector(w, w, 0, 0), r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// Two independent products issued interleaved (mul, mul, fma, fma): a packed FMA that reads the result of the packed
// multiply right before it costs a wait state (hipcc puts an s_nop between them — 88 of the 787 issue slots of a 2048-point
// tile); with the partner's multiply in between there is nothing to wait for.
RR_HD void cmul2(creg& a0, creg w0, creg& a1, creg w1) {
    creg t0 = a0 * __builtin_shufflevector(w0, w0, 0, 0);
    creg t1 = a1 * __builtin_shufflevector(w1, w1, 0, 0);
    creg r0, r1;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r0) : "v"(a0), "v"(w0), "v"(t0));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r1) : "v"(a1), "v"(w1), "v"(t1));
    a0 = r0; a1 = r1;
}
RR_HD void cmulc2(creg& a0, creg w0, creg& a1, creg w1) {
    creg t0 = a0 * __builtin_shufflevector(w0, w0, 0, 0);
    creg t1 = a1 * __builtin_shufflevector(w1, w1, 0, 0);
    creg r0, r1;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r0) : "v"(a0), "v"(w0), "v"(t0));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r1) : "v"(a1), "v"(w1), "v"(t1));
    a0 = r0; a1 = r1;
}
// the same with a compile-time constant w: the pair rides in SGPRs (one constant-bus operand) instead of two VGPRs that
// the compiler would hoist out of the tile loop and keep for the whole kernel
RR_HD creg cmul_k(creg a, creg w) {
    creg t = a * __builtin_shufflevector(w, w, 0, 0), r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
RR_HD creg cmulc_k(creg a, creg w) {
    creg t = a * __builtin_shufflevector(w, w, 0, 0), r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=v"(r) : "v"(a), "s"(w), "v"(t));
    return r;
}
// a * k (real constant): ONE packed multiply (this file is compiled without SLP vectorisation, which would otherwise
// be the only thing fusing the two scalar products)
RR_HD creg cscale(creg a, float k) { return a * k; }
RR_HD creg to_reg(cf a) { return mk(a.x, a.y); }
RR_HD cf from_reg(creg a) { cf r; r.x = a.x; r.y = a.y; return r; }
#else
typedef cf creg;
RR_HD creg mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
RR_HD creg cadd(creg a, creg b) { return mk(a.x + b.x, a.y + b.y); }
RR_HD creg csub(creg a, creg b) { return mk(a.x - b.x, a.y - b.y); }
RR_HD creg add_mj(creg a, creg b) { return mk(a.x + b.y, a.y - b.x); }
RR_HD creg add_pj(creg a, creg b) { return mk(a.x - b.y, a.y + b.x); }
RR_HD creg cmul(creg a, creg b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
RR_HD creg cmulc(creg a, creg b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
RR_HD creg cscale(creg a, float k) { return mk(a.x * k, a.y * k); }
RR_HD void cmul2(creg& a0, creg w0, creg& a1, creg w1) { a0 = cmul(a0, w0); a1 = cmul(a1, w1); }
RR_HD void cmulc2(creg& a0, creg w0, creg& a1, creg w1) { a0 = cmulc(a0, w0); a1 = cmulc(a1, w1); }
RR_HD creg cmul_k(creg a, creg b) { return cmul(a, b); }
RR_HD creg cmulc_k(creg a, creg b) { return cmulc(a, b); }
RR_HD creg to_reg(cf a) { return a; }
RR_HD cf from_reg(creg a) { return a; }
#endif

// RR_LDS_Q = volatile keeps hipcc from fusing neighbouring 8-byte LDS accesses into
// ds_read2_b64 / ds_write2_b64 (a tuning experiment; see profiles/TUNING_LOG.md).
#ifndef RR_LDS_Q
#define RR_LDS_Q
#endif

constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kCos8 = 0.92387953251128675613f;   // cos(pi/8)
constexpr float kSin8 = 0.38268343236508977173f;   // sin(pi/8)

// a + w4 b and a - w4 b, w4 = -j (forward) or +j (inverse): the rotation is free
template <bool INV> RR_HD creg add_w4(creg a, creg b) { return INV ? add_pj(a, b) : add_mj(a, b); }
template <bool INV> RR_HD creg sub_w4(creg a, creg b) { return INV ? add_mj(a, b) : add_pj(a, b); }

// multiply by w16^M = exp(-2 pi i M / 16) (forward) or its conjugate (INV)
template <int M, bool INV> RR_HD creg mul_w16(creg a) {
    constexpr int m = ((M % 16) + 16) % 16;
    const creg zero = mk(0.0f, 0.0f);
    if constexpr (m == 0) return a;
    else if constexpr (m == 4) return add_w4<INV>(zero, a);
    else if constexpr (m == 8) return csub(zero, a);
    else if constexpr (m == 12) return sub_w4<INV>(zero, a);
    else if constexpr (m == 2) {   // (1 -+ j)/sqrt2 : (a + w4 a) / sqrt2
        creg t = add_w4<INV>(a, a);
        return cscale(t, kSqrtHalf);
    } else if constexpr (m == 6) { // (-1 -+ j)/sqrt2 : (w4 a - a) / sqrt2 = -(a - w4 a)/sqrt2
        creg t = sub_w4<INV>(a, a);
        return cscale(t, -kSqrtHalf);
    } else if constexpr (m == 10) {
        creg t = add_w4<INV>(a, a);
        return cscale(t, -kSqrtHalf);
    } else if constexpr (m == 14) {
        creg t = sub_w4<INV>(a, a);
        return cscale(t, kSqrtHalf);
    } else {
        // generic: forward w = (c, -s), s = sin(pi m / 8), c = cos(pi m / 8)
        constexpr float c = (m == 1 || m == 15) ? kCos8 : (m == 3 || m == 13) ? kSin8
                          : (m == 5 || m == 11) ? -kSin8 : -kCos8;            // m == 7, 9
        constexpr float s = (m == 1 || m == 7) ? kSin8 : (m == 3 || m == 5) ? kCos8
                          : (m == 9 || m == 15) ? -kSin8 : -kCos8;            // m == 11, 13
        const creg w = mk(c, -s);
        return INV ? cmulc_k(a, w) : cmul_k(a, w);
    }
}

template <bool INV> RR_HD void bfly2(creg& a, creg& b) {
    creg t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// natural-order 4-point DFT in place
template <bool INV> RR_HD void bfly4(creg& a0, creg& a1, creg& a2, creg& a3) {
    creg t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = add_w4<INV>(t1, t3);
    a3 = sub_w4<INV>(t1, t3);
}

// R-point DFT of v[0..R) (stride 1 in the register array), natural order in and out.
template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<2, INV> {
    static RR_HD void run(creg* v) { bfly2<INV>(v[0], v[1]); }
};
template <bool INV> struct Dft<4, INV> {
    static RR_HD void run(creg* v) { bfly4<INV>(v[0], v[1], v[2], v[3]); }
};
template <bool INV> struct Dft<8, INV> {
    // n = 2 n1 + n2 (N1 = 4, N2 = 2), k = k1 + 4 k2
    static RR_HD void run(creg* v) {
        bfly4<INV>(v[0], v[2], v[4], v[6]);
        bfly4<INV>(v[1], v[3], v[5], v[7]);
        v[3] = mul_w16<2, INV>(v[3]);   // w8^1
        v[5] = mul_w16<4, INV>(v[5]);   // w8^2
        v[7] = mul_w16<6, INV>(v[7]);   // w8^3
        bfly2<INV>(v[0], v[1]); bfly2<INV>(v[2], v[3]); bfly2<INV>(v[4], v[5]); bfly2<INV>(v[6], v[7]);
        // v[2 k1 + k2] = X[k1 + 4 k2]  ->  natural order
        creg t1 = v[1], t2 = v[2], t3 = v[3], t4 = v[4], t5 = v[5], t6 = v[6];
        v[1] = t2; v[2] = t4; v[3] = t6; v[4] = t1; v[5] = t3; v[6] = t5;
    }
};
template <bool INV> struct Dft<16, INV> {
    // n = 4 n1 + n2, k = k1 + 4 k2
    static RR_HD void run(creg* v) {
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
        creg t;
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
// conflict-free for ds_read_b64/ds_write_b64; see profiles/TUNING_LOG.md).
RR_HD int lds_pad(int a) { return a + (a >> 4); }
constexpr int lds_elems(int F) { return F + (F >> 4); }

// Per-thread twiddles of pass I: twl[k-1] = w_{R P}^{k lo} = tw[k * lo * TWSTRIDE], k = 1..15.
// Only radix-16 passes (one group per thread) carry twiddles in every Plan above.
template <int LOG2F, int I> RR_HD constexpr bool pass_has_twiddles() { return PassGeom<LOG2F, I>::P > 1; }
template <int LOG2F, int I> RR_HD void load_twiddles(creg* twl, int t, const cf* __restrict__ tw) {
    using G = PassGeom<LOG2F, I>;
    static_assert(G::P == 1 || G::U == 1, "twiddled passes must be radix 16");
    if constexpr (G::P > 1) {
        const int lo = G::lo(t);
#pragma unroll
        for (int k = 1; k < 16; k++) twl[k - 1] = to_reg(tw[k * lo * G::TWSTRIDE]);
    }
}
// One forward pass on the 16 registers of a thread.  `v[u*R + n]` = element n of group
// (t + T u); twl = this thread's twiddles for the pass (unused when P == 1).
template <int LOG2F, int I> RR_HD void fwd_pass(creg* v, const creg* twl) {
    using G = PassGeom<LOG2F, I>;
#pragma unroll
    for (int u = 0; u < G::U; u++) Dft<G::R, false>::run(v + u * G::R);
    if constexpr (G::P > 1) {
        v[1] = cmul(v[1], twl[0]);
#pragma unroll
        for (int k = 2; k < 16; k += 2) cmul2(v[k], twl[k - 1], v[k + 1], twl[k]);
    }
}
// Mirror: conj twiddle first, then inverse DFT of the digit.
template <int LOG2F, int I> RR_HD void inv_pass(creg* v, const creg* twl) {
    using G = PassGeom<LOG2F, I>;
    if constexpr (G::P > 1) {
        v[1] = cmulc(v[1], twl[0]);
#pragma unroll
        for (int k = 2; k < 16; k += 2) cmulc2(v[k], twl[k - 1], v[k + 1], twl[k]);
    }
#pragma unroll
    for (int u = 0; u < G::U; u++) Dft<G::R, true>::run(v + u * G::R);
}

// LDS addressing of a pass layout, split into ONE per-thread base and compile-time
// offsets so that every ds_read/ds_write uses base VGPR + immediate:
//   lds_pad(pos(t + T u, n)) == lds_base<I>(t) + lds_off<I>(u, n)
// (holds because every twiddled pass has U == 1 and every multi-group pass has P == 1,
//  and n*P never carries into bit 4 together with the group's low part; see profiles/TUNING_LOG.md).
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
template <int LOG2F, int I> RR_HD void lds_store(const creg* v, int t, creg* lds) {
    using G = PassGeom<LOG2F, I>;
    RR_LDS_Q creg* b = lds + lds_base<LOG2F, I>(t);
#pragma unroll
    for (int u = 0; u < G::U; u++)
#pragma unroll
        for (int n = 0; n < G::R; n++) b[lds_off<LOG2F, I>(u, n)] = v[u * G::R + n];
}
template <int LOG2F, int I> RR_HD void lds_load(creg* v, int t, const creg* lds) {
    using G = PassGeom<LOG2F, I>;
    const RR_LDS_Q creg* b = lds + lds_base<LOG2F, I>(t);
#pragma unroll
    for (int u = 0; u < G::U; u++)
#pragma unroll
        for (int n = 0; n < G::R; n++) v[u * G::R + n] = b[lds_off<LOG2F, I>(u, n)];
}
// H in position order: this thread's 16 values for the layout of pass I
template <int LOG2F, int I> RR_HD void load_h(creg* h, int t, const cf* __restrict__ hpos) {
    using G = PassGeom<LOG2F, I>;
#pragma unroll
    for (int u = 0; u < G::U; u++)
#pragma unroll
        for (int n = 0; n < G::R; n++) h[u * G::R + n] = to_reg(hpos[G::pos(t + G::T * u, n)]);
}
RR_HD void apply_h(creg* v, const creg* h) {
#pragma unroll
    for (int n = 0; n < 16; n += 2) cmul2(v[n], h[n], v[n + 1], h[n + 1]);
}

}  // namespace rr
