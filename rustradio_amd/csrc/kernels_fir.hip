// kernels_fir.hip — direct-form FIR on gfx950: FirFilter<Complex>/<Float> incl. decimation
// (/root/reference/src/fir.rs:166-197, 492-550) and the Hilbert block's inner filter
// (/root/reference/src/hilbert.rs:113-116).
//
// y[m] = sum_k rev[k] * x[m*d + k]  is evaluated polyphase: for phase p < d the taps
// rev[q*d + p] run over the decimated sequence x_p[n] = x[n*d + p] as a d=1 FIR, so every
// thread keeps a sliding window of R inputs in VGPRs and needs ONE new LDS value per tap
// for R multiply-adds.  The input tile is staged in LDS transposed ([n % R][n / R]) so
// that lane t's window read x_p[t*R + c] is lane-consecutive (conflict-free).  Taps are
// wave-uniform and come through the scalar cache.  No MFMA: a FIR is a vector
// contraction, VALU (FP32) bound for long filters — see DESIGN.md for the roofline.
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "nan_fix.hpp"

namespace rr {

__device__ __forceinline__ void mac(cf& acc, float tap, cf w) {   // real tap (one v_pk_fma_f32)
    acc.x = fmaf(tap, w.x, acc.x);
    acc.y = fmaf(tap, w.y, acc.y);
}
__device__ __forceinline__ void mac(cf& acc, cf tap, cf w) {      // complex tap (4 FMA)
    acc.x = fmaf(tap.x, w.x, acc.x);
    acc.x = fmaf(-tap.y, w.y, acc.x);
    acc.y = fmaf(tap.x, w.y, acc.y);
    acc.y = fmaf(tap.y, w.x, acc.y);
}
__device__ __forceinline__ void mac(float& acc, float tap, float w) { acc = fmaf(tap, w, acc); }
__device__ __forceinline__ void mac(cf& acc, cf tap, float w) {   // complex tap, real sample (one v_pk_fma_f32)
    acc.x = fmaf(tap.x, w, acc.x);
    acc.y = fmaf(tap.y, w, acc.y);
}

__device__ __forceinline__ float add_of(float a, float b) { return a + b; }
__device__ __forceinline__ cf add_of(cf a, cf b) { return mkcf(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void opaque(cf& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
template <class T> __device__ __forceinline__ T zero_of();
template <> __device__ __forceinline__ float zero_of<float>() { return 0.0f; }
template <> __device__ __forceinline__ cf zero_of<cf>() { return mkcf(0.0f, 0.0f); }

// Tile geometry shared by host and device.  NT threads (per phase group) x R outputs each = NOUT outputs per tile.
//   input  LDS: per phase p, x_p[n] at  p*pstride + (n % R)*rstride + n / R      (transposed)
//   output LDS: output i of the tile at (i / R)*(R + 1) + i % R                    (re-coalescing)
struct FirGeom {
    int np;        // samples per phase staged in LDS
    int rstride;   // row stride (elements) of the transposed tile
    int pstride;   // phase stride
    size_t lds_bytes;
};
static FirGeom fir_geom(int NT, int R, int d, int qpad, size_t es_in, size_t es_out, int S = 1) {
    FirGeom g;
    g.np = NT * R + qpad;
    int rs = NT + qpad / R;
    // bank-friendly residue: the R rows of a phase must start in different banks, r*rs (in 8-byte
    // units for Complex, 4-byte for Float) a permutation of the even / multiple-of-4 residues
    const int mod = es_in == 8 ? 4 : 8, want = es_in == 8 ? 2 : 4;
    while (rs % mod != want) rs++;
    g.rstride = rs;
    g.pstride = R * rs + 1;
    const size_t in_b = (size_t)d * g.pstride * es_in;
    const size_t out_b = (size_t)NT * S * (R + 1) * es_out;
    g.lds_bytes = in_b > out_b ? in_b : out_b;
    return g;
}

// Staging is software-pipelined: the next tile's input is fetched into registers (PRE values
// per thread, all loads in flight) while the current tile is multiplied out, and only written to
// LDS after the barrier that ends the current tile.
constexpr int FIR_MAXPRE = 20;

//
// Phase split: a decimating filter's input tile is d times its output tile, so LDS capacity caps the
// outputs per tile and, with one thread per R outputs, R itself (R = 2 at d = 8), which leaves ONE
// packed multiply-add pair per LDS read.  With S > 1 the workgroup is S groups of NT/S threads; group
// s evaluates only the phases p = s (mod S) for its R outputs and the S partial sums meet in the
// output staging buffer.  Same tile, same thread count, S times fewer LDS reads per multiply-add.
// A group is a whole number of waves, so taps stay wave-uniform.
template <class T, class TapT, class OutT, int NT, int R, int S, int PRE, int RSC, bool HILBERT>
__global__ __launch_bounds__(NT, 4) void k_fir(NanFixCtx nfx, VSrc<T> src, OutT* __restrict__ out, long n_out, int L, int d,
                                            int qpad, int np, int rstride_arg, int pstride_arg,
                                            const TapT* __restrict__ tp) {
    (void)nfx;                                         // (nan_fix.hpp: read from the argument segment by nf_finish, never by the body)
    using AccT = typename std::conditional<HILBERT, T, OutT>::type;   // real samples x complex taps accumulate Complex
    // RSC > 0: the row stride is a compile-time constant, so the R window reads of a tap block are one
    // address register + immediates (the d = 1 shapes; the host sizes the tile for it)
    const int rstride = RSC > 0 ? RSC : rstride_arg;
    const int pstride = RSC > 0 ? R * RSC + 1 : pstride_arg;
    constexpr int UNROLL_TAPS = (R == 8 && S <= 2 && sizeof(T) == 8 && RSC == 0) ? 1 : 2;
    constexpr int NTC = NT / S;                        // threads (output columns) per phase group
    constexpr int NOUT = NTC * R;
    static_assert(R % 2 == 0 && R <= 8, "qpad is padded to a multiple of 8");
    static_assert(NTC % 64 == 0 && R % S == 0 && (S == 1 || !HILBERT), "phase groups are whole waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    OutT* lds_o = reinterpret_cast<OutT*>(smem_raw);
    const int t = threadIdx.x;
    const long ntiles = (n_out + NOUT - 1) / NOUT;
    nf_init();                                         // (the zero-padded taps spread a NaN up to 8 outputs too far)
    const int total = np * d;
    const int cnt_k = (total + NT - 1) / NT;           // staged values per thread
    const bool piped = cnt_k <= PRE;                   // PRE = staged values per thread the kernel is built for

    T pre[PRE];
    // Only INTERIOR tiles (whole window inside the caller's buffer) are pipelined through registers;
    // the few tiles that touch the carried history or the end of the window are staged in place by
    // stage_direct().  The test is workgroup-uniform and the loads below are straight-line: a
    // per-element in-range select compiles to an exec-masked branch per load and the loads then
    // complete one at a time (measured: 0.43 of 0.61 ms of the 255-tap /8 filter was that).
    auto interior = [&](long tile) {
        const long gi0 = tile * NOUT * d;
        return piped && gi0 >= src.plen && gi0 - src.plen + total <= src.in_len;
    };
    auto fetch = [&](long tile) {                       // tile input -> registers (no waits)
        if (!interior(tile)) return;
        const T* gp = src.in + (tile * NOUT * d - src.plen);
        unsigned tt = t;
        asm volatile("" : "+v"(tt));                    // recompute the offsets per tile: hoisted, they cost 20 VGPRs
        int cnt = cnt_k;
        asm volatile("" : "+s"(cnt));                   // ... and the 20 round predicates 40 SGPRs (spilled to lanes)
#pragma unroll
        for (int c = 0; c < PRE; c++) {
            if (c < cnt - 1) {
                pre[c] = (gp + c * NT)[tt];             // uniform base + lane offset: no per-load address math
            } else if (c == cnt - 1) {                  // ragged last round: clamped, commit() skips the slot
                const unsigned i = tt + c * NT;
                pre[c] = gp[i < (unsigned)total ? i : (unsigned)total - 1];
            }
        }
    };
    // Element i = t + c*NT of the tile is x_p[n] with p = i % d, n = i / d and goes to
    // lds[p*pstride + (n % R)*rstride + n / R].  When d divides NT and R divides NT/d (all power-of-two
    // decimations up to NT/R, and d = 1) the slot of round c is slot(0) + c * NT/d/R: one add per value.
    const bool fast_walk = NT % d == 0 && (NT / d) % R == 0;
    auto commit = [&]() {                               // registers -> transposed LDS tile
        unsigned tt = t;
        asm volatile("" : "+v"(tt));                    // (same: keep the 20 LDS addresses out of the live set)
        int cnt = cnt_k;
        asm volatile("" : "+s"(cnt));
        const unsigned ud = d, sp = NT % ud, sn = NT / ud;
        unsigned p = tt % ud, n = tt / ud;
        if (fast_walk) {
            T* slot = lds + p * pstride + (n % R) * rstride + n / R;
            const unsigned K = sn / R;
#pragma unroll
            for (int c = 0; c < PRE; c++) {
                if (c < cnt - 1) slot[c * K] = pre[c];
                else if (c == cnt - 1 && tt + c * NT < (unsigned)total) slot[c * K] = pre[c];
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < PRE; c++) {
            if (c < cnt - 1 || (c == cnt - 1 && tt + c * NT < (unsigned)total))
                lds[p * pstride + (n % R) * rstride + n / R] = pre[c];
            p += sp; n += sn;
            if (p >= ud) { p -= ud; n++; }
        }
    };
    auto stage_direct = [&](long tile) {                // boundary tiles and very large tiles
        const long gi0 = tile * NOUT * d;
        int p = t % d, n = t / d;
        const int sp = NT % d, sn = NT / d;
        for (int i = t; i < total; i += NT) {
            lds[p * pstride + (n % R) * rstride + n / R] = src.load(gi0 + i);
            p += sp; n += sn;
            if (p >= d) { p -= d; n++; }
        }
    };

    // The outputs of a tile are held in registers across the loop back-edge and stored only AFTER the next
    // tile's input has been committed: vmcnt counts loads and stores in one in-order queue, so a store
    // issued between a tile's fetch and its commit makes the commit wait for the store's completion as
    // well (the compiler emits vmcnt(0)) — measured 0.88 -> see profiles/TUNING_LOG.md for the 127-tap d = 1 filter.
    OutT hold[R / S];
    long hold_m0 = -1;
    auto flush = [&]() {
        if (hold_m0 < 0) return;
        const bool whole = hold_m0 + NOUT <= n_out;      // workgroup-uniform: unconditional stores for full tiles
#pragma unroll
        for (int c = 0; c < R / S; c++) {
            const int i = c * NT + t;
            if (whole || hold_m0 + i < n_out) out[hold_m0 + i] = hold[c];
        }
    };

    long tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const long m0 = tile * NOUT;
        __syncthreads();                       // previous tile's output reads are done
        if (interior(tile)) commit(); else stage_direct(tile);
        flush();                               // previous tile's outputs: behind this tile's loads in the queue
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);

        const int tc = S == 1 ? t : t % NTC;                                   // output column
        const int sg = S == 1 ? 0 : __builtin_amdgcn_readfirstlane(t / NTC);   // phase group (wave-uniform)
        AccT acc[R];
#pragma unroll
        for (int j = 0; j < R; j++) acc[j] = zero_of<AccT>();
        for (int p = sg; p < d; p += S) {
            const T* lp = lds + p * pstride + tc;
            const TapT* tpp = tp + (long)p * qpad;
            T w[R];
#pragma unroll
            for (int j = 0; j < R; j++) w[j] = lp[j * rstride];
            // 8 taps per iteration whatever R is (qpad is a multiple of 8): one wide scalar tap load and
            // 8 LDS reads in flight per 8*R multiply-adds.  Tap q needs x_p[t*R + q + R] next: row
            // (q % R), column tc + q/R + 1 of the transposed tile.  (Software-pipelining the loop by one
            // block — next block's samples and taps issued before this block's multiply-adds — measured
            // slower twice: +32 VGPRs cost a wave per SIMD and the scalar tap loads force lgkmcnt(0) anyway.)
            // two blocks per iteration when the registers allow it (R = 8 without a phase split does not:
            // the 128-VGPR bound then spills, and every scratch access waits vmcnt(0), i.e. for the
            // prefetched tile — measured 1.02 vs 0.6 ms for 127 taps at d = 1)
#pragma unroll UNROLL_TAPS
            for (int q0 = 0; q0 < qpad; q0 += 8) {
                const T* lq = lp + q0 / R + 1;
                TapT tap8[8];
#pragma unroll
#ifdef RR_FIR_EXP_NOTAPLOAD      /* measurement build: the same 8 taps every block (hoisted out of the loop) */
                for (int kk = 0; kk < 8; kk++) tap8[kk] = tpp[kk];
#else
                for (int kk = 0; kk < 8; kk++) tap8[kk] = tpp[q0 + kk];
#endif
#pragma unroll
                for (int kk = 0; kk < 8; kk++) {
#pragma unroll
                    for (int j = 0; j < R; j++) mac(acc[j], tap8[kk], w[(kk + j) % R]);
#ifdef RR_FIR_EXP_NOLDS          /* measurement build: the window is never refilled */
                    opaque(w[kk % R]);
#else
                    w[kk % R] = lq[(kk % R) * rstride + kk / R];
#endif
                }
            }
        }
        // ---- re-coalesce the outputs through LDS: thread (sg, tc) holds group sg's partial sums of
        //      outputs tc*R .. tc*R+R-1; the reader adds the S partials ----
        OutT res[R];
        if constexpr (HILBERT) {               // re = xp[k + L/2]  (hilbert.rs:115)
#pragma unroll
            for (int j = 0; j < R; j++) {
                const int n = t * R + j + L / 2;
                res[j] = mkcf(lds[(n % R) * rstride + n / R], acc[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) res[j] = acc[j];
        }
        __syncthreads();                       // everybody is done reading the input tile
#pragma unroll
        for (int j = 0; j < R; j++) lds_o[t * (R + 1) + j] = res[j];
        __syncthreads();
        bool tile_bad = false;
#pragma unroll
        for (int c = 0; c < R / S; c++) {
            const int i = c * NT + t;
            OutT v = lds_o[(i / R) * (R + 1) + i % R];
            if constexpr (S > 1) {
#pragma unroll
                for (int g = 1; g < S; g++) v = add_of(v, lds_o[(g * NTC + i / R) * (R + 1) + i % R]);
            }
            tile_bad |= nf_bad(v);
            hold[c] = v;
        }
        nf_mark(tile_bad);
        hold_m0 = m0;
    }
    flush();
    nf_finish<T, OutT>();
}

// Fallback for shapes whose tile does not fit LDS (very large d or L): one output per thread,
// inputs through L1/L2.
template <class T, class TapT, class OutT, bool HILBERT>
__global__ __launch_bounds__(256) void k_fir_direct(VSrc<T> src, OutT* __restrict__ out, long n_out, int L,
                                                    int d, const TapT* __restrict__ rev, NanFix fx) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < n_out; m += (long)gridDim.x * blockDim.x) {
        using AccT = typename std::conditional<HILBERT, T, OutT>::type;
        AccT acc = zero_of<AccT>();
        const long b = m * d;
        for (int k = 0; k < L; k++) mac(acc, rev[k], src.load(b + k));
        if constexpr (HILBERT) out[m] = mkcf(src.load(b + L / 2), acc);
        else out[m] = acc;
        if (fx.rev && nf_bad(acc)) {                   // (real-valued Complex taps skip the cross terms the reference's 0 * NaN keeps)
            out[m] = nf_direct<T, OutT>(src, fx.rev, fx.L, fx.d, fx.kind, m);
        }
    }
}

// Tile choice: the cheapest shape by a small cost model of the inner loops (below) among those whose
// LDS footprint lets >= 4 workgroups share a CU; shapes that would leave CUs idle on a small input
// are penalised, so small inputs get small tiles.
template <class T, class TapT, class OutT, bool HILBERT>
static void launch_fir_any(const FirPlan& pl, const TapT* tp, const TapT* rev, VSrc<T> src, OutT* out,
                           long n_out, hipStream_t s, NanFix fx) {
    if (n_out <= 0) return;
    const int cus = device_cu_count();
    struct Cfg { int NT, R, S; };
    // {256,8,1}: d = 1..2; the S > 1 shapes are for decimating filters; {128,8,1}/{64,8,1}: small inputs
    static const Cfg cfgs[] = {{256, 8, 1}, {128, 8, 1}, {64, 8, 1}, {256, 2, 1}, {128, 2, 1},
                               {256, 4, 2}, {256, 8, 2}, {256, 8, 4}};
    constexpr int NCFG = 8;
    const int Q = pl.qpad;
    auto geom = [&](int c) {
        return fir_geom(cfgs[c].NT / cfgs[c].S, cfgs[c].R, pl.d, Q, sizeof(T), sizeof(OutT), cfgs[c].S);
    };
    auto nout = [&](int c) { return (long)cfgs[c].NT / cfgs[c].S * cfgs[c].R; };
    // clocks per output: ceil(d/S) phases of Q taps per thread, each R packed multiply-adds (4 clk per
    // wave) + one LDS read (~2.1 clk, the 4 SIMDs share the LDS port), R reads to fill the window of a
    // phase; + a fixed per-tile cost (barriers, staging bookkeeping).
    auto cost = [&](int c) {
        const int R = cfgs[c].R, S = cfgs[c].S;
        const double ph = (pl.d + S - 1) / S;
        return (ph * (Q * R * 4.0 + (Q + R) * 8.4) + 600.0) / (double)nout(c) * (256.0 / cfgs[c].NT);
    };
    int pick = -1;
    FirGeom g{};
    const int force = pl.cfg;                                               // rr_build_opts.fir_cfg
    if (force >= 0 && force < NCFG && !(HILBERT && cfgs[force].S > 1)) {
        const FirGeom gc = geom(force);
        if (gc.lds_bytes <= 64 * 1024) { pick = force; g = gc; }
    }
    if (pick < 0) {
        double best = 0;
        for (int c = 0; c < NCFG; c++) {
            if (HILBERT && cfgs[c].S > 1) continue;
            const FirGeom gc = geom(c);
            if (gc.lds_bytes > 40 * 1024) continue;
            const long tiles = (n_out + nout(c) - 1) / nout(c);
            double cst = cost(c);
            if (tiles < 2L * cus) cst *= (2.0 * cus) / (double)tiles;       // does not fill the chip
            const long staged_c = ((long)gc.np * pl.d + cfgs[c].NT - 1) / cfgs[c].NT;
            if (staged_c > FIR_MAXPRE) cst *= 2.0;                          // staging not pipelined
            // the R = 8, S = 1 builds with the long register pipeline spill ~50 B for Complex data x Complex taps, and
            // every scratch access waits for the prefetched tile (profiles/TUNING_LOG.md): prefer a split shape there (measured
            // at 127 taps /2: 0.78 vs 0.81 ms; with real taps the 12 B spill still wins, 0.46 vs 0.56 ms)
            if (cfgs[c].R == 8 && cfgs[c].S == 1 && staged_c > 10 && sizeof(T) == 8 && sizeof(TapT) == 8) cst *= 1.5;
            if (pick < 0 || cst < best) { pick = c; g = gc; best = cst; }
        }
    }
    if (pick < 0) {
        long grid = (n_out + 255) / 256;
        if (grid > (long)cus * 16) grid = (long)cus * 16;
        hipLaunchKernelGGL((k_fir_direct<T, TapT, OutT, HILBERT>), dim3((unsigned)grid), dim3(256), 0, s, src, out,
                           n_out, pl.L, pl.d, rev, fx);
        RR_HIP(hipGetLastError());
        return;
    }
    const int NT = cfgs[pick].NT;
    const long ntiles = (n_out + nout(pick) - 1) / nout(pick);
    long per_cu = (long)(160 * 1024) / (long)(g.lds_bytes ? g.lds_bytes : 1);
    const long by_waves = 32 / (NT / 64);
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
#ifdef RR_MEASURE_KNOBS
    if (const char* pe = getenv("RR_FIR_PERCU")) per_cu = atoi(pe) > 0 ? atoi(pe) : per_cu;   // measurement builds only
#endif
    long grid = ntiles < (long)cus * per_cu ? ntiles : (long)cus * per_cu;
#ifdef RR_MEASURE_KNOBS
    const char* qe = getenv("RR_FIR_QCOMPUTE");                             // measurement builds only: taps actually multiplied
    const int qcomp = qe ? atoi(qe) : -1;
#else
    const int qcomp = -1;
#endif
    const int qrun = qcomp >= 0 && qcomp < pl.qpad ? qcomp : pl.qpad;
    // the R = 8, S = 1 shapes (d = 1) stage <= 10 values per thread: a build with the shorter register
    // pipeline leaves room for the held outputs without spilling
    const long staged = ((long)g.np * pl.d + NT - 1) / NT;
    // d = 1, R = 8, S = 1 and up to 512 taps per phase: fixed row stride NT + 66 (== 2 mod 4: bank-friendly)
    const bool fixed_rs = pick <= 2 && pl.d == 1 && staged <= 10 && pl.qpad / 8 + 1 <= 66;
    if (fixed_rs) {
        g.rstride = NT + 66;
        g.pstride = 8 * g.rstride + 1;
        const size_t in_b = (size_t)g.pstride * sizeof(T), out_b = (size_t)NT * 9 * sizeof(OutT);
        g.lds_bytes = in_b > out_b ? in_b : out_b;
    }
#define RR_FIR_LAUNCH(NTV, RV, SV, PREV)                                                                                \
    hipLaunchKernelGGL((k_fir<T, TapT, OutT, NTV, RV, SV, PREV, 0, HILBERT>), dim3((unsigned)grid), dim3(NTV), g.lds_bytes, \
                       s, nanfix_ctx(fx, src, out, (long)(NTV / SV * RV), 1, n_out, ntiles, 1), src, out, n_out, pl.L, pl.d, qrun, g.np, g.rstride, g.pstride, tp)
#define RR_FIR_LAUNCH_RSC(NTV, RV, PREV)                                                                                \
    hipLaunchKernelGGL((k_fir<T, TapT, OutT, NTV, RV, 1, PREV, NTV + 66, HILBERT>), dim3((unsigned)grid), dim3(NTV),     \
                       g.lds_bytes, s, nanfix_ctx(fx, src, out, (long)(NTV * RV), 1, n_out, ntiles, 1), src, out, n_out, pl.L, pl.d, qrun, g.np, g.rstride, g.pstride, tp)
#define RR_FIR_LAUNCH2(NTV, RV, SV)                                                   \
    do {                                                                              \
        if (fixed_rs) RR_FIR_LAUNCH_RSC(NTV, RV, 10);                                 \
        else if (staged <= 10) RR_FIR_LAUNCH(NTV, RV, SV, 10);                        \
        else RR_FIR_LAUNCH(NTV, RV, SV, FIR_MAXPRE);                                  \
    } while (0)
    switch (pick) {
    case 0: RR_FIR_LAUNCH2(256, 8, 1); break;
    case 1: RR_FIR_LAUNCH2(128, 8, 1); break;
    case 2: RR_FIR_LAUNCH2(64, 8, 1); break;
    case 3: RR_FIR_LAUNCH(256, 2, 1, FIR_MAXPRE); break;
    case 4: RR_FIR_LAUNCH(128, 2, 1, FIR_MAXPRE); break;
    default:
        if constexpr (!HILBERT) {
            switch (pick) {
            case 5: RR_FIR_LAUNCH(256, 4, 2, FIR_MAXPRE); break;
            case 6: RR_FIR_LAUNCH(256, 8, 2, FIR_MAXPRE); break;
            default:
                if (staged <= 18) RR_FIR_LAUNCH(256, 8, 4, 18); else RR_FIR_LAUNCH(256, 8, 4, FIR_MAXPRE);
                break;
            }
        }
        break;
    }
#undef RR_FIR_LAUNCH2
#undef RR_FIR_LAUNCH_RSC
#undef RR_FIR_LAUNCH
    RR_HIP(hipGetLastError());
}

// Does any tile shape of the direct form fit the LDS budget for this plan?  When none does, launch_fir_any falls back to
// k_fir_direct — one thread per output walking the taps in global memory, 10-30x slower (127 taps /20: 9.6 ms per 1e8 samples
// against 0.33 on overlap-save tiles with a decimating store) — so the block's path selection asks before it relies on it.
bool fir_direct_has_tile(const FirPlan& pl, size_t es_in, size_t es_out) {
    static const int shapes[8][3] = {{256, 8, 1}, {128, 8, 1}, {64, 8, 1}, {256, 2, 1}, {128, 2, 1}, {256, 4, 2}, {256, 8, 2}, {256, 8, 4}};
    for (auto& c : shapes)
        if (fir_geom(c[0] / c[2], c[1], pl.d, pl.qpad, es_in, es_out, c[2]).lds_bytes <= 40 * 1024) return true;
    return false;
}

void launch_fir_c32(const FirPlan& pl, const void* tp, const void* rev, VSrc<cf> src, cf* out, long n_out,
                    hipStream_t s, NanFix fx) {
    if (pl.complex_taps)
        launch_fir_any<cf, cf, cf, false>(pl, (const cf*)tp, (const cf*)rev, src, out, n_out, s, fx);
    else
        launch_fir_any<cf, float, cf, false>(pl, (const float*)tp, (const float*)rev, src, out, n_out, s, fx);
}
void launch_fir_f32(const FirPlan& pl, const float* tp, const float* rev, VSrc<float> src, float* out,
                    long n_out, hipStream_t s, NanFix fx) {
    launch_fir_any<float, float, float, false>(pl, tp, rev, src, out, n_out, s, fx);
}
void launch_fir_f32c(const FirPlan& pl, const cf* tp, const cf* rev, VSrc<float> src, cf* out, long n_out,
                     hipStream_t s, NanFix fx) {
    launch_fir_any<float, cf, cf, false>(pl, tp, rev, src, out, n_out, s, fx);
}
void launch_hilbert(const FirPlan& pl, const float* tp, const float* rev, VSrc<float> src, cf* out,
                    long n_out, hipStream_t s, NanFix fx) {
    launch_fir_any<float, float, cf, true>(pl, tp, rev, src, out, n_out, s, fx);
}

// ---- Hilbert with the zero taps skipped -------------------------------------------------------------
// hilbert() (src/fir.rs:660-680) is non-zero only at odd distances from the centre tap, i.e. the
// reversed taps rev[j] vanish unless j = par (mod 2), par = (L/2 + 1) % 2.  With h[q] = rev[2q + par]
// and the input read as PAIRS  P[n] = (xp[2n + par], xp[2n + par + 1]):
//     (Im a[2m], Im a[2m+1]) = sum_q h[q] * P[m + q]
// one d = 1 FIR of half the length whose "samples" are pairs of consecutive floats and whose taps are
// real: exactly the Complex-sample/real-tap FIR above (one v_pk_fma_f32 per tap and output pair, one
// ds_read_b64 per tap), 32 instead of 72 multiply-adds per output for the 65-tap transformer.  Each
// thread produces 8 consecutive output pairs.  Re a[k] = xp[k + L/2] (src/hilbert.rs:115) comes from
// the same tile: L/2 - par is odd, so Re a[2m] is the .y of pair m + (L/2-par-1)/2 and Re a[2m+1] the
// .x of the pair after it.
constexpr int HIL_PRE = 10;
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float* hil_sptr;      // constant address space: s_load_dword, a scalar register
__device__ __forceinline__ hil_sptr hil_scalar(const float* p) { return (hil_sptr)p; }
#else
inline const float* hil_scalar(const float* p) { return p; }
#endif
// acc |= lanes whose x is NaN or +-Inf: v_cmp_class_f32 into VCC and s_or_b64 at once (class bits: signalling NaN, quiet
// NaN, -Inf, +Inf) — written as one asm block so that no compare's mask waits in a scalar register pair of its own: sixteen
// of them in the unrolled store loop exhausted the kernel's scalar registers, and the overflow went into vector registers.
// acc is a WAVE mask: only ever updated under wave-uniform control flow.
__device__ __forceinline__ void hil_acc(unsigned long long& acc, float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_cmp_class_f32 vcc, %1, %2\n\ts_or_b64 %0, %0, vcc" : "+s"(acc) : "v"(x), "s"(0x207) : "vcc", "scc");   // (s_or_b64 writes SCC)
#else
    acc |= !(x - x == 0.0f);
#endif
}
// A pair = two consecutive input floats, the first of any parity, read with ONE 8-byte load (as a cf)
// although it is only 4-byte aligned: gfx9+ global memory instructions take dword-aligned addresses
// (compute queues run in unaligned access mode); with the honest alignment the compiler splits every
// load in two.  Covered by the Hilbert tests with even and odd L/2 and odd window offsets.
template <int NT>
__global__ __launch_bounds__(NT) void k_hilbert(VSrc<float> src, cf* __restrict__ out, long n_out, int L, int par,
                                                   int Q, int np, int rstride, const float* __restrict__ hq, int* __restrict__ wgflags) {
    constexpr int R = 8, NP = NT * R;                   // pairs per tile (2*NP outputs)
    // Non-finite samples (nan_fix.hpp), round 5.  The taps this kernel skips are zero and the reference multiplies them all
    // the same — 0 * NaN = NaN — so a bad sample reaches every output of its window there and only every other one here.
    // The tile's INPUT is tested where that costs no vector register (the kernel sits at 126 of the 128 that four waves per
    // SIMD allow; a packed accumulator in the staging code, a branch there, or nan_fix.hpp's in-kernel hook with its LDS
    // flags each tipped it to three waves: 0.23 -> 0.40 ms per 1e8 samples): the real part of every output IS an input
    // sample and is tested as it is stored (hil_acc: a compare into VCC and a scalar OR), the few dozen samples around them
    // by one wave from LDS, the zero-tap position in front of the first pair through a scalar load.  The verdict leaves the
    // kernel as one word per workgroup (wgflags), and a second, tiny launch (k_hilbert_repair, same grid) recomputes ALL
    // outputs of the tiles of a flagged workgroup with the reference's own fold — exactly the reference's set — and exits at
    // once where the flag is clear.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* lds = reinterpret_cast<cf*>(smem_raw);          // pair n at (n % R)*rstride + n / R
    cf* lds_o = reinterpret_cast<cf*>(smem_raw);
    const int t = threadIdx.x;
    const long ntiles = (n_out + 2 * NP - 1) / (2 * NP);
    const int cnt_k = (np + NT - 1) / NT;               // <= HIL_PRE (host checks)

    unsigned long long bad_any = 0;                     // lanes that met a non-finite input sample in any tile (a wave mask: scalar registers)
    cf pre[HIL_PRE];
    float pre_edge = 0.0f;                              // xp[tile start] when par == 1: a zero-tap position of the tile's FIRST
                                                        // output that the pairs do not cover (the reference multiplies it: 0 * NaN)
    auto interior = [&](long tile) {
        const long g0 = tile * 2 * NP + par;
        return g0 - par >= src.plen && g0 - src.plen + 2L * np <= src.in_len;     // (- par: pre_edge)
    };
    auto fetch = [&](long tile) {                       // (see k_fir for the two asm barriers)
        if (!interior(tile)) return;
        const cf* gp = reinterpret_cast<const cf*>(src.in + (tile * 2 * NP + par - src.plen));
        // (a scalar load into a scalar register — the index is workgroup-uniform; interior: g0 - 1 >= plen)
        if (par) pre_edge = *hil_scalar(src.in + (tile * 2 * NP - src.plen));
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
        int cnt = cnt_k;
        asm volatile("" : "+s"(cnt));
#pragma unroll
        for (int c = 0; c < HIL_PRE; c++) {
            if (c < cnt - 1) {
                pre[c] = (gp + c * NT)[tt];
            } else if (c == cnt - 1) {
                const unsigned i = tt + c * NT;
                pre[c] = gp[i < (unsigned)np ? i : (unsigned)np - 1];
            }
        }
    };
    auto commit = [&]() {
        unsigned tt = t;
        asm volatile("" : "+v"(tt));
        int cnt = cnt_k;
        asm volatile("" : "+s"(cnt));
        cf* slot = lds + (tt % R) * rstride + tt / R;   // R divides NT: round c lands NT/R columns further
        hil_acc(bad_any, pre_edge);
#pragma unroll
        for (int c = 0; c < HIL_PRE; c++) {
            if (c < cnt - 1) slot[c * (NT / R)] = pre[c];
            else if (c == cnt - 1 && tt + c * NT < (unsigned)np) slot[c * (NT / R)] = pre[c];
        }
    };
    auto stage_direct = [&](long tile) {                // tiles touching the carried history / window end
        const long g0 = tile * 2 * NP + par;
        if (par) hil_acc(bad_any, src.load(g0 - 1));
        for (int i = t; i < np; i += NT)
            lds[(i % R) * rstride + i / R] = mkcf(src.load(g0 + 2L * i), src.load(g0 + 2L * i + 1));
    };

    long tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const long m0 = tile * 2 * NP;
        __syncthreads();
        if (interior(tile)) commit(); else stage_direct(tile);
        __syncthreads();
        // The input samples of this tile that do NOT come by as the real part of one of its outputs (tested where they are
        // stored, below): the pairs in front of the first real part and behind the last one — a few dozen, one wave, here,
        // where next to nothing is live in registers.  (Scalar branch, uniform trip counts: bad_any stays a wave mask.)
        if (__builtin_amdgcn_readfirstlane(t >> 6) == 0) {
            const int hh = (L / 2 - par - 1) / 2;        // real parts come from the pairs [hh, NP + hh]
            for (int b = 0; b <= hh; b += 64) {
                const int n = b + t < hh ? b + t : hh;   // (clamped lanes re-test pair hh)
                const cf v = lds[(n % R) * rstride + n / R];
                hil_acc(bad_any, v.x); hil_acc(bad_any, v.y);
            }
            for (int b = NP + hh; b < np; b += 64) {
                const int n = b + t < np ? b + t : np - 1;
                const cf v = lds[(n % R) * rstride + n / R];
                hil_acc(bad_any, v.x); hil_acc(bad_any, v.y);
            }
        }
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);

        cf acc[R], w[R];
        const cf* lp = lds + t;
#pragma unroll
        for (int j = 0; j < R; j++) { acc[j] = mkcf(0.0f, 0.0f); w[j] = lp[j * rstride]; }
#pragma unroll 1
        for (int q0 = 0; q0 < Q; q0 += 8) {
            const cf* lq = lp + q0 / R + 1;
            float tap8[8];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) tap8[kk] = hq[q0 + kk];
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
#pragma unroll
                for (int j = 0; j < R; j++) mac(acc[j], tap8[kk], w[(kk + j) % R]);
                w[kk] = lq[kk * rstride];
            }
        }
        cf res[2 * R];
        const int h0 = (L / 2 - par - 1) / 2;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int n0 = t * R + j + h0, n1 = n0 + 1;
            res[2 * j] = mkcf(lds[(n0 % R) * rstride + n0 / R].y, acc[j].x);
            res[2 * j + 1] = mkcf(lds[(n1 % R) * rstride + n1 / R].x, acc[j].y);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 2 * R; c++) lds_o[t * (2 * R + 1) + c] = res[c];
        __syncthreads();
        cf vo[2 * R];
#pragma unroll
        for (int c = 0; c < 2 * R; c++) {
            const int i = c * NT + t;
            vo[c] = lds_o[(i / (2 * R)) * (2 * R + 1) + i % (2 * R)];
        }
#pragma unroll
        for (int c = 0; c < 2 * R; c++) {
            hil_acc(bad_any, vo[c].x);                   // the real part IS an input sample (hilbert.rs:115)
            if (m0 + c * NT + t < n_out) out[m0 + c * NT + t] = vo[c];
        }
    }
    if (wgflags && bad_any != 0 && (threadIdx.x & 63) == 0) wgflags[blockIdx.x] = 1;
}
// (force: outputs k_hilbert left finite are the reference's NaN too — 0 * NaN on the transformer's zero taps)
__global__ __launch_bounds__(256) void k_hilbert_repair(NanFixCtx nfx, int* __restrict__ wgflags) {
    (void)nfx;
    if (__hip_atomic_load(wgflags + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // (workgroup-uniform)
#if defined(__HIP_DEVICE_COMPILE__)
    nf_repair<float, cf>((nf_ctx_ptr)__builtin_amdgcn_kernarg_segment_ptr(), true);
#endif
    __syncthreads();
    if (threadIdx.x == 0) wgflags[blockIdx.x] = 0;
}

// hq = device [Q] (Q = taps per phase padded to a multiple of 8); returns false when the shape is
// not covered (caller falls back to the generic FIR kernel).
bool launch_hilbert_skip(int L, int par, int Q, const float* hq, VSrc<float> src, cf* out, long n_out, hipStream_t s, NanFix fx) {
    if (n_out <= 0) return true;
    constexpr int NT = 256, R = 8;
    const int np = NT * R + Q + 8;                      // pairs a tile may touch (taps, window refill, Re parts)
    if ((np + NT - 1) / NT > HIL_PRE || (L / 2 - par + 1) / 2 + 1 > Q + 8 || par != (L / 2 + 1) % 2) return false;
    int rs = np / R + 1;
    while (rs % 4 != 2) rs++;
    const size_t in_b = (size_t)R * rs * sizeof(cf), out_b = (size_t)NT * (2 * R + 1) * sizeof(cf);
    const size_t smem = in_b > out_b ? in_b : out_b;
    const long ntiles = (n_out + (long)NT * 2 * R - 1) / ((long)NT * 2 * R);
    long per_cu = (long)(160 * 1024) / (long)smem;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const long cap = std::min<long>((long)device_cu_count() * per_cu, HILBERT_MAX_GRID);
    const long grid = ntiles < cap ? ntiles : cap;
    int* flags = fx.rev ? fx.wgflags : nullptr;
    hipLaunchKernelGGL((k_hilbert<NT>), dim3((unsigned)grid), dim3(NT), smem, s, src, out, n_out, L, par, Q, np, rs, hq, flags);
    RR_HIP(hipGetLastError());
    if (flags) {
        // (nan_fix.hpp: tile k = blockIdx.x + j gridDim.x owns the outputs [k 2 NT R, (k + 1) 2 NT R); same grid as above)
        const NanFixCtx nfx = nanfix_ctx(fx, src, out, (long)NT * 2 * R, 1, n_out, ntiles, 1);
        hipLaunchKernelGGL(k_hilbert_repair, dim3((unsigned)grid), dim3(256), 0, s, nfx, flags);
        RR_HIP(hipGetLastError());
    }
    return true;
}

// ---- frequency-translation rotator (src/fir.rs:464-473) ---------------------------------
// RR_ROT_MODEL: phi_m = phase0 * step^m with the f32-rounded phase0/step of the reference,
// evaluated in f64 by binary powering per element (log2(m) complex multiplies).
__device__ __forceinline__ void zmul(double& ax, double& ay, double bx, double by) {
    const double x = ax * bx - ay * by, y = ax * by + ay * bx;
    ax = x; ay = y;
}
// One binary powering per THREAD (step^(m0 + first index), ~2 log2(m) f64 complex multiplies), then one multiply by
// step^(grid stride) per element (gx, gy: formed on the host in f64 the same way).  The per-element powering
// this replaces cost 0.17 ms on the 1.25e7 outputs of the /8 channelizer.
__global__ __launch_bounds__(256) void k_rotate_model(cf* __restrict__ y, long n, double p0x, double p0y,
                                                      double sx, double sy, long m0, double gx, double gy) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long e = (unsigned long)(m0 + i);
    double rx = p0x, ry = p0y, bx = sx, by = sy;
    while (e) {
        if (e & 1) zmul(rx, ry, bx, by);
        zmul(bx, by, bx, by);
        e >>= 1;
    }
    for (; i < n; i += (long)gridDim.x * blockDim.x) {
        const cf v = y[i];
        const double ox = (double)v.x * rx - (double)v.y * ry;
        const double oy = (double)v.x * ry + (double)v.y * rx;
        y[i] = mkcf((float)ox, (float)oy);
        zmul(rx, ry, gx, gy);
    }
}
// y[i] *= ring[(pos0 + i) & mask]: the phases the replay kernel left in its ring (mask = capacity - 1, a power of two)
__global__ __launch_bounds__(256) void k_rotate_table(cf* __restrict__ y, long n, const cf* __restrict__ tab, long pos0, long mask) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const cf v = y[i], p = tab[(pos0 + i) & mask];
        // sample * phase, un-contracted num-complex order (fir.rs:469)
        y[i] = mkcf(sub_rn(mul_rn(v.x, p.x), mul_rn(v.y, p.y)),
                  add_rn(mul_rn(v.x, p.y), mul_rn(v.y, p.x)));
    }
}
static unsigned rot_grid(long n) {
    long g = (n + 255) / 256;
    const long cap = (long)device_cu_count() * 8;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}
void launch_rotate_model(cf* y, long n, double p0x, double p0y, double sx, double sy, long m0, hipStream_t s) {
    if (n <= 0) return;
    const unsigned grid = rot_grid(n);
    double gx = 1.0, gy = 0.0, bx = sx, by = sy;                     // step^(grid * 256)
    for (unsigned long e = (unsigned long)grid * 256; e; e >>= 1) {
        if (e & 1) { const double x = gx * bx - gy * by, yy = gx * by + gy * bx; gx = x; gy = yy; }
        const double x = bx * bx - by * by, yy = 2.0 * bx * by; bx = x; by = yy;
    }
    hipLaunchKernelGGL(k_rotate_model, dim3(grid), dim3(256), 0, s, y, n, p0x, p0y, sx, sy, m0, gx, gy);
    RR_HIP(hipGetLastError());
}
// The reference's rotator (fir.rs:464-473) is a sequential f32 recurrence, phase <- phase * step once per output and
// never renormalised; replaying it bit for bit is inherently serial (every one of the six operations of a step rounds, and
// the rounding errors of step m feed step m + 1: no closed form, no parallel prefix).  ONE LANE walks the chain from the
// phase carried in *state (device memory: a chain of device-resident blocks never synchronises with the host), skips
// `nskip` steps without storing and leaves the next n phases in the ring tab[(pos0 + i) & mask] for k_rotate_table.
// A step is three packed instructions — (x sx, x sy), (y sy, y sx), and their sum with the low half negated: exactly the
// four rounded products and the two rounded sums of num-complex's `phase * step`, a - b being a + (-b) bit for bit — with
// the step in a scalar register pair; the stores are fire-and-forget.  The chain is data-independent, so the block runs it
// AHEAD of the filter on a side stream (blocks.cpp FirC32::rotate_output).
__device__ __forceinline__ creg rotor_step(creg z, creg st) {
#if defined(__HIP_DEVICE_COMPILE__)
    // ONE asm statement: written as three, the compiler guards each packed instruction that reads its predecessor's result
    // with an s_nop (23 per 16 steps), and a lone wave pays an issue slot for every one of them: 14 -> 11 ns per output
    // (tools/micro/rotor_rate.hip; bit-identical — test_translate_default_mode_is_the_reference_recurrence_for_1e7_outputs)
    creg p, q, r;
    asm("v_pk_mul_f32 %1, %3, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                              // (x sx, x sy)
        "v_pk_mul_f32 %2, %3, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"                              // (y sy, y sx)
        "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]"                                                    // (x sx - y sy, x sy + y sx)
        : "=v"(r), "=&v"(p), "=&v"(q) : "v"(z), "s"(st));
    return r;
#else
    return mk(z.x * st.x - z.y * st.y, z.x * st.y + z.y * st.x);
#endif
}
// A lone wave gets one issue slot every 4 clocks whatever the instruction (the CU's arbiter visits each SIMD in turn), so
// the cost of a step is its INSTRUCTION COUNT x 4 clocks, not its latency: the first version of this loop — three packed
// operations, one 8-byte store and seven scalar instructions of ring-index arithmetic per phase — ran at 24.5 ns per
// output (tools/replay_rate.py).  Here a block of 16 phases is 48 packed operations, 8 sixteen-byte stores at immediate
// offsets from one scalar pointer and 5 scalar instructions: 3.8 instructions per phase.  The ring wrap is taken out of
// the loop (two contiguous segments).
__device__ __forceinline__ void rotor_run(creg& z, const creg st, creg* __restrict__ p, long n) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef f4 f4u __attribute__((aligned(8)));
    long i = 0;
    // A block's phases are stored one block LATE, a pair after every second step of the next block: a store that follows the
    // step that produced its data waits for that step like the next step does and costs the lone wave an issue slot on the
    // chain's critical path; a store of finished data issues in the gap in which the next instruction of the chain waits for
    // its operand anyway (11.3 -> 9.x ns per output; tools/replay_rate.py).
    auto store_pair = [&](creg* q, const creg* o, int k) {
        f4 w; w.x = o[k].x; w.y = o[k].y; w.z = o[k + 1].x; w.w = o[k + 1].y;
        *reinterpret_cast<f4u*>(q + k) = w;
    };
    auto fill = [&](creg* o, creg* q, const creg* prev) {      // 16 phases into o; prev (if any) leaves for q as it goes
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            o[k] = z; z = rotor_step(z, st);
            o[k + 1] = z; z = rotor_step(z, st);
            if (prev) store_pair(q, prev, k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto flush = [&](creg* q, const creg* o) {
#pragma unroll
        for (int k = 0; k < 16; k += 2) store_pair(q, o, k);
    };
    if (n >= 16) {
        creg a[16], b[16];                                     // (two register blocks in turn: no copies)
        fill(a, nullptr, nullptr); i = 16;
        for (; i + 32 <= n; i += 32) {
            fill(b, p + i - 16, a);
            fill(a, p + i, b);
        }
        if (i + 16 <= n) { fill(b, p + i - 16, a); flush(p + i, b); i += 16; }
        else flush(p + i - 16, a);
        p += i;
    }
    for (; i < n; i++, p++) { *p = z; z = rotor_step(z, st); }
}
__global__ void k_rotor_replay(cf* __restrict__ state, float stx, float sty, cf* __restrict__ tab, long pos0, long mask,
                               long nskip, long n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const creg st = mk(stx, sty);
    creg z = to_reg(*state);
    for (long i = 0; i < nskip; i++) z = rotor_step(z, st);
    creg* ring = reinterpret_cast<creg*>(tab);
    const long start = pos0 & mask, cap = mask + 1;
    const long n0 = n < cap - start ? n : cap - start;              // up to the end of the ring, then from its start
    rotor_run(z, st, ring + start, n0);
    rotor_run(z, st, ring, n - n0);
    *state = from_reg(z);
}
void launch_rotor_replay(cf* state, float stx, float sty, cf* ring, long pos0, long mask, long nskip, long n, hipStream_t s) {
    if (n <= 0 && nskip <= 0) return;
    hipLaunchKernelGGL(k_rotor_replay, dim3(1), dim3(64), 0, s, state, stx, sty, ring, pos0, mask, nskip, n);
    RR_HIP(hipGetLastError());
}
void launch_rotate_table(cf* y, long n, const cf* ring, long pos0, long mask, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rotate_table, dim3(rot_grid(n)), dim3(256), 0, s, y, n, ring, pos0, mask);
    RR_HIP(hipGetLastError());
}

}  // namespace rr
