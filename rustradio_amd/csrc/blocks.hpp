// blocks.hpp — host-side block objects behind the C ABI (include/rustradio_amd.h).
#pragma once
#include <complex>
#include <memory>
#include <vector>

#include "../../include/rustradio_amd.h"
#include "kernels.hpp"
#include "stage.hpp"

namespace rr {

void set_thread_device(int d);
int thread_device();
// page-locked host ranges the library was told about (rr_host_register): their device view, for zero-copy host windows
void host_range_add(void* base, size_t bytes);
void host_range_remove(void* base);
bool host_range_registered(const void* host, size_t bytes);     // inside a live rr_host_register'd range (admitted to zero-copy or not)
void* device_view_of_host(const void* host, size_t bytes);      // nullptr: not wholly inside an rr_host_register'd range (staged copies)

struct Block {
    const char* name;
    size_t in_es, out_es;
    int device = 0;
    hipStream_t stream = nullptr;         // private stream: host-window work() and setup copies
    hipStream_t last_stream = nullptr;    // stream of the most recent work call (what sync() waits for)
    DevBuf<unsigned char> st_in, st_out;   // staging for host-window work() (pageable windows)
    std::unique_ptr<HostStage> hstage_;    // ... and the pinned chunks their bytes cross the bus in (stage.hpp), made on first use
    HostStage& hstage() { if (!hstage_) hstage_.reset(new HostStage()); return *hstage_; }
    // page-locked host INPUT windows are read by the kernels in place (zero copy); blocks whose kernels read the window more
    // than once (the N-channel blocks: every run of channel rounds re-reads its tiles) keep the upload
    bool zero_copy_in = true;
    // work_host on a page-locked OUTPUT window: set around work_dev, and the hook below runs once the call's work has
    // completed (the host then reads its own window for free where a kernel would cross PCIe again — FftFilter)
    const void* host_out = nullptr;
    virtual void host_out_done(const void* /*out_host*/, size_t /*produced*/) {}
    virtual void host_out_reset() {}      // drop a pass deferred by an earlier host-window call that never reached host_out_done

    Block(const char* nm, size_t ies, size_t oes);
    virtual ~Block();
    Block(const Block&) = delete;
    Block& operator=(const Block&) = delete;

    virtual int work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                         size_t* produced, size_t* need, hipStream_t s) = 0;
    virtual bool eof(bool src_eof);
    virtual int work_host(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                          size_t* produced, size_t* need);
    virtual size_t out_windows() const { return 1; }   // output streams (windows of out_cap elements each)
    void sync();

    // optional HIP-event timing of the block's dominant kernel (bench.py's roofline figure)
    bool prof_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_evs;
    size_t prof_used = 0;
    void prof_begin(hipStream_t s);
    void prof_end(hipStream_t s);
    void prof_read(double* total_ms, size_t* launches, bool reset);
};

// Forward, unnormalised FFT of consecutive N-sample frames in natural bin order for ANY N >= 2 (rustfft plans any size:
// fft_stream.rs:43-44, fft.rs:27-32).  One tile kernel up to 16384 = 2^14 points and for every size up to 2048; beyond
// that the four-step decomposition (powers of two) or Bluestein's chirp-z on a power of two M >= 2 N - 1 (kernels_misc.hip).
struct AnyFft {
    size_t N = 0;
    int log2n = 0;                    // one tile: power of two <= 16384 (k_fft_frames / k_fft_small)
    DevBuf<cf> d_tw, d_tw4096;
    int log2m = 0;                    // one tile: any size <= 2048 (fused k_fft_bluestein on a 2^log2m-point filter tile)
    DevBuf<cf> d_bh, d_chirp;
    size_t N1 = 0, N2 = 0;            // four-step: N = N1 N2, both powers of two
    std::unique_ptr<AnyFft> f1, f2;
    DevBuf<cf> d_twN;                 // w_N^m, m < N
    size_t M = 0;                     // generic Bluestein on M points
    std::unique_ptr<AnyFft> fm;
    DevBuf<cf> d_b;                   // FFT_M(conj chirp, wrapped) / M in natural order
    DevBuf<cf> t1, t2;                // work frames (grow on demand)
    AnyFft(size_t n, hipStream_t s);
    void forward(const cf* in, cf* out, long nframes, hipStream_t s);
};

struct FftFilter;
// Tables of k_fftfilt_prune (decimation by the last radix of the tile plan: 4 / 8 / 16 on 1024 / 2048 / 4096 points).
struct PruneTables {
    int log2f = 0;
    DevBuf<cf> d_tw, d_h2, d_h2b, d_twb;
    // t = taps in caller order (y[n] = sum_k t[k] x[n - k]); split: real-stream kernel, tables for Re t and Im t.
    // false when (L, d) is not covered.
    bool build(const std::vector<std::complex<double>>& t, size_t d, bool split, hipStream_t s);
};
// Tables of the decimate-first (polyphase) fused chains (kernels_poly.hip): per channel c and phase p the response of
// the taps t[D j + p] on 1024-point tiles, in the register order of the kernels.
struct PolyTables {
    DevBuf<cf> d_tw, d_h;
    // taps = [C][L] caller-order taps; false when (D, L) is not covered
    bool build(const rr_c32* taps, size_t C, size_t L, size_t D, bool multi, hipStream_t s);
};
struct FirC32 : Block {
    FirPlan pl;
    // Non-decimating filters beyond a few taps run as overlap-save FFT tiles (the FftFilter kernel on a window
    // with no history: y[m] = sum_k rev[k] x[m + k] is that kernel's output for an empty prefix): the direct form
    // costs 2-4 multiply-adds per tap and sample, the tiles a constant ~0.3 ms per 1e8 samples.  rr_build_opts.fir_path
    // forces the direct kernel or the tiles for any length (read at construction).
    std::unique_ptr<FftFilter> fftk;
    std::unique_ptr<PruneTables> prune;           // deci 4 / 8 / 16: pruned inverse transform (k_fftfilt_prune)
    size_t prune_D = 0, prune_sub = 1;            // round 4: deci = prune_D * prune_sub (every prune_sub-th kept sample stored)
    // other even decimations, <= 600 taps: 2048-point tiles with the half-size inverse (k_fftfilt_half)
    bool half_ok = false;
    // decimations 3 / 5 / 6 / 7 ...: decimate-first tiles (k_fm_chain_poly with the samples stored, kernels_poly.hip)
    std::unique_ptr<PolyTables> poly;
    bool window_aware = true;                     // per-call choice by window size (off when a path is forced: tests, probes)
    DevBuf<cf> d_htw, d_htw_half, d_hhpos;
    DevBuf<unsigned char> d_tp, d_rev;
    DevBuf<cf> d_fix;                             // the taps reversed, Complex: what the non-finite repair folds with (nan_fix.hpp)
    NanFix nanfix() const { NanFix f; f.rev = d_fix.p; f.L = pl.L; f.d = pl.d; f.kind = NANFIX_CC; return f; }
    bool rot_on = false;
    int rot_mode = RR_ROT_REPLAY;                 // the reference's own recurrence is the default (fir.rs:464-473)
    float ph0x = 1, ph0y = 0, stx = 1, sty = 0;   // f32-rounded phase0 / step (fir.rs:453-461)
    size_t n_rot = 0;                             // outputs rotated so far
    // RR_ROT_REPLAY: the phase chain is data-independent, so it is generated AHEAD of the filter — one lane on a side
    // stream walks the f32 recurrence into a ring of phases while the filter kernels of the current window run; a call
    // only waits for the part of its range the chain has not reached yet.
    DevBuf<cf> d_phase;                           // the carried f32 phase (the next one to generate), on the device
    DevBuf<cf> d_ring;                            // ring of generated phases: entry i at slot i & (ring_cap - 1)
    size_t ring_cap = 0;                          // power of two
    size_t rot_gen = 0;                           // phases generated (enqueued) so far
    hipStream_t rot_stream = nullptr;
    hipEvent_t ev_gen = nullptr, ev_used = nullptr;
    bool used_pending = false;
    hipStream_t used_stream = nullptr;            // stream of the last ev_used record
    void rotor_generate(size_t upto);             // enqueue the chain up to phase index `upto` (exclusive) on rot_stream
    // RR_ROT_REPLAY (default), round 5: the chain starts on the device (one lane, no host thread, no PCIe traffic) and only a
    // block whose calls keep ARRIVING BEFORE the chain's look-ahead has finished (`starving` calls in a row: the rotator, not
    // the filter, paces the block — back-to-back batch calls, never a graph paced by a 12.5 M outputs/s source) gets a host
    // generator: a HOST thread of this block (rotor_host.cpp) walks the same chain at 2.6 ns per output from the device
    // chain's current phase into a pinned ring, copied ahead into d_ring on rot_stream.  RR_ROT_REPLAY_DEVICE never
    // switches; RR_ROT_REPLAY_HOST starts on the host thread.  The same bits in every case.
    std::unique_ptr<struct HostRotor> hrot;
    size_t hrot_base = 0;                         // phase index of the host generator's phase 0
    int starving = 0;                             // consecutive calls that found the look-ahead still running
    bool lookahead_pending = false;               // ev_gen was last recorded by a look-ahead (not by a demand generation)
    void rotor_start_host();                      // spawn the host generator at the device chain's position
    size_t dphase_at = 0;                         // phase index d_phase holds (the device chain's position)
    struct CopyDone { hipEvent_t ev; size_t upto; };
    std::vector<CopyDone> copies;                 // host-ring ranges in flight to the device, oldest first
    std::vector<hipEvent_t> free_events;
    bool rotor_fetch(size_t upto, bool block);    // host-generated phases [rot_gen, upto) -> d_ring; false: not generated yet
    void rotor_reap(bool wait_oldest);
    std::vector<std::complex<float>> h_taps;      // caller-order taps after the translate pre-rotation
    // allow_fft = false: bookkeeping / direct form only (HilbertFir's inner object)
    FirC32(const rr_c32* taps, size_t ntaps, size_t deci, bool translate, float samp_rate, float freq, bool allow_fft = true);
    DevBuf<cf> dec_tmp;               // full-rate scratch of the any-size + strided-copy path (deci > 1, 15293..16383 taps)
    ~FirC32() override;
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
    void rotate_output(cf* out, size_t out_n, hipStream_t s);   // fir.rs:464-473 (no-op without translate)
};

// Graph-level fusion of Hilbert::new(src, hn, window) -> FirFilter::builder(taps).deci(d)[.translate()]
// (examples/ax25-1200-rx.rs:238-247 wiring; BASELINE configs[4]) as ONE decimating FIR: the analytic
// signal a[k] = sum_j c[j] iv[k+j], c = delta[j - hn/2] + i rev_h[j]  (hilbert.rs:113-116) followed by
// y[m] = sum_k rev[k] a[m d + k] (fir.rs:166-197) is y[m] = sum_n G[n] iv[m d + n] with the composite
// Complex taps G = rev (*) c of length ntaps + hn - 1, applied to the REAL input: 4 B in + 8/d B out
// per sample, neither the analytic stream nor a second kernel.  iv = hn zeros || input (hilbert.rs:55).
struct Hilbert;
struct HilbertFir : Block {
    std::unique_ptr<FirC32> fir;      // taps incl. translation, rotator state, decimation bookkeeping
    // Round 4: where the composite direct form loses — it costs ~0.004 ms x composite taps / deci per 1e8 samples, nothing
    // like the 0.23 + 0.3 ms of the two blocks on their own kernels — the SAME handle runs the two stages through an
    // analytic buffer in HBM: Hilbert's kernel, then the FirFilter's own path selection (tiles, decimate-first, pruned).
    std::unique_ptr<Hilbert> hil;     // tables of the Hilbert stage
    std::unique_ptr<FirC32> fir2;     // the FirFilter stage (pre-rotated taps, no rotator: `fir` rotates the output)
    DevBuf<cf> analytic;
    bool two_stage = false;
    size_t hn = 0;                    // Hilbert ntaps
    FirPlan plG;
    DevBuf<cf> d_tpG, d_revG;
    NanFix nanfixG() const { NanFix f; f.rev = d_revG.p; f.L = plG.L; f.d = plG.d; f.kind = NANFIX_FC; return f; }
    DevBuf<float> hist[2];            // the hn input samples before the window start
    int cur = 0;
    std::unique_ptr<PruneTables> prune;   // deci 4 / 8 / 16: two real segments per tile, pruned inverse (k_fftfilt_prune)
    size_t prune_D = 0, prune_sub = 1;    // deci = prune_D * prune_sub
    HilbertFir(size_t hilbert_ntaps, int window, float parm, const rr_c32* taps, size_t ntaps, size_t deci,
               bool translate, float samp_rate, float freq);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

struct FirF32 : Block {
    FirPlan pl;
    DevBuf<float> d_tp, d_rev;
    NanFix nanfix() const { NanFix f; f.rev = d_rev.p; f.L = pl.L; f.d = pl.d; f.kind = NANFIX_FF; return f; }
    std::unique_ptr<FftFilter> fftk;   // long filters: overlap-save tiles on the real stream (see FirC32::fftk)
    std::unique_ptr<PruneTables> prune; // deci 4 / 8 / 16: pruned inverse (k_fftfilt_prune, real stream x real taps)
    size_t prune_D = 0, prune_sub = 1;  // deci = prune_D * prune_sub
    bool window_aware = true;           // per-call choice by window size (off when a path is forced)
    // Beyond 3584 taps the real-stream tiles (4096 points at most) end and the direct form costs 0.0026 ms per tap and 1e8
    // samples — or has no tile at all (5000 taps: 12.9 ms, /32: 224 ms on the one-thread-per-output fallback).  Then the block
    // filters Complex(x, 0) with the FirFilter<Complex> kernels (tiles up to 16383 taps, decimate-first, pruned) in chunks
    // through two work buffers and keeps the real parts — what FftFilterFloat does for the same reason (fft_filter.rs:431-445).
    std::unique_ptr<FirC32> wide;
    DevBuf<cf> wide_in, wide_out;
    ~FirF32() override;
    FirF32(const float* taps, size_t ntaps, size_t deci);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

struct FftFilter : Block {
    size_t L = 0, fft_size = 0, nsamples = 0;
    int log2f = 10;
    DevBuf<cf> d_tw, d_hpos;
    // >= 8192-point tiles as nsub interleaved 4096-point sub-transforms (kernels_fft.hip k_fftfilt_split)
    int nsub = 0;
    DevBuf<cf> d_tw4096, d_hs, d_wk;
    // plain 4096-point tile beside the split tables, for windows of too few split tiles to fill the chip
    int alt_log2f = 0;
    DevBuf<cf> d_tw_alt, d_hpos_alt;
    // Does the plain 4096-point tile beat the split tile on a window of n_out filtered samples?  Costs in us on the 256 CUs
    // they were measured on (tools/chain_tile_probe.py, round 5: 1500 / 2467 / 3330 taps, 512 k ... 64 M samples, the FftFilter
    // block and the fused FM chain, Complex and RTL-SDR byte sources; profiles/r05_rtl_fm_tiles.txt):
    //   plain  13 + c na,  c = 0.0105 (FftFilter), 0.0112 -> 0.0127 beyond 2500 tiles (chain: the epilogue's share grows)
    //   split  max(25 + 0.022 ns, 12 + per-tile cost x ns): FftFilter 0.0207 per tile + 2.6e-6 per output sample (stores);
    //          chain 0.032 per tile, with a 200-tile tail (a window of 2-3 rounds of the 512 resident workgroups ends uneven)
    // na / ns = the tiles each form needs (3.5x more plain tiles at 2467 taps, 6.3x at 3330).  Round 4's fit stopped at 8 M
    // samples and handed the reference's own rtl_fm shape (2467 taps, 24 M samples) to the plain tile: 0.191 ms against 0.138.
    bool alt_wins(long n_out, bool chain = false) const {
        const double cu = (double)device_cu_count() / 256.0;
        const double na = (double)n_out / (double)(((size_t)1 << alt_log2f) - L + 1) / cu;
        const double ns = (double)n_out / (double)(((size_t)1 << log2f) - L + 1) / cu;
        const double plain = 13.0 + na * (chain ? (na < 2500.0 ? 0.0112 : 0.0127) : 0.0105);
        const double split = std::max(25.0 + 0.022 * ns, chain ? 12.0 + 0.032 * (ns + 200.0) : 12.0 + 0.0207 * ns + 2.6e-6 * (double)n_out / cu);
        return plain < split;
    }
    // more than 16383 taps: overlap-save frames of M = 2^m >= 2 L points through the any-size transform (AnyFft)
    NanFix nanfix;                    // set by a FirFilter that runs on these tiles (default: none — FftFilter's own reference is a transform)
    // The FftFilter / FftFilterFloat BLOCKS (not the chains and FirFilters built on this object): a non-finite input sample
    // poisons the reference's block of nsamples inputs (+ the ntaps points added to the next one, fft_filter.rs:326-347)
    // instead of the GPU's tile — one small launch behind the tile kernels of every call (kernels_misc.hip
    // k_ref_blocks_nonfinite; rr_build_opts.fft_nonfinite_tiles leaves it out)
    bool ref_blocks = false;
    bool rb_in_kernel = true;         // ... inside the tile kernel's tail where the kernel has one (round 6, nan_fix.hpp rb_finish)
    DevBuf<cf> d_rev;                 // the taps reversed (floats for a real stream)
    DevBuf<int> d_tail;               // [2]: was the last block of call seq - 1 poisoned?
    DevBuf<int> d_rb_work;            // the in-kernel form (nan_fix.hpp rb_finish): ticket, records, flagged tiles — zero between launches
    DevBuf<long> d_rb_recs;
    int seq = 1;
    long probe_stride = 0;            // of the current call: the smallest advance of the tiles that wrote it
    // a zero-copy host output window: the probe is the HOST's (its own memory, after the call's completion wait), and the pass
    // is launched only when it finds something (a probing kernel reads the window back over PCIe: +8 us per reference-sized call,
    // the host's own probe +2; the tile kernels' FIX instantiation could report for free but rounds differently in the last bit
    // than the one the other paths run, and the paths of one block are held bit-identical: tests/test_gpu_fuzz.py)
    struct { bool on = false, force0 = false; VSrc<cf> src{}; void* out = nullptr; long n_out = 0; } deferred;
    bool tail_host = false, tail_host_known = true;   // the host's copy of the carried verdict
    void host_out_done(const void* out_host, size_t produced) override;
    void host_out_reset() override { deferred.on = false; }
    void ref_blocks_on(const rr_c32* taps);
    template <class T> void ref_blocks_pass(VSrc<T> src, T* out, long n_out, hipStream_t s, bool force0 = false);
    std::unique_ptr<AnyFft> big;
    size_t bigM = 0;
    DevBuf<cf> d_hbig, bframes, bspec;
    DevBuf<cf> prefix[2];
    int cur = 0;
    size_t pend_len = 0;
    // for_chain: the object backs a fused chain kernel (k_fm_chain / k_fm_multi), whose >= 8192-point tiles are
    // still the one-workgroup-per-CU kind: the tile is then chosen with their cost
    // real_stream: f32 windows and real taps (imaginary parts ignored), two overlap-save segments per Complex tile
    // (k_fftfilt_real; tiles up to 4096 points, so <= 4094 taps); the carry prefix then holds floats
    bool real_stream = false;
    // Front FirFilter fused in (rr_fir_fftfilter_create / rr_fir_fm_chain_create): `taps` are then the composite
    // t1 (*) t2 of L1 + L2 - 1 taps and front = L1 - 1.  The block behaves like FirFilter(t1) -> FftFilter(t2):
    // the FIR's L1 - 1 held-back samples stay in the caller's window (fir.rs:537), the carried history is L2 - 1
    // samples, fft_size / nsamples are the reference's for L2 taps, and the first L2 - 1 outputs — where the two-block
    // chain sees FftFilter's ZERO history instead of a warm FIR — are recomputed from their definition (head fix).
    size_t front = 0, hist = 0;       // hist = L - 1 - front: history samples carried in `prefix`
    DevBuf<cf> d_t1, d_t2, d_zhead;   // front > 0: caller-order taps of the two stages, z[0 .. L2 - 1 + G]
    size_t emitted = 0;               // outputs emitted so far (the head fix applies while it is 0)
    // tiles_only: never the any-size frames below 16384 taps (a decimating FirFilter stores every d-th sample of the TILES;
    // the frames have no such store, and the block would fall to the one-thread-per-output direct form: ADVICE r4)
    FftFilter(const rr_c32* taps, size_t ntaps, bool for_chain = false, int max_log2f = 14, bool real_stream = false,
              size_t front = 0, bool tiles_only = false);
    // t1 (*) t2 in f64, rounded once
    static std::vector<rr_c32> composite(const rr_c32* t1, size_t n1, const rr_c32* t2, size_t n2);
    void set_stage_taps(const rr_c32* t1, size_t n1, const rr_c32* t2, size_t n2);
    // samples of the window that can be filtered now (all of them without a front FIR)
    size_t avail(size_t in_len) const { return front ? (in_len > front ? in_len - front : 0) : in_len; }
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
    // out[n] = sum_k t[k] src[n + L - 1 - k], n < n_out (the tile kernel of the chosen size)
    // (carry: the caller's carry-state update, written by the same launch — common.hpp CarryOut)
    bool filter(VSrc<cf> src, cf* out, long n_out, hipStream_t s, CarryOut carry = {}, bool rb_ok = false);
    void filter_real(VSrc<float> src, float* out, long n_out, int d, hipStream_t s, CarryOut carry = {});   // out[m] = y[m d]
};

// The fused chains end in a RationalResampler (+ demodulator / scaler) whose reference form makes progress on ANY output
// window: it emits what fits and carries a `pending` sample across a full buffer (rational_resampler.rs:162-173,190-196),
// and the FftFilter in front of it writes into an inner 4 MB stream, not into the caller's window.  One fused kernel emits
// whole filter blocks, so when the caller's window is smaller than the next block's outputs that ONE block is run into
// `buf` (C windows of `cap` elements) and handed out as room appears — the resampler's `pending`, a block long.
struct OutTail {
    DevBuf<float> buf;
    size_t cap = 0, len = 0, pos = 0;          // per window: capacity, valid outputs, outputs already handed out
    bool pending() const { return pos < len; }
    // up to out_cap of the pending outputs of each of the C windows -> out (windows out_stride apart); returns the count
    size_t drain(float* out, size_t out_stride, size_t out_cap, size_t C, hipStream_t s);
};
// work_dev of a fused chain B on top of its block-granular core:
//   B::next_block_outputs()                         outputs the next filter block will add
//   B::work_blocks(in, n, out, stride, cap, c, p, need, s, max_blocks)   the one-kernel path; needs cap >= next_block_outputs()
// Protocol: pending outputs first (WAIT_DST need 1 while some remain, like the resampler with a pending sample); a window
// that holds the next block runs the core unchanged; a smaller one gets ONE block through the tail — WAIT_DST(1), part of
// it is necessarily left — or WAIT_SRC(need) when there was no full block of input (which is consumed all the same).
template <class B>
int trickle_work(B& b, OutTail& t, size_t C, const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed, size_t* produced,
                 size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    float* o = static_cast<float*>(out);
    const size_t stride = out_cap;
    size_t room = out_cap;
    if (t.pending()) {
        const size_t m = t.drain(o, stride, room, C, s);
        *produced = m; o += m; room -= m;
        if (t.pending()) { *need = 1; return RR_WAIT_DST; }
    }
    const size_t nb = b.next_block_outputs();
    if (nb <= room) {
        size_t p = 0;
        const int st = b.work_blocks(in, in_len, o, stride, room, consumed, &p, need, s, ~(uint64_t)0);
        *produced += p;
        return st;
    }
    if (t.cap < nb) { t.cap = nb + 64; t.buf.reserve(C * t.cap); }
    size_t p = 0;
    const int st = b.work_blocks(in, in_len, t.buf.p, t.cap, t.cap, consumed, &p, need, s, 1);
    if (st == RR_ERR) return st;
    if (p == 0) return st;                             // no full filter block yet: WAIT_SRC(need) (input joined the pending samples)
    t.len = p; t.pos = 0;
    const size_t m = t.drain(o, stride, room, C, s);
    *produced += m;
    *need = t.pending() ? 1 : 0;
    return t.pending() ? RR_WAIT_DST : RR_AGAIN;
}

// Fused FftFilter -> RationalResampler(interp:deci) -> QuadratureDemod (one kernel per call).
// Whole-stream output = what the three reference blocks produce in sequence: after N input
// samples, N1 = floor(N/nsamples)*nsamples filtered, N2 = ceil(N1*I/D) resampled, N2-1 demodulated.
// The fused chains' non-finite pass (kernels_misc.hip k_chain_blocks_nonfinite, round 6): the reversed time-domain taps it folds
// with, the three sequence-numbered verdict slots and the call counter.  on == false: no pass (RTL-SDR byte sources cannot carry
// a non-finite sample; rr_build_opts.fft_nonfinite_tiles = 1; ratios beyond one reference block per output).
struct ChainNf {
    bool on = false;
    DevBuf<cf> rev_c;                 // [channels][L]
    DevBuf<float> rev_f;              // [L] (audio chain)
    DevBuf<int> slots;                // [6]
    int seq = 1;
    void init(hipStream_t s);
};
struct FmChain : Block {
    std::unique_ptr<FftFilter> f;     // owns taps tables, history/pending prefix and tile choice
    int64_t I = 1, D = 1;
    float gain;
    int mode;
    uint64_t n1 = 0;                  // filtered samples emitted so far
    DevBuf<cf> last_r[2];
    int cur_lr = 0;
    // iq8: the input stream is RTL-SDR bytes (u8 I/Q pairs) and RtlSdrDecode (rtlsdr_decode.rs:35-42)
    // is fused in front: windows, `consumed` and WAIT_SRC `need` are then counted in BYTES.
    bool iq8 = false;
    DevBuf<cf> decoded;               // only for odd-addressed byte windows (decoded out of line)
    DevBuf<cf> d_tw_half;             // w_(F/2)^k: half-size inverse (interp 1, even deci, 2048-point tiles; k_fm_chain_half)
    bool half_ok = false;
    std::unique_ptr<PolyTables> poly; // interp 1, integer deci: decimate-first tiles (k_fm_chain_poly)
    bool window_aware = true;         // windows of too few tiles run on the 2048-point kernels (off when fm_poly is forced)
    OutTail tail;                     // one block's outputs the caller's window could not take yet
    ChainNf nf;                       // non-finite samples on the reference's blocks (round 6)
    FmChain(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool iq8 = false,
            int max_log2f = 14, const rr_c32* fir_taps = nullptr, size_t fir_ntaps = 0);
    size_t next_block_outputs() const;
    int work_blocks(const void*, size_t, float*, size_t, size_t, size_t*, size_t*, size_t*, hipStream_t, uint64_t max_blocks);
    int work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* c, size_t* p, size_t* need, hipStream_t s) override {
        return trickle_work(*this, tail, 1, in, in_len, out, out_cap, c, p, need, s);
    }
    bool eof(bool src_eof) override { return src_eof && !tail.pending(); }       // as rational_resampler.rs:209-213
};

// FftFilterFloat(taps) -> RationalResampler(interp, deci) -> MultiplyConst(scale) fused (the audio stage of
// examples/rtl_fm.rs:398-418): f32 in, f32 out, one real-stream tile kernel (k_audio_chain).  Whole-stream output = the
// three reference blocks in sequence: after N input samples N1 = floor(N / nsamples) * nsamples filtered,
// ceil(N1 * I / D) resampled and scaled.
struct AudioChain : Block {
    std::unique_ptr<FftFilter> f;     // real-stream filter: tables, history / pending prefix, tile choice
    int64_t I = 1, D = 1;
    float scale;
    uint64_t n1 = 0;                  // filtered samples emitted so far
    OutTail tail;
    ChainNf nf;
    AudioChain(const float* taps, size_t ntaps, size_t interp, size_t deci, float scale);
    size_t next_block_outputs() const;
    int work_blocks(const void*, size_t, float*, size_t, size_t, size_t*, size_t*, size_t*, hipStream_t, uint64_t max_blocks);
    int work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* c, size_t* p, size_t* need, hipStream_t s) override {
        return trickle_work(*this, tail, 1, in, in_len, out, out_cap, c, p, need, s);
    }
    bool eof(bool src_eof) override { return src_eof && !tail.pending(); }
};

// N fused FM chains (FmChain) on ONE shared input: the reference's Tee fan-out + N x three blocks.
// The forward FFT of every tile is computed once for all channels.  Output = N windows.
struct FmMulti : Block {
    size_t C;
    std::unique_ptr<FmChain> chain;   // channel 0's chain object: shared bookkeeping + input carry state
    DevBuf<cf> d_hpos_all;            // [C][F]
    DevBuf<cf> d_tw_half;             // w_(F/2)^k: half-size inverse transforms (interp 1, even deci; k_fm_multi_half)
    bool half_ok = false;
    std::unique_ptr<PolyTables> poly; // interp 1, deci 2..8: decimate-first tiles (k_fm_multi_poly)
    int poly_waves = 0;               // rr_build_opts.fm_poly = 8 / 12 forces that kernel variant (parity tests of both)
    DevBuf<cf> last_r[2];             // [C]
    int cur_lr = 0;
    // iq8: RTL-SDR byte input, RtlSdrDecode fused in front (windows, `consumed` and WAIT_SRC `need` count BYTES)
    bool iq8 = false;
    DevBuf<cf> decoded;               // odd-addressed byte windows are decoded out of line
    OutTail tail;                     // [C] windows of one block's outputs
    ChainNf nf;                       // (its own: every channel's taps; `chain->nf` stays off)
    FmMulti(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool iq8 = false);
    size_t out_windows() const override { return C; }
    size_t next_block_outputs() const { return chain->next_block_outputs(); }
    int work_blocks(const void*, size_t, float*, size_t, size_t, size_t*, size_t*, size_t*, hipStream_t, uint64_t max_blocks);
    int work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* c, size_t* p, size_t* need, hipStream_t s) override {
        return trickle_work(*this, tail, C, in, in_len, out, out_cap, c, p, need, s);
    }
    bool eof(bool src_eof) override { return src_eof && !tail.pending(); }
};

struct FftFilterFloat : Block {
    std::unique_ptr<FftFilter> inner;
    size_t cap = 0;
    // inner streams (fft_filter.rs:393-420).  With a real-stream inner filter they hold f32 (the Complex(x, 0)
    // lift and the .re projection are no-ops on the data); filters too long for it keep the Complex pair.
    bool real_inner = false;
    DevBuf<float> fin[2], fout[2];
    DevBuf<cf> iin[2], iout[2];
    int ci = 0, co = 0;
    size_t iin_len = 0, iout_len = 0;
    FftFilterFloat(const float* taps, size_t ntaps);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

struct Resampler : Block {
    int64_t I = 1, D = 1, counter = 0;
    bool has_pending = false;
    DevBuf<unsigned char> d_pending;
    Resampler(size_t interp, size_t deci, size_t es);
    bool eof(bool src_eof) override;
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

struct QuadDemod : Block {
    float gain;
    int mode;
    QuadDemod(float gain, int mode);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

// FftStream (fft_stream.rs:26-117): forward FFT of consecutive `size`-sample frames (power-of-two sizes here).
struct FftStream : Block {
    size_t size = 0;
    std::unique_ptr<AnyFft> fft;
    explicit FftStream(size_t size);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

// #[rustradio(sync)] blocks (rustradio_macros_code/src/lib.rs:458-515): map min(input, output space) samples,
// then WaitForStream on the side that ran dry.
struct MultiplyConst : Block {        // multiply_const.rs:6-23, T = Float (es 4) or Complex (es 8)
    float vr, vi;
    MultiplyConst(size_t es, float re, float im);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};
struct FastFM : Block {               // quadrature_demod.rs:144-165; q2, q1 = the two previous samples (zeros at start)
    DevBuf<cf> hist[2];
    int cur = 0;
    FastFM();
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

// RtlSdrDecode (rtlsdr_decode.rs:9-47): stateless u8 pair -> Complex conversion.
struct RtlSdrDecode : Block {
    RtlSdrDecode();
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

// ---- compositions inside the library (compose.cpp) --------------------------------------------------------------------
// Blocks in series behind ONE handle: inner blocks chained through device-resident buffers of the reference's stream
// capacity, one Graph::run-style round loop per work() call.  What the fused kernels do not cover runs as this.
struct Series : Block {
    std::vector<std::unique_ptr<Block>> b;
    struct Link {
        DevBuf<unsigned char> buf[2];     // linear window; consume() moves the rest to the front of the other buffer
        int cur = 0;
        size_t len = 0, cap = 0, es = 0;
    };
    std::unique_ptr<Link[]> link;         // link[i] between b[i] and b[i + 1]
    Series(const char* name, std::vector<std::unique_ptr<Block>> blocks);
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
    bool eof(bool src_eof) override;
};
// N blocks of one shape on ONE shared input window, N output windows (the reference's Tee fan-out + N chains).
struct Parallel : Block {
    std::vector<std::unique_ptr<Block>> ch;
    static constexpr size_t POOL = 4;             // streams the channels are spread over inside one work() call
    hipStream_t pool[POOL] = {};
    hipEvent_t joined[POOL] = {}, forked = nullptr;
    Parallel(const char* name, std::vector<std::unique_ptr<Block>> channels);
    ~Parallel() override;
    size_t out_windows() const override { return ch.size(); }
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
    bool eof(bool src_eof) override;
};
// What rr_fm_chain*_create / rr_fm_multi*_create / rr_audio_chain_create build: the fused block where its kernels reach, the
// unfused composition (same protocol, same output) for every other shape the separate blocks take — no constructor cliffs.
// mode: RR_ATAN2_EXACT / RR_ATAN2_FAST (QuadratureDemod) or RR_DEMOD_FASTFM (FastFM, quadrature_demod.rs:144-165).
Block* make_fm_chain(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool u8,
                     const rr_c32* fir_taps, size_t fir_ntaps);
Block* make_fm_multi(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool u8);
Block* make_audio_chain(const float* taps, size_t ntaps, size_t interp, size_t deci, float scale);

struct Hilbert : Block {
    FirPlan pl;
    DevBuf<float> d_tp, d_rev;
    DevBuf<int> d_nf_flags;           // one word per workgroup of k_hilbert (zeroed once; the repair launch clears what it consumed)
    NanFix nanfix() const { NanFix f; f.rev = d_rev.p; f.L = pl.L; f.d = 1; f.kind = NANFIX_HILBERT; f.wgflags = d_nf_flags.p; return f; }
    DevBuf<float> hist[2];
    int cur = 0;
    // zero-tap skipping (kernels_fir.hip k_hilbert): taps of one parity only
    bool skip_ok = false;
    int par = 0, Q = 0;
    DevBuf<float> d_hq;
    // Round 4: long transformers.  The pair-sample kernel costs 0.24 ms per 1e8 samples up to ~129 taps and then grows with
    // the taps (255: 0.41, 1001: 1.2, 4001: 17.5 on the generic fallback); the real-stream overlap-save tiles with a Complex
    // store (k_fftfilt_real<.., HILB>) stay flat up to 3584 taps, and beyond them the transformer is the Complex filter
    // delta[j - L/2] + i h[j] on Complex(x, 0) (FirFilter<Complex>'s kernels, in chunks through two work buffers).
    std::unique_ptr<FftFilter> fftk;
    std::unique_ptr<FirC32> wide;
    DevBuf<cf> wide_in;
    // the tile kernel of fftk carries no non-finite hooks: a pass behind it (kernels_misc.hip k_hilbert_refold_nonfinite); on a
    // page-locked host output window the host probes its own memory after the completion wait and launches it only on a hit
    struct { bool on = false; VSrc<float> src{}; void* out = nullptr; long n = 0, P = 0; } deferred;
    void host_out_done(const void* out_host, size_t produced) override;
    void host_out_reset() override { deferred.on = false; }
    Hilbert(size_t ntaps, int window, float parm);
    ~Hilbert() override;
    int work_dev(const void*, size_t, void*, size_t, size_t*, size_t*, size_t*, hipStream_t) override;
};

}  // namespace rr
