// blocks.cpp — host side of each block: the reference's work() bookkeeping (thresholds,
// consumed/produced, WaitForStream amounts, carry state) restated over explicit windows,
// with the sample arithmetic handed to the HIP kernels.  Citations are to
// /root/reference/src.  There is no CPU compute path here: every branch that produces
// samples launches a kernel.
#include "blocks.hpp"

#include <mutex>
#include "rotor_host.hpp"

#include <algorithm>
#include <cstdint>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include "stage.hpp"
#include "taps.hpp"

namespace rr {

// The per-call kernel choices below compare a window's tiles / batches with the chip's resident workgroup slots.  The
// crossovers were measured on one MI355X (256 CUs; tools/call_overhead.py, tools/prune_window_probe*.py) and are kept as
// multiples of THAT chip's slots: on a partitioned (CPX) or smaller device they scale with its CU count.
static size_t chip_units(size_t measured_on_256_cus) {
    const size_t cus = (size_t)std::max(1, device_cu_count());
    return std::max<size_t>(1, (measured_on_256_cus * cus + 128) / 256);
}

static thread_local int g_device = 0;
void set_thread_device(int d) { g_device = d; }
int thread_device() { return g_device; }

// ---- Block base ---------------------------------------------------------------------------
Block::Block(const char* nm, size_t ies, size_t oes) : name(nm), in_es(ies), out_es(oes) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error("rustradio_amd: no usable HIP device (this library has no CPU fallback)");
    if (g_device < 0 || g_device >= n) throw Error("rustradio_amd: device ordinal out of range");
    device = g_device;
    RR_HIP(hipSetDevice(device));
    RR_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
}
Block::~Block() {
    (void)hipSetDevice(device);
    for (auto& e : prof_evs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (stream) {
        (void)hipStreamSynchronize(stream);
        (void)hipStreamDestroy(stream);
    }
}
void Block::sync() {
    RR_HIP(hipSetDevice(device));
    RR_HIP(hipStreamSynchronize(stream));
    if (last_stream != stream) RR_HIP(hipStreamSynchronize(last_stream));   // nullptr = default stream
}
void Block::prof_begin(hipStream_t s) {
    if (!prof_on) return;
    if (prof_used == prof_evs.size()) {
        hipEvent_t a, b;
        RR_HIP(hipEventCreate(&a));
        RR_HIP(hipEventCreate(&b));
        prof_evs.emplace_back(a, b);
    }
    RR_HIP(hipEventRecord(prof_evs[prof_used].first, s));
}
void Block::prof_end(hipStream_t s) {
    if (!prof_on) return;
    RR_HIP(hipEventRecord(prof_evs[prof_used].second, s));
    prof_used++;
}
void Block::prof_read(double* total_ms, size_t* launches, bool reset) {
    double t = 0;
    for (size_t i = 0; i < prof_used; i++) {
        RR_HIP(hipEventSynchronize(prof_evs[i].second));
        float ms = 0;
        RR_HIP(hipEventElapsedTime(&ms, prof_evs[i].first, prof_evs[i].second));
        t += ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = prof_used;
    if (reset) prof_used = 0;
}
bool Block::eof(bool src_eof) { return src_eof; }   // rustradio_macros_code/src/lib.rs:596-623

// The device's view of a host window that lies WHOLLY inside a range page-locked through rr_host_register (the caller's
// stream ring; the device address is taken once, at registration: a window costs a short scan under a lock), or nullptr
// for any other memory, which is staged through device memory.  Only ranges the library was told about qualify: memory
// page-locked behind its back (hipHostMalloc by the caller) can be freed behind its back too, and an answer remembered for
// such a pointer would outlive the mapping — it takes the staged path, whose copies are DMA from page-locked memory anyway.
//
// WHICH ranges run zero-copy (round 5).  Round 4's churn probe (tools/zerocopy_churn.py: register fresh heap arrays, run
// kernels on them in place, unregister, free, again) showed kernels that work in place on a RE-registered range now and
// then missing — runs of outputs never written, or computed from the previous contents: 24 of 4,500 calls in a bad pass —
// and only on ranges that shared their first or last PAGE with a neighbour of an earlier registration; page-aligned private
// mappings never missed, nor did a range registered once (12,000 calls), DMA copies or pageable windows.  The root cause
// (a stale device translation for a page that was page-locked, released and page-locked again) is the platform's and is not
// closed; the library therefore hands a range to kernels in place only when NO page of it can have such a history:
//   * base and size are multiples of the page size (a page-lock covers whole pages: nothing is shared with a neighbour) —
//     what the reference's ring is (circular_buffer.rs:98-128, one mmap of 2 x len) and what the shims register;
//   * no page of it overlaps a range that is registered now or was EVER registered and released in this process.
// Every rr_host_register call is tracked, admitted or not, so releasing a refused range retires its pages too.  Everything
// else is still page-locked (the staged copies are direct DMA) but goes through device memory.  RR_ZERO_COPY=0 in the
// environment turns the in-place path off altogether.
namespace {
struct HostRange { const unsigned char* base; size_t bytes; unsigned char* dev; };   // dev == nullptr: tracked, not admitted
std::mutex g_host_m;
std::vector<HostRange> g_host_ranges;
std::vector<std::pair<const unsigned char*, const unsigned char*>> g_retired;    // page-rounded [lo, hi) of released ranges
// the SYSTEM page size (a page-lock covers whole pages of it: on a 64K-page kernel a 4K-aligned range still shares its first
// and last page with its neighbours, the very condition this policy excludes)
const uintptr_t PAGE = [] { const long p = sysconf(_SC_PAGESIZE); return (uintptr_t)(p > 0 ? p : 4096); }();
bool zero_copy_enabled() {
    static const bool on = [] { const char* e = getenv("RR_ZERO_COPY"); return !(e && e[0] == '0'); }();
    return on;
}
}  // namespace
static const unsigned char* page_floor(const unsigned char* p) { return reinterpret_cast<const unsigned char*>(reinterpret_cast<uintptr_t>(p) & ~(PAGE - 1)); }
static const unsigned char* page_ceil(const unsigned char* p) { return page_floor(p + (PAGE - 1)); }
void host_range_add(void* base, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(base);
    const unsigned char* lo = page_floor(b);
    const unsigned char* hi = page_ceil(b + bytes);
    void* dev = nullptr;
    bool admit = zero_copy_enabled() && lo == b && hi == b + bytes;
    if (admit && (hipHostGetDevicePointer(&dev, base, 0) != hipSuccess || !dev)) { (void)hipGetLastError(); admit = false; }
    std::lock_guard<std::mutex> g(g_host_m);
    for (auto& r : g_retired)
        if (lo < r.second && r.first < hi) admit = false;     // a page of it was registered and released before
    for (auto& r : g_host_ranges)
        if (lo < page_ceil(r.base + r.bytes) && page_floor(r.base) < hi) admit = false;   // shares a page with a live registration
    g_host_ranges.push_back({b, bytes, admit ? static_cast<unsigned char*>(dev) : nullptr});
}
void host_range_remove(void* base) {
    std::lock_guard<std::mutex> g(g_host_m);
    for (size_t i = 0; i < g_host_ranges.size(); i++)
        if (g_host_ranges[i].base == base) {
            const unsigned char* lo = page_floor(g_host_ranges[i].base);
            const unsigned char* hi = page_ceil(g_host_ranges[i].base + g_host_ranges[i].bytes);
            // merge with overlapping / adjacent retired ranges so that the list stays short under churn
            for (size_t k = 0; k < g_retired.size();) {
                if (lo <= g_retired[k].second && g_retired[k].first <= hi) {
                    lo = std::min(lo, g_retired[k].first); hi = std::max(hi, g_retired[k].second);
                    g_retired.erase(g_retired.begin() + (long)k);
                } else k++;
            }
            g_retired.emplace_back(lo, hi);
            g_host_ranges.erase(g_host_ranges.begin() + (long)i);
            break;
        }
}
// whether [host, host + bytes) lies wholly inside a range that IS page-locked through rr_host_register right now (admitted to
// zero-copy or not): such memory may be handed to hipMemcpyAsync — the runtime finds the caller's registration.  Anything else
// goes through stage.hpp.
bool host_range_registered(const void* host, size_t bytes) {
    if (!host) return false;
    const unsigned char* h = static_cast<const unsigned char*>(host);
    std::lock_guard<std::mutex> g(g_host_m);
    for (auto& r : g_host_ranges)
        if (h >= r.base && bytes <= r.bytes && (size_t)(h - r.base) <= r.bytes - bytes) return true;
    return false;
}
void* device_view_of_host(const void* host, size_t bytes) {
    if (!host) return nullptr;
    const unsigned char* h = static_cast<const unsigned char*>(host);
    std::lock_guard<std::mutex> g(g_host_m);
    for (auto& r : g_host_ranges)
        if (r.dev && h >= r.base && bytes <= r.bytes && (size_t)(h - r.base) <= r.bytes - bytes) return r.dev + (h - r.base);
    return nullptr;
}

// Host-window work().  Page-locked windows (round 4): the kernels read the input window and write the output window over
// PCIe THEMSELVES — the link is full duplex, so the window going down overlaps the one coming up, which two DMA copies on
// this pool do not (tools/zerocopy_probe.py, 4,096,000-byte windows: FftFilter 193 -> 148 us, the fused
// RTL-SDR chain 175 -> 109, Hilbert -> FirFilter /8 316 -> 120).  Pageable windows are staged through device memory.
int Block::work_host(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                     size_t* produced, size_t* need) {
    RR_HIP(hipSetDevice(device));
    last_stream = stream;
    // (the window seen from the device, if it lies in a registered range: read in place by the block's kernels, or — blocks that
    //  read it more than once, or in narrow loads — copied down by a kernel first: 55 GB/s against hipMemcpyAsync's 16-18)
    void* vin = in_len ? device_view_of_host(in, in_len * in_es) : nullptr;
    void* din = zero_copy_in ? vin : nullptr;
    void* dout = out_cap ? device_view_of_host(out, out_cap * out_es * out_windows()) : nullptr;
    const size_t in_use = in_len;   // whole window: kernels may touch (zero-weighted) samples past the consumed range
    if (!din) {
        st_in.reserve(std::max<size_t>(in_use * in_es, 16));
        if (in_use && vin) launch_copy_bytes(vin, st_in.p, in_use * in_es, stream);
        else if (in_use && host_range_registered(in, in_use * in_es)) RR_HIP(hipMemcpyAsync(st_in.p, in, in_use * in_es, hipMemcpyHostToDevice, stream));
        else if (in_use) hstage().h2d(st_in.p, in, in_use * in_es, stream);     // pageable: never a DMA out of the caller's memory (stage.hpp)
        din = st_in.p;
    }
    if (!dout) st_out.reserve(std::max<size_t>(out_cap * out_es * out_windows(), 16));
    host_out = dout ? out : nullptr;
    host_out_reset();       // (a pass an earlier call deferred and then failed to run — a throw between work_dev and the wait — is stale)
    int st;
    try { st = work_dev(din, in_len, dout ? dout : st_out.p, out_cap, consumed, produced, need, stream); }
    catch (...) { host_out = nullptr; host_out_reset(); throw; }
    if (!dout && *produced) {
        if (host_range_registered(out, out_cap * out_es * out_windows())) {
            if (out_windows() == 1) RR_HIP(hipMemcpyAsync(out, st_out.p, *produced * out_es, hipMemcpyDeviceToHost, stream));
            else RR_HIP(hipMemcpy2DAsync(out, out_cap * out_es, st_out.p, out_cap * out_es, *produced * out_es, out_windows(), hipMemcpyDeviceToHost, stream));
        } else {
            hstage().d2h_2d(out, out_cap * out_es, st_out.p, out_cap * out_es, *produced * out_es, out_windows(), stream);
        }
    }
    // (polling an event from the calling thread instead was measured and changes nothing: 154 -> 158 us per reference-sized
    //  window; the call sits on the link — tools/micro/pcie_inplace.hip: a kernel moves 4,096,000 bytes each way in 128-145 us)
    RR_HIP(hipStreamSynchronize(stream));
    if (hstage_) hstage_->quiesced();
    if (host_out) {
        const void* h = host_out;
        host_out = nullptr;
        host_out_done(h, *produced);
    }
    return st;
}

// ---- FirFilter<T> (fir.rs:303-551) -----------------------------------------------------------
template <class TapT> static void build_poly(const std::vector<TapT>& rev, int d, int& qpad, std::vector<TapT>& tp) {
    const int L = (int)rev.size();
    const int qmax = (L + d - 1) / d;
    qpad = (qmax + 7) / 8 * 8;
    tp.assign((size_t)d * qpad, TapT{});
    for (int k = 0; k < L; k++) tp[(size_t)(k % d) * qpad + k / d] = rev[k];
}

static void compute_hpos(const rr_c32* taps, size_t ntaps, int log2f, std::vector<cf>& hpos);

// decimating FirFilter<Complex>: where the decimate-first tiles beat the block's other kernels (MI355X, tools/fir_poly_probe.py,
// ms per 1e8 samples, poly / best other, real-valued taps — Complex taps only slow the direct form down):
//   /3: 31 taps 0.264 / 0.410, 127 0.264 / 0.278, 401 0.281 / 0.346, 1000 0.346 / 0.427      /4: 401 0.260 / 0.261, 1000 0.292 / 0.413
//   /5: 127 0.251 / 0.320, 1000 0.276 / 0.417     /6: 127 0.248 / 0.259, 401 0.256 / 0.316, 2467 0.321 / 0.575
//   /7: 255 0.274 / 0.311, 2467 0.348 / 0.583     /8: 1000 0.321 / 0.306, 2467 0.355 / 0.593 (pruned inverse below that)
//   /10: 31 0.317 / 0.355, 255 0.319 / 0.518, 2467 0.349 / 0.580      /12: 2467 0.519 / 0.579      /2, /16: never
// deci = D * sub on the pruned tile of D (every sub-th kept sample stored): it beats or ties the block's other kernels wherever
// its tables exist (tools/prune_sub_probe.py, ms per 1e8 samples, pruned / other: 255 taps /32 0.222 / 0.295, 401 taps /24
// 0.215 / 0.320, 1000 taps /64 0.236 / 0.389, 31 taps /12 0.189 / 0.268; FirFilter<Float> 401 taps /40 0.120 / 0.190;
// Hilbert(65) -> FirFilter 255 taps /32 0.172 / 0.508, 401 taps /24 0.149 / 0.709, 2000 taps /32 0.255 / 0.823)
static bool FIR_PRUNE_SUB_DEFAULT(size_t, size_t, size_t) { return true; }
static bool fir_poly_wins(size_t ntaps, size_t deci) {
    // round 4: every decimation up to 16, and up to 768 taps per phase (tools/fir_long_probe.py, ms per 1e8 samples, other /
    // decimate-first: /6 3599 taps 0.747 / 0.458, /8 4799 taps 0.823 / 0.458, /10 7599 taps 1.058 / 0.804; /4 wins up to ~700
    // per phase, /3 up to ~560) — and tools/fir_cliff_probe.py for the new decimations: /9 1000 taps 0.396 / 0.310, /11 2467
    // 0.601 / 0.359, /13 2467 0.597 / 0.450, /16 5000 0.836 / 0.588; /8 1000 taps 0.312 / 0.240
    const size_t Ls = (ntaps + deci - 1) / deci;
    if (Ls > (deci == 3 ? 560u : deci == 4 ? 700u : 768u)) return false;
    switch (deci) {
    case 3: case 10: case 11: return ntaps >= 24;
    case 4: return ntaps >= 400;
    case 5: case 6: case 7: return ntaps >= 100;   // (/7 on the three-waves-per-SIMD kernel since round 4: 127 taps 0.225 / 0.250)
    case 8: case 12: return ntaps >= 700;
    case 9: return ntaps >= 300;
    case 13: case 14: case 15: case 16: return ntaps >= 2000;
    default: return false;
    }
}

FirC32::FirC32(const rr_c32* taps, size_t ntaps, size_t deci, bool translate, float samp_rate, float freq, bool allow_fft)
    : Block("FirFilter<Complex>", 8, 8) {
    if (ntaps == 0) throw Error("FirFilter: empty taps");            // fir.rs:372
    if (deci == 0) throw Error("FirFilter: decimation 0");           // fir.rs:319
    if (ntaps > 0x3fffffff || deci > 0x3fffffff) throw Error("FirFilter: taps/deci too large");
    std::vector<std::complex<float>> t(ntaps);
    for (size_t i = 0; i < ntaps; i++) t[i] = {taps[i].re, taps[i].im};
    if (translate) {                                                 // fir.rs:430-462
        if (!(samp_rate > 0.0f)) throw Error("FirFilter::translate: samp_rate must be > 0");
        if (freq != 0.0f) {
            const double input_step = 2.0 * 3.14159265358979323846 * (double)freq / (double)samp_rate;
            const float sr = (float)std::cos(input_step), si = (float)std::sin(input_step);
            float pr = 1.0f, pi = 0.0f;
            for (size_t k = 0; k < ntaps; k++) {                      // f32 recurrence, num-complex order
                const float ar = t[k].real(), ai = t[k].imag();
                t[k] = {ar * pr - ai * pi, ar * pi + ai * pr};
                const float nr = pr * sr - pi * si, ni = pr * si + pi * sr;
                pr = nr; pi = ni;
            }
            const double first = -input_step * (double)(ntaps - 1);
            const double ostep = -input_step * (double)deci;
            rot_on = true;
            ph0x = (float)std::cos(first); ph0y = (float)std::sin(first);
            stx = (float)std::cos(ostep); sty = (float)std::sin(ostep);
        }
    }
    pl.L = (int)ntaps; pl.d = (int)deci; pl.cfg = build_opts().fir_cfg;
    h_taps = t;
    bool real_taps = true;
    for (auto& c : t) if (c.imag() != 0.0f) real_taps = false;
    pl.complex_taps = !real_taps;
    if (real_taps) {   // Complex::new(t, 0) taps: 2 FMA per tap instead of 4
        std::vector<float> rev(ntaps), tp;
        for (size_t j = 0; j < ntaps; j++) rev[j] = t[ntaps - 1 - j].real();   // fir.rs:160
        build_poly(rev, pl.d, pl.qpad, tp);
        d_rev.upload(reinterpret_cast<unsigned char*>(rev.data()), rev.size() * 4, stream);
        d_tp.upload(reinterpret_cast<unsigned char*>(tp.data()), tp.size() * 4, stream);
    } else {
        std::vector<cf> rev(ntaps), tp;
        for (size_t j = 0; j < ntaps; j++) rev[j] = mkcf(t[ntaps - 1 - j].real(), t[ntaps - 1 - j].imag());
        build_poly(rev, pl.d, pl.qpad, tp);
        d_rev.upload(reinterpret_cast<unsigned char*>(rev.data()), rev.size() * 8, stream);
        d_tp.upload(reinterpret_cast<unsigned char*>(tp.data()), tp.size() * 8, stream);
    }
    {   // (nan_fix.hpp: the reference's fold over the reversed taps, Complex whatever the specialisation above)
        std::vector<cf> rev(ntaps);
        for (size_t j = 0; j < ntaps; j++) rev[j] = mkcf(t[ntaps - 1 - j].real(), t[ntaps - 1 - j].imag());
        d_fix.upload(rev.data(), rev.size(), stream);
    }
    RR_HIP(hipStreamSynchronize(stream));
    // d = 1: overlap-save tiles cost a flat ~0.32 ms per 1e8 samples; the direct form stays at its 0.30 ms of
    // staging up to ~32 real / ~24 Complex taps and then grows by 0.004 / 0.007 ms per tap
    // (tools/fir_paths_probe.py on MI355X: 127 real taps 0.65 vs 0.32 ms, 127 Complex taps 1.09 vs 0.32 ms)
    const BuildOpts& bo = build_opts();
    const bool force_direct = bo.fir_path == RR_PATH_DIRECT, force_fft = bo.fir_path == RR_PATH_FFT;
    window_aware = bo.fir_poly <= 0 && bo.fir_prune <= 0 && !force_fft;   // a forced path is used at every window size
    // d > 1: the same tiles with a decimating store (k_fftfilt_deci, k_fftfilt_split<.., true>).  The transform cost
    // per input sample does not shrink with d while the direct form's does, so the bar is on taps per output
    // phase; beyond ~320 taps the direct form's LDS tile no longer fits for most decimations and it collapses
    // (tools/fir_deci_probe.py: 401 taps /16 0.69 vs 0.32 ms, 1000 taps /16 95 vs 0.43 ms per 1e8 samples).
    const size_t min_taps = real_taps ? 40 : 28, min_per_phase = real_taps ? 36 : 16;
    const bool fits = ntaps <= 16383 && deci <= 4096;
    // Round 4 (tools/fir_cliff_probe.py): from /11 on the direct form's tile shapes get small or stop fitting (127 taps: /9
    // 0.20, /11 0.44, /15 0.60, /20 9.6 ms per 1e8 samples — the last one is the one-thread-per-output fallback) while the
    // tiles with a decimating store stay at 0.33 for any decimation
    const bool wins = deci == 1 ? ntaps >= min_taps
                                : (ntaps >= 320 || ntaps / deci >= min_per_phase || deci >= 11 || !fir_direct_has_tile(pl, sizeof(cf), sizeof(cf)));
    // deci 4 / 8 / 16: the tile whose last radix is the decimation, inverse transform pruned to 1/deci
    // (tools/prune_probe.py, ms per 1e8 samples, direct / decimating store / pruned: 255 Complex taps /8 0.31 / 0.31 /
    //  0.21, 401 taps /4 0.60 / 0.34 / 0.27, 1000 taps /16 98 / 0.45 / 0.29; short /8 filters stay direct: 127 taps 0.195 / 0.27 / 0.207)
    // Round 4: multiples of those — deci = D * sub, D the largest of 16 / 8 / 4 dividing it — run the tile of D and store every
    // sub-th kept sample (thresholds: tools/prune_sub_probe.py).
    const bool have_split = prune_split(deci, prune_D, prune_sub);
    const size_t per_phase = have_split ? ntaps / prune_D : 0;
    const bool prune_default = !have_split ? false
                               : prune_sub > 1 ? FIR_PRUNE_SUB_DEFAULT(ntaps, prune_D, prune_sub)
                               : prune_D == 4 ? per_phase >= 8 : prune_D == 8 ? per_phase >= (real_taps ? 28 : 16) : per_phase >= 4;
    const bool prune_wins = bo.fir_prune ? bo.fir_prune > 0 : prune_default;
    if (allow_fft && !force_direct && prune_wins && have_split) {
        std::vector<std::complex<double>> td(ntaps);
        for (size_t i = 0; i < ntaps; i++) td[i] = {t[i].real(), t[i].imag()};
        prune.reset(new PruneTables());
        if (!prune->build(td, prune_D, false, stream)) prune.reset();
    }
    // other even decimations: 2048-point tiles, folded 1024-point inverse (k_fftfilt_half); same bar as the tiles with a
    // decimating store, which it replaces where it applies
    const bool half_on = bo.fir_half >= 0;
    if (!prune && allow_fft && !force_direct && (force_fft || wins) && half_on && fftfilt_half_supported((int)std::min<size_t>(ntaps, 1 << 20), (long)std::min<size_t>(deci, 1 << 20))) {
        const size_t F = 2048;
        std::vector<rr_c32> ct(ntaps);
        for (size_t i = 0; i < ntaps; i++) ct[i] = {t[i].real(), t[i].imag()};
        std::vector<cf> hpos, tw(F), twh(F / 2);
        compute_hpos(ct.data(), ntaps, 11, hpos);
        for (size_t k = 0; k < F; k++) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)F;
            tw[k] = mkcf((float)std::cos(a), (float)std::sin(a));
        }
        for (size_t k = 0; k < F / 2; k++) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)(F / 2);
            twh[k] = mkcf((float)std::cos(a), (float)std::sin(a));
        }
        d_hhpos.upload(hpos.data(), F, stream);
        d_htw.upload(tw.data(), F, stream);
        d_htw_half.upload(twh.data(), F / 2, stream);
        RR_HIP(hipStreamSynchronize(stream));
        half_ok = true;
    }
    // decimations the last radix cannot prune (3, 5, 6, 7 ...): decimate-first tiles — deci phase transforms and ONE inverse
    // per 1024 output positions (kernels_poly.hip; VERDICT r1 #3 "likewise for FirFilter decimations divisible by 3").
    // Where they win is measured (tools/fir_poly_probe.py); fir_poly > 0 forces them wherever the kernel exists.
    if (allow_fft && !force_direct && deci >= 2 && bo.fir_poly >= 0 &&
        fm_poly_supported(1, (long)std::min<size_t>(deci, 1 << 20), (int)std::min<size_t>(ntaps, 1 << 24), false)) {
        const bool poly_default = fir_poly_wins(ntaps, deci);
        if (bo.fir_poly > 0 || poly_default) {
            std::vector<rr_c32> ct(ntaps);
            for (size_t i = 0; i < ntaps; i++) ct[i] = {t[i].real(), t[i].imag()};
            poly.reset(new PolyTables());
            if (!poly->build(ct.data(), 1, ntaps, deci, false, stream)) poly.reset();
            RR_HIP(hipStreamSynchronize(stream));
        }
    }
    if (poly) { prune.reset(); half_ok = false; }
    // (the pruned / decimate-first kernels need a window of many tiles to fill the chip — work_dev picks per call; the tiles
    //  with a decimating store stay available beside them as the small-window path of long filters)
    if (half_ok) {
    } else if (allow_fft && fits && !force_direct && (force_fft || wins || poly || prune)) {
        std::vector<rr_c32> ct(ntaps);
        for (size_t i = 0; i < ntaps; i++) ct[i] = {t[i].real(), t[i].imag()};
        // deci > 1 near the top of the tile range: a 16384-point tile keeps 16385 - L samples (16383 taps: 2), so — like the
        // plain filter — from where the any-size frames are cheaper (15293 taps on) the block runs THEM at the full rate
        // and keeps every deci-th output with a strided copy (work_dev: `big`); the tiles with a decimating store below that.
        fftk.reset(new FftFilter(ct.data(), ntaps, false, 14, false, 0, false));
        if (deci > 1 && !fftk->big && fftk->log2f >= 13 && !fftk->nsub) fftk.reset();   // (RR_FFT_NO_SPLIT measurement runs)
        if (fftk) { fftk->nanfix = nanfix(); fftk->nanfix.d = 1; }        // (fftk->filter is the full-rate form)
    }
}
FirC32::~FirC32() {
    (void)hipSetDevice(device);
    if (rot_stream) {
        (void)hipStreamSynchronize(rot_stream);                      // (every copy out of the host ring has landed)
        hrot.reset();
        for (auto& c : copies) (void)hipEventDestroy(c.ev);
        for (auto& e : free_events) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(rot_stream);
        (void)hipEventDestroy(ev_gen);
        (void)hipEventDestroy(ev_used);
    }
}

int FirC32::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                     size_t* produced, size_t* need, hipStream_t s) {
    const size_t L = pl.L, d = pl.d;
    *consumed = *produced = *need = 0;
    const size_t absolute_minimum = L + d - 1;                       // fir.rs:498-501
    if (in_len < absolute_minimum) { *need = absolute_minimum; return RR_WAIT_SRC; }
    size_t n = d * ((in_len - L + 1) / d);                           // fir.rs:502
    if (out_cap < 1) { *need = 1; return RR_WAIT_DST; }              // fir.rs:511-515
    n = std::min(n, out_cap * d);                                    // fir.rs:518
    const size_t out_n = n / d;
    VSrc<cf> src{nullptr, 0, static_cast<const cf*>(in), (long)in_len};
    prof_begin(s);
    // Window-size-aware choice (round 2, tools/call_overhead.py): the batched pruned inverse (D tiles per workgroup) and the
    // decimate-first tiles (d x 1024 inputs per workgroup) only pay once a window holds more units than the chip has
    // workgroup slots — at the reference's ring size (512,000 samples) they leave most CUs idle: 255 taps /8 pruned 34 us
    // against 8.7 us direct, 401 taps /6 decimate-first 18.7 against 9.9 us.  FirFilter carries no state between calls
    // (the window holds the history), so the choice is per call.
    bool use_poly = poly != nullptr, use_prune = prune != nullptr;
    if (window_aware && (use_poly || use_prune)) {
        if (use_poly) {
            const size_t Ls = (L + d - 1) / d, per_tile = 1024 - Ls;
            use_poly = out_n >= chip_units(1000) * per_tile;                   // one tile per resident workgroup slot
        } else {
            const size_t pd = prune_D, F = (size_t)1 << prune->log2f, per_batch = (F - L + 1) / pd * pd * pd;   // D tiles of F - L + 1 inputs
            // (crossover in batches, tools/prune_window_probe.py: ~1.0-1.5x the resident workgroup slots of the tile —
            //  64-thread tiles at /4: 2000, 128-thread at /8: 1500, 256-thread at /16: 550)
            use_prune = n >= chip_units(pd == 4 ? 2000 : pd == 8 ? 1500 : 550) * per_batch;
        }
    }
    bool small_direct = false;
    if ((poly && !use_poly) || (prune && !use_prune)) {
        // small window: direct form (5 us + 7.8e-8 us per tap-phase and sample, twice that for Complex taps) or the tiles
        // with a decimating store (9 us + 3.1e-6 us per sample), whichever the fitted costs favour
        const double t_direct = 5.0 + 7.8e-8 * (double)n * ((double)L / (double)d) * (pl.complex_taps ? 2.0 : 1.0);
        const double t_tiles = 9.0 + 3.1e-6 * (double)n;
        // (beyond ~320 taps the direct form's LDS tile stops fitting and it collapses: never there)
        small_direct = !fftk || (L <= 320 && t_direct <= t_tiles && fir_direct_has_tile(pl, sizeof(cf), sizeof(cf)));
    }
    if (use_poly) launch_fir_poly(src, static_cast<cf*>(out), (long)out_n, (int)L, (int)d, poly->d_tw.p, poly->d_h.p, s, nanfix());
    else if (small_direct) launch_fir_c32(pl, d_tp.p, d_rev.p, src, static_cast<cf*>(out), (long)out_n, s, nanfix());
    else if (half_ok) launch_fftfilt_half(src, static_cast<cf*>(out), (long)out_n, (int)L, (int)d, d_htw.p, d_htw_half.p, d_hhpos.p, s, nanfix());
    else if (use_prune) launch_fftfilt_prune_c32(prune->log2f, src, static_cast<cf*>(out), (long)out_n, (int)L, prune->d_tw.p, prune->d_h2.p, prune->d_twb.p, s, (int)prune_sub, nanfix());
    else if (fftk && d > 1 && fftk->big) {
        // full-rate overlap-save frames into a scratch buffer, chunk by chunk, then out[m] = y[m d] (the resampler's index
        // pick with I = 1: a strided copy).  A window position o0 d of the input starts chunk o0.
        const size_t chunk = std::max<size_t>(1, ((size_t)1 << 22) / d);
        for (size_t o0 = 0; o0 < out_n; o0 += chunk) {
            const size_t m = std::min(chunk, out_n - o0), nfull = (m - 1) * d + 1;
            VSrc<cf> sc{nullptr, 0, static_cast<const cf*>(in) + o0 * d, (long)(in_len - o0 * d)};
            dec_tmp.reserve(nfull);
            fftk->filter(sc, dec_tmp.p, (long)nfull, s);
            launch_resample(dec_tmp.p, static_cast<cf*>(out) + o0, sizeof(cf), 0, nullptr, (long)m, 1, (long)d, 0, s);
        }
    }
    else if (fftk && d > 1 && fftk->nsub) launch_fftfilt_split_deci(fftk->nsub, src, static_cast<cf*>(out), (long)out_n, (int)L, (int)d, fftk->d_tw4096.p, fftk->d_hs.p, fftk->d_wk.p, s, nanfix());
    else if (fftk && d > 1) launch_fftfilt_deci(fftk->log2f, src, static_cast<cf*>(out), (long)out_n, (int)L, (int)d, fftk->d_tw.p, fftk->d_hpos.p, s, nanfix());
    else if (fftk) fftk->filter(src, static_cast<cf*>(out), (long)out_n, s);
    else launch_fir_c32(pl, d_tp.p, d_rev.p, src, static_cast<cf*>(out), (long)out_n, s, nanfix());
    prof_end(s);
    rotate_output(static_cast<cf*>(out), out_n, s);                  // fir.rs:531
    *consumed = n; *produced = out_n;
    return RR_AGAIN;                                                 // fir.rs:549
}

// The reference's rotator (fir.rs:464-473): out[m] *= phase; phase *= step, in f32, never renormalised.
//   RR_ROT_REPLAY (default): that recurrence, bit for bit, for any stream length.  It is inherently serial (10 ns per
//     output on one device lane, 2.6 ns on a host core) but data-independent, so it is generated AHEAD: after every call the
//     side stream walks the chain on into a ring of phases for the next window while this window's filter kernels (and
//     whatever the graph runs next) execute; a call waits only for the part of its range the chain has not reached.  A
//     graph paced by its source (100 Msps / 8 = 12.5 M outputs/s in BASELINE configs[4]) never waits for the device
//     chain (100 M outputs/s); a block that does (back-to-back batch calls) is moved onto a host generator thread
//     (rotate_output: `starving`, rotor_start_host).
//   RR_ROT_MODEL (opt-in): phase0 * step^m in f64, parallel; NOT within 1e-5 of the reference beyond ~1e5 outputs
//     (it does not reproduce the recurrence's accumulated rounding; tests/test_gpu_edges_fullsize.py).
void FirC32::rotor_generate(size_t upto) {
    if (upto <= rot_gen) return;
    // the slots of [rot_gen, upto) still hold phases [rot_gen - cap, upto - cap): every rotate kernel that reads them has been
    // enqueued before the last ev_used record (callers keep upto - consumed <= cap)
    if (upto - rot_gen > ring_cap) throw Error("rotator: look-ahead request beyond the phase ring");   // (callers ask for <= cap / 2 at a time)
    if (used_pending) { RR_HIP(hipStreamWaitEvent(rot_stream, ev_used, 0)); used_pending = false; }
    // (phases the host generator delivered meanwhile: walk the device chain up to rot_gen without storing)
    if (dphase_at < rot_gen) launch_rotor_replay(d_phase.p, stx, sty, d_ring.p, 0, (long)(ring_cap - 1), (long)(rot_gen - dphase_at), 0, rot_stream);
    launch_rotor_replay(d_phase.p, stx, sty, d_ring.p, (long)rot_gen, (long)(ring_cap - 1), 0, (long)(upto - rot_gen), rot_stream);
    RR_HIP(hipEventRecord(ev_gen, rot_stream));
    rot_gen = dphase_at = upto;
}

// Host-ring ranges whose copy to the device has completed go back to the generator.
void FirC32::rotor_reap(bool wait_oldest) {
    while (!copies.empty()) {
        hipError_t e = wait_oldest ? hipEventSynchronize(copies.front().ev) : hipEventQuery(copies.front().ev);
        wait_oldest = false;
        if (e == hipErrorNotReady) break;
        if (e != hipSuccess) throw Error(std::string("rotator: copy event: ") + hipGetErrorString(e));
        hrot->release(copies.front().upto - hrot_base);
        free_events.push_back(copies.front().ev);
        copies.erase(copies.begin());
    }
}

// Phases [rot_gen, upto) from the host generator's pinned ring into d_ring, on rot_stream.  block = false: only if they
// are all there already (look-ahead).
bool FirC32::rotor_fetch(size_t upto, bool block) {
    if (upto <= rot_gen) return true;
    if (upto - rot_gen > ring_cap) throw Error("rotator: look-ahead request beyond the phase ring");
    rotor_reap(false);
    while (hrot->wait_for(upto - hrot_base, block ? 20 : 0) < upto - hrot_base) {
        if (!block) return false;
        // the generator stalls on a full ring only while copies of its oldest phases are still in flight
        if (hrot->gen.load() >= hrot->tail.load() + hrot->cap) rotor_reap(true);
    }
    if (used_pending) { RR_HIP(hipStreamWaitEvent(rot_stream, ev_used, 0)); used_pending = false; }
    for (size_t i = rot_gen; i < upto;) {
        const size_t hs = (i - hrot_base) & hrot->mask, ds = i & (ring_cap - 1);
        const size_t n = std::min({upto - i, hrot->cap - hs, ring_cap - ds});
        RR_HIP(hipMemcpyAsync(d_ring.p + ds, hrot->ring + hs, n * sizeof(cf), hipMemcpyHostToDevice, rot_stream));
        i += n;
    }
    RR_HIP(hipEventRecord(ev_gen, rot_stream));
    hipEvent_t ev;
    if (free_events.empty()) RR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else { ev = free_events.back(); free_events.pop_back(); }
    RR_HIP(hipEventRecord(ev, rot_stream));
    copies.push_back({ev, upto});
    rot_gen = upto;
    return true;
}

void FirC32::rotate_output(cf* out, size_t out_n, hipStream_t s) {
    if (!rot_on || out_n == 0) return;
    if (rot_mode == RR_ROT_MODEL) {
        launch_rotate_model(out, (long)out_n, ph0x, ph0y, stx, sty, (long)n_rot, s);
        n_rot += out_n;
        return;
    }
    if (!rot_stream) {
        RR_HIP(hipStreamCreateWithFlags(&rot_stream, hipStreamNonBlocking));
        RR_HIP(hipEventCreateWithFlags(&ev_gen, hipEventDisableTiming));
        RR_HIP(hipEventCreateWithFlags(&ev_used, hipEventDisableTiming));
        // ring: four windows of the first call's size, 2^16 .. 2^22 phases (0.5 .. 32 MB); larger calls go through in halves
        ring_cap = (size_t)1 << 16;
        while (ring_cap < 4 * out_n && ring_cap < ((size_t)1 << 22)) ring_cap <<= 1;
        d_ring.reserve(ring_cap);
        d_phase.reserve(1);
        const cf p0 = mkcf(ph0x, ph0y);
        RR_HIP(hipMemcpyAsync(d_phase.p, &p0, sizeof(cf), hipMemcpyHostToDevice, rot_stream));
        RR_HIP(hipStreamSynchronize(rot_stream));                    // (p0 is a stack variable; once per block)
        dphase_at = 0;
        rot_gen = n_rot;                                             // (outputs rotated by the model so far are skipped below)
    }
    if (rot_gen < n_rot) rot_gen = n_rot;
    // REPLAY -> MODEL -> REPLAY, or REPLAY chosen after outputs were produced: nothing below n_rot is needed any more — the
    // device chain skips to rot_gen without storing (rotor_generate), a host generator is started AT the device chain's phase.
    if (hrot && rot_mode == RR_ROT_REPLAY_DEVICE) {                  // (forced back: drop the thread; the device chain catches up)
        RR_HIP(hipStreamSynchronize(rot_stream));
        for (auto& c : copies) free_events.push_back(c.ev);
        copies.clear();
        hrot.reset();
    }
    if (!hrot && rot_mode == RR_ROT_REPLAY) {
        // the starvation test: the look-ahead enqueued at the end of the previous call is what this call's phases come from
        if (lookahead_pending && hipEventQuery(ev_gen) == hipErrorNotReady) starving++;
        else starving = 0;
        (void)hipGetLastError();
        if (starving >= 3) rotor_start_host();
    }
    if (!hrot && rot_mode == RR_ROT_REPLAY_HOST) rotor_start_host();
    const bool host = hrot != nullptr;
    if (host) {
        // (only once no copy out of the host ring is in flight: a slot handed back early could be overwritten under a copy)
        rotor_reap(false);
        if (copies.empty() && rot_gen > hrot_base) hrot->release(rot_gen - hrot_base);
    }
    // one event remembers the ring's readers: a call on another HIP stream first waits for the previous stream's rotate
    // kernels, so that the record below still covers every reader enqueued so far
    if (used_pending && used_stream != s) RR_HIP(hipStreamWaitEvent(s, ev_used, 0));
    used_stream = s;
    const size_t half = ring_cap / 2;
    for (size_t done = 0; done < out_n;) {
        const size_t chunk = std::min(out_n - done, half);
        if (host) rotor_fetch(n_rot + done + chunk, true);           // (no-op when the look-ahead already covers it)
        else rotor_generate(n_rot + done + chunk);
        RR_HIP(hipStreamWaitEvent(s, ev_gen, 0));
        launch_rotate_table(out + done, (long)chunk, d_ring.p, (long)((n_rot + done) & (ring_cap - 1)), (long)(ring_cap - 1), s);
        RR_HIP(hipEventRecord(ev_used, s));
        used_pending = true;
        done += chunk;
    }
    n_rot += out_n;
    // look-ahead: the next window of the same size (host mode: as far as the generator has got, without waiting)
    lookahead_pending = false;
    if (host) {
        const size_t want = n_rot + std::min(out_n, half);
        const size_t got = (size_t)std::min<uint64_t>(want, hrot->gen.load(std::memory_order_acquire) + hrot_base);
        if (got > rot_gen) rotor_fetch(got, false);
    } else {
        const size_t before = rot_gen;
        rotor_generate(n_rot + std::min(out_n, half));
        lookahead_pending = rot_gen > before;
    }
}

// The host generator takes over where the device chain stands: everything enqueued on rot_stream is waited for (once per
// block), the carried phase comes back over PCIe (8 bytes) and the thread walks on from there.
void FirC32::rotor_start_host() {
    RR_HIP(hipStreamSynchronize(rot_stream));
    if (dphase_at < rot_gen) {                                       // (a model interlude: bring the device chain up to rot_gen first)
        launch_rotor_replay(d_phase.p, stx, sty, d_ring.p, 0, (long)(ring_cap - 1), (long)(rot_gen - dphase_at), 0, rot_stream);
        dphase_at = rot_gen;
        RR_HIP(hipStreamSynchronize(rot_stream));
    }
    cf p{};
    RR_HIP(hipMemcpy(&p, d_phase.p, sizeof(cf), hipMemcpyDeviceToHost));
    hrot_base = dphase_at;                                           // == rot_gen: phases below it are in d_ring already
    hrot.reset(new HostRotor(p.x, p.y, stx, sty, 2 * ring_cap));
    starving = 0;
}

// ---- Hilbert -> FirFilter<Complex> as one composite decimating FIR ------------------------------------------
HilbertFir::HilbertFir(size_t hilbert_ntaps, int window, float parm, const rr_c32* taps, size_t ntaps, size_t deci,
                       bool translate, float samp_rate, float freq)
    : Block("Hilbert>FirFilter<Complex>", 4, 8), hn(hilbert_ntaps) {
    if (!(hn > 1 && (hn & 1) == 1)) throw Error("hilbert filter len must be odd and greater than 1");  // hilbert.rs:44-47
    if (hn > 0x3fffff) throw Error("Hilbert: too many taps");
    std::vector<float> win, ht;
    if (!make_window(window, parm, hn, win)) throw Error("Hilbert: unknown window type");
    hilbert_taps(win.data(), hn, ht);                                           // hilbert.rs:48
    fir.reset(new FirC32(taps, ntaps, deci, translate, samp_rate, freq, false)); // validates, pre-rotates, rotator
    // c[j] = delta[j - hn/2] + i * rev_h[j], rev_h = reversed Hilbert taps (Fir::new, fir.rs:160)
    std::vector<std::complex<double>> c(hn), G(ntaps + hn - 1);
    for (size_t j = 0; j < hn; j++) c[j] = {j == hn / 2 ? 1.0 : 0.0, (double)ht[hn - 1 - j]};
    for (size_t k = 0; k < ntaps; k++) {
        const std::complex<double> r(fir->h_taps[ntaps - 1 - k].real(), fir->h_taps[ntaps - 1 - k].imag());
        for (size_t j = 0; j < hn; j++) G[k + j] += r * c[j];
    }
    plG.L = (int)G.size(); plG.d = (int)deci; plG.complex_taps = true; plG.cfg = build_opts().fir_cfg;
    std::vector<cf> rev(G.size()), tp;
    for (size_t n = 0; n < G.size(); n++) rev[n] = mkcf((float)G[n].real(), (float)G[n].imag());
    build_poly(rev, plG.d, plG.qpad, tp);
    d_revG.upload(rev.data(), rev.size(), stream);
    d_tpG.upload(tp.data(), tp.size(), stream);
    // deci 4 / 8 / 16: real-stream tiles with the pruned inverse; taps in caller order t[k] = G[Lg - 1 - k]
    // (tools/prune_probe.py: 65 (*) 255 taps /8 0.219 -> 0.191 ms per 1e8 real samples, /4 0.37 -> 0.23, /16 a tie)
    const bool have_split = prune_split(deci, prune_D, prune_sub);         // (deci = D * sub: FirC32)
    const size_t g_per_phase = have_split ? G.size() / prune_D : 0;
    // (/16: 65 * 255 taps at 1.28e8 samples 202 us pruned against 376 us direct, tools/prune_window_probe2.py)
    const bool prune_default = have_split && (prune_sub > 1 || (prune_D == 4 ? g_per_phase >= 8 : prune_D == 8 ? g_per_phase >= 24 : g_per_phase >= 8));
    if (build_opts().fir_path != RR_PATH_DIRECT && (build_opts().fir_prune ? build_opts().fir_prune > 0 : prune_default) && have_split) {
        std::vector<std::complex<double>> td(G.size());
        for (size_t k = 0; k < G.size(); k++) td[k] = G[G.size() - 1 - k];
        prune.reset(new PruneTables());
        if (!prune->build(td, prune_D, true, stream)) prune.reset();
    }
    for (auto& h : hist) {                                                      // hilbert.rs:55 — hn zeros
        h.reserve(hn);
        RR_HIP(hipMemsetAsync(h.p, 0, hn * sizeof(float), stream));
    }
    RR_HIP(hipStreamSynchronize(stream));
    // Two stages or one composite?  Fitted on MI355X (tools/misc_cliff_probe.py, ms per 1e8 real samples): the composite
    // direct form 0.1 + 0.0039 x (ntaps + hn - 1) / deci (255 taps: /1 1.42, /2 0.70, /5 0.40; 1000 taps /1 3.98; no tile at
    // all for 2467 taps /32: 114 ms on the one-thread-per-output fallback); Hilbert 0.23 + the FirFilter's own kernels 0.30
    // (up to ~1200 taps) / 0.45 (~3000) / 0.85 (beyond).  rr_build_opts.fir_path = direct keeps the composite.
    if (!prune && build_opts().fir_path != RR_PATH_DIRECT) {
        const double est_direct = fir_direct_has_tile(plG, sizeof(float), sizeof(cf)) ? 0.1 + 0.0039 * (double)G.size() / (double)deci : 1e9;
        const double est_two = 0.23 + (ntaps <= 1200 ? 0.30 : ntaps <= 3000 ? 0.45 : 0.85);
        two_stage = est_two < est_direct;
    }
    if (two_stage) {
        hil.reset(new Hilbert(hn, window, parm));
        std::vector<rr_c32> pt(ntaps);
        for (size_t k = 0; k < ntaps; k++) pt[k] = rr_c32{fir->h_taps[k].real(), fir->h_taps[k].imag()};
        fir2.reset(new FirC32(pt.data(), ntaps, deci, false, 0.0f, 0.0f, true));
    }
}

// One input sample = one analytic sample, so the FirFilter bookkeeping (fir.rs:496-549) applies to the
// real input window unchanged; the Hilbert window of the first analytic sample reaches hn samples back.
int HilbertFir::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                         size_t* produced, size_t* need, hipStream_t s) {
    const size_t L = fir->pl.L, d = fir->pl.d;
    *consumed = *produced = *need = 0;
    if (in_len < L + d - 1) { *need = L + d - 1; return RR_WAIT_SRC; }
    size_t n = d * ((in_len - L + 1) / d);
    if (out_cap < 1) { *need = 1; return RR_WAIT_DST; }
    n = std::min(n, out_cap * d);
    const size_t out_n = n / d;
    VSrc<float> src{hist[cur].p, (long)hn, static_cast<const float*>(in), (long)in_len};
    prof_begin(s);
    // (as FirC32::work_dev: a batch is D / 2 tiles of 2 (F - Lg + 1) real samples; below ~2100 batches — 30 M samples at
    //  65 * 255 taps / 8 — the direct form is faster: 1 M samples 31.7 -> 10.1 us, 8 M 34.0 -> 22.7 us)
    bool use_prune = prune != nullptr;
    if (use_prune && fir->window_aware) {
        const size_t pd = prune_D, F = (size_t)1 << prune->log2f, per_batch = 2 * (F - (size_t)plG.L + 1) * (pd / 2);
        use_prune = n >= chip_units(pd == 4 ? 1500 : pd == 8 ? 1500 : 350) * per_batch;   // (tools/prune_window_probe2.py)
    }
    // (the hn samples before the new window become the next call's history: written by the tile kernel itself)
    const CarryOut carry{hist[cur ^ 1].p, (long)n, (long)hn};
    if (use_prune) launch_fftfilt_prune_real(prune->log2f, src, static_cast<cf*>(out), (long)out_n, plG.L, prune->d_tw.p, prune->d_h2.p, prune->d_h2b.p, prune->d_twb.p, s, carry, (int)prune_sub, nanfixG());
    else if (two_stage && (n >= 16384 || !fir_direct_has_tile(plG, sizeof(float), sizeof(cf)))) {
        // a[k] = (iv[k + hn/2], sum_j rev_h[j] iv[k + j]) over the virtual stream iv = hist ++ window (hilbert.rs:113-116), then
        // y[m] = sum_k rev[k] a[m d + k]: outputs [m0, m1) take a[m0 d, m1 d + L - 1).  In chunks, so that the analytic buffer
        // stays at 128 MB whatever the window.
        const size_t CH = (size_t)1 << 24;
        // (the FirFilter stage produces floor((len - L + 1) / d) outputs, fir.rs:503-510: k outputs take k d + L - 1 samples)
        // (CH <= L, or d > CH - L: one output per chunk; the buffer holds what a chunk really takes — ADVICE r4: `CH - L`
        //  wrapped around and a chunk of per d + L - 1 > CH samples overran a buffer of CH)
        const size_t per = CH > L ? std::max<size_t>(1, (CH - L) / d) : 1;
        analytic.reserve(std::min(per, out_n) * d + L - 1 + 8);
        for (size_t m0 = 0; m0 < out_n; m0 += per) {
            const size_t m1 = std::min(out_n, m0 + per), k0 = m0 * d, na = (m1 - m0) * d + L - 1;
            VSrc<float> sa = src;
            if (k0 >= hn) sa = VSrc<float>{hist[cur].p, 0, static_cast<const float*>(in) + (k0 - hn), (long)(in_len - (k0 - hn))};
            else if (k0 != 0) throw Error("HilbertFir: chunk inside the history");   // (per >> hn: only the first chunk starts in it)
            if (!(hil->skip_ok && launch_hilbert_skip(hil->pl.L, hil->par, hil->Q, hil->d_hq.p, sa, analytic.p, (long)na, s, hil->nanfix())))
                launch_hilbert(hil->pl, hil->d_tp.p, hil->d_rev.p, sa, analytic.p, (long)na, s, hil->nanfix());
            size_t c2 = 0, p2 = 0, n2 = 0;
            fir2->work_dev(analytic.p, na, static_cast<cf*>(out) + m0, m1 - m0, &c2, &p2, &n2, s);
            if (p2 != m1 - m0) throw Error("HilbertFir: the FirFilter stage disagrees on the output count");
        }
    }
    else launch_fir_f32c(plG, d_tpG.p, d_revG.p, src, static_cast<cf*>(out), (long)out_n, s, nanfixG());
    prof_end(s);
    fir->rotate_output(static_cast<cf*>(out), out_n, s);
    if (!use_prune) launch_carry(src, carry, s);
    cur ^= 1;
    *consumed = n; *produced = out_n;
    return RR_AGAIN;
}

FirF32::FirF32(const float* taps, size_t ntaps, size_t deci) : Block("FirFilter<Float>", 4, 4) {
    if (ntaps == 0) throw Error("FirFilter: empty taps");
    if (deci == 0) throw Error("FirFilter: decimation 0");
    if (ntaps > 0x3fffffff || deci > 0x3fffffff) throw Error("FirFilter: taps/deci too large");
    pl.L = (int)ntaps; pl.d = (int)deci; pl.complex_taps = false; pl.cfg = build_opts().fir_cfg;
    std::vector<float> rev(ntaps), tp;
    for (size_t j = 0; j < ntaps; j++) rev[j] = taps[ntaps - 1 - j];
    build_poly(rev, pl.d, pl.qpad, tp);
    d_rev.upload(rev.data(), rev.size(), stream);
    d_tp.upload(tp.data(), tp.size(), stream);
    RR_HIP(hipStreamSynchronize(stream));
    // long filters on overlap-save tiles, two real segments per Complex tile (k_fftfilt_real): a flat ~0.148 ms per
    // 1e8 samples (8 B/sample -> 5.4 TB/s) against 0.15 + 0.002 ms per tap for the direct form
    // (tools/fir_float_probe.py: 65 taps 0.26 -> 0.147 ms, 463 taps 1.25 -> 0.17 ms, 2467 taps 5.96 -> 0.33 ms)
    const BuildOpts& bo = build_opts();
    const bool force_direct = bo.fir_path == RR_PATH_DIRECT, force_fft = bo.fir_path == RR_PATH_FFT;
    const bool fits = ntaps <= 3584 && deci <= 4096;
    const bool wins = deci == 1 ? ntaps >= 24
                                : (ntaps >= 320 || ntaps / deci >= 40 || deci >= 10 ||   // (/10 on: 0.27-0.46 ms per 1e8 direct, 0.20 on the tiles)
                                   !fir_direct_has_tile(pl, sizeof(float), sizeof(float)));   // (see FirC32)
    const bool have_split = prune_split(deci, prune_D, prune_sub);         // (deci = D * sub: FirC32)
    const size_t per_phase = have_split ? ntaps / prune_D : 0;
    const bool prune_default = have_split && (prune_sub > 1 || (prune_D == 4 ? per_phase >= 8 : prune_D == 8 ? per_phase >= 16 : per_phase >= 4));
    const bool prune_wins = bo.fir_prune ? bo.fir_prune > 0 : prune_default;
    if (!force_direct && prune_wins && have_split) {
        std::vector<std::complex<double>> td(ntaps);
        for (size_t i = 0; i < ntaps; i++) td[i] = {(double)taps[i], 0.0};
        prune.reset(new PruneTables());
        if (!prune->build(td, prune_D, false, stream)) prune.reset();
    }
    window_aware = bo.fir_prune <= 0 && !force_fft;
    // (with the pruned inverse the tiles stay beside it as the small-window path of long filters, see work_dev)
    if (fits && !force_direct && (force_fft || wins || prune)) {
        std::vector<rr_c32> ct(ntaps);
        for (size_t i = 0; i < ntaps; i++) ct[i] = {taps[i], 0.0f};
        fftk.reset(new FftFilter(ct.data(), ntaps, false, 12, true));
        fftk->nanfix = nanfix();
    }
    if (!fftk && !prune && !force_direct && ntaps > 320) {   // (see blocks.hpp; shorter filters have direct-form tiles that fit)
        std::vector<rr_c32> ct(ntaps);
        for (size_t i = 0; i < ntaps; i++) ct[i] = {taps[i], 0.0f};
        wide.reset(new FirC32(ct.data(), ntaps, deci, false, 0.0f, 0.0f, true));
    }
}
FirF32::~FirF32() = default;
int FirF32::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                     size_t* produced, size_t* need, hipStream_t s) {
    const size_t L = pl.L, d = pl.d;
    *consumed = *produced = *need = 0;
    if (in_len < L + d - 1) { *need = L + d - 1; return RR_WAIT_SRC; }
    size_t n = d * ((in_len - L + 1) / d);
    if (out_cap < 1) { *need = 1; return RR_WAIT_DST; }
    n = std::min(n, out_cap * d);
    VSrc<float> src{nullptr, 0, static_cast<const float*>(in), (long)in_len};
    prof_begin(s);
    // as FirC32::work_dev: a batch of the pruned inverse is D tiles of two real segments; below ~1600 batches (46 M samples at
    // 255 taps / 8) the direct form or the plain tiles finish a call sooner (1 M samples: 33.9 -> 6.7 us, 8 M: 38.5 -> 14.6)
    bool use_prune = prune != nullptr;
    if (use_prune && window_aware) {
        const size_t pd = prune_D, F = (size_t)1 << prune->log2f, per_batch = 2 * (F - L + 1) * pd;
        use_prune = n >= chip_units(pd == 4 ? 2000 : pd == 8 ? 1700 : 370) * per_batch;   // (tools/prune_window_probe2.py)
    }
    const bool small_direct = prune && !use_prune && (!fftk || (L <= 320 && fir_direct_has_tile(pl, sizeof(float), sizeof(float))));
    if (wide) {
        // outputs [m0, m1) take samples [m0 d, m1 d + L - 1) (fir.rs:503-510); chunks keep the work buffers at 128 + 128 / d MB
        const size_t out_n = n / d, CH = (size_t)1 << 24, per = CH > L ? std::max<size_t>(1, (CH - L) / d) : 1;   // (see HilbertFir)
        wide_in.reserve(std::min(per, out_n) * d + L - 1 + 8);
        wide_out.reserve(std::min(per, out_n) + 8);
        for (size_t m0 = 0; m0 < out_n; m0 += per) {
            const size_t m1 = std::min(out_n, m0 + per), na = (m1 - m0) * d + L - 1;
            launch_f32_to_c32(static_cast<const float*>(in) + m0 * d, wide_in.p, (long)na, s);
            size_t c2 = 0, p2 = 0, n2 = 0;
            wide->work_dev(wide_in.p, na, wide_out.p, m1 - m0, &c2, &p2, &n2, s);
            if (p2 != m1 - m0) throw Error("FirFilter<Float>: the Complex stage disagrees on the output count");
            launch_c32_re(wide_out.p, static_cast<float*>(out) + m0, (long)(m1 - m0), s);
        }
    }
    else if (use_prune) launch_fftfilt_prune_f32(prune->log2f, src, static_cast<float*>(out), (long)(n / d), (int)L, prune->d_tw.p, prune->d_h2.p, prune->d_twb.p, s, (int)prune_sub, nanfix());
    else if (small_direct) launch_fir_f32(pl, d_tp.p, d_rev.p, src, static_cast<float*>(out), (long)(n / d), s, nanfix());
    else if (fftk) fftk->filter_real(src, static_cast<float*>(out), (long)(n / d), (int)d, s);
    else launch_fir_f32(pl, d_tp.p, d_rev.p, src, static_cast<float*>(out), (long)(n / d), s, nanfix());
    prof_end(s);
    *consumed = n; *produced = n / d;
    return RR_AGAIN;
}

// ---- FftFilter (fft_filter.rs:131-181, 210-355) --------------------------------------------------
static size_t calc_fft_size(size_t from) {   // fft_filter.rs:36-42
    size_t n = 1;
    while (n < from) n <<= 1;
    return 2 * n;
}

// Host f64 radix-2 FFT for the one-time taps transform (setup only).
static void fft64(std::vector<std::complex<double>>& a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                const std::complex<double> w = std::polar(1.0, -2.0 * 3.14159265358979323846 * (double)k / (double)len);
                const auto u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

template <int LOG2F> static void fill_hpos(const std::vector<std::complex<double>>& H, std::vector<cf>& hpos) {
    const int F = 1 << LOG2F;
    hpos.resize(F);
    for (int p = 0; p < F; p++) {
        const auto h = H[bin_of_pos<LOG2F>(p)];
        hpos[p] = mkcf((float)h.real(), (float)h.imag());
    }
}

// H = FFT(taps || 0) / F (fft_filter.rs:151-162) in f64, rounded once, in digit-reversed position order
static void compute_hpos(const rr_c32* taps, size_t ntaps, int log2f, std::vector<cf>& hpos) {
    const size_t F = (size_t)1 << log2f;
    std::vector<std::complex<double>> H(F, 0.0);
    for (size_t i = 0; i < ntaps; i++) H[i] = {taps[i].re, taps[i].im};
    fft64(H);
    for (auto& h : H) h /= (double)F;
    switch (log2f) {
    case 10: fill_hpos<10>(H, hpos); break;
    case 11: fill_hpos<11>(H, hpos); break;
    case 12: fill_hpos<12>(H, hpos); break;
    case 13: fill_hpos<13>(H, hpos); break;
    default: fill_hpos<14>(H, hpos); break;
    }
}

bool PruneTables::build(const std::vector<std::complex<double>>& t, size_t d, bool split, hipStream_t s) {
    const size_t L = t.size();
    log2f = d <= 16 ? prune_log2f_for_deci((int)d) : 0;
    if (!log2f || L == 0) return false;
    const size_t F = (size_t)1 << log2f, D = d;
    // keep at least half of every tile — a quarter for the real-stream x Complex-taps form (the fused Hilbert -> FirFilter),
    // whose alternative is two stages at 0.54 ms per 1e8 samples against ~0.14 / (share of the tile that is output) here
    if (L > (split ? 3 * F / 4 : F / 2 + 1)) return false;
    const size_t c = (L - 1) % D;
    const double two_pi = 2.0 * 3.14159265358979323846;
    auto table = [&](bool imag_part, DevBuf<cf>& dst) {
        std::vector<std::complex<double>> H(F, 0.0);
        for (size_t i = 0; i < L; i++) H[i] = split ? std::complex<double>(imag_part ? t[i].imag() : t[i].real(), 0.0) : t[i];
        fft64(H);
        for (size_t k = 0; k < F; k++) {                              // fold w_D^(-c k3) w_16D^(-c k2), and 1/F
            const size_t k2 = (k / 16) % 16, k3 = k / 256;
            const double a = two_pi * (double)c * ((double)k3 / (double)D + (double)k2 / (double)(16 * D));
            H[k] *= std::polar(1.0 / (double)F, a);
        }
        std::vector<cf> hpos;
        switch (log2f) {
        case 10: fill_hpos<10>(H, hpos); break;
        case 11: fill_hpos<11>(H, hpos); break;
        default: fill_hpos<12>(H, hpos); break;
        }
        dst.upload(hpos.data(), F, s);
    };
    table(false, d_h2);
    if (split) table(true, d_h2b);
    std::vector<cf> tw(F), twb(256);
    for (size_t k = 0; k < F; k++) {
        const double a = -two_pi * (double)k / (double)F;
        tw[k] = mkcf((float)std::cos(a), (float)std::sin(a));
    }
    for (size_t n2 = 0; n2 < 16; n2++)
        for (size_t k1 = 0; k1 < 16; k1++) {
            const double a = two_pi * (double)(k1 * (n2 * D + c)) / (double)F;
            twb[n2 * 16 + k1] = mkcf((float)std::cos(a), (float)std::sin(a));
        }
    d_tw.upload(tw.data(), F, s);
    d_twb.upload(twb.data(), 256, s);
    RR_HIP(hipStreamSynchronize(s));
    return true;
}

std::vector<rr_c32> FftFilter::composite(const rr_c32* t1, size_t n1, const rr_c32* t2, size_t n2) {
    if (n1 == 0 || n2 == 0) throw Error("FirFilter>FftFilter: empty taps");
    if (n1 > 0x3fffff || n2 > 0x3fffff) throw Error("FirFilter>FftFilter: too many taps");
    std::vector<std::complex<double>> g(n1 + n2 - 1);
    for (size_t a = 0; a < n1; a++)
        for (size_t b = 0; b < n2; b++)
            g[a + b] += std::complex<double>(t1[a].re, t1[a].im) * std::complex<double>(t2[b].re, t2[b].im);
    std::vector<rr_c32> out(g.size());
    for (size_t i = 0; i < g.size(); i++) out[i] = rr_c32{(float)g[i].real(), (float)g[i].imag()};
    return out;
}
void FftFilter::set_stage_taps(const rr_c32* t1, size_t n1, const rr_c32* t2, size_t n2) {
    if (n1 != front + 1 || n1 + n2 - 1 != L) throw Error("FirFilter>FftFilter: stage taps do not match the composite");
    d_t1.upload(reinterpret_cast<const cf*>(t1), n1, stream);
    d_t2.upload(reinterpret_cast<const cf*>(t2), n2, stream);
}

bool PolyTables::build(const rr_c32* taps, size_t C, size_t L, size_t D, bool multi, hipStream_t s) {
    if (!fm_poly_supported(1, (long)D, (int)std::min<size_t>(L, 1 << 24), multi)) return false;
    const size_t F = 1024;
    std::vector<cf> h(C * D * F), tw(F);
    std::vector<std::complex<double>> H(F);
    for (size_t c = 0; c < C; c++)
        for (size_t p = 0; p < D; p++) {
            std::fill(H.begin(), H.end(), std::complex<double>(0.0, 0.0));
            for (size_t j = 0; D * j + p < L; j++) H[j] = {taps[c * L + D * j + p].re, taps[c * L + D * j + p].im};
            fft64(H);
            cf* dst = h.data() + (c * D + p) * F;
            for (int j = 0; j < 16; j++)
                for (int lane = 0; lane < 64; lane++) {
                    const auto v = H[(size_t)fm_poly_bin(j, lane)] / (double)F;
                    // a lane's registers 2 jj, 2 jj + 1 side by side (one 16-byte load per pair)
                    dst[((j / 2) * 64 + lane) * 2 + (j & 1)] = mkcf((float)v.real(), (float)v.imag());
                }
        }
    for (size_t k = 0; k < F; k++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)F;
        tw[k] = mkcf((float)std::cos(a), (float)std::sin(a));
    }
    d_h.upload(h.data(), h.size(), s);
    d_tw.upload(tw.data(), F, s);
    return true;
}

FftFilter::FftFilter(const rr_c32* taps_in, size_t ntaps, bool for_chain, int max_log2f, bool real, size_t front_, bool tiles_only)
    : Block(front_ ? "FirFilter>FftFilter" : "FftFilter", real ? 4 : 8, real ? 4 : 8), real_stream(real), front(front_) {
    if (ntaps == 0) throw Error("FftFilter: empty taps");            // fft_filter.rs:146
    if (front >= ntaps || (front && real)) throw Error("FftFilter: bad front-filter length");
    std::vector<rr_c32> real_taps;
    const rr_c32* taps = taps_in;
    if (real) {
        if (max_log2f > 12) max_log2f = 12;
        real_taps.assign(taps_in, taps_in + ntaps);
        for (auto& c : real_taps) c.im = 0.0f;
        taps = real_taps.data();
    }
    L = ntaps;
    hist = L - 1 - front;
    fft_size = calc_fft_size(ntaps - front);                          // fft_filter.rs:261 (of the FftFilter stage)
    nsamples = fft_size - (ntaps - front);                            // fft_filter.rs:262
    // GPU tile: overlap-save with any F > L - 1 (S' = F - L + 1 new outputs per F-point transform; results do
    // not depend on F beyond f32 rounding).  F is the one that minimises measured tile cost / S'
    // (tools/taps_sweep.py: relative cost of one tile of 1024 .. 16384 points on MI355X; the 8192- and
    // 16384-point tiles are k_fftfilt_split, 2 / 4 sub-transforms of 4096 points).
    // cost of one tile ~ a + b * S' (transform + stores), fitted to tools/taps_sweep.py on MI355X
    static const double cost_a[5] = {200.0, 300.0, 800.0, 2900.0, 7900.0};
    static const double cost_b[5] = {0.10, 0.17, 0.16, 0.11, 0.17};
    (void)for_chain;
    double best = 0.0;
    log2f = -1;
    for (int lg = 10; lg <= max_log2f; lg++) {
        const size_t Fc = (size_t)1 << lg;
        if (Fc < L + 1) continue;                                     // need S' >= 2
        const double Sc = (double)(Fc - L + 1);
        const double c = (cost_a[lg - 10] + cost_b[lg - 10] * Sc) / Sc;
        if (log2f < 0 || c < best) { best = c; log2f = lg; }
    }
    if (log2f < 0) log2f = 15;                                        // -> the any-size path (or refused) below
    // Towards 16383 taps the largest tile is all overlap (16385 - L samples per 16384-point tile: 16383 taps = 2 samples,
    // tools/misc_cliff_probe.py: 1795 ms per 5e7 samples against 3.7 ms for 16385 taps on the any-size frames).  The frames
    // cost ~7.4 units of the tile model: from 15293 taps on they are cheaper.
    if (log2f == 14 && !real && !for_chain && !tiles_only && max_log2f >= 14 && 7900.0 / (double)(16385 - L) + 0.17 > 7.4) log2f = 15;
    if (const int v = build_opts().fft_log2f) {            // rr_build_opts: force a tile size
        if (v >= 10 && v <= max_log2f && ((size_t)1 << v) >= L + 1) log2f = v;
    }
    if (!fftfilt_supported(log2f)) {
        // more than 16383 taps (the reference has no limit, fft_filter.rs:36-42): overlap-save frames of M = 2^m >= 2 L
        // points, every frame one any-size transform (four-step beyond 16384 points) — plain HBM-streaming passes
        if (real || for_chain || max_log2f < 14 || L > ((size_t)1 << 19))
            throw Error("FftFilter: more than " + std::to_string(((size_t)1 << max_log2f) - 1) +
                        " taps is not supported by this block's tile kernels");
        bigM = 1;
        while (bigM < 2 * L) bigM <<= 1;
        log2f = 0;
        while (((size_t)1 << log2f) < bigM) log2f++;
        big.reset(new AnyFft(bigM, stream));
        std::vector<std::complex<double>> H(bigM, 0.0);
        for (size_t i = 0; i < ntaps; i++) H[i] = {taps[i].re, taps[i].im};
        fft64(H);
        std::vector<cf> hb(bigM);
        for (size_t k = 0; k < bigM; k++) hb[k] = mkcf((float)(H[k].real() / (double)bigM), (float)(H[k].imag() / (double)bigM));
        d_hbig.upload(hb.data(), bigM, stream);
        const size_t pcap = hist + nsamples + 1;
        for (auto& p : prefix) {
            p.reserve(pcap);
            RR_HIP(hipMemsetAsync(p.p, 0, pcap * sizeof(cf), stream));
        }
        RR_HIP(hipStreamSynchronize(stream));
        return;
    }
    const size_t F = (size_t)1 << log2f;
    std::vector<cf> hpos, tw(F);
    compute_hpos(taps, ntaps, log2f, hpos);
    for (size_t k = 0; k < F; k++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)F;
        tw[k] = mkcf((float)std::cos(a), (float)std::sin(a));
    }
    d_hpos.upload(hpos.data(), F, stream);
    d_tw.upload(tw.data(), F, stream);
    if (log2f >= 13 && !build_opts().fft_no_split) {      // split-tile tables (thread-major order, see kernels.hpp)
        nsub = 1 << (log2f - 12);
        const size_t M = 4096;
        std::vector<std::complex<double>> H(F, 0.0);
        for (size_t i = 0; i < ntaps; i++) H[i] = {taps[i].re, taps[i].im};
        fft64(H);
        std::vector<cf> hs((size_t)nsub * M), wk(M), tw4(M);
        for (size_t pp = 0; pp < M; pp++) {             // hs[r][p] = H[nsub bin(p) + r] / F
            const size_t k = (size_t)fftfilt_split_bin((int)pp);
            for (int r = 0; r < nsub; r++) {
                const auto h = H[(size_t)nsub * k + r] / (double)F;
                hs[(size_t)r * M + pp] = mkcf((float)h.real(), (float)h.imag());
            }
        }
        for (int t = 0; t < 256; t++) {                 // wk[t] = w_F^t (the kernels build w_F^(n 256 + t) from it)
            const double a = -2.0 * 3.14159265358979323846 * (double)t / (double)F;
            wk[(size_t)t] = mkcf((float)std::cos(a), (float)std::sin(a));
        }
        for (size_t k = 0; k < M; k++) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)M;
            tw4[k] = mkcf((float)std::cos(a), (float)std::sin(a));
        }
        d_hs.upload(hs.data(), hs.size(), stream);
        d_wk.upload(wk.data(), wk.size(), stream);
        d_tw4096.upload(tw4.data(), tw4.size(), stream);
        // Small windows: a split tile is one workgroup of 512 threads per 8192 / 16384 points, two per CU — a ring-sized
        // window (512,000 samples = 90 tiles at 2467 taps) leaves most of the chip idle.  The plain 4096-point tile is less
        // efficient per sample but four to ten times as many workgroups (tools/call_overhead.py, 2467 taps: 512 k samples
        // 26.2 -> 16.8 us, 2 M 30.3 -> 25.3, 8 M 57.4 -> 64.3): kept beside the split tables, chosen per call (results do
        // not depend on the tile beyond f32 rounding, the carried state is tile-independent).
        if (!real && L + 512 <= 4096 && build_opts().fft_log2f == 0) {
            std::vector<cf> hp, twa(4096);
            compute_hpos(taps, ntaps, 12, hp);
            for (size_t k = 0; k < 4096; k++) {
                const double a = -2.0 * 3.14159265358979323846 * (double)k / 4096.0;
                twa[k] = mkcf((float)std::cos(a), (float)std::sin(a));
            }
            d_hpos_alt.upload(hp.data(), 4096, stream);
            d_tw_alt.upload(twa.data(), 4096, stream);
            alt_log2f = 12;
        }
    }
    // prefix = [L-1 history samples][pending < nsamples]; zero history at stream start (A.4)
    const size_t pcap = hist + nsamples + 1;
    for (auto& p : prefix) {
        p.reserve(pcap);
        RR_HIP(hipMemsetAsync(p.p, 0, pcap * sizeof(cf), stream));
    }
    RR_HIP(hipStreamSynchronize(stream));
}

void FftFilter::ref_blocks_on(const rr_c32* taps) {
    if (build_opts().fft_nonfinite_tiles == 1) return;
    // 3: the pass inside the tile kernel's tail (nan_fix.hpp rb_finish, round 6): ONE launch per work(), but every workgroup
    // must release its outputs to the device before the last one may overwrite some of them — on this part that is a write-back
    // of the XCD's L2 per workgroup.  Measured on one box, FftFilter 401 taps (tools/ab_nonfinite_pass.sh): 1e8 samples 0.360 ms against
    // 0.332 with the pass as its own launch (0.325 without any pass); a 512,000-sample call 22.2 us against 17.1 (12.9).  The
    // launch it saves costs less than the releases it needs: opt-in, for the record.
    rb_in_kernel = build_opts().fft_nonfinite_tiles == 3;
    if (real_stream) {
        std::vector<float> r(L);
        for (size_t j = 0; j < L; j++) r[j] = taps[L - 1 - j].re;
        d_rev.reserve((L + 1) / 2);
        stage_upload_sync(d_rev.p, r.data(), L * sizeof(float), stream);
    } else {
        std::vector<cf> r(L);
        for (size_t j = 0; j < L; j++) r[j] = mkcf(taps[L - 1 - j].re, taps[L - 1 - j].im);
        d_rev.upload(r.data(), L, stream);
    }
    const int none[2] = {-1, -1};
    d_tail.upload(none, 2, stream);
    const int zero[4] = {0, 0, 0, 0};
    d_rb_work.upload(zero, 4, stream);
    RR_HIP(hipStreamSynchronize(stream));
    ref_blocks = true;
}
template <class T> void FftFilter::ref_blocks_pass(VSrc<T> src, T* out, long n_out, hipStream_t s, bool force0) {
    if (!ref_blocks || n_out <= 0) return;
    if constexpr (std::is_same<T, cf>::value) {
        if (host_out) { deferred.on = true; deferred.force0 = force0; deferred.src = src; deferred.out = out; deferred.n_out = n_out; return; }
    }
    tail_host_known = false;
    launch_ref_blocks_nonfinite(src, out, n_out, (long)nsamples, probe_stride, (long)hist, (int)L, (int)front,
                                reinterpret_cast<const T*>(d_rev.p), d_tail.p, seq, force0, s);
    seq++;
}

void FftFilter::host_out_done(const void* out_host, size_t) {
    if (!deferred.on) return;
    deferred.on = false;
    const std::complex<float>* y = static_cast<const std::complex<float>*>(out_host);
    const long n = deferred.n_out, P = std::max<long>(1, probe_stride);
    auto bad = [&](long m) { return !(std::isfinite(y[m].real()) && std::isfinite(y[m].imag())); };
    bool hit = !tail_host_known || tail_host || deferred.force0;
    if (!hit) {                                                 // a tile that read a non-finite sample has NO finite output
        for (long m = 0; m < n; m += P) __builtin_prefetch(&y[m]);   // (the lines come from memory the device just wrote: all misses)
        hit = bad(n - 1);
        for (long m = 0; m < n && !hit; m += P) hit = bad(m);
    }
    if (!hit) { seq++; return; }                               // (a call without the pass: its tail slot keeps a stale number)
    launch_ref_blocks_nonfinite(deferred.src, static_cast<cf*>(deferred.out), n, (long)nsamples, probe_stride, (long)hist, (int)L, (int)front,
                                d_rev.p, d_tail.p, seq, deferred.force0, stream);
    int t = -1;
    RR_HIP(hipMemcpyAsync(&t, d_tail.p + (seq & 1), sizeof t, hipMemcpyDeviceToHost, stream));
    RR_HIP(hipStreamSynchronize(stream));
    tail_host = t == seq; tail_host_known = true;
    seq++;
}

// rb_ok: this call may put the reference's blocks in place inside the tile kernel (nan_fix.hpp rb_finish) -> whether it did
// (the split tiles and the any-size frames have no such tail: the caller runs the pass behind them)
bool FftFilter::filter(VSrc<cf> src, cf* out, long n_out, hipStream_t s, CarryOut carry, bool rb_ok) {
    probe_stride = (long)(big ? bigM : (size_t)1 << (nsub && alt_log2f && alt_wins(n_out) ? alt_log2f : log2f)) - (long)L + 1;
    if (big) {
        // y = conj(FFT_M(conj(FFT_M(frame) H))) per overlap-save frame, in chunks of <= 2^24 elements of work space
        const long M = (long)bigM, S = M - (long)L + 1;
        const long nfr = (n_out + S - 1) / S;
        const long per = std::max<long>(1, ((long)1 << 24) / M);
        bframes.reserve((size_t)std::min(per, nfr) * M);
        bspec.reserve((size_t)std::min(per, nfr) * M);
        for (long f0 = 0; f0 < nfr; f0 += per) {
            const long nf = std::min(per, nfr - f0);
            launch_ols_gather(src, bframes.p, S, M, f0, nf, s);
            big->forward(bframes.p, bspec.p, nf, s);
            launch_mul_conj(bspec.p, d_hbig.p, M, nf, s);
            big->forward(bspec.p, bframes.p, nf, s);
            launch_ols_scatter(bframes.p, out, S, M, (long)L, f0, nf, n_out, s);
        }
        launch_carry(src, carry, s);
        return false;
    }
    const bool use_alt = nsub && alt_log2f && alt_wins(n_out);
    if (nsub && !use_alt) { launch_fftfilt_split(nsub, src, out, n_out, (int)L, d_tw4096.p, d_hs.p, d_wk.p, s, carry, nanfix); return false; }
    NanFix fx = nanfix;
    const bool rb = rb_ok && ref_blocks && rb_in_kernel && n_out > 0;
    if (rb) {
        const long P = probe_stride, ntiles = (n_out + P - 1) / P;
        const long cap = 4 * ntiles + 8;                             // (a flagged tile writes at most four records)
        d_rb_recs.reserve((size_t)(2 * cap));
        fx = NanFix{};
        fx.rev = d_rev.p; fx.L = (int)L; fx.d = 1; fx.kind = NANFIX_CC;
        fx.rb_work = d_rb_work.p; fx.rb_recs = d_rb_recs.p; fx.rb_cap = cap; fx.rb_S = (long)nsamples; fx.rb_hist = (long)hist;
        fx.rb_front = (int)front; fx.rb_seq = seq; fx.rb_tail = d_tail.p;
    }
    if (use_alt) launch_fftfilt_os(alt_log2f, src, out, n_out, (int)L, d_tw_alt.p, d_hpos_alt.p, s, carry, fx);
    else launch_fftfilt_os(log2f, src, out, n_out, (int)L, d_tw.p, d_hpos.p, s, carry, fx);
    return rb;
}

void FftFilter::filter_real(VSrc<float> src, float* out, long n_out, int d, hipStream_t s, CarryOut carry) {
    NanFix fx = nanfix; fx.d = d;
    probe_stride = ((long)1 << log2f) - (long)L + 1;
    launch_fftfilt_real(log2f, src, out, n_out, (int)L, d, d_tw.p, d_hpos.p, s, carry, fx);
}

int FftFilter::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                        size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    const size_t S = nsamples;
    if (S > out_cap) { *need = S; return RR_WAIT_DST; }               // fft_filter.rs:294-303
    if (front && in_len < front + 1) { *need = front + 1; return RR_WAIT_SRC; }   // the front FirFilter: fir.rs:498-501
    const size_t av = avail(in_len);   // front FIR: its last L1 - 1 samples stay in the caller's window (fir.rs:537)
    const size_t total = pend_len + av;
    const size_t k_in = total / S, k_out = out_cap / S;
    size_t k, new_pend;
    int st;
    if (k_in >= k_out) {               // output space runs out first: the loop stops before reading more
        k = k_out; *consumed = k * S - pend_len; new_pend = 0;
        st = RR_WAIT_DST; *need = S;
    } else {                           // input runs out: everything is taken into `buf` (:306-314)
        k = k_in; *consumed = av; new_pend = total - k * S;
        st = RR_WAIT_SRC; *need = S - new_pend + front;                // :323-326
    }
    const size_t n_out = k * S;
    const long plen = (long)(hist + pend_len);
    if (real_stream) {
        VSrc<float> rsrc{reinterpret_cast<const float*>(prefix[cur].p), plen, static_cast<const float*>(in), (long)in_len};
        // new carry = last `hist` samples before the first unprocessed one, then the unprocessed tail: written by the
        // filter kernel itself (any workgroup: it reads this call's stream and writes the OTHER prefix buffer)
        CarryOut carry;
        if (*consumed) carry = CarryOut{prefix[cur ^ 1].p, (long)n_out, (long)(hist + new_pend)};
        if (k) {
            prof_begin(s);
            filter_real(rsrc, static_cast<float*>(out), (long)n_out, 1, s, carry);
            prof_end(s);
            ref_blocks_pass(rsrc, static_cast<float*>(out), (long)n_out, s);
        } else {
            launch_carry(rsrc, carry, s);
        }
        if (*consumed) {
            cur ^= 1;
            pend_len = new_pend;
        }
        *produced = n_out;
        return st;
    }
    VSrc<cf> src{prefix[cur].p, plen, static_cast<const cf*>(in), (long)in_len};
    CarryOut carry;      // (as above)
    if (*consumed) carry = CarryOut{prefix[cur ^ 1].p, (long)n_out, (long)(hist + new_pend)};
    if (k) {
        const bool head = front && emitted == 0;
        prof_begin(s);
        // (the head fix below overwrites the first outputs of a possibly poisoned tile AFTER the tile kernel: that one call keeps the
        //  pass behind it)
        const bool in_kernel = filter(src, static_cast<cf*>(out), (long)n_out, s, carry, !head);
        prof_end(s);
        if (head) {   // head fix: FftFilter's zero history under the first L2 - 1 outputs (n_out >= S > L2 - 1)
            const long L2 = (long)(L - front);
            d_zhead.reserve((size_t)L2);
            launch_head_z(src, (long)hist, d_t1.p, (int)(front + 1), d_zhead.p, L2 - 1, s);
            launch_head_y(d_zhead.p, d_t2.p, (int)L2, static_cast<cf*>(out), L2 - 1, s);
        }
        // (after the head fix, which a poisoned first block covers — and whose finite values may hide a poisoned tile from the
        //  probes: block 0 of that one call is looked at regardless)
        if (in_kernel) { tail_host_known = false; seq++; }      // (the verdict on this call's last block is in d_tail, not on the host)
        else ref_blocks_pass(src, static_cast<cf*>(out), (long)n_out, s, head);
        emitted += n_out;
    } else {
        launch_carry(src, carry, s);
    }
    if (*consumed) {
        cur ^= 1;
        pend_len = new_pend;
    }
    *produced = n_out;
    return st;
}

// ---- fused FM chain ------------------------------------------------------------------------------------
static int64_t gcd64(int64_t a, int64_t b);
void ChainNf::init(hipStream_t s) {
    const int none[6] = {-1, -1, -1, -1, -1, -1};
    slots.upload(none, 6, s);
    on = true;
}
// whether a chain gets the pass: not with rr_build_opts.fft_nonfinite_tiles = 1, not from RTL-SDR bytes (always finite)
static bool chain_nf_wanted(bool u8, int64_t, int64_t, size_t) {
    return !u8 && build_opts().fft_nonfinite_tiles != 1;
}
static void reversed_taps(const rr_c32* t, size_t L, cf* dst) {
    for (size_t j = 0; j < L; j++) dst[j] = mkcf(t[L - 1 - j].re, t[L - 1 - j].im);
}
// outputs (r positions) of the smallest tile a chain kernel of this call may have run on: P_y filtered samples hold at least
// floor(P_y I / D) of them
// (the chain kernels advance by fm_advance((F - L + 1) - ceil(D / I)), which rounds down to whole epilogue rounds but never
//  below 9/10 of it — kernels_fft.hip; the half-size inverse takes two more off: a stride of 0.85 of the nominal count, less two,
//  is below every variant's, and a lattice finer than the tiles only costs a few more loads)
static long chain_probe_stride(long P_y, int64_t I, int64_t D) {
    const long G = (long)((D + I - 1) / I);
    const __int128 q = (__int128)(P_y - G) * I * 85 / (D * 100) - 2;
    return q < 1 ? 1 : (long)q;
}
FmChain::FmChain(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float g, int m, bool u8, int max_log2f,
                 const rr_c32* fir_taps, size_t fir_ntaps)
    : Block(fir_taps ? "FirFilter>FftFilter>RationalResampler>QuadratureDemod"
            : u8 ? "RtlSdrDecode>FftFilter>RationalResampler>QuadratureDemod" : "FftFilter>RationalResampler>QuadratureDemod",
            u8 ? 1 : 8, 4), gain(g), mode(m), iq8(u8) {
    if (deci == 0) throw Error("RationalResampler created using deci 0");
    if (interp == 0) throw Error("RationalResampler created using interp 0");
    if (m != RR_ATAN2_EXACT && m != RR_ATAN2_FAST) throw Error("QuadratureDemod: bad atan2 mode");
    // the kernels index (A + y) * I and u * D in 64-bit: with the reduced ratio below 2^31 every product of a stream
    // position (< 2^32 per call, rebased) stays far below 2^63
    if (interp > (size_t)1 << 31 || deci > (size_t)1 << 31) throw NotFusedShape("FmChain: interp and deci must be <= 2^31");
    const int64_t gg = gcd64((int64_t)deci, (int64_t)interp);
    D = (int64_t)deci / gg; I = (int64_t)interp / gg;
    // decided before any table is built: the fused kernels run on LDS-resident tiles of up to 2^max_log2f points
    {
        const size_t Lc = fir_taps ? fir_ntaps + ntaps - 1 : ntaps, lim = ((size_t)1 << std::min(max_log2f, 14)) - 1;
        if (ntaps && Lc > lim) throw NotFusedShape("FmChain: more taps than the largest tile of the fused kernels holds");
    }
    if (fir_taps) {                      // front FirFilter fused in: composite taps, head fix at stream start
        if (u8) throw Error("FmChain: the front FirFilter takes Complex input");
        const std::vector<rr_c32> gt = FftFilter::composite(fir_taps, fir_ntaps, taps, ntaps);
        f.reset(new FftFilter(gt.data(), gt.size(), true, max_log2f, false, fir_ntaps - 1));
        f->set_stage_taps(fir_taps, fir_ntaps, taps, ntaps);
        ntaps = gt.size();
    } else {
        f.reset(new FftFilter(taps, ntaps, true, max_log2f));
    }
    if (f->big) throw NotFusedShape("FmChain: at most 16383 taps (the fused kernels run on LDS-resident tiles)");
    const int64_t G = (D + I - 1) / I;
    if (G >= (int64_t)(((size_t)1 << f->log2f) - f->L + 1)) throw NotFusedShape("FmChain: decimation too large for the FFT tile");
    half_ok = !f->nsub && fm_multi_half_supported(f->log2f, I, D, (int)ntaps) && !build_opts().fm_full;
    if (half_ok) {
        const size_t FH = ((size_t)1 << f->log2f) / 2;
        std::vector<cf> th(FH);
        for (size_t k = 0; k < FH; k++) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)FH;
            th[k] = mkcf((float)std::cos(a), (float)std::sin(a));
        }
        d_tw_half.upload(th.data(), th.size(), stream);
    }
    // decimate-first tiles: D phase transforms + one inverse per 1024 OUTPUT-rate positions (kernels_poly.hip).  Where they
    // win (tools/poly_probe.py on MI355X, ms per 2.4e7 samples, poly / best other kernel): 463 taps 1:2 0.084 / 0.102, 1:4
    // 0.065 / 0.087, 1:6 0.067 / 0.085, 1:7 0.083 / 0.095, 1:8 0.096 / 0.083; 2467 taps 1:6 0.086 / 0.157, 1:10 0.113 / 0.152,
    // 1:12 0.159 / 0.149 — every decimation up to 6 (7 phases and more run in two register batches per wave), and up to
    // 10 for filters long enough to push the other kernels onto 4096-point or split tiles.  fm_poly > 0 forces them.
    window_aware = build_opts().fm_poly == 0;          // any forced choice is used at every window size
    // Round 4: up to 768 taps per phase (256 of a tile's 1024 positions are output) instead of 448 — measured for 1:4 … 1:10
    // (tools/poly_long_probe.py, ms per 2.4e7 samples, decimate-first / best other: 1:6 3599 taps 0.104 / 0.171, 4559 taps
    // 0.158 / 0.217; 1:4 3039 taps 0.158 / 0.162; 1:10 7599 taps 0.168 / 0.252).
    const uint64_t Lsp = (f->L + (uint64_t)D - 1) / (uint64_t)D;
    // Decimations 9 and 11 … 16 (round 4; they fell to the 2048-point / split tiles before: 2467 taps 1:9 0.177 ms per 2.4e7
    // samples against 0.081 now) — tools/poly_probe.py, forced decimate-first / other: 1:9 1000 taps 0.079 / 0.105, 1:11 2467
    // 0.090 / 0.140, 1:12 2467 0.092 / 0.139, 1:13 2467 0.120 / 0.138, 1:14 4000 0.141 / 0.185, 1:16 4000 0.150 / 0.182; the
    // two-wave kernel takes a third batch of phases from 1:13 on, hence the later crossovers.
    // (1:7 on the three-waves-per-SIMD kernel, four waves of 2 + 2 + 2 + 1 phases: 463 taps 0.090 -> 0.059, wins at every length)
    const bool d_wins = D <= 7 || (D <= 11 && f->L >= 800) || (D == 12 && f->L >= 1000) || (D == 13 && f->L >= 2000) ||
                        (D >= 14 && D <= 16 && f->L >= 2000);
    // (1:2 wins up to 500 taps per phase — 1000 taps 0.122 / 0.126, 1300 taps 0.159 / 0.141 — and 1:3 up to 600)
    const bool poly_wins = d_wins && Lsp <= (D == 2 ? 500u : D == 3 ? 600u : 768u);
    if ((build_opts().fm_poly > 0 || (build_opts().fm_poly == 0 && poly_wins)) && !build_opts().fm_full && I == 1 && D >= 2) {
        std::vector<rr_c32> ct(f->L);
        if (fir_taps) ct = FftFilter::composite(fir_taps, fir_ntaps, taps, f->L - (fir_ntaps - 1));
        else std::copy(taps, taps + f->L, ct.begin());
        poly.reset(new PolyTables());
        if (!poly->build(ct.data(), 1, f->L, (size_t)D, false, stream)) poly.reset();
    }
    for (auto& b : last_r) { b.reserve(1); RR_HIP(hipMemsetAsync(b.p, 0, sizeof(cf), stream)); }
    if (chain_nf_wanted(u8, I, D, f->nsamples)) {
        std::vector<rr_c32> ct(f->L);
        if (fir_taps) ct = FftFilter::composite(fir_taps, fir_ntaps, taps, f->L - (fir_ntaps - 1));
        else std::copy(taps, taps + f->L, ct.begin());
        std::vector<cf> r(f->L);
        reversed_taps(ct.data(), f->L, r.data());
        nf.rev_c.upload(r.data(), r.size(), stream);
        nf.init(stream);
    }
    RR_HIP(hipStreamSynchronize(stream));
}

size_t OutTail::drain(float* out, size_t out_stride, size_t out_cap, size_t C, hipStream_t s) {
    const size_t m = std::min(out_cap, len - pos);
    // (hipMemcpyDefault: `out` is a device window, or a page-locked host window the kernels write in place)
    if (m) RR_HIP(hipMemcpy2DAsync(out, out_stride * sizeof(float), buf.p + pos, cap * sizeof(float), m * sizeof(float), C,
                                   hipMemcpyDefault, s));
    pos += m;
    return m;
}

size_t FmChain::next_block_outputs() const {
    auto N3 = [&](uint64_t y) { const uint64_t r = (uint64_t)(((__int128)y * I + D - 1) / D); return r ? r - 1 : 0; };
    return (size_t)(N3(n1 + f->nsamples) - N3(n1));
}

int FmChain::work_blocks(const void* in, size_t in_len, float* out, size_t, size_t out_cap, size_t* consumed,
                         size_t* produced, size_t* need, hipStream_t s, uint64_t max_blocks) {
    *consumed = *produced = *need = 0;
    if (iq8) in_len /= 2;                          // whole I/Q pairs; an odd trailing byte is never consumed (:23)
    bool packed = iq8;
    if (iq8 && ((uintptr_t)in & 1)) {              // odd-addressed byte window: decode out of line
        decoded.reserve(std::max<size_t>(in_len, 1));
        launch_rtlsdr_decode(static_cast<const unsigned char*>(in), decoded.p, (long)in_len, s);
        in = decoded.p;
        packed = false;
    }
    const uint64_t S = f->nsamples;
    auto N2 = [&](uint64_t y) { return (uint64_t)(((__int128)y * I + D - 1) / D); };
    auto N3 = [&](uint64_t y) { const uint64_t r = N2(y); return r ? r - 1 : 0; };
    const uint64_t o_old = N3(n1);
    // like FftFilter::work (fft_filter.rs:294-303): room for one block's worth of output first
    const uint64_t need_next = N3(n1 + S) - o_old;
    if (need_next > out_cap) { *need = need_next; return RR_WAIT_DST; }
    if (f->front && in_len < f->front + 1) { *need = f->front + 1; return RR_WAIT_SRC; }   // front FirFilter: fir.rs:498-501
    const uint64_t av = f->avail(in_len);
    const uint64_t total = f->pend_len + av;
    const uint64_t k_in = total / S;
    // largest k with N3(n1 + k S) - o_old <= out_cap  <=>  ceil((n1+kS) I / D) <= o_old + out_cap + 1
    const __int128 X = (__int128)(o_old + out_cap + 1) * D / I;        // (n1 + k S) <= X
    uint64_t k_out = X >= (__int128)n1 ? (uint64_t)((X - n1) / S) : 0;
    while (k_out > 0 && N3(n1 + k_out * S) - o_old > out_cap) k_out--;
    k_out = std::min(k_out, max_blocks);
    uint64_t k, new_pend;
    int st;
    if (k_in > k_out) {
        k = k_out; *consumed = k * S - f->pend_len; new_pend = 0;
        st = RR_WAIT_DST; *need = N3(n1 + (k + 1) * S) - N3(n1 + k * S);
    } else {
        k = k_in; *consumed = av; new_pend = total - k * S;
        st = RR_WAIT_SRC; *need = S - new_pend + f->front;
    }
    const uint64_t n_y = k * S;
    const long plen = (long)(f->hist + f->pend_len);
    VSrc<cf> src{f->prefix[f->cur].p, plen, static_cast<const cf*>(in), (long)in_len};
    VSrcIQ8 src8{f->prefix[f->cur].p, plen, static_cast<const rr::iq8*>(in), (long)in_len};
    if (k) {
        FmChainArgs a;
        a.A = (long)n1; a.n_y = (long)n_y; a.r_lo = (long)N2(n1); a.r_hi = (long)N2(n1 + n_y);
        a.o_base = (long)o_old; a.I = I; a.D = D; a.gain = gain; a.mode = mode;
        if (*consumed) a.carry = CarryOut{f->prefix[f->cur ^ 1].p, (long)n_y, (long)(f->hist + new_pend)};   // (see FftFilter::work_dev)
        // decimate-first tiles cover D x 946 inputs each: below one tile per resident workgroup slot (~6 M samples at 1:6)
        // the 2048-point tiles keep more of the chip busy (tools/call_overhead.py, 463 taps 1:6, 512 k samples: 21.7 us
        // decimate-first, 15.4-18.8 us on the 2048-point tiles; 8 M samples: 32.9 against 35.1).  All kernels share the
        // carried state (prefix, pending samples, last r), so the choice is per call.
        bool use_poly = poly != nullptr;
        if (use_poly && window_aware) {
            const uint64_t Ls = (f->L + (uint64_t)D - 1) / (uint64_t)D;
            use_poly = (a.r_hi - a.r_lo) >= (long)(chip_units(1000) * (1024 - Ls));
        }
        // (and below ~1.2 M samples the plain 2048-point tiles — more, smaller workgroups — beat the half-size inverse,
        //  which finishes two tiles per workgroup: 512 k samples 15.4 against 18.8 us)
        const bool use_half = half_ok && (!window_aware || n_y >= chip_units(1200000));
        const bool use_alt = f->nsub && f->alt_log2f && window_aware && f->alt_wins((long)n_y, true) &&
                             (int64_t)((D + I - 1) / I) < (int64_t)(((size_t)1 << f->alt_log2f) - f->L + 1);
        prof_begin(s);
        if (use_poly && packed)
            launch_fm_chain_poly_iq8(src8, out, (int)f->L, poly->d_tw.p, poly->d_h.p, a, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (use_poly)
            launch_fm_chain_poly(src, out, (int)f->L, poly->d_tw.p, poly->d_h.p, a, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (use_alt && packed)        // long filter, window of too few split tiles: the plain 4096-point chain tile (see FftFilter)
            launch_fm_chain_iq8(f->alt_log2f, src8, out, (int)f->L, f->d_tw_alt.p, f->d_hpos_alt.p, a,
                                last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (use_alt)
            launch_fm_chain(f->alt_log2f, src, out, (int)f->L, f->d_tw_alt.p, f->d_hpos_alt.p, a,
                            last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (f->nsub && packed)
            launch_fm_chain_split_iq8(f->nsub, src8, out, (int)f->L, f->d_tw4096.p, f->d_hs.p, f->d_wk.p, a,
                                      last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (f->nsub)
            launch_fm_chain_split(f->nsub, src, out, (int)f->L, f->d_tw4096.p, f->d_hs.p, f->d_wk.p, a,
                                  last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (use_half && packed)
            launch_fm_chain_half_iq8(src8, out, (int)f->L, f->d_tw.p, d_tw_half.p, f->d_hpos.p, a,
                                     last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (use_half)
            launch_fm_chain_half(src, out, (int)f->L, f->d_tw.p, d_tw_half.p, f->d_hpos.p, a,
                                 last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (packed)
            launch_fm_chain_iq8(f->log2f, src8, out, (int)f->L, f->d_tw.p, f->d_hpos.p, a,
                                last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else
            launch_fm_chain(f->log2f, src, out, (int)f->L, f->d_tw.p, f->d_hpos.p, a,
                            last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        prof_end(s);
        if (f->front && n1 == 0) {
            // head fix (see FftFilter): every demodulated sample that touches y[n], n < L2 - 1, is recomputed from the
            // two-stage definition.  n1 == 0: the whole head lies in this first emitting call (n_y >= S > L2 - 1).
            const long L2 = (long)(f->L - f->front);
            const long nz = std::min<long>((long)n_y, L2 - 1 + (long)((D + I - 1) / I) + 1);
            f->d_zhead.reserve((size_t)nz);
            launch_head_z(src, (long)f->hist, f->d_t1.p, (int)(f->front + 1), f->d_zhead.p, nz, s);
            launch_head_demod(f->d_zhead.p, nz, f->d_t2.p, (int)L2, I, D, gain, mode, a.r_hi, out,
                              last_r[cur_lr ^ 1].p, s);
        }
        if (nf.on && !iq8) {
            // the smallest tile a kernel above may have used (decimate-first: 1024 - Ls output positions per tile)
            const uint64_t Lsd = (f->L + (uint64_t)D - 1) / (uint64_t)D;
            const long P_os = (long)((size_t)1 << (use_alt ? f->alt_log2f : use_half ? 11 : f->log2f)) - (long)f->L + 1;
            const long P = use_poly ? (long)(1024 - Lsd) : chain_probe_stride(P_os, I, D);
            launch_chain_blocks_nonfinite(src, out, 0, 1, a, (long)S, (long)f->hist, P, (int)f->L, (int)f->front, nf.rev_c.p, 0,
                                          last_r[cur_lr].p, last_r[cur_lr ^ 1].p, nf.slots.p, nf.seq, (f->front && n1 == 0) ? 1 : 0, s);
            nf.seq++;
        }
        if (a.r_hi > a.r_lo) cur_lr ^= 1;
    } else if (*consumed) {                        // nothing to filter yet: the window only joins the pending samples
        if (packed) launch_vcopy_iq8(src8, (long)n_y, f->prefix[f->cur ^ 1].p, (long)(f->hist + new_pend), s);
        else launch_vcopy_c32(src, (long)n_y, f->prefix[f->cur ^ 1].p, (long)(f->hist + new_pend), s);
    }
    if (*consumed) {
        f->cur ^= 1;
        f->pend_len = new_pend;
    }
    *produced = N3(n1 + n_y) - o_old;
    n1 += n_y;
    if (iq8) {                                     // the input stream counts bytes
        *consumed *= 2;
        if (st == RR_WAIT_SRC) *need *= 2;
    }
    return st;
}

// ---- fused audio stage: FftFilterFloat -> RationalResampler -> MultiplyConst --------------------------------------------
AudioChain::AudioChain(const float* taps, size_t ntaps, size_t interp, size_t deci, float sc)
    : Block("FftFilterFloat>RationalResampler>MultiplyConst", 4, 4), scale(sc) {
    if (ntaps == 0) throw Error("FftFilterFloat: empty taps");
    if (deci == 0) throw Error("RationalResampler created using deci 0");
    if (interp == 0) throw Error("RationalResampler created using interp 0");
    if (interp > (size_t)1 << 31 || deci > (size_t)1 << 31) throw NotFusedShape("AudioChain: interp and deci must be <= 2^31");
    if (ntaps > 3584) throw NotFusedShape("AudioChain: at most 3584 taps (real-stream tiles of up to 4096 points); use the three blocks");
    const int64_t gg = gcd64((int64_t)deci, (int64_t)interp);
    D = (int64_t)deci / gg; I = (int64_t)interp / gg;
    std::vector<rr_c32> ct(ntaps);
    for (size_t i = 0; i < ntaps; i++) ct[i] = rr_c32{taps[i], 0.0f};  // fft_filter.rs:398
    f.reset(new FftFilter(ct.data(), ntaps, false, 12, true));
    if (chain_nf_wanted(false, I, D, f->nsamples)) {
        std::vector<float> r(ntaps);
        for (size_t j = 0; j < ntaps; j++) r[j] = taps[ntaps - 1 - j];
        nf.rev_f.upload(r.data(), r.size(), stream);
        nf.init(stream);
    }
}

// Bookkeeping as FmChain's without the demodulator's one-sample lag: a call emits whole filter blocks and every resampled
// sample whose source lies in them.
size_t AudioChain::next_block_outputs() const {
    auto N2 = [&](uint64_t y) { return (uint64_t)(((__int128)y * I + D - 1) / D); };
    return (size_t)(N2(n1 + f->nsamples) - N2(n1));
}

int AudioChain::work_blocks(const void* in, size_t in_len, float* out, size_t, size_t out_cap, size_t* consumed, size_t* produced,
                            size_t* need, hipStream_t s, uint64_t max_blocks) {
    *consumed = *produced = *need = 0;
    const uint64_t S = f->nsamples;
    auto N2 = [&](uint64_t y) { return (uint64_t)(((__int128)y * I + D - 1) / D); };
    const uint64_t o_old = N2(n1);
    const uint64_t need_next = N2(n1 + S) - o_old;                     // room for one filter block's outputs first
    if (need_next > out_cap) { *need = need_next; return RR_WAIT_DST; }
    const uint64_t total = f->pend_len + in_len, k_in = total / S;
    const __int128 X = (__int128)(o_old + out_cap) * D / I;            // N2(n1 + k S) <= o_old + out_cap
    uint64_t k_out = X >= (__int128)n1 ? (uint64_t)((X - n1) / S) : 0;
    while (k_out > 0 && N2(n1 + k_out * S) - o_old > out_cap) k_out--;
    k_out = std::min(k_out, max_blocks);
    uint64_t k, new_pend;
    int st;
    if (k_in > k_out) {
        k = k_out; *consumed = k * S - f->pend_len; new_pend = 0;
        st = RR_WAIT_DST; *need = N2(n1 + (k + 1) * S) - N2(n1 + k * S);
    } else {
        k = k_in; *consumed = in_len; new_pend = total - k * S;
        st = RR_WAIT_SRC; *need = S - new_pend;
    }
    const uint64_t n_y = k * S;
    VSrc<float> src{reinterpret_cast<const float*>(f->prefix[f->cur].p), (long)(f->hist + f->pend_len),
                    static_cast<const float*>(in), (long)in_len};
    if (k) {
        AudioChainArgs a{(long)n1, (long)n_y, (long)o_old, (long)N2(n1 + n_y), I, D, scale, {}};
        if (*consumed) a.carry = CarryOut{f->prefix[f->cur ^ 1].p, (long)n_y, (long)(f->hist + new_pend)};
        prof_begin(s);
        launch_audio_chain(f->log2f, src, out, (int)f->L, f->d_tw.p, f->d_hpos.p, a, s);
        prof_end(s);
        if (nf.on) {
            const long P_y = (long)((size_t)1 << f->log2f) - (long)f->L + 1;       // one real segment of a tile
            launch_chain_blocks_nonfinite(src, out, a, (long)S, (long)f->hist, chain_probe_stride(P_y, I, D), (int)f->L, nf.rev_f.p,
                                          nf.slots.p, nf.seq, s);
            nf.seq++;
        }
    } else if (*consumed) {
        launch_vcopy_f32(src, (long)n_y, reinterpret_cast<float*>(f->prefix[f->cur ^ 1].p), (long)(f->hist + new_pend), s);
    }
    if (*consumed) {
        f->cur ^= 1;
        f->pend_len = new_pend;
    }
    *produced = N2(n1 + n_y) - o_old;
    n1 += n_y;
    return st;
}

// ---- N FM chains on one shared source ---------------------------------------------------------------------
FmMulti::FmMulti(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float g, int m, bool u8)
    : Block(u8 ? "RtlSdrDecode>Tee>N x (FftFilter>RationalResampler>QuadratureDemod)" : "Tee>N x (FftFilter>RationalResampler>QuadratureDemod)",
            u8 ? 1 : 8, 4), C(nchan), iq8(u8) {
    if (nchan == 0 || nchan > 4096) throw Error("FmMulti: channel count must be 1..4096");
    zero_copy_in = false;                            // (its kernels read a tile once per run of channel rounds: upload it)
    if (deci == 0) throw Error("RationalResampler created using deci 0");
    if (interp == 0) throw Error("RationalResampler created using interp 0");
    // Integer decimations up to 8 run on the decimate-first tiles (kernels_poly.hip) for up to 768 taps PER PHASE — they
    // beat everything else there (tools/poly_long_probe.py, 32 channels, ms per 2.4e6 samples, decimate-first / other: 1:6
    // 2687 taps 0.097 / 0.426, 3119 taps 0.106 / 0.569; 1:4 3039 taps 0.210 / 0.566) and need nothing of the 4096-point
    // shared-forward kernels, so the filter may be longer than those take.
    const int64_t gg0 = gcd64((int64_t)std::min<size_t>(deci, (size_t)1 << 31), (int64_t)std::min<size_t>(interp, (size_t)1 << 31));
    const bool want_poly = build_opts().fm_poly >= 0 && !build_opts().fm_full && interp <= ((size_t)1 << 31) && deci <= ((size_t)1 << 31) &&
                           (int64_t)interp / gg0 == 1 && ntaps <= 16383 &&
                           fm_poly_supported(1, (long)((int64_t)deci / gg0), (int)ntaps, true);
    // Everything else is the shared-forward kernels on tiles of at most 4096 points, which yield 4097 - ntaps filtered samples
    // each: towards 4094 taps a tile is all overlap (tools/multi_taps_probe.py, 32 channels 1:6 without decimate-first
    // tiles, ms per 2.4e6 samples: 3000 taps 0.55, 3800 taps 1.80) while one fused chain per channel on 8192-point split tiles
    // (compose.cpp Parallel) costs 0.83 ms at 5000 taps and less below.  The crossover is where a tile still yields ~768 samples.
    if (!want_poly && ntaps > 4097 - 768)
        throw NotFusedShape("FmMulti: beyond 3329 taps one fused chain per channel on larger tiles is cheaper");
    chain.reset(new FmChain(taps, ntaps, interp, deci, g, m, false, want_poly ? 14 : 12));   // bookkeeping, carry state; tile-agnostic
    if (want_poly) {
        poly.reset(new PolyTables());
        if (!poly->build(taps, C, ntaps, (size_t)chain->D, true, stream)) poly.reset();
        if (build_opts().fm_poly == 8 || build_opts().fm_poly == 12) poly_waves = build_opts().fm_poly;
    }
    if (!poly) {
        const int lg = chain->f->log2f;
        if (chain->f->nsub || !fm_multi_supported(lg)) throw NotFusedShape("FmMulti: at most 4094 taps (3-pass tiles)");
        const size_t F = (size_t)1 << lg;
        std::vector<cf> all(C * F), one;
        for (size_t c = 0; c < C; c++) {
            compute_hpos(taps + c * ntaps, ntaps, lg, one);
            std::copy(one.begin(), one.end(), all.begin() + c * F);
        }
        d_hpos_all.upload(all.data(), all.size(), stream);
        half_ok = fm_multi_half_supported(lg, chain->I, chain->D, (int)ntaps) && !build_opts().fm_full;
        if (half_ok) {
            std::vector<cf> th(F / 2);
            for (size_t k = 0; k < F / 2; k++) {
                const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)(F / 2);
                th[k] = mkcf((float)std::cos(a), (float)std::sin(a));
            }
            d_tw_half.upload(th.data(), th.size(), stream);
        }
    }
    for (auto& b : last_r) { b.reserve(C); RR_HIP(hipMemsetAsync(b.p, 0, C * sizeof(cf), stream)); }
    chain->nf.on = false;
    if (chain_nf_wanted(u8, chain->I, chain->D, chain->f->nsamples)) {
        std::vector<cf> r(C * ntaps);
        for (size_t c = 0; c < C; c++) reversed_taps(taps + c * ntaps, ntaps, r.data() + c * ntaps);
        nf.rev_c.upload(r.data(), r.size(), stream);
        nf.init(stream);
    }
    RR_HIP(hipStreamSynchronize(stream));
}

// `out` holds C windows of out_cap elements each (channel c at out + c*out_cap); all channels
// consume and produce the same counts, so the bookkeeping is FmChain's.
int FmMulti::work_blocks(const void* in, size_t in_len, float* out, size_t out_stride, size_t out_cap, size_t* consumed,
                         size_t* produced, size_t* need, hipStream_t s, uint64_t max_blocks) {
    FmChain& ch = *chain;
    FftFilter* f = ch.f.get();
    *consumed = *produced = *need = 0;
    if (iq8) in_len /= 2;                          // whole I/Q pairs; an odd trailing byte is never consumed
    bool packed = iq8;
    if (iq8 && ((uintptr_t)in & 1)) {              // odd-addressed byte window: decode out of line
        decoded.reserve(std::max<size_t>(in_len, 1));
        launch_rtlsdr_decode(static_cast<const unsigned char*>(in), decoded.p, (long)in_len, s);
        in = decoded.p;
        packed = false;
    }
    const uint64_t S = f->nsamples;
    const int64_t I = ch.I, D = ch.D;
    auto N2 = [&](uint64_t y) { return (uint64_t)(((__int128)y * I + D - 1) / D); };
    auto N3 = [&](uint64_t y) { const uint64_t r = N2(y); return r ? r - 1 : 0; };
    const uint64_t n1 = ch.n1, o_old = N3(n1);
    const uint64_t need_next = N3(n1 + S) - o_old;
    if (need_next > out_cap) { *need = need_next; return RR_WAIT_DST; }
    const uint64_t total = f->pend_len + in_len, k_in = total / S;
    const __int128 X = (__int128)(o_old + out_cap + 1) * D / I;
    uint64_t k_out = X >= (__int128)n1 ? (uint64_t)((X - n1) / S) : 0;
    while (k_out > 0 && N3(n1 + k_out * S) - o_old > out_cap) k_out--;
    k_out = std::min(k_out, max_blocks);
    uint64_t k, new_pend;
    int st;
    if (k_in > k_out) {
        k = k_out; *consumed = k * S - f->pend_len; new_pend = 0;
        st = RR_WAIT_DST; *need = N3(n1 + (k + 1) * S) - N3(n1 + k * S);
    } else {
        k = k_in; *consumed = in_len; new_pend = total - k * S;
        st = RR_WAIT_SRC; *need = S - new_pend;
    }
    const uint64_t n_y = k * S;
    VSrc<cf> src{f->prefix[f->cur].p, (long)(f->hist + f->pend_len), static_cast<const cf*>(in), (long)in_len};
    VSrcIQ8 src8{f->prefix[f->cur].p, (long)(f->hist + f->pend_len), static_cast<const rr::iq8*>(in), (long)in_len};
    if (k) {
        FmChainArgs a;
        a.A = (long)n1; a.n_y = (long)n_y; a.r_lo = (long)N2(n1); a.r_hi = (long)N2(n1 + n_y);
        a.o_base = (long)o_old; a.I = I; a.D = D; a.gain = ch.gain; a.mode = ch.mode;
        if (*consumed) a.carry = CarryOut{f->prefix[f->cur ^ 1].p, (long)n_y, (long)(f->hist + new_pend)};
        a.multi_waves = poly_waves;
        prof_begin(s);
        if (poly && packed)
            launch_fm_multi_poly_iq8(src8, out, (long)out_stride, (int)f->L, poly->d_tw.p, poly->d_h.p, (int)C, a,
                                     last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (poly)
            launch_fm_multi_poly(src, out, (long)out_stride, (int)f->L, poly->d_tw.p, poly->d_h.p, (int)C, a,
                                 last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (half_ok && packed)
            launch_fm_multi_half_iq8(f->log2f, src8, out, (long)out_stride, (int)f->L, f->d_tw.p, d_tw_half.p,
                                     d_hpos_all.p, (int)C, a, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (packed)
            launch_fm_multi_iq8(f->log2f, src8, out, (long)out_stride, (int)f->L, f->d_tw.p, d_hpos_all.p,
                                (int)C, a, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else if (half_ok)
            launch_fm_multi_half(f->log2f, src, out, (long)out_stride, (int)f->L, f->d_tw.p, d_tw_half.p,
                                 d_hpos_all.p, (int)C, a, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        else
            launch_fm_multi(f->log2f, src, out, (long)out_stride, (int)f->L, f->d_tw.p, d_hpos_all.p,
                            (int)C, a, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, s);
        prof_end(s);
        if (nf.on && !iq8) {
            const uint64_t Lsd = (f->L + (uint64_t)D - 1) / (uint64_t)D;
            const long P_os = (long)((size_t)1 << f->log2f) - (long)f->L + 1;
            const long P = poly ? (long)(1024 - Lsd) : chain_probe_stride(P_os, I, D);
            launch_chain_blocks_nonfinite(src, out, (long)out_stride, (int)C, a, (long)S, (long)f->hist, P, (int)f->L, 0, nf.rev_c.p,
                                          (long)f->L, last_r[cur_lr].p, last_r[cur_lr ^ 1].p, nf.slots.p, nf.seq, 0, s);
            nf.seq++;
        }
        if (a.r_hi > a.r_lo) cur_lr ^= 1;
    } else if (*consumed) {
        if (packed) launch_vcopy_iq8(src8, (long)n_y, f->prefix[f->cur ^ 1].p, (long)(f->hist + new_pend), s);
        else launch_vcopy_c32(src, (long)n_y, f->prefix[f->cur ^ 1].p, (long)(f->hist + new_pend), s);
    }
    if (*consumed) {
        f->cur ^= 1;
        f->pend_len = new_pend;
    }
    *produced = N3(n1 + n_y) - o_old;
    ch.n1 += n_y;
    if (iq8) {                                     // the input stream counts bytes
        *consumed *= 2;
        if (st == RR_WAIT_SRC) *need *= 2;
    }
    return st;
}

// ---- FftFilterFloat (fft_filter.rs:365-491) ---------------------------------------------------------
FftFilterFloat::FftFilterFloat(const float* taps, size_t ntaps) : Block("FftFilterFloat", 4, 4) {
    if (ntaps == 0) throw Error("FftFilterFloat: empty taps");
    std::vector<rr_c32> ct(ntaps);
    for (size_t i = 0; i < ntaps; i++) ct[i] = rr_c32{taps[i], 0.0f};  // fft_filter.rs:398
    real_inner = ntaps <= 3584 && !build_opts().fftfloat_complex;       // (the option keeps the three-kernel path testable)
    inner.reset(real_inner ? new FftFilter(ct.data(), ntaps, false, 12, true) : new FftFilter(ct.data(), ntaps));
    inner->ref_blocks_on(ct.data());
    cap = 4096000 / sizeof(cf);                                         // inner streams: stream.rs:105,336-339
    if (real_inner) {
        for (auto& b : fin) b.reserve(cap);
        for (auto& b : fout) b.reserve(cap);
    } else {
        for (auto& b : iin) b.reserve(cap);
        for (auto& b : iout) b.reserve(cap);
    }
}

int FftFilterFloat::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                             size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    // outer input -> inner_in as Complex(x, 0)   (fft_filter.rs:431-445)
    const size_t n = std::min(in_len, cap - iin_len);
    if (real_inner) {
        if (n) RR_HIP(hipMemcpyAsync(fin[ci].p + iin_len, in, n * sizeof(float), hipMemcpyDefault, s));
    } else {
        launch_f32_to_c32(static_cast<const float*>(in), iin[ci].p + iin_len, (long)n, s);
    }
    iin_len += n;
    *consumed = n;
    // inner complex filter (fft_filter.rs:450)
    size_t ic = 0, ip = 0, ineed = 0;
    const int st = real_inner
        ? inner->work_dev(fin[ci].p, iin_len, fout[co].p + iout_len, cap - iout_len, &ic, &ip, &ineed, s)
        : inner->work_dev(iin[ci].p, iin_len, iout[co].p + iout_len, cap - iout_len, &ic, &ip, &ineed, s);
    if (ic) {
        if (real_inner) {
            VSrc<float> v{nullptr, 0, fin[ci].p, (long)iin_len};
            launch_vcopy_f32(v, (long)ic, fin[ci ^ 1].p, (long)(iin_len - ic), s);
        } else {
            VSrc<cf> v{nullptr, 0, iin[ci].p, (long)iin_len};
            launch_vcopy_c32(v, (long)ic, iin[ci ^ 1].p, (long)(iin_len - ic), s);
        }
        ci ^= 1; iin_len -= ic;
    }
    iout_len += ip;
    // inner_out -> outer output, real part   (fft_filter.rs:453-470)
    const size_t m = std::min(iout_len, out_cap);
    if (m == 0 && iout_len != 0) { *need = 1; return RR_WAIT_DST; }   // :457-459
    if (real_inner) {
        if (m) RR_HIP(hipMemcpyAsync(out, fout[co].p, m * sizeof(float), hipMemcpyDefault, s));
    } else {
        launch_c32_re(iout[co].p, static_cast<float*>(out), (long)m, s);
    }
    if (m) {
        if (real_inner) {
            VSrc<float> v{nullptr, 0, fout[co].p, (long)iout_len};
            launch_vcopy_f32(v, (long)m, fout[co ^ 1].p, (long)(iout_len - m), s);
        } else {
            VSrc<cf> v{nullptr, 0, iout[co].p, (long)iout_len};
            launch_vcopy_c32(v, (long)m, iout[co ^ 1].p, (long)(iout_len - m), s);
        }
        co ^= 1; iout_len -= m;
    }
    *produced = m;
    *need = ineed;                                                      // :474-489
    return st;
}

// ---- RationalResampler (rational_resampler.rs:100-213) -------------------------------------------------
static int64_t gcd64(int64_t a, int64_t b) {
    while (b != 0) { const int64_t t = b; b = a % b; a = t; }
    return a;
}
Resampler::Resampler(size_t interp, size_t deci, size_t es) : Block("RationalResampler", es, es) {
    if (deci == 0) throw Error("RationalResampler created using deci 0");      // :130-132
    if (interp == 0) throw Error("RationalResampler created using interp 0");  // :133-135
    if (!(es == 1 || es == 2 || es == 4 || es == 8 || es == 16)) throw Error("RationalResampler: element size must be 1,2,4,8 or 16");
    if (interp > (size_t)INT64_MAX || deci > (size_t)INT64_MAX) throw Error("RationalResampler: ratio out of range");  // i64::try_from :142-143
    // the gather kernel indexes m * D - c0 in 64-bit with m < 2^32 per call: the ratio has to stay below 2^31
    if (interp > (size_t)1 << 31 || deci > (size_t)1 << 31) throw Error("RationalResampler: interp and deci must be <= 2^31 on the GPU");
    const int64_t g = gcd64((int64_t)deci, (int64_t)interp);
    D = (int64_t)deci / g; I = (int64_t)interp / g;
    d_pending.reserve(16);
}
bool Resampler::eof(bool src_eof) { return !has_pending && src_eof; }        // :209-213

int Resampler::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                        size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = 0; *need = 1;
    if (out_cap == 0) return RR_WAIT_DST;                                      // :158-160
    int64_t r = 0;
    if (has_pending) {                                                         // :162-173
        const int64_t r_full = counter > 0 ? (counter + D - 1) / D : 0;
        if (r_full >= (int64_t)out_cap) {
            r = (int64_t)out_cap; counter -= r * D;
            launch_resample(in, out, in_es, r, d_pending.p, 0, I, D, 0, s);
            *produced = (size_t)r;
            return RR_WAIT_DST;
        }
        r = r_full; counter -= r * D; has_pending = false;
    }
    if (in_len == 0) {                                                         // :176-179
        if (r) launch_resample(in, out, in_es, r, d_pending.p, 0, I, D, 0, s);
        *produced = (size_t)r;
        return RR_WAIT_SRC;
    }
    const int64_t c0 = counter, capp = (int64_t)out_cap - r, n = (int64_t)in_len;
    const __int128 a = (__int128)c0 + (__int128)n * I;
    const int64_t m_total = a <= 0 ? 0 : (int64_t)((a + D - 1) / D);
    if (m_total < capp) {                                                      // all input taken
        prof_begin(s);
        launch_resample(in, out, in_es, r, d_pending.p, m_total, I, D, c0, s);
        prof_end(s);
        counter = (int64_t)(a - (__int128)m_total * D);
        *consumed = in_len; *produced = (size_t)(r + m_total);
        return RR_WAIT_SRC;                                                    // :204
    }
    // output fills at emit number capp (:190-196)
    const int64_t kstar = (int64_t)((((__int128)(capp - 1)) * D - c0) / I);
    launch_resample(in, out, in_es, r, d_pending.p, capp, I, D, c0, s);
    counter = (int64_t)((__int128)c0 + (__int128)(kstar + 1) * I - (__int128)capp * D);
    if (counter > 0) {
        RR_HIP(hipMemcpyAsync(d_pending.p, static_cast<const unsigned char*>(in) + (size_t)kstar * in_es, in_es,
                              hipMemcpyDefault, s));     // (`in`: a device window or a page-locked host window read in place)
        has_pending = true;
    }
    *consumed = (size_t)(kstar + 1); *produced = out_cap;
    return RR_WAIT_DST;                                                        // :202
}

// ---- QuadratureDemod (quadrature_demod.rs:32-114) ---------------------------------------------------------
QuadDemod::QuadDemod(float g, int m) : Block("QuadratureDemod", 8, 4), gain(g), mode(m) {
    if (m != RR_ATAN2_EXACT && m != RR_ATAN2_FAST) throw Error("QuadratureDemod: bad atan2 mode");
}
int QuadDemod::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                        size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    if (in_len < 2) { *need = 2; return RR_WAIT_SRC; }                         // :49-51
    if (out_cap == 0) { *need = 1; return RR_WAIT_DST; }                       // :53-55
    const size_t n1 = std::min(in_len - 1, out_cap);                           // :56
    prof_begin(s);
    launch_quaddemod(static_cast<const cf*>(in), static_cast<float*>(out), (long)n1, gain, mode, s);
    prof_end(s);
    *consumed = *produced = n1;                                                // :110-111
    // the reference loops: the next iteration returns the wait
    if (in_len - n1 < 2) { *need = 2; return RR_WAIT_SRC; }
    *need = 1; return RR_WAIT_DST;
}

// ---- FftStream (fft_stream.rs:26-117) ------------------------------------------------------------------------
static void fill_tw(std::vector<cf>& tw, size_t n) {
    tw.resize(n);
    for (size_t k = 0; k < n; k++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
        tw[k] = mkcf((float)std::cos(a), (float)std::sin(a));
    }
}
// c[k] = exp(-i pi k^2 / n), k^2 reduced mod 2 n in integers (exact phase for any n), and b[m] = conj(c[|m|]) wrapped mod M
static void bluestein_tables(size_t n, size_t M, std::vector<std::complex<double>>& c, std::vector<std::complex<double>>& B) {
    c.resize(n);
    B.assign(M, 0.0);
    for (size_t k = 0; k < n; k++) {
        const unsigned __int128 q = ((unsigned __int128)k * k) % (2 * n);
        c[k] = std::polar(1.0, -3.14159265358979323846 * (double)(size_t)q / (double)n);
    }
    B[0] = std::conj(c[0]);
    for (size_t m = 1; m < n; m++) B[m] = B[M - m] = std::conj(c[m]);
    fft64(B);
    for (auto& h : B) h /= (double)M;
}

AnyFft::AnyFft(size_t n, hipStream_t s) : N(n) {
    if (n < 2) throw Error("FFT size must be at least 2");
    const bool pow2 = (n & (n - 1)) == 0;
    if (n > ((size_t)1 << 20) || (!pow2 && n > ((size_t)1 << 19))) throw Error("FFT size beyond 524288 (2^20 for powers of two) is not supported");
    std::vector<cf> tw;
    if (pow2 && n <= 16384) {
        while (((size_t)1 << log2n) < n) log2n++;
        fill_tw(tw, n);
        d_tw.upload(tw.data(), n, s);
        if (n >= 8192 && !build_opts().fft_no_split) {
            fill_tw(tw, 4096);
            d_tw4096.upload(tw.data(), 4096, s);
        }
    } else if (pow2) {                                    // four-step: N1 >= N2, both <= 1024
        int lg = 0;
        while (((size_t)1 << lg) < n) lg++;
        N1 = (size_t)1 << ((lg + 1) / 2); N2 = n / N1;
        f1.reset(new AnyFft(N1, s));
        f2.reset(new AnyFft(N2, s));
        fill_tw(tw, n);
        d_twN.upload(tw.data(), n, s);
    } else if (n <= 2048) {                               // Bluestein fused into one filter tile of M >= 2 n - 1 points
        log2m = 10;
        while (((size_t)1 << log2m) < 2 * n - 1) log2m++;
        const size_t Mt = (size_t)1 << log2m;
        std::vector<std::complex<double>> c, B;
        bluestein_tables(n, Mt, c, B);
        std::vector<cf> hpos, chirp(n);
        switch (log2m) {
        case 10: fill_hpos<10>(B, hpos); break;
        case 11: fill_hpos<11>(B, hpos); break;
        default: fill_hpos<12>(B, hpos); break;
        }
        fill_tw(tw, Mt);
        for (size_t k = 0; k < n; k++) chirp[k] = mkcf((float)c[k].real(), (float)c[k].imag());
        d_tw.upload(tw.data(), Mt, s);
        d_bh.upload(hpos.data(), Mt, s);
        d_chirp.upload(chirp.data(), n, s);
    } else {                                              // Bluestein on M points through the power-of-two engine
        M = 1;
        while (M < 2 * n - 1) M <<= 1;
        fm.reset(new AnyFft(M, s));
        std::vector<std::complex<double>> c, B;
        bluestein_tables(n, M, c, B);
        std::vector<cf> bn(M), chirp(n);
        for (size_t k = 0; k < M; k++) bn[k] = mkcf((float)B[k].real(), (float)B[k].imag());
        for (size_t k = 0; k < n; k++) chirp[k] = mkcf((float)c[k].real(), (float)c[k].imag());
        d_b.upload(bn.data(), M, s);
        d_chirp.upload(chirp.data(), n, s);
    }
}

void AnyFft::forward(const cf* in, cf* out, long nf, hipStream_t s) {
    if (nf <= 0) return;
    if (log2n) { launch_fft_frames(log2n, in, out, nf, d_tw.p, d_tw4096.p, s); return; }
    if (log2m) { launch_fft_bluestein(log2m, in, out, nf, (int)N, d_tw.p, d_bh.p, d_chirp.p, s); return; }
    if (N1) {                                             // n = N2 n1 + n2, k = k1 + N1 k2
        t1.reserve((size_t)nf * N); t2.reserve((size_t)nf * N);
        launch_transpose_tw(in, t1.p, (int)N1, (int)N2, nf, nullptr, s);            // [n1][n2] -> [n2][n1]
        f1->forward(t1.p, t2.p, nf * (long)N2, s);                                   // over n1 -> [n2][k1]
        launch_transpose_tw(t2.p, t1.p, (int)N2, (int)N1, nf, d_twN.p, s);          // * w_N^(n2 k1), -> [k1][n2]
        f2->forward(t1.p, t2.p, nf * (long)N1, s);                                   // over n2 -> [k1][k2]
        launch_transpose_tw(t2.p, out, (int)N1, (int)N2, nf, nullptr, s);           // -> [k2][k1] = natural order
        return;
    }
    t1.reserve((size_t)nf * M); t2.reserve((size_t)nf * M);
    launch_chirp_pre(in, t1.p, (long)N, (long)M, nf, d_chirp.p, s);
    fm->forward(t1.p, t2.p, nf, s);
    launch_mul_conj(t2.p, d_b.p, (long)M, nf, s);
    fm->forward(t2.p, t1.p, nf, s);
    launch_chirp_post(t1.p, out, (long)N, (long)M, nf, d_chirp.p, s);
}

FftStream::FftStream(size_t n) : Block("FftStream", 8, 8), size(n) {
    if (n == 0) throw Error("FFT size must be nonzero");                                   // :42
    if (n > 4096000 / sizeof(cf)) throw Error("FFT size must be no bigger than stream size");   // :46-50
    if (n < 2) throw Error("FftStream: size 1 is the identity (not a GPU block)");
    fft.reset(new AnyFft(n, stream));
    RR_HIP(hipStreamSynchronize(stream));
}
int FftStream::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                        size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    if (in_len < size) { *need = size; return RR_WAIT_SRC; }                   // :74-76
    if (out_cap < size) { *need = size; return RR_WAIT_DST; }                  // :79-81
    size_t len = std::min(in_len, out_cap);                                    // :82-83
    len -= len % size;
    prof_begin(s);
    fft->forward(static_cast<const cf*>(in), static_cast<cf*>(out), (long)(len / size), s);
    prof_end(s);
    *consumed = *produced = len;
    return RR_AGAIN;                                                           // :116
}

// ---- MultiplyConst, FastFM (sync blocks) ------------------------------------------------------------------------
static int sync_counts(size_t in_len, size_t out_cap, size_t* n, size_t* need) {
    *need = 1;
    if (in_len == 0) { *n = 0; return RR_WAIT_SRC; }
    if (out_cap == 0) { *n = 0; return RR_WAIT_DST; }
    *n = std::min(in_len, out_cap);
    return in_len - *n == 0 ? RR_WAIT_SRC : RR_WAIT_DST;      // the loop's next iteration
}
MultiplyConst::MultiplyConst(size_t es, float re, float im) : Block("MultiplyConst", es, es), vr(re), vi(im) {
    if (es != 4 && es != 8) throw Error("MultiplyConst: Float or Complex only");
}
int MultiplyConst::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                            size_t* produced, size_t* need, hipStream_t s) {
    size_t n = 0;
    const int st = sync_counts(in_len, out_cap, &n, need);
    prof_begin(s);
    if (in_es == 4) launch_mulconst_f32(static_cast<const float*>(in), static_cast<float*>(out), (long)n, vr, s);
    else launch_mulconst_c32(static_cast<const cf*>(in), static_cast<cf*>(out), (long)n, vr, vi, s);
    prof_end(s);
    *consumed = *produced = n;
    return st;
}
FastFM::FastFM() : Block("FastFM", 8, 4) {
    for (auto& h : hist) { h.reserve(2); RR_HIP(hipMemsetAsync(h.p, 0, 2 * sizeof(cf), stream)); }
    RR_HIP(hipStreamSynchronize(stream));
}
int FastFM::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                     size_t* produced, size_t* need, hipStream_t s) {
    size_t n = 0;
    const int st = sync_counts(in_len, out_cap, &n, need);
    if (n) {
        VSrc<cf> src{hist[cur].p, 2, static_cast<const cf*>(in), (long)in_len};
        prof_begin(s);
        launch_fastfm(src, static_cast<float*>(out), (long)n, s);
        prof_end(s);
        launch_vcopy_c32(src, (long)n, hist[cur ^ 1].p, 2, s);       // q2, q1 for the next call
        cur ^= 1;
    }
    *consumed = *produced = n;
    return st;
}

// ---- RtlSdrDecode (rtlsdr_decode.rs:9-47) ----------------------------------------------------------------------
RtlSdrDecode::RtlSdrDecode() : Block("RtlSdrDecode", 1, 8) {}
int RtlSdrDecode::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                           size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    size_t isamples = in_len & ~(size_t)1;                                     // :23
    if (isamples == 0) { *need = 2; return RR_WAIT_SRC; }                      // :24-26
    if (out_cap == 0) { *need = 1; return RR_WAIT_DST; }                       // :28-30
    isamples = std::min(isamples, out_cap * 2);                                // :31
    const size_t osamples = isamples / 2;
    prof_begin(s);
    launch_rtlsdr_decode(static_cast<const unsigned char*>(in), static_cast<cf*>(out), (long)osamples, s);
    prof_end(s);
    *consumed = isamples; *produced = osamples;                                // :43-44
    // the reference loops: the next iteration returns the wait
    if (((in_len - isamples) & ~(size_t)1) == 0) { *need = 2; return RR_WAIT_SRC; }
    *need = 1; return RR_WAIT_DST;
}

// ---- Hilbert (hilbert.rs:22-129) -------------------------------------------------------------------------------
Hilbert::Hilbert(size_t ntaps, int window, float parm) : Block("Hilbert", 4, 8) {
    if (!(ntaps > 1 && (ntaps & 1) == 1)) throw Error("hilbert filter len must be odd and greater than 1");  // :44-47
    if (ntaps > 0x3fffffff) throw Error("Hilbert: too many taps");
    std::vector<float> win, taps;
    if (!make_window(window, parm, ntaps, win)) throw Error("Hilbert: unknown window type");
    hilbert_taps(win.data(), ntaps, taps);                                      // :48
    pl.L = (int)ntaps; pl.d = 1; pl.complex_taps = false; pl.cfg = build_opts().fir_cfg;
    std::vector<float> rev(ntaps), tp;
    for (size_t j = 0; j < ntaps; j++) rev[j] = taps[ntaps - 1 - j];           // Fir::new, fir.rs:160
    build_poly(rev, 1, pl.qpad, tp);
    d_rev.upload(rev.data(), rev.size(), stream);
    d_tp.upload(tp.data(), tp.size(), stream);
    // the transformer is exactly zero on one parity of tap positions (fir.rs:667-676): use the
    // half-length polyphase form when that holds for these taps
    par = (int)((((ntaps - 1) / 2) + 1) % 2);
    skip_ok = true;
    for (size_t j = 0; j < ntaps; j++) if ((int)(j % 2) != par && rev[j] != 0.0f) skip_ok = false;
    d_nf_flags.reserve(HILBERT_MAX_GRID);
    RR_HIP(hipMemsetAsync(d_nf_flags.p, 0, HILBERT_MAX_GRID * sizeof(int), stream));
    if (skip_ok) {
        const size_t q = (ntaps - par + 1) / 2;
        Q = (int)((q + 7) / 8 * 8);
        std::vector<float> hq(Q, 0.0f);
        for (size_t i = 0; i < q && 2 * i + par < ntaps; i++) hq[i] = rev[2 * i + par];
        d_hq.upload(hq.data(), hq.size(), stream);
    }
    for (auto& h : hist) {                                                      // :55 — ntaps zeros
        h.reserve(ntaps);
        RR_HIP(hipMemsetAsync(h.p, 0, ntaps * sizeof(float), stream));
    }
    RR_HIP(hipStreamSynchronize(stream));
    if (build_opts().fir_path != RR_PATH_DIRECT && (ntaps >= 200 || build_opts().fir_path == RR_PATH_FFT)) {
        if (ntaps <= 3584) {
            std::vector<rr_c32> ct(ntaps);
            for (size_t i = 0; i < ntaps; i++) ct[i] = {taps[i], 0.0f};
            fftk.reset(new FftFilter(ct.data(), ntaps, false, 12, true));
        } else if (ntaps <= 16383) {
            // a[k] = sum_j c[j] xp[k + j], c[j] = delta[j - L/2] + i rev[j]: caller-order taps t[k] = c[L - 1 - k]
            std::vector<rr_c32> ct(ntaps);
            for (size_t k = 0; k < ntaps; k++) ct[k] = {k == ntaps / 2 ? 1.0f : 0.0f, taps[k]};
            wide.reset(new FirC32(ct.data(), ntaps, 1, false, 0.0f, 0.0f, true));
        }
    }
}
Hilbert::~Hilbert() = default;
void Hilbert::host_out_done(const void* out_host, size_t) {
    if (!deferred.on) return;
    deferred.on = false;
    const std::complex<float>* y = static_cast<const std::complex<float>*>(out_host);
    const long n = deferred.n, P = std::max<long>(1, deferred.P);
    auto bad = [&](long m) { return !(std::isfinite(y[m].real()) && std::isfinite(y[m].imag())); };
    for (long m = 0; m < n; m += P) __builtin_prefetch(&y[m]);
    bool hit = bad(n - 1);
    for (long m = 0; m < n && !hit; m += P) hit = bad(m);
    if (!hit) return;
    launch_hilbert_refold_nonfinite(deferred.src, static_cast<cf*>(deferred.out), n, deferred.P, pl.L, d_rev.p, stream);
    RR_HIP(hipStreamSynchronize(stream));
}
int Hilbert::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                      size_t* produced, size_t* need, hipStream_t s) {
    *consumed = *produced = 0; *need = 1;
    if (in_len == 0) return RR_WAIT_SRC;                                        // :76-78
    if (out_cap == 0) return RR_WAIT_DST;                                       // :81-83
    const size_t n = std::min(in_len, out_cap);                                 // :85-87
    VSrc<float> src{hist[cur].p, (long)pl.L, static_cast<const float*>(in), (long)in_len};
    prof_begin(s);
    // (windows of a few tiles leave the chip idle: the pair-sample / direct kernels keep those, as in FirC32::work_dev)
    if (fftk && n >= 16 * (((size_t)1 << fftk->log2f) - (size_t)pl.L + 1)) {
        launch_fftfilt_real_hilbert(fftk->log2f, src, static_cast<cf*>(out), (long)n, pl.L, fftk->d_tw.p, fftk->d_hpos.p, s, CarryOut{}, nanfix());
        const long P = ((long)1 << fftk->log2f) - (long)pl.L + 1;
        if (host_out) { deferred.on = true; deferred.src = src; deferred.out = out; deferred.n = (long)n; deferred.P = P; }
        else launch_hilbert_refold_nonfinite(src, static_cast<cf*>(out), (long)n, P, pl.L, d_rev.p, s);
    } else if (wide && n >= 4096) {
        // a[k] over the virtual stream xp = hist ++ window needs xp[k, k + L): Complex(xp, 0) in chunks, then the filter
        const size_t L = (size_t)pl.L, CH = (size_t)1 << 24, per = CH - L;
        wide_in.reserve(std::min(CH, n + L - 1) + 8);
        for (size_t m0 = 0; m0 < n; m0 += per) {
            const size_t m1 = std::min(n, m0 + per), na = (m1 - m0) + L - 1;
            // xp[m0 + i], i < na: the first L of the stream come from hist
            size_t done = 0;
            if (m0 < L) {
                const size_t nh = std::min(na, L - m0);
                launch_f32_to_c32(hist[cur].p + m0, wide_in.p, (long)nh, s);
                done = nh;
            }
            if (done < na) launch_f32_to_c32(static_cast<const float*>(in) + (m0 + done - L), wide_in.p + done, (long)(na - done), s);
            size_t c2 = 0, p2 = 0, n2 = 0;
            wide->work_dev(wide_in.p, na, static_cast<cf*>(out) + m0, m1 - m0, &c2, &p2, &n2, s);
            if (p2 != m1 - m0) throw Error("Hilbert: the Complex filter stage disagrees on the output count");
        }
    } else if (!(skip_ok && launch_hilbert_skip(pl.L, par, Q, d_hq.p, src, static_cast<cf*>(out), (long)n, s, nanfix())))
        launch_hilbert(pl, d_tp.p, d_rev.p, src, static_cast<cf*>(out), (long)n, s, nanfix());
    prof_end(s);
    launch_vcopy_f32(src, (long)n, hist[cur ^ 1].p, (long)pl.L, s);            // :125
    cur ^= 1;
    *consumed = *produced = n; *need = 0;
    return RR_AGAIN;                                                            // :127
}

}  // namespace rr
