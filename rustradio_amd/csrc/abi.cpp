// abi.cpp — the extern "C" surface declared in include/rustradio_amd.h.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>

#include "blocks.hpp"
#include "dstream.hpp"
#include "taps.hpp"

struct rr_block {
    std::unique_ptr<rr::Block> b;
    int tag_rule = RR_TAGS_DROP;      // rr_block_tag_rule: what the reference block(s) behind the handle do with tags
    size_t tag_param = 1;
};
// One ring + what makes it a two-ended stream between threads: every entry point takes `m` (two rings: both, deadlock-free),
// produce / consume / close wake `cv`.  closed[side] is the reference's Arc strong count dropping to 1
// (src/stream.rs:148-150,166-168): the shim sets it when it drops its WriteStream / ReadStream end.
struct rr_dstream {
    std::unique_ptr<rr::DStream> s;
    std::mutex m;
    std::condition_variable cv;
    bool closed[2] = {false, false};
    size_t id = 0;
};

namespace rr {
std::atomic<unsigned long long> g_kernel_launches{0};
static thread_local std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
static thread_local BuildOpts g_opts;
const BuildOpts& build_opts() { return g_opts; }
void set_build_opts(const BuildOpts* o) { g_opts = o ? *o : BuildOpts(); }
// the pending overrides belong to ONE create call: cleared when it returns, whether it succeeded or not
struct OptsScope { ~OptsScope() { set_build_opts(nullptr); } };
}  // namespace rr

template <class F> static rr_block* make_block(F&& f, int tag_rule = RR_TAGS_DROP, size_t tag_param = 1) {
    rr::OptsScope scope;
    try {
        std::unique_ptr<rr::Block> b(f());         // a throwing constructor leaks nothing
        if (const int v = rr::build_opts().host_in_staged) b->zero_copy_in = v < 0;   // (rr_build_opts: the block's default otherwise)
        auto* h = new rr_block;
        h->b = std::move(b);
        h->tag_rule = tag_rule;
        h->tag_param = tag_param;
        return h;
    } catch (const std::exception& e) {
        rr::set_last_error(e.what());
        return nullptr;
    } catch (...) {
        rr::set_last_error("non-standard exception");
        return nullptr;
    }
}

template <class F> static int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; } catch (...) { rr::set_last_error("non-standard exception"); return RR_ERR; }
}

extern "C" {

int rr_abi_version(void) { return RR_ABI_VERSION; }
int rr_next_create_options(const rr_build_opts* o) {
    if (!o) { rr::set_build_opts(nullptr); return 0; }
    rr::BuildOpts b;
    b.fir_path = o->fir_path; b.fir_prune = o->fir_prune; b.fir_half = o->fir_half; b.fir_cfg = o->fir_cfg_plus1 - 1;
    b.fft_log2f = o->fft_log2f; b.fft_no_split = o->fft_no_split; b.fftfloat_complex = o->fftfloat_complex;
    b.fm_full = o->fm_full; b.fm_poly = o->fm_poly; b.dstream_no_vmm = o->dstream_no_vmm;
    b.fir_poly = o->fir_poly; b.fft_nonfinite_tiles = o->fft_nonfinite_tiles; b.host_in_staged = o->host_in_staged;
    if (b.fir_path < 0 || b.fir_path > 2 || b.fir_cfg > 7 || (b.fft_log2f != 0 && (b.fft_log2f < 10 || b.fft_log2f > 14))) {
        rr::set_last_error("rr_next_create_options: value out of range");
        return RR_ERR;
    }
    rr::set_build_opts(&b);
    return 0;
}
const char* rr_last_error(void) { return rr::g_err.c_str(); }

int rr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int rr_set_device(int ordinal) {
    int n = rr_device_count();
    if (ordinal < 0 || ordinal >= n) { rr::set_last_error("rr_set_device: ordinal out of range"); return RR_ERR; }
    rr::set_thread_device(ordinal);
    return 0;
}

float rr_max_attenuation(int window) { return rr::max_attenuation(window); }
int rr_make_window(int window, float parm, size_t ntaps, float* out) {
    std::vector<float> w;
    if (!rr::make_window(window, parm, ntaps, w)) { rr::set_last_error("unknown window type"); return RR_ERR; }
    if (ntaps) std::memcpy(out, w.data(), ntaps * sizeof(float));
    return 0;
}
size_t rr_compute_ntaps(float samp_rate, float twidth, int window) { return rr::compute_ntaps(samp_rate, twidth, window); }
size_t rr_low_pass(float samp_rate, float cutoff, float twidth, int window, float parm, float* out, size_t cap) {
    std::vector<float> t;
    if (!rr::low_pass(samp_rate, cutoff, twidth, window, parm, t)) {
        rr::set_last_error("low_pass: samp_rate, cutoff and twidth must be > 0 and the window type valid");
        return 0;
    }
    for (size_t i = 0; i < t.size() && i < cap; i++) out[i] = t[i];
    return t.size();
}
size_t rr_low_pass_complex(float samp_rate, float cutoff, float twidth, int window, float parm, rr_c32* out, size_t cap) {
    std::vector<float> t;
    if (!rr::low_pass(samp_rate, cutoff, twidth, window, parm, t)) {
        rr::set_last_error("low_pass_complex: samp_rate, cutoff and twidth must be > 0 and the window type valid");
        return 0;
    }
    for (size_t i = 0; i < t.size() && i < cap; i++) out[i] = rr_c32{t[i], 0.0f};   // fir.rs:602
    return t.size();
}
int rr_multiband(const float* bands, size_t nbands, const float* window, size_t ntaps, rr_c32* out) {
    std::vector<float> t;
    if (!bands && nbands) { rr::set_last_error("multiband: null bands"); return RR_ERR; }
    if (!rr::multiband(bands, nbands, window, ntaps, t)) { rr::set_last_error("multiband: None (fir.rs:558-569)"); return RR_ERR; }
    std::memcpy(out, t.data(), ntaps * sizeof(rr_c32));
    return 0;
}
int rr_hilbert_taps(const float* window, size_t ntaps, float* out) {
    std::vector<float> t;
    if (!rr::hilbert_taps(window, ntaps, t)) { rr::set_last_error("hilbert: window must have more than 1 tap"); return RR_ERR; }
    std::memcpy(out, t.data(), ntaps * sizeof(float));
    return 0;
}

rr_block* rr_fir_c32_create(const rr_c32* taps, size_t ntaps, size_t deci, int translate, float samp_rate, float freq) {
    return make_block([&] { return new rr::FirC32(taps, ntaps, deci, translate != 0, samp_rate, freq); }, RR_TAGS_FORWARD, deci);
}
rr_block* rr_fir_f32_create(const float* taps, size_t ntaps, size_t deci) {
    return make_block([&] { return new rr::FirF32(taps, ntaps, deci); }, RR_TAGS_FORWARD, deci);
}
rr_block* rr_fftfilter_create(const rr_c32* taps, size_t ntaps) {
    return make_block([&] { auto* f = new rr::FftFilter(taps, ntaps); std::unique_ptr<rr::FftFilter> own(f); f->ref_blocks_on(taps); return own.release(); },
                      RR_TAGS_FORWARD, 1);
}
rr_block* rr_fftfilter_float_create(const float* taps, size_t ntaps) {
    return make_block([&] { return new rr::FftFilterFloat(taps, ntaps); }, RR_TAGS_FORWARD, 1);
}
rr_block* rr_resampler_create(size_t interp, size_t deci, size_t elem_size) {
    return make_block([&] { return new rr::Resampler(interp, deci, elem_size); });
}
rr_block* rr_fm_chain_u8_create(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain,
                                int atan2_mode) {
    return make_block([&] { return rr::make_fm_chain(taps, ntaps, interp, deci, gain, atan2_mode, true, nullptr, 0); });
}
rr_block* rr_hilbert_fir_create(size_t hilbert_ntaps, int window, float window_parm, const rr_c32* taps, size_t ntaps,
                                size_t deci, int translate, float samp_rate, float freq) {
    return make_block([&] {
        return new rr::HilbertFir(hilbert_ntaps, window, window_parm, taps, ntaps, deci, translate != 0, samp_rate, freq);
    }, RR_TAGS_FORWARD, deci);
}
rr_block* rr_fftstream_create(size_t size) {
    return make_block([&] { return new rr::FftStream(size); }, RR_TAGS_FRAMES, size);
}
int rr_fft_process(rr_block* b, const rr_c32* msg, size_t n, rr_c32* out) {
    auto* f = b ? dynamic_cast<rr::FftStream*>(b->b.get()) : nullptr;
    if (!f || !msg || !out) { rr::set_last_error("rr_fft_process: not an FftStream handle / null argument"); return RR_ERR; }
    if (n != f->size) {                                                   // fft.rs:46-52
        rr::set_last_error("FFT expected " + std::to_string(f->size) + " samples, got " + std::to_string(n));
        return RR_ERR;
    }
    size_t c = 0, p = 0, need = 0;
    const int st = rr_block_work(b, msg, n, out, n, &c, &p, &need);
    return st == RR_ERR ? RR_ERR : (p == n ? 0 : RR_ERR);
}
rr_block* rr_multiply_const_f32_create(float val) {
    return make_block([&] { return new rr::MultiplyConst(4, val, 0.0f); }, RR_TAGS_FORWARD, 1);
}
rr_block* rr_multiply_const_c32_create(float re, float im) {
    return make_block([&] { return new rr::MultiplyConst(8, re, im); }, RR_TAGS_FORWARD, 1);
}
rr_block* rr_fastfm_create(void) {
    return make_block([&] { return new rr::FastFM(); }, RR_TAGS_FORWARD, 1);
}
rr_block* rr_rtlsdr_decode_create(void) {
    return make_block([&] { return new rr::RtlSdrDecode(); });
}
rr_block* rr_quaddemod_create(float gain, int atan2_mode) {
    return make_block([&] { return new rr::QuadDemod(gain, atan2_mode); });
}
rr_block* rr_hilbert_create(size_t ntaps, int window, float window_parm) {
    return make_block([&] { return new rr::Hilbert(ntaps, window, window_parm); }, RR_TAGS_FORWARD, 1);
}
rr_block* rr_fm_chain_create(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain, int atan2_mode) {
    return make_block([&] { return rr::make_fm_chain(taps, ntaps, interp, deci, gain, atan2_mode, false, nullptr, 0); });
}
rr_block* rr_fir_fftfilter_create(const rr_c32* fir_taps, size_t fir_ntaps, const rr_c32* fft_taps, size_t fft_ntaps) {
    return make_block([&] {
        const std::vector<rr_c32> g = rr::FftFilter::composite(fir_taps, fir_ntaps, fft_taps, fft_ntaps);
        std::unique_ptr<rr::FftFilter> f(new rr::FftFilter(g.data(), g.size(), false, 14, false, fir_ntaps - 1));
        f->set_stage_taps(fir_taps, fir_ntaps, fft_taps, fft_ntaps);
        f->ref_blocks_on(g.data());
        return f.release();
    }, RR_TAGS_FORWARD, 1);
}
rr_block* rr_fir_fm_chain_create(const rr_c32* fir_taps, size_t fir_ntaps, const rr_c32* fft_taps, size_t fft_ntaps,
                                 size_t interp, size_t deci, float gain, int atan2_mode) {
    return make_block([&] {
        if (!fir_taps || fir_ntaps == 0) throw rr::Error("FirFilter: empty taps");
        return rr::make_fm_chain(fft_taps, fft_ntaps, interp, deci, gain, atan2_mode, false, fir_taps, fir_ntaps);
    });
}
rr_block* rr_audio_chain_create(const float* taps, size_t ntaps, size_t interp, size_t deci, float scale) {
    return make_block([&] { return rr::make_audio_chain(taps, ntaps, interp, deci, scale); });
}
rr_block* rr_fm_multi_create(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float gain,
                             int atan2_mode) {
    return make_block([&] { return rr::make_fm_multi(taps, nchan, ntaps, interp, deci, gain, atan2_mode, false); });
}
rr_block* rr_fm_multi_u8_create(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float gain,
                                int atan2_mode) {
    return make_block([&] { return rr::make_fm_multi(taps, nchan, ntaps, interp, deci, gain, atan2_mode, true); });
}
size_t rr_block_out_windows(const rr_block* b) { return b ? b->b->out_windows() : 0; }
void rr_block_destroy(rr_block* b) { try { delete b; } catch (...) {} }

static int guarded(rr_block* b, size_t* consumed, size_t* produced, size_t* need, const char* what,
                   const std::function<int()>& f) {
    size_t dummy = 0;
    (void)dummy;
    if (!b || !consumed || !produced || !need) { rr::set_last_error(std::string(what) + ": null argument"); return RR_ERR; }
    try {
        return f();
    } catch (const std::exception& e) {
        rr::set_last_error(std::string(what) + ": " + e.what());
        *consumed = *produced = *need = 0;
        return RR_ERR;
    } catch (...) {
        rr::set_last_error("non-standard exception");
        *consumed = *produced = *need = 0;
        return RR_ERR;
    }
}

int rr_block_work(rr_block* b, const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed,
                  size_t* produced, size_t* need) {
    return guarded(b, consumed, produced, need, "rr_block_work",
                   [&] { return b->b->work_host(in, in_len, out, out_cap, consumed, produced, need); });
}
int rr_block_work_dev(rr_block* b, const void* d_in, size_t in_len, void* d_out, size_t out_cap, size_t* consumed,
                      size_t* produced, size_t* need, void* hip_stream) {
    return guarded(b, consumed, produced, need, "rr_block_work_dev", [&] {
        RR_HIP(hipSetDevice(b->b->device));
        // the handle is used as given: NULL is HIP's default stream (what torch's default stream is),
        // NOT the block's private stream — anything else would run unordered against the caller's work
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        b->b->last_stream = s;
        return b->b->work_dev(d_in, in_len, d_out, out_cap, consumed, produced, need, s);
    });
}
int rr_block_eof(rr_block* b, int src_eof) { return b && b->b->eof(src_eof != 0) ? 1 : 0; }
const char* rr_block_name(const rr_block* b) { return b ? b->b->name : ""; }
int rr_block_tag_rule(const rr_block* b, size_t* param) {
    if (!b) { rr::set_last_error("rr_block_tag_rule: null handle"); return RR_ERR; }
    if (param) *param = b->tag_param;
    return b->tag_rule;
}
size_t rr_block_in_elem_size(const rr_block* b) { return b ? b->b->in_es : 0; }
size_t rr_block_out_elem_size(const rr_block* b) { return b ? b->b->out_es : 0; }
int rr_block_sync(rr_block* b) {
    if (!b) return RR_ERR;
    try { b->b->sync(); return 0; } catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; } catch (...) { rr::set_last_error("non-standard exception"); return RR_ERR; }
}

int rr_block_set_profiling(rr_block* b, int on) {
    if (!b) return RR_ERR;
    b->b->prof_on = on != 0;
    return 0;
}
int rr_block_profile(rr_block* b, double* total_ms, size_t* launches, int reset) {
    if (!b) return RR_ERR;
    try { RR_HIP(hipSetDevice(b->b->device)); b->b->prof_read(total_ms, launches, reset != 0); return 0; }
    catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; } catch (...) { rr::set_last_error("non-standard exception"); return RR_ERR; }
}

unsigned long long rr_debug_kernel_launches(void) { return rr::g_kernel_launches.load(std::memory_order_relaxed); }

/* measurement builds (make ABLATE=1): phase time stamps of one FftFilter tile; 0 in product builds */
int rr_debug_fft_stamps(unsigned long long* out16) {
    try { return rr::fft_read_stamps(out16); } catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; } catch (...) { rr::set_last_error("non-standard exception"); return RR_ERR; }
}

int rr_fftfilter_dims(const rr_block* b, size_t* fft_size, size_t* nsamples, size_t* gpu_fft_size) {
    if (!b) return RR_ERR;
    const rr::FftFilter* f = dynamic_cast<const rr::FftFilter*>(b->b.get());
    if (!f) {
        if (auto* ff = dynamic_cast<const rr::FftFilterFloat*>(b->b.get())) f = ff->inner.get();
    }
    if (!f) { rr::set_last_error("rr_fftfilter_dims: not an FftFilter"); return RR_ERR; }
    if (fft_size) *fft_size = f->fft_size;
    if (nsamples) *nsamples = f->nsamples;
    if (gpu_fft_size) *gpu_fft_size = (size_t)1 << f->log2f;
    return 0;
}
size_t rr_fir_fft_tile(const rr_block* b) {
    if (!b) return 0;
    const rr::FirC32* f = dynamic_cast<const rr::FirC32*>(b->b.get());
    if (f && f->poly) return 1024;                            // decimate-first tiles: 1024 output-rate positions
    if (f && f->prune) return (size_t)1 << f->prune->log2f;
    if (f && f->half_ok) return 2048;
    return f && f->fftk ? (size_t)1 << f->fftk->log2f : 0;
}
// ---- host memory registration -----------------------------------------------------------------------
int rr_host_register(void* ptr, size_t bytes) {
    if (!ptr || !bytes) { rr::set_last_error("rr_host_register: null / empty range"); return RR_ERR; }
    return guarded([&] { RR_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault)); rr::host_range_add(ptr, bytes); });
}
int rr_host_window_in_place(const void* ptr, size_t bytes) { return ptr && bytes && rr::device_view_of_host(ptr, bytes) ? 1 : 0; }
int rr_host_unregister(void* ptr) {
    if (!ptr) { rr::set_last_error("rr_host_unregister: null"); return RR_ERR; }
    return guarded([&] { rr::host_range_remove(ptr); RR_HIP(hipHostUnregister(ptr)); });
}

// ---- device-resident streams ------------------------------------------------------------------------
rr_dstream* rr_dstream_create(size_t elem_size, size_t capacity_bytes) {
    rr::OptsScope scope;
    try {
        static std::atomic<size_t> next_id{1};
        std::unique_ptr<rr::DStream> d(new rr::DStream(elem_size, capacity_bytes));
        auto* h = new rr_dstream;
        h->s = std::move(d);
        h->id = next_id++;
        return h;
    } catch (const std::exception& e) {
        rr::set_last_error(e.what());
        return nullptr;
    } catch (...) {
        rr::set_last_error("non-standard exception");
        return nullptr;
    }
}
void rr_dstream_destroy(rr_dstream* s) { try { delete s; } catch (...) {} }
size_t rr_dstream_capacity(const rr_dstream* s) { return s ? s->s->cap : 0; }
int rr_dstream_is_double_mapped(const rr_dstream* s) { return s && s->s->vmm ? 1 : 0; }
size_t rr_dstream_id(const rr_dstream* s) { return s ? s->id : 0; }
size_t rr_dstream_read_buf(rr_dstream* s, const void** dev_ptr) {
    if (!s) return 0;
    std::lock_guard<std::mutex> g(s->m);
    if (dev_ptr) *dev_ptr = s->s->read_ptr();
    return s->s->used();
}
size_t rr_dstream_write_buf(rr_dstream* s, void** dev_ptr, void* hip_stream) {
    if (!s) return 0;
    std::lock_guard<std::mutex> g(s->m);
    try {
        if (!dev_ptr) return s->s->free();         // a count only: nothing is about to be written
        RR_HIP(hipSetDevice(s->s->device));        // write_ptr may enqueue the fallback ring's move
        s->s->will_write(static_cast<hipStream_t>(hip_stream));   // the caller writes the window on this stream
        *dev_ptr = s->s->write_ptr(static_cast<hipStream_t>(hip_stream));
        return s->s->free();
    } catch (const std::exception& e) {
        rr::set_last_error(e.what());
        return 0;
    } catch (...) {
        rr::set_last_error("non-standard exception");
        return 0;
    }
}
int rr_dstream_consume(rr_dstream* s, size_t n) {
    if (!s) { rr::set_last_error("null dstream"); return RR_ERR; }
    std::lock_guard<std::mutex> g(s->m);
    const int rc = guarded([&] { s->s->consume(n); });
    if (n) s->cv.notify_all();
    return rc;
}
int rr_dstream_produce(rr_dstream* s, size_t n) {
    if (!s) { rr::set_last_error("null dstream"); return RR_ERR; }
    std::lock_guard<std::mutex> g(s->m);
    const int rc = guarded([&] { s->s->produce(n); });
    if (n) s->cv.notify_all();
    return rc;
}
int rr_dstream_close(rr_dstream* s, int side) {
    if (!s || (side != RR_SIDE_WRITER && side != RR_SIDE_READER)) { rr::set_last_error("rr_dstream_close: bad argument"); return RR_ERR; }
    std::lock_guard<std::mutex> g(s->m);
    s->closed[side] = true;
    s->cv.notify_all();
    return 0;
}
int rr_dstream_closed(rr_dstream* s, int side) {
    if (!s || (side != RR_SIDE_WRITER && side != RR_SIDE_READER)) return 1;
    std::lock_guard<std::mutex> g(s->m);
    return s->closed[side] ? 1 : 0;
}
size_t rr_dstream_wait(rr_dstream* s, int side, size_t need, unsigned timeout_ms, int* never) {
    if (never) *never = 1;
    if (!s || (side != RR_SIDE_WRITER && side != RR_SIDE_READER)) return 0;
    std::unique_lock<std::mutex> g(s->m);
    // the READER waits for samples and gives up when the writer is gone; the WRITER waits for room and gives up when
    // the reader is gone (ReadStream::wait_for_read / WriteStream::wait_for_write, src/stream.rs:222-224,311-313)
    const int other = side == RR_SIDE_READER ? RR_SIDE_WRITER : RR_SIDE_READER;
    auto have = [&] { return side == RR_SIDE_READER ? s->s->used() : s->s->free(); };
    s->cv.wait_for(g, std::chrono::milliseconds(timeout_ms), [&] { return have() >= need || s->closed[other]; });
    const size_t n = have();
    if (never) *never = (n < need && s->closed[other]) ? 1 : 0;
    return n;
}
int rr_dstream_copy_in(rr_dstream* s, size_t offset, const void* host, size_t n, void* hip_stream) {
    if (!s) { rr::set_last_error("null dstream"); return RR_ERR; }
    auto st = static_cast<hipStream_t>(hip_stream);
    int rc;
    {
        std::lock_guard<std::mutex> g(s->m);
        rc = guarded([&] {
            rr::DStream& d = *s->s;
            RR_HIP(hipSetDevice(d.device));
            d.will_write(st);
            unsigned char* w = static_cast<unsigned char*>(d.write_ptr(st));
            if (offset + n > d.free()) throw rr::Error("dstream copy_in: beyond the write window");
            // a window of a ring the shim registered (zero-copy admitted): a copy kernel reads it in place over PCIe, 3x the
            // rate hipMemcpyAsync gets out of such a range on this platform (kernels_misc.hip launch_copy_bytes)
            void* hv = n ? rr::device_view_of_host(host, n * d.es) : nullptr;
            if (hv) rr::launch_copy_bytes(hv, w + offset * d.es, n * d.es, st);
            else if (n && rr::host_range_registered(host, n * d.es)) RR_HIP(hipMemcpyAsync(w + offset * d.es, host, n * d.es, hipMemcpyHostToDevice, st));
            else if (n) rr::stage_upload_sync(w + offset * d.es, host, n * d.es, st);     // pageable: stage.hpp
        });
    }
    if (rc != 0 || n == 0) return rc;
    // `host` is the caller's to reuse on return: a source ring's window is consumed right after this call and its writer
    // overwrites it.  From pageable memory the runtime has already staged the bytes; from a page-locked (rr_host_register'd)
    // ring the DMA is still reading, so wait for it -- outside the lock, the reader may go on consuming meanwhile.
    return guarded([&] { RR_HIP(hipStreamSynchronize(st)); });
}
int rr_dstream_copy_out(rr_dstream* s, size_t offset, void* host, size_t n, void* hip_stream) {
    if (!s) { rr::set_last_error("null dstream"); return RR_ERR; }
    auto st = static_cast<hipStream_t>(hip_stream);
    int rc;
    {
        std::lock_guard<std::mutex> g(s->m);
        rc = guarded([&] {
            rr::DStream& d = *s->s;
            if (offset + n > d.used()) throw rr::Error("dstream copy_out: beyond the read window");
            RR_HIP(hipSetDevice(d.device));
            d.will_read(st);
            void* hv = n ? rr::device_view_of_host(host, n * d.es) : nullptr;
            const unsigned char* rp = static_cast<const unsigned char*>(d.read_ptr()) + offset * d.es;
            if (hv) rr::launch_copy_bytes(rp, hv, n * d.es, st);              // (see copy_in)
            else if (n && rr::host_range_registered(host, n * d.es)) RR_HIP(hipMemcpyAsync(host, rp, n * d.es, hipMemcpyDeviceToHost, st));
            else if (n) rr::stage_download_sync(host, rp, n * d.es, st);                   // pageable: stage.hpp
        });
    }
    if (rc != 0) return rc;
    // outside the lock: the writer may go on filling the free space while this window crosses the bus (the read window
    // is not released before the caller's rr_dstream_consume)
    return guarded([&] { RR_HIP(hipStreamSynchronize(st)); });          // the host may read `host` on return
}
int rr_dstream_copy(rr_dstream* dst, size_t dst_offset, rr_dstream* src, size_t src_offset, size_t n, void* hip_stream) {
    if (!dst || !src || dst == src) { rr::set_last_error("rr_dstream_copy: null dstream / same ring on both sides"); return RR_ERR; }
    std::scoped_lock g(src->m, dst->m);
    return guarded([&] {
        rr::DStream& d = *dst->s;
        rr::DStream& r = *src->s;
        auto st = static_cast<hipStream_t>(hip_stream);
        if (d.es != r.es) throw rr::Error("dstream copy: element sizes differ");
        if (d.device != r.device) throw rr::Error("dstream copy: rings on different devices");
        if (src_offset + n > r.used()) throw rr::Error("dstream copy: beyond the read window");
        RR_HIP(hipSetDevice(d.device));
        r.will_read(st);
        d.will_write(st);
        unsigned char* w = static_cast<unsigned char*>(d.write_ptr(st));
        if (dst_offset + n > d.free()) throw rr::Error("dstream copy: beyond the write window");
        if (n) RR_HIP(hipMemcpyAsync(w + dst_offset * d.es, static_cast<const unsigned char*>(r.read_ptr()) + src_offset * r.es, n * d.es,
                                     hipMemcpyDeviceToDevice, st));
    });
}
int rr_block_work_streams(rr_block* b, rr_dstream* src, rr_dstream* dst, size_t* consumed, size_t* produced,
                          size_t* need, void* hip_stream) {
    size_t c = 0, p = 0, nd = 0;
    if (!b || !src || !dst || src == dst) { rr::set_last_error("rr_block_work_streams: null argument / same ring on both sides"); return RR_ERR; }
    if (b->b->out_windows() != 1) { rr::set_last_error("rr_block_work_streams: single-output blocks only"); return RR_ERR; }
    if (src->s->es != b->b->in_es || dst->s->es != b->b->out_es) {
        rr::set_last_error("rr_block_work_streams: stream element size does not match the block");
        return RR_ERR;
    }
    int st = RR_ERR;
    // both rings for the whole call: windows, enqueue and the consume / produce are one step to every other thread
    std::scoped_lock g(src->m, dst->m);
    const int rc = guarded([&] {
        auto hs = static_cast<hipStream_t>(hip_stream);
        RR_HIP(hipSetDevice(dst->s->device));
        src->s->will_read(hs);
        dst->s->will_write(hs);
        void* out = dst->s->write_ptr(hs);
        st = rr_block_work_dev(b, src->s->read_ptr(), src->s->used(), out, dst->s->free(), &c, &p, &nd, hip_stream);
        if (st == RR_ERR) return;
        src->s->consume(c);
        dst->s->produce(p);
    });
    if (c) src->cv.notify_all();
    if (p) dst->cv.notify_all();
    if (consumed) *consumed = c;
    if (produced) *produced = p;
    if (need) *need = nd;
    return rc == 0 ? st : RR_ERR;
}

int rr_fir_set_rotator_mode(rr_block* b, int mode) {
    auto* f = b ? dynamic_cast<rr::FirC32*>(b->b.get()) : nullptr;
    if (!f && b) if (auto* hf = dynamic_cast<rr::HilbertFir*>(b->b.get())) f = hf->fir.get();
    if (!f || (mode != RR_ROT_MODEL && mode != RR_ROT_REPLAY && mode != RR_ROT_REPLAY_DEVICE && mode != RR_ROT_REPLAY_HOST)) { rr::set_last_error("rr_fir_set_rotator_mode: bad argument"); return RR_ERR; }
    f->rot_mode = mode;
    return 0;
}

}  // extern "C"
