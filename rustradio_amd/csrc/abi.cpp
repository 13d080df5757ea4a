// abi.cpp — the extern "C" surface declared in include/rustradio_amd.h.
#include <cstring>
#include <functional>
#include <string>

#include "blocks.hpp"
#include "taps.hpp"

struct rr_block {
    std::unique_ptr<rr::Block> b;
};

namespace rr {
static thread_local std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
}  // namespace rr

template <class F> static rr_block* make_block(F&& f) {
    try {
        auto* h = new rr_block;
        h->b.reset(f());
        return h;
    } catch (const std::exception& e) {
        rr::set_last_error(e.what());
        return nullptr;
    }
}

extern "C" {

int rr_abi_version(void) { return RR_ABI_VERSION; }
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
int rr_hilbert_taps(const float* window, size_t ntaps, float* out) {
    std::vector<float> t;
    if (!rr::hilbert_taps(window, ntaps, t)) { rr::set_last_error("hilbert: window must have more than 1 tap"); return RR_ERR; }
    std::memcpy(out, t.data(), ntaps * sizeof(float));
    return 0;
}

rr_block* rr_fir_c32_create(const rr_c32* taps, size_t ntaps, size_t deci, int translate, float samp_rate, float freq) {
    return make_block([&] { return new rr::FirC32(taps, ntaps, deci, translate != 0, samp_rate, freq); });
}
rr_block* rr_fir_f32_create(const float* taps, size_t ntaps, size_t deci) {
    return make_block([&] { return new rr::FirF32(taps, ntaps, deci); });
}
rr_block* rr_fftfilter_create(const rr_c32* taps, size_t ntaps) {
    return make_block([&] { return new rr::FftFilter(taps, ntaps); });
}
rr_block* rr_fftfilter_float_create(const float* taps, size_t ntaps) {
    return make_block([&] { return new rr::FftFilterFloat(taps, ntaps); });
}
rr_block* rr_resampler_create(size_t interp, size_t deci, size_t elem_size) {
    return make_block([&] { return new rr::Resampler(interp, deci, elem_size); });
}
rr_block* rr_fm_chain_u8_create(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain,
                                int atan2_mode) {
    return make_block([&] { return new rr::FmChain(taps, ntaps, interp, deci, gain, atan2_mode, true); });
}
rr_block* rr_hilbert_fir_create(size_t hilbert_ntaps, int window, float window_parm, const rr_c32* taps, size_t ntaps,
                                size_t deci, int translate, float samp_rate, float freq) {
    return make_block([&] {
        return new rr::HilbertFir(hilbert_ntaps, window, window_parm, taps, ntaps, deci, translate != 0, samp_rate, freq);
    });
}
rr_block* rr_rtlsdr_decode_create(void) {
    return make_block([&] { return new rr::RtlSdrDecode(); });
}
rr_block* rr_quaddemod_create(float gain, int atan2_mode) {
    return make_block([&] { return new rr::QuadDemod(gain, atan2_mode); });
}
rr_block* rr_hilbert_create(size_t ntaps, int window, float window_parm) {
    return make_block([&] { return new rr::Hilbert(ntaps, window, window_parm); });
}
rr_block* rr_fm_chain_create(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain, int atan2_mode) {
    return make_block([&] { return new rr::FmChain(taps, ntaps, interp, deci, gain, atan2_mode); });
}
rr_block* rr_fm_multi_create(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float gain,
                             int atan2_mode) {
    return make_block([&] { return new rr::FmMulti(taps, nchan, ntaps, interp, deci, gain, atan2_mode); });
}
size_t rr_block_out_windows(const rr_block* b) { return b ? b->b->out_windows() : 0; }
void rr_block_destroy(rr_block* b) { delete b; }

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
size_t rr_block_in_elem_size(const rr_block* b) { return b ? b->b->in_es : 0; }
size_t rr_block_out_elem_size(const rr_block* b) { return b ? b->b->out_es : 0; }
int rr_block_sync(rr_block* b) {
    if (!b) return RR_ERR;
    try { b->b->sync(); return 0; } catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; }
}

int rr_block_set_profiling(rr_block* b, int on) {
    if (!b) return RR_ERR;
    b->b->prof_on = on != 0;
    return 0;
}
int rr_block_profile(rr_block* b, double* total_ms, size_t* launches, int reset) {
    if (!b) return RR_ERR;
    try { RR_HIP(hipSetDevice(b->b->device)); b->b->prof_read(total_ms, launches, reset != 0); return 0; }
    catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; }
}

/* measurement builds (make ABLATE=1): phase time stamps of one FftFilter tile; 0 in product builds */
int rr_debug_fft_stamps(unsigned long long* out16) {
    try { return rr::fft_read_stamps(out16); } catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; }
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
int rr_fir_set_rotator_mode(rr_block* b, int mode) {
    auto* f = b ? dynamic_cast<rr::FirC32*>(b->b.get()) : nullptr;
    if (!f && b) if (auto* hf = dynamic_cast<rr::HilbertFir*>(b->b.get())) f = hf->fir.get();
    if (!f || (mode != RR_ROT_MODEL && mode != RR_ROT_REPLAY)) { rr::set_last_error("rr_fir_set_rotator_mode: bad argument"); return RR_ERR; }
    f->rot_mode = mode;
    return 0;
}

}  // extern "C"
