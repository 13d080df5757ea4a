// taps.cpp — setup-time tap designers (host, f32), same formulas and operation order as
// /root/reference/src/window.rs:98-185 and /root/reference/src/fir.rs:594-680, so that a
// graph built against this library designs the same filters as one built against rustradio.
#include "taps.hpp"

#include <cmath>

namespace rr {

static const float kPi = static_cast<float>(3.14159265358979323846);   // window.rs:34 / fir.rs:628

float max_attenuation(int window) {   // window.rs:67-75
    if (window == 1) return 74.0f;    // Blackman
    if (window == 2) return 92.0f;    // BlackmanHarris
    return 53.0f;                     // Hamming / HammingParm
}

static std::vector<float> hamming(size_t n, float a0) {
    std::vector<float> w(n);
    if (n == 1) { w[0] = 1.0f; return w; }
    const float a1 = 1.0f - a0;
    const float m = static_cast<float>(n - 1);            // symmetric form (denominator n-1)
    for (size_t i = 0; i < n; i++) w[i] = a0 - a1 * std::cos(2.0f * kPi * static_cast<float>(i) / m);
    return w;
}
static std::vector<float> blackman(size_t n) {
    std::vector<float> w(n);
    if (n == 1) { w[0] = 1.0f; return w; }
    const float A = 0.16f;
    const float a0 = (1.0f - A) / 2.0f, a1 = 0.5f, a2 = A / 2.0f;
    const float m = static_cast<float>(n);                 // periodic form (denominator n)
    for (size_t i = 0; i < n; i++) {
        const float x = static_cast<float>(i);
        const float t1 = 2.0f * kPi * x / m, t2 = 4.0f * kPi * x / m;
        w[i] = a0 - a1 * std::cos(t1) + a2 * std::cos(t2);
    }
    return w;
}
static std::vector<float> blackman_harris(size_t n) {
    std::vector<float> w(n);
    if (n == 1) { w[0] = 1.0f; return w; }
    const float A0 = 0.35875f, A1 = 0.48829f, A2 = 0.14128f, A3 = 0.01168f;
    const float m = static_cast<float>(n);
    for (size_t i = 0; i < n; i++) {
        const float x = static_cast<float>(i);
        const float t1 = 2.0f * kPi * x / m, t2 = 4.0f * kPi * x / m, t3 = 6.0f * kPi * x / m;
        w[i] = A0 - A1 * std::cos(t1) + A2 * std::cos(t2) - A3 * std::cos(t3);
    }
    return w;
}

bool make_window(int window, float parm, size_t n, std::vector<float>& out) {
    if (n == 0) { out.clear(); return window >= 0 && window <= 3; }
    switch (window) {
    case 0: out = hamming(n, 25.0f / 46.0f); return true;   // window.rs:37
    case 1: out = blackman(n); return true;
    case 2: out = blackman_harris(n); return true;
    case 3: out = hamming(n, parm); return true;
    }
    return false;
}

size_t compute_ntaps(float samp_rate, float twidth, int window) {   // fir.rs:606-610
    const float a = max_attenuation(window);
    const size_t t = static_cast<size_t>(a * samp_rate / (22.0f * twidth));
    return (t & 1) == 0 ? t + 1 : t;
}

bool low_pass(float samp_rate, float cutoff, float twidth, int window, float parm, std::vector<float>& taps) {
    if (!(samp_rate > 0.0f) || !(cutoff > 0.0f) || !(twidth > 0.0f)) return false;   // fir.rs:623-625
    const size_t ntaps = compute_ntaps(samp_rate, twidth, window);
    std::vector<float> win;
    if (!make_window(window, parm, ntaps, win)) return false;
    const size_t mid = (ntaps - 1) / 2;
    const float fwt0 = 2.0f * kPi * cutoff / samp_rate;
    taps.resize(ntaps);
    for (size_t i = 0; i < ntaps; i++) {
        const long n = static_cast<long>(i) - static_cast<long>(mid);
        const float nf = static_cast<float>(n);
        taps[i] = n == 0 ? fwt0 / kPi * win[i] : (std::sin(nf * fwt0) / (nf * kPi)) * win[i];
    }
    float fmax = taps[mid];                                   // DC-gain normalisation, fir.rs:647-655
    for (size_t n = 1; n <= mid; n++) fmax += 2.0f * taps[n + mid];
    const float gain = 1.0f / fmax;
    for (auto& t : taps) t *= gain;
    return true;
}

bool hilbert_taps(const float* window, size_t ntaps, std::vector<float>& taps) {   // fir.rs:660-680
    if (ntaps < 2) return false;
    const size_t mid = (ntaps - 1) / 2;
    float gain = 0.0f;
    taps.assign(ntaps, 0.0f);
    for (size_t i = 1; i <= mid; i++) {
        if (i & 1) {
            const float x = 1.0f / static_cast<float>(i);
            taps[mid + i] = x * window[mid + i];
            taps[mid - i] = -x * window[mid - i];
            gain = taps[mid + i] - gain;
        }
    }
    gain = 1.0f / (2.0f * std::fabs(gain));
    for (auto& t : taps) t = gain * t;
    return true;
}

// multiband (fir.rs:552-590): ideal brick response sampled at ntaps points (mirrored), inverse DFT of that size,
// rotated by ntaps/2, windowed, scaled by 1/sqrt(ntaps).  The reference runs rustfft's f32 inverse of any size;
// this is a direct O(N^2) sum in double, rounded once (setup time; the reference marks the function untested).
bool multiband(const float* bands, size_t nbands, const float* window, size_t ntaps, std::vector<float>& re_im) {
    if (ntaps == 0) return false;
    std::vector<char> ideal(ntaps, 0);
    const float scale = (float)ntaps / 2.0f;
    for (size_t bi = 0; bi < nbands; bi++) {
        const size_t a = (size_t)std::floor(bands[2 * bi] * scale), b = (size_t)std::ceil(bands[2 * bi + 1] * scale);
        if (a > b || a > ntaps || b > ntaps) return false;
        for (size_t n = a; n < b; n++) { ideal[n] = 1; ideal[ntaps - n - 1] = 1; }
    }
    const float fscale = std::sqrt((float)ntaps);
    re_im.assign(2 * ntaps, 0.0f);
    for (size_t n = 0; n < ntaps; n++) {
        const size_t m = (n + ntaps - ntaps / 2) % ntaps;
        double re = 0.0, im = 0.0;
        for (size_t k = 0; k < ntaps; k++) {
            if (!ideal[k]) continue;
            const double ang = 2.0 * 3.14159265358979323846 * (double)((k * m) % ntaps) / (double)ntaps;
            re += std::cos(ang); im += std::sin(ang);
        }
        re_im[2 * n] = ((float)re * window[n]) / fscale;
        re_im[2 * n + 1] = ((float)im * window[n]) / fscale;
    }
    return true;
}

}  // namespace rr
