#pragma once
#include <cstddef>
#include <vector>

namespace rr {
float max_attenuation(int window);
bool make_window(int window, float parm, size_t n, std::vector<float>& out);
size_t compute_ntaps(float samp_rate, float twidth, int window);
bool low_pass(float samp_rate, float cutoff, float twidth, int window, float parm, std::vector<float>& taps);
bool hilbert_taps(const float* window, size_t ntaps, std::vector<float>& taps);
// multiband(bands, taps, window) (fir.rs:552-590): bands = pairs (low, high) in units of Nyquist; false = None
bool multiband(const float* bands, size_t nbands, const float* window, size_t ntaps, std::vector<float>& re_im);
}  // namespace rr
