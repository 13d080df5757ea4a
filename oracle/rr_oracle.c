/*
 * rr_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See rr_oracle.h for scope, pinning status and the usage rule.
 *
 * Restates, in the reference's operation order and in strict IEEE f32:
 *   /root/reference/src/window.rs, fir.rs, fft_filter.rs,
 *   rational_resampler.rs, quadrature_demod.rs, hilbert.rs
 * Complex arithmetic follows num-complex 0.4.6 (Cargo.lock:1645):
 *   (a*b).re = a.re*b.re - a.im*b.im ; (a*b).im = a.re*b.im + a.im*b.re
 * with no FMA (rustc never contracts) — hence -ffp-contract=off in the Makefile.
 */
#include "rr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* src/stream.rs:105 DEFAULT_STREAM_SIZE (bytes). */
#define ORC_DEFAULT_STREAM_SIZE 4096000u

static char g_err[256];
const char *orc_last_error(void) { return g_err; }
static void set_err(const char *m) { snprintf(g_err, sizeof g_err, "%s", m); }

/* ---- num-complex arithmetic -------------------------------------------------- */
static inline orc_c32 c_mul(orc_c32 a, orc_c32 b) {
    orc_c32 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline orc_c32 c_add(orc_c32 a, orc_c32 b) {
    orc_c32 r = { a.re + b.re, a.im + b.im };
    return r;
}
static inline orc_c32 c_sub(orc_c32 a, orc_c32 b) {
    orc_c32 r = { a.re - b.re, a.im - b.im };
    return r;
}

/* ---- windows: src/window.rs --------------------------------------------------- */
/* window.rs:34  const PI: Float = std::f64::consts::PI as Float */
static const float PI_F = (float)3.14159265358979323846;

float orc_max_attenuation(int wtype) { /* window.rs:67-75 */
    switch (wtype) {
    case ORC_WIN_BLACKMAN: return 74.0f;
    case ORC_WIN_BLACKMAN_HARRIS: return 92.0f;
    default: return 53.0f;
    }
}

static void win_hamming(size_t ntaps, float a0, float *out) { /* window.rs:98-112 */
    if (ntaps == 0) return;
    if (ntaps == 1) { out[0] = 1.0f; return; }
    float a1 = 1.0f - a0;
    float m = (float)(ntaps - 1);
    for (size_t n = 0; n < ntaps; n++)
        out[n] = a0 - a1 * cosf(2.0f * PI_F * (float)n / m);
}

static void win_blackman(size_t mm, float *out) { /* window.rs:117-154 */
    const float A = 0.16f;
    if (mm == 0) return;
    if (mm == 1) { out[0] = 1.0f; return; }
    for (size_t k = 0; k < mm; k++) {
        float n = (float)k, m = (float)mm;
        float a0 = (1.0f - A) / 2.0f, a1 = 0.5f, a2 = A / 2.0f;
        float t1 = 2.0f * PI_F * n / m;
        float t2 = 4.0f * PI_F * n / m;
        out[k] = a0 - a1 * cosf(t1) + a2 * cosf(t2);
    }
}

static void win_blackman_harris(size_t mm, float *out) { /* window.rs:159-185 */
    const float A0 = 0.35875f, A1 = 0.48829f, A2 = 0.14128f, A3 = 0.01168f;
    if (mm == 0) return;
    if (mm == 1) { out[0] = 1.0f; return; }
    for (size_t k = 0; k < mm; k++) {
        float n = (float)k, m = (float)mm;
        float t1 = 2.0f * PI_F * n / m;
        float t2 = 4.0f * PI_F * n / m;
        float t3 = 6.0f * PI_F * n / m;
        out[k] = A0 - A1 * cosf(t1) + A2 * cosf(t2) - A3 * cosf(t3);
    }
}

int orc_make_window(int wtype, float parm, size_t ntaps, float *out) { /* window.rs:79-86 */
    switch (wtype) {
    case ORC_WIN_HAMMING: win_hamming(ntaps, 25.0f / 46.0f, out); return 0; /* window.rs:37 */
    case ORC_WIN_HAMMING_PARM: win_hamming(ntaps, parm, out); return 0;
    case ORC_WIN_BLACKMAN: win_blackman(ntaps, out); return 0;
    case ORC_WIN_BLACKMAN_HARRIS: win_blackman_harris(ntaps, out); return 0;
    }
    set_err("unknown window type");
    return -1;
}

/* ---- tap designers: src/fir.rs:594-680 ---------------------------------------- */
size_t orc_compute_ntaps(float samp_rate, float twidth, int wtype) { /* fir.rs:606-610 */
    float a = orc_max_attenuation(wtype);
    size_t t = (size_t)(a * samp_rate / (22.0f * twidth));
    return (t & 1) == 0 ? t + 1 : t;
}

size_t orc_low_pass(float samp_rate, float cutoff, float twidth, int wtype, float parm,
                    float *out, size_t cap) { /* fir.rs:617-656 */
    if (!(samp_rate > 0.0f) || !(cutoff > 0.0f) || !(twidth > 0.0f)) {
        set_err("low_pass: arguments must be > 0");
        return 0;
    }
    const float pi = PI_F;
    size_t ntaps = orc_compute_ntaps(samp_rate, twidth, wtype);
    float *taps = (float *)malloc(sizeof(float) * ntaps);
    orc_make_window(wtype, parm, ntaps, taps); /* taps[] holds the window first */
    size_t m = (ntaps - 1) / 2;
    float fwt0 = 2.0f * pi * cutoff / samp_rate;
    for (size_t nm = 0; nm < ntaps; nm++) {
        float win = taps[nm];
        long n = (long)nm - (long)m;
        float nf = (float)n;
        if (n == 0) taps[nm] = fwt0 / pi * win;
        else taps[nm] = (sinf(nf * fwt0) / (nf * pi)) * win;
    }
    float fmax = taps[m];
    for (size_t n = 1; n <= m; n++) fmax += 2.0f * taps[n + m];
    float gain = 1.0f / fmax;
    for (size_t i = 0; i < ntaps; i++) {
        float v = taps[i] * gain;
        if (i < cap) out[i] = v;
    }
    free(taps);
    return ntaps;
}

void orc_hilbert_taps(const float *window, size_t ntaps, float *out) { /* fir.rs:660-680 */
    size_t mid = (ntaps - 1) / 2;
    float gain = 0.0f;
    for (size_t i = 0; i < ntaps; i++) out[i] = 0.0f;
    for (size_t i = 1; i <= mid; i++) {
        if (i & 1) {
            float x = 1.0f / (float)i;
            out[mid + i] = x * window[mid + i];
            out[mid - i] = -x * window[mid - i];
            gain = out[mid + i] - gain;
        } else {
            out[mid + i] = 0.0f;
            out[mid - i] = 0.0f;
        }
    }
    gain = 1.0f / (2.0f * fabsf(gain));
    for (size_t i = 0; i < ntaps; i++) out[i] = gain * out[i];
}

/* ---- Fir<T>: src/fir.rs:150-198 ------------------------------------------------ */
/* Fir::new reverses the taps (fir.rs:160); Fir::filter folds left to right from
 * T::default(): acc = acc + tap_rev[j] * input[j] (fir.rs:173-176). */
static inline orc_c32 fir_c32_one(const orc_c32 *rev, size_t ntaps, const orc_c32 *in) {
    orc_c32 acc = { 0.0f, 0.0f };
    for (size_t j = 0; j < ntaps; j++) acc = c_add(acc, c_mul(rev[j], in[j]));
    return acc;
}
static inline float fir_f32_one(const float *rev, size_t ntaps, const float *in) {
    float acc = 0.0f;
    for (size_t j = 0; j < ntaps; j++) acc = acc + rev[j] * in[j];
    return acc;
}

void orc_fir_c32_n(const orc_c32 *taps, size_t ntaps, size_t deci, const orc_c32 *in,
                   orc_c32 *out, size_t n_out) { /* fir.rs:192-197 */
    orc_c32 *rev = (orc_c32 *)malloc(sizeof(orc_c32) * ntaps);
    for (size_t j = 0; j < ntaps; j++) rev[j] = taps[ntaps - 1 - j];
    for (size_t i = 0; i < n_out; i++) out[i] = fir_c32_one(rev, ntaps, in + i * deci);
    free(rev);
}
void orc_fir_f32_n(const float *taps, size_t ntaps, size_t deci, const float *in, float *out,
                   size_t n_out) {
    float *rev = (float *)malloc(sizeof(float) * ntaps);
    for (size_t j = 0; j < ntaps; j++) rev[j] = taps[ntaps - 1 - j];
    for (size_t i = 0; i < n_out; i++) out[i] = fir_f32_one(rev, ntaps, in + i * deci);
    free(rev);
}

/* ---- FFT (stand-in for rustfft 6.4.1; see header) ------------------------------ */
/* Stockham autosort, radix-4 stages with a radix-2 tail.  Twiddles are computed
 * in f64 and rounded once to f32, as rustfft's twiddles::compute_twiddle does. */
typedef struct {
    size_t n;
    orc_c32 *tw_fwd; /* e^{-2 pi i k / n}, k < n */
    orc_c32 *tw_inv; /* e^{+2 pi i k / n} */
    orc_c32 *scratch;
} orc_fftplan;

static orc_fftplan *fftplan_new(size_t n) {
    orc_fftplan *p = (orc_fftplan *)calloc(1, sizeof *p);
    p->n = n;
    p->tw_fwd = (orc_c32 *)malloc(sizeof(orc_c32) * n);
    p->tw_inv = (orc_c32 *)malloc(sizeof(orc_c32) * n);
    p->scratch = (orc_c32 *)malloc(sizeof(orc_c32) * n);
    for (size_t k = 0; k < n; k++) {
        double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
        p->tw_fwd[k].re = (float)cos(ang);
        p->tw_fwd[k].im = (float)sin(ang);
        p->tw_inv[k].re = (float)cos(-ang);
        p->tw_inv[k].im = (float)sin(-ang);
    }
    return p;
}
static void fftplan_free(orc_fftplan *p) {
    if (!p) return;
    free(p->tw_fwd); free(p->tw_inv); free(p->scratch); free(p);
}

static void fftplan_run(orc_fftplan *pl, orc_c32 *buf, int inverse) {
    const size_t N = pl->n;
    const orc_c32 *tw = inverse ? pl->tw_inv : pl->tw_fwd;
    orc_c32 *x = buf, *y = pl->scratch;
    size_t n = N, s = 1;
    while (n > 1) {
        if (n == 2) {
            for (size_t q = 0; q < s; q++) {
                orc_c32 a = x[q], b = x[q + s];
                y[q] = c_add(a, b);
                y[q + s] = c_sub(a, b);
            }
            n = 1; s *= 2;
        } else {
            const size_t n1 = n / 4, n2 = n / 2, n3 = n1 + n2;
            const size_t tstep = N / n;
            for (size_t p = 0; p < n1; p++) {
                const orc_c32 w1 = tw[p * tstep], w2 = tw[2 * p * tstep], w3 = tw[3 * p * tstep];
                for (size_t q = 0; q < s; q++) {
                    const orc_c32 a = x[q + s * p], b = x[q + s * (p + n1)];
                    const orc_c32 c = x[q + s * (p + n2)], d = x[q + s * (p + n3)];
                    const orc_c32 apc = c_add(a, c), amc = c_sub(a, c);
                    const orc_c32 bpd = c_add(b, d), bmd = c_sub(b, d);
                    /* forward: -j*(b-d) ; inverse: +j*(b-d) */
                    orc_c32 jbmd;
                    if (!inverse) { jbmd.re = bmd.im; jbmd.im = -bmd.re; }
                    else          { jbmd.re = -bmd.im; jbmd.im = bmd.re; }
                    y[q + s * (4 * p + 0)] = c_add(apc, bpd);
                    y[q + s * (4 * p + 1)] = c_mul(w1, c_add(amc, jbmd));
                    y[q + s * (4 * p + 2)] = c_mul(w2, c_sub(apc, bpd));
                    y[q + s * (4 * p + 3)] = c_mul(w3, c_sub(amc, jbmd));
                }
            }
            n /= 4; s *= 4;
        }
        orc_c32 *t = x; x = y; y = t;
    }
    if (x != buf) memcpy(buf, x, sizeof(orc_c32) * N);
}

void orc_fft(orc_c32 *buf, size_t n, int inverse) {
    orc_fftplan *p = fftplan_new(n);
    fftplan_run(p, buf, inverse);
    fftplan_free(p);
}

/* ---- fast-math 0.1.1 atan2 (unpinned restatement; see header) ------------------ */
static inline float flip_sign_nonnan(float v, float sign_src) {
    union { float f; uint32_t u; } a, b;
    a.f = v; b.f = sign_src;
    a.u ^= (b.u & 0x80000000u);
    return a.f;
}
static inline float fm_atan_raw(float x) {
    const float N2 = 0.273f;
    const float FRAC_PI_4 = 0.78539816339744830962f;
    return (FRAC_PI_4 + N2 - N2 * fabsf(x)) * x;
}
float orc_fast_atan2(float y, float x) {
    const float PI32 = 3.14159265358979323846f, FRAC_PI_2 = 1.57079632679489661923f;
    if (fabsf(y) < fabsf(x)) {
        float bias = x > 0.0f ? 0.0f : PI32;
        return flip_sign_nonnan(bias, y) + fm_atan_raw(y / x);
    } else if (x == 0.0f) {
        if (y == 0.0f) return 0.0f; /* pinned by quad_nulls, quadrature_demod.rs:211-219 */
        return flip_sign_nonnan(FRAC_PI_2, y);
    } else {
        return flip_sign_nonnan(FRAC_PI_2, y) - fm_atan_raw(x / y);
    }
}

int orc_multiband(const float *bands, size_t nbands, const float *window, size_t ntaps, orc_c32 *out) {
    if (ntaps == 0) return -1;                                                   /* :558-560 */
    double *ideal = (double *)calloc(ntaps, sizeof(double));
    const float scale = (float)ntaps / 2.0f;                                     /* :563 */
    for (size_t bi = 0; bi < nbands; bi++) {
        const size_t a = (size_t)floorf(bands[2 * bi] * scale), b = (size_t)ceilf(bands[2 * bi + 1] * scale);   /* :565-566 */
        if (a > b || a > ntaps || b > ntaps) { free(ideal); return -1; }         /* :567-569 */
        for (size_t n = a; n < b; n++) { ideal[n] = 1.0; ideal[ntaps - n - 1] = 1.0; }   /* :570-573 */
    }
    const float fscale = sqrtf((float)ntaps);                                    /* :580 */
    for (size_t n = 0; n < ntaps; n++) {
        /* ifft (unnormalised, +i) then rotate_right(taps / 2): out[n] = ifft[(n + N - N/2) % N]   (:575-579) */
        const size_t m = (n + ntaps - ntaps / 2) % ntaps;
        double re = 0.0, im = 0.0;
        for (size_t k = 0; k < ntaps; k++) {
            if (ideal[k] == 0.0) continue;
            const double ang = 2.0 * 3.14159265358979323846 * (double)((k * m) % ntaps) / (double)ntaps;
            re += cos(ang); im += sin(ang);
        }
        /* v * window[n] / Complex(scale, 0)   (:584-586) */
        out[n].re = ((float)re * window[n]) / fscale;
        out[n].im = ((float)im * window[n]) / fscale;
    }
    free(ideal);
    return 0;
}

/* ---- streaming blocks ----------------------------------------------------------- */
enum { K_FIR_C32, K_FIR_F32, K_FFTFILT, K_FFTFILT_F, K_RESAMP, K_QUAD, K_HILBERT, K_RTLSDR, K_MULC_F, K_MULC_C, K_FASTFM, K_FFTSTREAM };

struct orc_block {
    int kind;
    size_t in_es, out_es;
    /* FIR */
    size_t ntaps, deci;
    orc_c32 *rev_c;   /* reversed (and, if translating, pre-rotated) taps */
    orc_c32 *taps_c;  /* caller-order taps after pre-rotation */
    float *rev_f;
    int rot_on;
    orc_c32 rot_phase, rot_step;
    /* FftFilter */
    size_t fft_size, nsamples;
    orc_c32 *buf; size_t buf_len;
    orc_c32 *tail;
    orc_c32 *taps_fft;
    orc_fftplan *plan;
    /* FftFilterFloat: inner streams (fft_filter.rs:373-374) */
    orc_block *inner;
    orc_c32 *inner_in; size_t inner_in_len;   /* samples waiting in inner_in stream */
    orc_c32 *inner_out; size_t inner_out_len; /* samples waiting in inner_out stream */
    size_t inner_cap;
    /* Resampler */
    int64_t r_interp, r_deci, counter;
    int has_pending;
    unsigned char pending[16];
    /* QuadDemod */
    float gain; int atan2_mode;
    /* Hilbert */
    float *history;
    /* MultiplyConst / FastFM */
    orc_c32 mval;
    orc_c32 q1, q2;
};

size_t orc_block_in_elem_size(const orc_block *b) { return b->in_es; }
size_t orc_block_out_elem_size(const orc_block *b) { return b->out_es; }

orc_block *orc_fir_c32_new(const orc_c32 *taps, size_t ntaps, size_t deci, int translate,
                           float samp_rate, float freq) {
    if (ntaps == 0) { set_err("FirFilter: empty taps (fir.rs:372 assert)"); return NULL; }
    if (deci == 0) { set_err("FirFilter: deci 0 (fir.rs:319 assert)"); return NULL; }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_FIR_C32; b->in_es = b->out_es = sizeof(orc_c32);
    b->ntaps = ntaps; b->deci = deci;
    b->taps_c = (orc_c32 *)malloc(sizeof(orc_c32) * ntaps);
    memcpy(b->taps_c, taps, sizeof(orc_c32) * ntaps);
    if (translate) { /* ComplexFrequencyTranslator::new_translator, fir.rs:430-462 */
        if (!(samp_rate > 0.0f)) { set_err("translate: samp_rate <= 0 (fir.rs:436)"); orc_block_free(b); return NULL; }
        if (freq != 0.0f) { /* fir.rs:438-440 */
            double input_step = 2.0 * 3.14159265358979323846 * (double)freq / (double)samp_rate;
            orc_c32 tap_step = { (float)cos(input_step), (float)sin(input_step) };
            orc_c32 phase = { 1.0f, 0.0f };
            for (size_t k = 0; k < ntaps; k++) { /* fir.rs:446-449 */
                b->taps_c[k] = c_mul(b->taps_c[k], phase);
                phase = c_mul(phase, tap_step);
            }
            double first_output_phase = -input_step * (double)(ntaps - 1);
            double output_step = -input_step * (double)deci;
            b->rot_on = 1;
            b->rot_phase.re = (float)cos(first_output_phase);
            b->rot_phase.im = (float)sin(first_output_phase);
            b->rot_step.re = (float)cos(output_step);
            b->rot_step.im = (float)sin(output_step);
        }
    }
    b->rev_c = (orc_c32 *)malloc(sizeof(orc_c32) * ntaps);
    for (size_t j = 0; j < ntaps; j++) b->rev_c[j] = b->taps_c[ntaps - 1 - j]; /* fir.rs:160 */
    return b;
}

orc_block *orc_fir_f32_new(const float *taps, size_t ntaps, size_t deci) {
    if (ntaps == 0) { set_err("FirFilter: empty taps"); return NULL; }
    if (deci == 0) { set_err("FirFilter: deci 0"); return NULL; }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_FIR_F32; b->in_es = b->out_es = sizeof(float);
    b->ntaps = ntaps; b->deci = deci;
    b->rev_f = (float *)malloc(sizeof(float) * ntaps);
    for (size_t j = 0; j < ntaps; j++) b->rev_f[j] = taps[ntaps - 1 - j];
    return b;
}

size_t orc_fir_get_taps(const orc_block *b, orc_c32 *out, size_t cap) {
    if (b->kind != K_FIR_C32) return 0;
    for (size_t i = 0; i < b->ntaps && i < cap; i++) out[i] = b->taps_c[i];
    return b->ntaps;
}
void orc_fir_get_rotator(const orc_block *b, orc_c32 *phase, orc_c32 *step, int *enabled) {
    *phase = b->rot_phase; *step = b->rot_step; *enabled = b->rot_on;
}

static size_t calc_fft_size(size_t from) { /* fft_filter.rs:36-42 */
    size_t n = 1;
    while (n < from) n <<= 1;
    return 2 * n;
}

orc_block *orc_fftfilter_new(const orc_c32 *taps, size_t ntaps) { /* fft_filter.rs:144-169, 259-278 */
    if (ntaps == 0) { set_err("FftFilter: empty taps (fft_filter.rs:146 assert)"); return NULL; }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_FFTFILT; b->in_es = b->out_es = sizeof(orc_c32);
    b->ntaps = ntaps;
    b->fft_size = calc_fft_size(ntaps);
    b->nsamples = b->fft_size - ntaps;
    b->plan = fftplan_new(b->fft_size);
    b->taps_fft = (orc_c32 *)calloc(b->fft_size, sizeof(orc_c32));
    memcpy(b->taps_fft, taps, sizeof(orc_c32) * ntaps);
    fftplan_run(b->plan, b->taps_fft, 0);
    float f = 1.0f / (float)b->fft_size; /* fft_filter.rs:158-161 */
    for (size_t i = 0; i < b->fft_size; i++) { b->taps_fft[i].re *= f; b->taps_fft[i].im *= f; }
    b->buf = (orc_c32 *)calloc(b->fft_size, sizeof(orc_c32));
    b->tail = (orc_c32 *)calloc(ntaps, sizeof(orc_c32));
    return b;
}

void orc_fftfilter_dims(const orc_block *b, size_t *fft_size, size_t *nsamples) {
    const orc_block *f = b->kind == K_FFTFILT_F ? b->inner : b;
    *fft_size = f->fft_size; *nsamples = f->nsamples;
}

orc_block *orc_fftfilter_float_new(const float *taps, size_t ntaps) { /* fft_filter.rs:397-425 */
    if (ntaps == 0) { set_err("FftFilterFloat: empty taps"); return NULL; }
    orc_c32 *ct = (orc_c32 *)malloc(sizeof(orc_c32) * ntaps);
    for (size_t i = 0; i < ntaps; i++) { ct[i].re = taps[i]; ct[i].im = 0.0f; }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_FFTFILT_F; b->in_es = b->out_es = sizeof(float);
    b->inner = orc_fftfilter_new(ct, ntaps);
    free(ct);
    b->inner_cap = ORC_DEFAULT_STREAM_SIZE / sizeof(orc_c32); /* stream.rs:336-339 */
    b->inner_in = (orc_c32 *)malloc(sizeof(orc_c32) * b->inner_cap);
    b->inner_out = (orc_c32 *)malloc(sizeof(orc_c32) * b->inner_cap);
    return b;
}

static int64_t gcd64(int64_t a, int64_t b) { /* rational_resampler.rs:10-17 */
    while (b != 0) { int64_t t = b; b = a % b; a = t; }
    return a;
}

orc_block *orc_resampler_new(size_t interp, size_t deci, size_t elem_size) { /* :125-151 */
    if (deci == 0) { set_err("RationalResampler created using deci 0"); return NULL; }
    if (interp == 0) { set_err("RationalResampler created using interp 0"); return NULL; }
    if (elem_size == 0 || elem_size > 16) { set_err("resampler elem_size must be 1..16"); return NULL; }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_RESAMP; b->in_es = b->out_es = elem_size;
    int64_t g = gcd64((int64_t)deci, (int64_t)interp);
    b->r_deci = (int64_t)deci / g;
    b->r_interp = (int64_t)interp / g;
    b->counter = 0; b->has_pending = 0;
    return b;
}

orc_block *orc_quaddemod_new(float gain, int atan2_mode) {
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_QUAD; b->in_es = sizeof(orc_c32); b->out_es = sizeof(float);
    b->gain = gain; b->atan2_mode = atan2_mode;
    return b;
}

orc_block *orc_multiply_const_f32_new(float val) { /* multiply_const.rs:6-23 */
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_MULC_F; b->in_es = b->out_es = sizeof(float); b->mval.re = val;
    return b;
}
orc_block *orc_multiply_const_c32_new(float re, float im) {
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_MULC_C; b->in_es = b->out_es = sizeof(orc_c32); b->mval.re = re; b->mval.im = im;
    return b;
}
orc_block *orc_fastfm_new(void) { /* quadrature_demod.rs:144-165; q1 = q2 = 0 (#[rustradio(default)]) */
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_FASTFM; b->in_es = sizeof(orc_c32); b->out_es = sizeof(float);
    return b;
}

/* FftStream::new (fft_stream.rs:40-60).  The reference plans any size with rustfft (forward, unnormalised,
 * e^{-2 pi i k n / N}); this restatement has a radix-4/2 transform for powers of two and evaluates the defining sum in
 * f64 (rounded once to f32) for every other size. */
orc_block *orc_fftstream_new(size_t size) {
    if (size == 0) { set_err("FFT size must be nonzero (fft_stream.rs:42)"); return NULL; }
    if (size > 4096000 / sizeof(orc_c32)) { set_err("FFT size must be no bigger than stream size (fft_stream.rs:46-50)"); return NULL; }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_FFTSTREAM; b->in_es = b->out_es = sizeof(orc_c32);
    b->fft_size = size;
    b->plan = (size & (size - 1)) ? NULL : fftplan_new(size);
    return b;
}

/* the defining sum X[k] = sum_n x[n] e^{-2 pi i k n / N} in f64, in place */
static void dft_f64_inplace(orc_c32 *x, size_t n) {
    double *wr = (double *)malloc(sizeof(double) * n), *wi = (double *)malloc(sizeof(double) * n);
    orc_c32 *y = (orc_c32 *)malloc(sizeof(orc_c32) * n);
    for (size_t j = 0; j < n; j++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)j / (double)n;
        wr[j] = cos(a); wi[j] = sin(a);
    }
    for (size_t k = 0; k < n; k++) {
        double sr = 0.0, si = 0.0;
        size_t idx = 0;                                 /* k * m mod n */
        for (size_t m = 0; m < n; m++) {
            sr += (double)x[m].re * wr[idx] - (double)x[m].im * wi[idx];
            si += (double)x[m].re * wi[idx] + (double)x[m].im * wr[idx];
            idx += k; if (idx >= n) idx -= n;
        }
        y[k].re = (float)sr; y[k].im = (float)si;
    }
    memcpy(x, y, sizeof(orc_c32) * n);
    free(wr); free(wi); free(y);
}

orc_block *orc_rtlsdr_decode_new(void) { /* rtlsdr_decode.rs:9-16 */
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_RTLSDR; b->in_es = 1; b->out_es = sizeof(orc_c32);
    return b;
}

orc_block *orc_hilbert_new(size_t ntaps, int wtype, float parm) { /* hilbert.rs:38-61 */
    if (!(ntaps > 1 && (ntaps & 1) == 1)) {
        set_err("hilbert filter len must be odd and greater than 1"); /* hilbert.rs:44-47 */
        return NULL;
    }
    orc_block *b = (orc_block *)calloc(1, sizeof *b);
    b->kind = K_HILBERT; b->in_es = sizeof(float); b->out_es = sizeof(orc_c32);
    b->ntaps = ntaps;
    float *w = (float *)malloc(sizeof(float) * ntaps);
    float *t = (float *)malloc(sizeof(float) * ntaps);
    if (orc_make_window(wtype, parm, ntaps, w) != 0) { free(w); free(t); free(b); return NULL; }
    orc_hilbert_taps(w, ntaps, t);
    b->rev_f = (float *)malloc(sizeof(float) * ntaps);
    for (size_t j = 0; j < ntaps; j++) b->rev_f[j] = t[ntaps - 1 - j]; /* Fir::new, fir.rs:160 */
    b->history = (float *)calloc(ntaps, sizeof(float)); /* hilbert.rs:55: ntaps zeros */
    free(w); free(t);
    return b;
}

void orc_block_free(orc_block *b) {
    if (!b) return;
    free(b->rev_c); free(b->taps_c); free(b->rev_f);
    free(b->buf); free(b->tail); free(b->taps_fft);
    fftplan_free(b->plan);
    if (b->inner) orc_block_free(b->inner);
    free(b->inner_in); free(b->inner_out);
    free(b->history);
    free(b);
}

int orc_block_eof(orc_block *b, int src_eof) {
    if (b->kind == K_RESAMP) return !b->has_pending && src_eof; /* rational_resampler.rs:209-213 */
    return src_eof; /* rustradio_macros_code/src/lib.rs:596-623 */
}

/* FirFilter::work, fir.rs:492-550 */
static int work_fir(orc_block *b, const void *in, size_t in_len, void *out, size_t out_cap,
                    size_t *consumed, size_t *produced, size_t *need) {
    size_t absolute_minimum = b->ntaps + b->deci - 1;
    if (in_len < absolute_minimum) { *need = absolute_minimum; return ORC_WAIT_SRC; }
    size_t n = b->deci * ((in_len - b->ntaps + 1) / b->deci);
    if (out_cap < 1) { *need = 1; return ORC_WAIT_DST; }
    if (n > out_cap * b->deci) n = out_cap * b->deci;
    size_t out_n = n / b->deci;
    if (b->kind == K_FIR_C32) {
        const orc_c32 *x = (const orc_c32 *)in;
        orc_c32 *y = (orc_c32 *)out;
        for (size_t i = 0; i < out_n; i++) y[i] = fir_c32_one(b->rev_c, b->ntaps, x + i * b->deci);
        if (b->rot_on) { /* translate_output, fir.rs:464-473 */
            for (size_t i = 0; i < out_n; i++) {
                y[i] = c_mul(y[i], b->rot_phase);
                b->rot_phase = c_mul(b->rot_phase, b->rot_step);
            }
        }
    } else {
        const float *x = (const float *)in;
        float *y = (float *)out;
        for (size_t i = 0; i < out_n; i++) y[i] = fir_f32_one(b->rev_f, b->ntaps, x + i * b->deci);
    }
    *consumed = n; *produced = out_n;
    return ORC_AGAIN;
}

/* FftFilter::work, fft_filter.rs:290-354 */
static int work_fftfilter(orc_block *b, const orc_c32 *in, size_t in_len, orc_c32 *out,
                          size_t out_cap, size_t *consumed, size_t *produced, size_t *need) {
    size_t ipos = 0, opos = 0;
    for (;;) {
        if (b->nsamples > out_cap - opos) { /* :294-303 */
            *consumed = ipos; *produced = opos; *need = b->nsamples;
            return ORC_WAIT_DST;
        }
        size_t avail = in_len - ipos;
        size_t add = b->nsamples - b->buf_len;
        if (avail < add) add = avail; /* :306 */
        memcpy(b->buf + b->buf_len, in + ipos, sizeof(orc_c32) * add); /* :308 */
        b->buf_len += add;
        ipos += add; /* :314 */
        if (b->buf_len < b->nsamples) { /* :315-327 */
            *consumed = ipos; *produced = opos; *need = b->nsamples - b->buf_len;
            return ORC_WAIT_SRC;
        }
        for (size_t i = b->nsamples; i < b->fft_size; i++) { b->buf[i].re = 0.0f; b->buf[i].im = 0.0f; } /* :332 */
        /* RustFftEngine::run, :172-176 ; sum_vec :281-287 */
        fftplan_run(b->plan, b->buf, 0);
        for (size_t i = 0; i < b->fft_size; i++) b->buf[i] = c_mul(b->buf[i], b->taps_fft[i]);
        fftplan_run(b->plan, b->buf, 1);
        for (size_t i = 0; i < b->ntaps; i++) b->buf[i] = c_add(b->buf[i], b->tail[i]); /* :336-338 */
        memcpy(out + opos, b->buf, sizeof(orc_c32) * b->nsamples); /* :342-343 */
        opos += b->nsamples;
        for (size_t i = 0; i < b->ntaps; i++) b->tail[i] = b->buf[b->nsamples + i]; /* :346-348 */
        b->buf_len = 0; /* :351 */
    }
}

/* FftFilterFloat::work, fft_filter.rs:429-490 */
static int work_fftfilter_float(orc_block *b, const float *in, size_t in_len, float *out,
                                size_t out_cap, size_t *consumed, size_t *produced, size_t *need) {
    /* convert input to Complex into the inner_in stream (:431-445) */
    size_t room = b->inner_cap - b->inner_in_len;
    size_t n = in_len < room ? in_len : room;
    for (size_t i = 0; i < n; i++) {
        b->inner_in[b->inner_in_len + i].re = in[i];
        b->inner_in[b->inner_in_len + i].im = 0.0f;
    }
    b->inner_in_len += n;
    *consumed = n;
    /* run the inner complex filter (:450) */
    size_t ic = 0, ip = 0, ineed = 0;
    int ret = work_fftfilter(b->inner, b->inner_in, b->inner_in_len,
                             b->inner_out + b->inner_out_len, b->inner_cap - b->inner_out_len,
                             &ic, &ip, &ineed);
    memmove(b->inner_in, b->inner_in + ic, sizeof(orc_c32) * (b->inner_in_len - ic));
    b->inner_in_len -= ic;
    b->inner_out_len += ip;
    /* replicate stream write (:453-470) */
    size_t m = b->inner_out_len < out_cap ? b->inner_out_len : out_cap;
    if (m == 0 && b->inner_out_len != 0) { *produced = 0; *need = 1; return ORC_WAIT_DST; } /* :457-459 */
    for (size_t i = 0; i < m; i++) out[i] = b->inner_out[i].re;
    memmove(b->inner_out, b->inner_out + m, sizeof(orc_c32) * (b->inner_out_len - m));
    b->inner_out_len -= m;
    *produced = m;
    *need = ineed; /* inner WaitForStream mapped onto the outer streams (:474-489) */
    return ret;
}

/* RationalResampler::work, rational_resampler.rs:155-206 */
static int work_resampler(orc_block *b, const unsigned char *in, size_t in_len, unsigned char *out,
                          size_t out_cap, size_t *consumed, size_t *produced, size_t *need) {
    const size_t es = b->in_es;
    *need = 1;
    if (out_cap == 0) return ORC_WAIT_DST; /* :158-160 */
    size_t opos = 0;
    if (b->has_pending) { /* :162-173 */
        while (b->counter > 0) {
            memcpy(out + opos * es, b->pending, es);
            b->counter -= b->r_deci;
            opos++;
            if (opos == out_cap) { *produced = opos; return ORC_WAIT_DST; }
        }
        b->has_pending = 0;
    }
    if (in_len == 0) { *produced = opos; return ORC_WAIT_SRC; } /* :176-179 */
    size_t taken = 0;
    int out_full = 0;
    for (size_t k = 0; k < in_len && !out_full; k++) { /* :183-198 */
        const unsigned char *s = in + k * es;
        taken++;
        b->counter += b->r_interp;
        while (b->counter > 0) {
            memcpy(out + opos * es, s, es);
            b->counter -= b->r_deci;
            opos++;
            if (opos == out_cap) {
                out_full = 1;
                if (b->counter > 0) { memcpy(b->pending, s, es); b->has_pending = 1; }
                break;
            }
        }
    }
    *consumed = taken; *produced = opos;
    return out_full ? ORC_WAIT_DST : ORC_WAIT_SRC;
}

/* QuadratureDemod::work, quadrature_demod.rs:46-113 */
static int work_quad(orc_block *b, const orc_c32 *in, size_t in_len, float *out, size_t out_cap,
                     size_t *consumed, size_t *produced, size_t *need) {
    size_t ipos = 0, opos = 0;
    for (;;) {
        size_t ilen = in_len - ipos, olen = out_cap - opos;
        if (ilen < 2) { *consumed = ipos; *produced = opos; *need = 2; return ORC_WAIT_SRC; }
        if (olen == 0) { *consumed = ipos; *produced = opos; *need = 1; return ORC_WAIT_DST; }
        size_t n1 = ilen - 1 < olen ? ilen - 1 : olen;
        const orc_c32 *i = in + ipos;
        float *o = out + opos;
        for (size_t t = 0; t < n1; t++) {
            orc_c32 cj = { i[t].re, -i[t].im };          /* :72 i[t].conj() * i[t+1] */
            orc_c32 z = c_mul(cj, i[t + 1]);
            if (b->atan2_mode == ORC_ATAN2_FAST) o[t] = b->gain * orc_fast_atan2(z.im, z.re); /* :80 */
            else o[t] = b->gain * atan2f(z.im, z.re);                                         /* :107 */
        }
        ipos += n1; opos += n1; /* :110-111 */
    }
}

/* RtlSdrDecode::work, rtlsdr_decode.rs:18-47: a loop, so one call converts what fits and then
 * reports the side that ran dry. */
static int work_rtlsdr(const unsigned char *in, size_t in_len, orc_c32 *out, size_t out_cap,
                       size_t *consumed, size_t *produced, size_t *need) {
    size_t ipos = 0, opos = 0;
    for (;;) {
        size_t isamples = (in_len - ipos) & ~(size_t)1;                         /* :23 */
        if (isamples == 0) { *consumed = ipos; *produced = opos; *need = 2; return ORC_WAIT_SRC; } /* :24-26 */
        size_t olen = out_cap - opos;
        if (olen == 0) { *consumed = ipos; *produced = opos; *need = 1; return ORC_WAIT_DST; }     /* :28-30 */
        if (isamples > olen * 2) isamples = olen * 2;                          /* :31 */
        size_t osamples = isamples / 2;
        for (size_t i = 0; i < osamples; i++) {                                /* :35-42 */
            float a = (float)in[ipos + 2 * i], b2 = (float)in[ipos + 2 * i + 1];
            out[opos + i].re = (a - 127.0f) * 0.008f;
            out[opos + i].im = (b2 - 127.0f) * 0.008f;
        }
        ipos += isamples; opos += osamples;                                    /* :43-44 */
    }
}

/* work() of a #[rustradio(sync)] block (rustradio_macros_code/src/lib.rs:458-515): a loop that maps
 * min(input, output space) samples through process_sync and stops on the side that ran dry. */
static int work_sync(orc_block *b, const void *in, size_t in_len, void *out, size_t out_cap,
                     size_t *consumed, size_t *produced, size_t *need) {
    size_t ipos = 0, opos = 0;
    *need = 1;
    for (;;) {
        if (in_len - ipos == 0) { *consumed = ipos; *produced = opos; return ORC_WAIT_SRC; }
        if (out_cap - opos == 0) { *consumed = ipos; *produced = opos; return ORC_WAIT_DST; }
        size_t n = in_len - ipos < out_cap - opos ? in_len - ipos : out_cap - opos;
        for (size_t i = 0; i < n; i++) {
            if (b->kind == K_MULC_F) {                                  /* multiply_const.rs:20-22: x * self.val */
                ((float *)out)[opos + i] = ((const float *)in)[ipos + i] * b->mval.re;
            } else if (b->kind == K_MULC_C) {
                ((orc_c32 *)out)[opos + i] = c_mul(((const orc_c32 *)in)[ipos + i], b->mval);
            } else {                                                    /* FastFM::process_sync, quadrature_demod.rs:158-164 */
                const orc_c32 s = ((const orc_c32 *)in)[ipos + i];
                const float top = (s.im - b->q2.im) * b->q1.re;
                const float bottom = (s.re - b->q2.re) * b->q1.im;
                b->q2 = b->q1; b->q1 = s;
                ((float *)out)[opos + i] = top - bottom;
            }
        }
        ipos += n; opos += n;
    }
}

/* FftStream::work, fft_stream.rs:71-117 (frame tags are the wrapper's business) */
static int work_fftstream(orc_block *b, const orc_c32 *in, size_t in_len, orc_c32 *out, size_t out_cap,
                          size_t *consumed, size_t *produced, size_t *need) {
    const size_t size = b->fft_size;
    if (in_len < size) { *need = size; return ORC_WAIT_SRC; }      /* :74-76 */
    if (out_cap < size) { *need = size; return ORC_WAIT_DST; }     /* :79-81 */
    size_t len = in_len < out_cap ? in_len : out_cap;              /* :82-83 */
    len -= len % size;
    memcpy(out, in, len * sizeof(orc_c32));                        /* :84 */
    for (size_t f = 0; f < len; f += size) {                       /* :93-96 forward, unnormalised */
        if (b->plan) fftplan_run(b->plan, out + f, 0);
        else dft_f64_inplace(out + f, size);
    }
    *consumed = len; *produced = len;
    return ORC_AGAIN;
}

/* Hilbert::work, hilbert.rs:72-128 */
static int work_hilbert(orc_block *b, const float *in, size_t in_len, orc_c32 *out, size_t out_cap,
                        size_t *consumed, size_t *produced, size_t *need) {
    *need = 1;
    if (in_len == 0) return ORC_WAIT_SRC; /* :76-78 */
    if (out_cap == 0) return ORC_WAIT_DST; /* :81-83 */
    size_t inout = in_len < out_cap ? in_len : out_cap; /* :85 */
    size_t len = b->ntaps + inout;
    size_t n = len - b->ntaps;
    float *iv = (float *)malloc(sizeof(float) * len); /* :102-104 */
    memcpy(iv, b->history, sizeof(float) * b->ntaps);
    memcpy(iv + b->ntaps, in, sizeof(float) * inout);
    for (size_t i = 0; i < n; i++) { /* :113-116 ; filter_float -> scalar fold fir.rs:146 */
        out[i].re = iv[i + b->ntaps / 2];
        out[i].im = fir_f32_one(b->rev_f, b->ntaps, iv + i);
    }
    memcpy(b->history, iv + n, sizeof(float) * b->ntaps); /* :125 */
    free(iv);
    *consumed = n; *produced = n; *need = 0;
    return ORC_AGAIN;
}

int orc_block_work(orc_block *b, const void *in, size_t in_len, void *out, size_t out_cap,
                   size_t *consumed, size_t *produced, size_t *need) {
    *consumed = 0; *produced = 0; *need = 0;
    switch (b->kind) {
    case K_FIR_C32:
    case K_FIR_F32: return work_fir(b, in, in_len, out, out_cap, consumed, produced, need);
    case K_FFTFILT: return work_fftfilter(b, (const orc_c32 *)in, in_len, (orc_c32 *)out, out_cap, consumed, produced, need);
    case K_FFTFILT_F: return work_fftfilter_float(b, (const float *)in, in_len, (float *)out, out_cap, consumed, produced, need);
    case K_RESAMP: return work_resampler(b, (const unsigned char *)in, in_len, (unsigned char *)out, out_cap, consumed, produced, need);
    case K_QUAD: return work_quad(b, (const orc_c32 *)in, in_len, (float *)out, out_cap, consumed, produced, need);
    case K_HILBERT: return work_hilbert(b, (const float *)in, in_len, (orc_c32 *)out, out_cap, consumed, produced, need);
    case K_MULC_F:
    case K_MULC_C:
    case K_FASTFM: return work_sync(b, in, in_len, out, out_cap, consumed, produced, need);
    case K_FFTSTREAM: return work_fftstream(b, (const orc_c32 *)in, in_len, (orc_c32 *)out, out_cap, consumed, produced, need);
    case K_RTLSDR: return work_rtlsdr((const unsigned char *)in, in_len, (orc_c32 *)out, out_cap, consumed, produced, need);
    }
    set_err("bad block kind");
    return ORC_ERR;
}
