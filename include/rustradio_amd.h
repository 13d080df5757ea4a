/*
 * rustradio_amd.h — C ABI of the MI355X-native rustradio hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point cites the interface of ThomasHabets/rustradio (v0.18.2) it
 * replaces; INTEGRATION.md shows the Rust `impl Block` shim that binds them.
 *
 * One opaque `rr_block` per block instance (the reference's struct fields:
 * taps, carry state).  A handle is not thread-safe but may move between
 * threads (`Block: Send`, src/block.rs:115); it owns a HIP stream and its
 * device buffers, and there is no global mutable state, so N handles can run
 * on N host threads or N GPUs.
 *
 * The library is GPU-only: every entry point that computes fails (NULL /
 * RR_ERR, message in rr_last_error()) when no HIP device is usable.  There is
 * no CPU fallback.
 */
#ifndef RUSTRADIO_AMD_H
#define RUSTRADIO_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 3   /* 3 (round 5): rr_block_tag_rule; zero-copy host windows only for page-aligned ranges (rr_host_register), RR_ZERO_COPY=0.  2 (round 4): rr_dstream_close / _closed / _wait / _id, RR_ROT_REPLAY_DEVICE; rr_build_opts.host_sync_copies removed */

/* Complex<f32>: interleaved [re, im], 8 bytes (src/lib.rs:268-271). */
typedef struct { float re, im; } rr_c32;

typedef struct rr_block rr_block;

/* BlockRet (src/block.rs:12-70).  RR_WAIT_SRC / RR_WAIT_DST are
 * WaitForStream(&self.src, need) / WaitForStream(&self.dst, need). */
enum rr_status {
    RR_AGAIN = 0,
    RR_WAIT_SRC = 1,
    RR_WAIT_DST = 2,
    RR_EOF = 3,
    RR_PENDING = 4,
    RR_ERR = -1
};

/* WindowType (src/window.rs:42-60). */
enum rr_window { RR_WIN_HAMMING = 0, RR_WIN_BLACKMAN = 1, RR_WIN_BLACKMAN_HARRIS = 2, RR_WIN_HAMMING_PARM = 3 };

/* QuadratureDemod atan2 flavour: RR_ATAN2_EXACT = `f32::atan2`
 * (--no-default-features build, src/quadrature_demod.rs:96-109);
 * RR_ATAN2_FAST = `fast_math::atan2` (default Cargo feature, :77-81). */
enum rr_atan2 { RR_ATAN2_EXACT = 0, RR_ATAN2_FAST = 1,
                /* fused-chain constructors only (rr_fm_chain*_create, rr_fm_multi*_create): the chain's demodulator is the
                 * FastFM block (src/quadrature_demod.rs:144-165: one output per resampled sample, two samples of history,
                 * `gain` unused) instead of QuadratureDemod */
                RR_DEMOD_FASTFM = 2 };

/* Rotator evaluation for FirFilter::translate (src/fir.rs:464-473: `sample *= phase; phase *= step` in f32, never
 * renormalised):
 * RR_ROT_REPLAY = DEFAULT.  The reference's sequential f32 recurrence replayed bit for bit, for any stream length.  The
 *                 chain is data-independent, so it is generated AHEAD of the filter on a side stream while the filter
 *                 kernels of the current window run; a call waits only for what the chain has not reached.  It starts
 *                 on the device: one lane walks it from the phase carried in device memory (three packed f32
 *                 instructions per step, 10 ns per output = 100 M outputs/s; no host thread, no PCIe traffic) — enough
 *                 for any graph paced by its source (BASELINE configs[4]: 12.5 M outputs/s).  A block whose calls keep
 *                 arriving before the look-ahead has finished (three in a row: back-to-back batch calls) is given a host
 *                 generator instead: one thread walks the same chain in strict f32 at an x86 core's multiply + add
 *                 latency (2.6 ns per output) from the device chain's current phase into a pinned ring, copied ahead
 *                 into the device ring (8 B per output over PCIe).  The sequential chain bounds such a block either
 *                 way ("sequential_rotator" in bench.py).
 * RR_ROT_REPLAY_DEVICE = the device chain only (never a thread).  RR_ROT_REPLAY_HOST = the host generator from the first
 *                 call.  Bit-identical to each other and to the reference in every case.
 * RR_ROT_MODEL  = opt-in: closed form phase0 * step^m evaluated in f64 from the SAME f32-rounded phase0 / step
 *                 (parallel, as fast as the filter).  NOT parity-faithful for long streams: it differs from the
 *                 recurrence by its accumulated rounding, <= 1e-7 * n after n outputs (measured 1.5e-8 * n:
 *                 tests/test_gpu_edges_fullsize.py::test_rotator_drift_vs_length), i.e. inside the 1e-5 bar only for
 *                 the first ~1e2 (bound) .. 1e5 (measured) outputs of a stream. */
enum rr_rotator { RR_ROT_MODEL = 0, RR_ROT_REPLAY = 1, RR_ROT_REPLAY_DEVICE = 2, RR_ROT_REPLAY_HOST = 3 };

/* ---- library / device ------------------------------------------------------ */
int         rr_abi_version(void);
const char *rr_last_error(void);            /* thread-local message of the last failure */
int         rr_device_count(void);          /* number of visible HIP devices (0 if none) */
int         rr_set_device(int ordinal);     /* device used by blocks created afterwards on this thread */

/* Path-selection overrides for ONE block: rr_next_create_options(&o) applies `o` to the next rr_*_create /
 * rr_dstream_create call made on the calling thread and is cleared when that call returns (thread-local, so
 * handles stay independent and nothing in the process environment changes what a block does).  A zeroed struct =
 * every choice automatic.  Every path stays within the 1e-5 parity bar; the overrides exist so that the parity
 * tests and A/B probes can reach each kernel. */
enum rr_path { RR_PATH_AUTO = 0, RR_PATH_DIRECT = 1, RR_PATH_FFT = 2 };
typedef struct {
    int fir_path;           /* FirFilter / Hilbert>FirFilter: RR_PATH_DIRECT = direct-form kernels only,
                               RR_PATH_FFT = overlap-save tiles for every shape they cover */
    int fir_prune;          /* decimations 4 / 8 / 16: pruned inverse transform, > 0 on, < 0 off */
    int fir_half;           /* other even decimations: half-size inverse, < 0 off */
    int fir_cfg_plus1;      /* direct-form FIR tile shape 0..7, stored + 1 (0 = automatic) */
    int fft_log2f;          /* FftFilter: overlap-save tile of 2^n points, n = 10..14 */
    int fft_no_split;       /* tiles of 8192 / 16384 points as ONE workgroup instead of 2 / 4 sub-transforms */
    int fftfloat_complex;   /* FftFilterFloat: the reference's f32 -> Complex -> FftFilter -> .re inner path */
    int fm_full;            /* fused FM chains: full-size inverse transforms for 1:even ratios too */
    int fm_poly;            /* fused FM chains: decimate-first (polyphase) tiles, > 0 wherever supported, < 0 never;
                               8 / 12: also fixes the multi-channel kernel's waves per workgroup (0 / 1: by predicted cost) */
    int dstream_no_vmm;     /* rr_dstream_create: the copying fallback ring instead of the double mapping */
    int fir_poly;           /* decimating FirFilter<Complex>: decimate-first (polyphase) tiles, > 0 wherever supported, < 0 never */
    int fft_nonfinite_tiles;/* FftFilter / FftFilterFloat: 1 = leave a NaN / Inf input sample's damage on the GPU's tile instead of
                               moving it onto the reference's block of nsamples inputs (no pass behind the tile kernels);
                               3 = do the pass inside the tile kernel's tail (one launch per work(), slower: csrc/blocks.cpp
                               ref_blocks_on); 0 / 2 = the default, a small launch of its own behind every call */
    int host_in_staged;     /* rr_block_work on a page-locked INPUT window: > 0 copy it to device memory first (a copy kernel, 55 GB/s)
                               instead of letting the block's kernels read it in place, < 0 always in place; 0 = the block's default */
    int reserved[3];
} rr_build_opts;
int rr_next_create_options(const rr_build_opts *opts);   /* NULL clears a pending override */

/* ---- tap designers (setup time; host f32, same operation order as the reference) */
float  rr_max_attenuation(int window);                                   /* src/window.rs:67-75 */
int    rr_make_window(int window, float parm, size_t ntaps, float *out); /* src/window.rs:79-185 */
size_t rr_compute_ntaps(float samp_rate, float twidth, int window);      /* src/fir.rs:606-610 */
/* low_pass (src/fir.rs:617-656): returns ntaps, writes min(ntaps, cap) taps; 0 on bad args. */
size_t rr_low_pass(float samp_rate, float cutoff, float twidth, int window, float parm,
                   float *out, size_t cap);
/* low_pass_complex (src/fir.rs:594-604). */
size_t rr_low_pass_complex(float samp_rate, float cutoff, float twidth, int window, float parm,
                           rr_c32 *out, size_t cap);
/* multiband(bands, taps, window) (src/fir.rs:552-590): bands = nbands (low, high) pairs in units of Nyquist,
 * window = ntaps window values; RR_ERR for the reference's None. */
int    rr_multiband(const float *bands, size_t nbands, const float *window, size_t ntaps, rr_c32 *out);
/* hilbert(window) (src/fir.rs:660-680). */
int    rr_hilbert_taps(const float *window, size_t ntaps, float *out);

/* ---- block constructors ------------------------------------------------------ */
/* FirFilter::<Complex>::builder(taps).deci(deci)[.translate(samp_rate, freq)].build(src)
 * (src/fir.rs:303-386, 476-486).  translate != 0 requests frequency translation
 * (freq == 0 disables it, fir.rs:438-440).  NULL on invalid args (the reference asserts).
 * Arithmetic: Fir::filter_n (fir.rs:166-197) in direct form or on overlap-save FFT tiles (plain, decimating store,
 * half-size / pruned inverse, decimate-first), whichever is cheaper for (ntaps, deci) AND the window of the call — the
 * block carries no arithmetic state between calls, so a 512,000-sample ring and a 1e8-sample batch may take different
 * kernels; all within 1e-5 of the reference (rr_fir_fft_tile tells the large-window choice; rr_build_opts forces one). */
rr_block *rr_fir_c32_create(const rr_c32 *taps, size_t ntaps, size_t deci,
                            int translate, float samp_rate, float freq);
/* FirFilter::<Float> (same generic block, src/fir.rs:343-386; long filters on real-stream overlap-save tiles). */
rr_block *rr_fir_f32_create(const float *taps, size_t ntaps, size_t deci);
/* FftFilter::new(src, taps) (src/fft_filter.rs:242-279).  Up to 16383 taps run on LDS-resident overlap-save tiles; longer
 * filters (up to 524,288 taps; the reference has no limit) as overlap-save frames of 2^m >= 2 ntaps points through the
 * any-size transform (plain HBM-streaming passes).  The tile size is internal (rr_fftfilter_dims reports it) and, for long
 * filters, depends on the window of the call; outputs do not depend on it beyond f32 rounding.  A NaN / Inf input sample
 * makes the reference's outputs non-finite — the fft_size outputs from the start of its block of nsamples inputs
 * (src/fft_filter.rs:326-347) — whatever tile ran (a small pass behind the tile kernels of every call; FftFilterFloat too). */
rr_block *rr_fftfilter_create(const rr_c32 *taps, size_t ntaps);
/* FftFilterFloat::new(src, taps) (src/fft_filter.rs:391-426). */
rr_block *rr_fftfilter_float_create(const float *taps, size_t ntaps);
/* RationalResampler::<T>::new(src, interp, deci) for any Copy T of elem_size
 * bytes, elem_size in {1,2,4,8,16} (src/rational_resampler.rs:125-151).
 * NULL (reference: Err) when interp or deci is 0. */
rr_block *rr_resampler_create(size_t interp, size_t deci, size_t elem_size);
/* QuadratureDemod::new(src, gain) (src/quadrature_demod.rs:32-43). */
rr_block *rr_quaddemod_create(float gain, int atan2_mode);
/* RtlSdrDecode::new(src) (src/rtlsdr_decode.rs:9-47): u8 I/Q byte pairs -> Complex,
 * (b - 127) * 0.008; input windows are counted in BYTES, an odd trailing byte is left unconsumed. */
rr_block *rr_rtlsdr_decode_create(void);
/* FftStream::new(src, size) (src/fft_stream.rs:40-117): the forward, unnormalised FFT of every consecutive
 * `size`-sample frame, natural bin order; work(): WAIT_SRC(size) / WAIT_DST(size), else whole frames of
 * min(in, out), RR_AGAIN.  The frame tags (TAG_FRAME, TAG_FRAME_SIZE) are added by the shim from
 * `produced`.  ANY size from 2 to the stream capacity (512,000; rustfft plans any size): one tile kernel up to 16384 = 2^14
 * points and for every size up to 2048 (Bluestein chirp-z on a filter tile), the four-step decomposition for larger powers
 * of two, Bluestein on a power of two M >= 2 size - 1 for everything else. */
rr_block *rr_fftstream_create(size_t size);
/* Fft::from_fft_size(prev, size) (src/fft.rs:19-56): the message (PDU) form — one Vec<Complex> of exactly `size` samples
 * in, its forward FFT out.  The handle is an rr_fftstream_create(size) block; rr_fft_process transforms ONE message held in
 * host memory (n != size: RR_ERR, "FFT expected {size} samples, got {n}" as fft.rs:46-52).  The shim pops the message from
 * its NCReadStream, calls this, pushes the result with the message's tags. */
int rr_fft_process(rr_block *fftstream, const rr_c32 *msg, size_t n, rr_c32 *out);
/* MultiplyConst::<Float|Complex>::new(src, val) (src/multiply_const.rs:6-23) and FastFM::new(src)
 * (src/quadrature_demod.rs:144-165): #[rustradio(sync)] blocks — work() maps min(input, output space)
 * samples and returns WaitForStream(src|dst, 1) (rustradio_macros_code/src/lib.rs:458-515).  Bit-exact. */
rr_block *rr_multiply_const_f32_create(float val);
rr_block *rr_multiply_const_c32_create(float re, float im);
rr_block *rr_fastfm_create(void);
/* Hilbert::new(src, ntaps, &window_type) (src/hilbert.rs:38-61); ntaps odd > 1. */
rr_block *rr_hilbert_create(size_t ntaps, int window, float window_parm);

/* Graph-level fusion of three reference blocks wired as in examples/rtl_fm.rs:381-419:
 *   FftFilter::new(src, taps) -> RationalResampler::new(_, interp, deci) -> QuadratureDemod::new(_, gain)
 * as ONE block (Complex in, f32 out) whose whole-stream output equals that of the three blocks in
 * sequence (src/fft_filter.rs:290-354, src/rational_resampler.rs:155-206,
 * src/quadrature_demod.rs:46-113); the filtered and resampled streams never reach HBM.
 * work(): WAIT_DST(n) when the next filter block's outputs do not fit, else consumes like
 * FftFilter (whole pending block) and WAIT_SRC(nsamples - pending).
 * The output stream must be able to hold one filter block's worth of demodulated samples,
 * ceil(nsamples * interp / deci) (the three separate blocks need only `nsamples` Complex slots in THEIR rings):
 * with the reference's 4,096,000-byte streams that is any ratio up to interp/deci ~ 60 at 16384-point blocks; a
 * smaller output window returns WAIT_DST(n) forever — use the three blocks there.
 * No tap-count limit (the reference has none, src/fft_filter.rs:36-42): up to 16383 taps the chain is one kernel per call;
 * beyond that — or for a decimation larger than a tile — the SAME constructor returns the unfused composition of the three
 * GPU blocks behind the one handle (device-resident intermediates, one work() call, same whole-stream output).
 * atan2_mode = RR_DEMOD_FASTFM puts FastFM (src/quadrature_demod.rs:144-165) in the demodulator's place: one output per
 * resampled sample, bit-exact against FftFilter -> RationalResampler -> FastFM (composition; `gain` unused). */
rr_block *rr_fm_chain_create(const rr_c32 *taps, size_t ntaps, size_t interp, size_t deci,
                             float gain, int atan2_mode);

/* Graph-level fusion of FirFilter::<Complex>::builder(fir_taps).build(src) (deci 1, src/fir.rs:303-386) ->
 * FftFilter::new(_, fft_taps) (src/fft_filter.rs:242-279) — the north star's "127-tap FIR + 1024-pt FftFilter chain" —
 * as ONE convolution with the composite taps fir_taps (*) fft_taps (formed in f64, rounded once).  Whole-stream output
 * equals the two blocks in sequence to f32 rounding, INCLUDING the start of the stream, where FftFilter sees zero
 * history rather than a warm FIR: those fft_ntaps - 1 outputs are recomputed from the two-stage definition.
 * work(): WAIT_DST(nsamples) when a block of the FftFilter stage does not fit (fft_filter.rs:294-303), WAIT_SRC(fir_ntaps)
 * below the FIR's minimum (fir.rs:498-501); otherwise consumes len - (fir_ntaps - 1) samples like the FIR (its history
 * stays in the caller's ring, fir.rs:537), emits whole blocks of nsamples = fft_size(fft_ntaps) - fft_ntaps and returns
 * WAIT_SRC(nsamples - pending + fir_ntaps - 1) / WAIT_DST(nsamples).  Non-finite input samples: the reference's set — the
 * FIR outputs whose fir_ntaps windows hold one, and through them the FftFilter stage's whole blocks (see rr_fftfilter_create). */
rr_block *rr_fir_fftfilter_create(const rr_c32 *fir_taps, size_t fir_ntaps, const rr_c32 *fft_taps, size_t fft_ntaps);
/* ... followed by RationalResampler(interp, deci) -> QuadratureDemod(gain): the metric's whole chain
 * FIR + FftFilter + Resampler + QuadDemod as one kernel (rr_fm_chain_create with the composite filter and the same
 * head fix).  Complex in, f32 out. */
rr_block *rr_fir_fm_chain_create(const rr_c32 *fir_taps, size_t fir_ntaps, const rr_c32 *fft_taps, size_t fft_ntaps,
                                 size_t interp, size_t deci, float gain, int atan2_mode);

/* The same with RtlSdrDecode::new(src) (src/rtlsdr_decode.rs:9-47) fused in front, the receive chain of
 * examples/rtl_fm.rs:328-419 from the RTL-SDR byte stream onwards: u8 in, f32 out, 2 B instead of 8 B
 * read per sample.  Input windows, `consumed` and the WAIT_SRC `need` count BYTES; an odd trailing
 * byte stays unconsumed (rtlsdr_decode.rs:23).  Whole-stream output == RtlSdrDecode -> rr_fm_chain. */
rr_block *rr_fm_chain_u8_create(const rr_c32 *taps, size_t ntaps, size_t interp, size_t deci,
                                float gain, int atan2_mode);

/* Graph-level fusion of the audio stage of examples/rtl_fm.rs:398-418:
 *   FftFilterFloat::new(src, taps) -> RationalResampler::new(_, interp, deci) -> MultiplyConst::new(_, scale)
 * (src/fft_filter.rs:365-491, src/rational_resampler.rs:125-206, src/multiply_const.rs:6-23) as ONE real-valued kernel:
 * f32 in, f32 out; whole-stream output equals the three blocks in sequence.  work() like rr_fm_chain_create: WAIT_DST(n)
 * when the next filter block's resampled samples do not fit, else consumes like FftFilter (whole pending block) and
 * WAIT_SRC(nsamples - pending).  Up to 3584 taps one kernel per call; longer filters run as the unfused composition of
 * the three GPU blocks behind the same handle (no tap-count limit, as in the reference). */
rr_block *rr_audio_chain_create(const float *taps, size_t ntaps, size_t interp, size_t deci, float scale);

/* Graph-level fusion of Hilbert::new(src, hilbert_ntaps, &window) (src/hilbert.rs:38-61) ->
 * FirFilter::<Complex>::builder(taps).deci(deci)[.translate(samp_rate, freq)].build(_) (src/fir.rs:303-386,476-486)
 * as wired in examples/ax25-1200-rx.rs:238-247: f32 in, Complex out, ONE decimating FIR with the composite
 * Complex taps (Hilbert transformer convolved with `taps`, formed in f64).  Whole-stream output equals the
 * two blocks in sequence to f32 rounding; work() follows FirFilter's protocol on the real input
 * (WAIT_SRC(ntaps+deci-1), consumes deci*floor((len-ntaps+1)/deci), RR_AGAIN). */
rr_block *rr_hilbert_fir_create(size_t hilbert_ntaps, int window, float window_parm, const rr_c32 *taps, size_t ntaps,
                                size_t deci, int translate, float samp_rate, float freq);

/* `nchan` fused FM chains (rr_fm_chain_create) fed by ONE input stream — the reference's Tee fan-out
 * (src/tee.rs:10-24) plus nchan x {FftFilter, RationalResampler, QuadratureDemod}.  taps =
 * [nchan][ntaps] (each channel its own, e.g. the prototype shifted to the channel centre); every
 * tile's forward FFT is computed once for all channels.  The block has nchan OUTPUT windows:
 * rr_block_work[_dev] takes `out` as nchan consecutive windows of out_cap elements (channel c at
 * out + c*out_cap) and reports the per-channel consumed/produced (identical for all channels).
 * Up to 4094 taps the channels share every tile's forward transform in one kernel; longer filters (the reference has no
 * limit) run as one rr_fm_chain per channel on the shared window behind the same handle.  atan2_mode as rr_fm_chain_create. */
rr_block *rr_fm_multi_create(const rr_c32 *taps, size_t nchan, size_t ntaps, size_t interp, size_t deci,
                             float gain, int atan2_mode);
/* The same fed by the RTL-SDR byte stream: RtlSdrDecode (src/rtlsdr_decode.rs:9-47) fused in front of the Tee, as in
 * rr_fm_chain_u8_create (input windows, `consumed` and the WAIT_SRC `need` count BYTES). */
rr_block *rr_fm_multi_u8_create(const rr_c32 *taps, size_t nchan, size_t ntaps, size_t interp, size_t deci,
                                float gain, int atan2_mode);
/* number of output windows of a block (1 except rr_fm_multi[_u8]_create) */
size_t rr_block_out_windows(const rr_block *b);

void rr_block_destroy(rr_block *b);

/* ---- Block trait -------------------------------------------------------------- */
/* Block::work() (src/block.rs:115-126) over the stream windows the Rust shim
 * obtained from `self.src.read_buf()` / `self.dst.write_buf()`
 * (src/stream.rs:208-217, 301-310): `in`/`out` are HOST pointers to contiguous
 * windows of in_len / out_cap ELEMENTS.  The block copies the window to the
 * GPU, runs its HIP kernels and copies the produced samples back before
 * returning.  `*consumed` / `*produced` are what the shim must pass to
 * `consume()` / `produce()`; on RR_WAIT_* `*need` is the WaitForStream amount.
 * Thresholds and return codes follow each block's reference `work()` exactly
 * (src/fir.rs:492-550, src/fft_filter.rs:290-354,
 * src/rational_resampler.rs:155-206, src/quadrature_demod.rs:46-113,
 * src/hilbert.rs:72-128). */
int rr_block_work(rr_block *b, const void *in, size_t in_len, void *out, size_t out_cap,
                  size_t *consumed, size_t *produced, size_t *need);

/* Same contract with DEVICE pointers (device-resident streams between GPU
 * blocks: no PCIe hop).  Kernels are enqueued on `hip_stream` exactly as given (a
 * hipStream_t; NULL = HIP's default stream, which is also what PyTorch's default stream
 * is), so they are ordered with the caller's other work on that stream.  The call returns
 * without waiting; counts are final on return (they depend on lengths only). */
int rr_block_work_dev(rr_block *b, const void *d_in, size_t in_len, void *d_out, size_t out_cap,
                      size_t *consumed, size_t *produced, size_t *need, void *hip_stream);

/* BlockEOF::eof() (src/block.rs:103-110): `src_eof` = all input streams are at
 * EOF; the resampler additionally requires no pending sample
 * (src/rational_resampler.rs:209-213). */
int         rr_block_eof(rr_block *b, int src_eof);
/* BlockName::block_name() (src/block.rs:91-97). */
const char *rr_block_name(const rr_block *b);

/* What the reference block(s) behind this handle do with stream tags (src/stream.rs:48-93).  Tags never cross the
 * C ABI: the shim keeps them on the host and re-bases them from `consumed` / `produced`.  This call
 * tells a GENERIC shim block (GpuResident, GpuFused: one type for every handle) which of the reference's rules
 * applies, in whole-stream positions — a tag on input sample a (counted from the start of the stream):
 *   RR_TAGS_DROP     not forwarded: RationalResampler (rational_resampler.rs:156), QuadratureDemod
 *                    (quadrature_demod.rs:46-113), RtlSdrDecode (rtlsdr_decode.rs:21), every chain containing one.
 *   RR_TAGS_FORWARD  re-emitted on output sample a / *param (integer division) once that output exists:
 *                    FirFilter `pos < n` kept at `pos / deci` (fir.rs:536-545; every window starts at a multiple
 *                    of deci, so the window-relative rule is this one), *param = deci; FftFilter / FftFilterFloat
 *                    (fft_filter.rs:307-313,343 / :441-445,467-472), Hilbert (hilbert.rs:119-123), MultiplyConst,
 *                    FastFM (the sync macro, rustradio_macros_code/src/lib.rs:458-515): *param = 1; the fused
 *                    FirFilter -> FftFilter: 1; Hilbert -> FirFilter(deci): deci.
 *   RR_TAGS_FRAMES   input tags dropped, frame tags added per *param = frame size outputs: FftStream
 *                    (fft_stream.rs:98-111).
 * Returns the rule, or RR_ERR for a NULL handle. */
enum rr_tag_rule { RR_TAGS_DROP = 0, RR_TAGS_FORWARD = 1, RR_TAGS_FRAMES = 2 };
int rr_block_tag_rule(const rr_block *b, size_t *param);
size_t      rr_block_in_elem_size(const rr_block *b);
size_t      rr_block_out_elem_size(const rr_block *b);
/* Wait for everything the block enqueued (its private stream and the stream of the last work call). */
int         rr_block_sync(rr_block *b);

/* Page-lock a host range the library will be handed windows of (hipHostRegister): the reference's stream
 * ring is one stable mapping (src/nowasm/circular_buffer.rs:98-128: base, 2 x len), so the shim registers
 * it once at stream creation.  rr_block_work on windows inside registered ranges then runs ZERO-COPY
 * (round 4): the kernels read the input window and write the output window over PCIe themselves, so the
 * window going down overlaps the one coming up on the full-duplex link — two DMA copies do not on this
 * pool (4,096,000-byte windows: FftFilter 193 -> 148 us per call, the fused RTL-SDR chain 175 -> 109);
 * rr_dstream_copy_in/out on such windows run as copy KERNELS on the range's device view (55 GB/s either way; hipMemcpyAsync
 * gets 16-18 GB/s up and 50 down out of a registered range here: profiles/r05_pcie_inplace.txt).  Windows
 * that are not WHOLLY inside a range registered here (pageable memory, memory the caller page-locked by
 * other means) are staged through device memory — and since round 6 through pinned chunks the library owns,
 * copied by the CPU: no GPU engine is ever pointed at memory the caller did not register (csrc/stage.hpp: the
 * runtime's own pinning of a recycled pageable address was the abort() of rounds 4-5).  Expect one host core's
 * memcpy rate there.  Optional; unregister before the memory is unmapped, and not while a work call on one of
 * its windows is running.
 * Zero-copy is granted only to a range whose base and size are multiples of the SYSTEM page size (sysconf(_SC_PAGESIZE)) and none of
 * whose pages is, or ever was, part of another registration in this process — the reference's ring qualifies
 * (one page-aligned mmap, registered once).  Any other range is still page-locked (its staged copies run as
 * direct DMA) but is never handed to kernels in place: kernels working in place on pages that had been
 * page-locked, released and page-locked again were seen to miss on this platform (csrc/blocks.cpp "WHICH ranges
 * run zero-copy").  RR_ZERO_COPY=0 in the environment turns the in-place path off altogether. */
int rr_host_register(void *ptr, size_t bytes);
/* 1 when rr_block_work would let kernels work IN PLACE on the host window [ptr, ptr + bytes), 0 when it would be staged
 * through device memory (tests assert the path they mean to exercise; a shim can log it once per ring). */
int rr_host_window_in_place(const void *ptr, size_t bytes);
int rr_host_unregister(void *ptr);

/* ---- device-resident streams ----------------------------------------------------------
 * A stream ring in HBM with the reference's window contract (src/stream.rs:187-310 over
 * src/nowasm/circular_buffer.rs:98-128): read window = ALL readable elements, write window = ALL free
 * space, both contiguous (double mapping); capacity in bytes (the reference default is 4,096,000, src/stream.rs:105).
 * GPU blocks chained through these never cross PCIe and never wait for each other: all counts are
 * host-side, kernels and the occasional ring move are enqueued on the caller's HIP stream. */
typedef struct rr_dstream rr_dstream;
rr_dstream *rr_dstream_create(size_t elem_size, size_t capacity_bytes);          /* new_stream(), stream.rs:336-339 */
void        rr_dstream_destroy(rr_dstream *s);
size_t      rr_dstream_capacity(const rr_dstream *s);                            /* elements */
/* 1: the ring is one physical allocation mapped twice back to back (HIP virtual-memory API), the
 * reference's own trick (circular_buffer.rs:98-128), no data is ever moved; 0: linear fallback */
int         rr_dstream_is_double_mapped(const rr_dstream *s);
/* ReadStream::read_buf() (stream.rs:208-217): returns the readable element count, *dev_ptr = window */
size_t      rr_dstream_read_buf(rr_dstream *s, const void **dev_ptr);
/* WriteStream::write_buf() (stream.rs:301-310): returns the free element count, *dev_ptr = window (NULL: the count only) */
size_t      rr_dstream_write_buf(rr_dstream *s, void **dev_ptr, void *hip_stream);
int         rr_dstream_consume(rr_dstream *s, size_t n);                         /* BufferReader::consume */
int         rr_dstream_produce(rr_dstream *s, size_t n);                         /* BufferWriter::produce */
/* The two ends of a stream, and how a graph ENDS (round 4).  The reference's ReadStream / WriteStream share one
 * Arc'd buffer; an end is "closed" when the other end has been dropped (strong count 1: src/stream.rs:148-150,166-168),
 * ReadStream::eof() = writer gone AND ring empty (:237-246), and StreamWait::wait(need) returns true when `need` can
 * never be met (:222-224,311-313) — the three facts Graph::run (src/graph.rs:126-147) and MTGraph (src/mtgraph.rs:98-116)
 * finish a block on.  A shim holds one rr_dstream behind a writer handle and a reader handle, calls rr_dstream_close(s, side)
 * when it drops one of them, and destroys the ring with the last.  Every rr_dstream_* call and rr_block_work_streams
 * takes the ring's own lock, so the two ends may live on different threads (one thread per end). */
enum { RR_SIDE_WRITER = 0, RR_SIDE_READER = 1 };
int         rr_dstream_close(rr_dstream *s, int side);                           /* Drop of that end */
int         rr_dstream_closed(rr_dstream *s, int side);                          /* 1 once `side` has been dropped */
/* StreamWait::wait(need) for the `side` end: blocks until `need` elements are readable (READER) / free (WRITER), the
 * other end closes, or `timeout_ms` passes; returns the count it saw last and sets *never (may be NULL) to 1 when that is
 * below `need` and the other end is closed, i.e. the block should go ahead and EOF. */
size_t      rr_dstream_wait(rr_dstream *s, int side, size_t need, unsigned timeout_ms, int *never);
size_t      rr_dstream_id(const rr_dstream *s);                                  /* StreamWait::id(), shared by both ends */
/* Streams: every call that takes a ring and a `hip_stream` may use its own stream — a source pushing window k + 1 on a copy
 * stream while the blocks of window k run on a compute stream.  The ring orders them (an event from the last writer, and
 * from every stream that read since, before a write; from the last writer before a read); with one stream driving a ring it
 * costs nothing.  rr_dstream_read_buf hands out a raw window without a stream: callers that launch their own work on it are
 * ordered like a block if they go through rr_block_work_streams, and on their own otherwise. */
/* BufferWriter::fill_from_slice from HOST memory into the write window at `offset` (not yet produced).  `host` is the
 * caller's again on return (the call waits for the DMA out of a page-locked ring), so the source window can be consumed
 * and overwritten straight away. */
int         rr_dstream_copy_in(rr_dstream *s, size_t offset, const void *host, size_t n, void *hip_stream);
/* copy `n` elements at `offset` of the read window to HOST memory; waits for the stream (data valid on return) */
int         rr_dstream_copy_out(rr_dstream *s, size_t offset, void *host, size_t n, void *hip_stream);
/* `n` elements at `src_offset` of src's READ window -> dst's WRITE window at `dst_offset` (not yet produced), device to
 * device on `hip_stream`, no host involvement: what Tee (src/tee.rs:10-24) and a ring-to-ring copy need between two HBM rings */
int         rr_dstream_copy(rr_dstream *dst, size_t dst_offset, rr_dstream *src, size_t src_offset, size_t n, void *hip_stream);
/* One Block::work() between two device streams: rr_block_work_dev over their windows + consume/produce. */
int rr_block_work_streams(rr_block *b, rr_dstream *src, rr_dstream *dst, size_t *consumed, size_t *produced,
                          size_t *need, void *hip_stream);

/* ---- multi-GPU fan-out of a shared source ------------------------------------------------------------------
 * One process per GPU, chains sharded by channel; the only exchange is the source, which the reference fans out with a Tee
 * tree inside one process (src/tee.rs:10-24).  An rr_fanout is a double buffer of `tile_bytes` in HBM on every rank plus a
 * communication stream: the owning rank's source block writes tile t into one half, an RCCL broadcast (xGMI) delivers it
 * into the same half on every other rank while the blocks of every rank still read tile t - 1 from the other half.  All
 * ordering is by HIP events between the caller's streams and the communication stream; no call blocks the host.
 *
 *     id:  rank src calls rr_fanout_unique_id and ships the 128 bytes to the other processes (file, socket, MPI, env)
 *     f = rr_fanout_create(id, rank, world, src, tile_bytes, 0)                  -- collective: every rank calls it
 *     per tile t = 0, 1, 2 …, on every rank, in order:
 *         rank src:  p = rr_fanout_produce_buf(f, t, s_src);  … enqueue the source's writes of tile t to p on s_src …
 *         rr_fanout_submit(f, t, s_src)                      -- collective: broadcast of tile t on the communication stream
 *         x = rr_fanout_acquire(f, t, s_blk);  rr_block_work_dev(b, x, …, s_blk) …;  rr_fanout_release(f, t, s_blk)
 *     submit(t + 1) may (and for overlap should) be called before acquire(t); tile t + 2 needs release(t) first.
 *     release(t) is called ONCE per tile, on the one stream that has (transitively) waited for every reader of the tile:
 *     blocks reading a tile on several streams join them into one (hipStreamWaitEvent) before the release — a second
 *     release of the same tile, or one of a tile that has left the double buffer, is an error.
 * A one-rank fan-out (world 1) needs neither an id nor RCCL and keeps the same calls, so a graph is written once.
 * RCCL is bound at run time (dlopen), the library itself does not depend on it.  Like a block, a handle is driven by one
 * thread at a time; calls made out of protocol order (a tile skipped, a half not yet released) fail with RR_ERR / NULL and
 * rr_last_error instead of racing. */
typedef struct rr_fanout rr_fanout;
#define RR_FANOUT_ID_BYTES 128
enum rr_fanout_flags {
    RR_FANOUT_TIMING      = 1,  /* time every 4th broadcast with HIP events on the communication stream (rr_fanout_stats) */
    RR_FANOUT_RCCL_ALWAYS = 2,  /* run a one-rank group through RCCL as well (tests of the RCCL path on a one-GPU machine) */
    RR_FANOUT_MESH        = 4   /* fan out by ncclScatter (1/world of the tile to each rank) + in-place ncclAllGather between the
                                   receivers instead of one ncclBroadcast: a broadcast is bounded by ONE of the owner's xGMI
                                   links, this form spreads the tile over the full mesh (same flag on every rank) */
};
int         rr_fanout_unique_id(void *id128);
rr_fanout  *rr_fanout_create(const void *id128, int rank, int world, int src_rank, size_t tile_bytes, int flags);
void        rr_fanout_destroy(rr_fanout *f);
void       *rr_fanout_produce_buf(rr_fanout *f, unsigned long long t, void *producer_stream);   /* NULL + rr_last_error */
int         rr_fanout_submit(rr_fanout *f, unsigned long long t, void *producer_stream);        /* non-owning ranks: stream unused */
const void *rr_fanout_acquire(rr_fanout *f, unsigned long long t, void *compute_stream);        /* NULL + rr_last_error */
int         rr_fanout_release(rr_fanout *f, unsigned long long t, void *compute_stream);
/* waits for the communication stream; summed duration and number of the broadcasts timed since the last call */
int         rr_fanout_stats(rr_fanout *f, double *broadcast_ms, size_t *broadcasts);

/* Measurement aid: when enabled, every work call brackets the block's dominant kernel
 * with HIP events on the stream it is launched on; rr_block_profile waits for them and
 * returns the summed kernel time and the number of launches (reset != 0 clears them). */
int rr_block_set_profiling(rr_block *b, int on);
int rr_block_profile(rr_block *b, double *total_ms, size_t *launches, int reset);

/* Measurement builds only (make TIMING=1): 32 s_memtime stamps taken at the phase boundaries of one tile of the
 * last stamped kernel (out must hold 32 values); returns 0 in product builds. */
int rr_debug_fft_stamps(unsigned long long *out32);
/* Kernel launches the library has made in this process so far (every hipLaunchKernelGGL of a block's work(), incl. carry
 * copies and repair passes; not the copies of host windows).  Tests use the difference around one call: a clean
 * FftFilter work() on a device window is ONE launch (round 6; the reference's one work() = one pass over the window,
 * src/fft_filter.rs:289-355). */
unsigned long long rr_debug_kernel_launches(void);

/* ---- per-block knobs / introspection ------------------------------------------- */
/* FftFilter: reference fft_size and nsamples (src/fft_filter.rs:261-262) and the
 * internal overlap-save tile the GPU kernel uses. */
int rr_fftfilter_dims(const rr_block *b, size_t *fft_size, size_t *nsamples, size_t *gpu_fft_size);
/* FirFilter<Complex>: size of the overlap-save FFT tile a non-decimating filter runs on, 0 when the block uses the
 * direct-form kernel (introspection for tests and benches; the result of Fir::filter, src/fir.rs:166-197, either way). */
size_t rr_fir_fft_tile(const rr_block *b);
/* FirFilter translate (also inside rr_hilbert_fir_create): rotator mode (default RR_ROT_REPLAY, the on-parity one).
 * Switching to a REPLAY mode after outputs were produced walks the chain from the stream start first (once). */
int rr_fir_set_rotator_mode(rr_block *b, int mode);

#ifdef __cplusplus
}
#endif
#endif /* RUSTRADIO_AMD_H */
