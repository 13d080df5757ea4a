//! GPU (MI355X / gfx950) replacements for rustradio's hot-path blocks, behind the
//! unchanged `Block` / `ReadStream` / `WriteStream` API: a graph from `examples/` swaps
//! `FftFilter::new(prev, taps)` for `GpuFftFilter::new(prev, taps)` and nothing else.
//!
//! SOURCE ONLY: the build image has no cargo/rustc, so this file is not compiled in CI.
//! It binds the C ABI of `include/rustradio_amd.h` one to one — `tests/test_rust_shim.py` parses the
//! `extern "C"` block below and checks every declaration (name, arity, argument and return types)
//! against the header — and the C++ mirror `rustradio_amd/host/rustradio.hpp` implements the same shim
//! logic and IS tested (`tests/cpp/test_host_api.cpp`).
use std::ffi::{c_int, c_uint, c_void, CStr};
use std::sync::{Arc, Condvar, Mutex};

use rustradio::block::{Block, BlockEOF, BlockName, BlockRet};
use rustradio::stream::{new_stream, ReadStream, StreamWait, Tag, TagValue, WriteStream};
use rustradio::window::WindowType;
use rustradio::{Complex, Error, Float, Result, Sample};

#[repr(C)]
pub struct RrBlock {
    _private: [u8; 0],
}
#[repr(C)]
pub struct RrDStream {
    _private: [u8; 0],
}
#[repr(C)]
pub struct RrFanout {
    _private: [u8; 0],
}

// enum rr_status (include/rustradio_amd.h)
const RR_AGAIN: c_int = 0;
const RR_WAIT_SRC: c_int = 1;
const RR_WAIT_DST: c_int = 2;
const RR_EOF: c_int = 3;
const RR_ERR: c_int = -1;
// enum rr_tag_rule
const RR_TAGS_FORWARD: c_int = 1;
const RR_TAGS_FRAMES: c_int = 2;
// enum { RR_SIDE_WRITER, RR_SIDE_READER }
const RR_SIDE_WRITER: c_int = 0;
const RR_SIDE_READER: c_int = 1;

unsafe extern "C" {
    fn rr_last_error() -> *const libc::c_char;
    fn rr_fir_c32_create(taps: *const Complex, ntaps: usize, deci: usize, translate: c_int, samp_rate: f32, freq: f32) -> *mut RrBlock;
    fn rr_fir_f32_create(taps: *const f32, ntaps: usize, deci: usize) -> *mut RrBlock;
    fn rr_fftfilter_create(taps: *const Complex, ntaps: usize) -> *mut RrBlock;
    fn rr_fftfilter_float_create(taps: *const f32, ntaps: usize) -> *mut RrBlock;
    fn rr_resampler_create(interp: usize, deci: usize, elem_size: usize) -> *mut RrBlock;
    fn rr_quaddemod_create(gain: f32, atan2_mode: c_int) -> *mut RrBlock;
    fn rr_hilbert_create(ntaps: usize, window: c_int, window_parm: f32) -> *mut RrBlock;
    fn rr_rtlsdr_decode_create() -> *mut RrBlock;
    fn rr_multiply_const_f32_create(val: f32) -> *mut RrBlock;
    fn rr_fastfm_create() -> *mut RrBlock;
    fn rr_fm_chain_create(taps: *const Complex, ntaps: usize, interp: usize, deci: usize, gain: f32, atan2_mode: c_int) -> *mut RrBlock;
    fn rr_fm_chain_u8_create(taps: *const Complex, ntaps: usize, interp: usize, deci: usize, gain: f32, atan2_mode: c_int) -> *mut RrBlock;
    fn rr_audio_chain_create(taps: *const f32, ntaps: usize, interp: usize, deci: usize, scale: f32) -> *mut RrBlock;
    fn rr_fir_fftfilter_create(fir_taps: *const Complex, fir_ntaps: usize, fft_taps: *const Complex, fft_ntaps: usize) -> *mut RrBlock;
    fn rr_fir_fm_chain_create(fir_taps: *const Complex, fir_ntaps: usize, fft_taps: *const Complex, fft_ntaps: usize,
                              interp: usize, deci: usize, gain: f32, atan2_mode: c_int) -> *mut RrBlock;
    fn rr_hilbert_fir_create(hilbert_ntaps: usize, window: c_int, window_parm: f32, taps: *const Complex, ntaps: usize,
                             deci: usize, translate: c_int, samp_rate: f32, freq: f32) -> *mut RrBlock;
    fn rr_fm_multi_create(taps: *const Complex, nchan: usize, ntaps: usize, interp: usize, deci: usize, gain: f32, atan2_mode: c_int) -> *mut RrBlock;
    fn rr_fm_multi_u8_create(taps: *const Complex, nchan: usize, ntaps: usize, interp: usize, deci: usize, gain: f32, atan2_mode: c_int) -> *mut RrBlock;
    fn rr_fanout_unique_id(id128: *mut c_void) -> c_int;
    fn rr_fanout_create(id128: *const c_void, rank: c_int, world: c_int, src_rank: c_int, tile_bytes: usize, flags: c_int) -> *mut RrFanout;
    fn rr_fanout_destroy(f: *mut RrFanout);
    fn rr_fanout_produce_buf(f: *mut RrFanout, t: u64, producer_stream: *mut c_void) -> *mut c_void;
    fn rr_fanout_submit(f: *mut RrFanout, t: u64, producer_stream: *mut c_void) -> c_int;
    fn rr_fanout_acquire(f: *mut RrFanout, t: u64, compute_stream: *mut c_void) -> *const c_void;
    fn rr_fanout_release(f: *mut RrFanout, t: u64, compute_stream: *mut c_void) -> c_int;
    fn rr_fanout_stats(f: *mut RrFanout, broadcast_ms: *mut f64, broadcasts: *mut usize) -> c_int;
    fn rr_fftstream_create(size: usize) -> *mut RrBlock;
    fn rr_fft_process(fftstream: *mut RrBlock, msg: *const Complex, n: usize, out: *mut Complex) -> c_int;
    fn rr_multiply_const_c32_create(re: f32, im: f32) -> *mut RrBlock;
    fn rr_block_out_windows(b: *const RrBlock) -> usize;
    fn rr_block_destroy(b: *mut RrBlock);
    fn rr_block_work(b: *mut RrBlock, inp: *const c_void, in_len: usize, out: *mut c_void, out_cap: usize,
                     consumed: *mut usize, produced: *mut usize, need: *mut usize) -> c_int;
    fn rr_block_eof(b: *mut RrBlock, src_eof: c_int) -> c_int;
    fn rr_block_tag_rule(b: *const RrBlock, param: *mut usize) -> c_int;
    fn rr_block_work_dev(b: *mut RrBlock, d_in: *const c_void, in_len: usize, d_out: *mut c_void, out_cap: usize,
                         consumed: *mut usize, produced: *mut usize, need: *mut usize, hip_stream: *mut c_void) -> c_int;
    fn rr_block_sync(b: *mut RrBlock) -> c_int;
    fn rr_fir_set_rotator_mode(b: *mut RrBlock, mode: c_int) -> c_int;
    fn rr_host_register(ptr: *mut c_void, bytes: usize) -> c_int;
    fn rr_host_window_in_place(ptr: *const c_void, bytes: usize) -> c_int;
    fn rr_host_unregister(ptr: *mut c_void) -> c_int;
    fn rr_dstream_create(elem_size: usize, capacity_bytes: usize) -> *mut RrDStream;
    fn rr_dstream_destroy(s: *mut RrDStream);
    fn rr_dstream_capacity(s: *const RrDStream) -> usize;
    fn rr_dstream_read_buf(s: *mut RrDStream, dev_ptr: *mut *const c_void) -> usize;
    fn rr_dstream_write_buf(s: *mut RrDStream, dev_ptr: *mut *mut c_void, hip_stream: *mut c_void) -> usize;
    fn rr_dstream_consume(s: *mut RrDStream, n: usize) -> c_int;
    fn rr_dstream_produce(s: *mut RrDStream, n: usize) -> c_int;
    fn rr_dstream_close(s: *mut RrDStream, side: c_int) -> c_int;
    fn rr_dstream_closed(s: *mut RrDStream, side: c_int) -> c_int;
    fn rr_dstream_wait(s: *mut RrDStream, side: c_int, need: usize, timeout_ms: c_uint, never: *mut c_int) -> usize;
    fn rr_dstream_id(s: *const RrDStream) -> usize;
    fn rr_dstream_copy_in(s: *mut RrDStream, offset: usize, host: *const c_void, n: usize, hip_stream: *mut c_void) -> c_int;
    fn rr_dstream_copy_out(s: *mut RrDStream, offset: usize, host: *mut c_void, n: usize, hip_stream: *mut c_void) -> c_int;
    fn rr_dstream_copy(dst: *mut RrDStream, dst_offset: usize, src: *mut RrDStream, src_offset: usize, n: usize, hip_stream: *mut c_void) -> c_int;
    fn rr_block_work_streams(b: *mut RrBlock, src: *mut RrDStream, dst: *mut RrDStream, consumed: *mut usize,
                             produced: *mut usize, need: *mut usize, hip_stream: *mut c_void) -> c_int;
}

fn last_error() -> Error {
    // SAFETY: rr_last_error returns a NUL-terminated thread-local string.
    let s = unsafe { CStr::from_ptr(rr_last_error()) }.to_string_lossy().into_owned();
    Error::msg(s)
}

/// Owning handle of one `rr_block`.
struct Handle(*mut RrBlock);
// SAFETY: a handle owns its HIP stream and device buffers and has no thread affinity;
// `Block: Send` only requires moving between threads, never sharing.
unsafe impl Send for Handle {}
impl Drop for Handle {
    fn drop(&mut self) {
        // SAFETY: created by an rr_*_create call, destroyed once.
        unsafe { rr_block_destroy(self.0) }
    }
}
impl Handle {
    fn new(p: *mut RrBlock) -> Result<Self> {
        if p.is_null() { Err(last_error()) } else { Ok(Self(p)) }
    }
    /// One `rr_block_work` call over the two stream windows.
    fn work<I: Sample, O: Sample>(&mut self, input: &[I], out: &mut [O]) -> Result<(c_int, usize, usize, usize)> {
        let (mut c, mut p, mut need) = (0usize, 0usize, 0usize);
        // SAFETY: pointers/lengths describe live, contiguous windows for the duration of the call.
        let st = unsafe {
            rr_block_work(self.0, input.as_ptr().cast(), input.len(), out.as_mut_ptr().cast(), out.len(),
                          &mut c, &mut p, &mut need)
        };
        if st == RR_ERR { Err(last_error()) } else { Ok((st, c, p, need)) }
    }
}

/// Tags through a block that is known only by its handle (`GpuFused`, `GpuResident`): the reference rule of the block(s)
/// behind the handle as `rr_block_tag_rule` states it in whole-stream positions.  `step` takes the tags of the read
/// window (positions relative to it; only those on consumed samples travel, the rest stay in the stream) and returns the
/// tags of the `produced` outputs of this call, relative to the first of them.  A tag whose output sample does not exist
/// yet waits here, as it does in FftFilter's `self.tags` (src/fft_filter.rs:309-313).
/// C++ twin: `detail::TagForwarder` (rustradio_amd/host/rustradio.hpp), tested by tests/cpp/test_resident_graph.cpp.
struct TagForwarder {
    rule: c_int,
    param: usize,
    pending: Vec<(u64, Tag)>,   // (output position counted from the start of the stream, tag)
    in_abs: u64,
    out_abs: u64,
}
impl TagForwarder {
    fn new(h: &Handle) -> Result<Self> {
        let mut param = 1usize;
        // SAFETY: valid handle, live out-parameter.
        let rule = unsafe { rr_block_tag_rule(h.0, &mut param) };
        if rule == RR_ERR { return Err(last_error()); }
        Ok(Self { rule, param: param.max(1), pending: Vec::new(), in_abs: 0, out_abs: 0 })
    }
    fn step(&mut self, tags: Vec<Tag>, consumed: usize, produced: usize) -> Vec<Tag> {
        let mut emit = Vec::new();
        if self.rule == RR_TAGS_FORWARD {
            for t in tags.into_iter().filter(|t| t.pos() < consumed) {
                self.pending.push(((self.in_abs + t.pos() as u64) / self.param as u64, t));
            }
            let limit = self.out_abs + produced as u64;
            let (now, keep): (Vec<_>, Vec<_>) = self.pending.drain(..).partition(|(abs, _)| *abs < limit);
            self.pending = keep;
            emit = now.into_iter()
                .map(|(abs, t)| Tag::new((abs - self.out_abs) as usize, t.key(), t.val().clone()))
                .collect();
        } else if self.rule == RR_TAGS_FRAMES {
            // src/fft_stream.rs:98-111 (produced is a whole number of frames)
            let mut pos = 0usize;
            while pos + self.param <= produced {
                emit.push(Tag::new(pos, TAG_FRAME_SIZE, TagValue::U64(self.param as u64)));
                emit.push(Tag::new(pos, TAG_FRAME, TagValue::Bool(true)));
                emit.push(Tag::new(pos + self.param - 1, TAG_FRAME, TagValue::Bool(false)));
                pos += self.param;
            }
        }
        self.in_abs += consumed as u64;
        self.out_abs += produced as u64;
        emit
    }
}

/// `FftFilter` on the GPU (replaces `rustradio::blocks::FftFilter`, src/fft_filter.rs:210-355).
pub struct GpuFftFilter {
    h: Handle,
    src: ReadStream<Complex>,
    dst: WriteStream<Complex>,
    fwd: TagForwarder,   // tags of samples still inside the filter's block buffer wait here (fft_filter.rs:309-313)
}
impl GpuFftFilter {
    pub fn new<T: Into<Vec<Complex>>>(src: ReadStream<Complex>, taps: T) -> Result<(Self, ReadStream<Complex>)> {
        let taps = taps.into();
        // SAFETY: taps is a live slice of repr(C) Complex<f32>.
        let h = Handle::new(unsafe { rr_fftfilter_create(taps.as_ptr(), taps.len()) })?;
        let fwd = TagForwarder::new(&h)?;
        let (dst, dr) = new_stream();
        Ok((Self { h, src, dst, fwd }, dr))
    }
}
impl BlockName for GpuFftFilter {
    fn block_name(&self) -> &str { "GpuFftFilter" }
}
impl BlockEOF for GpuFftFilter {
    fn eof(&mut self) -> bool {
        // SAFETY: valid handle.
        unsafe { rr_block_eof(self.h.0, self.src.eof() as c_int) != 0 }
    }
}
impl Block for GpuFftFilter {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        // a tag travels with its sample (fft_filter.rs:307-313,343)
        let out_tags = self.fwd.step(tags, consumed, produced);
        input.consume(consumed);
        out.produce(produced, &out_tags);
        Ok(match st {
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            _ => BlockRet::Again,
        })
    }
}

/// `FirFilter<Complex>` on the GPU incl. `.deci()` and `.translate()` (src/fir.rs:303-551).
pub struct GpuFirFilter {
    h: Handle,
    deci: usize,
    src: ReadStream<Complex>,
    dst: WriteStream<Complex>,
}
pub struct GpuFirFilterBuilder {
    taps: Vec<Complex>,
    deci: usize,
    translate: Option<(Float, Float)>,
}
impl GpuFirFilterBuilder {
    #[must_use] pub fn deci(mut self, deci: usize) -> Self { assert_ne!(deci, 0); self.deci = deci; self }
    #[must_use] pub fn translate(mut self, samp_rate: Float, freq: Float) -> Self { self.translate = Some((samp_rate, freq)); self }
    pub fn build(self, src: ReadStream<Complex>) -> Result<(GpuFirFilter, ReadStream<Complex>)> {
        let (fs, f) = self.translate.unwrap_or((0.0, 0.0));
        // SAFETY: taps is a live slice.
        let h = Handle::new(unsafe {
            rr_fir_c32_create(self.taps.as_ptr(), self.taps.len(), self.deci, self.translate.is_some() as c_int, fs, f)
        })?;
        let (dst, dr) = new_stream();
        Ok((GpuFirFilter { h, deci: self.deci, src, dst }, dr))
    }
}
impl GpuFirFilter {
    pub fn builder(taps: impl Into<Vec<Complex>>) -> GpuFirFilterBuilder {
        GpuFirFilterBuilder { taps: taps.into(), deci: 1, translate: None }
    }
}
impl BlockName for GpuFirFilter { fn block_name(&self) -> &str { "GpuFirFilter" } }
impl BlockEOF for GpuFirFilter { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuFirFilter {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, mut tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        match st {
            RR_WAIT_SRC => return Ok(BlockRet::WaitForStream(&self.src, need)),
            RR_WAIT_DST => return Ok(BlockRet::WaitForStream(&self.dst, need)),
            _ => {}
        }
        tags.retain(|t| t.pos() < consumed);                     // fir.rs:536
        for t in &mut tags { t.set_pos(t.pos() / self.deci); }   // fir.rs:541-543
        input.consume(consumed);
        out.produce(produced, &tags);
        debug_assert_eq!(st, RR_AGAIN);
        Ok(BlockRet::Again)
    }
}

/// `RtlSdrDecode -> FftFilter -> RationalResampler -> QuadratureDemod` (examples/rtl_fm.rs:328-419) as ONE
/// GPU block: RTL-SDR bytes in, demodulated f32 out.  All four reference blocks drop tags.
pub struct GpuRtlFmChain {
    h: Handle,
    src: ReadStream<u8>,
    dst: WriteStream<Float>,
}
impl GpuRtlFmChain {
    pub fn new(src: ReadStream<u8>, taps: &[Complex], interp: usize, deci: usize, gain: Float, fast_math: bool)
        -> Result<(Self, ReadStream<Float>)> {
        // SAFETY: taps is a live slice of repr(C) Complex<f32>.
        let h = Handle::new(unsafe { rr_fm_chain_u8_create(taps.as_ptr(), taps.len(), interp, deci, gain, fast_math as c_int) })?;
        let (dst, dr) = new_stream();
        Ok((Self { h, src, dst }, dr))
    }
}
impl BlockName for GpuRtlFmChain { fn block_name(&self) -> &str { "GpuRtlFmChain" } }
impl BlockEOF for GpuRtlFmChain { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuRtlFmChain {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, _tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;   // byte counts on the input side
        input.consume(consumed);
        out.produce(produced, &[]);
        Ok(if st == RR_WAIT_DST { BlockRet::WaitForStream(&self.dst, need) } else { BlockRet::WaitForStream(&self.src, need) })
    }
}

/// Every hot-path block whose tags are dropped (`RationalResampler`, `QuadratureDemod`, `RtlSdrDecode`) or pass
/// through position-for-position (`MultiplyConst`, `FastFM`): one input stream, one output stream, the C ABI
/// does the rest.  `keep_tags` selects the second behaviour (rustradio_macros_code/src/lib.rs:458-515).
pub struct GpuMap<I: Sample, O: Sample> {
    h: Handle,
    name: &'static str,
    keep_tags: bool,
    src: ReadStream<I>,
    dst: WriteStream<O>,
}
impl<I: Sample, O: Sample> GpuMap<I, O> {
    fn wrap(h: *mut RrBlock, name: &'static str, keep_tags: bool, src: ReadStream<I>) -> Result<(Self, ReadStream<O>)> {
        let h = Handle::new(h)?;
        let (dst, dr) = new_stream();
        Ok((Self { h, name, keep_tags, src, dst }, dr))
    }
}
impl<T: Sample> GpuMap<T, T> {
    /// `RationalResampler::new(src, interp, deci)` (src/rational_resampler.rs:125-151)
    pub fn rational_resampler(src: ReadStream<T>, interp: usize, deci: usize) -> Result<(Self, ReadStream<T>)> {
        // SAFETY: plain values.
        Self::wrap(unsafe { rr_resampler_create(interp, deci, std::mem::size_of::<T>()) }, "GpuRationalResampler", false, src)
    }
}
impl GpuMap<Complex, Float> {
    /// `QuadratureDemod::new(src, gain)`; `fast_math` = the Cargo feature the application is built with
    pub fn quadrature_demod(src: ReadStream<Complex>, gain: Float, fast_math: bool) -> Result<(Self, ReadStream<Float>)> {
        Self::wrap(unsafe { rr_quaddemod_create(gain, fast_math as c_int) }, "GpuQuadratureDemod", false, src)
    }
    /// `FastFM::new(src)` (src/quadrature_demod.rs:144-165)
    pub fn fast_fm(src: ReadStream<Complex>) -> Result<(Self, ReadStream<Float>)> {
        Self::wrap(unsafe { rr_fastfm_create() }, "GpuFastFM", true, src)
    }
}
impl GpuMap<u8, Complex> {
    /// `RtlSdrDecode::new(src)` (src/rtlsdr_decode.rs:9-47)
    pub fn rtlsdr_decode(src: ReadStream<u8>) -> Result<(Self, ReadStream<Complex>)> {
        Self::wrap(unsafe { rr_rtlsdr_decode_create() }, "GpuRtlSdrDecode", false, src)
    }
}
impl GpuMap<Float, Float> {
    /// `MultiplyConst::new(src, val)` (src/multiply_const.rs:6-23)
    pub fn multiply_const(src: ReadStream<Float>, val: Float) -> Result<(Self, ReadStream<Float>)> {
        Self::wrap(unsafe { rr_multiply_const_f32_create(val) }, "GpuMultiplyConst", true, src)
    }
}
impl<I: Sample, O: Sample> BlockName for GpuMap<I, O> { fn block_name(&self) -> &str { self.name } }
impl<I: Sample, O: Sample> BlockEOF for GpuMap<I, O> {
    fn eof(&mut self) -> bool {
        // SAFETY: valid handle.  (The resampler also needs its pending sample flushed: rational_resampler.rs:209-213.)
        unsafe { rr_block_eof(self.h.0, self.src.eof() as c_int) != 0 }
    }
}
impl<I: Sample, O: Sample> Block for GpuMap<I, O> {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, mut tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        if self.keep_tags { tags.retain(|t| t.pos() < produced); } else { tags.clear(); }
        input.consume(consumed);
        out.produce(produced, &tags);
        Ok(match st {
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            _ => BlockRet::Again,
        })
    }
}

/// `Hilbert::new(src, ntaps, &window_type)` (src/hilbert.rs:38-129): f32 in, analytic Complex out; tags with
/// `pos < n` are forwarded unchanged (hilbert.rs:119-123).
pub struct GpuHilbert {
    h: Handle,
    src: ReadStream<Float>,
    dst: WriteStream<Complex>,
}
impl GpuHilbert {
    pub fn new(src: ReadStream<Float>, ntaps: usize, window_type: &WindowType) -> Result<(Self, ReadStream<Complex>)> {
        let (w, parm) = window_code(window_type);
        // SAFETY: plain values.
        let h = Handle::new(unsafe { rr_hilbert_create(ntaps, w, parm) })?;
        let (dst, dr) = new_stream();
        Ok((Self { h, src, dst }, dr))
    }
}
impl BlockName for GpuHilbert { fn block_name(&self) -> &str { "GpuHilbert" } }
impl BlockEOF for GpuHilbert { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuHilbert {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, mut tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        tags.retain(|t| t.pos() < produced);
        input.consume(consumed);
        out.produce(produced, &tags);
        Ok(match st {
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            _ => BlockRet::Again,
        })
    }
}

/// `FftFilterFloat::new(src, taps)` (src/fft_filter.rs:365-491).  The reference carries tags into its inner Complex
/// stream, through the inner FftFilter and out again (fft_filter.rs:441-445,467-472): a tag travels with its sample.
pub struct GpuFftFilterFloat {
    h: Handle,
    src: ReadStream<Float>,
    dst: WriteStream<Float>,
    fwd: TagForwarder,
}
impl GpuFftFilterFloat {
    pub fn new(src: ReadStream<Float>, taps: &[Float]) -> Result<(Self, ReadStream<Float>)> {
        // SAFETY: taps is a live slice.
        let h = Handle::new(unsafe { rr_fftfilter_float_create(taps.as_ptr(), taps.len()) })?;
        let fwd = TagForwarder::new(&h)?;
        let (dst, dr) = new_stream();
        Ok((Self { h, src, dst, fwd }, dr))
    }
}
impl BlockName for GpuFftFilterFloat { fn block_name(&self) -> &str { "GpuFftFilterFloat" } }
impl BlockEOF for GpuFftFilterFloat { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuFftFilterFloat {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        let out_tags = self.fwd.step(tags, consumed, produced);
        input.consume(consumed);
        out.produce(produced, &out_tags);
        Ok(match st {
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            _ => BlockRet::Again,
        })
    }
}

/// `FftStream::new(src, size)` (src/fft_stream.rs:40-117): frame tags are rebuilt from `produced` exactly as
/// fft_stream.rs:98-111 (input tags are dropped there too).
pub const TAG_FRAME: &str = "FftStream::frame";
pub const TAG_FRAME_SIZE: &str = "FftStream::size";
pub struct GpuFftStream {
    h: Handle,
    size: usize,
    src: ReadStream<Complex>,
    dst: WriteStream<Complex>,
}
impl GpuFftStream {
    pub fn new(src: ReadStream<Complex>, size: usize) -> Result<(Self, ReadStream<Complex>)> {
        // SAFETY: plain value.
        let h = Handle::new(unsafe { rr_fftstream_create(size) })?;
        let (dst, dr) = new_stream();
        Ok((Self { h, size, src, dst }, dr))
    }
}
impl BlockName for GpuFftStream { fn block_name(&self) -> &str { "GpuFftStream" } }
impl BlockEOF for GpuFftStream { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuFftStream {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, _tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        let mut tags = Vec::with_capacity((produced / self.size) * 3);
        for pos in (0..produced).step_by(self.size) {
            tags.push(Tag::new(pos, TAG_FRAME_SIZE, TagValue::U64(self.size as u64)));
            tags.push(Tag::new(pos, TAG_FRAME, TagValue::Bool(true)));
            tags.push(Tag::new(pos + self.size - 1, TAG_FRAME, TagValue::Bool(false)));
        }
        input.consume(consumed);
        out.produce(produced, &tags);
        Ok(match st {
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            _ => BlockRet::Again,
        })
    }
}

/// `Fft::from_fft_size(prev, size)` (src/fft.rs:19-56): the message (PDU) form — one `Vec<Complex>` of exactly `size`
/// samples in, its forward FFT out, tags passed along; a wrong-sized message is a fatal `Err` as in fft.rs:46-52.
pub struct GpuFft {
    h: Handle,
    src: rustradio::stream::NCReadStream<Vec<Complex>>,
    dst: rustradio::stream::NCWriteStream<Vec<Complex>>,
}
impl GpuFft {
    pub fn from_fft_size(src: rustradio::stream::NCReadStream<Vec<Complex>>, size: usize)
        -> Result<(Self, rustradio::stream::NCReadStream<Vec<Complex>>)> {
        if size == 0 { return Err(Error::msg("FFT called with size 0")); }
        // SAFETY: plain value.
        let h = Handle::new(unsafe { rr_fftstream_create(size) })?;
        let (dst, dr) = rustradio::stream::new_nocopy_stream();
        Ok((Self { h, src, dst }, dr))
    }
}
impl BlockName for GpuFft { fn block_name(&self) -> &str { "GpuFft" } }
impl BlockEOF for GpuFft { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuFft {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        loop {
            if self.dst.remaining() == 0 { return Ok(BlockRet::WaitForStream(&self.dst, 1)); }
            let Some((msg, tags)) = self.src.pop() else { return Ok(BlockRet::WaitForStream(&self.src, 1)); };
            let mut out = vec![Complex::default(); msg.len()];
            // SAFETY: msg and out are live slices of msg.len() repr(C) Complex<f32>.
            check(unsafe { rr_fft_process(self.h.0, msg.as_ptr(), msg.len(), out.as_mut_ptr()) })?;
            self.dst.push(out, tags);
        }
    }
}

/// Several reference blocks behind one handle and one fused kernel, one Complex / f32 input stream to one output stream:
/// `GpuFused::fm_chain` = FftFilter -> RationalResampler -> QuadratureDemod (rr_fm_chain_create,
/// examples/rtl_fm.rs:381-419), `fir_fm_chain` = FirFilter -> FftFilter -> RationalResampler -> QuadratureDemod
/// (rr_fir_fm_chain_create), `fir_fftfilter` = FirFilter -> FftFilter (rr_fir_fftfilter_create), `hilbert_fir` =
/// Hilbert -> FirFilter (rr_hilbert_fir_create, examples/ax25-1200-rx.rs:238-247).
/// Tags: what the reference blocks in sequence deliver (`rr_block_tag_rule`): a chain holding a RationalResampler or a
/// QuadratureDemod drops them (rational_resampler.rs:156), `fir_fftfilter` keeps a tag on its sample (fir.rs:536-545 with
/// deci 1, fft_filter.rs:307-313,343), `hilbert_fir` re-emits it at `pos / deci` (hilbert.rs:119-123, fir.rs:536-545).
pub struct GpuFused<I: Sample, O: Sample> {
    h: Handle,
    name: &'static str,
    src: ReadStream<I>,
    dst: WriteStream<O>,
    fwd: TagForwarder,
}
impl<I: Sample, O: Sample> GpuFused<I, O> {
    fn wrap(h: *mut RrBlock, name: &'static str, src: ReadStream<I>) -> Result<(Self, ReadStream<O>)> {
        let h = Handle::new(h)?;
        let fwd = TagForwarder::new(&h)?;
        let (dst, dr) = new_stream();
        Ok((Self { h, name, src, dst, fwd }, dr))
    }
}
impl GpuFused<Complex, Float> {
    /// `FftFilter -> RationalResampler -> QuadratureDemod` behind one handle: one kernel per call up to 16383 taps, the unfused
    /// composition of the three GPU blocks beyond (no tap-count limit, like src/fft_filter.rs:36-42).
    pub fn fm_chain(src: ReadStream<Complex>, taps: &[Complex], interp: usize, deci: usize, gain: Float, fast_math: bool)
        -> Result<(Self, ReadStream<Float>)> {
        // SAFETY: taps is a live slice of repr(C) Complex<f32>.
        Self::wrap(unsafe { rr_fm_chain_create(taps.as_ptr(), taps.len(), interp, deci, gain, fast_math as c_int) }, "GpuFmChain", src)
    }
    /// `FftFilter -> RationalResampler -> FastFM` (src/quadrature_demod.rs:144-165) behind one handle: RR_DEMOD_FASTFM = 2
    pub fn fm_chain_fastfm(src: ReadStream<Complex>, taps: &[Complex], interp: usize, deci: usize) -> Result<(Self, ReadStream<Float>)> {
        // SAFETY: taps is a live slice of repr(C) Complex<f32>.
        Self::wrap(unsafe { rr_fm_chain_create(taps.as_ptr(), taps.len(), interp, deci, 1.0, 2) }, "GpuFmChainFastFM", src)
    }
    pub fn fir_fm_chain(src: ReadStream<Complex>, fir_taps: &[Complex], fft_taps: &[Complex], interp: usize, deci: usize,
                        gain: Float, fast_math: bool) -> Result<(Self, ReadStream<Float>)> {
        // SAFETY: live slices.
        Self::wrap(unsafe {
            rr_fir_fm_chain_create(fir_taps.as_ptr(), fir_taps.len(), fft_taps.as_ptr(), fft_taps.len(), interp, deci, gain,
                                   fast_math as c_int)
        }, "GpuFirFmChain", src)
    }
}
impl GpuFused<Float, Float> {
    /// `FftFilterFloat -> RationalResampler -> MultiplyConst` (the audio stage of examples/rtl_fm.rs:398-418) as one kernel
    pub fn audio_chain(src: ReadStream<Float>, taps: &[Float], interp: usize, deci: usize, scale: Float) -> Result<(Self, ReadStream<Float>)> {
        // SAFETY: taps is a live slice.
        Self::wrap(unsafe { rr_audio_chain_create(taps.as_ptr(), taps.len(), interp, deci, scale) }, "GpuAudioChain", src)
    }
}
impl GpuFused<Complex, Complex> {
    pub fn fir_fftfilter(src: ReadStream<Complex>, fir_taps: &[Complex], fft_taps: &[Complex]) -> Result<(Self, ReadStream<Complex>)> {
        // SAFETY: live slices.
        Self::wrap(unsafe { rr_fir_fftfilter_create(fir_taps.as_ptr(), fir_taps.len(), fft_taps.as_ptr(), fft_taps.len()) },
                   "GpuFirFftFilter", src)
    }
    /// `MultiplyConst::<Complex>::new(src, val)` (src/multiply_const.rs:6-23)
    pub fn multiply_const(src: ReadStream<Complex>, val: Complex) -> Result<(Self, ReadStream<Complex>)> {
        Self::wrap(unsafe { rr_multiply_const_c32_create(val.re, val.im) }, "GpuMultiplyConst", src)
    }
}
impl GpuFused<Float, Complex> {
    /// `translate` = `Some((samp_rate, freq))` for `.translate()`; `replay_rotator` = true keeps the library default, the
    /// reference's own f32 rotator recurrence replayed bit for bit (RR_ROT_REPLAY, generated ahead on a side stream: by a
    /// device lane, or by a host thread for a block whose calls outrun that lane — include/rustradio_amd.h);
    /// false opts into the parallel closed-form model (RR_ROT_MODEL), which is outside 1e-5 parity beyond ~1e5 outputs.
    pub fn hilbert_fir(src: ReadStream<Float>, hilbert_ntaps: usize, window_type: &WindowType, taps: &[Complex], deci: usize,
                       translate: Option<(Float, Float)>, replay_rotator: bool) -> Result<(Self, ReadStream<Complex>)> {
        let (w, parm) = window_code(window_type);
        let (fs, f) = translate.unwrap_or((0.0, 0.0));
        // SAFETY: live slice, plain values.
        let p = unsafe { rr_hilbert_fir_create(hilbert_ntaps, w, parm, taps.as_ptr(), taps.len(), deci, translate.is_some() as c_int, fs, f) };
        if !p.is_null() && translate.is_some() {
            // SAFETY: p is a valid handle.
            unsafe { rr_fir_set_rotator_mode(p, replay_rotator as c_int) };
        }
        Self::wrap(p, "GpuHilbertFir", src)
    }
}
impl<I: Sample, O: Sample> BlockName for GpuFused<I, O> { fn block_name(&self) -> &str { self.name } }
impl<I: Sample, O: Sample> BlockEOF for GpuFused<I, O> { fn eof(&mut self) -> bool { self.src.eof() } }
impl<I: Sample, O: Sample> Block for GpuFused<I, O> {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, tags) = self.src.read_buf()?;
        let mut out = self.dst.write_buf()?;
        let (st, consumed, produced, need) = self.h.work(input.slice(), out.slice())?;
        let out_tags = self.fwd.step(tags, consumed, produced);
        input.consume(consumed);
        out.produce(produced, &out_tags);
        Ok(match st {
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            _ => BlockRet::Again,
        })
    }
}

/// `Tee` + N x (FftFilter -> RationalResampler -> QuadratureDemod) on ONE input (rr_fm_multi_create; the reference needs
/// a Tee tree, src/tee.rs:10-24): N output streams.  The C ABI takes the N output windows as one buffer of N x out_cap
/// elements, so the block stages them in `scratch` and copies each channel into its own stream (4 B per output sample).
pub struct GpuFmMulti {
    h: Handle,
    nchan: usize,
    src: ReadStream<Complex>,
    dsts: Vec<WriteStream<Float>>,
    scratch: Vec<Float>,
}
impl GpuFmMulti {
    /// `taps[c]` = channel c's filter (all of one length).
    pub fn new(src: ReadStream<Complex>, taps: &[Vec<Complex>], interp: usize, deci: usize, gain: Float, fast_math: bool)
        -> Result<(Self, Vec<ReadStream<Float>>)> {
        let nchan = taps.len();
        let ntaps = taps.first().map_or(0, Vec::len);
        if taps.iter().any(|t| t.len() != ntaps) { return Err(Error::msg("GpuFmMulti: all channels need the same number of taps")); }
        let flat: Vec<Complex> = taps.iter().flatten().copied().collect();
        // SAFETY: flat is a live [nchan][ntaps] array.
        let h = Handle::new(unsafe { rr_fm_multi_create(flat.as_ptr(), nchan, ntaps, interp, deci, gain, fast_math as c_int) })?;
        debug_assert_eq!(unsafe { rr_block_out_windows(h.0) }, nchan);
        let (dsts, drs): (Vec<_>, Vec<_>) = (0..nchan).map(|_| new_stream()).unzip();
        Ok((Self { h, nchan, src, dsts, scratch: Vec::new() }, drs))
    }
}
impl BlockName for GpuFmMulti { fn block_name(&self) -> &str { "GpuFmMulti" } }
impl BlockEOF for GpuFmMulti { fn eof(&mut self) -> bool { self.src.eof() } }
impl Block for GpuFmMulti {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, _tags) = self.src.read_buf()?;
        let mut outs = Vec::with_capacity(self.nchan);
        for d in &self.dsts { outs.push(d.write_buf()?); }
        let cap = outs.iter_mut().map(|o| o.slice().len()).min().unwrap_or(0);
        self.scratch.resize(self.nchan * cap.max(1), 0.0);
        let (mut c, mut p, mut need) = (0usize, 0usize, 0usize);
        // SAFETY: scratch holds nchan windows of `cap` elements; input is a live window.
        let st = unsafe {
            rr_block_work(self.h.0, input.slice().as_ptr().cast(), input.slice().len(), self.scratch.as_mut_ptr().cast(), cap,
                          &mut c, &mut p, &mut need)
        };
        if st == RR_ERR { return Err(last_error()); }
        for (ch, mut o) in outs.into_iter().enumerate() {
            o.slice()[..p].copy_from_slice(&self.scratch[ch * cap..ch * cap + p]);
            o.produce(p, &[]);
        }
        input.consume(c);
        Ok(if st == RR_WAIT_DST { BlockRet::WaitForStream(&self.dsts[0], need) } else { BlockRet::WaitForStream(&self.src, need) })
    }
}

// ---- device-resident streams (include/rustradio_amd.h rr_dstream_*) ---------------------------------------
// The reference's stream is ONE ring behind TWO handles, `WriteStream<T>` and `ReadStream<T>`, each an `Arc` of it
// (src/stream.rs:187-190,256-258); an end is `closed()` when the other handle has been dropped (strong count 1,
// :148-150,166-168), `ReadStream::eof()` = writer dropped AND ring empty (:237-246), and `wait(need)` is true when `need`
// can never be met (:222-224,311-313).  `Graph::run` (src/graph.rs:126-147) and `MTGraph` (src/mtgraph.rs:98-116) end a
// block on exactly those three facts, so the HBM rings carry them too: `new_gpu_stream()` returns the same two handles,
// dropping one calls `rr_dstream_close(side)`, and both implement `StreamWait`.  The ring's counters live in the library
// under the ring's own lock, so the two handles may sit in blocks on different threads (MTGraph).

/// Tags of an HBM ring: the host-side side-band the reference's `Buffer` keeps next to its samples
/// (src/nowasm/circular_buffer.rs:518-557: `produce()` files the tags of the samples it publishes, `consume()` drops those
/// of the samples it retires).  Positions are counted from the start of the stream with the side-band's OWN totals.
#[derive(Default)]
struct TagBand {
    posted: u64,             // samples whose tags are filed (writer side)
    taken: u64,              // samples whose tags have been handed on (reader side)
    writer_gone: bool,
    tags: Vec<(u64, Tag)>,
}
/// The ring both ends share (the reference's `Arc<Buffer<T>>`); destroyed with the last handle.
struct GpuRing<T: Sample> {
    s: *mut RrDStream,
    band: Mutex<TagBand>,    // tags only: the ring's counters and their lock live in the library
    band_cv: Condvar,
    _t: std::marker::PhantomData<T>,
}
impl<T: Sample> GpuRing<T> {
    /// Writer side: the next `n` samples of the stream carry `tags` (positions relative to the first of them).
    fn post(&self, n: usize, tags: &[Tag]) {
        if n == 0 { return; }
        let mut b = self.band.lock().unwrap();
        let base = b.posted;
        for t in tags.iter().filter(|t| t.pos() < n) { b.tags.push((base + t.pos() as u64, t.clone())); }
        b.posted += n as u64;
        drop(b);
        self.band_cv.notify_all();
    }
    /// Reader side: the tags of the next `n` samples, relative to the first of them; the samples are retired.
    /// `rr_block_work_streams` publishes a block's output inside the library BEFORE the block can post the tags, so a
    /// reader may hold samples whose tags are a moment away: wait until the side-band covers what was consumed (the
    /// writer posts right after its call returns, or is gone).
    fn take(&self, n: usize) -> Vec<Tag> {
        if n == 0 { return Vec::new(); }
        let mut b = self.band.lock().unwrap();
        while b.posted < b.taken + n as u64 && !b.writer_gone { b = self.band_cv.wait(b).unwrap(); }
        let (taken, limit) = (b.taken, b.taken + n as u64);
        let (now, keep): (Vec<_>, Vec<_>) = b.tags.drain(..).partition(|(abs, _)| *abs < limit);
        b.tags = keep;
        b.taken = limit;
        now.into_iter().map(|(abs, t)| Tag::new((abs - taken) as usize, t.key(), t.val().clone())).collect()
    }
    fn writer_dropped(&self) {
        self.band.lock().unwrap().writer_gone = true;
        self.band_cv.notify_all();
    }
}
// SAFETY: every rr_dstream_* entry point and rr_block_work_streams lock the ring inside the library.
unsafe impl<T: Sample> Send for GpuRing<T> {}
unsafe impl<T: Sample> Sync for GpuRing<T> {}
impl<T: Sample> Drop for GpuRing<T> {
    fn drop(&mut self) {
        // SAFETY: created by rr_dstream_create, destroyed once (the Arc's last owner).
        unsafe { rr_dstream_destroy(self.s) }
    }
}
/// How long one `StreamWait::wait` blocks at most before the runner's loop gets to look again.
const WAIT_SLICE_MS: c_uint = 100;

/// Writing end of an HBM ring (`WriteStream<T>`, src/stream.rs:256-313).
pub struct GpuWriteStream<T: Sample> { ring: Arc<GpuRing<T>> }
/// Reading end of an HBM ring (`ReadStream<T>`, src/stream.rs:187-246).
pub struct GpuReadStream<T: Sample> { ring: Arc<GpuRing<T>> }

/// `new_stream()` in HBM (src/stream.rs:336-339); `capacity_bytes` as `DEFAULT_STREAM_SIZE` (:105).
pub fn new_gpu_stream<T: Sample>(capacity_bytes: usize) -> Result<(GpuWriteStream<T>, GpuReadStream<T>)> {
    // SAFETY: plain values.
    let s = unsafe { rr_dstream_create(std::mem::size_of::<T>(), capacity_bytes) };
    if s.is_null() { return Err(last_error()); }
    let ring = Arc::new(GpuRing { s, band: Mutex::new(TagBand::default()), band_cv: Condvar::new(), _t: std::marker::PhantomData });
    Ok((GpuWriteStream { ring: ring.clone() }, GpuReadStream { ring }))
}
impl<T: Sample> Drop for GpuWriteStream<T> {
    // SAFETY (both drops): the ring outlives the handle (Arc); the flag is what `closed()` / `eof()` / `wait()` of the other end read.
    fn drop(&mut self) { unsafe { rr_dstream_close(self.ring.s, RR_SIDE_WRITER); } self.ring.writer_dropped(); }
}
impl<T: Sample> Drop for GpuReadStream<T> {
    fn drop(&mut self) { unsafe { rr_dstream_close(self.ring.s, RR_SIDE_READER); } }
}
impl<T: Sample> GpuWriteStream<T> {
    #[must_use] pub fn capacity(&self) -> usize { unsafe { rr_dstream_capacity(self.ring.s) } }
    /// `WriteStream::free()` (src/stream.rs:274-276).
    #[must_use] pub fn free(&self) -> usize { unsafe { rr_dstream_write_buf(self.ring.s, std::ptr::null_mut(), std::ptr::null_mut()) } }
}
impl<T: Sample> GpuReadStream<T> {
    #[must_use] pub fn capacity(&self) -> usize { unsafe { rr_dstream_capacity(self.ring.s) } }
    #[must_use] pub fn readable(&self) -> usize { unsafe { rr_dstream_read_buf(self.ring.s, std::ptr::null_mut()) } }
    /// `ReadStream::eof()` (src/stream.rs:237-246): the writer is gone and nothing is left to read.
    #[must_use] pub fn eof(&self) -> bool {
        // the flag first: once the writer is closed `readable()` can only fall
        unsafe { rr_dstream_closed(self.ring.s, RR_SIDE_WRITER) != 0 } && self.readable() == 0
    }
}
fn ring_wait(s: *mut RrDStream, side: c_int, need: usize) -> bool {
    let mut never: c_int = 0;
    // SAFETY: valid ring; `never` is a live out-parameter.
    unsafe { rr_dstream_wait(s, side, need, WAIT_SLICE_MS, &mut never) };
    never != 0
}
#[cfg_attr(feature = "async", async_trait::async_trait)]
impl<T: Sample + Sync + Send + 'static> StreamWait for GpuReadStream<T> {
    fn id(&self) -> usize { unsafe { rr_dstream_id(self.ring.s) } }
    fn wait(&self, need: usize) -> bool { ring_wait(self.ring.s, RR_SIDE_READER, need) }
    fn closed(&self) -> bool { unsafe { rr_dstream_closed(self.ring.s, RR_SIDE_WRITER) != 0 } }
    #[cfg(feature = "async")]
    async fn wait_async(&self, need: usize) -> bool { self.wait(need) }
}
#[cfg_attr(feature = "async", async_trait::async_trait)]
impl<T: Sample + Sync + Send + 'static> StreamWait for GpuWriteStream<T> {
    fn id(&self) -> usize { unsafe { rr_dstream_id(self.ring.s) } }
    fn wait(&self, need: usize) -> bool { ring_wait(self.ring.s, RR_SIDE_WRITER, need) }
    fn closed(&self) -> bool { unsafe { rr_dstream_closed(self.ring.s, RR_SIDE_READER) != 0 } }
    #[cfg(feature = "async")]
    async fn wait_async(&self, need: usize) -> bool { self.wait(need) }
}
fn check(rc: c_int) -> Result<()> { if rc == RR_ERR { Err(last_error()) } else { Ok(()) } }

/// Host ring -> HBM ring (graph edge): `fill_from_slice` + `produce` across PCIe.  Register the host ring once with
/// `register_ring` (the reference's ring is one stable mapping) and the copies run as direct DMA.
pub struct GpuUpload<T: Sample> {
    src: ReadStream<T>,
    dst: GpuWriteStream<T>,
}
impl<T: Sample> GpuUpload<T> {
    /// `let (up, on_gpu) = GpuUpload::new(prev, 256 << 20)?;` — the block and the reading end of its HBM ring.
    pub fn new(src: ReadStream<T>, capacity_bytes: usize) -> Result<(Self, GpuReadStream<T>)> {
        let (dst, dr) = new_gpu_stream(capacity_bytes)?;
        Ok((Self { src, dst }, dr))
    }
}
impl<T: Sample> BlockName for GpuUpload<T> { fn block_name(&self) -> &str { "GpuUpload" } }
impl<T: Sample> BlockEOF for GpuUpload<T> { fn eof(&mut self) -> bool { self.src.eof() } }
impl<T: Sample + Sync + Send + 'static> Block for GpuUpload<T> {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (input, tags) = self.src.read_buf()?;
        let have = input.slice().len();
        let n = have.min(self.dst.free());
        if n == 0 {
            // starved -> wait on the source; ring full -> wait on the ring (its reader dropping ends this block too)
            return Ok(if have == 0 { BlockRet::WaitForStream(&self.src, 1) } else { BlockRet::WaitForStream(&self.dst, 1) });
        }
        // SAFETY: input is a live window of n elements; the ring has room for n (only this block writes it).
        check(unsafe { rr_dstream_copy_in(self.dst.ring.s, 0, input.slice().as_ptr().cast(), n, std::ptr::null_mut()) })?;
        self.dst.ring.post(n, &tags);   // tags move with their samples (pos < n; the rest stay in the host ring)
        check(unsafe { rr_dstream_produce(self.dst.ring.s, n) })?;
        input.consume(n);
        Ok(BlockRet::Again)
    }
}
/// HBM ring -> host ring (graph edge).
pub struct GpuDownload<T: Sample> {
    src: GpuReadStream<T>,
    dst: WriteStream<T>,
}
impl<T: Sample> GpuDownload<T> {
    pub fn new(src: GpuReadStream<T>) -> (Self, ReadStream<T>) {
        let (dst, dr) = new_stream();
        (Self { src, dst }, dr)
    }
}
impl<T: Sample> BlockName for GpuDownload<T> { fn block_name(&self) -> &str { "GpuDownload" } }
impl<T: Sample> BlockEOF for GpuDownload<T> { fn eof(&mut self) -> bool { self.src.eof() } }
impl<T: Sample + Sync + Send + 'static> Block for GpuDownload<T> {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let mut out = self.dst.write_buf()?;
        let have = self.src.readable();
        let n = out.slice().len().min(have);
        if n == 0 {
            return Ok(if have == 0 { BlockRet::WaitForStream(&self.src, 1) } else { BlockRet::WaitForStream(&self.dst, 1) });
        }
        // SAFETY: out is a live window with room for n elements; the ring holds n readable elements (only this block reads it).
        check(unsafe { rr_dstream_copy_out(self.src.ring.s, 0, out.slice().as_mut_ptr().cast(), n, std::ptr::null_mut()) })?;
        check(unsafe { rr_dstream_consume(self.src.ring.s, n) })?;
        out.produce(n, &self.src.ring.take(n));
        Ok(BlockRet::Again)
    }
}
/// One GPU block between two HBM rings: `Block::work()` without a PCIe hop (rr_block_work_streams).  The status is the
/// block's own — `WaitForStream(src, need)` when starved, `WaitForStream(dst, need)` when the output ring is full — so
/// both runners end it the reference's way: upstream dropped ∧ ring drained ∧ (resampler) no pending sample.
/// Tags are re-based from `consumed` / `produced` by the reference rule of the block behind the handle
/// (`rr_block_tag_rule`): FirFilter `pos / deci` (fir.rs:536-545), FftFilter's waiting list (fft_filter.rs:307-313,343),
/// Hilbert `pos < n` (hilbert.rs:119-123), FftStream's frame tags; RationalResampler / QuadratureDemod drop them.
pub struct GpuResident<I: Sample, O: Sample> {
    h: Handle,
    name: &'static str,
    src: GpuReadStream<I>,
    dst: GpuWriteStream<O>,
    fwd: TagForwarder,
}
impl<I: Sample, O: Sample> GpuResident<I, O> {
    /// `create` = any `rr_*_create` call, e.g. `|| unsafe { rr_fftfilter_create(taps.as_ptr(), taps.len()) }`.
    /// Returns the block and the reading end of its output ring (`out_capacity_bytes` in HBM).
    pub fn new(create: impl FnOnce() -> *mut RrBlock, name: &'static str, src: GpuReadStream<I>,
               out_capacity_bytes: usize) -> Result<(Self, GpuReadStream<O>)> {
        let h = Handle::new(create())?;
        let fwd = TagForwarder::new(&h)?;
        let (dst, dr) = new_gpu_stream(out_capacity_bytes)?;
        Ok((Self { h, name, src, dst, fwd }, dr))
    }
}
impl<I: Sample, O: Sample> BlockName for GpuResident<I, O> { fn block_name(&self) -> &str { self.name } }
impl<I: Sample, O: Sample> BlockEOF for GpuResident<I, O> {
    fn eof(&mut self) -> bool {
        // SAFETY: valid handle.  rr_block_eof adds the block's own condition (src/rational_resampler.rs:209-213).
        unsafe { rr_block_eof(self.h.0, self.src.eof() as c_int) != 0 }
    }
}
impl<I: Sample + Sync + Send + 'static, O: Sample + Sync + Send + 'static> Block for GpuResident<I, O> {
    fn work(&mut self) -> Result<BlockRet<'_>> {
        let (mut c, mut p, mut need) = (0usize, 0usize, 0usize);
        // SAFETY: valid handles; counts are final on return (they depend on lengths only), kernels run asynchronously.
        let st = unsafe {
            rr_block_work_streams(self.h.0, self.src.ring.s, self.dst.ring.s, &mut c, &mut p, &mut need, std::ptr::null_mut())
        };
        check(st)?;
        let out_tags = self.fwd.step(self.src.ring.take(c), c, p);
        self.dst.ring.post(p, &out_tags);
        Ok(match st {
            RR_WAIT_SRC => BlockRet::WaitForStream(&self.src, need),
            RR_WAIT_DST => BlockRet::WaitForStream(&self.dst, need),
            RR_EOF => BlockRet::EOF,
            _ => BlockRet::Again,
        })
    }
}

// ---- multi-GPU fan-out (include/rustradio_amd.h rr_fanout_*) --------------------------------------------------
/// The shared source of a channel-sharded graph, one process per GPU: replaces the in-process `Tee` tree
/// (src/tee.rs:10-24) by a double buffer in HBM on every rank and an RCCL broadcast per tile on a communication stream.
/// The owning rank's source writes tile `t` (`produce_buf` / `submit`), every rank's blocks read it
/// (`acquire` / `release`) while tile `t + 1` is in flight.
pub struct GpuFanout {
    f: *mut RrFanout,
    next: u64,
}
// SAFETY: the handle has no thread affinity; it is driven by one graph thread.
unsafe impl Send for GpuFanout {}
impl Drop for GpuFanout {
    fn drop(&mut self) {
        // SAFETY: created by rr_fanout_create, destroyed once.
        unsafe { rr_fanout_destroy(self.f) }
    }
}
impl GpuFanout {
    /// The 128-byte group id: made on the owning rank, shipped to the other processes by the application.
    pub fn unique_id() -> Result<[u8; 128]> {
        let mut id = [0u8; 128];
        // SAFETY: 128 writable bytes.
        check(unsafe { rr_fanout_unique_id(id.as_mut_ptr().cast()) })?;
        Ok(id)
    }
    /// Collective: every rank of the group calls it (`id` may be `None` for a one-rank group).
    pub fn new(id: Option<&[u8; 128]>, rank: usize, world: usize, src_rank: usize, tile_bytes: usize) -> Result<Self> {
        let p = id.map_or(std::ptr::null(), |i| i.as_ptr().cast());
        // SAFETY: p is null or 128 readable bytes.
        let f = unsafe { rr_fanout_create(p, rank as c_int, world as c_int, src_rank as c_int, tile_bytes, 0) };
        if f.is_null() { Err(last_error()) } else { Ok(Self { f, next: 0 }) }
    }
    /// Owning rank: device pointer the source writes the next tile to (its stream waits for the half to be free).
    pub fn produce_buf(&mut self, producer_stream: *mut c_void) -> Result<*mut c_void> {
        // SAFETY: valid handle.
        let p = unsafe { rr_fanout_produce_buf(self.f, self.next, producer_stream) };
        if p.is_null() { Err(last_error()) } else { Ok(p) }
    }
    /// Collective, once per tile in order: the broadcast of the next tile; returns its index.
    pub fn submit(&mut self, producer_stream: *mut c_void) -> Result<u64> {
        // SAFETY: valid handle.
        check(unsafe { rr_fanout_submit(self.f, self.next, producer_stream) })?;
        self.next += 1;
        Ok(self.next - 1)
    }
    /// Device pointer of tile `t`; `compute_stream` waits for its broadcast.
    pub fn acquire(&mut self, t: u64, compute_stream: *mut c_void) -> Result<*const c_void> {
        // SAFETY: valid handle.
        let p = unsafe { rr_fanout_acquire(self.f, t, compute_stream) };
        if p.is_null() { Err(last_error()) } else { Ok(p) }
    }
    /// The blocks' reads of tile `t` are enqueued on `compute_stream`: its half may be overwritten after them.
    pub fn release(&mut self, t: u64, compute_stream: *mut c_void) -> Result<()> {
        // SAFETY: valid handle.
        check(unsafe { rr_fanout_release(self.f, t, compute_stream) })
    }
    /// (summed broadcast ms, broadcasts timed) since the last call; waits for the communication stream.
    pub fn stats(&mut self) -> Result<(f64, usize)> {
        let (mut ms, mut n) = (0f64, 0usize);
        // SAFETY: valid handle, two writable scalars.
        check(unsafe { rr_fanout_stats(self.f, &mut ms, &mut n) })?;
        Ok((ms, n))
    }
}

/// Page-lock a host ring the blocks will be handed windows of (rr_host_register): once per stream, at creation.
/// Zero-copy windows need a page-aligned range of whole pages — the reference's ring is one (circular_buffer.rs:98-128).
pub fn register_ring(base: *mut u8, bytes: usize) -> Result<()> { check(unsafe { rr_host_register(base.cast(), bytes) }) }
/// Whether `rr_block_work` lets kernels work in place on this host window (else it is staged through device memory).
pub fn window_in_place(ptr: *const u8, bytes: usize) -> bool { unsafe { rr_host_window_in_place(ptr.cast(), bytes) != 0 } }
pub fn unregister_ring(base: *mut u8) -> Result<()> { check(unsafe { rr_host_unregister(base.cast()) }) }
/// Wait for everything a handle enqueued (device-pointer work is asynchronous).
pub fn sync_block(h: *mut RrBlock) -> Result<()> { check(unsafe { rr_block_sync(h) }) }
/// `Block::work()` on raw device windows (rr_block_work_dev), for graphs that manage their own device memory.
///
/// # Safety
/// `d_in` / `d_out` must be device pointers to `in_len` / `out_cap` elements of the block's element types.
pub unsafe fn work_dev(h: *mut RrBlock, d_in: *const c_void, in_len: usize, d_out: *mut c_void, out_cap: usize,
                       hip_stream: *mut c_void) -> Result<(c_int, usize, usize, usize)> {
    let (mut c, mut p, mut need) = (0usize, 0usize, 0usize);
    let st = unsafe { rr_block_work_dev(h, d_in, in_len, d_out, out_cap, &mut c, &mut p, &mut need, hip_stream) };
    if st == RR_ERR { Err(last_error()) } else { Ok((st, c, p, need)) }
}

pub fn window_code(w: &WindowType) -> (c_int, f32) {
    match w {
        WindowType::Hamming => (0, 0.0),
        WindowType::Blackman => (1, 0.0),
        WindowType::BlackmanHarris => (2, 0.0),
        WindowType::HammingParm(p) => (3, *p),
    }
}
