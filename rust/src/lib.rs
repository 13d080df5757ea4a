//! GPU (MI355X / gfx950) replacements for rustradio's hot-path blocks, behind the
//! unchanged `Block` / `ReadStream` / `WriteStream` API: a graph from `examples/` swaps
//! `FftFilter::new(prev, taps)` for `GpuFftFilter::new(prev, taps)` and nothing else.
//!
//! SOURCE ONLY: the build image has no cargo/rustc, so this file is not compiled in CI.
//! It binds the C ABI of `include/rustradio_amd.h` one to one; the C++ mirror
//! `rustradio_amd/host/rustradio.hpp` implements the same shim logic and IS tested
//! (`tests/cpp/test_host_api.cpp`).
use std::ffi::{c_int, c_void, CStr};

use rustradio::block::{Block, BlockEOF, BlockName, BlockRet};
use rustradio::stream::{new_stream, ReadStream, Tag, WriteStream};
use rustradio::window::WindowType;
use rustradio::{Complex, Error, Float, Result, Sample};

#[repr(C)]
pub struct RrBlock {
    _private: [u8; 0],
}

// enum rr_status (include/rustradio_amd.h)
const RR_AGAIN: c_int = 0;
const RR_WAIT_SRC: c_int = 1;
const RR_WAIT_DST: c_int = 2;
const RR_ERR: c_int = -1;

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
    fn rr_block_destroy(b: *mut RrBlock);
    fn rr_block_work(b: *mut RrBlock, inp: *const c_void, in_len: usize, out: *mut c_void, out_cap: usize,
                     consumed: *mut usize, produced: *mut usize, need: *mut usize) -> c_int;
    fn rr_block_eof(b: *mut RrBlock, src_eof: c_int) -> c_int;
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

/// `FftFilter` on the GPU (replaces `rustradio::blocks::FftFilter`, src/fft_filter.rs:210-355).
pub struct GpuFftFilter {
    h: Handle,
    src: ReadStream<Complex>,
    dst: WriteStream<Complex>,
    pending_tags: Vec<(u64, Tag)>,
    in_abs: u64,
    out_abs: u64,
}
impl GpuFftFilter {
    pub fn new<T: Into<Vec<Complex>>>(src: ReadStream<Complex>, taps: T) -> Result<(Self, ReadStream<Complex>)> {
        let taps = taps.into();
        // SAFETY: taps is a live slice of repr(C) Complex<f32>.
        let h = Handle::new(unsafe { rr_fftfilter_create(taps.as_ptr(), taps.len()) })?;
        let (dst, dr) = new_stream();
        Ok((Self { h, src, dst, pending_tags: Vec::new(), in_abs: 0, out_abs: 0 }, dr))
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
        for t in tags.into_iter().filter(|t| t.pos() < consumed) {
            self.pending_tags.push((self.in_abs + t.pos() as u64, t));
        }
        let limit = self.out_abs + produced as u64;
        let (emit, keep): (Vec<_>, Vec<_>) = self.pending_tags.drain(..).partition(|(abs, _)| *abs < limit);
        self.pending_tags = keep;
        let out_tags: Vec<Tag> = emit.into_iter()
            .map(|(abs, t)| Tag::new((abs - self.out_abs) as usize, t.key(), t.val().clone()))
            .collect();
        self.in_abs += consumed as u64;
        self.out_abs += produced as u64;
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

// The Complex-input fused chain (rr_fm_chain_create), the fused Hilbert -> FirFilter (rr_hilbert_fir_create),
// Hilbert (tags with `pos < n` kept), FftFilterFloat and FftStream (frame tags added from `produced`) follow
// the same patterns: see the C++ mirror for the exact work() bodies — rustradio_amd/host/rustradio.hpp — and
// INTEGRATION.md.
pub fn window_code(w: &WindowType) -> (c_int, f32) {
    match w {
        WindowType::Hamming => (0, 0.0),
        WindowType::Blackman => (1, 0.0),
        WindowType::BlackmanHarris => (2, 0.0),
        WindowType::HammingParm(p) => (3, *p),
    }
}
