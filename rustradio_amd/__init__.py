"""rustradio_amd — MI355X-native implementation of rustradio's streaming-DSP hot path.

Host-side mirror (Python, over the C ABI of include/rustradio_amd.h) of the reference's
block interface for this path: the constructors take what `X::new(src, ...)` / the
builders take (minus the stream, which the caller owns), and `work()` is
`Block::work()` over explicit stream windows, returning the BlockRet variant plus
what the reference would pass to `consume()`/`produce()`.

    FirFilter            src/fir.rs:303-551      (builder: taps, .deci(), .translate())
    FftFilter            src/fft_filter.rs:210-355
    FftFilterFloat       src/fft_filter.rs:365-491
    RationalResampler    src/rational_resampler.rs:100-213
    QuadratureDemod      src/quadrature_demod.rs:32-114
    RtlSdrDecode         src/rtlsdr_decode.rs:9-47
    Hilbert              src/hilbert.rs:22-129
    low_pass / low_pass_complex / hilbert_taps / make_window   src/fir.rs:594-680, src/window.rs

All sample arithmetic runs in HIP kernels on the GPU; nothing here computes samples.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

import contextlib
import threading

from ._lib import BuildOpts, lib as _raw_lib, last_error, LIB_PATH  # noqa: F401

# BlockRet (src/block.rs:12-70)
AGAIN, WAIT_SRC, WAIT_DST, EOF, PENDING, ERR = 0, 1, 2, 3, 4, -1
# WindowType (src/window.rs:42-60)
WIN_HAMMING, WIN_BLACKMAN, WIN_BLACKMAN_HARRIS, WIN_HAMMING_PARM = 0, 1, 2, 3
ATAN2_EXACT, ATAN2_FAST = 0, 1
ROT_MODEL, ROT_REPLAY, ROT_REPLAY_DEVICE, ROT_REPLAY_HOST = 0, 1, 2, 3
DEMOD_FASTFM = 2        # fused-chain constructors: FastFM in the demodulator's place (RR_DEMOD_FASTFM)
DEFAULT_STREAM_SIZE = 4_096_000  # bytes, src/stream.rs:105


# ---- per-block path overrides (rr_build_opts) ---------------------------------------------
# `build_options(...)` is a context manager: every block / DeviceStream created inside it is built with these
# overrides (each create call gets its own rr_next_create_options; nothing is read from the environment).
PATH_AUTO, PATH_DIRECT, PATH_FFT = 0, 1, 2
_build_opts: dict = {}                 # process-wide defaults (tests patch this through harness.knob)
_tls = threading.local()               # build_options() is per THREAD, like the C side's pending overrides
_PATHS = {"auto": PATH_AUTO, "direct": PATH_DIRECT, "fft": PATH_FFT}
# the create calls that CONSUME rr_next_create_options (abi.cpp OptsScope): every block constructor and rr_dstream_create.
# rr_fanout_create does not — handing it the overrides would leave them pending for the thread's next block
_NO_OPTS_CREATES = ("rr_fanout_create",)


def _effective_opts() -> dict:
    o = getattr(_tls, "opts", None)
    return _build_opts if o is None else o


@contextlib.contextmanager
def build_options(**kw):
    """fir_path='direct'|'fft', fir_prune=+-1, fir_half=-1, fir_cfg=0..7, fft_log2f=10..14, fft_no_split=1,
    fftfloat_complex=1, fm_full=1, fm_poly=-1 (8 / 12: the multi-channel kernel variant), dstream_no_vmm=1, fir_poly=+-1,
    fft_nonfinite_tiles=1 (none) | 3 (in the tile kernel's tail), host_in_staged=+-1"""
    prev = getattr(_tls, "opts", None)
    _tls.opts = dict(_effective_opts(), **kw)
    try:
        yield
    finally:
        _tls.opts = prev


class _CreateProxy:
    """the loaded library; rr_*_create / rr_dstream_create first hand the pending overrides to the C ABI"""

    def __getattr__(self, name):
        f = getattr(_raw_lib(), name)
        opts = _effective_opts()
        if not (name.endswith("_create") and name not in _NO_OPTS_CREATES and opts):
            return f

        def create(*a):
            o = BuildOpts()
            for k, v in opts.items():
                if k == "fir_path":
                    v = _PATHS.get(v, v)
                if k == "fir_cfg":
                    k, v = "fir_cfg_plus1", int(v) + 1
                setattr(o, k, int(v))
            if _raw_lib().rr_next_create_options(C.byref(o)) != 0:
                raise ValueError(last_error())
            return f(*a)
        return create


_proxy = _CreateProxy()


def lib():
    return _proxy


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    return lib().rr_device_count()


def set_device(ordinal: int) -> None:
    if lib().rr_set_device(ordinal) != 0:
        raise ValueError(last_error())


# ---- tap designers --------------------------------------------------------------------
def make_window(wtype: int, ntaps: int, parm: float = 0.0) -> np.ndarray:
    out = np.zeros(ntaps, np.float32)
    if lib().rr_make_window(wtype, parm, ntaps, _ptr(out)) != 0:
        raise ValueError(last_error())
    return out


def compute_ntaps(samp_rate, twidth, wtype=WIN_HAMMING) -> int:
    return lib().rr_compute_ntaps(samp_rate, twidth, wtype)


def low_pass(samp_rate, cutoff, twidth, wtype=WIN_HAMMING, parm=0.0) -> np.ndarray:
    n = lib().rr_low_pass(samp_rate, cutoff, twidth, wtype, parm, None, 0)
    if n == 0:
        raise ValueError(last_error())
    out = np.zeros(n, np.float32)
    lib().rr_low_pass(samp_rate, cutoff, twidth, wtype, parm, _ptr(out), n)
    return out


def low_pass_complex(samp_rate, cutoff, twidth, wtype=WIN_HAMMING, parm=0.0) -> np.ndarray:
    n = lib().rr_low_pass_complex(samp_rate, cutoff, twidth, wtype, parm, None, 0)
    if n == 0:
        raise ValueError(last_error())
    out = np.zeros(n, np.complex64)
    lib().rr_low_pass_complex(samp_rate, cutoff, twidth, wtype, parm, _ptr(out), n)
    return out


def multiband(bands, window):
    """fir::multiband(bands, taps, window) (src/fir.rs:552-590) -> complex64 taps, or None (the reference's None)"""
    b = np.ascontiguousarray(bands, np.float32).reshape(-1, 2)
    w = np.ascontiguousarray(window, np.float32)
    out = np.zeros(len(w), np.complex64)
    return out if lib().rr_multiband(_ptr(b), len(b), _ptr(w), len(w), _ptr(out)) == 0 else None


def hilbert_taps(window: np.ndarray) -> np.ndarray:
    w = np.ascontiguousarray(window, np.float32)
    out = np.zeros(len(w), np.float32)
    if lib().rr_hilbert_taps(_ptr(w), len(w), _ptr(out)) != 0:
        raise ValueError(last_error())
    return out


# ---- blocks ---------------------------------------------------------------------------
class Block:
    """One block instance (an opaque `rr_block`)."""

    def __init__(self, handle, in_dtype, out_dtype):
        if not handle:
            raise ValueError(last_error())
        self._h = handle
        self.in_dtype = np.dtype(in_dtype)
        self.out_dtype = np.dtype(out_dtype)
        assert lib().rr_block_in_elem_size(handle) == self.in_dtype.itemsize
        assert lib().rr_block_out_elem_size(handle) == self.out_dtype.itemsize

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().rr_block_destroy(h)
            except Exception:
                pass

    @property
    def name(self) -> str:
        """BlockName::block_name (src/block.rs:91-97)."""
        return lib().rr_block_name(self._h).decode()

    def work(self, inp: np.ndarray, out_cap: int):
        """Block::work() over host windows -> (status, consumed, produced, need, out[:produced])."""
        inp = np.ascontiguousarray(inp, self.in_dtype)
        nw = lib().rr_block_out_windows(self._h)
        out = np.zeros(max(out_cap, 1) * nw, self.out_dtype)
        c, p, n = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        st = lib().rr_block_work(self._h, _ptr(inp), len(inp), _ptr(out), out_cap,
                                 C.byref(c), C.byref(p), C.byref(n))
        if st == ERR:
            raise RuntimeError(last_error())
        if nw > 1:      # multi-output block: (windows, produced)
            return st, c.value, p.value, n.value, out.reshape(nw, max(out_cap, 1))[:, :p.value].copy()
        return st, c.value, p.value, n.value, out[:p.value]

    def work_into(self, inp: np.ndarray, out: np.ndarray, out_cap: int):
        """Block::work() over caller-owned host windows (no allocation, no copy on the Python side): exactly the
        call the Rust shim makes with read_buf() / write_buf() slices -> (status, consumed, produced, need)"""
        assert inp.dtype == self.in_dtype and out.dtype == self.out_dtype and inp.flags.c_contiguous and out.flags.c_contiguous
        assert len(out) >= out_cap * lib().rr_block_out_windows(self._h)
        c, p, n = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        st = lib().rr_block_work(self._h, _ptr(inp), len(inp), _ptr(out), out_cap, C.byref(c), C.byref(p), C.byref(n))
        if st == ERR:
            raise RuntimeError(last_error())
        return st, c.value, p.value, n.value

    def work_dev(self, d_in: int, in_len: int, d_out: int, out_cap: int, stream: int = 0):
        """Block::work() over DEVICE windows (raw device pointers, e.g. tensor.data_ptr());
        asynchronous on `stream` (a hipStream_t handle used as given; 0 = the default stream, which is
        what torch.cuda.current_stream().cuda_stream returns for torch's default stream)
        -> (status, consumed, produced, need)."""
        c, p, n = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        st = lib().rr_block_work_dev(self._h, C.c_void_p(d_in), in_len, C.c_void_p(d_out), out_cap,
                                     C.byref(c), C.byref(p), C.byref(n), C.c_void_p(stream))
        if st == ERR:
            raise RuntimeError(last_error())
        return st, c.value, p.value, n.value

    def work_streams(self, src: "DeviceStream", dst: "DeviceStream", stream: int = 0):
        """Block::work() between two device-resident streams (windows, consume and produce handled by the
        library) -> (status, consumed, produced, need); asynchronous on `stream`."""
        c, p, n = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        st = lib().rr_block_work_streams(self._h, src._h, dst._h, C.byref(c), C.byref(p), C.byref(n), C.c_void_p(stream))
        if st == ERR:
            raise RuntimeError(last_error())
        return st, c.value, p.value, n.value

    def eof(self, src_eof: bool) -> bool:
        """BlockEOF::eof (src/block.rs:103-110)."""
        return bool(lib().rr_block_eof(self._h, int(src_eof)))

    def set_rotator_mode(self, mode: int) -> None:
        """rr_fir_set_rotator_mode on a translating FirFilter / HilbertFir: ROT_REPLAY (default, the reference's recurrence) or ROT_MODEL"""
        if lib().rr_fir_set_rotator_mode(self._h, mode) != 0:
            raise ValueError(last_error())

    def sync(self) -> None:
        if lib().rr_block_sync(self._h) != 0:
            raise RuntimeError(last_error())

    def set_profiling(self, on: bool) -> None:
        lib().rr_block_set_profiling(self._h, int(on))

    def profile(self, reset: bool = True):
        """-> (summed dominant-kernel time in ms, launches), from HIP events on the launch stream."""
        ms, n = C.c_double(0), C.c_size_t(0)
        if lib().rr_block_profile(self._h, C.byref(ms), C.byref(n), int(reset)) != 0:
            raise RuntimeError(last_error())
        return ms.value, n.value


def host_ring(nbytes: int) -> np.ndarray:
    """a page-aligned, whole-pages host buffer (one private anonymous mapping, like the reference's ring:
    src/nowasm/circular_buffer.rs:98-128) as a uint8 array of at least `nbytes` — the only kind of range rr_host_register
    grants zero-copy windows on (include/rustradio_amd.h)"""
    import mmap
    size = -(-max(int(nbytes), 1) // mmap.PAGESIZE) * mmap.PAGESIZE
    return np.frombuffer(mmap.mmap(-1, size), dtype=np.uint8)


def host_register(a: np.ndarray) -> None:
    """page-lock a host array that will be handed to work()/push()/pop() (rr_host_register)"""
    if lib().rr_host_register(_ptr(a), a.nbytes) != 0:
        raise RuntimeError(last_error())


def host_window_in_place(a: np.ndarray) -> bool:
    """whether work_into() lets the kernels read / write this host window in place (rr_host_window_in_place)"""
    return bool(lib().rr_host_window_in_place(_ptr(a), a.nbytes))


def host_unregister(a: np.ndarray) -> None:
    if lib().rr_host_unregister(_ptr(a)) != 0:
        raise RuntimeError(last_error())


class DeviceStream:
    """new_stream() (src/stream.rs:336-339) in HBM: rr_dstream.  Same window contract as ReadStream /
    WriteStream: everything readable / all free space, contiguous; `capacity_bytes` as the reference's
    DEFAULT_STREAM_SIZE."""

    def __init__(self, dtype, capacity_bytes: int = 4_096_000):
        self.dtype = np.dtype(dtype)
        self._h = lib().rr_dstream_create(self.dtype.itemsize, capacity_bytes)
        if not self._h:
            raise RuntimeError(last_error())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().rr_dstream_destroy(h)

    @property
    def capacity(self) -> int:
        return lib().rr_dstream_capacity(self._h)

    @property
    def double_mapped(self) -> bool:
        """True when the ring is one physical allocation mapped twice (no data movement, ever)"""
        return bool(lib().rr_dstream_is_double_mapped(self._h))

    def readable(self) -> int:
        return lib().rr_dstream_read_buf(self._h, None)

    def free(self, stream: int = 0) -> int:
        return lib().rr_dstream_write_buf(self._h, None, C.c_void_p(stream))

    SIDE_WRITER, SIDE_READER = 0, 1

    def close(self, side: int) -> None:
        """drop of the WriteStream / ReadStream end (rr_dstream_close): what the other end's closed() / eof() / wait() see"""
        if lib().rr_dstream_close(self._h, side) != 0:
            raise RuntimeError(last_error())

    def closed(self, side: int) -> bool:
        return bool(lib().rr_dstream_closed(self._h, side))

    def wait(self, side: int, need: int, timeout_ms: int = 100):
        """StreamWait::wait(need) of the `side` end (src/stream.rs:121-126) -> (count seen, never)"""
        never = C.c_int(0)
        n = lib().rr_dstream_wait(self._h, side, need, timeout_ms, C.byref(never))
        return n, bool(never.value)

    def push(self, x: np.ndarray, stream: int = 0) -> int:
        """fill_from_slice + produce of as much of the host array `x` as fits -> elements taken"""
        x = np.ascontiguousarray(x, self.dtype)
        n = min(len(x), self.free(stream))
        if n and (lib().rr_dstream_copy_in(self._h, 0, _ptr(x), n, C.c_void_p(stream)) != 0
                  or lib().rr_dstream_produce(self._h, n) != 0):
            raise RuntimeError(last_error())
        return n

    def produce_resident(self, stream: int = 0) -> int:
        """a device-resident source: everything free becomes readable as it stands in the ring (write_buf + produce, no copy) -> n"""
        n = self.free(stream)
        if n and lib().rr_dstream_produce(self._h, n) != 0:
            raise RuntimeError(last_error())
        return n

    def discard(self) -> int:
        """consume everything readable without copying it anywhere (NullSink, src/null_sink.rs:15-25)"""
        n = self.readable()
        if n and lib().rr_dstream_consume(self._h, n) != 0:
            raise RuntimeError(last_error())
        return n

    def pop(self, n: int = None, stream: int = 0) -> np.ndarray:
        """copy the first n readable elements (default all) to the host and consume them"""
        m = self.readable()
        n = m if n is None else min(n, m)
        out = np.empty(n, self.dtype)
        if n and (lib().rr_dstream_copy_out(self._h, 0, _ptr(out), n, C.c_void_p(stream)) != 0
                  or lib().rr_dstream_consume(self._h, n) != 0):
            raise RuntimeError(last_error())
        return out


FANOUT_TIMING, FANOUT_RCCL_ALWAYS, FANOUT_MESH, FANOUT_ID_BYTES = 1, 2, 4, 128


def fanout_unique_id() -> bytes:
    """rr_fanout_unique_id: the 128-byte group id, made on the owning rank and shipped to the others"""
    buf = C.create_string_buffer(FANOUT_ID_BYTES)
    if lib().rr_fanout_unique_id(buf) != 0:
        raise RuntimeError(last_error())
    return buf.raw


class Fanout:
    """rr_fanout (include/rustradio_amd.h): the double-buffered streaming fan-out of a shared source across GPUs, one
    process per GPU, broadcast over RCCL on a communication stream (replaces the reference's in-process Tee tree,
    src/tee.rs:10-24).  Streams are raw HIP stream handles (0 = the default stream)."""

    def __init__(self, group_id, rank: int, world: int, tile_bytes: int, src_rank: int = 0, flags: int = 0):
        self.rank, self.world, self.src, self.tile_bytes = rank, world, src_rank, tile_bytes
        self._h = lib().rr_fanout_create(group_id, rank, world, src_rank, tile_bytes, flags)
        if not self._h:
            raise RuntimeError(last_error())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:                            # (module globals are gone at interpreter shutdown)
            lib().rr_fanout_destroy(h)

    def produce_buf(self, t: int, stream: int = 0) -> int:
        p = lib().rr_fanout_produce_buf(self._h, t, C.c_void_p(stream))
        if not p:
            raise RuntimeError(last_error())
        return p

    def submit(self, t: int, stream: int = 0) -> None:
        if lib().rr_fanout_submit(self._h, t, C.c_void_p(stream)) != 0:
            raise RuntimeError(last_error())

    def acquire(self, t: int, stream: int = 0) -> int:
        p = lib().rr_fanout_acquire(self._h, t, C.c_void_p(stream))
        if not p:
            raise RuntimeError(last_error())
        return p

    def release(self, t: int, stream: int = 0) -> None:
        if lib().rr_fanout_release(self._h, t, C.c_void_p(stream)) != 0:
            raise RuntimeError(last_error())

    def stats(self):
        """-> (summed broadcast ms, broadcasts timed) since the last call (FANOUT_TIMING)"""
        ms, n = C.c_double(0), C.c_size_t(0)
        if lib().rr_fanout_stats(self._h, C.byref(ms), C.byref(n)) != 0:
            raise RuntimeError(last_error())
        return ms.value, n.value


def FirFilter(taps, deci: int = 1, translate=None, rotator: int = ROT_REPLAY) -> Block:
    """FirFilter::builder(taps).deci(deci).translate(samp_rate, freq).build(src)."""
    if np.iscomplexobj(np.asarray(taps)):
        t = np.ascontiguousarray(taps, np.complex64)
        fs, f = translate if translate is not None else (0.0, 0.0)
        h = lib().rr_fir_c32_create(_ptr(t), len(t), deci, 1 if translate is not None else 0, fs, f)
        b = Block(h, np.complex64, np.complex64)
        if translate is not None and lib().rr_fir_set_rotator_mode(b._h, rotator) != 0:
            raise ValueError(last_error())
        return b
    if translate is not None:
        raise ValueError("FirFilter asked to translate on non-Complex")  # fir.rs:401
    t = np.ascontiguousarray(taps, np.float32)
    return Block(lib().rr_fir_f32_create(_ptr(t), len(t), deci), np.float32, np.float32)


def FftFilter(taps) -> Block:
    t = np.ascontiguousarray(taps, np.complex64)
    return Block(lib().rr_fftfilter_create(_ptr(t), len(t)), np.complex64, np.complex64)


def FftFilterFloat(taps) -> Block:
    t = np.ascontiguousarray(taps, np.float32)
    return Block(lib().rr_fftfilter_float_create(_ptr(t), len(t)), np.float32, np.float32)


def fftfilter_dims(block: Block):
    """-> (reference fft_size, reference nsamples, GPU tile size)."""
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    if lib().rr_fftfilter_dims(block._h, C.byref(a), C.byref(b), C.byref(c)) != 0:
        raise ValueError(last_error())
    return a.value, b.value, c.value


def fir_uses_fft_tiles(block: Block) -> bool:
    """True when a FirFilter<Complex> runs on overlap-save FFT tiles (deci 1, more than a few taps)."""
    return lib().rr_fir_fft_tile(block._h) != 0


def RationalResampler(interp: int, deci: int, dtype=np.complex64) -> Block:
    dt = np.dtype(dtype)
    return Block(lib().rr_resampler_create(interp, deci, dt.itemsize), dt, dt)


def MultiplyConst(val, dtype=np.float32) -> Block:
    """MultiplyConst::new(src, val) (src/multiply_const.rs:6-23), Float or Complex."""
    if np.dtype(dtype) == np.complex64:
        v = complex(val)
        return Block(lib().rr_multiply_const_c32_create(v.real, v.imag), np.complex64, np.complex64)
    return Block(lib().rr_multiply_const_f32_create(float(val)), np.float32, np.float32)


def FastFM() -> Block:
    """FastFM::new(src) (src/quadrature_demod.rs:144-165)."""
    return Block(lib().rr_fastfm_create(), np.complex64, np.float32)


def FftStream(size: int) -> Block:
    """FftStream::new(src, size) (src/fft_stream.rs:40-117); any size from 2 to the stream capacity (512,000)."""
    h = lib().rr_fftstream_create(size)
    if not h:
        raise ValueError(last_error())
    return Block(h, np.complex64, np.complex64)


class Fft:
    """Fft::from_fft_size(prev, size) (src/fft.rs:19-56): the message (PDU) form of the forward FFT —
    `process(msg)` takes one array of exactly `size` Complex samples and returns its transform."""

    def __init__(self, size: int):
        if size == 0:
            raise ValueError("FFT called with size 0")                    # fft.rs:24-26
        self.size = size
        self._blk = FftStream(size)

    def process(self, msg) -> np.ndarray:
        m = np.ascontiguousarray(msg, np.complex64)
        out = np.empty(len(m), np.complex64) if len(m) else np.empty(1, np.complex64)
        if lib().rr_fft_process(self._blk._h, _ptr(m), len(m), _ptr(out)) != 0:
            raise ValueError(last_error())
        return out[:len(m)]


def RtlSdrDecode() -> Block:
    """RtlSdrDecode::new(src) (src/rtlsdr_decode.rs:9-47): u8 I/Q pairs -> Complex."""
    return Block(lib().rr_rtlsdr_decode_create(), np.uint8, np.complex64)


def QuadratureDemod(gain: float = 1.0, mode: int = ATAN2_EXACT) -> Block:
    return Block(lib().rr_quaddemod_create(gain, mode), np.complex64, np.float32)


def FmChain(taps, interp: int, deci: int, gain: float = 1.0, mode: int = ATAN2_EXACT) -> Block:
    """FftFilter(taps) -> RationalResampler(interp, deci) -> QuadratureDemod(gain) fused into one
    block / one kernel (examples/rtl_fm.rs:381-419 wiring); Complex in, f32 out."""
    t = np.ascontiguousarray(taps, np.complex64)
    return Block(lib().rr_fm_chain_create(_ptr(t), len(t), interp, deci, gain, mode), np.complex64, np.float32)


def FirFftFilter(fir_taps, fft_taps) -> Block:
    """FirFilter<Complex>(fir_taps) -> FftFilter(fft_taps) fused into one convolution with the composite taps
    (rr_fir_fftfilter_create); whole-stream output == the two blocks, incl. FftFilter's zero-history start."""
    t1 = np.ascontiguousarray(fir_taps, np.complex64)
    t2 = np.ascontiguousarray(fft_taps, np.complex64)
    return Block(lib().rr_fir_fftfilter_create(_ptr(t1), len(t1), _ptr(t2), len(t2)), np.complex64, np.complex64)


def FirFmChain(fir_taps, fft_taps, interp: int, deci: int, gain: float = 1.0, mode: int = ATAN2_EXACT) -> Block:
    """FirFilter(fir_taps) -> FftFilter(fft_taps) -> RationalResampler(interp, deci) -> QuadratureDemod(gain) as one
    kernel (rr_fir_fm_chain_create): the metric's whole chain; Complex in, f32 out."""
    t1 = np.ascontiguousarray(fir_taps, np.complex64)
    t2 = np.ascontiguousarray(fft_taps, np.complex64)
    return Block(lib().rr_fir_fm_chain_create(_ptr(t1), len(t1), _ptr(t2), len(t2), interp, deci, gain, mode),
                 np.complex64, np.float32)


def AudioChain(taps, interp: int, deci: int, scale: float = 1.0) -> Block:
    """FftFilterFloat(taps) -> RationalResampler(interp, deci) -> MultiplyConst(scale) fused into one real-valued kernel
    (rr_audio_chain_create; the audio stage of examples/rtl_fm.rs:398-418); f32 in, f32 out."""
    t = np.ascontiguousarray(taps, np.float32)
    return Block(lib().rr_audio_chain_create(_ptr(t), len(t), interp, deci, scale), np.float32, np.float32)


def FmChainU8(taps, interp: int, deci: int, gain: float = 1.0, mode: int = ATAN2_EXACT) -> Block:
    """RtlSdrDecode -> FmChain fused (examples/rtl_fm.rs:328-419): RTL-SDR bytes in, f32 out; windows,
    consumed and the WAIT_SRC need count bytes."""
    t = np.ascontiguousarray(taps, np.complex64)
    return Block(lib().rr_fm_chain_u8_create(_ptr(t), len(t), interp, deci, gain, mode), np.uint8, np.float32)


def HilbertFir(hilbert_ntaps: int, taps, deci: int = 1, translate=None, wtype: int = WIN_HAMMING, parm: float = 0.0,
               rotator: int = ROT_REPLAY) -> Block:
    """Hilbert(hilbert_ntaps, wtype) -> FirFilter<Complex>(taps, deci[, translate]) fused into one composite
    decimating FIR on the real input (examples/ax25-1200-rx.rs:238-247 wiring); f32 in, Complex out."""
    t = np.ascontiguousarray(taps, np.complex64)
    fs, f = translate if translate is not None else (0.0, 0.0)
    h = lib().rr_hilbert_fir_create(hilbert_ntaps, wtype, parm, _ptr(t), len(t), deci,
                                    1 if translate is not None else 0, fs, f)
    b = Block(h, np.float32, np.complex64)
    if translate is not None and lib().rr_fir_set_rotator_mode(b._h, rotator) != 0:
        raise ValueError(last_error())
    return b


def FmMulti(taps_per_channel, interp: int, deci: int, gain: float = 1.0, mode: int = ATAN2_EXACT) -> Block:
    """N fused FM chains on one shared input (Tee + N x FmChain); taps_per_channel = [N][ntaps].
    work() returns out with shape (N, produced); work_dev() takes N windows of out_cap elements."""
    t = np.ascontiguousarray(taps_per_channel, np.complex64)
    if t.ndim != 2:
        raise ValueError("taps_per_channel must be [nchan][ntaps]")
    h = lib().rr_fm_multi_create(_ptr(t), t.shape[0], t.shape[1], interp, deci, gain, mode)
    return Block(h, np.complex64, np.float32)


def FmMultiU8(taps_per_channel, interp: int, deci: int, gain: float = 1.0, mode: int = ATAN2_EXACT) -> Block:
    """RtlSdrDecode -> Tee -> N x FmChain fused: RTL-SDR bytes in, N f32 windows out (windows, consumed and the
    WAIT_SRC need count bytes)."""
    t = np.ascontiguousarray(taps_per_channel, np.complex64)
    if t.ndim != 2:
        raise ValueError("taps_per_channel must be [nchan][ntaps]")
    h = lib().rr_fm_multi_u8_create(_ptr(t), t.shape[0], t.shape[1], interp, deci, gain, mode)
    return Block(h, np.uint8, np.float32)


def Hilbert(ntaps: int, wtype: int = WIN_HAMMING, parm: float = 0.0) -> Block:
    return Block(lib().rr_hilbert_create(ntaps, wtype, parm), np.float32, np.complex64)
