"""ctypes binding of the CPU ORACLE (oracle/rr_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from rustradio_amd/ (the product).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

AGAIN, WAIT_SRC, WAIT_DST, EOF, PENDING, ERR = 0, 1, 2, 3, 4, -1
WIN_HAMMING, WIN_BLACKMAN, WIN_BLACKMAN_HARRIS, WIN_HAMMING_PARM = 0, 1, 2, 3
ATAN2_EXACT, ATAN2_FAST = 0, 1


def build(force: bool = False) -> str:
    """Compile liboracle.so with oracle/Makefile (gcc, strict f32)."""
    src = os.path.join(_HERE, "rr_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rr_oracle.h")))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    # bench.py's cpu_baseline leg times a build of the same source made for its host (-O3 -march=native) and
    # names it here; tests and smoke() never set this and always check against the portable build
    path = os.environ.get("RR_ORACLE_LIB") or _LIB_PATH
    L = C.CDLL(path)
    sz, f32, vp, i32 = C.c_size_t, C.c_float, C.c_void_p, C.c_int
    L.orc_make_window.argtypes = [i32, f32, sz, vp]; L.orc_make_window.restype = i32
    L.orc_compute_ntaps.argtypes = [f32, f32, i32]; L.orc_compute_ntaps.restype = sz
    L.orc_low_pass.argtypes = [f32, f32, f32, i32, f32, vp, sz]; L.orc_low_pass.restype = sz
    L.orc_hilbert_taps.argtypes = [vp, sz, vp]; L.orc_hilbert_taps.restype = None
    L.orc_multiband.argtypes = [vp, sz, vp, sz, vp]; L.orc_multiband.restype = i32
    L.orc_fir_c32_n.argtypes = [vp, sz, sz, vp, vp, sz]; L.orc_fir_c32_n.restype = None
    L.orc_fir_f32_n.argtypes = [vp, sz, sz, vp, vp, sz]; L.orc_fir_f32_n.restype = None
    L.orc_fft.argtypes = [vp, sz, i32]; L.orc_fft.restype = None
    L.orc_fast_atan2.argtypes = [f32, f32]; L.orc_fast_atan2.restype = f32
    L.orc_fir_c32_new.argtypes = [vp, sz, sz, i32, f32, f32]; L.orc_fir_c32_new.restype = vp
    L.orc_fir_f32_new.argtypes = [vp, sz, sz]; L.orc_fir_f32_new.restype = vp
    L.orc_fftfilter_new.argtypes = [vp, sz]; L.orc_fftfilter_new.restype = vp
    L.orc_fftfilter_float_new.argtypes = [vp, sz]; L.orc_fftfilter_float_new.restype = vp
    L.orc_resampler_new.argtypes = [sz, sz, sz]; L.orc_resampler_new.restype = vp
    L.orc_quaddemod_new.argtypes = [f32, i32]; L.orc_quaddemod_new.restype = vp
    L.orc_hilbert_new.argtypes = [sz, i32, f32]; L.orc_hilbert_new.restype = vp
    L.orc_rtlsdr_decode_new.argtypes = []; L.orc_rtlsdr_decode_new.restype = vp
    L.orc_multiply_const_f32_new.argtypes = [f32]; L.orc_multiply_const_f32_new.restype = vp
    L.orc_multiply_const_c32_new.argtypes = [f32, f32]; L.orc_multiply_const_c32_new.restype = vp
    L.orc_fastfm_new.argtypes = []; L.orc_fastfm_new.restype = vp
    L.orc_fftstream_new.argtypes = [sz]; L.orc_fftstream_new.restype = vp
    L.orc_block_free.argtypes = [vp]; L.orc_block_free.restype = None
    L.orc_block_work.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.orc_block_work.restype = i32
    L.orc_block_eof.argtypes = [vp, i32]; L.orc_block_eof.restype = i32
    L.orc_block_in_elem_size.argtypes = [vp]; L.orc_block_in_elem_size.restype = sz
    L.orc_block_out_elem_size.argtypes = [vp]; L.orc_block_out_elem_size.restype = sz
    L.orc_fir_get_taps.argtypes = [vp, vp, sz]; L.orc_fir_get_taps.restype = sz
    L.orc_fir_get_rotator.argtypes = [vp, vp, vp, C.POINTER(i32)]; L.orc_fir_get_rotator.restype = None
    L.orc_fftfilter_dims.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]; L.orc_fftfilter_dims.restype = None
    L.orc_last_error.argtypes = []; L.orc_last_error.restype = C.c_char_p
    _lib = L
    return L


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---- tap designers ---------------------------------------------------------------
def make_window(wtype: int, ntaps: int, parm: float = 0.0) -> np.ndarray:
    out = np.zeros(ntaps, np.float32)
    if lib().orc_make_window(wtype, parm, ntaps, _ptr(out)) != 0:
        raise ValueError(lib().orc_last_error().decode())
    return out


def compute_ntaps(samp_rate, twidth, wtype=WIN_HAMMING) -> int:
    return lib().orc_compute_ntaps(samp_rate, twidth, wtype)


def low_pass(samp_rate, cutoff, twidth, wtype=WIN_HAMMING, parm=0.0) -> np.ndarray:
    n = lib().orc_low_pass(samp_rate, cutoff, twidth, wtype, parm, None, 0)
    if n == 0:
        raise ValueError(lib().orc_last_error().decode())
    out = np.zeros(n, np.float32)
    lib().orc_low_pass(samp_rate, cutoff, twidth, wtype, parm, _ptr(out), n)
    return out


def low_pass_complex(samp_rate, cutoff, twidth, wtype=WIN_HAMMING, parm=0.0) -> np.ndarray:
    """fir.rs:594-604: Complex::new(t, 0.0)."""
    return low_pass(samp_rate, cutoff, twidth, wtype, parm).astype(np.complex64)


def hilbert_taps(window: np.ndarray) -> np.ndarray:
    w = np.ascontiguousarray(window, np.float32)
    out = np.zeros(len(w), np.float32)
    lib().orc_hilbert_taps(_ptr(w), len(w), _ptr(out))
    return out


def fir_n(taps: np.ndarray, x: np.ndarray, deci: int, n_out: int) -> np.ndarray:
    """Fir::filter_n_inplace over a window (fir.rs:192-197)."""
    if np.iscomplexobj(x):
        t = np.ascontiguousarray(taps, np.complex64); xx = np.ascontiguousarray(x, np.complex64)
        out = np.zeros(n_out, np.complex64)
        lib().orc_fir_c32_n(_ptr(t), len(t), deci, _ptr(xx), _ptr(out), n_out)
    else:
        t = np.ascontiguousarray(taps, np.float32); xx = np.ascontiguousarray(x, np.float32)
        out = np.zeros(n_out, np.float32)
        lib().orc_fir_f32_n(_ptr(t), len(t), deci, _ptr(xx), _ptr(out), n_out)
    return out


def fft(x: np.ndarray, inverse: bool = False) -> np.ndarray:
    b = np.ascontiguousarray(x, np.complex64).copy()
    lib().orc_fft(_ptr(b), len(b), 1 if inverse else 0)
    return b


def fast_atan2(y: float, x: float) -> float:
    return lib().orc_fast_atan2(y, x)


# ---- streaming blocks -------------------------------------------------------------
class OracleBlock:
    """One reference block; `work()` restates Block::work() over explicit windows."""

    def __init__(self, handle, in_dtype, out_dtype, name):
        if not handle:
            raise ValueError(lib().orc_last_error().decode())
        self._h = handle
        self.in_dtype = np.dtype(in_dtype)
        self.out_dtype = np.dtype(out_dtype)
        self.name = name

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.orc_block_free(h)

    def work(self, inp: np.ndarray, out_cap: int):
        """-> (status, consumed, produced, need, out[:produced])"""
        inp = np.ascontiguousarray(inp, self.in_dtype)
        out = np.zeros(max(out_cap, 1), self.out_dtype)
        c, p, n = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        st = lib().orc_block_work(self._h, _ptr(inp), len(inp), _ptr(out), out_cap,
                                  C.byref(c), C.byref(p), C.byref(n))
        if st == ERR:
            raise RuntimeError(lib().orc_last_error().decode())
        return st, c.value, p.value, n.value, out[:p.value]

    def eof(self, src_eof: bool) -> bool:
        return bool(lib().orc_block_eof(self._h, int(src_eof)))


def FirFilter(taps, deci: int = 1, translate=None) -> OracleBlock:
    """FirFilter::builder(taps).deci(d).translate(fs, f) (fir.rs:303-386,476-486)."""
    if np.iscomplexobj(np.asarray(taps)):
        t = np.ascontiguousarray(taps, np.complex64)
        fs, f = translate if translate is not None else (0.0, 0.0)
        h = lib().orc_fir_c32_new(_ptr(t), len(t), deci, 1 if translate is not None else 0, fs, f)
        return OracleBlock(h, np.complex64, np.complex64, "FirFilter<Complex>")
    t = np.ascontiguousarray(taps, np.float32)
    if translate is not None:
        raise ValueError("translate only on Complex")
    h = lib().orc_fir_f32_new(_ptr(t), len(t), deci)
    return OracleBlock(h, np.float32, np.float32, "FirFilter<Float>")


def fir_translated_taps(block: OracleBlock) -> np.ndarray:
    n = lib().orc_fir_get_taps(block._h, None, 0)
    out = np.zeros(n, np.complex64)
    lib().orc_fir_get_taps(block._h, _ptr(out), n)
    return out


def fir_rotator(block: OracleBlock):
    ph = np.zeros(1, np.complex64); st = np.zeros(1, np.complex64); en = C.c_int(0)
    lib().orc_fir_get_rotator(block._h, _ptr(ph), _ptr(st), C.byref(en))
    return ph[0], st[0], bool(en.value)


def FftFilter(taps) -> OracleBlock:
    t = np.ascontiguousarray(taps, np.complex64)
    return OracleBlock(lib().orc_fftfilter_new(_ptr(t), len(t)), np.complex64, np.complex64, "FftFilter")


def multiband(bands, window):
    """fir::multiband(bands, taps, window) (fir.rs:552-590) -> complex64 taps, or None"""
    b = np.ascontiguousarray(bands, np.float32).reshape(-1, 2)
    w = np.ascontiguousarray(window, np.float32)
    out = np.zeros(len(w), np.complex64)
    return out if lib().orc_multiband(_ptr(b), len(b), _ptr(w), len(w), _ptr(out)) == 0 else None


def FftFilterFloat(taps) -> OracleBlock:
    t = np.ascontiguousarray(taps, np.float32)
    return OracleBlock(lib().orc_fftfilter_float_new(_ptr(t), len(t)), np.float32, np.float32, "FftFilterFloat")


def fftfilter_dims(block: OracleBlock):
    a, b = C.c_size_t(0), C.c_size_t(0)
    lib().orc_fftfilter_dims(block._h, C.byref(a), C.byref(b))
    return a.value, b.value


def RationalResampler(interp: int, deci: int, dtype=np.complex64) -> OracleBlock:
    dt = np.dtype(dtype)
    return OracleBlock(lib().orc_resampler_new(interp, deci, dt.itemsize), dt, dt, "RationalResampler")


def QuadratureDemod(gain: float = 1.0, mode: int = ATAN2_EXACT) -> OracleBlock:
    return OracleBlock(lib().orc_quaddemod_new(gain, mode), np.complex64, np.float32, "QuadratureDemod")


def MultiplyConst(val, dtype=np.float32) -> OracleBlock:
    if np.dtype(dtype) == np.complex64:
        v = complex(val)
        return OracleBlock(lib().orc_multiply_const_c32_new(v.real, v.imag), np.complex64, np.complex64, "MultiplyConst")
    return OracleBlock(lib().orc_multiply_const_f32_new(float(val)), np.float32, np.float32, "MultiplyConst")


def FastFM() -> OracleBlock:
    return OracleBlock(lib().orc_fastfm_new(), np.complex64, np.float32, "FastFM")


def FftStream(size: int) -> OracleBlock:
    h = lib().orc_fftstream_new(size)
    if not h:
        raise ValueError(lib().orc_last_error().decode())
    return OracleBlock(h, np.complex64, np.complex64, "FftStream")


def RtlSdrDecode() -> OracleBlock:
    return OracleBlock(lib().orc_rtlsdr_decode_new(), np.uint8, np.complex64, "RtlSdrDecode")


def Hilbert(ntaps: int, wtype: int = WIN_HAMMING, parm: float = 0.0) -> OracleBlock:
    return OracleBlock(lib().orc_hilbert_new(ntaps, wtype, parm), np.float32, np.complex64, "Hilbert")
