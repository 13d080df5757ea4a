"""CPU-only: the C-ABI library builds, loads, exports every symbol include/rustradio_amd.h
declares, fails loudly (no CPU fallback) when asked to compute without a GPU, and its
setup-time tap designers agree with the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

import rustradio_amd as rr
from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rustradio_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(rr.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/rustradio_amd.h but not exported"
    from rustradio_amd._lib import SYMBOLS
    assert sorted(SYMBOLS) == syms
    assert rr.lib().rr_abi_version() == 3


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ValueError, match="no usable HIP device"):
        rr.FftFilter(np.ones(4, np.complex64))
    with pytest.raises(ValueError, match="no usable HIP device"):
        rr.QuadratureDemod(1.0)


def test_product_never_references_oracle():
    pkg = os.path.join(ROOT, "rustradio_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "rr_oracle" not in txt, fn


@pytest.mark.parametrize("args", [(10000.0, 1000.0, 1000.0), (10e6, 1e6, 190e3), (10e6, 1e6, 60e3),
                                  (2.4e6, 100e3, 12.5e3), (100e6, 5e6, 943e3), (1.024e6, 100e3, 1e3),
                                  (48000.0, 4000.0, 1000.0), (8000.0, 1000.0, 100.0)])
def test_low_pass_matches_oracle(args):
    for w in (rr.WIN_HAMMING, rr.WIN_BLACKMAN, rr.WIN_BLACKMAN_HARRIS):
        a = rr.low_pass(*args, w)
        b = orc.low_pass(*args, w)
        assert len(a) == len(b) == rr.compute_ntaps(args[0], args[2], w)
        assert np.max(np.abs(a - b)) <= 1e-7 * max(1.0, np.max(np.abs(b)))
    assert np.array_equal(rr.low_pass_complex(*args).real, rr.low_pass(*args))


def test_windows_and_hilbert_taps_match_oracle():
    for n in (1, 2, 3, 65, 128, 401):
        for w, parm in ((rr.WIN_HAMMING, 0.0), (rr.WIN_BLACKMAN, 0.0), (rr.WIN_BLACKMAN_HARRIS, 0.0),
                        (rr.WIN_HAMMING_PARM, 0.54)):
            assert np.allclose(rr.make_window(w, n, parm), orc.make_window(w, n, parm), atol=1e-7)
    for n in (3, 65, 129):
        win = rr.make_window(rr.WIN_HAMMING, n)
        assert np.allclose(rr.hilbert_taps(win), orc.hilbert_taps(win), atol=1e-7)
    with pytest.raises(ValueError):
        rr.make_window(9, 4)
    with pytest.raises(ValueError):
        rr.low_pass(0.0, 1.0, 1.0)


def test_cpu_emulation_of_fft_tile():
    """The per-thread pass functions of the HIP FFT kernel, run thread-by-thread on the CPU."""
    import subprocess, tempfile
    exe = os.path.join(tempfile.gettempdir(), "rr_emu_fft")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "rustradio_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "emu_fft.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_build_opts_struct_matches_the_header():
    """rr_build_opts (include/rustradio_amd.h) and its ctypes mirror: the same int fields in the same order, 16 ints in all —
    a field added to one side only would shift every override behind it"""
    from rustradio_amd._lib import BuildOpts
    src = open(os.path.join(ROOT, "include", "rustradio_amd.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} rr_build_opts;", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\bint\s+([a-z0-9_]+)(?:\[(\d+)\])?\s*;", body)
    names = [n for n, _ in fields]
    total = sum(int(k) if k else 1 for _, k in fields)
    assert names == [n for n, _ in BuildOpts._fields_] and names[-1] == "reserved"
    assert total == 16 and ctypes.sizeof(BuildOpts) == 16 * ctypes.sizeof(ctypes.c_int)
