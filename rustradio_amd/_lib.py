"""Loader of librustradio_amd.so (the C ABI in include/rustradio_amd.h).

The library is the product; this module only binds it.  Import fails loudly when
the shared object is missing — there is no Python or CPU fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# RR_LIB_PATH: measurement tools load a differently-built library (tools/*.sh: ablation / timing builds) without ever
# overwriting the product .so
LIB_PATH = os.environ.get("RR_LIB_PATH") or os.path.join(_HERE, "lib", "librustradio_amd.so")

# every symbol include/rustradio_amd.h declares (checked by tests/test_abi_symbols.py)
SYMBOLS = [
    "rr_abi_version", "rr_last_error", "rr_device_count", "rr_set_device", "rr_next_create_options",
    "rr_max_attenuation", "rr_make_window", "rr_compute_ntaps", "rr_low_pass", "rr_low_pass_complex",
    "rr_hilbert_taps", "rr_multiband",
    "rr_fir_c32_create", "rr_fir_f32_create", "rr_fftfilter_create", "rr_fftfilter_float_create",
    "rr_resampler_create", "rr_quaddemod_create", "rr_rtlsdr_decode_create", "rr_fftstream_create", "rr_fft_process", "rr_multiply_const_f32_create", "rr_multiply_const_c32_create", "rr_fastfm_create", "rr_hilbert_create", "rr_fm_chain_create", "rr_fm_chain_u8_create", "rr_fir_fftfilter_create", "rr_fir_fm_chain_create", "rr_audio_chain_create", "rr_hilbert_fir_create", "rr_fm_multi_create", "rr_fm_multi_u8_create", "rr_block_out_windows", "rr_block_destroy",
    "rr_block_work", "rr_block_work_dev", "rr_block_eof", "rr_block_name", "rr_block_tag_rule", "rr_block_in_elem_size",
    "rr_block_out_elem_size", "rr_block_sync", "rr_fftfilter_dims", "rr_fir_fft_tile", "rr_fir_set_rotator_mode",
    "rr_block_set_profiling", "rr_block_profile", "rr_debug_fft_stamps", "rr_debug_kernel_launches",
    "rr_host_register", "rr_host_unregister", "rr_host_window_in_place",
    "rr_dstream_create", "rr_dstream_destroy", "rr_dstream_capacity", "rr_dstream_is_double_mapped", "rr_dstream_read_buf", "rr_dstream_write_buf",
    "rr_dstream_consume", "rr_dstream_produce", "rr_dstream_close", "rr_dstream_closed", "rr_dstream_wait", "rr_dstream_id", "rr_dstream_copy_in", "rr_dstream_copy_out", "rr_dstream_copy", "rr_block_work_streams",
    "rr_fanout_unique_id", "rr_fanout_create", "rr_fanout_destroy", "rr_fanout_produce_buf", "rr_fanout_submit",
    "rr_fanout_acquire", "rr_fanout_release", "rr_fanout_stats",
]

_lib = None


class BuildOpts(C.Structure):
    """rr_build_opts (include/rustradio_amd.h): path-selection overrides for the next create call"""
    _fields_ = [(n, C.c_int) for n in ("fir_path", "fir_prune", "fir_half", "fir_cfg_plus1", "fft_log2f", "fft_no_split",
                                        "fftfloat_complex", "fm_full", "fm_poly", "dstream_no_vmm", "fir_poly", "fft_nonfinite_tiles", "host_in_staged")] + [("reserved", C.c_int * 3)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rustradio_amd/csrc` (hipcc --offload-arch=gfx950). rustradio_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64, and whichever copy is loaded
    # first serves both (same SONAME).  If this library pulled in /opt/rocm's copy before torch was imported,
    # torch would later run on a runtime it was not built against (observed: torch.cuda.is_available() False,
    # or "no usable HIP device" here).  So when torch is installed, let it load its runtime first.
    if "torch" not in sys.modules and not os.environ.get("RR_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    sz, f32, vp, i32 = C.c_size_t, C.c_float, C.c_void_p, C.c_int
    psz = C.POINTER(sz)
    L.rr_abi_version.restype = i32
    L.rr_last_error.restype = C.c_char_p
    L.rr_device_count.restype = i32
    L.rr_set_device.argtypes = [i32]; L.rr_set_device.restype = i32
    L.rr_next_create_options.argtypes = [vp]; L.rr_next_create_options.restype = i32
    L.rr_max_attenuation.argtypes = [i32]; L.rr_max_attenuation.restype = f32
    L.rr_make_window.argtypes = [i32, f32, sz, vp]; L.rr_make_window.restype = i32
    L.rr_compute_ntaps.argtypes = [f32, f32, i32]; L.rr_compute_ntaps.restype = sz
    L.rr_low_pass.argtypes = [f32, f32, f32, i32, f32, vp, sz]; L.rr_low_pass.restype = sz
    L.rr_low_pass_complex.argtypes = [f32, f32, f32, i32, f32, vp, sz]; L.rr_low_pass_complex.restype = sz
    L.rr_hilbert_taps.argtypes = [vp, sz, vp]; L.rr_hilbert_taps.restype = i32
    L.rr_multiband.argtypes = [vp, sz, vp, sz, vp]; L.rr_multiband.restype = i32
    L.rr_fir_c32_create.argtypes = [vp, sz, sz, i32, f32, f32]; L.rr_fir_c32_create.restype = vp
    L.rr_fir_f32_create.argtypes = [vp, sz, sz]; L.rr_fir_f32_create.restype = vp
    L.rr_fftfilter_create.argtypes = [vp, sz]; L.rr_fftfilter_create.restype = vp
    L.rr_fftfilter_float_create.argtypes = [vp, sz]; L.rr_fftfilter_float_create.restype = vp
    L.rr_resampler_create.argtypes = [sz, sz, sz]; L.rr_resampler_create.restype = vp
    L.rr_quaddemod_create.argtypes = [f32, i32]; L.rr_quaddemod_create.restype = vp
    L.rr_rtlsdr_decode_create.argtypes = []; L.rr_rtlsdr_decode_create.restype = vp
    L.rr_multiply_const_f32_create.argtypes = [f32]; L.rr_multiply_const_f32_create.restype = vp
    L.rr_multiply_const_c32_create.argtypes = [f32, f32]; L.rr_multiply_const_c32_create.restype = vp
    L.rr_fastfm_create.argtypes = []; L.rr_fastfm_create.restype = vp
    L.rr_fftstream_create.argtypes = [sz]; L.rr_fftstream_create.restype = vp
    L.rr_fft_process.argtypes = [vp, vp, sz, vp]; L.rr_fft_process.restype = i32
    L.rr_hilbert_create.argtypes = [sz, i32, f32]; L.rr_hilbert_create.restype = vp
    L.rr_fm_chain_create.argtypes = [vp, sz, sz, sz, f32, i32]; L.rr_fm_chain_create.restype = vp
    L.rr_fir_fftfilter_create.argtypes = [vp, sz, vp, sz]; L.rr_fir_fftfilter_create.restype = vp
    L.rr_fir_fm_chain_create.argtypes = [vp, sz, vp, sz, sz, sz, f32, i32]; L.rr_fir_fm_chain_create.restype = vp
    L.rr_audio_chain_create.argtypes = [vp, sz, sz, sz, f32]; L.rr_audio_chain_create.restype = vp
    L.rr_fm_chain_u8_create.argtypes = [vp, sz, sz, sz, f32, i32]; L.rr_fm_chain_u8_create.restype = vp
    L.rr_hilbert_fir_create.argtypes = [sz, i32, f32, vp, sz, sz, i32, f32, f32]; L.rr_hilbert_fir_create.restype = vp
    L.rr_fm_multi_create.argtypes = [vp, sz, sz, sz, sz, f32, i32]; L.rr_fm_multi_create.restype = vp
    L.rr_fm_multi_u8_create.argtypes = [vp, sz, sz, sz, sz, f32, i32]; L.rr_fm_multi_u8_create.restype = vp
    L.rr_block_out_windows.argtypes = [vp]; L.rr_block_out_windows.restype = sz
    L.rr_block_destroy.argtypes = [vp]; L.rr_block_destroy.restype = None
    L.rr_block_work.argtypes = [vp, vp, sz, vp, sz, psz, psz, psz]; L.rr_block_work.restype = i32
    L.rr_block_work_dev.argtypes = [vp, vp, sz, vp, sz, psz, psz, psz, vp]; L.rr_block_work_dev.restype = i32
    L.rr_block_eof.argtypes = [vp, i32]; L.rr_block_eof.restype = i32
    L.rr_block_name.argtypes = [vp]; L.rr_block_name.restype = C.c_char_p
    L.rr_block_tag_rule.argtypes = [vp, psz]; L.rr_block_tag_rule.restype = i32
    L.rr_block_in_elem_size.argtypes = [vp]; L.rr_block_in_elem_size.restype = sz
    L.rr_block_out_elem_size.argtypes = [vp]; L.rr_block_out_elem_size.restype = sz
    L.rr_block_sync.argtypes = [vp]; L.rr_block_sync.restype = i32
    L.rr_fftfilter_dims.argtypes = [vp, psz, psz, psz]; L.rr_fftfilter_dims.restype = i32
    L.rr_fir_fft_tile.argtypes = [vp]; L.rr_fir_fft_tile.restype = sz
    L.rr_fir_set_rotator_mode.argtypes = [vp, i32]; L.rr_fir_set_rotator_mode.restype = i32
    L.rr_debug_fft_stamps.argtypes = [vp]; L.rr_debug_fft_stamps.restype = i32
    L.rr_debug_kernel_launches.argtypes = []; L.rr_debug_kernel_launches.restype = C.c_ulonglong
    pvp = C.POINTER(vp)
    L.rr_host_register.argtypes = [vp, sz]; L.rr_host_register.restype = i32
    L.rr_host_window_in_place.argtypes = [vp, sz]; L.rr_host_window_in_place.restype = i32
    L.rr_host_unregister.argtypes = [vp]; L.rr_host_unregister.restype = i32
    L.rr_dstream_create.argtypes = [sz, sz]; L.rr_dstream_create.restype = vp
    L.rr_dstream_destroy.argtypes = [vp]; L.rr_dstream_destroy.restype = None
    L.rr_dstream_capacity.argtypes = [vp]; L.rr_dstream_capacity.restype = sz
    L.rr_dstream_is_double_mapped.argtypes = [vp]; L.rr_dstream_is_double_mapped.restype = i32
    L.rr_dstream_read_buf.argtypes = [vp, pvp]; L.rr_dstream_read_buf.restype = sz
    L.rr_dstream_write_buf.argtypes = [vp, pvp, vp]; L.rr_dstream_write_buf.restype = sz
    L.rr_dstream_consume.argtypes = [vp, sz]; L.rr_dstream_consume.restype = i32
    L.rr_dstream_produce.argtypes = [vp, sz]; L.rr_dstream_produce.restype = i32
    L.rr_dstream_close.argtypes = [vp, i32]; L.rr_dstream_close.restype = i32
    L.rr_dstream_closed.argtypes = [vp, i32]; L.rr_dstream_closed.restype = i32
    L.rr_dstream_wait.argtypes = [vp, i32, sz, C.c_uint, C.POINTER(C.c_int)]; L.rr_dstream_wait.restype = sz
    L.rr_dstream_id.argtypes = [vp]; L.rr_dstream_id.restype = sz
    L.rr_dstream_copy_in.argtypes = [vp, sz, vp, sz, vp]; L.rr_dstream_copy_in.restype = i32
    L.rr_dstream_copy_out.argtypes = [vp, sz, vp, sz, vp]; L.rr_dstream_copy_out.restype = i32
    L.rr_dstream_copy.argtypes = [vp, sz, vp, sz, sz, vp]; L.rr_dstream_copy.restype = i32
    L.rr_block_work_streams.argtypes = [vp, vp, vp, psz, psz, psz, vp]; L.rr_block_work_streams.restype = i32
    u64 = C.c_ulonglong
    L.rr_fanout_unique_id.argtypes = [vp]; L.rr_fanout_unique_id.restype = i32
    L.rr_fanout_create.argtypes = [vp, i32, i32, i32, sz, i32]; L.rr_fanout_create.restype = vp
    L.rr_fanout_destroy.argtypes = [vp]; L.rr_fanout_destroy.restype = None
    L.rr_fanout_produce_buf.argtypes = [vp, u64, vp]; L.rr_fanout_produce_buf.restype = vp
    L.rr_fanout_submit.argtypes = [vp, u64, vp]; L.rr_fanout_submit.restype = i32
    L.rr_fanout_acquire.argtypes = [vp, u64, vp]; L.rr_fanout_acquire.restype = vp
    L.rr_fanout_release.argtypes = [vp, u64, vp]; L.rr_fanout_release.restype = i32
    L.rr_fanout_stats.argtypes = [vp, C.POINTER(C.c_double), psz]; L.rr_fanout_stats.restype = i32
    L.rr_block_set_profiling.argtypes = [vp, i32]; L.rr_block_set_profiling.restype = i32
    L.rr_block_profile.argtypes = [vp, C.POINTER(C.c_double), psz, i32]; L.rr_block_profile.restype = i32
    _lib = L
    return L


def last_error() -> str:
    return lib().rr_last_error().decode()
