"""GPU: the reference's unit tests written against the C++ host mirror of the Block/Stream
API (rustradio_amd/host/rustradio.hpp -> C ABI -> HIP kernels): tests/cpp/test_host_api.cpp."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_api_reference_tests():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_api.bin")
    src = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
    lib = os.path.join(ROOT, "rustradio_amd", "lib")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-L", lib, "-lrustradio_amd",
                    f"-Wl,-rpath,{lib}", "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.strip().endswith("OK")
