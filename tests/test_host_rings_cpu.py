"""CPU: the C++ host mirror's stream rings and runners (rustradio_amd/host/rustradio.hpp) without any GPU block — a
graph must end by itself under Graph::run and under the thread-per-block MTGraph (tests/cpp/test_host_rings.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_rings_and_both_runners_terminate():
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_rings.bin")
    src = os.path.join(ROOT, "tests", "cpp", "test_host_rings.cpp")
    lib = os.path.join(ROOT, "rustradio_amd", "lib")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", src, "-L", lib, "-lrustradio_amd", f"-Wl,-rpath,{lib}", "-o", exe],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)       # a graph that never ends = failure
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.strip().endswith("OK")
