// graph_rate.cpp — end-to-end rate of an examples/rtl_fm.rs-style Graph through the C++ mirror, host memory in,
// NullSink out: rings in host memory (every work() is a PCIe round trip) vs rings in HBM, reference-sized
// (4,096,000 B) vs HBM-sized rings.  Build: g++ -O2 -std=c++17 tests/cpp/graph_rate.cpp -L rustradio_amd/lib -lrustradio_amd
#include <chrono>
#include <cmath>
#include <cstdio>

#include "../../rustradio_amd/host/rustradio.hpp"

using namespace rustradio;
using window::WindowType;

template <class G = Graph>
static double run(Memory mem, size_t ring_bytes, const std::vector<Complex>& x, uint64_t repeats, bool fused_free_blocks) {
    (void)fused_free_blocks;
    default_memory() = mem;
    default_stream_size() = ring_bytes;
    auto taps = fir::low_pass_complex(2.4e6f, 100e3f, 12.5e3f, WindowType::Hamming());
    auto [src, s0] = VectorSource<Complex>::new_(x, Repeat::finite(repeats));
    auto [fft, s1] = FftFilter::new_(std::move(s0), taps);
    auto [rs, s2] = RationalResampler<Complex>::new_(std::move(s1), 1, 6);
    auto [qd, s3] = QuadratureDemod::new_(std::move(s2), 1.0f);
    auto sink = std::make_unique<NullSink<Float>>(std::move(s3));
    G g;
    g.add(std::move(src)); g.add(std::move(fft)); g.add(std::move(rs)); g.add(std::move(qd)); g.add(std::move(sink));
    const auto t0 = std::chrono::steady_clock::now();
    g.run();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    default_memory() = Memory::Host;
    default_stream_size() = DEFAULT_STREAM_SIZE;
    return (double)x.size() * (double)repeats / dt / 1e6;
}

int main() {
    std::vector<Complex> x(6'000'000);
    uint32_t lcg = 1;
    for (auto& v : x) { lcg = lcg * 1664525u + 1013904223u; v = Complex((float)(lcg >> 8) / 8388608.0f - 1.0f, (float)((lcg * 31u) >> 8) / 8388608.0f - 1.0f); }
    run(Memory::Device, DEFAULT_STREAM_SIZE, x, 1, false);                       // warm-up (module load, plans)
    printf("host rings   4,096,000 B : %8.1f Msamples/s\n", run(Memory::Host, DEFAULT_STREAM_SIZE, x, 4, false));
    printf("HBM rings    4,096,000 B : %8.1f Msamples/s\n", run(Memory::Device, DEFAULT_STREAM_SIZE, x, 8, false));
    printf("HBM rings   64,000,000 B : %8.1f Msamples/s\n", run(Memory::Device, 64'000'000, x, 16, false));
    printf("HBM rings  512,000,000 B : %8.1f Msamples/s\n", run(Memory::Device, 512'000'000, x, 32, false));
    // the same graphs under the thread-per-block runner (mtgraph.rs): source copy, PCIe and kernels of different windows overlap
    printf("host rings   4,096,000 B, MTGraph : %8.1f Msamples/s\n", run<MTGraph>(Memory::Host, DEFAULT_STREAM_SIZE, x, 4, false));
    printf("HBM rings    4,096,000 B, MTGraph : %8.1f Msamples/s\n", run<MTGraph>(Memory::Device, DEFAULT_STREAM_SIZE, x, 8, false));
    printf("HBM rings   64,000,000 B, MTGraph : %8.1f Msamples/s\n", run<MTGraph>(Memory::Device, 64'000'000, x, 16, false));
    return 0;
}
