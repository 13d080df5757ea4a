// test_resident_graph.cpp — the Rust shim's device-resident graph design (rust/src/lib.rs: GpuUpload -> GpuResident ... ->
// GpuDownload over new_gpu_stream() handles), compiled from its C++ twin (rustradio_amd/host/resident.hpp) and driven TO
// TERMINATION by both of the reference's runners: Graph::run (src/graph.rs:126-147) and MTGraph, one thread per block
// (src/mtgraph.rs:98-116).  tests/test_gpu_resident_twin.py runs it under a timeout (a graph that never ends = failure)
// and compares the sink with the oracle chain's whole-stream output.  Needs a GPU.
//
//   test_resident_graph <graph|mt> <in.c32> <taps.c32> <out.f32> <ring_bytes> <interp> <deci> <fused 0|1> [out_ring_bytes]
//
//   VectorSource<Complex> -> GpuUpload -> [FftFilter -> RationalResampler -> QuadratureDemod | fused FmChain] -> GpuDownload
//   -> VectorSink<Float>; every HBM ring holds `ring_bytes`, the last one `out_ring_bytes` (default = ring_bytes).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../rustradio_amd/host/resident.hpp"

using namespace rustradio;

template <class T> static std::vector<T> read_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)bytes / sizeof(T));
    if (!v.empty() && fread(v.data(), sizeof(T), v.size(), f) != v.size()) { perror("fread"); exit(2); }
    fclose(f);
    return v;
}

template <class G> static std::vector<Float> run(G& g, const std::vector<Complex>& x, const std::vector<Complex>& taps, size_t ring,
                                                 size_t out_ring, size_t interp, size_t deci, bool fused) {
    const rr_c32* t = reinterpret_cast<const rr_c32*>(taps.data());
    auto [src, s0] = VectorSource<Complex>::new_(x);
    auto [up, d0] = GpuUpload<Complex>::new_(std::move(s0), ring);
    g.add(std::move(src));
    g.add(std::move(up));
    GpuReadStream<Float> last;
    if (fused) {
        auto [b, d1] = GpuResident<Complex, Float>::new_(rr_fm_chain_create(t, taps.size(), interp, deci, 1.0f, RR_ATAN2_EXACT),
                                                         "FmChain", std::move(d0), out_ring);
        g.add(std::move(b));
        last = std::move(d1);
    } else {
        auto [f, d1] = GpuResident<Complex, Complex>::new_(rr_fftfilter_create(t, taps.size()), "FftFilter", std::move(d0), ring);
        auto [r, d2] = GpuResident<Complex, Complex>::new_(rr_resampler_create(interp, deci, sizeof(Complex)), "RationalResampler",
                                                           std::move(d1), ring);
        auto [q, d3] = GpuResident<Complex, Float>::new_(rr_quaddemod_create(1.0f, RR_ATAN2_EXACT), "QuadratureDemod", std::move(d2), out_ring);
        g.add(std::move(f));
        g.add(std::move(r));
        g.add(std::move(q));
        last = std::move(d3);
    }
    auto [down, h] = GpuDownload<Float>::new_(std::move(last));
    auto sink = std::make_unique<VectorSink<Float>>(std::move(h));
    auto hook = sink->hook();
    g.add(std::move(down));
    g.add(std::move(sink));
    g.run();                                   // must return by itself
    return *hook;
}

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage: %s graph|mt in.c32 taps.c32 out.f32 ring_bytes interp deci fused [out_ring_bytes]\n", argv[0]); return 2; }
    const std::string runner = argv[1];
    const auto x = read_file<Complex>(argv[2]);
    const auto taps = read_file<Complex>(argv[3]);
    const size_t ring = strtoull(argv[5], nullptr, 10), interp = strtoull(argv[6], nullptr, 10), deci = strtoull(argv[7], nullptr, 10);
    const bool fused = atoi(argv[8]) != 0;
    const size_t out_ring = argc > 9 ? strtoull(argv[9], nullptr, 10) : ring;
    std::vector<Float> y;
    try {
        if (runner == "mt") { MTGraph g; y = run(g, x, taps, ring, out_ring, interp, deci, fused); }
        else { Graph g; y = run(g, x, taps, ring, out_ring, interp, deci, fused); }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    FILE* f = fopen(argv[4], "wb");
    if (!f) { perror(argv[4]); return 2; }
    if (!y.empty()) fwrite(y.data(), sizeof(Float), y.size(), f);
    fclose(f);
    printf("OK %zu samples\n", y.size());
    return 0;
}
