// test_resident_graph.cpp — the Rust shim's device-resident graph design (rust/src/lib.rs: GpuUpload -> GpuResident ... ->
// GpuDownload over new_gpu_stream() handles), compiled from its C++ twin (rustradio_amd/host/resident.hpp) and driven TO
// TERMINATION by both of the reference's runners: Graph::run (src/graph.rs:126-147) and MTGraph, one thread per block
// (src/mtgraph.rs:98-116).  tests/test_gpu_resident_twin.py runs it under a timeout (a graph that never ends = failure)
// and compares the sink with the oracle chain's whole-stream output.  Needs a GPU.
//
//   test_resident_graph <graph|mt> <in.c32> <taps.c32> <out.f32> <ring_bytes> <interp> <deci> <fused 0|1> [out_ring_bytes]
//
//   test_resident_graph <graph|mt> tags        (tag forwarding across the device-resident boundary, see tag_tests)
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

// ---- tags across the device-resident boundary ------------------------------------------------------------
//   test_resident_graph <graph|mt> tags
// The reference's own tag tests, driven through GpuUpload -> GpuResident -> GpuDownload, and every forwarding block kind
// against the host-window block of the same kind (whose tag code is the reference's, window-relative).
static int g_fail = 0;
#define EXPECT(c, ...) do { if (!(c)) { g_fail++; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static std::string show(const std::vector<Tag>& t) {
    std::string s;
    for (auto& x : t) {
        s += "(" + std::to_string(x.pos()) + "," + x.key() + ",";
        if (auto* b = std::get_if<bool>(&x.val())) s += *b ? "true" : "false";
        else if (auto* u = std::get_if<uint64_t>(&x.val())) s += std::to_string(*u);
        else s += "?";
        s += ") ";
    }
    return s;
}

template <class T> struct Sunk { std::vector<T> data; std::vector<Tag> tags; };

// source -> [make(stream) ...] -> sink on one runner; `make` adds its blocks to g and returns the last host ReadStream
template <class G, class In, class Out, class Make> static Sunk<Out> drive(const std::vector<In>& x, uint64_t repeats, Make&& make) {
    G g;
    auto [src, s0] = VectorSource<In>::new_(x, Repeat::finite(repeats));
    g.add(std::move(src));
    ReadStream<Out> last = make(g, std::move(s0));
    auto sink = std::make_unique<VectorSink<Out>>(std::move(last));
    auto hook = sink->hook();
    auto thook = sink->tag_hook();
    g.add(std::move(sink));
    g.run();
    return {*hook, *thook};
}
// one GPU block between an upload and a download edge
template <class G, class In, class Out, class Create> static Sunk<Out> resident1(const std::vector<In>& x, uint64_t repeats, size_t ring,
                                                                                 const char* name, Create&& create) {
    return drive<G, In, Out>(x, repeats, [&](G& g, ReadStream<In> s0) {
        auto [up, d0] = GpuUpload<In>::new_(std::move(s0), ring);
        auto [b, d1] = GpuResident<In, Out>::new_(create(), name, std::move(d0), ring);
        auto [down, h] = GpuDownload<Out>::new_(std::move(d1));
        g.add(std::move(up));
        g.add(std::move(b));
        g.add(std::move(down));
        return std::move(h);
    });
}

template <class G> static void tag_tests() {
    const Tag S0(0, "VectorSource::start", true), R0(0, "VectorSource::repeat", (uint64_t)0), F0(0, "VectorSource::first", true);
    // fft_filter.rs:551-574 tag_propagation: 1024 zeros twice through FftFilter([0]); tags at 0 and 1024, 2048 samples out
    {
        const std::vector<Complex> x(1024), taps(1);
        auto r = resident1<G, Complex, Complex>(x, 2, 1 << 20, "FftFilter", [&] { return rr_fftfilter_create(c32(taps), 1); });
        const std::vector<Tag> want = {S0, R0, F0, Tag(1024, "VectorSource::start", true), Tag(1024, "VectorSource::repeat", (uint64_t)1)};
        EXPECT(r.tags == want, "tag_propagation: got %s", show(r.tags).c_str());
        EXPECT(r.data.size() == 2048, "tag_propagation: %zu samples", r.data.size());
    }
    // fir.rs:691-741 test_identity: 6 samples twice, one unit tap, deci 1..18: tags at 0 and 6 / deci, samples = every deci-th
    {
        const std::vector<Complex> in = {{1, 0}, {2, 0}, {3, 0.2f}, {4.1f, 0}, {5, 0}, {6, 0.2f}};
        const std::vector<Complex> taps = {{1, 0}};
        for (size_t deci = 1; deci <= 3 * in.size(); deci++) {
            auto r = resident1<G, Complex, Complex>(in, 2, 4096, "FirFilter", [&] { return rr_fir_c32_create(c32(taps), 1, deci, 0, 0, 0); });
            const size_t max = 2 * in.size() / deci;
            EXPECT(r.data.size() == max, "test_identity deci %zu: %zu samples, want %zu", deci, r.data.size(), max);
            for (size_t i = 0; i < std::min(max, r.data.size()); i++) {
                const Complex want = in[(i * deci) % in.size()];
                EXPECT(std::abs(r.data[i] - want) < 1e-6f, "test_identity deci %zu sample %zu", deci, i);
            }
            std::vector<Tag> want;
            if (max) want = {S0, R0, F0, Tag(6 / deci, "VectorSource::start", true), Tag(6 / deci, "VectorSource::repeat", (uint64_t)1)};
            EXPECT(r.tags == want, "test_identity deci %zu: got %s", deci, show(r.tags).c_str());
        }
    }
    // every forwarding kind, small rings (many work() calls, tags waiting inside FftFilter across calls), against the
    // host-window block of the same kind; 3 repeats of a length that is no multiple of anything
    std::vector<Complex> xc(20011);
    std::vector<Float> xf(20011);
    for (size_t i = 0; i < xc.size(); i++) { xc[i] = Complex(std::sin(0.01f * i), std::cos(0.013f * i)); xf[i] = std::sin(0.02f * i); }
    const auto lp31 = fir::low_pass_complex(1e6f, 1e5f, 2e5f * 53.0f / 22.0f / 3.1f, window::WindowType::Hamming());
    const auto lp = fir::low_pass_complex(1e6f, 1e5f, 5e4f, window::WindowType::Hamming());
    const auto W = window::WindowType::Hamming();
    auto same = [&](const char* what, const auto& a, const auto& b) {
        EXPECT(a.data.size() == b.data.size(), "%s: %zu vs %zu samples", what, a.data.size(), b.data.size());
        EXPECT(a.tags == b.tags, "%s: resident %s\n   host-window %s", what, show(a.tags).c_str(), show(b.tags).c_str());
        EXPECT(!b.tags.empty(), "%s: the host-window block delivered no tag at all", what);
    };
    for (size_t deci : {1, 3, 8}) {
        auto a = resident1<G, Complex, Complex>(xc, 3, 8192, "FirFilter", [&] { return rr_fir_c32_create(c32(lp), lp.size(), deci, 0, 0, 0); });
        auto b = drive<G, Complex, Complex>(xc, 3, [&](G& g, ReadStream<Complex> s0) {
            auto [f, o] = FirFilter<Complex>::builder(lp).deci(deci).build(std::move(s0));
            g.add(std::move(f));
            return std::move(o);
        });
        same(("FirFilter deci " + std::to_string(deci)).c_str(), a, b);
        // and the closed form: input position a -> a / deci while that output exists
        std::vector<size_t> want;
        for (size_t rep = 0; rep < 3; rep++) { const size_t o = rep * xc.size() / deci; if (o < a.data.size()) for (int k = 0; k < (rep ? 2 : 3); k++) want.push_back(o); }
        std::vector<size_t> got;
        for (auto& t : a.tags) got.push_back(t.pos());
        EXPECT(got == want, "FirFilter deci %zu: positions differ from a / deci", deci);
    }
    {
        auto a = resident1<G, Complex, Complex>(xc, 3, 8192, "FftFilter", [&] { return rr_fftfilter_create(c32(lp), lp.size()); });
        auto b = drive<G, Complex, Complex>(xc, 3, [&](G& g, ReadStream<Complex> s0) {
            auto [f, o] = FftFilter::new_(std::move(s0), lp);
            g.add(std::move(f));
            return std::move(o);
        });
        same("FftFilter", a, b);
    }
    {
        std::vector<Float> tf(lp.size());
        for (size_t i = 0; i < tf.size(); i++) tf[i] = lp[i].real();
        // (rings that hold the whole stream, as the reference's 4,096,000-byte rings do here: FftFilterFloat moves min(inner_out,
        //  free) samples out per work() and reports the inner filter's WaitForStream(src) even when some are left inside
        //  (fft_filter.rs:453-489), so a thread-per-block run whose output ring fills up ends with those samples unsent — in
        //  the reference too.  Not a tag matter.)
        auto a = resident1<G, Float, Float>(xf, 3, 1 << 20, "FftFilterFloat", [&] { return rr_fftfilter_float_create(tf.data(), tf.size()); });
        auto b = drive<G, Float, Float>(xf, 3, [&](G& g, ReadStream<Float> s0) {
            auto [f, o] = FftFilterFloat::new_(std::move(s0), tf);
            g.add(std::move(f));
            return std::move(o);
        });
        same("FftFilterFloat", a, b);
    }
    {
        auto a = resident1<G, Float, Complex>(xf, 3, 8192, "Hilbert", [&] { return rr_hilbert_create(65, W.kind, W.parm); });
        auto b = drive<G, Float, Complex>(xf, 3, [&](G& g, ReadStream<Float> s0) {
            auto [f, o] = Hilbert::new_(std::move(s0), 65, W);
            g.add(std::move(f));
            return std::move(o);
        });
        same("Hilbert", a, b);
    }
    // the fused forms against the reference's TWO blocks in sequence on host windows
    {
        auto a = resident1<G, Complex, Complex>(xc, 3, 16384, "FirFftFilter",
                                                [&] { return rr_fir_fftfilter_create(c32(lp31), lp31.size(), c32(lp), lp.size()); });
        auto two = drive<G, Complex, Complex>(xc, 3, [&](G& g, ReadStream<Complex> s0) {
            auto [f1, o1] = FirFilter<Complex>::builder(lp31).build(std::move(s0));
            auto [f2, o2] = FftFilter::new_(std::move(o1), lp);
            g.add(std::move(f1));
            g.add(std::move(f2));
            return std::move(o2);
        });
        same("fused FirFilter -> FftFilter (resident)", a, two);
        auto hostfused = drive<G, Complex, Complex>(xc, 3, [&](G& g, ReadStream<Complex> s0) {
            auto [f, o] = FirFftFilter(std::move(s0), lp31, lp);
            g.add(std::move(f));
            return std::move(o);
        });
        same("fused FirFilter -> FftFilter (host windows)", hostfused, two);
    }
    for (size_t deci : {1, 8}) {
        auto a = resident1<G, Float, Complex>(xf, 3, 16384, "HilbertFir", [&] {
            return rr_hilbert_fir_create(65, W.kind, W.parm, c32(lp), lp.size(), deci, 0, 0, 0);
        });
        auto two = drive<G, Float, Complex>(xf, 3, [&](G& g, ReadStream<Float> s0) {
            auto [f1, o1] = Hilbert::new_(std::move(s0), 65, W);
            auto [f2, o2] = FirFilter<Complex>::builder(lp).deci(deci).build(std::move(o1));
            g.add(std::move(f1));
            g.add(std::move(f2));
            return std::move(o2);
        });
        same(("fused Hilbert -> FirFilter (resident) deci " + std::to_string(deci)).c_str(), a, two);
        auto hostfused = drive<G, Float, Complex>(xf, 3, [&](G& g, ReadStream<Float> s0) {
            auto [f, o] = HilbertFir(std::move(s0), 65, W, lp, deci);
            g.add(std::move(f));
            return std::move(o);
        });
        same(("fused Hilbert -> FirFilter (host windows) deci " + std::to_string(deci)).c_str(), hostfused, two);
    }
    // FftStream: input tags dropped, frame tags added (fft_stream.rs:98-111)
    {
        auto a = resident1<G, Complex, Complex>(xc, 3, 8192, "FftStream", [&] { return rr_fftstream_create(64); });
        auto b = drive<G, Complex, Complex>(xc, 3, [&](G& g, ReadStream<Complex> s0) {
            auto [f, o] = FftStream::new_(std::move(s0), 64);
            g.add(std::move(f));
            return std::move(o);
        });
        same("FftStream", a, b);
    }
    // blocks that drop tags in the reference drop them here: FftFilter -> RationalResampler, and the fused FM chain
    {
        auto a = drive<G, Complex, Complex>(xc, 3, [&](G& g, ReadStream<Complex> s0) {
            auto [up, d0] = GpuUpload<Complex>::new_(std::move(s0), 8192);
            auto [f, d1] = GpuResident<Complex, Complex>::new_(rr_fftfilter_create(c32(lp), lp.size()), "FftFilter", std::move(d0), 8192);
            auto [r, d2] = GpuResident<Complex, Complex>::new_(rr_resampler_create(2, 3, sizeof(Complex)), "RationalResampler", std::move(d1), 8192);
            auto [down, h] = GpuDownload<Complex>::new_(std::move(d2));
            g.add(std::move(up)); g.add(std::move(f)); g.add(std::move(r)); g.add(std::move(down));
            return std::move(h);
        });
        EXPECT(a.tags.empty() && !a.data.empty(), "RationalResampler forwarded %zu tags (%zu samples)", a.tags.size(), a.data.size());
        auto c = resident1<G, Complex, Float>(xc, 3, 16384, "FmChain", [&] { return rr_fm_chain_create(c32(lp), lp.size(), 1, 6, 1.0f, RR_ATAN2_EXACT); });
        EXPECT(c.tags.empty() && !c.data.empty(), "FmChain forwarded %zu tags", c.tags.size());
    }
}

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[2]) == "tags") {
        try {
            if (std::string(argv[1]) == "mt") tag_tests<MTGraph>(); else tag_tests<Graph>();
        } catch (const std::exception& e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
        if (g_fail) { fprintf(stderr, "%d tag checks failed\n", g_fail); return 1; }
        printf("OK tags\n");
        return 0;
    }
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
