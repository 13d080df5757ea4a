// test_host_api.cpp — the reference's unit tests for the hot path, written against the C++
// host mirror (rustradio_amd/host/rustradio.hpp) the way the reference writes them
// (VectorSource -> block, work() by hand or Graph, then read_buf()).  Needs a GPU.
// Build: g++ -O2 -std=c++17 tests/cpp/test_host_api.cpp -L rustradio_amd/lib -lrustradio_amd
#include <cmath>
#include <cstdio>
#include <thread>

#include "../../rustradio_amd/host/rustradio.hpp"

using namespace rustradio;
using window::WindowType;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)

static void assert_almost_equal_complex(const Complex* l, size_t ln, const std::vector<Complex>& r) {   // src/lib.rs:846-861
    if (ln != r.size()) { printf("FAIL length %zu vs %zu\n", ln, r.size()); g_fail++; return; }
    for (size_t i = 0; i < ln; i++)
        if (std::abs(l[i] - r[i]) > 0.001f) { printf("FAIL element %zu: (%g,%g) vs (%g,%g)\n", i, l[i].real(), l[i].imag(), r[i].real(), r[i].imag()); g_fail++; return; }
}
static bool is_wait(const BlockRet& r) { return r.kind == BlockRet::WaitForStream; }

static const std::vector<Complex> SIX = {{1, 0}, {2, 0}, {3, 0.2f}, {4.1f, 0}, {5, 0}, {6, 0.2f}};

static void test_complex() {   // src/fir.rs:921-950
    Fir<Complex> filter({{0.1f, 0}, {1, 0}, {0, 0.2f}});
    auto a = filter.filter_n(SIX, 1);
    assert_almost_equal_complex(a.data(), a.size(), {{2.3f, 0.22f}, {3.41f, 0.6f}, {4.56f, 0.6f}, {5.6f, 0.84f}});
    auto b = filter.filter_n(SIX, 2);
    assert_almost_equal_complex(b.data(), b.size(), {{2.3f, 0.22f}, {4.56f, 0.6f}});
    std::vector<Complex> c(2);
    filter.filter_n_inplace(SIX, 2, c);
    assert_almost_equal_complex(c.data(), c.size(), {{2.3f, 0.22f}, {4.56f, 0.6f}});
    CHECK(std::abs(filter.filter(SIX) - Complex(2.3f, 0.22f)) < 1e-3f);
}

static void test_identity() {  // src/fir.rs:691-741
    for (size_t deci = 1; deci <= 3 * SIX.size(); deci++) {
        auto [src, src_out] = VectorSource<Complex>::new_(SIX, Repeat::finite(2));
        CHECK(src->work().kind == BlockRet::Again);
        CHECK(src->work().kind == BlockRet::EOF_);
        auto [b, os] = FirFilter<Complex>::builder({{1, 0}}).deci(deci).build(std::move(src_out));
        if (deci <= 2 * SIX.size()) CHECK(b->work().kind == BlockRet::Again);
        CHECK(is_wait(b->work()));
        auto [res, tags] = os.read_buf();
        const size_t max = 2 * SIX.size() / deci;
        if (!res.is_empty()) {
            std::vector<Tag> want = {Tag(0, "VectorSource::start", true), Tag(0, "VectorSource::repeat", (uint64_t)0),
                                     Tag(0, "VectorSource::first", true), Tag(6 / deci, "VectorSource::start", true),
                                     Tag(6 / deci, "VectorSource::repeat", (uint64_t)1)};
            CHECK(tags == want);
        }
        std::vector<Complex> want;
        for (size_t i = 0; i < 2 * SIX.size() && want.size() < max; i += deci) want.push_back(SIX[i % SIX.size()]);
        assert_almost_equal_complex(res.slice(), res.len(), want);
    }
}

static void moving_avg() {     // src/fir.rs:880-919
    const std::vector<Complex> full = {{1.5f, 0}, {2.5f, 0.1f}, {3.55f, 0.1f}, {4.55f, 0}, {5.5f, 0.1f}};
    for (size_t deci = 1; deci <= SIX.size() + 1; deci++) {
        auto [src, src_out] = VectorSource<Complex>::new_(SIX);
        src->work();
        auto [b, os] = FirFilter<Complex>::builder({{0.5f, 0}, {0.5f, 0}}).deci(deci).build(std::move(src_out));
        if (deci < SIX.size()) CHECK(b->work().kind == BlockRet::Again);
        CHECK(is_wait(b->work()));
        auto [res, tags] = os.read_buf();
        std::vector<Complex> want;
        for (size_t i = 0; i < full.size() && want.size() < (SIX.size() - 1) / deci; i += deci) want.push_back(full[i]);
        assert_almost_equal_complex(res.slice(), res.len(), want);
    }
}

static void translate_matches_mixed_input() {   // src/fir.rs:744-789
    std::vector<Complex> input, mixed;
    for (int i = 0; i < 32; i++) input.emplace_back((float)i, (float)i * 0.25f);
    const std::vector<Complex> taps = {{0.5f, -0.1f}, {1.0f, 0.2f}, {-0.25f, 0.05f}, {0.125f, -0.3f}};
    const double step = -2.0 * M_PI * 2.0 / 8.0;
    const Complex rot((float)std::cos(step), (float)std::sin(step));
    Complex phase(1, 0);
    for (auto& s : input) { mixed.push_back(s * phase); phase *= rot; }
    auto [sa, sa_out] = VectorSource<Complex>::new_(input);
    CHECK(sa->work().kind == BlockRet::EOF_);
    auto [tr, tr_out] = FirFilter<Complex>::builder(taps).deci(3).translate(8.0f, 2.0f).build(std::move(sa_out));
    CHECK(tr->work().kind == BlockRet::Again);
    CHECK(is_wait(tr->work()));
    auto [sb, sb_out] = VectorSource<Complex>::new_(mixed);
    CHECK(sb->work().kind == BlockRet::EOF_);
    auto [mn, mn_out] = FirFilter<Complex>::builder(taps).deci(3).build(std::move(sb_out));
    CHECK(mn->work().kind == BlockRet::Again);
    CHECK(is_wait(mn->work()));
    auto [r1, t1] = tr_out.read_buf();
    auto [r2, t2] = mn_out.read_buf();
    assert_almost_equal_complex(r1.slice(), r1.len(), std::vector<Complex>(r2.begin(), r2.end()));
}

static void test_filter_generator() {   // src/fir.rs:952-986 (first/centre taps)
    auto taps = fir::low_pass_complex(10000.0f, 1000.0f, 1000.0f, WindowType::Hamming());
    CHECK(taps.size() == 25);
    CHECK(std::abs(taps[12] - Complex(0.19922684f, 0)) < 1e-3f);
    CHECK(std::abs(taps[0] - Complex(0.002010403f, 0)) < 1e-3f);
}

static void fft_tag_propagation() {     // src/fft_filter.rs:551-574
    auto [src, o] = VectorSource<Complex>::new_(std::vector<Complex>(1024), Repeat::finite(2));
    auto [fft, out] = FftFilter::new_(std::move(o), {Complex(0, 0)});
    src->work(); src->work();
    fft->work();
    auto [res, tags] = out.read_buf();
    std::vector<Tag> want = {Tag(0, "VectorSource::start", true), Tag(0, "VectorSource::repeat", (uint64_t)0),
                             Tag(0, "VectorSource::first", true), Tag(1024, "VectorSource::start", true),
                             Tag(1024, "VectorSource::repeat", (uint64_t)1)};
    CHECK(tags == want);
    CHECK(res.len() == 2048);
}

static void resampler_examples() {      // src/rational_resampler.rs:249-277, 130-135
    std::vector<uint32_t> input(50);
    for (uint32_t i = 0; i < 50; i++) input[i] = i;
    {
        auto [src, so] = VectorSource<uint32_t>::new_(input);
        CHECK(src->work().kind == BlockRet::EOF_);
        auto [b, os] = RationalResampler<uint32_t>::new_(std::move(so), 25, 64);
        CHECK(is_wait(b->work()));
        auto [res, tags] = os.read_buf();
        const std::vector<uint32_t> want = {0, 2, 5, 7, 10, 12, 15, 17, 20, 23, 25, 28, 30, 33, 35, 38, 40, 43, 46, 48};
        CHECK(std::vector<uint32_t>(res.begin(), res.end()) == want);
    }
    bool threw = false;
    try { auto [src, so] = VectorSource<uint32_t>::new_(input); RationalResampler<uint32_t>::new_(std::move(so), 0, 1); }
    catch (const Error&) { threw = true; }
    CHECK(threw);
}

static void quad_known() {              // src/quadrature_demod.rs:210-264
    {
        auto [b, prev] = VectorSource<Complex>::new_(std::vector<Complex>(4));
        b->work();
        auto [q, out] = QuadratureDemod::new_(std::move(prev), 1.0f);
        q->work();
        auto [o, tags] = out.read_buf();
        CHECK(o.len() == 3);
        for (auto v : o) CHECK(v == 0.0f);
    }
    {
        auto [b, prev] = VectorSource<Complex>::new_({{1, 0}, {0.707f, -0.707f}, {0, -1}, {-1, 0}});
        b->work();
        auto [q, out] = QuadratureDemod::new_(std::move(prev), 1.0f);
        q->work();
        auto [o, tags] = out.read_buf();
        const float want[3] = {-(float)M_PI / 4, -(float)M_PI / 4, -(float)M_PI / 2};
        CHECK(o.len() == 3);
        for (size_t i = 0; i < 3 && i < o.len(); i++) CHECK(std::fabs(o.slice()[i] - want[i]) < 1e-3f);
    }
}

static void rtlsdr_decode_tests() {     // src/rtlsdr_decode.rs:54-100
    {   // empty
        auto [b, prev] = VectorSource<uint8_t>::new_(std::vector<uint8_t>{});
        CHECK(b->work().kind == BlockRet::EOF_);
        auto [d, out] = RtlSdrDecode::new_(std::move(prev));
        CHECK(d->work().kind == BlockRet::WaitForStream);
        auto [o, tags] = out.read_buf();
        CHECK(o.len() == 0);
    }
    {   // some_input: the reference asserts exact equality with these literals
        auto [b, prev] = VectorSource<uint8_t>::new_({0, 10, 20, 10, 0, 13});
        b->work();
        auto [d, out] = RtlSdrDecode::new_(std::move(prev));
        CHECK(d->work().kind == BlockRet::WaitForStream);
        auto [o, tags] = out.read_buf();
        const Complex want[3] = {{-1.016f, -0.93600005f}, {-0.85600007f, -0.93600005f}, {-1.016f, -0.91200006f}};
        CHECK(o.len() == 3);
        for (size_t i = 0; i < 3 && i < o.len(); i++) CHECK(o.slice()[i] == want[i]);
    }
    {   // uneven
        auto [b, prev] = VectorSource<uint8_t>::new_({0, 10, 20, 10, 0});
        b->work();
        auto [d, out] = RtlSdrDecode::new_(std::move(prev));
        d->work();
        auto [o, tags] = out.read_buf();
        CHECK(o.len() == 2);
    }
}

static void hilbert_rejects_even() {    // src/hilbert.rs:44-47
    bool threw = false;
    try { auto [s, so] = VectorSource<Float>::new_({1.f, 2.f}); Hilbert::new_(std::move(so), 64, WindowType::Hamming()); }
    catch (const Error&) { threw = true; }
    CHECK(threw);
}

static void graph_fm_chain() {
    // examples/rtl_fm.rs:381-419 style chain on a Graph: VectorSource -> FftFilter -> RationalResampler
    // -> QuadratureDemod -> VectorSink; a constant-frequency tone must demodulate to its phase step.
    const size_t n = 600000;
    const double w = 2.0 * M_PI * 10e3 / 2.4e6;
    std::vector<Complex> x(n);
    for (size_t i = 0; i < n; i++) x[i] = Complex((float)std::cos(w * i), (float)std::sin(w * i));
    auto taps = fir::low_pass_complex(2.4e6f, 100e3f, 12.5e3f, WindowType::Hamming());
    CHECK(taps.size() == 463);
    auto [src, s0] = VectorSource<Complex>::new_(x);
    auto [fft, s1] = FftFilter::new_(std::move(s0), taps);
    auto [rs, s2] = RationalResampler<Complex>::new_(std::move(s1), 1, 6);
    auto [qd, s3] = QuadratureDemod::new_(std::move(s2), 1.0f);
    auto sink = std::make_unique<VectorSink<Float>>(std::move(s3));
    auto hook = sink->hook();
    Graph g;
    g.add(std::move(src)); g.add(std::move(fft)); g.add(std::move(rs)); g.add(std::move(qd)); g.add(std::move(sink));
    g.run();
    const size_t n1 = (n / 561) * 561, n2 = (n1 + 5) / 6;
    CHECK(hook->size() == n2 - 1);
    double worst = 0;
    for (size_t i = 200; i < hook->size(); i++) worst = std::max(worst, std::fabs((double)(*hook)[i] - 6.0 * w));
    CHECK(worst < 1e-4);
}

// Graph-level fusions against the blocks they replace, both through Graph on the GPU: FirFilter -> FftFilter as one
// convolution, the metric's four-block chain as one kernel, the rtl_fm audio stage as one kernel.
template <class T> static double max_rel(const std::vector<T>& a, const std::vector<T>& b, double floor_ = 0.0) {
    double e = 0, m = floor_;
    for (size_t i = 0; i < b.size(); i++) m = std::max(m, (double)std::abs(b[i]));
    for (size_t i = 0; i < std::min(a.size(), b.size()); i++) e = std::max(e, (double)std::abs(a[i] - b[i]));
    return m > 0 ? e / m : e;
}
// fft.rs:58-105 — the reference's own two tests of the message block, plus one DC check
static void fft_message_block() {
    Fft f(1024);
    auto z = f.process(std::vector<Complex>(1024));
    for (auto v : z) CHECK(v == Complex(0, 0));
    auto dc = f.process(std::vector<Complex>(1024, Complex(1, 0)));
    CHECK(std::abs(dc[0] - Complex(1024, 0)) < 1e-2f && std::abs(dc[1]) < 1e-3f);
    bool threw = false;
    try { Fft(4).process(std::vector<Complex>(3)); } catch (const Error& e) { threw = std::string(e.what()).find("FFT expected 4 samples, got 3") != std::string::npos; }
    CHECK(threw);
    threw = false;
    try { Fft bad(0); } catch (const Error&) { threw = true; }
    CHECK(threw);
}

// rr_fanout_* from compiled code, as a Rust graph would drive it (one-rank group: the double buffer without a communicator).
// The "source block" and the consumer are rr blocks working on device pointers; device memory comes from rr_dstream rings.
static void fanout_from_c_abi() {
    const size_t n = 100000, tiles = 5;
    std::vector<float> host(n * tiles);
    for (size_t i = 0; i < host.size(); i++) host[i] = (float)(i % 1000) * 0.25f;
    rr_dstream* src = rr_dstream_create(4, 4 * host.size());
    rr_dstream* dst = rr_dstream_create(4, 4 * host.size());
    CHECK(src && dst);
    CHECK(rr_dstream_copy_in(src, 0, host.data(), host.size(), nullptr) == 0 && rr_dstream_produce(src, host.size()) == 0);
    const void* dsrc = nullptr; void* ddst = nullptr;
    CHECK(rr_dstream_read_buf(src, &dsrc) == host.size());
    CHECK(rr_dstream_write_buf(dst, &ddst, nullptr) >= host.size());
    rr_fanout* f = rr_fanout_create(nullptr, 0, 1, 0, 4 * n, 0);
    CHECK(f != nullptr);
    rr_block* producer = rr_multiply_const_f32_create(1.0f);
    rr_block* consumer = rr_multiply_const_f32_create(3.0f);
    size_t c = 0, p = 0, need = 0;
    auto produce = [&](unsigned long long t) {
        void* buf = rr_fanout_produce_buf(f, t, nullptr);
        CHECK(buf != nullptr);
        rr_block_work_dev(producer, static_cast<const float*>(dsrc) + t * n, n, buf, n, &c, &p, &need, nullptr);
        CHECK(c == n && p == n);
        CHECK(rr_fanout_submit(f, t, nullptr) == 0);
    };
    produce(0);
    for (unsigned long long t = 0; t < tiles; t++) {
        if (t + 1 < tiles) produce(t + 1);                       // tile t + 1 travels while tile t is consumed
        const void* x = rr_fanout_acquire(f, t, nullptr);
        CHECK(x != nullptr);
        rr_block_work_dev(consumer, x, n, static_cast<float*>(ddst) + t * n, n, &c, &p, &need, nullptr);
        CHECK(c == n && p == n);
        CHECK(rr_fanout_release(f, t, nullptr) == 0);
    }
    CHECK(rr_fanout_produce_buf(f, tiles + 1, nullptr) == nullptr);   // out of order: an error, not a race
    CHECK(rr_dstream_produce(dst, host.size()) == 0);
    std::vector<float> back(host.size());
    CHECK(rr_dstream_copy_out(dst, 0, back.data(), back.size(), nullptr) == 0);
    bool same = true;
    for (size_t i = 0; i < host.size(); i++) same = same && back[i] == host[i] * 3.0f;
    CHECK(same);
    double ms = -1; size_t nb = 99;
    CHECK(rr_fanout_stats(f, &ms, &nb) == 0 && nb == 0);           // no communicator: nothing to time
    rr_block_destroy(producer); rr_block_destroy(consumer);
    rr_fanout_destroy(f); rr_dstream_destroy(src); rr_dstream_destroy(dst);
}

static void fused_blocks_equal_their_chains() {
    const size_t n = 300000;
    std::vector<Complex> x(n);
    uint64_t st = 12345;
    auto rnd = [&] { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (float)((st >> 40) / 8388608.0 - 1.0); };
    for (auto& v : x) v = Complex(rnd(), rnd());
    auto t1 = fir::low_pass_complex(10e6f, 1e6f, 190e3f, WindowType::Hamming());
    auto t2 = fir::low_pass_complex(10e6f, 1e6f, 60e3f, WindowType::Hamming());
    CHECK(t1.size() == 127 && t2.size() == 401);
    auto run_c = [&](bool fused) {
        auto [src, s0] = VectorSource<Complex>::new_(x);
        Graph g;
        g.add(std::move(src));
        ReadStream<Complex> last = std::move(s0);
        if (fused) {
            auto [b, s1] = FirFftFilter(std::move(last), t1, t2);
            g.add(std::move(b)); last = std::move(s1);
        } else {
            auto [a, s1] = FirFilter<Complex>::builder(t1).build(std::move(last));
            auto [b, s2] = FftFilter::new_(std::move(s1), t2);
            g.add(std::move(a)); g.add(std::move(b)); last = std::move(s2);
        }
        auto sink = std::make_unique<VectorSink<Complex>>(std::move(last));
        auto hook = sink->hook();
        g.add(std::move(sink));
        g.run();
        return *hook;
    };
    auto yu = run_c(false), yf = run_c(true);
    CHECK(yu.size() == yf.size() && yu.size() > 290000);
    CHECK(max_rel(yf, yu) < 2e-5);
    auto run_f = [&](bool fused) {
        auto [src, s0] = VectorSource<Complex>::new_(x);
        Graph g;
        g.add(std::move(src));
        ReadStream<Float> out;
        if (fused) {
            auto [b, s1] = FirFmChain(std::move(s0), t1, t2, 1, 4, 0.5f);
            g.add(std::move(b)); out = std::move(s1);
        } else {
            auto [a, s1] = FirFilter<Complex>::builder(t1).build(std::move(s0));
            auto [b, s2] = FftFilter::new_(std::move(s1), t2);
            auto [c, s3] = RationalResampler<Complex>::new_(std::move(s2), 1, 4);
            auto [d, s4] = QuadratureDemod::new_(std::move(s3), 0.5f);
            g.add(std::move(a)); g.add(std::move(b)); g.add(std::move(c)); g.add(std::move(d)); out = std::move(s4);
        }
        auto sink = std::make_unique<VectorSink<Float>>(std::move(out));
        auto hook = sink->hook();
        g.add(std::move(sink));
        g.run();
        return *hook;
    };
    auto du = run_f(false), df = run_f(true);
    CHECK(du.size() == df.size() && du.size() > 70000);
    // noise input filtered to |r| ~ 0.3: angle errors stay far below 1e-3 rad except where |r| is tiny; compare robustly
    size_t bad = 0;
    for (size_t i = 0; i < du.size(); i++) {
        double d = std::fabs((double)du[i] - (double)df[i]);
        d = std::min(d, 2 * M_PI * 0.5 - d);
        if (d > 1e-3) bad++;
    }
    CHECK(bad < du.size() / 1000);
    // audio stage
    std::vector<Float> xr(n);
    for (auto& v : xr) v = rnd();
    auto ta = fir::low_pass(200000.0f, 44100.0f, 500.0f, WindowType::Hamming());
    CHECK(ta.size() == 963);
    auto run_a = [&](bool fused) {
        auto [src, s0] = VectorSource<Float>::new_(xr);
        Graph g;
        g.add(std::move(src));
        ReadStream<Float> out;
        if (fused) {
            auto [b, s1] = AudioChain(std::move(s0), ta, 48000, 200000, 0.25f);
            g.add(std::move(b)); out = std::move(s1);
        } else {
            auto [a, s1] = FftFilterFloat::new_(std::move(s0), ta);
            auto [b, s2] = RationalResampler<Float>::new_(std::move(s1), 48000, 200000);
            auto [c, s3] = MultiplyConst<Float>::new_(std::move(s2), 0.25f);
            g.add(std::move(a)); g.add(std::move(b)); g.add(std::move(c)); out = std::move(s3);
        }
        auto sink = std::make_unique<VectorSink<Float>>(std::move(out));
        auto hook = sink->hook();
        g.add(std::move(sink));
        g.run();
        return *hook;
    };
    auto au = run_a(false), af = run_a(true);
    CHECK(au.size() == af.size() && au.size() > 70000);
    CHECK(max_rel(af, au) < 2e-5);
}

// The same Graph, rings in host memory vs rings in HBM: identical samples and tags.
template <class F> static auto with_memory(Memory m, F&& f) {
    const Memory old = default_memory();
    default_memory() = m;
    auto r = f();
    default_memory() = old;
    return r;
}
static void device_resident_graph() {
    const size_t n = 300000;
    std::vector<uint8_t> bytes(2 * n + 1);
    uint32_t lcg = 12345;
    for (auto& b : bytes) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
    auto taps = fir::low_pass_complex(2.4e6f, 100e3f, 50e3f, WindowType::Hamming());
    auto run = [&]() {
        auto [src, s0] = VectorSource<uint8_t>::new_(bytes);
        auto [dec, s1] = RtlSdrDecode::new_(std::move(s0));
        auto [fft, s2] = FftFilter::new_(std::move(s1), taps);
        auto [rs, s3] = RationalResampler<Complex>::new_(std::move(s2), 3, 7);
        auto [qd, s4] = QuadratureDemod::new_(std::move(s3), 0.5f);
        auto sink = std::make_unique<VectorSink<Float>>(std::move(s4));
        auto hook = sink->hook();
        Graph g;
        g.add(std::move(src)); g.add(std::move(dec)); g.add(std::move(fft)); g.add(std::move(rs)); g.add(std::move(qd));
        g.add(std::move(sink));
        g.run();
        return *hook;
    };
    const auto yh = with_memory(Memory::Host, run);
    const auto yd = with_memory(Memory::Device, run);
    CHECK(yh.size() > 1000);
    CHECK(yh == yd);
    // tags travel host-side with a device-resident FIR exactly as with a host one (fir.rs:536-545)
    auto run_fir = [&]() {
        std::vector<Complex> x(5000);
        for (size_t i = 0; i < x.size(); i++) x[i] = Complex((float)i, -(float)i);
        auto [src, s0] = VectorSource<Complex>::new_(x, Repeat::finite(3));
        auto [f, s1] = FirFilter<Complex>::builder({{0.5f, 0}, {0.25f, 0}, {0, 0.25f}}).deci(4).build(std::move(s0));
        auto sink = std::make_unique<VectorSink<Complex>>(std::move(s1));
        auto hook = sink->hook();
        auto th = sink->tag_hook();
        Graph g;
        g.add(std::move(src)); g.add(std::move(f)); g.add(std::move(sink));
        g.run();
        std::vector<std::pair<size_t, std::string>> tags;
        for (auto& t : *th) tags.emplace_back(t.pos(), t.key());
        return std::make_pair(*hook, tags);
    };
    const auto fh = with_memory(Memory::Host, run_fir);
    const auto fd = with_memory(Memory::Device, run_fir);
    CHECK(fh.first.size() > 3000 && fh.first == fd.first);
    CHECK(!fh.second.empty() && fh.second == fd.second);
    // MemCopy: a CPU-side source feeding a device-resident chain and back
    {
        std::vector<Float> x(70000);
        for (size_t i = 0; i < x.size(); i++) x[i] = (float)std::sin(0.5 * (double)i);
        auto [src, s0] = VectorSource<Float>::new_(x);                                  // host ring
        auto [up, s1] = MemCopy<Float>::new_(std::move(s0), Memory::Device);
        auto [hb, s2] = with_memory(Memory::Device, [&, s = std::move(s1)]() mutable {
            return Hilbert::new_(std::move(s), 65, WindowType::Hamming()); });
        auto [down, s3] = MemCopy<Complex>::new_(std::move(s2), Memory::Host);
        auto sink = std::make_unique<VectorSink<Complex>>(std::move(s3));
        auto hook = sink->hook();
        Graph g;
        g.add(std::move(src)); g.add(std::move(up)); g.add(std::move(hb)); g.add(std::move(down)); g.add(std::move(sink));
        g.run();
        CHECK(hook->size() == x.size());
        double worst = 0;                       // analytic signal of a sine: |a| -> 1, re = the input delayed by 32
        for (size_t i = 2000; i < hook->size(); i++) worst = std::max(worst, std::fabs(std::abs((*hook)[i]) - 1.0));
        CHECK(worst < 2e-2);
        CHECK((*hook)[5000].real() == x[5000 - 33]);
    }
}

static void sync_blocks() {                      // multiply_const.rs:20-22, quadrature_demod.rs:158-164
    auto [src, s0] = VectorSource<Float>::new_({1.5f, -2.0f, 0.25f});
    src->work();
    auto [m, s1] = MultiplyConst<Float>::new_(std::move(s0), 3.0f);
    CHECK(is_wait(m->work()));
    auto [o, tags] = s1.read_buf();
    CHECK(o.len() == 3 && o.slice()[0] == 4.5f && o.slice()[1] == -6.0f && o.slice()[2] == 0.75f);
    CHECK(tags.size() >= 1 && tags[0].pos() == 0);          // VectorSource::start passes through
    auto [src2, c0] = VectorSource<Complex>::new_({{1, 0}, {0, 1}, {-1, 0}, {0, -1}});
    src2->work();
    auto [f, c1] = FastFM::new_(std::move(c0));
    CHECK(is_wait(f->work()));
    auto [fo, ft] = c1.read_buf();
    // q1 = q2 = 0 at start: out = [0, (1-0)*1 - (0-0)*0, (0-0)*0 - (-1-1)*1, (-1-1)*(-1) - ...]
    const float want[4] = {0.0f, 1.0f, 2.0f, 2.0f};
    CHECK(fo.len() == 4);
    for (size_t i = 0; i < 4 && i < fo.len(); i++) CHECK(fo.slice()[i] == want[i]);
}

static void fftstream_adds_frame_tags() {        // src/fft_stream.rs:125-150
    auto src = ReadStream<Complex>::from_slice(std::vector<Complex>(8).data(), 8);
    auto [fft, out] = FftStream::new_(std::move(src), 4);
    CHECK(fft->work().kind == BlockRet::Again);
    auto [buf, tags] = out.read_buf();
    CHECK(buf.len() == 8);
    const std::vector<std::pair<size_t, std::string>> want = {{0, TAG_FRAME_SIZE}, {0, TAG_FRAME}, {3, TAG_FRAME},
                                                              {4, TAG_FRAME_SIZE}, {4, TAG_FRAME}, {7, TAG_FRAME}};
    CHECK(tags.size() == want.size());
    for (size_t i = 0; i < want.size() && i < tags.size(); i++) CHECK(tags[i].pos() == want[i].first && tags[i].key() == want[i].second);
    for (size_t i = 0; i < 8; i++) CHECK(buf.slice()[i] == Complex(0, 0));
    // a tone lands in its bin
    std::vector<Complex> x(2048);
    for (size_t i = 0; i < x.size(); i++) x[i] = Complex((float)std::cos(2 * M_PI * 5 * i / 1024.0), (float)std::sin(2 * M_PI * 5 * i / 1024.0));
    auto [f2, o2] = FftStream::new_(ReadStream<Complex>::from_slice(x.data(), x.size()), 1024);
    CHECK(f2->work().kind == BlockRet::Again);
    auto [b2, t2] = o2.read_buf();
    CHECK(b2.len() == 2048 && std::abs(b2.slice()[5] - Complex(1024, 0)) < 0.05f && std::abs(b2.slice()[6]) < 0.05f
          && std::abs(b2.slice()[1024 + 5] - Complex(1024, 0)) < 0.05f);
    bool threw = false;
    try { FftStream::new_(ReadStream<Complex>::from_slice(x.data(), 8), 0); } catch (const Error&) { threw = true; }
    CHECK(threw);
}

static void file_source_tests() {                 // src/file_source.rs:171-258
    const std::string fn = "/tmp/rr_filesource_test.bin";
    auto write = [&](std::vector<uint8_t> b) { FILE* f = std::fopen(fn.c_str(), "wb"); std::fwrite(b.data(), 1, b.size(), f); std::fclose(f); };
    const std::vector<uint8_t> four = {0, 0, 128, 63, 0, 0, 64, 64, 195, 245, 72, 64, 195, 245, 72, 192};
    {   // source_f32
        write(four);
        auto [src, out] = FileSource<Float>::new_(fn);
        CHECK(src->work().kind == BlockRet::Again);
        CHECK(src->work().kind == BlockRet::EOF_);
        auto [res, tags] = out.read_buf();
        CHECK(res.len() == 4 && res.slice()[0] == 1.0f && res.slice()[1] == 3.0f && res.slice()[2] == 3.14f && res.slice()[3] == -3.14f);
    }
    {   // source_f32_partial_tail
        write(std::vector<uint8_t>(four.begin(), four.end() - 1));
        auto [src, out] = FileSource<Float>::new_(fn);
        CHECK(src->work().kind == BlockRet::Again);
        CHECK(src->work().kind == BlockRet::EOF_);
        auto [res, tags] = out.read_buf();
        CHECK(res.len() == 3 && res.slice()[2] == 3.14f);
    }
    {   // source_f32_twice
        write(four);
        auto [src, out] = FileSource<Float>::new_(fn, Repeat::finite(2));
        CHECK(src->work().kind == BlockRet::Again);
        CHECK(src->work().kind == BlockRet::Again);
        CHECK(src->work().kind == BlockRet::Again);
        CHECK(src->work().kind == BlockRet::EOF_);
        auto [res, tags] = out.read_buf();
        CHECK(res.len() == 8 && res.slice()[4] == 1.0f && res.slice()[7] == -3.14f);
    }
    {   // a .c32 file straight into an HBM ring and through a GPU filter == the same data from a VectorSource
        std::vector<Complex> x(100000);
        for (size_t i = 0; i < x.size(); i++) x[i] = Complex((float)std::sin(0.01 * i), (float)std::cos(0.017 * i));
        { FILE* f = std::fopen(fn.c_str(), "wb"); std::fwrite(x.data(), sizeof(Complex), x.size(), f); std::fwrite("abc", 1, 3, f); std::fclose(f); }
        auto taps = fir::low_pass_complex(2.4e6f, 100e3f, 50e3f, WindowType::Hamming());
        auto run = [&](bool from_file) {
            default_memory() = Memory::Device;
            std::unique_ptr<Block> src; ReadStream<Complex> s0;
            if (from_file) { auto [b, r] = FileSource<Complex>::new_(fn); src = std::move(b); s0 = std::move(r); }
            else { auto [b, r] = VectorSource<Complex>::new_(x); src = std::move(b); s0 = std::move(r); }
            auto [fft, s1] = FftFilter::new_(std::move(s0), taps);
            auto sink = std::make_unique<VectorSink<Complex>>(std::move(s1));
            auto hook = sink->hook();
            Graph g;
            g.add(std::move(src)); g.add(std::move(fft)); g.add(std::move(sink));
            g.run();
            default_memory() = Memory::Host;
            return *hook;
        };
        const auto a = run(true), b = run(false);
        CHECK(a.size() > 90000 && a == b);
    }
    std::remove(fn.c_str());
}

// `Block: Send`, one thread per block instance (MTGraph, src/mtgraph.rs:80-82): N handles driven from N host
// threads at the same time give the single-threaded results.
static void handles_on_concurrent_threads() {
    std::vector<Complex> x(400000);
    uint32_t lcg = 7;
    for (auto& v : x) { lcg = lcg * 1664525u + 1013904223u; v = Complex((float)(lcg >> 8) / 8388608.0f - 1.0f, (float)((lcg * 13u) >> 8) / 8388608.0f - 1.0f); }
    auto taps = fir::low_pass_complex(2.4e6f, 100e3f, 12.5e3f, WindowType::Hamming());
    auto job = [&](int kind) {
        std::vector<Complex> yc; std::vector<Float> yf;
        if (kind == 0) {
            auto [src, s0] = VectorSource<Complex>::new_(x);
            auto [b, s1] = FftFilter::new_(std::move(s0), taps);
            auto sink = std::make_unique<VectorSink<Complex>>(std::move(s1)); auto hook = sink->hook();
            Graph g; g.add(std::move(src)); g.add(std::move(b)); g.add(std::move(sink)); g.run();
            yc = *hook;
        } else if (kind == 1) {
            auto [src, s0] = VectorSource<Complex>::new_(x);
            auto [b, s1] = FirFilter<Complex>::builder(taps).deci(8).build(std::move(s0));
            auto sink = std::make_unique<VectorSink<Complex>>(std::move(s1)); auto hook = sink->hook();
            Graph g; g.add(std::move(src)); g.add(std::move(b)); g.add(std::move(sink)); g.run();
            yc = *hook;
        } else {
            auto [src, s0] = VectorSource<Complex>::new_(x);
            auto [b, s1] = QuadratureDemod::new_(std::move(s0), 1.0f);
            auto sink = std::make_unique<VectorSink<Float>>(std::move(s1)); auto hook = sink->hook();
            Graph g; g.add(std::move(src)); g.add(std::move(b)); g.add(std::move(sink)); g.run();
            yf = *hook;
        }
        return std::make_pair(yc, yf);
    };
    std::vector<std::pair<std::vector<Complex>, std::vector<Float>>> ref(3), got(6);
    for (int k = 0; k < 3; k++) ref[k] = job(k);
    std::vector<std::thread> th;
    for (int i = 0; i < 6; i++) th.emplace_back([&, i] { got[i] = job(i % 3); });
    for (auto& t : th) t.join();
    for (int i = 0; i < 6; i++) CHECK(got[i] == ref[i % 3] && (got[i].first.size() + got[i].second.size()) > 1000);
}

static void tee_and_signal_source() {            // src/tee.rs:10-24, src/signal_source.rs:9-63
    auto [ss, s0] = SignalSourceComplex::new_(1200.0f, 100.0f, 1.0f);
    CHECK(is_wait(ss->work()));
    {
        auto [o, tags] = s0.read_buf();
        CHECK(o.len() == 512000);                // fills the whole ring (quadrature_demod.rs:176-183)
        const double w = 2.0 * M_PI * 100.0 / 1200.0;
        CHECK(std::abs(o.slice()[0] - Complex((float)std::sin(w), (float)std::sin(w - M_PI / 2))) < 1e-6f);
        o.consume(512000 - 10);
    }
    auto [tee, a, b] = Tee<Complex>::new_(std::move(s0));
    CHECK(tee->work().kind == BlockRet::Again);
    auto [ra, ta] = a.read_buf();
    auto [rb, tb] = b.read_buf();
    CHECK(ra.len() == 10 && rb.len() == 10);
    for (size_t i = 0; i < 10; i++) CHECK(ra.slice()[i] == rb.slice()[i]);
}

// Tee and MemCopy between HBM rings stay on the device (rr_dstream_copy; round 2 bounced them through host memory):
// VectorSource(host) -> MemCopy -> device ring -> Tee -> two device rings -> MemCopy each -> host sinks, tags included.
static void tee_and_memcopy_between_device_rings() {
    std::vector<Complex> x(300000);
    for (size_t i = 0; i < x.size(); i++) x[i] = Complex((float)i, -(float)(i % 977));
    auto [src, s0] = VectorSource<Complex>::new_(x);
    auto [up, s1] = MemCopy<Complex>::new_(std::move(s0), Memory::Device);
    auto [tee, a, b] = with_memory(Memory::Device, [&, s = std::move(s1)]() mutable { return Tee<Complex>::new_(std::move(s)); });
    auto [mid, a2] = MemCopy<Complex>::new_(std::move(a), Memory::Device);          // device ring to device ring
    auto [da, ha] = MemCopy<Complex>::new_(std::move(a2), Memory::Host);
    auto [db, hb] = MemCopy<Complex>::new_(std::move(b), Memory::Host);
    auto ka = std::make_unique<VectorSink<Complex>>(std::move(ha));
    auto kb = std::make_unique<VectorSink<Complex>>(std::move(hb));
    auto oa = ka->hook();
    auto ob = kb->hook();
    auto ta = ka->tag_hook();
    Graph g;
    g.add(std::move(src)); g.add(std::move(up)); g.add(std::move(tee)); g.add(std::move(mid)); g.add(std::move(da)); g.add(std::move(db));
    g.add(std::move(ka)); g.add(std::move(kb));
    g.run();
    CHECK(oa->size() == x.size() && ob->size() == x.size());
    CHECK(*oa == x && *ob == x);
    CHECK(!ta->empty() && (*ta)[0].pos() == 0);       // VectorSource::start travels through every copy
}

int main() {
    test_complex(); test_identity(); moving_avg(); translate_matches_mixed_input(); test_filter_generator();
    fft_tag_propagation(); resampler_examples(); quad_known(); rtlsdr_decode_tests(); hilbert_rejects_even(); graph_fm_chain();
    device_resident_graph(); fused_blocks_equal_their_chains(); fanout_from_c_abi(); fft_message_block(); tee_and_signal_source(); tee_and_memcopy_between_device_rings(); sync_blocks(); fftstream_adds_frame_tags(); file_source_tests(); handles_on_concurrent_threads();
    printf(g_fail ? "FAILED (%d)\n" : "OK\n", g_fail);
    return g_fail ? 1 : 0;
}
