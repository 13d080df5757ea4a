// test_host_rings.cpp — CPU only: the host mirror's stream rings (one memfd mapped twice, circular_buffer.rs:98-128) and its
// two runners.  A graph of harness blocks (no GPU block) must END by itself with every sample and tag delivered, under
// Graph::run (src/graph.rs:126-147) and under MTGraph, one thread per block (src/mtgraph.rs:98-116), on rings far
// smaller than the data so that every block waits on both of its streams many times.
#include <cstdio>

#include "../../rustradio_amd/host/rustradio.hpp"

using namespace rustradio;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)

template <class G> static void tee_graph(size_t ring_bytes, size_t n, uint64_t repeats) {
    default_stream_size() = ring_bytes;
    std::vector<Float> x(n);
    for (size_t i = 0; i < n; i++) x[i] = (float)(i % 9973) - 0.25f * (float)(i % 7);
    auto [src, s0] = VectorSource<Float>::new_(x, Repeat::finite(repeats));
    auto [cp, s1] = MemCopy<Float>::new_(std::move(s0), Memory::Host);
    auto [tee, a, b] = Tee<Float>::new_(std::move(s1));
    auto k1 = std::make_unique<VectorSink<Float>>(std::move(a));
    auto k2 = std::make_unique<VectorSink<Float>>(std::move(b));
    auto h1 = k1->hook(), h2 = k2->hook();
    auto t1 = k1->tag_hook();
    G g;
    g.add(std::move(src)); g.add(std::move(cp)); g.add(std::move(tee)); g.add(std::move(k1)); g.add(std::move(k2));
    g.run();                                            // must return
    CHECK(h1->size() == n * repeats);
    CHECK(*h1 == *h2);
    bool same = h1->size() == n * repeats;
    for (size_t i = 0; same && i < h1->size(); i++) same = (*h1)[i] == x[i % n];
    CHECK(same);
    // VectorSource tags (start / repeat per pass, first once) arrive at their samples
    size_t starts = 0;
    for (auto& t : *t1) if (t.key() == "VectorSource::start") { CHECK(t.pos() % n == 0); starts++; }
    CHECK(starts == repeats);
    default_stream_size() = DEFAULT_STREAM_SIZE;
}

static void ring_wraps_and_windows_are_contiguous() {
    auto [w, r] = new_stream<uint32_t>(4096 * 3, Memory::Host);     // 3072 samples
    uint32_t next_w = 0, next_r = 0;
    for (int round = 0; round < 200; round++) {
        {
            auto o = w.write_buf();
            const size_t n = std::min<size_t>(o.len(), 700 + (size_t)round % 13);
            for (size_t i = 0; i < n; i++) o.slice()[i] = next_w++;
            o.produce(n, {Tag(0, "k", (uint64_t)round)});
        }
        auto [in, tags] = r.read_buf();
        CHECK(in.len() <= 3072);
        for (size_t i = 0; i < in.len(); i++) if (in.slice()[i] != next_r + i) { CHECK(false); break; }
        const size_t c = std::min<size_t>(in.len(), 650);
        in.consume(c);
        next_r += (uint32_t)c;
    }
    CHECK(next_r > 100000);
    // closing: the reader sees eof only once the ring is drained
    CHECK(!r.eof());
    { auto gone = std::move(w); }
    CHECK(r.wait_handle().closed());
    auto [in, tags] = r.read_buf();
    CHECK(!r.eof() || in.len() == 0);
    in.consume(in.len());
    CHECK(r.eof());
    CHECK(r.wait_handle().wait(1));                                 // never: writer gone, nothing left
}

int main() {
    ring_wraps_and_windows_are_contiguous();
    for (size_t ring : {(size_t)4096, (size_t)40000, DEFAULT_STREAM_SIZE}) {
        tee_graph<Graph>(ring, 100003, 3);
        tee_graph<MTGraph>(ring, 100003, 3);
    }
    printf(g_fail ? "FAILED (%d)\n" : "OK\n", g_fail);
    return g_fail ? 1 : 0;
}
