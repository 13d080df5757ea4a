// resident.hpp — the C++ twin of the Rust shim's device-resident graph (rust/src/lib.rs: new_gpu_stream,
// GpuWriteStream / GpuReadStream, GpuUpload, GpuDownload, GpuResident), type for type and branch for branch, so that the
// design the Rust source uses is the design that is compiled, run and tested here (tests/cpp/test_resident_graph.cpp
// drives it to termination under Graph::run and under MTGraph).  The Rust file cannot be compiled in this image.
//
// The reference's stream is ONE ring behind TWO handles (src/stream.rs:187-190,256-258): an end is closed() when the other
// handle has been dropped (:148-150,166-168), ReadStream::eof() = writer dropped and ring empty (:237-246), wait(need) is
// true when `need` can never be met (:222-224,311-313).  Graph::run (src/graph.rs:126-147) and MTGraph
// (src/mtgraph.rs:98-116) finish a block on those facts; rr_dstream_close / _closed / _wait carry them for an HBM ring.
#pragma once
#include "rustradio.hpp"

namespace rustradio {

namespace detail {
// the ring both ends share (the reference's Arc<Buffer<T>>); destroyed with the last handle.
//
// Tags are the ring's host-side side-band, as in the reference's Buffer (circular_buffer.rs:518-557: produce() files the tags
// of the samples it publishes, consume() drops those of the samples it retires).  Positions are kept counted from the start
// of the stream, with the side-band's OWN totals: the writing block posts (n samples + their tags) and the reading block
// takes (the tags of the n samples it consumed), each from its own thread.  rr_block_work_streams publishes a block's
// output inside the library BEFORE the block can post the tags, so a reader may hold samples whose tags are a moment away:
// take() waits until the side-band covers what was consumed (the writer posts right after its call returns, or is gone).
struct GpuRing {
    rr_dstream* s;
    std::mutex m;
    std::condition_variable cv;
    uint64_t posted = 0, taken = 0;          // samples whose tags are filed / have been handed on
    bool writer_gone = false;
    std::vector<std::pair<uint64_t, Tag>> tags;
    explicit GpuRing(rr_dstream* p) : s(p) {}
    ~GpuRing() { rr_dstream_destroy(s); }
    GpuRing(const GpuRing&) = delete;
    // writer side: the next n samples of the stream carry `t` (positions relative to the first of them)
    void post(size_t n, const std::vector<Tag>& t) {
        { std::lock_guard<std::mutex> g(m);
          for (auto& x : t) if (x.pos() < n) tags.emplace_back(posted + x.pos(), x);
          posted += n; }
        cv.notify_all();
    }
    // reader side: the tags of the next n samples (positions relative to the first of them); the samples are retired
    std::vector<Tag> take(size_t n) {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return posted >= taken + n || writer_gone; });
        std::vector<Tag> out;
        std::vector<std::pair<uint64_t, Tag>> keep;
        for (auto& pt : tags) {
            if (pt.first < taken + n) out.emplace_back((size_t)(pt.first - taken), pt.second.key(), pt.second.val());
            else keep.push_back(std::move(pt));
        }
        tags.swap(keep);
        taken += n;
        return out;
    }
    void writer_dropped() { { std::lock_guard<std::mutex> g(m); writer_gone = true; } cv.notify_all(); }
};
inline bool ring_wait(rr_dstream* s, int side, size_t need) {
    int never = 0;
    rr_dstream_wait(s, side, need, WAIT_SLICE_MS, &never);
    return never != 0;
}
inline void check(int rc) { if (rc == RR_ERR) throw Error(rr_last_error()); }
}  // namespace detail

// Writing end of an HBM ring (WriteStream<T>, src/stream.rs:256-313).
template <class T> class GpuWriteStream : public StreamWait {
    std::shared_ptr<detail::GpuRing> ring_;
public:
    GpuWriteStream() = default;
    explicit GpuWriteStream(std::shared_ptr<detail::GpuRing> r) : ring_(std::move(r)) {}
    ~GpuWriteStream() override { drop(); }                                                          // Drop
    GpuWriteStream(GpuWriteStream&& o) noexcept : ring_(std::move(o.ring_)) {}
    GpuWriteStream& operator=(GpuWriteStream&& o) noexcept {
        if (this != &o) { drop(); ring_ = std::move(o.ring_); }
        return *this;
    }
    GpuWriteStream(const GpuWriteStream&) = delete;
    rr_dstream* raw() const { return ring_->s; }
    // BufferWriter::produce's tag half (circular_buffer.rs:518-557): n samples went (or are about to go) into the ring with these tags
    void post_tags(size_t n, const std::vector<Tag>& tags) const { if (n) ring_->post(n, tags); }
    size_t capacity() const { return rr_dstream_capacity(ring_->s); }
    size_t free() const { return rr_dstream_write_buf(ring_->s, nullptr, nullptr); }              // WriteStream::free, :274-276
    size_t id() const override { return rr_dstream_id(ring_->s); }
    bool wait(size_t need) const override { return detail::ring_wait(ring_->s, RR_SIDE_WRITER, need); }
    bool closed() const override { return rr_dstream_closed(ring_->s, RR_SIDE_READER) != 0; }
private:
    void drop() { if (ring_) { rr_dstream_close(ring_->s, RR_SIDE_WRITER); ring_->writer_dropped(); } }
};

// Reading end of an HBM ring (ReadStream<T>, src/stream.rs:187-246).
template <class T> class GpuReadStream : public StreamWait {
    std::shared_ptr<detail::GpuRing> ring_;
public:
    GpuReadStream() = default;
    explicit GpuReadStream(std::shared_ptr<detail::GpuRing> r) : ring_(std::move(r)) {}
    ~GpuReadStream() override { if (ring_) rr_dstream_close(ring_->s, RR_SIDE_READER); }          // Drop
    GpuReadStream(GpuReadStream&& o) noexcept : ring_(std::move(o.ring_)) {}
    GpuReadStream& operator=(GpuReadStream&& o) noexcept {
        if (this != &o) { if (ring_) rr_dstream_close(ring_->s, RR_SIDE_READER); ring_ = std::move(o.ring_); }
        return *this;
    }
    GpuReadStream(const GpuReadStream&) = delete;
    rr_dstream* raw() const { return ring_->s; }
    size_t capacity() const { return rr_dstream_capacity(ring_->s); }
    size_t readable() const { return rr_dstream_read_buf(ring_->s, nullptr); }
    // BufferReader::consume's tag half (circular_buffer.rs:472-513): the tags of the n samples just consumed, relative to the first
    std::vector<Tag> take_tags(size_t n) const { return n ? ring_->take(n) : std::vector<Tag>{}; }
    // the flag first: once the writer is closed readable() can only fall (:237-246)
    bool eof() const { return rr_dstream_closed(ring_->s, RR_SIDE_WRITER) != 0 && readable() == 0; }
    size_t id() const override { return rr_dstream_id(ring_->s); }
    bool wait(size_t need) const override { return detail::ring_wait(ring_->s, RR_SIDE_READER, need); }
    bool closed() const override { return rr_dstream_closed(ring_->s, RR_SIDE_WRITER) != 0; }
};

// new_stream() in HBM (src/stream.rs:336-339)
template <class T> std::pair<GpuWriteStream<T>, GpuReadStream<T>> new_gpu_stream(size_t capacity_bytes) {
    rr_dstream* s = rr_dstream_create(sizeof(T), capacity_bytes);
    if (!s) throw Error(rr_last_error());
    auto ring = std::make_shared<detail::GpuRing>(s);
    return {GpuWriteStream<T>(ring), GpuReadStream<T>(ring)};
}

// Host ring -> HBM ring (graph edge).
template <class T> class GpuUpload : public Block {
    ReadStream<T> src_;
    GpuWriteStream<T> dst_;
public:
    GpuUpload(ReadStream<T> src, GpuWriteStream<T> dst) : src_(std::move(src)), dst_(std::move(dst)) {}
    static std::pair<std::unique_ptr<GpuUpload<T>>, GpuReadStream<T>> new_(ReadStream<T> src, size_t capacity_bytes) {
        auto [w, r] = new_gpu_stream<T>(capacity_bytes);
        return {std::make_unique<GpuUpload<T>>(std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return "GpuUpload"; }
    bool eof() override { return src_.eof(); }
    BlockRet work() override {
        auto [input, tags] = src_.read_buf();
        const size_t have = input.len();
        const size_t n = std::min(have, dst_.free());
        if (n == 0)   // starved -> wait on the source; ring full -> wait on the ring (its reader dropping ends this block too)
            return have == 0 ? BlockRet::wait(src_.wait_handle(), 1) : BlockRet::wait(dst_, 1);
        detail::check(rr_dstream_copy_in(dst_.raw(), 0, input.slice(), n, nullptr));
        dst_.post_tags(n, tags);                  // tags move with their samples (those with pos < n; the rest stay in the host ring)
        detail::check(rr_dstream_produce(dst_.raw(), n));
        input.consume(n);
        return BlockRet::again();
    }
};

// HBM ring -> host ring (graph edge).
template <class T> class GpuDownload : public Block {
    GpuReadStream<T> src_;
    WriteStream<T> dst_;
public:
    GpuDownload(GpuReadStream<T> src, WriteStream<T> dst) : src_(std::move(src)), dst_(std::move(dst)) {}
    static std::pair<std::unique_ptr<GpuDownload<T>>, ReadStream<T>> new_(GpuReadStream<T> src) {
        auto [w, r] = new_stream<T>(default_stream_size(), Memory::Host);
        return {std::make_unique<GpuDownload<T>>(std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return "GpuDownload"; }
    bool eof() override { return src_.eof(); }
    BlockRet work() override {
        auto out = dst_.write_buf();
        const size_t have = src_.readable();
        const size_t n = std::min(out.len(), have);
        if (n == 0) return have == 0 ? BlockRet::wait(src_, 1) : BlockRet::wait(dst_.wait_handle(), 1);
        detail::check(rr_dstream_copy_out(src_.raw(), 0, out.slice(), n, nullptr));
        detail::check(rr_dstream_consume(src_.raw(), n));
        out.produce(n, src_.take_tags(n));
        return BlockRet::again();
    }
};

// One GPU block between two HBM rings (rr_block_work_streams).  The status is the block's own — WaitForStream(src, need)
// when starved, WaitForStream(dst, need) when the output ring is full — so both runners end it the reference's way:
// upstream dropped AND ring drained AND (resampler) no pending sample.
//
// Tags: re-based from consumed / produced by the reference rule of the block behind the handle (rr_block_tag_rule) — the
// host-window blocks' own code (FirFilter `pos / deci`, FftFilter's pending list, Hilbert `pos < n`) in one place.
template <class I, class O> class GpuResident : public Block {
    detail::Handle h_;
    const char* name_;
    GpuReadStream<I> src_;
    GpuWriteStream<O> dst_;
    detail::TagForwarder fwd_;
public:
    GpuResident(rr_block* created, const char* name, GpuReadStream<I> src, GpuWriteStream<O> dst)
        : h_(created), name_(name), src_(std::move(src)), dst_(std::move(dst)), fwd_(h_.h) {}
    static std::pair<std::unique_ptr<GpuResident<I, O>>, GpuReadStream<O>> new_(rr_block* created, const char* name, GpuReadStream<I> src,
                                                                               size_t out_capacity_bytes) {
        detail::Handle guard(created);                    // throws rr_last_error() on a failed create; freed if the ring cannot be made
        auto [w, r] = new_gpu_stream<O>(out_capacity_bytes);
        rr_block* h = guard.h;
        guard.h = nullptr;
        return {std::make_unique<GpuResident<I, O>>(h, name, std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return name_; }
    bool eof() override { return rr_block_eof(h_.h, src_.eof() ? 1 : 0) != 0; }    // src/rational_resampler.rs:209-213 inside
    BlockRet work() override {
        size_t c = 0, p = 0, need = 0;
        const int st = rr_block_work_streams(h_.h, src_.raw(), dst_.raw(), &c, &p, &need, nullptr);
        detail::check(st);
        detail::activity() += c + p;
        dst_.post_tags(p, fwd_.step(src_.take_tags(c), c, p));
        switch (st) {
        case RR_WAIT_SRC: return BlockRet::wait(src_, need);
        case RR_WAIT_DST: return BlockRet::wait(dst_, need);
        case RR_EOF: return BlockRet::eof();
        default: return BlockRet::again();
        }
    }
};

}  // namespace rustradio
