// rustradio.hpp — C++17 host-side mirror of rustradio's operator API for the hot path,
// layered on the C ABI (include/rustradio_amd.h).  The reference is Rust; this image has no
// Rust toolchain, so the host side above the C ABI is C++ with the same names, argument
// meaning and error behaviour as the reference:
//
//   Block / BlockRet / BlockName / BlockEOF      src/block.rs:12-126
//   ReadStream / WriteStream / Tag / new_stream  src/stream.rs:48-93, 105, 187-339
//   VectorSource / VectorSink / NullSink / Graph  src/vector_source.rs, src/vector_sink.rs,
//                                                 src/null_sink.rs, src/graph.rs:99-173
//   FirFilter (+builder .deci() .translate()), Fir::filter_n   src/fir.rs:150-198, 303-551
//   FftFilter / FftFilterFloat                    src/fft_filter.rs:210-491
//   RationalResampler (+builder)                  src/rational_resampler.rs:19-213
//   QuadratureDemod                               src/quadrature_demod.rs:32-114
//   Hilbert                                       src/hilbert.rs:22-129
//   fir::low_pass / low_pass_complex / hilbert, window::WindowType   src/fir.rs:594-680, src/window.rs
//
// A block's work() does exactly what the Rust shim in INTEGRATION.md does: take the stream
// windows, hand plain pointers to rr_block_work(), then consume()/produce() what the GPU
// block reported, re-basing tags the way the reference block does.  No sample arithmetic
// happens in this file.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <variant>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include "../../include/rustradio_amd.h"

namespace rustradio {

using Float = float;                    // src/lib.rs:268
using Complex = std::complex<float>;    // src/lib.rs:271 (interleaved re, im)

struct Error : std::runtime_error {     // src/lib.rs:276-311
    using std::runtime_error::runtime_error;
    static Error msg(const std::string& m) { return Error(m); }
};

constexpr size_t DEFAULT_STREAM_SIZE = 4'096'000;   // bytes, src/stream.rs:105

// ---- tags (src/stream.rs:48-93) ----------------------------------------------------------------
using TagValue = std::variant<bool, uint64_t, float, std::string>;
struct Tag {
    size_t pos_;
    std::string key_;
    TagValue val_;
    Tag(size_t p, std::string k, TagValue v) : pos_(p), key_(std::move(k)), val_(std::move(v)) {}
    size_t pos() const { return pos_; }
    void set_pos(size_t p) { pos_ = p; }
    const std::string& key() const { return key_; }
    const TagValue& val() const { return val_; }
    bool operator==(const Tag& o) const { return pos_ == o.pos_ && key_ == o.key_ && val_ == o.val_; }
};

namespace detail {
inline uint64_t& activity() { static thread_local uint64_t a = 0; return a; }   // samples moved (Graph idle detection)
}

// ---- streams --------------------------------------------------------------------------------------
// The reference hands out contiguous windows of a double-mapped ring shared by a ReadStream and a
// WriteStream that may live on different threads (src/nowasm/circular_buffer.rs:98-128, 572-615).
// This harness does the same: one memfd mapped twice back to back, read_buf() = ALL readable samples,
// write_buf() = ALL free space, fixed capacity; counters and tags under one lock, windows stable while
// the other end works, a condition variable behind StreamWait::wait (src/stream.rs:114-138).
struct StreamWait {
    virtual ~StreamWait() = default;
    virtual size_t id() const = 0;
    // true = `need` will never be satisfied: go ahead and EOF (stream.rs:121-126)
    virtual bool wait(size_t need) const = 0;
    virtual bool closed() const = 0;
};

// Where new_stream() places rings.  Memory::Device = an HBM-resident ring (rr_dstream): GPU
// blocks chained through such streams run rr_block_work_dev on the windows — no PCIe hop and no
// host/device synchronisation per work(); only sources (fill_from_slice) and sinks (copy_to) cross
// the bus.  Tags stay a host-side side-band in both cases.  Set once before building a graph.
enum class Memory { Host, Device };
inline Memory& default_memory() { static thread_local Memory m = Memory::Host; return m; }

namespace detail {
// How long one StreamWait::wait sleeps at most before it reports back (the reference's condvar wait has no
// timeout because every produce / consume / drop notifies; so does this one — the timeout only bounds a lost race).
constexpr unsigned WAIT_SLICE_MS = 100;
// One physical range mapped at two consecutive virtual ranges (circular_buffer.rs:98-128).
struct DoubleMap {
    unsigned char* base = nullptr;
    size_t phys = 0;
    explicit DoubleMap(size_t bytes) {
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        phys = std::max(page, (bytes + page - 1) / page * page);
        const int fd = memfd_create("rustradio_ring", 0);
        if (fd < 0 || ftruncate(fd, (off_t)phys) != 0) { if (fd >= 0) close(fd); throw Error("stream ring: memfd_create failed"); }
        void* va = mmap(nullptr, 2 * phys, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        bool ok = va != MAP_FAILED;
        ok = ok && mmap(va, phys, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) != MAP_FAILED;
        ok = ok && mmap((unsigned char*)va + phys, phys, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) != MAP_FAILED;
        close(fd);
        if (!ok) { if (va != MAP_FAILED) munmap(va, 2 * phys); throw Error("stream ring: mmap failed"); }
        base = static_cast<unsigned char*>(va);
    }
    ~DoubleMap() { if (base) munmap(base, 2 * phys); }
    DoubleMap(const DoubleMap&) = delete;
};
}  // namespace detail

template <class T> struct StreamState {
    std::unique_ptr<detail::DoubleMap> ring;   // host ring: bytes [rb, rb + used * sizeof(T)) readable
    size_t rb = 0, used_ = 0;
    rr_dstream* ds = nullptr;     // device ring (then `ring` is unused; its counters live in the library, under its own lock)
    size_t cap;
    std::vector<Tag> tags;        // positions relative to the read window
    bool writer_alive = true, reader_alive = true;
    size_t id_;
    mutable std::mutex m;         // rb, used_, tags, the two flags
    mutable std::condition_variable cv;
    // the two ends as StreamWait objects: BlockRet::WaitForStream carries one of them (block.rs:61)
    struct Side : StreamWait {
        const StreamState* st; bool reader;
        Side(const StreamState* s, bool r) : st(s), reader(r) {}
        size_t id() const override { return st->id_; }
        bool closed() const override { return st->closed(); }
        bool wait(size_t need) const override { return st->wait(reader, need); }
    } rside{this, true}, wside{this, false};
    explicit StreamState(size_t bytes, Memory m_ = default_memory()) : cap(bytes / sizeof(T)) {
        static std::atomic<size_t> next_id{1};
        id_ = next_id++;
        if (cap == 0) throw Error("stream: capacity smaller than one sample");
        if (m_ == Memory::Device) {
            ds = rr_dstream_create(sizeof(T), bytes);
            if (!ds) throw Error(rr_last_error());
        } else {
            ring = std::make_unique<detail::DoubleMap>(cap * sizeof(T));
            // page-lock the whole double mapping once: GPU blocks then read and write the windows of this
            // ring in place over PCIe (zero copy).  Without a GPU the call fails and the ring is an ordinary host ring.
            registered = rr_host_register(ring->base, 2 * ring->phys) == 0;
        }
    }
    ~StreamState() {
        if (ds) rr_dstream_destroy(ds);
        if (registered) rr_host_unregister(ring->base);
    }
    bool registered = false;
    StreamState(const StreamState&) = delete;
    bool device() const { return ds != nullptr; }
    size_t used() const {
        if (ds) return rr_dstream_read_buf(ds, nullptr);
        std::lock_guard<std::mutex> g(m);
        return used_;
    }
    size_t free() const { return cap - used(); }
    bool closed() const {                                    // src/stream.rs:148-150,166-168: the other end is gone
        std::lock_guard<std::mutex> g(m);
        return !writer_alive || !reader_alive;
    }
    void drop_end(bool reader) {
        { std::lock_guard<std::mutex> g(m); (reader ? reader_alive : writer_alive) = false; }
        if (ds) rr_dstream_close(ds, reader ? RR_SIDE_READER : RR_SIDE_WRITER);
        cv.notify_all();
    }
    // wait_for_read / wait_for_write (stream.rs:222-224,311-313): what is there < need AND the other end dropped
    bool wait(bool reader, size_t need) const {
        if (ds) {
            int never = 0;
            rr_dstream_wait(ds, reader ? RR_SIDE_READER : RR_SIDE_WRITER, need, detail::WAIT_SLICE_MS, &never);
            return never != 0;
        }
        std::unique_lock<std::mutex> g(m);
        auto have = [&] { return reader ? used_ : cap - used_; };
        auto other_gone = [&] { return reader ? !writer_alive : !reader_alive; };
        cv.wait_for(g, std::chrono::milliseconds(detail::WAIT_SLICE_MS), [&] { return have() >= need || other_gone(); });
        return have() < need && other_gone();
    }
};

template <class T> class BufferReader {   // circular_buffer.rs:233-251
    std::shared_ptr<StreamState<T>> s_;
    const T* ptr_ = nullptr;      // the window as it was when read_buf() was called (host or DEVICE pointer)
    size_t len_ = 0;
public:
    explicit BufferReader(std::shared_ptr<StreamState<T>> s) : s_(std::move(s)) {
        if (s_->ds) {
            const void* p = nullptr;
            len_ = rr_dstream_read_buf(s_->ds, &p);
            ptr_ = static_cast<const T*>(p);
        } else {
            std::lock_guard<std::mutex> g(s_->m);
            ptr_ = reinterpret_cast<const T*>(s_->ring->base + s_->rb);
            len_ = s_->used_;
        }
    }
    bool device() const { return s_->device(); }
    const T* slice() const { return ptr_; }
    size_t len() const { return len_; }
    bool is_empty() const { return len() == 0; }
    const T* begin() const { host_only(); return slice(); }
    const T* end() const { host_only(); return slice() + len(); }
    // first n samples of the window into host memory (any placement)
    rr_dstream* dstream() const { return s_->ds; }
    void copy_to(T* host, size_t n) const {
        if (n > len()) throw Error("copy_to: n > readable");
        if (!s_->ds) { if (n) std::memcpy(host, slice(), n * sizeof(T)); return; }
        if (rr_dstream_copy_out(s_->ds, 0, host, n, nullptr) != 0) throw Error(rr_last_error());
    }
    void consume(size_t n) {               // circular_buffer.rs:472-513
        if (n > len_) throw Error("consume: n > used");
        detail::activity() += n;
        {
            std::lock_guard<std::mutex> g(s_->m);
            std::vector<Tag> keep;
            for (auto& t : s_->tags)
                if (t.pos() >= n) keep.emplace_back(t.pos() - n, t.key(), t.val());
            s_->tags.swap(keep);
            if (!s_->ds) {
                s_->rb = (s_->rb + n * sizeof(T)) % s_->ring->phys;
                s_->used_ -= n;
            } else if (rr_dstream_consume(s_->ds, n) != 0) {   // under m, like produce: tags and the ring's counters move together
                throw Error(rr_last_error());
            }
        }
        ptr_ += n; len_ -= n;
        if (n) s_->cv.notify_all();
    }
private:
    void host_only() const { if (s_->ds) throw Error("host iteration over a device-resident stream: use copy_to()"); }
};

template <class T> class BufferWriter {   // circular_buffer.rs:284-310
    std::shared_ptr<StreamState<T>> s_;
    T* ptr_ = nullptr;            // the write window as it was when write_buf() was called (host or DEVICE pointer)
    size_t len_ = 0;
public:
    explicit BufferWriter(std::shared_ptr<StreamState<T>> s) : s_(std::move(s)) {
        if (s_->ds) {
            void* p = nullptr;
            len_ = rr_dstream_write_buf(s_->ds, &p, nullptr);
            ptr_ = static_cast<T*>(p);
            return;
        }
        std::lock_guard<std::mutex> g(s_->m);
        ptr_ = reinterpret_cast<T*>(s_->ring->base + (s_->rb + s_->used_ * sizeof(T)) % s_->ring->phys);
        len_ = s_->cap - s_->used_;
    }
    BufferWriter(BufferWriter&& o) noexcept : s_(std::move(o.s_)), ptr_(o.ptr_), len_(o.len_) { o.s_.reset(); }
    BufferWriter(const BufferWriter&) = delete;
    bool device() const { return s_->device(); }
    T* slice() { return ptr_; }                                         // host or DEVICE pointer
    size_t len() const { return len_; }
    bool is_empty() const { return len() == 0; }
    // n samples of a DEVICE-resident read window into this DEVICE-resident write window: no host hop (rr_dstream_copy)
    void fill_from_device(const BufferReader<T>& src, size_t n) {
        if (n > len()) throw Error("fill_from_device: n > free");
        if (!s_->ds || !src.dstream()) throw Error("fill_from_device: both streams must be device-resident");
        if (rr_dstream_copy(s_->ds, 0, src.dstream(), 0, n, nullptr) != 0) throw Error(rr_last_error());
    }
    void fill_from_slice(const T* src, size_t n) {                      // src = host memory
        if (n > len()) throw Error("fill_from_slice: n > free");
        if (!s_->ds) { if (n) std::memcpy(slice(), src, n * sizeof(T)); return; }
        if (rr_dstream_copy_in(s_->ds, 0, src, n, nullptr) != 0) throw Error(rr_last_error());
    }
    void produce(size_t n, const std::vector<Tag>& tags) {   // circular_buffer.rs:518-557
        if (n > len()) throw Error("produce: n > free");
        if (n == 0) return;                                  // tags dropped (:528-533)
        detail::activity() += n;
        {
            // tag positions are relative to the READ window as it is now (the reader may have consumed since write_buf())
            std::lock_guard<std::mutex> g(s_->m);
            const size_t base = s_->ds ? rr_dstream_read_buf(s_->ds, nullptr) : s_->used_;
            for (auto& t : tags) s_->tags.emplace_back(t.pos() + base, t.key(), t.val());
            if (!s_->ds) s_->used_ += n;
            else if (rr_dstream_produce(s_->ds, n) != 0) throw Error(rr_last_error());
        }
        ptr_ += n; len_ -= n;
        s_->cv.notify_all();
    }
};

template <class T> class ReadStream {      // src/stream.rs:187-246
    std::shared_ptr<StreamState<T>> s_;
public:
    ReadStream() = default;
    explicit ReadStream(std::shared_ptr<StreamState<T>> s) : s_(std::move(s)) {}
    ~ReadStream() { if (s_) s_->drop_end(true); }
    ReadStream(ReadStream&&) = default;
    ReadStream& operator=(ReadStream&&) = default;
    ReadStream(const ReadStream&) = delete;
    static ReadStream from_slice(const T* d, size_t n) {    // src/stream.rs:187-195 (test helper)
        auto st = std::make_shared<StreamState<T>>(DEFAULT_STREAM_SIZE, Memory::Host);
        if (n > st->cap) throw Error("from_slice: more samples than the ring holds");
        if (n) std::memcpy(st->ring->base, d, n * sizeof(T));
        st->used_ = n;
        st->writer_alive = false;
        return ReadStream(st);
    }
    std::pair<BufferReader<T>, std::vector<Tag>> read_buf() const {   // :208-217
        BufferReader<T> r(s_);
        std::lock_guard<std::mutex> g(s_->m);
        std::vector<Tag> t;
        for (auto& x : s_->tags) if (x.pos() < r.len()) t.push_back(x);
        return {std::move(r), std::move(t)};
    }
    bool eof() const {                                                 // :237-246: writer gone and nothing left
        { std::lock_guard<std::mutex> g(s_->m); if (s_->writer_alive) return false; }
        return s_->used() == 0;
    }
    const StreamWait& wait_handle() const { return s_->rside; }
    size_t id() const { return s_->id_; }
};

template <class T> class WriteStream {     // src/stream.rs:288-310
    std::shared_ptr<StreamState<T>> s_;
public:
    WriteStream() = default;
    explicit WriteStream(std::shared_ptr<StreamState<T>> s) : s_(std::move(s)) {}
    ~WriteStream() { if (s_) s_->drop_end(false); }
    WriteStream(WriteStream&&) = default;
    WriteStream& operator=(WriteStream&&) = default;
    WriteStream(const WriteStream&) = delete;
    BufferWriter<T> write_buf() const { return BufferWriter<T>(s_); }  // :301-310
    size_t free() const { return s_->free(); }
    const StreamWait& wait_handle() const { return s_->wside; }
    size_t id() const { return s_->id_; }
};

// Ring size used by new_stream() when none is given.  The reference's 4,096,000 bytes suit CPU caches; rings
// in HBM want hundreds of MB so that one work() is milliseconds, not microseconds, of kernel.
inline size_t& default_stream_size() { static thread_local size_t b = DEFAULT_STREAM_SIZE; return b; }

template <class T> std::pair<WriteStream<T>, ReadStream<T>> new_stream(size_t bytes = default_stream_size(),
                                                                       Memory m = default_memory()) {  // :336-339
    auto st = std::make_shared<StreamState<T>>(bytes, m);
    return {WriteStream<T>(st), ReadStream<T>(st)};
}

// ---- Block (src/block.rs) -----------------------------------------------------------------------------
struct BlockRet {
    enum Kind { Again, Pending, WaitForStream, EOF_ } kind;
    const StreamWait* stream = nullptr;
    size_t need = 0;
    static BlockRet again() { return {Again, nullptr, 0}; }
    static BlockRet eof() { return {EOF_, nullptr, 0}; }
    static BlockRet wait(const StreamWait& s, size_t n) { return {WaitForStream, &s, n}; }
};

struct Block {
    virtual ~Block() = default;
    virtual const char* block_name() const = 0;   // BlockName, :91-97
    virtual bool eof() = 0;                       // BlockEOF, :103-110
    virtual BlockRet work() = 0;                  // Block::work, :115-126 (throws Error on failure)
};

// ---- window / tap designers (src/window.rs, src/fir.rs:594-680) -----------------------------------------
namespace window {
struct WindowType {
    int kind; float parm;
    static WindowType Hamming() { return {RR_WIN_HAMMING, 0}; }
    static WindowType Blackman() { return {RR_WIN_BLACKMAN, 0}; }
    static WindowType BlackmanHarris() { return {RR_WIN_BLACKMAN_HARRIS, 0}; }
    static WindowType HammingParm(float a0) { return {RR_WIN_HAMMING_PARM, a0}; }
    float max_attenuation() const { return rr_max_attenuation(kind); }
    std::vector<Float> make_window(size_t ntaps) const {
        std::vector<Float> w(ntaps);
        if (rr_make_window(kind, parm, ntaps, w.data()) != 0) throw Error(rr_last_error());
        return w;
    }
};
}  // namespace window

namespace fir {
inline std::vector<Float> low_pass(Float samp_rate, Float cutoff, Float twidth, const window::WindowType& w) {
    const size_t n = rr_low_pass(samp_rate, cutoff, twidth, w.kind, w.parm, nullptr, 0);
    if (n == 0) throw Error(rr_last_error());       // the reference asserts (fir.rs:623-625)
    std::vector<Float> t(n);
    rr_low_pass(samp_rate, cutoff, twidth, w.kind, w.parm, t.data(), n);
    return t;
}
inline std::vector<Complex> low_pass_complex(Float samp_rate, Float cutoff, Float twidth, const window::WindowType& w) {
    auto t = low_pass(samp_rate, cutoff, twidth, w);
    return std::vector<Complex>(t.begin(), t.end());   // Complex::new(t, 0.0), fir.rs:602
}
inline std::vector<Float> hilbert(const std::vector<Float>& window) {
    std::vector<Float> t(window.size());
    if (rr_hilbert_taps(window.data(), window.size(), t.data()) != 0) throw Error(rr_last_error());
    return t;
}
}  // namespace fir

// ---- GPU block plumbing -----------------------------------------------------------------------------------
namespace detail {
struct Handle {
    rr_block* h;
    explicit Handle(rr_block* p) : h(p) { if (!h) throw Error(rr_last_error()); }
    ~Handle() { rr_block_destroy(h); }
    Handle(const Handle&) = delete;
};
struct WorkOut { int st; size_t consumed, produced, need; };
inline WorkOut work(rr_block* h, const void* in, size_t in_len, void* out, size_t out_cap) {
    WorkOut w{};
    w.st = rr_block_work(h, in, in_len, out, out_cap, &w.consumed, &w.produced, &w.need);
    if (w.st == RR_ERR) throw Error(rr_last_error());
    return w;
}
// Block::work() over a read and a write window of the same placement: host windows go through
// rr_block_work (PCIe round trip inside), device windows through rr_block_work_dev on the default stream.
template <class I, class O> inline WorkOut work(rr_block* h, const BufferReader<I>& in, BufferWriter<O>& out) {
    if (in.device() != out.device())
        throw Error("a GPU block needs both streams in the same memory: put a MemCopy block in between");
    if (!in.device()) return work(h, in.slice(), in.len(), out.slice(), out.len());
    WorkOut w{};
    w.st = rr_block_work_dev(h, in.slice(), in.len(), out.slice(), out.len(), &w.consumed, &w.produced, &w.need, nullptr);
    if (w.st == RR_ERR) throw Error(rr_last_error());
    return w;
}
static_assert(sizeof(Complex) == sizeof(rr_c32), "Complex<f32> must be interleaved re, im");

// Tags through a block that is known only by its handle (Fused, GpuResident): the reference rule of the block(s) behind the
// handle, as rr_block_tag_rule states it in whole-stream positions.  step() takes the tags of the read window (positions
// relative to it; only those on consumed samples travel, the rest stay in the stream) and returns the tags of the
// `produced` outputs of this call (positions relative to them).  A tag whose output sample does not exist yet waits here,
// as it does in FftFilter's `self.tags` (fft_filter.rs:309-313).
class TagForwarder {
    int rule_ = RR_TAGS_DROP;
    size_t param_ = 1;
    std::vector<std::pair<uint64_t, Tag>> pending_;   // (output position counted from the start of the stream, tag)
    uint64_t in_abs_ = 0, out_abs_ = 0;
public:
    explicit TagForwarder(const rr_block* h) {
        size_t p = 1;
        rule_ = rr_block_tag_rule(h, &p);
        if (rule_ == RR_ERR) throw Error(rr_last_error());
        param_ = p ? p : 1;
    }
    bool drops() const { return rule_ == RR_TAGS_DROP; }
    std::vector<Tag> step(const std::vector<Tag>& tags, size_t consumed, size_t produced) {
        std::vector<Tag> emit;
        if (rule_ == RR_TAGS_FORWARD) {
            for (auto& t : tags)
                if (t.pos() < consumed) pending_.emplace_back((in_abs_ + t.pos()) / param_, t);
            std::vector<std::pair<uint64_t, Tag>> keep;
            for (auto& pt : pending_) {
                if (pt.first < out_abs_ + produced) emit.emplace_back((size_t)(pt.first - out_abs_), pt.second.key(), pt.second.val());
                else keep.push_back(std::move(pt));
            }
            pending_.swap(keep);
        } else if (rule_ == RR_TAGS_FRAMES) {             // fft_stream.rs:98-111 (produced is a whole number of frames)
            for (size_t pos = 0; pos + param_ <= produced; pos += param_) {
                emit.emplace_back(pos, "FftStream::size", (uint64_t)param_);
                emit.emplace_back(pos, "FftStream::frame", true);
                emit.emplace_back(pos + param_ - 1, "FftStream::frame", false);
            }
        }
        in_abs_ += consumed;
        out_abs_ += produced;
        return emit;
    }
};
}  // namespace detail

// ---- FirFilter (src/fir.rs) ---------------------------------------------------------------------------------
template <class T> class FirFilter;

template <class T> class FirFilterBuilder {      // fir.rs:303-340, 476-486
    std::vector<T> taps_;
    size_t deci_ = 1;
    bool translate_ = false;
    Float samp_rate_ = 0, freq_ = 0;
    friend class FirFilter<T>;
public:
    explicit FirFilterBuilder(std::vector<T> taps) : taps_(std::move(taps)) {}
    FirFilterBuilder& deci(size_t d) { if (d == 0) throw Error("FirFilter: deci 0"); deci_ = d; return *this; }
    FirFilterBuilder& translate(Float samp_rate, Float freq) {
        static_assert(std::is_same<T, Complex>::value, "FirFilter asked to translate on non-Complex");
        translate_ = true; samp_rate_ = samp_rate; freq_ = freq; return *this;
    }
    std::pair<std::unique_ptr<FirFilter<T>>, ReadStream<T>> build(ReadStream<T> src);
};

template <class T> class FirFilter : public Block {
    detail::Handle h_;
    size_t deci_;
    ReadStream<T> src_;
    WriteStream<T> dst_;
    static rr_block* make(const std::vector<T>& taps, size_t deci, bool tr, Float fs, Float f) {
        if constexpr (std::is_same<T, Complex>::value)
            return rr_fir_c32_create(reinterpret_cast<const rr_c32*>(taps.data()), taps.size(), deci, tr, fs, f);
        else
            return rr_fir_f32_create(taps.data(), taps.size(), deci);
    }
public:
    FirFilter(ReadStream<T> src, WriteStream<T> dst, const FirFilterBuilder<T>& b)
        : h_(make(b.taps_, b.deci_, b.translate_, b.samp_rate_, b.freq_)), deci_(b.deci_), src_(std::move(src)), dst_(std::move(dst)) {}
    static FirFilterBuilder<T> builder(std::vector<T> taps) { return FirFilterBuilder<T>(std::move(taps)); }
    static std::pair<std::unique_ptr<FirFilter<T>>, ReadStream<T>> new_(ReadStream<T> src, std::vector<T> taps) {
        return FirFilterBuilder<T>(std::move(taps)).build(std::move(src));
    }
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {                    // fir.rs:492-550
        auto [input, tags] = src_.read_buf();
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        if (w.st == RR_WAIT_SRC) { out.produce(0, {}); return BlockRet::wait(src_.wait_handle(), w.need); }
        if (w.st == RR_WAIT_DST) { out.produce(0, {}); return BlockRet::wait(dst_.wait_handle(), w.need); }
        std::vector<Tag> keep;                    // :536-545
        for (auto& t : tags)
            if (t.pos() < w.consumed) keep.emplace_back(t.pos() / deci_, t.key(), t.val());
        input.consume(w.consumed);
        out.produce(w.produced, keep);
        return BlockRet::again();
    }
};
template <class T>
std::pair<std::unique_ptr<FirFilter<T>>, ReadStream<T>> FirFilterBuilder<T>::build(ReadStream<T> src) {
    if (taps_.empty()) throw Error("FirFilter: empty taps");
    auto [w, r] = new_stream<T>();
    return {std::make_unique<FirFilter<T>>(std::move(src), std::move(w), *this), std::move(r)};
}

// Fir<T>::filter_n (fir.rs:181-189): `filter()` across an input range; runs the same HIP kernel.
template <class T> class Fir {
    std::vector<T> taps_;
public:
    explicit Fir(std::vector<T> taps) : taps_(std::move(taps)) { if (taps_.empty()) throw Error("Fir: empty taps"); }
    std::vector<T> filter_n(const std::vector<T>& input, size_t deci) const {
        if (deci == 0) throw Error("filter_n: deci 0");
        if (input.size() < taps_.size()) throw Error("filter_n: input shorter than taps");
        const size_t n_out = (input.size() - taps_.size()) / deci + 1;
        // pad so that the block's "consume whole groups of deci" rule (fir.rs:502) covers n_out outputs
        std::vector<T> in(input);
        in.resize((n_out - 1) * deci + taps_.size() + deci - 1, T{});
        detail::Handle h(FirFilterMaker(taps_, deci));
        std::vector<T> out(n_out);
        auto w = detail::work(h.h, in.data(), in.size(), out.data(), out.size());
        if (w.st != RR_AGAIN || w.produced != n_out) throw Error("filter_n: unexpected block state");
        return out;
    }
    // filter_n_inplace (fir.rs:191-197): out.size() outputs, out[i] = filter(&input[i * deci ..])
    void filter_n_inplace(const std::vector<T>& input, size_t deci, std::vector<T>& out) const {
        if (out.empty()) return;
        if (input.size() < (out.size() - 1) * deci + taps_.size()) throw Error("filter_n_inplace: input too short");
        std::vector<T> in(input.begin(), input.begin() + (std::ptrdiff_t)((out.size() - 1) * deci + taps_.size()));
        const auto r = filter_n(in, deci);
        std::copy(r.begin(), r.begin() + (std::ptrdiff_t)out.size(), out.begin());
    }
    // filter (fir.rs:166-177): one output from the first taps.size() samples
    T filter(const std::vector<T>& input) const { return filter_n(std::vector<T>(input.begin(), input.begin() + (std::ptrdiff_t)taps_.size()), 1)[0]; }
private:
    static rr_block* FirFilterMaker(const std::vector<T>& taps, size_t deci) {
        if constexpr (std::is_same<T, Complex>::value)
            return rr_fir_c32_create(reinterpret_cast<const rr_c32*>(taps.data()), taps.size(), deci, 0, 0, 0);
        else
            return rr_fir_f32_create(taps.data(), taps.size(), deci);
    }
};

// ---- FftFilter / FftFilterFloat (src/fft_filter.rs) ---------------------------------------------------------------
template <class T> class FftFilterBase : public Block {
protected:
    detail::Handle h_;
    ReadStream<T> src_;
    WriteStream<T> dst_;
    detail::TagForwarder fwd_;                        // tags of samples still inside `buf` wait here (:309-313)
public:
    FftFilterBase(rr_block* h, ReadStream<T> src, WriteStream<T> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)), fwd_(h_.h) {}
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {                    // fft_filter.rs:290-354 / 429-490
        auto [input, tags] = src_.read_buf();
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        // a tag travels with its sample: output sample i is input sample i of the stream
        const std::vector<Tag> emit = fwd_.step(tags, w.consumed, w.produced);
        input.consume(w.consumed);
        out.produce(w.produced, emit);
        if (w.st == RR_WAIT_SRC) return BlockRet::wait(src_.wait_handle(), w.need);
        if (w.st == RR_WAIT_DST) return BlockRet::wait(dst_.wait_handle(), w.need);
        return BlockRet::again();
    }
};

class FftFilter : public FftFilterBase<Complex> {
public:
    using FftFilterBase<Complex>::FftFilterBase;
    static std::pair<std::unique_ptr<FftFilter>, ReadStream<Complex>> new_(ReadStream<Complex> src, const std::vector<Complex>& taps) {
        auto [w, r] = new_stream<Complex>();
        rr_block* h = rr_fftfilter_create(reinterpret_cast<const rr_c32*>(taps.data()), taps.size());
        return {std::make_unique<FftFilter>(h, std::move(src), std::move(w)), std::move(r)};
    }
};
class FftFilterFloat : public FftFilterBase<Float> {
public:
    using FftFilterBase<Float>::FftFilterBase;
    static std::pair<std::unique_ptr<FftFilterFloat>, ReadStream<Float>> new_(ReadStream<Float> src, const std::vector<Float>& taps) {
        auto [w, r] = new_stream<Float>();
        rr_block* h = rr_fftfilter_float_create(taps.data(), taps.size());
        return {std::make_unique<FftFilterFloat>(h, std::move(src), std::move(w)), std::move(r)};
    }
};

// ---- RationalResampler (src/rational_resampler.rs) ------------------------------------------------------------------
template <class T> class RationalResampler : public Block {
    detail::Handle h_;
    ReadStream<T> src_;
    WriteStream<T> dst_;
public:
    RationalResampler(rr_block* h, ReadStream<T> src, WriteStream<T> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)) {}
    // Err (here: throws Error) when interp or deci is 0 (:130-135)
    static std::pair<std::unique_ptr<RationalResampler<T>>, ReadStream<T>> new_(ReadStream<T> src, size_t interp, size_t deci) {
        auto [w, r] = new_stream<T>();
        rr_block* h = rr_resampler_create(interp, deci, sizeof(T));
        return {std::make_unique<RationalResampler<T>>(h, std::move(src), std::move(w)), std::move(r)};
    }
    struct Builder {                              // typestate builder (:19-92), flattened
        size_t interp_ = 0, deci_ = 0;
        Builder& interp(size_t i) { interp_ = i; return *this; }
        Builder& deci(size_t d) { deci_ = d; return *this; }
        auto build(ReadStream<T> src) { return RationalResampler<T>::new_(std::move(src), interp_, deci_); }
    };
    static Builder builder() { return Builder{}; }
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }   // pending.is_none() && src.eof() (:209-213)
    BlockRet work() override {                    // :155-206 — tags are dropped (:156)
        auto [input, tags] = src_.read_buf();
        (void)tags;
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        input.consume(w.consumed);
        out.produce(w.produced, {});
        return w.st == RR_WAIT_DST ? BlockRet::wait(dst_.wait_handle(), w.need) : BlockRet::wait(src_.wait_handle(), w.need);
    }
};

// ---- QuadratureDemod (src/quadrature_demod.rs) -------------------------------------------------------------------------
class QuadratureDemod : public Block {
    detail::Handle h_;
    ReadStream<Complex> src_;
    WriteStream<Float> dst_;
public:
    QuadratureDemod(rr_block* h, ReadStream<Complex> src, WriteStream<Float> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)) {}
    // `exact_atan2 = true` is a --no-default-features build (f32::atan2); false = fast-math feature
    static std::pair<std::unique_ptr<QuadratureDemod>, ReadStream<Float>> new_(ReadStream<Complex> src, Float gain, bool exact_atan2 = true) {
        auto [w, r] = new_stream<Float>();
        rr_block* h = rr_quaddemod_create(gain, exact_atan2 ? RR_ATAN2_EXACT : RR_ATAN2_FAST);
        return {std::make_unique<QuadratureDemod>(h, std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {                    // :46-113 — tags dropped
        auto [input, tags] = src_.read_buf();
        (void)tags;
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        input.consume(w.consumed);
        out.produce(w.produced, {});
        return w.st == RR_WAIT_DST ? BlockRet::wait(dst_.wait_handle(), w.need) : BlockRet::wait(src_.wait_handle(), w.need);
    }
};

// ---- RtlSdrDecode (src/rtlsdr_decode.rs) ----------------------------------------------------------------------------------
class RtlSdrDecode : public Block {
    detail::Handle h_;
    ReadStream<uint8_t> src_;
    WriteStream<Complex> dst_;
public:
    RtlSdrDecode(rr_block* h, ReadStream<uint8_t> src, WriteStream<Complex> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)) {}
    static std::pair<std::unique_ptr<RtlSdrDecode>, ReadStream<Complex>> new_(ReadStream<uint8_t> src) {   // #[rustradio(new)], :9-16
        auto [w, r] = new_stream<Complex>();
        return {std::make_unique<RtlSdrDecode>(rr_rtlsdr_decode_create(), std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {                    // :18-47 — tags dropped (":21 TODO: handle tags")
        auto [input, tags] = src_.read_buf();
        (void)tags;
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        input.consume(w.consumed);
        out.produce(w.produced, {});
        return w.st == RR_WAIT_DST ? BlockRet::wait(dst_.wait_handle(), w.need) : BlockRet::wait(src_.wait_handle(), w.need);
    }
};

// ---- FftStream (src/fft_stream.rs) -------------------------------------------------------------------------------------------
inline constexpr const char* TAG_FRAME = "FftStream::frame";            // :18
inline constexpr const char* TAG_FRAME_SIZE = "FftStream::size";        // :21
class FftStream : public Block {
    detail::Handle h_;
    size_t size_;
    ReadStream<Complex> src_;
    WriteStream<Complex> dst_;
public:
    FftStream(rr_block* h, size_t size, ReadStream<Complex> src, WriteStream<Complex> dst)
        : h_(h), size_(size), src_(std::move(src)), dst_(std::move(dst)) {}
    static std::pair<std::unique_ptr<FftStream>, ReadStream<Complex>> new_(ReadStream<Complex> src, size_t size) {   // :40-60
        auto [w, r] = new_stream<Complex>();
        if (size > w.free()) throw Error("FFT size must be no bigger than stream size");
        return {std::make_unique<FftStream>(rr_fftstream_create(size), size, std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {                    // :71-117 — input tags dropped, frame tags added
        auto [input, tags] = src_.read_buf();
        (void)tags;
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        if (w.st == RR_WAIT_SRC) { out.produce(0, {}); return BlockRet::wait(src_.wait_handle(), w.need); }
        if (w.st == RR_WAIT_DST) { out.produce(0, {}); return BlockRet::wait(dst_.wait_handle(), w.need); }
        std::vector<Tag> ft;                      // :98-111
        ft.reserve(w.produced / size_ * 3);
        for (size_t pos = 0; pos < w.produced; pos += size_) {
            ft.emplace_back(pos, TAG_FRAME_SIZE, (uint64_t)size_);
            ft.emplace_back(pos, TAG_FRAME, true);
            ft.emplace_back(pos + size_ - 1, TAG_FRAME, false);
        }
        input.consume(w.consumed);
        out.produce(w.produced, ft);
        return BlockRet::again();
    }
};

// ---- Fft (src/fft.rs:19-56): the message (PDU) form; process() is what work() does to each popped message ---------------------
class Fft {
    detail::Handle h_;
    size_t size_;
    static rr_block* make(size_t size) {
        if (!size) throw Error("FFT called with size 0");                                        // :24-26
        return rr_fftstream_create(size);
    }
public:
    explicit Fft(size_t size) : h_(make(size)), size_(size) {}
    std::vector<Complex> process(const std::vector<Complex>& msg) {                              // :41-55
        std::vector<Complex> out(msg.size());
        Complex dummy{};
        if (rr_fft_process(h_.h, reinterpret_cast<const rr_c32*>(msg.empty() ? &dummy : msg.data()), msg.size(),
                           reinterpret_cast<rr_c32*>(msg.empty() ? &dummy : out.data())) != 0)
            throw Error(rr_last_error());                                                        // "FFT expected {} samples, got {}"
        return out;
    }
    size_t size() const { return size_; }
};

// ---- #[rustradio(sync)] blocks: MultiplyConst (src/multiply_const.rs), FastFM (src/quadrature_demod.rs:144-165) ----------
// work() per rustradio_macros_code/src/lib.rs:458-515; a tag at position pos < n passes through at pos.
template <class In, class Out> class SyncBlock : public Block {
protected:
    detail::Handle h_;
    ReadStream<In> src_;
    WriteStream<Out> dst_;
public:
    SyncBlock(rr_block* h, ReadStream<In> src, WriteStream<Out> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)) {}
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {
        auto [input, tags] = src_.read_buf();
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        std::vector<Tag> keep;
        for (auto& t : tags) if (t.pos() < w.produced) keep.push_back(t);
        input.consume(w.consumed);
        out.produce(w.produced, keep);
        return w.st == RR_WAIT_DST ? BlockRet::wait(dst_.wait_handle(), w.need) : BlockRet::wait(src_.wait_handle(), w.need);
    }
};
template <class T> class MultiplyConst : public SyncBlock<T, T> {
    static rr_block* make(T val) {
        if constexpr (std::is_same<T, Complex>::value) return rr_multiply_const_c32_create(val.real(), val.imag());
        else return rr_multiply_const_f32_create(val);
    }
public:
    using SyncBlock<T, T>::SyncBlock;
    static std::pair<std::unique_ptr<MultiplyConst<T>>, ReadStream<T>> new_(ReadStream<T> src, T val) {
        auto [w, r] = new_stream<T>();
        return {std::make_unique<MultiplyConst<T>>(make(val), std::move(src), std::move(w)), std::move(r)};
    }
};
class FastFM : public SyncBlock<Complex, Float> {
public:
    using SyncBlock<Complex, Float>::SyncBlock;
    static std::pair<std::unique_ptr<FastFM>, ReadStream<Float>> new_(ReadStream<Complex> src) {
        auto [w, r] = new_stream<Float>();
        return {std::make_unique<FastFM>(rr_fastfm_create(), std::move(src), std::move(w)), std::move(r)};
    }
};

// ---- graph-level fusions (one block, one kernel; whole-stream output = the reference blocks in sequence) ------------------
// Tags: what the reference blocks in sequence would deliver (rr_block_tag_rule) — a chain holding a RationalResampler or a
// QuadratureDemod drops them, FirFilter -> FftFilter and Hilbert -> FirFilter forward them.
template <class In, class Out> class Fused : public Block {
    detail::Handle h_;
    ReadStream<In> src_;
    WriteStream<Out> dst_;
    detail::TagForwarder fwd_;
public:
    Fused(rr_block* h, ReadStream<In> src, WriteStream<Out> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)), fwd_(h_.h) {}
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {
        auto [input, tags] = src_.read_buf();
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        const std::vector<Tag> emit = fwd_.step(tags, w.consumed, w.produced);
        input.consume(w.consumed);
        out.produce(w.produced, emit);
        if (w.st == RR_AGAIN) return BlockRet::again();
        return w.st == RR_WAIT_DST ? BlockRet::wait(dst_.wait_handle(), w.need) : BlockRet::wait(src_.wait_handle(), w.need);
    }
    template <class MakeHandle> static std::pair<std::unique_ptr<Fused>, ReadStream<Out>> make(ReadStream<In> src, MakeHandle&& mk) {
        auto [w, r] = new_stream<Out>();
        return {std::make_unique<Fused>(mk(), std::move(src), std::move(w)), std::move(r)};
    }
};
inline const rr_c32* c32(const std::vector<Complex>& v) { return reinterpret_cast<const rr_c32*>(v.data()); }
// FftFilter -> RationalResampler -> QuadratureDemod (examples/rtl_fm.rs:381-419)
inline auto FmChain(ReadStream<Complex> src, const std::vector<Complex>& taps, size_t interp, size_t deci, Float gain, bool exact_atan2 = true) {
    return Fused<Complex, Float>::make(std::move(src), [&] { return rr_fm_chain_create(c32(taps), taps.size(), interp, deci, gain, exact_atan2 ? RR_ATAN2_EXACT : RR_ATAN2_FAST); });
}
// FirFilter -> FftFilter as one convolution (the north star's pair)
inline auto FirFftFilter(ReadStream<Complex> src, const std::vector<Complex>& fir_taps, const std::vector<Complex>& fft_taps) {
    return Fused<Complex, Complex>::make(std::move(src), [&] { return rr_fir_fftfilter_create(c32(fir_taps), fir_taps.size(), c32(fft_taps), fft_taps.size()); });
}
// FirFilter -> FftFilter -> RationalResampler -> QuadratureDemod (BASELINE's metric chain)
inline auto FirFmChain(ReadStream<Complex> src, const std::vector<Complex>& fir_taps, const std::vector<Complex>& fft_taps, size_t interp,
                       size_t deci, Float gain, bool exact_atan2 = true) {
    return Fused<Complex, Float>::make(std::move(src), [&] {
        return rr_fir_fm_chain_create(c32(fir_taps), fir_taps.size(), c32(fft_taps), fft_taps.size(), interp, deci, gain,
                                      exact_atan2 ? RR_ATAN2_EXACT : RR_ATAN2_FAST);
    });
}
// Hilbert -> FirFilter(deci)[.translate(samp_rate, freq)] (BASELINE configs[4]; hilbert.rs:38-129 + fir.rs:303-551)
inline auto HilbertFir(ReadStream<Float> src, size_t hilbert_ntaps, const window::WindowType& w, const std::vector<Complex>& taps,
                       size_t deci, bool translate = false, Float samp_rate = 0, Float freq = 0) {
    return Fused<Float, Complex>::make(std::move(src), [&] {
        return rr_hilbert_fir_create(hilbert_ntaps, w.kind, w.parm, c32(taps), taps.size(), deci, translate, samp_rate, freq);
    });
}
// FftFilterFloat -> RationalResampler -> MultiplyConst (examples/rtl_fm.rs:398-418)
inline auto AudioChain(ReadStream<Float> src, const std::vector<Float>& taps, size_t interp, size_t deci, Float scale) {
    return Fused<Float, Float>::make(std::move(src), [&] { return rr_audio_chain_create(taps.data(), taps.size(), interp, deci, scale); });
}

// ---- Hilbert (src/hilbert.rs) ----------------------------------------------------------------------------------------------
class Hilbert : public Block {
    detail::Handle h_;
    ReadStream<Float> src_;
    WriteStream<Complex> dst_;
public:
    Hilbert(rr_block* h, ReadStream<Float> src, WriteStream<Complex> dst) : h_(h), src_(std::move(src)), dst_(std::move(dst)) {}
    // asserts odd ntaps > 1 in the reference (:44-47); here: throws Error
    static std::pair<std::unique_ptr<Hilbert>, ReadStream<Complex>> new_(ReadStream<Float> src, size_t ntaps, const window::WindowType& w) {
        auto [ws, r] = new_stream<Complex>();
        rr_block* h = rr_hilbert_create(ntaps, w.kind, w.parm);
        return {std::make_unique<Hilbert>(h, std::move(src), std::move(ws)), std::move(r)};
    }
    const char* block_name() const override { return rr_block_name(h_.h); }
    bool eof() override { return rr_block_eof(h_.h, src_.eof()) != 0; }
    BlockRet work() override {                    // :72-128
        auto [input, tags] = src_.read_buf();
        auto out = dst_.write_buf();
        auto w = detail::work(h_.h, input, out);
        if (w.st == RR_WAIT_SRC) { out.produce(0, {}); return BlockRet::wait(src_.wait_handle(), w.need); }
        if (w.st == RR_WAIT_DST) { out.produce(0, {}); return BlockRet::wait(dst_.wait_handle(), w.need); }
        std::vector<Tag> keep;                    // :119-123
        for (auto& t : tags) if (t.pos() < w.produced) keep.push_back(t);
        out.produce(w.produced, keep);
        input.consume(w.consumed);
        return BlockRet::again();
    }
};

// ---- harness blocks (restated from the reference; host-side plumbing only) ----------------------------------------------------
struct Repeat {                                   // src/lib.rs Repeat
    uint64_t count; bool infinite;
    static Repeat finite(uint64_t n) { return {n, false}; }
    static Repeat infinite_() { return {0, true}; }
};

template <class T> class VectorSource : public Block {   // src/vector_source.rs:97-146
    WriteStream<T> dst_;
    std::vector<T> data_;
    Repeat repeat_;
    uint64_t repeat_count_ = 0;
    size_t pos_ = 0;
public:
    VectorSource(WriteStream<T> dst, std::vector<T> d, Repeat r) : dst_(std::move(dst)), data_(std::move(d)), repeat_(r) {}
    static std::pair<std::unique_ptr<VectorSource<T>>, ReadStream<T>> new_(std::vector<T> data, Repeat r = Repeat::finite(1)) {
        auto [w, rd] = new_stream<T>();
        return {std::make_unique<VectorSource<T>>(std::move(w), std::move(data), r), std::move(rd)};
    }
    const char* block_name() const override { return "VectorSource"; }
    bool eof() override { return !repeat_.infinite && repeat_count_ >= repeat_.count; }
    BlockRet work() override {
        if (eof() || data_.empty()) return BlockRet::eof();
        auto out = dst_.write_buf();
        const size_t n = std::min(out.len(), data_.size() - pos_);
        if (n == 0) { out.produce(0, {}); return BlockRet::wait(dst_.wait_handle(), 1); }
        std::vector<Tag> tags;
        if (pos_ == 0) {
            tags.emplace_back(0, "VectorSource::start", true);
            tags.emplace_back(0, "VectorSource::repeat", (uint64_t)repeat_count_);
            if (repeat_count_ == 0) tags.emplace_back(0, "VectorSource::first", true);
        }
        out.fill_from_slice(data_.data() + pos_, n);
        out.produce(n, tags);
        pos_ += n;
        if (pos_ == data_.size()) { pos_ = 0; repeat_count_++; }
        return eof() ? BlockRet::eof() : BlockRet::again();
    }
};

// FileSource<T> (src/file_source.rs:43-153): raw little-endian samples (`.c32` = interleaved f32 re, im; `.u8`
// RTL-SDR bytes, ...) from disk into a stream of any placement; a partial trailing sample is dropped at EOF
// (and on repeat), as the reference does.
template <class T> class FileSource : public Block {
    WriteStream<T> dst_;
    std::string filename_;
    FILE* f_ = nullptr;
    Repeat repeat_;
    uint64_t count_ = 0;                        // Repeat::again() counter (src/lib.rs:477-490)
    std::vector<uint8_t> buf_;                  // bytes of an incomplete sample + not yet emitted samples
    std::vector<uint8_t> rd_;
    bool again() { count_++; return repeat_.infinite || count_ < repeat_.count; }
public:
    FileSource(WriteStream<T> dst, std::string filename, Repeat r) : dst_(std::move(dst)), filename_(std::move(filename)), repeat_(r) {
        f_ = std::fopen(filename_.c_str(), "rb");
        if (!f_) throw Error("FileSource: cannot open " + filename_);       // Error::file_io (:64-66)
    }
    ~FileSource() override { if (f_) std::fclose(f_); }
    static std::pair<std::unique_ptr<FileSource<T>>, ReadStream<T>> new_(const std::string& filename, Repeat r = Repeat::finite(1)) {
        auto [w, rd] = new_stream<T>();
        return {std::make_unique<FileSource<T>>(std::move(w), filename, r), std::move(rd)};
    }
    const char* block_name() const override { return "FileSource"; }
    bool eof() override { return false; }
    BlockRet work() override {                  // :89-152
        auto o = dst_.write_buf();
        constexpr size_t ss = sizeof(T);
        size_t have = buf_.size() / ss;
        const size_t want = o.len();
        if (want == 0) { o.produce(0, {}); return BlockRet::wait(dst_.wait_handle(), 1); }
        if (have < want) {
            rd_.resize((want - have) * ss);
            const size_t n = std::fread(rd_.data(), 1, rd_.size(), f_);
            if (n == 0) {
                o.produce(0, {});
                if (again()) {
                    buf_.clear();
                    std::fseek(f_, 0, SEEK_SET);
                    return BlockRet::again();
                }
                return BlockRet::eof();
            }
            if (buf_.empty() && n % ss == 0) {  // fast path: whole samples only (:120-128)
                o.fill_from_slice(reinterpret_cast<const T*>(rd_.data()), n / ss);
                o.produce(n / ss, {});
                return BlockRet::again();
            }
            buf_.insert(buf_.end(), rd_.begin(), rd_.begin() + (std::ptrdiff_t)n);
        }
        have = buf_.size() / ss;
        if (have == 0) { o.produce(0, {}); return BlockRet{BlockRet::Pending, nullptr, 0}; }
        const size_t nemit = std::min(have, want);
        std::vector<T> tmp(nemit);
        std::memcpy(tmp.data(), buf_.data(), nemit * ss);
        buf_.erase(buf_.begin(), buf_.begin() + (std::ptrdiff_t)(nemit * ss));
        o.fill_from_slice(tmp.data(), nemit);
        o.produce(nemit, {});
        return BlockRet::again();
    }
};

template <class T> class VectorSink : public Block {     // src/vector_sink.rs:113-137
    ReadStream<T> src_;
    std::shared_ptr<std::vector<T>> data_ = std::make_shared<std::vector<T>>();
    std::shared_ptr<std::vector<Tag>> tags_ = std::make_shared<std::vector<Tag>>();
public:
    explicit VectorSink(ReadStream<T> src) : src_(std::move(src)) {}
    std::shared_ptr<std::vector<T>> hook() const { return data_; }           // vector_sink.rs:103-111
    std::shared_ptr<std::vector<Tag>> tag_hook() const { return tags_; }
    const char* block_name() const override { return "VectorSink"; }
    bool eof() override { return src_.eof(); }
    BlockRet work() override {
        auto [input, tags] = src_.read_buf();
        const size_t n = input.len();
        for (auto& t : tags) tags_->emplace_back(t.pos() + data_->size(), t.key(), t.val());
        const size_t at = data_->size();
        data_->resize(at + n);
        input.copy_to(data_->data() + at, n);
        input.consume(n);
        return BlockRet::wait(src_.wait_handle(), 1);
    }
};

template <class T> class NullSink : public Block {       // src/null_sink.rs:15-25
    ReadStream<T> src_;
public:
    explicit NullSink(ReadStream<T> src) : src_(std::move(src)) {}
    const char* block_name() const override { return "NullSink"; }
    bool eof() override { return src_.eof(); }
    BlockRet work() override {
        auto [input, tags] = src_.read_buf();
        (void)tags;
        input.consume(input.len());
        return BlockRet::wait(src_.wait_handle(), 1);
    }
};

// Moves samples (and tags) between streams of any placement: the explicit upload / download step of a
// graph that mixes CPU blocks and device-resident GPU chains.  No reference counterpart.
template <class T> class MemCopy : public Block {
    ReadStream<T> src_;
    WriteStream<T> dst_;
    std::vector<T> bounce_;
public:
    MemCopy(ReadStream<T> src, WriteStream<T> dst) : src_(std::move(src)), dst_(std::move(dst)) {}
    static std::pair<std::unique_ptr<MemCopy<T>>, ReadStream<T>> new_(ReadStream<T> src, Memory to) {
        auto [w, r] = new_stream<T>(default_stream_size(), to);
        return {std::make_unique<MemCopy<T>>(std::move(src), std::move(w)), std::move(r)};
    }
    const char* block_name() const override { return "MemCopy"; }
    bool eof() override { return src_.eof(); }
    BlockRet work() override {
        auto [input, tags] = src_.read_buf();
        auto out = dst_.write_buf();
        const size_t n = std::min(input.len(), out.len());
        if (n == 0) {
            out.produce(0, {});
            return input.len() == 0 ? BlockRet::wait(src_.wait_handle(), 1) : BlockRet::wait(dst_.wait_handle(), 1);
        }
        std::vector<Tag> keep;
        for (auto& t : tags) if (t.pos() < n) keep.push_back(t);
        if (!input.device()) out.fill_from_slice(input.slice(), n);
        else if (!out.device()) input.copy_to(out.slice(), n);
        else out.fill_from_device(input, n);                              // HBM ring to HBM ring: device-to-device
        input.consume(n);
        out.produce(n, keep);
        return BlockRet::again();
    }
};

// Tee (src/tee.rs:10-24): every input sample and tag goes to both outputs.
template <class T> class Tee : public Block {
    ReadStream<T> src_;
    WriteStream<T> dst1_, dst2_;
    std::vector<T> bounce_;
public:
    Tee(ReadStream<T> src, WriteStream<T> d1, WriteStream<T> d2) : src_(std::move(src)), dst1_(std::move(d1)), dst2_(std::move(d2)) {}
    static std::tuple<std::unique_ptr<Tee<T>>, ReadStream<T>, ReadStream<T>> new_(ReadStream<T> src) {
        auto [w1, r1] = new_stream<T>();
        auto [w2, r2] = new_stream<T>();
        return {std::make_unique<Tee<T>>(std::move(src), std::move(w1), std::move(w2)), std::move(r1), std::move(r2)};
    }
    const char* block_name() const override { return "Tee"; }
    bool eof() override { return src_.eof(); }
    BlockRet work() override {                    // the `sync` macro loop: min over the input and both outputs
        auto [input, tags] = src_.read_buf();
        auto o1 = dst1_.write_buf();
        auto o2 = dst2_.write_buf();
        const size_t n = std::min(input.len(), std::min(o1.len(), o2.len()));
        if (n == 0) {
            o1.produce(0, {}); o2.produce(0, {});
            if (input.len() == 0) return BlockRet::wait(src_.wait_handle(), 1);
            return BlockRet::wait(o1.len() == 0 ? dst1_.wait_handle() : dst2_.wait_handle(), 1);
        }
        std::vector<Tag> keep;
        for (auto& t : tags) if (t.pos() < n) keep.push_back(t);
        // every output takes the samples the cheapest way its placement allows: device rings straight from a device input
        // (rr_dstream_copy, no host hop), everything else through host memory (one download at most)
        const T* hp = nullptr;
        auto feed = [&](BufferWriter<T>& o) {
            if (input.device() && o.device()) { o.fill_from_device(input, n); return; }
            if (!hp) {
                if (input.device()) { bounce_.resize(n); input.copy_to(bounce_.data(), n); hp = bounce_.data(); }
                else hp = input.slice();
            }
            o.fill_from_slice(hp, n);
        };
        feed(o1);
        feed(o2);
        input.consume(n);
        o1.produce(n, keep);
        o2.produce(n, keep);
        return BlockRet::again();
    }
};

// SignalSourceComplex (src/signal_source.rs:9-63): amplitude * (sin(c), sin(c - pi/2)), the phase kept in
// f64 and wrapped with fmod each sample; fills the whole write window per work().
class SignalSourceComplex : public Block {
    WriteStream<Complex> dst_;
    Float amplitude_;
    double rad_per_sample_, current_ = 0.0;
    std::vector<Complex> tmp_;
public:
    SignalSourceComplex(WriteStream<Complex> dst, Float samp_rate, Float freq, Float amplitude)
        : dst_(std::move(dst)), amplitude_(amplitude),
          rad_per_sample_(2.0 * M_PI * (double)freq / (double)samp_rate) {
        if (!(samp_rate > 0.0f)) throw Error("SignalSourceComplex: samp_rate must be > 0");   // :26 assert
    }
    static std::pair<std::unique_ptr<SignalSourceComplex>, ReadStream<Complex>> new_(Float samp_rate, Float freq, Float amplitude) {
        auto [w, r] = new_stream<Complex>();
        return {std::make_unique<SignalSourceComplex>(std::move(w), samp_rate, freq, amplitude), std::move(r)};
    }
    const char* block_name() const override { return "SignalSourceComplex"; }
    bool eof() override { return false; }
    BlockRet work() override {                    // :55-63
        auto o = dst_.write_buf();
        const size_t n = o.len();
        tmp_.resize(n);
        for (size_t i = 0; i < n; i++) {          // Iterator::next, :41-52
            current_ = std::fmod(current_ + rad_per_sample_, 2.0 * M_PI);
            const Complex v((Float)std::sin(current_), (Float)std::sin(current_ - M_PI / 2.0));
            tmp_[i] = amplitude_ * v;
        }
        o.fill_from_slice(tmp_.data(), n);
        o.produce(n, {});
        return BlockRet::wait(dst_.wait_handle(), 1);
    }
};

// Single-threaded round-robin runner (src/graph.rs:99-173), minus timing statistics.  A block
// that reaches EOF is dropped, which closes its output streams (the reference: Arc strong
// count, src/stream.rs:148-150, 237-246); the run ends when a whole round moves no sample
// and returns no Again/Pending (src/graph.rs:108-154).
class Graph {
    std::vector<std::unique_ptr<Block>> blocks_;
public:
    void add(std::unique_ptr<Block> b) { blocks_.push_back(std::move(b)); }
    void run() {
        for (;;) {
            bool done = true;
            const uint64_t before = detail::activity();
            for (auto& bp : blocks_) {
                if (!bp) continue;
                const BlockRet ret = bp->work();
                switch (ret.kind) {
                case BlockRet::Again:
                case BlockRet::Pending: done = false; break;
                case BlockRet::EOF_: bp.reset(); done = false; break;
                case BlockRet::WaitForStream:
                    if (bp->eof() || ret.stream->closed()) { bp.reset(); done = false; }   // graph.rs:136-143
                    break;
                }
            }
            if (done && detail::activity() == before) break;
        }
    }
};


// Thread-per-block runner (src/mtgraph.rs:84-135).  Each block runs on its own thread until its work() says EOF, or says
// WaitForStream and either the block's own eof() holds or the stream's wait(need) reports that `need` can never be met
// (:109-115); Pending sleeps with a doubling back-off (:116-122).  The block is dropped when its thread ends, which closes
// its ends of its streams and so wakes and ends its neighbours.  A throwing work() cancels the graph (:99-105).
class MTGraph {
    std::vector<std::unique_ptr<Block>> blocks_;
public:
    void add(std::unique_ptr<Block> b) { blocks_.push_back(std::move(b)); }
    void run() {
        using namespace std::chrono;
        constexpr auto MIN_IDLE_SLEEP = microseconds(1000), MAX_IDLE_SLEEP = microseconds(100000);   // mtgraph.rs:14-15
        std::atomic<bool> cancel{false};
        std::mutex em;
        std::string err;
        std::vector<std::thread> threads;
        for (auto& slot : blocks_) {
            threads.emplace_back([&cancel, &em, &err, MIN_IDLE_SLEEP, MAX_IDLE_SLEEP, b = std::move(slot)]() mutable {
                auto idle = MIN_IDLE_SLEEP;
                bool running = true;
                while (running && !cancel.load()) {
                    BlockRet ret = BlockRet::again();
                    try {
                        ret = b->work();
                    } catch (const std::exception& e) {
                        std::lock_guard<std::mutex> g(em);
                        if (err.empty()) err = std::string("in block ") + b->block_name() + ": " + e.what();
                        cancel = true;
                        break;
                    }
                    switch (ret.kind) {
                    case BlockRet::Again: idle = MIN_IDLE_SLEEP; break;
                    case BlockRet::EOF_: running = false; break;
                    case BlockRet::WaitForStream: {
                        const bool never = ret.stream->wait(ret.need);
                        if (b->eof() || never) running = false;
                        break;
                    }
                    case BlockRet::Pending:
                        std::this_thread::sleep_for(idle);
                        idle = std::min(idle * 2, MAX_IDLE_SLEEP);
                        break;
                    }
                }
                b.reset();                                   // the drop that closes this block's stream ends
            });
        }
        blocks_.clear();
        for (auto& t : threads) t.join();
        if (!err.empty()) throw Error(err);
    }
};

}  // namespace rustradio
