// dstream.hpp — a stream ring resident in HBM: what ReadStream<T>/WriteStream<T> (src/stream.rs:187-310) over
// the double-mapped circular buffer (src/nowasm/circular_buffer.rs:98-128) are on the host.
//
// Same window contract as the reference: the read window is ALL readable elements, the write window
// ALL free space, both contiguous, fixed capacity (default 4,096,000 bytes, src/stream.rs:105).
// Contiguity without a double mapping: the ring lives in a linear buffer of twice its capacity; when
// the write window would run off the end the readable part is moved to the front first — at that
// point it starts beyond `cap` and is at most `cap` long, so source and destination never overlap,
// and every element is moved at most once per `cap` elements written.  All bookkeeping is on the
// host (counts never depend on data), device work is enqueued on the caller's HIP stream.
#pragma once
#include "common.hpp"

namespace rr {

struct DStream {
    size_t es, cap;              // element size, capacity in elements
    DevBuf<unsigned char> buf;   // 2 * cap * es bytes
    size_t r = 0, w = 0;         // readable = [r, w) in elements
    int device;
    DStream(size_t elem_size, size_t capacity_bytes);
    size_t used() const { return w - r; }
    size_t free() const { return cap - used(); }
    const void* read_ptr() const { return buf.p + r * es; }
    void* write_ptr(hipStream_t s);          // makes the free space contiguous (may enqueue the move on s)
    void consume(size_t n);
    void produce(size_t n);
};

}  // namespace rr
