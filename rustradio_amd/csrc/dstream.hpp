// dstream.hpp — a stream ring resident in HBM: what ReadStream<T>/WriteStream<T> (src/stream.rs:187-310) over
// the double-mapped circular buffer (src/nowasm/circular_buffer.rs:98-128) are on the host.
//
// Same window contract as the reference: the read window is ALL readable elements, the write window
// ALL free space, both contiguous, fixed capacity (default 4,096,000 bytes, src/stream.rs:105).
//
// Contiguity, the reference's way: ONE physical allocation mapped at two consecutive virtual ranges
// through HIP's virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap twice), so a window
// that runs past the end of the ring simply continues into the second mapping of its start; no byte is
// ever moved.  Where that API is unavailable (or RR_DSTREAM_NO_VMM is set) the ring lives in a linear
// buffer of twice its capacity and the readable part is moved to the front when the write window would
// run off the end — at that point it starts beyond `cap` and is at most `cap` long, so source and
// destination never overlap, and every element is moved at most once per `cap` elements written.
// All bookkeeping is on the host (counts never depend on data); device work is enqueued on the caller's
// HIP stream.
#pragma once
#include <vector>

#include "common.hpp"

namespace rr {

struct DStream {
    size_t es, cap;              // element size, capacity in elements
    int device;
    // double-mapped ring
    bool vmm = false;
    unsigned char* va = nullptr; // 2 * phys bytes of address space
    size_t phys = 0;             // bytes of the physical allocation (>= cap * es, multiple of the granularity)
    hipMemGenericAllocationHandle_t handle{};
    size_t rb = 0, used_ = 0;    // read offset in bytes (< phys), readable elements
    // fallback: linear buffer of 2 * cap elements, readable = [r, w)
    DevBuf<unsigned char> buf;
    size_t r = 0, w = 0;

    DStream(size_t elem_size, size_t capacity_bytes);
    ~DStream();
    DStream(const DStream&) = delete;
    size_t used() const { return vmm ? used_ : w - r; }
    size_t free() const { return cap - used(); }
    const void* read_ptr() const { return vmm ? va + rb : buf.p + r * es; }
    void* write_ptr(hipStream_t s);          // makes the free space contiguous (fallback: may enqueue the move on s)
    void consume(size_t n);
    void produce(size_t n);

    // Streams (round 3).  Every entry point that touches the ring on a HIP stream declares it here first:
    //   will_read(s)  — work on `s` is about to READ the read window   (a block's kernels, copy_out, the source side of a copy)
    //   will_write(s) — work on `s` is about to WRITE the write window (copy_in, a block's output, the destination of a copy)
    // As long as one stream drives the ring (the common case) this costs two pointer compares.  When a different stream
    // shows up — a source pushing window k + 1 on a copy stream while the blocks of window k run on the compute stream —
    // it is made to wait for the other side first: an event is recorded on the earlier stream NOW (which covers everything
    // enqueued on it so far, the earlier access included) and the new stream waits for it.  Reads wait for the last
    // writer; writes wait for the last writer and for every stream that has read since.
    void will_read(hipStream_t s);
    void will_write(hipStream_t s);
private:
    hipStream_t last_writer_ = nullptr;
    bool has_writer_ = false;
    std::vector<hipStream_t> readers_;       // streams that read since the last write
    hipEvent_t ev_ = nullptr;
    void order_after(hipStream_t later, hipStream_t earlier);
    bool try_vmm();
};

}  // namespace rr
