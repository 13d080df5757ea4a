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
private:
    bool try_vmm();
};

}  // namespace rr
