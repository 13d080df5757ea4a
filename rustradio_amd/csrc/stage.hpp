// stage.hpp — host <-> device copies of memory the library does NOT know to be page-locked, through pinned buffers of its own.
//
// Why (round 6, the `abort()` of rounds 4 and 5 named at last — profiles/r06_abort_backtrace.txt): handing hipMemcpyAsync a
// PAGEABLE host pointer makes the runtime page-lock that range itself and keep the pinned object for reuse (ROCclr
// DmaBlitManager::hsaCopyStagedOrPinned: "HSA Copy Using Pinned resource").  A host allocator recycles virtual addresses:
// glibc trims the heap top when a 4 MB window is freed and grows it again for the next one, big vectors are mmap'ed and
// munmap'ed, and the next array of the same size lands on the SAME address with OTHER pages behind it.  A copy engine that still
// trusts its earlier pin of that address reads the old pages (round 4: "outputs computed from the previous contents") or
// faults ("Memory access fault by GPU node-2 on address 0x6423...[heap]. Reason: Unknown", then abort() inside rr_block_work:
// 1 of 14 full suite runs).  So no GPU engine is ever pointed at caller memory the caller did not register (rr_host_register, a
// page-aligned ring registered once: INTEGRATION.md): such windows are copied by the CPU into / out of two pinned chunks owned
// by the library, and only those cross the bus — 12 GB/s of memcpy instead of a 47 GB/s DMA.  Register the rings.
#pragma once
#include <cstddef>

#include "common.hpp"

namespace rr {

struct HostStage {
    static constexpr size_t CHUNK = (size_t)2 << 20;
    unsigned char* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool pending[2] = {false, false};
    HostStage() = default;
    HostStage(const HostStage&) = delete;
    HostStage& operator=(const HostStage&) = delete;
    ~HostStage();
    // src (any host memory) -> dst (device), ordered on `s`.  On return all of src has been read (the last chunks may still be
    // on their way from the pinned buffers): the caller may reuse or free src.
    void h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);
    // src (device, as of everything enqueued on `s` so far) -> dst (any host memory).  Returns when dst is complete.
    void d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s);
    // rows x row_bytes out of a pitched device buffer into a pitched host buffer
    void d2h_2d(void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t row_bytes, size_t rows, hipStream_t s);
    // the caller has just synchronised the stream(s) every pending chunk was enqueued on: nothing to wait for any more.  (Events
    // are never waited on across calls otherwise either — the stream an event was last recorded on may be gone by then, and
    // hipEventSynchronize on such an event fails: "operation not permitted on an event last recorded in a capturing stream",
    // six of fifteen suite runs in the multi-threaded resident-graph test before this.)
    void quiesced() { pending[0] = pending[1] = false; }

private:
    void init();
    void wait(int i);
};

// one process-wide stage under a lock, for the tables blocks upload at construction (DevBuf::upload) and the copies of
// rr_dstream_copy_in / _out from unregistered memory; blocking
void stage_upload_sync(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s);
void stage_download_sync(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s);

}  // namespace rr
