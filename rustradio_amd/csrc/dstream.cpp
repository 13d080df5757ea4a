// dstream.cpp — device-resident stream rings (see dstream.hpp).
#include "dstream.hpp"

#include <cstdlib>

#include "blocks.hpp"

namespace rr {

bool DStream::try_vmm() {
    if (build_opts().dstream_no_vmm) return false;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return false;
    const size_t bytes = cap * es;
    phys = (bytes + gran - 1) / gran * gran;
    if (hipMemCreate(&handle, phys, &prop, 0) != hipSuccess) return false;
    void* p = nullptr;
    if (hipMemAddressReserve(&p, 2 * phys, 0, nullptr, 0) != hipSuccess) { (void)hipMemRelease(handle); return false; }
    va = static_cast<unsigned char*>(p);
    bool ok = hipMemMap(va, phys, 0, handle, 0) == hipSuccess;
    bool ok2 = ok && hipMemMap(va + phys, phys, 0, handle, 0) == hipSuccess;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (!ok2 || hipMemSetAccess(va, 2 * phys, &acc, 1) != hipSuccess) {
        if (ok2) (void)hipMemUnmap(va + phys, phys);
        if (ok) (void)hipMemUnmap(va, phys);
        (void)hipMemAddressFree(va, 2 * phys);
        (void)hipMemRelease(handle);
        va = nullptr;
        (void)hipGetLastError();
        return false;
    }
    return true;
}

DStream::DStream(size_t elem_size, size_t capacity_bytes) : es(elem_size), cap(0), device(thread_device()) {
    if (!(es == 1 || es == 2 || es == 4 || es == 8 || es == 16)) throw Error("dstream: element size must be 1,2,4,8 or 16");
    cap = capacity_bytes / es;
    if (cap == 0) throw Error("dstream: capacity smaller than one element");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error("no usable HIP device");
    RR_HIP(hipSetDevice(device));
    vmm = try_vmm();
    if (!vmm) buf.reserve(2 * cap * es);
}

DStream::~DStream() {
    if (ev_) (void)hipEventDestroy(ev_);
    if (vmm) {
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();            // nothing may still be running on the mapping
        (void)hipMemUnmap(va, phys);
        (void)hipMemUnmap(va + phys, phys);
        (void)hipMemAddressFree(va, 2 * phys);
        (void)hipMemRelease(handle);
    }
}

void DStream::order_after(hipStream_t later, hipStream_t earlier) {
    if (later == earlier) return;
    if (!ev_) RR_HIP(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
    RR_HIP(hipEventRecord(ev_, earlier));            // everything enqueued on `earlier` so far, the access in question included
    RR_HIP(hipStreamWaitEvent(later, ev_, 0));
}
void DStream::will_read(hipStream_t s) {
    if (has_writer_) order_after(s, last_writer_);
    for (hipStream_t r : readers_) if (r == s) return;
    readers_.push_back(s);
}
void DStream::will_write(hipStream_t s) {
    if (has_writer_) order_after(s, last_writer_);
    for (hipStream_t r : readers_) order_after(s, r);
    readers_.clear();
    last_writer_ = s;
    has_writer_ = true;
}

void* DStream::write_ptr(hipStream_t s) {
    if (vmm) return va + (rb + used_ * es) % phys;
    if (r == w) r = w = 0;                          // empty: start over at the front (here, on the writer's side, never in consume)
    if (w + free() > 2 * cap) {                     // the write window would run off the end
        const size_t n = used();                    // here r > cap >= n: source and destination are disjoint
        if (n) RR_HIP(hipMemcpyAsync(buf.p, buf.p + r * es, n * es, hipMemcpyDeviceToDevice, s));
        r = 0; w = n;
    }
    return buf.p + w * es;
}
void DStream::consume(size_t n) {
    if (n > used()) throw Error("dstream consume: n > readable");
    if (vmm) {
        // the offsets only ever advance: a writer on another thread may hold the write window it was handed before
        // this call, and it must stay where it is
        rb = (rb + n * es) % phys;
        used_ -= n;
        return;
    }
    r += n;
}
void DStream::produce(size_t n) {
    if (n > free()) throw Error("dstream produce: n > free");
    if (vmm) { used_ += n; return; }
    if (w + n > 2 * cap) throw Error("dstream produce: n > free");
    w += n;
}

}  // namespace rr
