// dstream.cpp — device-resident stream rings (see dstream.hpp).
#include "dstream.hpp"

#include "blocks.hpp"

namespace rr {

DStream::DStream(size_t elem_size, size_t capacity_bytes) : es(elem_size), cap(0), device(thread_device()) {
    if (!(es == 1 || es == 2 || es == 4 || es == 8 || es == 16)) throw Error("dstream: element size must be 1,2,4,8 or 16");
    cap = capacity_bytes / es;
    if (cap == 0) throw Error("dstream: capacity smaller than one element");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error("no usable HIP device");
    RR_HIP(hipSetDevice(device));
    buf.reserve(2 * cap * es);
}

void* DStream::write_ptr(hipStream_t s) {
    if (w + free() > 2 * cap) {                     // the write window would run off the end
        const size_t n = used();                    // here r > cap >= n: source and destination are disjoint
        if (n) RR_HIP(hipMemcpyAsync(buf.p, buf.p + r * es, n * es, hipMemcpyDeviceToDevice, s));
        r = 0; w = n;
    }
    return buf.p + w * es;
}
void DStream::consume(size_t n) {
    if (n > used()) throw Error("dstream consume: n > readable");
    r += n;
    if (r == w) r = w = 0;
}
void DStream::produce(size_t n) {
    if (n > free() || w + n > 2 * cap) throw Error("dstream produce: n > free");
    w += n;
}

}  // namespace rr
