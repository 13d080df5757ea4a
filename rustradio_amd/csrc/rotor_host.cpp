// rotor_host.cpp — see rotor_host.hpp.  Host code: compiled with -ffp-contract=off (csrc/Makefile CXXFLAGS), no fast-math.
#include "rotor_host.hpp"

#include <algorithm>
#include <chrono>

namespace rr {

HostRotor::HostRotor(float p0x, float p0y, float stx, float sty, size_t capacity)
    : cap(capacity), mask(capacity - 1), px(p0x), py(p0y), sx(stx), sy(sty) {
    if (cap == 0 || (cap & mask)) throw Error("HostRotor: capacity must be a power of two");
    void* p = nullptr;
    RR_HIP(hipHostMalloc(&p, cap * sizeof(cf), hipHostMallocDefault));
    ring = static_cast<cf*>(p);
    th = std::thread([this] { run(); });
}

HostRotor::~HostRotor() {
    stop.store(true);
    { std::lock_guard<std::mutex> g(m); }
    cv.notify_all();
    if (th.joinable()) th.join();
    if (ring) (void)hipHostFree(ring);
}

void HostRotor::run() {
    uint64_t g = 0;
    // volatile-free strict f32: every product and every sum below is one rounded IEEE operation (no contraction), in
    // num-complex's order for `phase * step`: (re*re - im*im, re*im + im*re)
    float x = px, y = py;
    const float cx = sx, cy = sy;
    while (!stop.load(std::memory_order_relaxed)) {
        const uint64_t lim = tail.load(std::memory_order_acquire) + cap;
        if (g >= lim) {                                           // ring full: wait for the consumer
            std::unique_lock<std::mutex> lk(m);
            cv.wait_for(lk, std::chrono::milliseconds(2), [&] { return stop.load() || tail.load() + cap > g; });
            continue;
        }
        const uint64_t n = std::min<uint64_t>(lim - g, 1u << 14);
        for (uint64_t i = 0; i < n; i++) {
            ring[(g + i) & mask] = mkcf(x, y);                    // the phase that rotates output g + i (fir.rs:468-469)
            const float nx = x * cx - y * cy;
            const float ny = x * cy + y * cx;
            x = nx; y = ny;
        }
        g += n;
        gen.store(g, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(m); }                    // (a waiter between its test and its sleep must not miss this)
        cv.notify_all();
    }
}

uint64_t HostRotor::wait_for(uint64_t upto, unsigned ms) {
    uint64_t g = gen.load(std::memory_order_acquire);
    if (g >= upto || ms == 0) return g;
    std::unique_lock<std::mutex> lk(m);
    cv.wait_for(lk, std::chrono::milliseconds(ms), [&] { return gen.load(std::memory_order_acquire) >= upto; });
    return gen.load(std::memory_order_acquire);
}

void HostRotor::release(uint64_t upto) {
    uint64_t t = tail.load(std::memory_order_relaxed);
    while (t < upto && !tail.compare_exchange_weak(t, upto, std::memory_order_release)) {}
    { std::lock_guard<std::mutex> lk(m); }
    cv.notify_all();
}

}  // namespace rr
