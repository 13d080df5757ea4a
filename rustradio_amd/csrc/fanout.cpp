// fanout.cpp — rr_fanout_*: the streaming fan-out of a shared source across GPUs, one process per GPU.
//
// The reference fans one source out to its chains with a Tee tree inside one process (src/tee.rs:10-24).  Across GPUs the
// chains are sharded by channel and the only exchange is the source itself: the owning rank's source block writes tile t
// into one half of a double buffer in HBM, a broadcast on a COMMUNICATION stream (RCCL over xGMI) delivers it into the same
// half on every other rank, and every rank's blocks read tile t on their own stream while tile t+1 is in flight.  Two events
// per half order the streams: `ready` (broadcast done -> blocks may read) and `freed` (blocks done -> the half may be
// overwritten).  Nothing here blocks the host.
//
// RCCL is bound at run time (dlopen of the already-loaded librccl when there is one), so the library loads — and every
// single-GPU entry point works — on a machine without it; a one-rank fan-out needs no communicator at all.
#include <dlfcn.h>

#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "common.hpp"
#include "blocks.hpp"
#include "../../include/rustradio_amd.h"

namespace rr {

namespace {
// the five RCCL entry points this file uses (rccl.h:187,220,260,339,591); ncclUniqueId is 128 opaque bytes passed by value
struct UniqueId { char internal[RR_FANOUT_ID_BYTES]; };
using Comm = void*;
struct Rccl {
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Scatter)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;       // rccl.h:767 (RCCL extension)
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;          // rccl.h:678
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        void* h = nullptr;
        for (const char* name : {"librccl.so", "librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);           // the copy the process already runs on (PyTorch's)
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!h) { x.why = std::string("librccl not found: ") + dlerror(); return x; }
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.Broadcast = reinterpret_cast<decltype(x.Broadcast)>(dlsym(h, "ncclBroadcast"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        x.Scatter = reinterpret_cast<decltype(x.Scatter)>(dlsym(h, "ncclScatter"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(h, "ncclAllGather"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.Broadcast && x.GetErrorString;
        if (!x.ok) x.why = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclBroadcast";
        return x;
    }();
    return r;
}
void nccl_check(int rc, const char* what) {
    if (rc != 0) throw Error(std::string(what) + ": " + rccl().GetErrorString(rc));
}
constexpr int kNcclUint8 = 1;                                            // rccl.h:460
}  // namespace

struct Fanout {
    int rank, world, src, device = 0;
    size_t tile_bytes, piece;                                            // piece: bytes per rank of the scattered form
    bool timing, mesh;
    Comm comm = nullptr;
    void* buf[2] = {nullptr, nullptr};
    hipStream_t cs = nullptr;                                            // the communication stream
    hipEvent_t ready[2] = {}, freed[2] = {}, written[2] = {};
    long long rel[2] = {-1, -1};                                         // last tile released from each half
    long long issued = -1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;                // one per timed broadcast

    Fanout(const void* id, int rank_, int world_, int src_, size_t bytes, int flags)
        : rank(rank_), world(world_), src(src_), tile_bytes(bytes), piece(0), timing((flags & RR_FANOUT_TIMING) != 0),
          mesh((flags & RR_FANOUT_MESH) != 0) {
        if (world < 1 || rank < 0 || rank >= world || src < 0 || src >= world) throw Error("rr_fanout_create: rank / world / src_rank out of range");
        if (bytes == 0) throw Error("rr_fanout_create: tile_bytes must be nonzero");
        piece = (bytes + (size_t)world - 1) / (size_t)world;
        device = thread_device();
        RR_HIP(hipSetDevice(device));
        if (world > 1 || (flags & RR_FANOUT_RCCL_ALWAYS)) {
            if (!id) throw Error("rr_fanout_create: a group of more than one rank needs the id of rr_fanout_unique_id");
            if (!rccl().ok) throw Error(rccl().why);
            if (mesh && !(rccl().Scatter && rccl().AllGather)) throw Error("RR_FANOUT_MESH: this librccl has no ncclScatter / ncclAllGather");
            UniqueId u;
            std::memcpy(u.internal, id, sizeof u.internal);
            nccl_check(rccl().CommInitRank(&comm, world, u, rank), "ncclCommInitRank");
        }
        try {
            RR_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            for (int h = 0; h < 2; h++) {
                RR_HIP(hipMalloc(&buf[h], piece * (size_t)world));
                RR_HIP(hipEventCreateWithFlags(&ready[h], hipEventDisableTiming));
                RR_HIP(hipEventCreateWithFlags(&freed[h], hipEventDisableTiming));
                RR_HIP(hipEventCreateWithFlags(&written[h], hipEventDisableTiming));
            }
        } catch (...) { close(); throw; }
    }
    void close() {
        (void)hipSetDevice(device);
        if (cs) (void)hipStreamSynchronize(cs);
        if (comm) { (void)rccl().CommDestroy(comm); comm = nullptr; }
        for (auto& e : spans) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        spans.clear();
        for (int h = 0; h < 2; h++) {
            if (buf[h]) (void)hipFree(buf[h]);
            if (ready[h]) (void)hipEventDestroy(ready[h]);
            if (freed[h]) (void)hipEventDestroy(freed[h]);
            if (written[h]) (void)hipEventDestroy(written[h]);
            buf[h] = nullptr; ready[h] = freed[h] = written[h] = nullptr;
        }
        if (cs) { (void)hipStreamDestroy(cs); cs = nullptr; }
    }
    ~Fanout() { close(); }

    void half_is_released(unsigned long long t, const char* who) const {
        if (t >= 2 && rel[t & 1] != (long long)t - 2)
            throw Error(std::string(who) + ": tile t - 2 has not been released from this half of the double buffer");
    }
    // owning rank: where the source block writes tile t; its stream first waits until the blocks have released the half
    void* produce_buf(unsigned long long t, hipStream_t producer) {
        if (rank != src) throw Error("rr_fanout_produce_buf: only the owning rank produces");
        if ((long long)t != issued + 1) throw Error("rr_fanout_produce_buf: tiles are produced in order, one ahead of the last submit");
        const int h = (int)(t & 1);
        RR_HIP(hipSetDevice(device));
        half_is_released(t, "rr_fanout_produce_buf");
        if (t >= 2) RR_HIP(hipStreamWaitEvent(producer, freed[h], 0));
        return buf[h];
    }
    // every rank, once per tile and in order: the broadcast of tile t (on the owning rank after the producer's writes)
    void submit(unsigned long long t, hipStream_t producer) {
        if ((long long)t != issued + 1) throw Error("rr_fanout_submit: tiles are submitted in order");
        const int h = (int)(t & 1);
        half_is_released(t, "rr_fanout_submit");
        RR_HIP(hipSetDevice(device));
        if (rank == src) {
            RR_HIP(hipEventRecord(written[h], producer));
            RR_HIP(hipStreamWaitEvent(cs, written[h], 0));
        } else if (t >= 2) {
            RR_HIP(hipStreamWaitEvent(cs, freed[h], 0));
        }
        if (comm) {
            hipEvent_t b = nullptr, e = nullptr;
            // a sample of the broadcasts is timed (timing events cost the communication stream ~20 us each on this
            // runtime); bounded: rr_fanout_stats drains the list
            const bool timed = timing && (t % 4 == 0) && spans.size() < 4096;
            if (timed) {
                RR_HIP(hipEventCreate(&b));
                RR_HIP(hipEventCreate(&e));
                RR_HIP(hipEventRecord(b, cs));
            }
            if (mesh) {
                // the full mesh instead of one link: 1/world of the tile from the owner to every rank (each into its own
                // slot), then an in-place all-gather between the receivers
                char* own = static_cast<char*>(buf[h]) + (size_t)rank * piece;
                nccl_check(rccl().Scatter(buf[h], own, piece, kNcclUint8, src, comm, cs), "ncclScatter");
                nccl_check(rccl().AllGather(own, buf[h], piece, kNcclUint8, comm, cs), "ncclAllGather");
            } else {
                nccl_check(rccl().Broadcast(buf[h], buf[h], tile_bytes, kNcclUint8, src, comm, cs), "ncclBroadcast");
            }
            if (timed) {
                RR_HIP(hipEventRecord(e, cs));
                spans.emplace_back(b, e);
            }
        }
        RR_HIP(hipEventRecord(ready[h], cs));
        issued = (long long)t;
    }
    const void* acquire(unsigned long long t, hipStream_t compute) {
        if ((long long)t > issued || (long long)t + 2 <= issued) throw Error("rr_fanout_acquire: tile is not in the double buffer");
        const int h = (int)(t & 1);
        RR_HIP(hipSetDevice(device));
        RR_HIP(hipStreamWaitEvent(compute, ready[h], 0));
        return buf[h];
    }
    void release(unsigned long long t, hipStream_t compute) {
        const int h = (int)(t & 1);
        RR_HIP(hipSetDevice(device));
        if ((long long)t > issued) throw Error("rr_fanout_release: tile was never submitted");
        if ((long long)t + 2 <= issued) throw Error("rr_fanout_release: tile is no longer in the double buffer");
        // ONE release per tile (the header's protocol): a second record on another stream would replace this event and the
        // next broadcast into the half would wait for that stream only
        if (rel[h] == (long long)t) throw Error("rr_fanout_release: tile already released (one compute stream per fan-out: join the readers' streams first)");
        RR_HIP(hipEventRecord(freed[h], compute));
        rel[h] = (long long)t;
    }
    void stats(double* ms, size_t* n) {
        RR_HIP(hipSetDevice(device));
        RR_HIP(hipStreamSynchronize(cs));
        std::vector<std::pair<hipEvent_t, hipEvent_t>> done;
        done.swap(spans);                                              // the list is drained whatever happens below
        double sum = 0;
        size_t ok = 0;
        for (auto& e : done) {
            float x = 0;
            if (hipEventElapsedTime(&x, e.first, e.second) == hipSuccess) { sum += x; ok++; }
            (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
        }
        if (ms) *ms = sum;
        if (n) *n = ok;
    }
};
}  // namespace rr

struct rr_fanout { std::unique_ptr<rr::Fanout> f; };

template <class F> static int fan_guard(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { rr::set_last_error(e.what()); return RR_ERR; } catch (...) { rr::set_last_error("non-standard exception"); return RR_ERR; }
}

extern "C" {

int rr_fanout_unique_id(void* id128) {
    return fan_guard([&] {
        if (!id128) throw rr::Error("rr_fanout_unique_id: null buffer");
        if (!rr::rccl().ok) throw rr::Error(rr::rccl().why);
        rr::UniqueId u;
        rr::nccl_check(rr::rccl().GetUniqueId(&u), "ncclGetUniqueId");
        std::memcpy(id128, u.internal, sizeof u.internal);
    });
}
rr_fanout* rr_fanout_create(const void* id128, int rank, int world, int src_rank, size_t tile_bytes, int flags) {
    try {
        std::unique_ptr<rr::Fanout> f(new rr::Fanout(id128, rank, world, src_rank, tile_bytes, flags));
        return new rr_fanout{std::move(f)};
    } catch (const std::exception& e) {
        rr::set_last_error(e.what());
        return nullptr;
    } catch (...) {
        rr::set_last_error("non-standard exception");
        return nullptr;
    }
}
void rr_fanout_destroy(rr_fanout* f) { try { delete f; } catch (...) {} }
void* rr_fanout_produce_buf(rr_fanout* f, unsigned long long t, void* producer_stream) {
    void* p = nullptr;
    if (!f) return nullptr;
    return fan_guard([&] { p = f->f->produce_buf(t, static_cast<hipStream_t>(producer_stream)); }) == 0 ? p : nullptr;
}
int rr_fanout_submit(rr_fanout* f, unsigned long long t, void* producer_stream) {
    if (!f) return RR_ERR;
    return fan_guard([&] { f->f->submit(t, static_cast<hipStream_t>(producer_stream)); });
}
const void* rr_fanout_acquire(rr_fanout* f, unsigned long long t, void* compute_stream) {
    const void* p = nullptr;
    if (!f) return nullptr;
    return fan_guard([&] { p = f->f->acquire(t, static_cast<hipStream_t>(compute_stream)); }) == 0 ? p : nullptr;
}
int rr_fanout_release(rr_fanout* f, unsigned long long t, void* compute_stream) {
    if (!f) return RR_ERR;
    return fan_guard([&] { f->f->release(t, static_cast<hipStream_t>(compute_stream)); });
}
int rr_fanout_stats(rr_fanout* f, double* broadcast_ms, size_t* broadcasts) {
    if (!f) return RR_ERR;
    return fan_guard([&] { f->f->stats(broadcast_ms, broadcasts); });
}

}  // extern "C"
