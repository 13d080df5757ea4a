// stage.cpp — see stage.hpp.
#include "stage.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>

namespace rr {

void HostStage::init() {
    if (buf[0]) return;
    for (int i = 0; i < 2; i++) {
        void* p = nullptr;
        RR_HIP(hipHostMalloc(&p, CHUNK, hipHostMallocDefault));
        buf[i] = static_cast<unsigned char*>(p);
        RR_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        pending[i] = false;
    }
}

HostStage::~HostStage() {
    for (int i = 0; i < 2; i++) {
        if (ev[i]) { if (pending[i]) (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); }
        if (buf[i]) (void)hipHostFree(buf[i]);
    }
}

void HostStage::wait(int i) {
    if (!pending[i]) return;
    RR_HIP(hipEventSynchronize(ev[i]));
    pending[i] = false;
}

void HostStage::h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    if (!bytes) return;
#ifdef RR_STAGE_OFF   /* reproducer builds only (tools/pageable_churn.py): the rounds 1-5 behaviour, a DMA straight out of caller memory */
    RR_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    return;
#endif
    init();
    const unsigned char* src = static_cast<const unsigned char*>(src_host);
    unsigned char* dst = static_cast<unsigned char*>(dst_dev);
    int i = 0;
    for (size_t off = 0; off < bytes; off += CHUNK, i ^= 1) {
        const size_t n = std::min(CHUNK, bytes - off);
        wait(i);                                            // the DMA that last read this chunk has finished
        std::memcpy(buf[i], src + off, n);                  // (while the other chunk crosses the bus)
        RR_HIP(hipMemcpyAsync(dst + off, buf[i], n, hipMemcpyHostToDevice, s));
        RR_HIP(hipEventRecord(ev[i], s));
        pending[i] = true;
    }
}

void HostStage::d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s) {
    d2h_2d(dst_host, bytes, src_dev, bytes, bytes, 1, s);
}

void HostStage::d2h_2d(void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t row_bytes, size_t rows,
                       hipStream_t s) {
    if (!row_bytes || !rows) return;
#ifdef RR_STAGE_OFF
    RR_HIP(hipMemcpy2DAsync(dst_host, dst_pitch, src_dev, src_pitch, row_bytes, rows, hipMemcpyDeviceToHost, s));
    RR_HIP(hipStreamSynchronize(s));
    return;
#endif
    init();
    unsigned char* dst = static_cast<unsigned char*>(dst_host);
    const unsigned char* src = static_cast<const unsigned char*>(src_dev);
    // the pieces in order: (row, offset, length); piece k + 1 is enqueued before piece k is copied out of its chunk
    struct Piece { size_t row, off, n; };
    auto piece = [&](size_t k, Piece& p) {
        const size_t per_row = (row_bytes + CHUNK - 1) / CHUNK;
        if (k >= per_row * rows) return false;
        p.row = k / per_row;
        p.off = (k % per_row) * CHUNK;
        p.n = std::min(CHUNK, row_bytes - p.off);
        return true;
    };
    auto enqueue = [&](const Piece& p, int i) {
        wait(i);
        RR_HIP(hipMemcpyAsync(buf[i], src + p.row * src_pitch + p.off, p.n, hipMemcpyDeviceToHost, s));
        RR_HIP(hipEventRecord(ev[i], s));
        pending[i] = true;
    };
    Piece cur, nxt;
    if (!piece(0, cur)) return;
    enqueue(cur, 0);
    int i = 0;
    for (size_t k = 0;; k++, i ^= 1) {
        const bool more = piece(k + 1, nxt);
        if (more) enqueue(nxt, i ^ 1);
        wait(i);
        std::memcpy(dst + cur.row * dst_pitch + cur.off, buf[i], cur.n);
        if (!more) break;
        cur = nxt;
    }
}

namespace {
std::mutex g_stage_m;
HostStage& global_stage() { static HostStage* st = new HostStage(); return *st; }   // (never destroyed: no HIP calls at exit)
}  // namespace

void stage_upload_sync(void* dst_dev, const void* src_host, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    std::lock_guard<std::mutex> g(g_stage_m);
    global_stage().h2d(dst_dev, src_host, bytes, s);
    RR_HIP(hipStreamSynchronize(s));
    global_stage().quiesced();
}
void stage_download_sync(void* dst_host, const void* src_dev, size_t bytes, hipStream_t s) {
    if (!bytes) return;
    std::lock_guard<std::mutex> g(g_stage_m);
    global_stage().d2h(dst_host, src_dev, bytes, s);
    global_stage().quiesced();               // (d2h waited for every piece)
}

}  // namespace rr
