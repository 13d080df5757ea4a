// compose.cpp — compositions of blocks INSIDE the library, behind one handle and one work() call.
//
// The fused chains (FmChain, FmMulti, AudioChain: one kernel per call) cover what their tile kernels cover: filters up to
// 16383 / 4094 / 3584 taps, the atan2 demodulators.  The reference has no such limits (fft_filter.rs:36-42 sizes its
// transform from any tap count), so a constructor must not refuse a shape the three separate blocks would take
// ("no constructor cliffs").  What the fused kernels do not cover runs here as the UNFUSED composition —
// the same GPU blocks a graph would wire by hand (examples/rtl_fm.rs:381-419), chained through device-resident
// buffers of the reference's stream capacity (stream.rs:105), driven round-robin like Graph::run (graph.rs:108-154)
// inside ONE work() call: same handle type, same window protocol, whole-stream output identical to the separate blocks.
// Also here: FastFM (quadrature_demod.rs:144-165) as the demodulator of a chain (RR_DEMOD_FASTFM).
#include <algorithm>

#include "blocks.hpp"

namespace rr {

Series::Series(const char* nm, std::vector<std::unique_ptr<Block>> blocks)
    : Block(nm, blocks.front()->in_es, blocks.back()->out_es), b(std::move(blocks)) {
    link.reset(new Link[b.size()]);
    for (size_t i = 0; i + 1 < b.size(); i++) {
        if (b[i]->out_es != b[i + 1]->in_es) throw Error("Series: element sizes of adjacent blocks differ");
        link[i].es = b[i]->out_es;
        link[i].cap = 4096000 / link[i].es;                           // stream.rs:105,336-339
        // ... but never smaller than what one work() of a neighbour needs at once: an FftFilter wants room for / a window of
        // `nsamples` (> 512000 beyond 262144 taps), and a link that can never hold it would stall the chain for good
        for (Block* nb : {b[i].get(), b[i + 1].get()})
            if (auto* ff = dynamic_cast<FftFilter*>(nb)) link[i].cap = std::max<size_t>(link[i].cap, 2 * ff->nsamples);
        for (auto& d : link[i].buf) d.reserve(link[i].cap * link[i].es);
    }
}

int Series::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed, size_t* produced,
                     size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    const size_t m = b.size();
    std::vector<int> last(m, RR_AGAIN);       // status and need of each block's most recent work()
    std::vector<size_t> lneed(m, 0);
    prof_begin(s);
    // rounds of one work() per block, in order, until a round moves nothing (Graph::run's loop over a straight chain)
    for (int round = 0; round < 1 << 20; round++) {
        bool progress = false;
        for (size_t i = 0; i < m; i++) {
            const unsigned char* ip;
            size_t in_n;
            if (i == 0) { ip = static_cast<const unsigned char*>(in) + *consumed * in_es; in_n = in_len - *consumed; }
            else { Link& L = link[i - 1]; ip = L.buf[L.cur].p; in_n = L.len; }
            unsigned char* op;
            size_t room;
            if (i + 1 == m) { op = static_cast<unsigned char*>(out) + *produced * out_es; room = out_cap - *produced; }
            else { Link& L = link[i]; op = L.buf[L.cur].p + L.len * L.es; room = L.cap - L.len; }
            size_t c = 0, p = 0, nd = 0;
            const int st = b[i]->work_dev(ip, in_n, op, room, &c, &p, &nd, s);
            if (st == RR_ERR) { prof_end(s); return st; }
            last[i] = st; lneed[i] = nd;
            if (i == 0) *consumed += c;
            else if (c) {
                // consume(c) on a linear buffer: what is left moves to the front of the other buffer (a few samples of
                // carry — QuadratureDemod keeps one, FastFM none, the filters take whole windows)
                Link& L = link[i - 1];
                const size_t left = L.len - c;
                if (left) RR_HIP(hipMemcpyAsync(L.buf[L.cur ^ 1].p, L.buf[L.cur].p + c * L.es, left * L.es, hipMemcpyDeviceToDevice, s));
                L.cur ^= 1;
                L.len = left;
            }
            if (i + 1 == m) *produced += p;
            else link[i].len += p;
            if (c || p) progress = true;
        }
        if (!progress) break;
    }
    prof_end(s);
    if (last[m - 1] == RR_WAIT_DST) { *need = lneed[m - 1]; return RR_WAIT_DST; }     // the caller's output window is the limit
    if (last[0] == RR_WAIT_SRC) {
        // The caller's input is the limit.  What it must offer is what the block that is actually starved needs, carried
        // back through the blocks in front of it — the fused form of the same chain reports exactly this
        // (FmChain: nsamples - pending + front, in bytes for u8) — not the few samples / bytes block 0 alone asks for,
        // which would wake a scheduler for nothing.
        size_t nd = lneed[0];
        for (size_t j = 1; j < m; j++) {
            if (last[j] != RR_WAIT_SRC || lneed[j] <= link[j - 1].len) continue;
            size_t want = lneed[j] - link[j - 1].len;           // more elements block j wants to see in its window
            bool ok = true;
            for (size_t i = j; i-- > 0 && ok;) {                 // ... in units of the stream in front of block i
                if (dynamic_cast<RtlSdrDecode*>(b[i].get())) want *= 2;                       // two bytes per sample
                else if (auto* fc = dynamic_cast<FirC32*>(b[i].get())) {
                    if (fc->pl.d != 1) ok = false;
                    else if (i == 0) want += (size_t)fc->pl.L - 1;                            // the window keeps ntaps - 1 (fir.rs:496-549)
                } else if (i > 0 || !dynamic_cast<FftFilter*>(b[i].get())) ok = false;        // other blocks: keep block 0's own figure
            }
            if (ok) nd = std::max(nd, want);
            break;
        }
        *need = nd;
        return RR_WAIT_SRC;
    }
    if (*consumed == 0 && *produced == 0) {
        // nothing moved and neither end is the limit: an inner link is both too full for its writer and too short for its
        // reader.  Saying AGAIN would spin the caller (block.rs:21-40: AGAIN is not for polling).
        throw Error("Series: an inner stream can hold neither what its writer must emit at once nor what its reader needs");
    }
    return RR_AGAIN;
}

bool Series::eof(bool src_eof) {
    bool e = src_eof;
    for (auto& blk : b) e = blk->eof(e);
    return e;
}

Parallel::Parallel(const char* nm, std::vector<std::unique_ptr<Block>> channels)
    : Block(nm, channels.front()->in_es, channels.front()->out_es), ch(std::move(channels)) {
    for (auto& c : ch)
        if (c->in_es != in_es || c->out_es != out_es) throw Error("Parallel: channels of different stream types");
    zero_copy_in = false;                            // (every channel reads the whole window: upload it once)
}

// `out` holds ch.size() windows of out_cap elements; every channel sees the same input window.  The channels are built
// from the same shape (tap count, ratio), so their bookkeeping is identical: one (status, consumed, produced, need).
int Parallel::work_dev(const void* in, size_t in_len, void* out, size_t out_cap, size_t* consumed, size_t* produced,
                       size_t* need, hipStream_t s) {
    *consumed = *produced = *need = 0;
    int st0 = RR_AGAIN;
    // The channels are independent chains on one shared window: they go out on a small pool of streams forked from and
    // joined back into the caller's stream, so that a chain whose launch does not fill the chip (short windows, the single
    // workgroup tail of a launch) overlaps with its neighbours instead of queueing behind them.
    const size_t NP = std::min<size_t>(ch.size(), POOL);
    if (NP > 1 && !pool[0]) {
        for (size_t i = 0; i < POOL; i++) {
            RR_HIP(hipStreamCreateWithFlags(&pool[i], hipStreamNonBlocking));
            RR_HIP(hipEventCreateWithFlags(&joined[i], hipEventDisableTiming));
        }
        RR_HIP(hipEventCreateWithFlags(&forked, hipEventDisableTiming));
    }
    prof_begin(s);
    // Whatever happens below — a channel's work_dev throwing, the channels disagreeing — the pool streams are joined back into
    // the caller's stream before this call returns: work the channels already enqueued must not run on past the caller's next
    // step (its copy of the output windows, its next window, the destruction of this block) un-ordered.
    struct Join {
        Parallel& p; hipStream_t s; size_t np; bool armed = false;
        void run() {
            if (!armed) return;
            armed = false;
            for (size_t i = 0; i < np; i++)
                if (hipEventRecord(p.joined[i], p.pool[i]) != hipSuccess || hipStreamWaitEvent(s, p.joined[i], 0) != hipSuccess)
                    (void)hipStreamSynchronize(p.pool[i]);             // (the event path failed: wait here instead)
        }
        ~Join() { run(); }
    } join{*this, s, NP};
    if (NP > 1) {
        RR_HIP(hipEventRecord(forked, s));
        for (size_t i = 0; i < NP; i++) RR_HIP(hipStreamWaitEvent(pool[i], forked, 0));
        join.armed = true;
    }
    for (size_t c = 0; c < ch.size(); c++) {
        size_t cc = 0, pp = 0, nn = 0;
        const int st = ch[c]->work_dev(in, in_len, static_cast<unsigned char*>(out) + c * out_cap * out_es, out_cap, &cc, &pp, &nn,
                                       NP > 1 ? pool[c % NP] : s);
        if (c == 0) { st0 = st; *consumed = cc; *produced = pp; *need = nn; }
        else if (st != st0 || cc != *consumed || pp != *produced || nn != *need)
            throw Error("Parallel: channels of one shape disagree on the window protocol");
    }
    join.run();
    prof_end(s);
    return st0;
}

Parallel::~Parallel() {
    (void)hipSetDevice(device);
    for (size_t i = 0; i < POOL; i++) {
        if (pool[i]) { (void)hipStreamSynchronize(pool[i]); (void)hipStreamDestroy(pool[i]); }
        if (joined[i]) (void)hipEventDestroy(joined[i]);
    }
    if (forked) (void)hipEventDestroy(forked);
}

bool Parallel::eof(bool src_eof) {
    bool e = true;
    for (auto& c : ch) e = c->eof(src_eof) && e;
    return e;
}

// ---- factories behind the C ABI's fused-chain constructors -------------------------------------------------------------
static std::unique_ptr<Block> demodulator(float gain, int mode) {
    if (mode == RR_DEMOD_FASTFM) return std::unique_ptr<Block>(new FastFM());
    return std::unique_ptr<Block>(new QuadDemod(gain, mode));
}

static Block* fm_chain_unfused(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool u8,
                               const rr_c32* fir_taps, size_t fir_ntaps) {
    std::vector<std::unique_ptr<Block>> v;
    if (u8) v.emplace_back(new RtlSdrDecode());
    if (fir_taps) v.emplace_back(new FirC32(fir_taps, fir_ntaps, 1, false, 0.0f, 0.0f));
    v.emplace_back(new FftFilter(taps, ntaps));
    v.emplace_back(new Resampler(interp, deci, sizeof(cf)));
    v.push_back(demodulator(gain, mode));
    return new Series(fir_taps ? "FirFilter>FftFilter>RationalResampler>demod (unfused)"
                      : u8 ? "RtlSdrDecode>FftFilter>RationalResampler>demod (unfused)" : "FftFilter>RationalResampler>demod (unfused)",
                      std::move(v));
}

Block* make_fm_chain(const rr_c32* taps, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool u8,
                     const rr_c32* fir_taps, size_t fir_ntaps) {
    if (mode != RR_DEMOD_FASTFM) {
        try {
            return new FmChain(taps, ntaps, interp, deci, gain, mode, u8, 14, fir_taps, fir_ntaps);
        } catch (const NotFusedShape&) {
            // not a shape of the fused kernels (more than 16383 taps, a decimation beyond the tile, ...): the unfused
            // composition takes everything the three blocks take.  Anything else (HIP failure, bad argument) propagates.
        }
    }
    return fm_chain_unfused(taps, ntaps, interp, deci, gain, mode, u8, fir_taps, fir_ntaps);
}

Block* make_fm_multi(const rr_c32* taps, size_t nchan, size_t ntaps, size_t interp, size_t deci, float gain, int mode, bool u8) {
    if (nchan == 0 || nchan > 4096) throw Error("FmMulti: channel count must be 1..4096");
    if (mode != RR_DEMOD_FASTFM) {
        try {
            return new FmMulti(taps, nchan, ntaps, interp, deci, gain, mode, u8);
        } catch (const NotFusedShape&) {
        }
    }
    // beyond the shared-forward-transform kernels (more than 4094 taps, ...): one chain per channel on the shared window —
    // each still fused where its own kernels reach (split tiles up to 16383 taps)
    std::vector<std::unique_ptr<Block>> ch;
    for (size_t c = 0; c < nchan; c++) ch.emplace_back(make_fm_chain(taps + c * ntaps, ntaps, interp, deci, gain, mode, u8, nullptr, 0));
    return new Parallel(u8 ? "RtlSdrDecode>Tee>N x chain (per channel)" : "Tee>N x chain (per channel)", std::move(ch));
}

Block* make_audio_chain(const float* taps, size_t ntaps, size_t interp, size_t deci, float scale) {
    try {
        return new AudioChain(taps, ntaps, interp, deci, scale);
    } catch (const NotFusedShape&) {
    }
    std::vector<std::unique_ptr<Block>> v;
    v.emplace_back(new FftFilterFloat(taps, ntaps));
    v.emplace_back(new Resampler(interp, deci, sizeof(float)));
    v.emplace_back(new MultiplyConst(sizeof(float), scale, 0.0f));
    return new Series("FftFilterFloat>RationalResampler>MultiplyConst (unfused)", std::move(v));
}

}  // namespace rr
