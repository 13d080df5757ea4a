"""CPU: the set algebra of FftFilter's in-kernel non-finite pass (csrc/nan_fix.hpp rb_own_tiles / rb_last, round 6) restated in
Python and checked against the reference's definition over random tile sizes, block sizes, tap counts, fused front filters,
poisoned positions and a poisoned previous call — the kernel itself is checked on the GPU against the oracle
(tests/test_gpu_edges_fullsize.py, tests/test_gpu_fuzz.py); this pins WHY it is right, and found a wrong record range before
the GPU did.

Reference (src/fft_filter.rs:326-347): a non-finite input sample of block b (nsamples = S inputs, + `front` samples of a fused
FirFilter) poisons the outputs [b S, (b + 1) S + Lf), Lf = the FftFilter stage's taps.  GPU: a tile of P outputs that read such
a sample has no finite output; the workgroup that owns it rewrites its own outputs (NaN on the reference's set, the
reference-order fold elsewhere) and records what the reference poisons outside the tile; the last workgroup fills the records."""
import numpy as np


def model(P, S, L, front, nblocks, bad_positions, tail_bad):
    Lf, hist = L - front, L - 1 - front
    nfin = nblocks * S
    badv = np.zeros(nfin + L - 1, bool)
    badv[bad_positions] = True

    def blockbad(b):
        return tail_bad if b < 0 else bool(badv[hist + b * S: hist + (b + 1) * S + front].any())
    ref = np.zeros(nfin, bool)
    for b in range(-1, nblocks):
        if blockbad(b):
            ref[max(b * S, 0):min((b + 1) * S + Lf, nfin)] = True
    out = np.zeros(nfin, int)                      # 0 finite (or folded), 1 smeared by a tile, 2 NaN written on purpose
    flagged = []
    for k in range((nfin + P - 1) // P):
        o0, o1 = k * P, min(k * P + P, nfin)
        if badv[o0:o1 + L - 1].any():
            out[o0:o1] = 1
            flagged.append(k)
    recs = []

    def record(a, b):
        a, b = max(a, 0), min(b, nfin)
        if b > a:
            recs.append((a, b))
    for k in flagged:                              # rb_own_tiles
        o0, o1 = k * P, min(k * P + P, nfin)
        b_first, b_last = o0 // S, (o1 - 1) // S
        bad_prev = blockbad(b_first - 1)
        if bad_prev:
            record((b_first - 1) * S, min(o0, b_first * S + Lf))
            record(o1, b_first * S + Lf)
        for b in range(b_first, b_last + 1):
            bad_cur = blockbad(b)
            for m in range(max(b * S, o0), min((b + 1) * S, o1)):
                out[m] = 2 if (bad_cur or (bad_prev and m - b * S < Lf)) else 0
            if bad_cur:
                record(b * S, o0)
                record(o1, (b + 1) * S + Lf)
            bad_prev = bad_cur
    for a, b in recs:                              # rb_last
        out[a:b] = 2
    if tail_bad:
        out[:min(Lf, nfin)] = 2
    gpu = out != 0
    for k in flagged:                              # a fold over a window that holds a poisoned sample is not finite either
        for m in range(k * P, min(k * P + P, nfin)):
            if out[m] == 0 and badv[m:m + L].any():
                gpu[m] = True
    return ref, gpu


def test_in_kernel_pass_reaches_the_references_set():
    rng = np.random.default_rng(0)
    for _ in range(1500):
        P = int(rng.choice([1630, 898, 5726, 100, 3000]))
        S = int(rng.choice([5726, 623, 561]))
        L = min(int(rng.integers(2, P + 1000)), S - 1)
        front = int(rng.integers(0, min(L - 1, 50))) if rng.random() < 0.3 else 0
        nb = int(rng.integers(1, 12))
        hist = L - 1 - front
        pos = rng.integers(hist, nb * S + L - 1, int(rng.integers(0, 4)))
        tail_bad = bool(rng.random() < 0.2)
        if tail_bad and hist > 0 and rng.random() < 0.5:      # (the poisoned sample of the previous call may lie in the carried history)
            pos = np.append(pos, rng.integers(0, hist))
        ref, gpu = model(P, S, L, front, nb, pos, tail_bad)
        assert np.array_equal(ref, gpu), (P, S, L, front, nb, sorted(pos), tail_bad, np.flatnonzero(ref != gpu)[:5])
