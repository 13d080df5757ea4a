"""bench_verify.py — bench.py checks what it timed.

After a workload's timed passes, and outside them, ONE more step runs on FRESH handles of the same constructors (zero
history: the same kernels on the same window of the same input) and `segments` random output segments of `seglen` outputs
are compared with a direct float64 evaluation of the reference's chain on the input slice each of them depends on.  The
evaluation restates the blocks' definitions in numpy (not the oracle: `oracle/` is test infrastructure and bench.py may
touch it only in its cpu_baseline leg):

    fir            out[i] = sum_k taps[k] x[i d + L-1-k]                       /root/reference/src/fir.rs:166-197,488-551
    fir_translate  taps pre-rotated by the f32 recurrence, out[i] *= phase_i   src/fir.rs:430-473
    fft            out[k] = sum_j taps[j] x[k-j], x[<0] = 0 (whole-stream view) src/fft_filter.rs:289-355
    rs             out[m] = in[floor(m D / I)]                                  src/rational_resampler.rs:155-213
    demod          out[m] = gain * arg(conj(x[m]) x[m+1])                       src/quadrature_demod.rs:65-109
    hilbert        out[i] = (x[i - (N - N/2)], sum_k th[k] x[i-1-k])            src/hilbert.rs:72-128, src/fir.rs:660-680
    u8             (b - 127) * 0.008 in f32                                     src/rtlsdr_decode.rs:42

Bar: 1e-5 max-normalised for sample streams; for a demodulated stream the stage bound propagated through atan2,
tol pi + eps / |r[m]| + eps / |r[m+1]| with eps = tol max|r| (tests/harness.py angle_parity).
"""
from __future__ import annotations

import numpy as np

TOL = 1e-5


def hilbert_taps_f64(ntaps):
    """fir::hilbert(Hamming window) as the reference forms it, in f32 steps (src/fir.rs:660-680, src/window.rs hamming)"""
    import rustradio_amd as rr
    return rr.hilbert_taps(rr.make_window(rr.WIN_HAMMING, ntaps)).astype(np.float64)


def rotated_taps_and_phases(taps, deci, fs, freq, n_phases, replay=True):
    """FirFilter::translate (src/fir.rs:430-473): taps pre-rotated and the output rotator, both by f32 recurrences in
    num-complex operation order -> (complex64 taps, phases[0..n_phases)); replay=False: the library's opt-in MODEL rotator,
    the closed form cos/sin(first + m step) in f64 (off the reference's recurrence by ~3e-8 m)"""
    f32 = np.float32
    step = 2.0 * np.pi * float(f32(freq)) / float(f32(fs))
    sr, si = f32(np.cos(step)), f32(np.sin(step))
    t = np.asarray(taps, np.complex64).copy()
    pr, pi = f32(1.0), f32(0.0)
    for k in range(len(t)):
        ar, ai = f32(t[k].real), f32(t[k].imag)
        t[k] = complex(f32(ar * pr) - f32(ai * pi), f32(ar * pi) + f32(ai * pr))
        pr, pi = f32(f32(pr * sr) - f32(pi * si)), f32(f32(pr * si) + f32(pi * sr))
    first = -step * (len(t) - 1)
    ostep = -step * deci
    if not replay:
        ang = first + ostep * np.arange(n_phases, dtype=np.float64)
        return t, np.cos(ang) + 1j * np.sin(ang)
    ph = np.zeros(n_phases, np.complex64)
    pr, pi = f32(np.cos(first)), f32(np.sin(first))
    sx, sy = f32(np.cos(ostep)), f32(np.sin(ostep))
    for m in range(n_phases):
        ph[m] = complex(pr, pi)
        pr, pi = f32(f32(pr * sx) - f32(pi * sy)), f32(f32(pr * sy) + f32(pi * sx))
    return t, ph


class _Chain:
    """lazy f64 evaluation of a stage list over index ranges; indices < 0 of any stream read as zero"""
    ROT_LIMIT = 65536          # the replayed rotator is checkable only where its recurrence is replayed here

    def __init__(self, stages, src_get):
        self.stages, self.src_get = list(stages), src_get
        self._rot = {}

    def get(self, i, a, b):
        """outputs [a, b) of stage i (i = -1: the source)"""
        a0 = max(a, 0)
        if b <= a0:
            return np.zeros(b - a, np.complex128)
        y = self.src_get(a0, b) if i < 0 else self._run(i, a0, b)
        return y if a0 == a else np.concatenate([np.zeros(a0 - a, y.dtype), y])

    def _run(self, i, a, b):
        st = self.stages[i]
        kind = st[0]
        up = lambda ia, ib: self.get(i - 1, ia, ib)          # noqa: E731
        if kind == "u8":
            return up(a, b)                                  # (the source already decoded the bytes)
        if kind == "fft":
            t = np.asarray(st[1]).astype(np.complex128)
            L = len(t)
            x = up(a - L + 1, b)
            return np.convolve(x, t)[L - 1:L - 1 + (b - a)]
        if kind in ("fir", "fir_translate"):
            taps, d = st[1], st[2]
            rot = None
            if kind == "fir_translate":
                key = i
                if key not in self._rot:
                    self._rot[key] = rotated_taps_and_phases(taps, d, st[3], st[4], self.ROT_LIMIT, st[5])
                taps, rot = self._rot[key]
                assert b <= self.ROT_LIMIT
            t = np.asarray(taps).astype(np.complex128 if np.iscomplexobj(taps) else np.float64)
            L = len(t)
            x = up(a * d, (b - 1) * d + L)
            y = np.convolve(x, t)[L - 1::d][:b - a]
            return y if rot is None else y * rot[a:b].astype(np.complex128)
        if kind == "rs":
            I, D = st[1], st[2]
            ia, ib = (a * D) // I, ((b - 1) * D) // I + 1
            x = up(ia, ib)
            return x[(np.arange(a, b, dtype=np.int64) * D) // I - ia]
        if kind == "demod":
            x = up(a, b + 1)
            return st[1] * np.angle(np.conj(x[:-1]) * x[1:])
        if kind == "hilbert":
            N = st[1]
            th = hilbert_taps_f64(N)
            x = np.real(up(a - N, b)).astype(np.float64)            # x[p], p = a-N .. b-1
            im = np.convolve(x, th)[N - 1:N - 1 + (b - a)]          # position i-1 -> index i-1-(a-N)
            re = x[N // 2:N // 2 + (b - a)]                          # position i-(N-N/2) -> index i-(N-N/2)-(a-N)
            return re + 1j * im
        raise ValueError(kind)


def _source_getter(w):
    buf = w.bufs[0]
    dt = np.dtype(w.blocks[0].in_dtype)
    if dt == np.uint8:
        def get(a, b):
            raw = buf[2 * a:2 * b].cpu().numpy().astype(np.float32)
            v = (raw - np.float32(127.0)) * np.float32(0.008)
            return v[0::2].astype(np.float64) + 1j * v[1::2].astype(np.float64)
    elif dt == np.complex64:
        def get(a, b):
            return buf[2 * a:2 * b].cpu().numpy().view(np.complex64).astype(np.complex128)
    else:
        def get(a, b):
            return buf[a:b].cpu().numpy().astype(np.float64)
    return get


def verify(w, stream, segments=16, seglen=256, seed=0xBE7C4):
    """-> {"ok", "segments", "seglen", "max_err", "tol", "metric", "produced"} for workload `w` (see the module docstring)."""
    import torch
    old = w.blocks
    w.blocks = w.make_blocks()                   # fresh handles: zero history, the same kernels and window
    try:
        w.dom_units = 0
        w.step(stream.cuda_stream)
        torch.cuda.synchronize()
        p = int(w.last_p)
    finally:
        w.blocks = old
    out = w.bufs[-1]
    out_dt = np.dtype(w.blocks[-1].out_dtype)
    cap = w.caps[-1]
    rng = np.random.default_rng(seed)
    get = _source_getter(w)
    demod = False
    translate = False
    plans = []
    for s in range(segments):
        ch = s % w.n_windows if w.n_windows > 1 else 0
        stages = w.ref_of(ch) if w.n_windows > 1 else w.ref
        demod = stages[-1][0] == "demod"
        translate = any(st[0] == "fir_translate" for st in stages)
        hi = min(p, _Chain.ROT_LIMIT) if translate else p
        if hi < seglen + 1:
            return {"ok": False, "segments": 0, "why": f"only {p} outputs produced"}
        a = 0 if s == 0 else (hi - seglen - (1 if demod else 0)) if s == 1 else int(rng.integers(0, hi - seglen - 1))
        plans.append((ch, stages, a))
    refs, rmag, gots = [], [], []
    for ch, stages, a in plans:
        chain = _Chain(stages, get)
        last = len(stages) - 1
        if demod:
            r = chain.get(last - 1, a, a + seglen + 1)           # the resampled stream the demodulator sees
            rmag.append(np.abs(r))
            refs.append(stages[-1][1] * np.angle(np.conj(r[:-1]) * r[1:]))
        else:
            refs.append(chain.get(last, a, a + seglen))
        base = ch * cap
        if out_dt == np.complex64:
            g = out[2 * (base + a):2 * (base + a + seglen)].cpu().numpy().view(np.complex64).astype(np.complex128)
        else:
            g = out[base + a:base + a + seglen].cpu().numpy().astype(np.float64)
        gots.append(g)
    worst = 0.0
    if demod:
        eps = TOL * max(float(m.max()) for m in rmag)
        for g, r, m in zip(gots, refs, rmag):
            d = np.abs(g - r)
            d = np.minimum(d, 2 * np.pi - d)
            bound = TOL * np.pi + eps / np.maximum(m[:-1], 1e-30) + eps / np.maximum(m[1:], 1e-30)
            worst = max(worst, float(np.max(d / bound)))
        ok, metric, tol = worst <= 1.0, "max |d angle| / propagated bound (tol pi + eps/|r[m]| + eps/|r[m+1]|, eps = 1e-5 max|r|)", 1.0
    else:
        scale = max(float(np.max(np.abs(r))) for r in refs) or 1.0
        for g, r in zip(gots, refs):
            worst = max(worst, float(np.max(np.abs(g - r))) / scale)
        ok, metric, tol = worst <= TOL, "max |y - y_f64| / max |y_f64| over the segments", TOL
    res = {"ok": bool(ok and np.isfinite(worst)), "segments": segments, "seglen": seglen, "max_err": float(worst), "tol": tol,
           "metric": metric, "produced": p, "how": "one step on fresh handles after the timed passes; f64 direct evaluation of the chain"}
    if translate:
        res["note"] = (f"segments lie in the first {_Chain.ROT_LIMIT} outputs: the rotator is a sequential f32 recurrence, replayed here only "
                       "that far (bit-exact replay over 1e7 outputs: tests/test_gpu_parity.py)")
    return res
