"""Shared test harness: a minimal restatement of the reference's stream windows
and single-threaded Graph loop (src/stream.rs:105,208-217,301-310; src/graph.rs:99-160),
so the oracle blocks and the HIP blocks are driven through IDENTICAL work()
call sequences and their (status, consumed, produced, need) can be compared
exactly and their samples within tolerance."""
from __future__ import annotations

import json
import os

import numpy as np

AGAIN, WAIT_SRC, WAIT_DST, EOF, PENDING = 0, 1, 2, 3, 4
DEFAULT_STREAM_SIZE = 4_096_000  # bytes, src/stream.rs:105

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden():
    with open(os.path.join(GOLDEN_DIR, "reference_known_answers.json")) as f:
        return json.load(f)


def cplx(pairs):
    a = np.asarray(pairs, np.float64)
    return (a[:, 0] + 1j * a[:, 1]).astype(np.complex64)


def max_norm_err(got, ref, scale=None) -> float:
    """SURVEY §8d parity metric: max|got-ref| / max|ref| (scale overrides the denominator)."""
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape, f"length mismatch {got.shape} vs {ref.shape}"
    if ref.size == 0:
        return 0.0
    den = float(np.max(np.abs(ref))) if scale is None else float(scale)
    if den == 0.0:
        den = 1.0
    return float(np.max(np.abs(got.astype(np.complex128) - ref.astype(np.complex128)))) / den


def angle_parity(yg, yo, ro, tol=1e-5, skip=0):
    """Chain-level check of a demodulated stream against the oracle's (say how much of the allowance is
    used).  atan2 amplifies a 1e-5 error of the resampled stream `ro` by 1 / |r|, so the bound per sample is the stage
    bound propagated through it, tol pi + eps / |r[m]| + eps / |r[m+1]| with eps = tol max|r|.  Returns
      used        = max over samples of |d angle| / bound          (must be <= 1)
      above_plain = fraction of samples (after `skip`) whose error exceeds the PLAIN tol pi
      max_err_pi  = the largest error in units of pi"""
    yg = np.asarray(yg, np.float64); yo = np.asarray(yo, np.float64)
    assert yg.shape == yo.shape and len(yg) > 0, (yg.shape, yo.shape)
    mag = np.abs(np.asarray(ro).astype(np.complex128))
    eps = tol * float(np.max(mag))
    bound = (tol * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30))[:len(yg)]
    d = np.abs(yg - yo)
    d = np.minimum(d, 2 * np.pi - d)                              # +-pi wrap
    return {"used": float(np.max(d / bound)), "above_plain": float(np.mean(d[skip:] > tol * np.pi)) if len(d) > skip else 0.0,
            "max_err_pi": float(np.max(d) / np.pi)}


class Ring:
    """A stream as the blocks see it: `buf` = readable window, `free()` = writable window."""

    def __init__(self, dtype, nbytes=DEFAULT_STREAM_SIZE):
        self.dtype = np.dtype(dtype)
        self.cap = nbytes // self.dtype.itemsize
        self.buf = np.zeros(0, self.dtype)

    def free(self):
        return self.cap - len(self.buf)

    def push(self, a):
        assert len(a) <= self.free()
        self.buf = np.concatenate([self.buf, np.asarray(a, self.dtype)])

    def consume(self, n):
        assert n <= len(self.buf)
        self.buf = self.buf[n:]


def run_chain(blocks, x, stream_bytes=DEFAULT_STREAM_SIZE, log=None, max_rounds=1_000_000):
    """VectorSource(x) -> blocks... -> VectorSink, scheduled like Graph::run:
    round-robin, each block's work() once per round, until a full round makes no progress.
    Returns the concatenated sink contents."""
    rings = [Ring(blocks[0].in_dtype, stream_bytes)] + [Ring(b.out_dtype, stream_bytes) for b in blocks]
    x = np.asarray(x, blocks[0].in_dtype)
    pos = 0
    sink = []
    for _ in range(max_rounds):
        progress = False
        take = min(rings[0].free(), len(x) - pos)  # VectorSource::work, src/vector_source.rs:101-146
        if take:
            rings[0].push(x[pos:pos + take]); pos += take; progress = True
        for i, b in enumerate(blocks):
            st, c, p, need, out = b.work(rings[i].buf, rings[i + 1].free())
            if log is not None:
                log.append((i, st, c, p, need))
            rings[i].consume(c)
            rings[i + 1].push(out)
            if c or p or st == AGAIN:
                progress = True
        if len(rings[-1].buf):
            sink.append(rings[-1].buf.copy()); rings[-1].consume(len(rings[-1].buf)); progress = True
        if not progress:
            break
    else:
        raise RuntimeError("run_chain did not terminate")
    dt = blocks[-1].out_dtype
    return np.concatenate(sink) if sink else np.zeros(0, dt)


def signal_source_complex(samp_rate, freq, amplitude, n, state=None):
    """SignalSourceComplex iterator (src/signal_source.rs:39-52); test-side source restatement.
    `state` = [current] carried between calls."""
    rad = 2.0 * np.pi * float(np.float32(freq)) / float(np.float32(samp_rate))
    cur = 0.0 if state is None else state[0]
    out = np.zeros(n, np.complex64)
    twopi = 2.0 * np.pi
    for i in range(n):
        cur = (cur + rad) % twopi
        out[i] = np.float32(amplitude) * np.complex64(complex(np.float32(np.sin(cur)), np.float32(np.sin(cur - np.pi / 2.0))))
    if state is not None:
        state[0] = cur
    return out


def signal_source_complex_fast(samp_rate, freq, amplitude, n):
    """Vectorised variant for big windows (phase accumulated in f64 without the per-sample
    fmod; differs from the iterator only by f64 rounding of the phase)."""
    rad = 2.0 * np.pi * float(np.float32(freq)) / float(np.float32(samp_rate))
    cur = np.mod(rad * np.arange(1, n + 1, dtype=np.float64), 2.0 * np.pi)
    return (np.float32(amplitude) * (np.sin(cur).astype(np.float32) + 1j * np.sin(cur - np.pi / 2).astype(np.float32))).astype(np.complex64)


def knob(rr, monkeypatch, **opts):
    """Build every block this test creates from now on with the given rr_build_opts overrides (rustradio_amd.build_options
    keys: fir_path, fir_prune, fir_half, fir_cfg, fft_log2f, fft_no_split, fftfloat_complex, fm_full, fm_poly,
    dstream_no_vmm, fir_poly); undone by monkeypatch at the end of the test.  The oracle ignores them."""
    monkeypatch.setattr(rr, "_build_opts", dict(rr._build_opts, **opts))


# ---- one block under a Graph::run-style loop on host windows: pageable (staged copies) and page-locked (zero copy) ----
_ARENAS = {}


def _arena(rr, which, nbytes):
    """Two page-locked arenas per process, registered ONCE (rr_host_register) and never unregistered — what the shim does
    with a stream's ring: a page-aligned mapping of whole pages (rr.host_ring), the only kind of range the library runs
    zero-copy on (csrc/blocks.cpp "WHICH ranges run zero-copy")."""
    a = _ARENAS.get(which)
    if a is None or a.nbytes < nbytes:
        if a is not None:
            rr.host_unregister(a)
        a = rr.host_ring(max(nbytes, (160 << 20) if which == "out" else (32 << 20)))     # never grown in practice: a second mapping could land on retired pages
        a[:] = 0
        rr.host_register(a)
        _ARENAS[which] = a
    return a[:nbytes]


def drive_registered(rr, blk, x, in_cap, out_cap):
    """Graph::run around one block on PAGE-LOCKED rings (rr_host_register, what the shim does once per stream): windows are
    slices of the two registered arenas, work_into() on them — the zero-copy path of Block::work_host.  Like a ring's, the
    windows START ANYWHERE: the read window moves on by what was consumed (element-aligned only: 1 byte for the RTL-SDR
    stream, 4 for Float), the write window begins at a different odd offset on every call."""
    nw = int(rr.lib().rr_block_out_windows(blk._h))
    ring_in = _arena(rr, "in", (3 * in_cap + 16) * blk.in_dtype.itemsize).view(blk.in_dtype)
    ring_out = _arena(rr, "out", (nw * out_cap + 16) * blk.out_dtype.itemsize).view(blk.out_dtype)
    rpos, have, pos, outs, log = 3, 0, 0, [], []
    for k in range(200_000):
        if rpos + in_cap > len(ring_in):                          # the ring "wraps": move what is left to another odd start
            ring_in[5:5 + have] = ring_in[rpos:rpos + have].copy(); rpos = 5
        take = min(in_cap - have, len(x) - pos)
        ring_in[rpos + have:rpos + have + take] = x[pos:pos + take]; have += take; pos += take
        wo = (7 * k + 1) % 13
        if k == 0:                                                # the path under test IS the in-place one
            assert rr.host_window_in_place(ring_in[rpos:rpos + max(have, 1)]) and rr.host_window_in_place(ring_out[wo:wo + nw * out_cap])
        st, c, p, need = blk.work_into(ring_in[rpos:rpos + have], ring_out[wo:], out_cap)
        log.append((st, c, p, need))
        rpos += c; have -= c
        outs.append(ring_out[wo:wo + nw * out_cap].reshape(nw, out_cap)[:, :p].copy())
        if take == 0 and c == 0 and p == 0:
            break
    else:
        raise AssertionError("no termination")
    return np.concatenate(outs, axis=1), log


def drive_pageable(blk, x, in_cap, out_cap):
    have, pos, outs, log = 0, 0, [], []
    ring = np.zeros(0, blk.in_dtype)
    for _ in range(200_000):
        take = min(in_cap - len(ring), len(x) - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, out_cap)
        log.append((st, c, p, need))
        ring = ring[c:]
        outs.append(np.atleast_2d(out))
        if take == 0 and c == 0 and p == 0:
            break
    else:
        raise AssertionError("no termination")
    return np.concatenate(outs, axis=1), log
