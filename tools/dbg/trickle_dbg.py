import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import rustradio_amd as rr
from oracle import pyoracle as orc
from harness import run_chain
from test_gpu_trickle import _fm, _drive

taps = orc.low_pass_complex(2.4e6, 100e3, 2.4e3)
x = _fm(40000, 11)
chain = [orc.FftFilter(taps), orc.RationalResampler(5, 1), orc.QuadratureDemod(1.0)]
yo = run_chain(chain, x)
for in_cap, out_cap in [(512000, 1 << 20), (9000, 1 << 20), (512000, 18000), (9000, 18000), (9000, 30000)]:
    blk = rr.FmChain(taps, 5, 1, 1.0)
    yg, log = _drive(blk, x, in_cap, out_cap)
    d = np.abs(yg[0] - yo[:yg.shape[1]]); d = np.minimum(d, 2 * np.pi - d)
    bad = np.nonzero(d > 1e-3)[0]
    print(in_cap, out_cap, yg.shape, len(yo), "bad", len(bad), bad[:10], bad[-5:] if len(bad) else "", log[:12])
ro = run_chain(chain[:2], x)
blk = rr.FmChain(taps, 5, 1, 1.0)
yg, log = _drive(blk, x, 512000, 1 << 20)
mag = np.abs(ro.astype(np.complex128))
eps = 1e-5 * mag.max()
bound = (1e-5 * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30))[:len(yo)]
d = np.abs(yg[0].astype(np.float64) - yo); d = np.minimum(d, 2 * np.pi - d)
w = np.argsort(d / bound)[-8:]
for i in w:
    print(i, d[i], bound[i], mag[i], mag[i + 1], yg[0][i], yo[i], ro[i], ro[i + 1])
print("max mag", mag.max())
# the filter stage alone
f = rr.FftFilter(taps)
st, c, p, need, yf = f.work(x, 1 << 20)
yfo = run_chain([orc.FftFilter(taps)], x)
e = np.abs(yf[:len(yfo)] - yfo)
print("filter stage: max abs err", e.max(), "at", e.argmax(), "max|y|", np.abs(yfo).max(), "first |y|", np.abs(yfo[:8]), "err first", e[:8])
