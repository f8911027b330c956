"""Why does style-swap at C = 512, 32 x 32 flip patches when the eigensolver's tile update changes?  (a) the transform error of the
same features against the float64 oracle, (b) the margins of the oracle's own patch matches (best vs second-best correlation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from oracle import wct_oracle
from wct_tf_amd.context import Context
from wct_tf_amd import _lib, ops
from wct_tf_amd.weights import synthetic_features

def rel(a, b):
    a = np.float64(a); b = np.float64(b)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))

ctx = Context(0)
for (c, hc, wc, hs, ws) in [(512, 32, 32, 32, 32), (512, 22, 22, 12, 20), (512, 16, 16, 16, 16)]:
    fc = synthetic_features(90 + c, c, hc, wc, 1.5)
    fs = synthetic_features(95 + c, c, hs, ws, 1.5)
    for alpha in (1.0, 0.6):
        got = ctx.transform(fc.reshape(-1, c), fs.reshape(-1, c), alpha, _lib.WCT_TF)
        w32 = np.asarray(oracle.wct_tf(fc, fs, alpha)).reshape(-1, c)
        w64 = np.asarray(oracle.wct_tf(np.float64(fc), np.float64(fs), alpha, dtype=np.float64)).reshape(-1, c)
        print('C=%d %dx%d / %dx%d alpha %.1f: wct_tf GPU vs fp64 oracle %.2e | fp32 oracle vs fp64 %.2e' % (c, hc, wc, hs, ws, alpha, rel(got, w64), rel(w32, w64)))
    want = oracle.wct_style_swap(fc, fs, 0.6, 3, 1)
    got = ops.wct_style_swap(fc, fs, 0.6, 3, 1, ctx=ctx)
    bad = np.abs(got - want).max(-1) > 1e-3 * np.abs(want).max()
    print('   style-swap: rel %.2e, pixels differing %d of %d' % (rel(got, want), int(bad.sum()), bad.size))
ctx.close()

# which matches flipped, and how decided the oracle was about them
ctx = Context(0)
c, hc, wc, hs, ws = 512, 32, 32, 32, 32
fc = synthetic_features(90 + c, c, hc, wc, 1.5)
fs = synthetic_features(95 + c, c, hs, ws, 1.5)
want, margins = oracle.wct_style_swap(fc, fs, 0.6, 3, 1, return_margins=True)
got = ops.wct_style_swap(fc, fs, 0.6, 3, 1, ctx=ctx)
diff = np.abs(got - want).max(-1) > 1e-3 * np.abs(want).max()
cand = [(float(margins[y, x]), y, x) for y in range(margins.shape[0]) for x in range(margins.shape[1]) if diff[y:y + 3, x:x + 3].all()]
print('positions whose whole 3 x 3 footprint differs (margin, y, x):', sorted(cand)[:12])
print('the eight smallest margins of the oracle:', np.sort(margins.ravel())[:8])
ctx.close()
