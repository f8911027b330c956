"""One case of the style-swap sweep (tests/test_gpu_fuzz.py::test_style_swap_random_shapes), taken apart: C = 128, 15 x 18 content, 13 x 7 style
(N = 91 < C on the style side), 3 x 3 patches.  Where do this path's pixels differ from the oracle's, and by what margin did the oracle decide there?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from wct_tf_amd.context import Context
from wct_tf_amd import ops, _lib
from wct_tf_amd.weights import synthetic_features

ctx = Context(0)
for (c, hc, wc, hs, ws, patch, alpha, seed) in [(128, 15, 18, 13, 7, 3, 0.7341779593261556, 189), (128, 15, 18, 13, 7, 1, 0.7341779593261556, 189), (128, 15, 18, 13, 12, 3, 0.7341779593261556, 189)]:
    fc, fs = synthetic_features(seed, c, hc, wc, 1.5), synthetic_features(seed + 1, c, hs, ws, 1.5)
    want, margins = oracle.wct_style_swap(fc, fs, alpha, patch, 1, return_margins=True)
    got = ops.wct_style_swap(fc, fs, alpha, patch, 1, ctx=ctx)
    diff = np.abs(got - want).max(-1)[0] > 1e-3 * np.abs(want).max()
    print('C=%d %dx%d / %dx%d patch %d: rel %.2e, pixels differing %d of %d; oracle margins: min %.2e, below 1e-3: %d, below 1e-2: %d of %d positions' % (
        c, hc, wc, hs, ws, patch, np.linalg.norm(got - want) / np.linalg.norm(want), int(diff.sum()), diff.size, margins.min(), int((margins < 1e-3).sum()), int((margins < 1e-2).sum()), margins.size))
    # a differing pixel (y, x) is covered by the positions (y - patch + 1 .. y, x - patch + 1 .. x): the smallest margin among them
    ys, xs = np.nonzero(diff)
    cover = []
    for y, x in zip(ys, xs):
        m = [margins[yy, xx] for yy in range(max(0, y - patch + 1), min(margins.shape[0], y + 1)) for xx in range(max(0, x - patch + 1), min(margins.shape[1], x + 1))]
        cover.append(min(m))
    if cover:
        print('   smallest oracle margin among the positions covering each differing pixel: median %.2e, max %.2e' % (np.median(cover), max(cover)))
    # the whitened maps: this path's transform accuracy on the two sides (alpha = 1 against an identity-covariance style is not available here;
    # instead the plain transform of content by style, both semantics, as a yardstick)
    o64 = np.asarray(oracle.wct_tf(np.float64(fc), np.float64(fs), 1.0, dtype=np.float64)).reshape(-1, c)
    t = ctx.transform(fc.reshape(-1, c), fs.reshape(-1, c), 1.0, _lib.WCT_TF)
    print('   wct_tf of the same features, alpha 1: this path vs float64 %.2e' % (np.linalg.norm(t - o64) / np.linalg.norm(o64)))
ctx.close()
