"""Cases of the WCT fuzz sweep, re-run one by one through the sweep's own judge (tests/test_gpu_fuzz.py::_wct_case): the failing ones a wide run
(tools/gpu_fuzz_wide.sh) reported -- for an A-B of library builds.  usage: python tools/probe/r06_fuzz_cases.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd import _lib
import oracle
import test_gpu_fuzz as tf

CASES = [
    (64, 12, 6, 20, 10, 0.959527218831276, 'tf', 1.2633897218325192, 5076692),
    (128, 12, 25, 25, 23, 0.8727483594839052, 'np', -1.8471459453454355, 38),
    (256, 8, 19, 9, 17, 0.8948437019837017, 'np', -1.7328125291679197, 176),
    (512, 14, 31, 12, 29, 0.9304784057570625, 'tf', -1.0695702389412896, 501),
]
ctx = Context(0)
for case in CASES:
    try:
        tf._wct_case(ctx, *case)
        print('   -> PASS')
    except AssertionError as e:
        print('   -> FAIL %s' % str(e).splitlines()[0][:300])
    c, hc, wc, hs, ws, alpha, mode, log_scale, seed = case
    rng = np.random.default_rng(seed)
    scale = 10.0 ** log_scale
    fc, fs = tf.features(rng, hc * wc, c, scale), tf.features(rng, hs * ws, c, scale * 10.0 ** rng.uniform(-1, 1))
    fn = oracle.wct_np if mode == 'np' else oracle.wct_tf
    got, sw = ctx.transform(fc, fs, alpha, _lib.WCT_NP if mode == 'np' else _lib.WCT_TF, return_sweeps=True)
    sh = (fc.reshape(1, hc, wc, c), fs.reshape(1, hs, ws, c))
    o64 = np.asarray(fn(np.float64(sh[0]), np.float64(sh[1]), alpha, **({'dtype': np.float64} if mode == 'tf' else {}))).reshape(-1, c)
    o32 = np.asarray(fn(*sh, alpha)).reshape(-1, c)
    print('      default kept counts: this path vs float64 %.3e | float32 oracle vs float64 %.3e | this path vs float32 oracle %.3e | sweeps %s' % (
        tf.rel_err(got, o64), tf.rel_err(o32, o64), tf.rel_err(got, o32), [int(x) for x in sw]))
    ev = np.sort(np.linalg.eigvalsh(np.cov(np.float64(fc).T)))[::-1]
    k = int((ev > 1e-5).sum())
    print('      content covariance (float64): largest %.3e, kept %d, around the cut-off: %s' % (ev[0], k, ' '.join('%.3e' % t for t in ev[max(k - 4, 0):k + 4])), flush=True)
ctx.close()
