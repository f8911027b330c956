"""Accuracy of the transform on the teacher-forced level features of a 512x512 five-level frame, as a function of the
eigensolver's convergence tolerance (WCT_JACOBI_CONV_TOL, read once per process): prints rel. error vs the oracle and
the sweeps used, per level.  usage: python tests/probes/wct_tol_probe.py [cache.npz]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from wct_tf_amd import _lib
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, synthetic_image, RELU_TARGETS

cache = sys.argv[1] if len(sys.argv) > 1 else '/tmp/wct_levels.npz'
if not os.path.exists(cache):
    w = synthetic_weights(42)
    c, s = synthetic_image(1000, 512, 512), synthetic_image(2000, 512, 512)
    _, levels = oracle.stylize(c, s, w, RELU_TARGETS, alpha=0.8, wct_mode='tf', return_levels=True)
    np.savez(cache, **{'fc%d' % i: l[0] for i, l in enumerate(levels)}, **{'fs%d' % i: l[1] for i, l in enumerate(levels)},
             **{'t%d' % i: l[2] for i, l in enumerate(levels)})
z = np.load(cache)
ctx = Context(0)
line = []
for i, relu in enumerate(RELU_TARGETS):
    fc, fs, t = z['fc%d' % i], z['fs%d' % i], z['t%d' % i]
    c = fc.shape[-1]
    got, sweeps = ctx.transform(fc.reshape(-1, c), fs.reshape(-1, c), 0.8, _lib.WCT_TF, return_sweeps=True)
    e = np.linalg.norm(got.reshape(t.shape).astype(np.float64) - t) / np.linalg.norm(t)
    line.append('%s %.2e %s' % (relu, e, sweeps))
print('tol=%s pw=%s | ' % (os.environ.get('WCT_JACOBI_CONV_TOL', 'default'), os.environ.get('WCT_JACOBI_PW', '1')) + ' | '.join(line))
