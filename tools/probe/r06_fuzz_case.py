"""One case of the fuzz sweep, reproduced and taken apart: C = 256, 12 x 7 content / 13 x 6 style pixels, feature scale 1e3, wct_np, alpha 0.237
(N << C at a feature scale whose rounding noise, ~1e-7 ||cov|| ~ 1, is five decades ABOVE the reference's 1e-5 cut-off).

For the drawn seed and a few more of the same shape: this path against the float64 oracle, the float32 oracle against the same, and --
through wct_eigh on the float32 covariance NumPy forms -- where this path's rounding-noise eigenvalues of the content covariance lie
(the whitening gain of a kept noise direction is (d + 1e-5)^-1/2: 1 at d = 1, 220 at d = 1e-5)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
from wct_tf_amd.context import Context
from wct_tf_amd import _lib
import test_gpu_fuzz as tf

ctx = Context(0)
c, hc, wc, hs, ws, alpha, mode, log_scale = 256, 12, 7, 13, 6, 0.23694110562368303, 'np', 3.0
nc, ns = hc * wc, hs * ws
for seed in [464496, 1, 2, 3, 4, 5]:
    rng = np.random.default_rng(seed)
    scale = 10.0 ** log_scale
    fc, fs = tf.features(rng, nc, c, scale), tf.features(rng, ns, c, scale * 10.0 ** rng.uniform(-1, 1))
    got, sw = ctx.transform(fc, fs, alpha, _lib.WCT_NP, return_sweeps=True)
    o64 = np.asarray(oracle.wct_np(np.float64(fc.reshape(1, hc, wc, c)), np.float64(fs.reshape(1, hs, ws, c)), alpha)).reshape(nc, c)
    o32 = np.asarray(oracle.wct_np(fc.reshape(1, hc, wc, c), fs.reshape(1, hs, ws, c), alpha)).reshape(nc, c)
    x = np.float32(fc - fc.mean(0, keepdims=True)).T
    cov32 = np.float32(x @ x.T / np.float32(nc - 1))
    w, v = ctx.eigh(cov32[None])
    w, v = w[0], v[0]
    s32 = np.linalg.svd(cov32, compute_uv=False)
    pos = np.sort(w[(w > 1e-5) & (w < 1e3)])
    print('seed %7d: GPU vs float64 oracle %.3e | float32 oracle vs float64 %.3e | sweeps %s' % (
        seed, np.linalg.norm(got - o64) / np.linalg.norm(o64), np.linalg.norm(o32 - o64) / np.linalg.norm(o64), list(np.ravel(sw))))
    print('      wct_eigh of the fp32 covariance: %d eigenvalues > 1e3 (signal), %d in (1e-5, 1e3) [smallest five: %s], %d <= 1e-5 (most negative %.2e); '
          'LAPACK sgesdd noise singular values: smallest five %s' % (
              (w >= 1e3).sum(), pos.size, ' '.join('%.2e' % t for t in pos[:5]), (w <= 1e-5).sum(), w.min(),
              ' '.join('%.2e' % t for t in np.sort(s32)[:5])), flush=True)
    # what the reference's formula gives with THIS path's eigenpairs of that matrix (float64 arithmetic from here on)
    keep = w > 1e-5
    vk = np.float64(v[:, keep])
    white = vk @ np.diag((np.float64(w[keep]) + 1e-5) ** -0.5) @ vk.T @ np.float64(x)
    keep_sig = w > 1e3
    vs_ = np.float64(v[:, keep_sig])
    white_sig = vs_ @ np.diag((np.float64(w[keep_sig]) + 1e-5) ** -0.5) @ vs_.T @ np.float64(x)
    print('      whitened features from those eigenpairs: signal + kept noise directions vs signal only: rel %.3e' % (
        np.linalg.norm(white - white_sig) / np.linalg.norm(white_sig)), flush=True)
ctx.close()
