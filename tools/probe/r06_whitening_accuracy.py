"""How accurate is the WHITENING this path applies?  Transform with alpha = 1 against a style whose covariance is the identity
(orthonormal zero-mean columns x sqrt(N - 1)) and whose mean is zero: the output IS the whitened content.  Compared with float64
whitening, per pixel row -- for an A-B of two builds of the eigensolver."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd import _lib
from wct_tf_amd.weights import synthetic_features

ctx = Context(0)
rng = np.random.default_rng(3)
for (c, h, w, seed) in [(512, 32, 32, 90), (512, 32, 32, 95), (256, 64, 64, 7)]:
    fc = synthetic_features(seed + c, c, h, w, 1.5).reshape(-1, c)
    n = max(2 * c, 1024)
    g = rng.standard_normal((n, c)); g -= g.mean(0)
    q, _ = np.linalg.qr(g)                       # orthonormal, zero-mean columns
    q -= q.mean(0)
    fs = np.float32(q * np.sqrt(n - 1))
    got = np.float64(ctx.transform(fc, fs, 1.0, _lib.WCT_TF))
    x = np.float64(fc); xc = x - x.mean(0)
    cov = xc.T @ xc / (x.shape[0] - 1) + 1e-8 * np.eye(c)
    u, s, _ = np.linalg.svd(cov); k = s > 1e-5
    W = (u[:, k] / np.sqrt(s[k])) @ u[:, k].T
    want = xc @ W.T
    # the style is the identity only to float32 accuracy: colour it exactly as well
    fs64 = np.float64(fs); sc = fs64 - fs64.mean(0)
    cs = sc.T @ sc / (n - 1) + 1e-8 * np.eye(c)
    us, ss, _ = np.linalg.svd(cs)
    want = want @ ((us * np.sqrt(ss)) @ us.T).T + fs64.mean(0)
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    print('C=%d N=%d: whitened content vs float64: rel %.2e overall, per pixel row median %.2e, 99th percentile %.2e, max %.2e'
          % (c, h * w, np.linalg.norm(got - want) / np.linalg.norm(want), np.median(err), np.percentile(err, 99), err.max()))
ctx.close()
