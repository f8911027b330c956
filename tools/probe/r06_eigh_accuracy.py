"""Accuracy of wct_eigh on the covariances of the style-swap test features (C = 512): eigenvalues against LAPACK in float64
(relative, per eigenvalue), residual ||A V - V L|| / ||A||, orthogonality ||V^T V - I||_max -- for an A-B of two builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_features

ctx = Context(0)
for (c, h, w, seed) in [(512, 32, 32, 90), (512, 32, 32, 95), (512, 22, 22, 90), (256, 64, 64, 7)]:
    f = np.float64(synthetic_features(seed + c, c, h, w, 1.5)).reshape(-1, c)
    A = np.cov(f.T) + 1e-8 * np.eye(c)
    ev, V, sw = ctx.eigh(np.float32(A), return_sweeps=True)
    ev = np.float64(ev[0]); V = np.float64(V[0])
    ref = np.linalg.eigvalsh(A)
    o = np.argsort(ev)
    rel = np.abs(ev[o] - ref) / np.maximum(ref, 1e-30)
    big = ref > 1e-5
    res = np.linalg.norm(A @ V - V * ev[None, :]) / np.linalg.norm(A)
    orth = np.abs(V.T @ V - np.eye(c)).max()
    # what the transform needs: W = V D^-1/2 V^T over the kept eigenvalues against the float64 one
    k = ev > 1e-5
    W = (V[:, k] / np.sqrt(ev[k])) @ V[:, k].T
    u, s, _ = np.linalg.svd(A); kk = s > 1e-5
    W64 = (u[:, kk] / np.sqrt(s[kk])) @ u[:, kk].T
    print('C=%d N=%d sweeps %s: eigenvalues kept %d, rel err max %.2e median %.2e (smallest kept %.2e / largest %.2e) | residual %.2e | orthogonality %.2e | whitening matrix vs float64 %.2e'
          % (c, h * w, sw, int(big.sum()), rel[big].max(), np.median(rel[big]), ref[big].min(), ref.max(), res, orth, np.linalg.norm(W - W64) / np.linalg.norm(W64)))
ctx.close()
