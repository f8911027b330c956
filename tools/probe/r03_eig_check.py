"""Accuracy + timing of the batched eigensolver (wct_eigh) against LAPACK (float64): eigenvalues, residual of the
decomposition, orthogonality, sweeps; WCT_JACOBI_FUSED=0/1 selects the round-2 / look-ahead launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context


def spd(rng, c, n, decades=1.0):
    x = np.maximum(rng.standard_normal((n, c)) @ (rng.standard_normal((c, c)) / np.sqrt(c)), 0) * 10.0 ** rng.uniform(-decades, decades, c)
    x = x - x.mean(0)
    return (x.T @ x / (n - 1)).astype(np.float32)


ctx = Context(0)
rng = np.random.default_rng(0)
print('WCT_JACOBI_FUSED=%s' % os.environ.get('WCT_JACOBI_FUSED', '(default)'))
for c, nmat in [(32, 3), (64, 2), (128, 2), (256, 2), (512, 2), (512, 16), (256, 16), (128, 16), (64, 16), (512, 64)]:
    mats = np.stack([spd(rng, c, 4 * c if i % 2 == 0 else c // 2, 1.0 + (i % 3)) for i in range(nmat)])
    ev, vec, sw = ctx.eigh(mats, return_sweeps=True)
    worst = [0, 0, 0]
    for i in range(min(nmat, 4)):
        a = mats[i].astype(np.float64)
        ref = np.linalg.eigvalsh(a)
        v = vec[i].astype(np.float64)
        worst[0] = max(worst[0], np.abs(np.sort(ev[i]) - ref).max() / ref.max())
        worst[1] = max(worst[1], np.abs(v.T @ a @ v - np.diag(ev[i])).max() / ref.max())
        worst[2] = max(worst[2], np.abs(v.T @ v - np.eye(c)).max())
    ctx.prof_reset(); ctx.prof_enable(True)
    reps = 3
    for _ in range(reps):
        ctx.eigh(mats)
    ctx.prof_enable(False)
    p = ctx.prof_read()['jacobi']
    print('C=%3d nmat=%2d: %.2f ms (events) sweeps %s  eig err %.1e  |V^T A V - D| %.1e  |V^T V - I| %.1e' % (
        c, nmat, p['ms'] / reps, sorted(set(sw)), worst[0], worst[1], worst[2]), flush=True)
