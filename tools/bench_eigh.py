"""Micro-benchmark of the batched eigensolver (jacobi class timing from the library's HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context

def spd(rng, c, n):
    x = np.maximum(rng.standard_normal((n, c)) @ (rng.standard_normal((c, c)) / np.sqrt(c)), 0) * 10.0 ** rng.uniform(-1, 1, c)
    x = x - x.mean(0)
    return (x.T @ x / (n - 1)).astype(np.float32)

ctx = Context(0)
rng = np.random.default_rng(0)
for c, nmat in [(512, 16), (256, 16), (128, 16), (64, 16)]:
    mats = np.stack([spd(rng, c, 4 * c) for _ in range(nmat)])
    ctx.eigh(mats)
    ctx.prof_reset(); ctx.prof_enable(True)
    reps = 3
    for _ in range(reps):
        ev, vec, sw = ctx.eigh(mats, return_sweeps=True)
    ctx.prof_enable(False)
    p = ctx.prof_read()['jacobi']
    ref = np.linalg.eigvalsh(mats[0].astype(np.float64))
    err = np.abs(np.sort(ev[0]) - ref).max() / ref.max()
    print('C=%d nmat=%d: %.2f ms per call, sweeps %s, eig err %.1e' % (c, nmat, p['ms'] / reps, sorted(set(sw)), err), flush=True)
