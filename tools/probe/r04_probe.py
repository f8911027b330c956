"""Per-call timing and sweep counts of wct_eigh (the plain eigensolver entry point) under the WCT_JACOBI_R4 switch:
every call timed on its own (prof events of the solver class), the sweeps each matrix took, the residual of the
decomposition -- to tell a slow call from a call that ran more sweeps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd._lib import WCTNotConverged
def spd(rng, c, n):
    x = np.maximum(rng.standard_normal((n, c)) @ (rng.standard_normal((c, c)) / np.sqrt(c)), 0) * 10.0 ** rng.uniform(-1, 1, c)
    x = x - x.mean(0)
    return (x.T @ x / (n - 1)).astype(np.float32)
ctx = Context(0)
rng = np.random.default_rng(0)
for c, nmat in [(512, 64), (512, 16), (512, 2), (256, 64), (256, 2)]:
    mats = np.stack([spd(rng, c, 4 * c) for _ in range(nmat)])
    times, sw = [], None
    for it in range(6):
        ctx.prof_reset(); ctx.prof_enable(True)
        try:
            w, v, sw = ctx.eigh(mats, return_sweeps=True)
            status = 'ok'
        except WCTNotConverged as e:
            status = 'NOCONV'
            sw = list(ctx.last_sweeps)
        ctx.prof_enable(False)
        times.append(ctx.prof_read()['jacobi']['ms'])
    a = mats[0].astype(np.float64)
    res = np.abs(v[0].astype(np.float64) @ np.diag(w[0].astype(np.float64)) @ v[0].T - a).max() / np.abs(a).max() if status == 'ok' else float('nan')
    print('%s C=%d nmat=%d: %s  ms per call %s  sweeps min %d max %d  |V W V^T - A| %.2e' %
          (os.environ.get('TAG', ''), c, nmat, status, ' '.join('%.2f' % t for t in times), min(sw), max(sw), res), flush=True)
