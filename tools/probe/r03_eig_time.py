"""Timing of wct_eigh at a fixed sweep count (WCT_JACOBI_MAX_SWEEPS) for experiments with the WCT_JACOBI_* switches."""
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
cases = [(512, 64), (512, 16), (512, 2), (256, 64)] if len(sys.argv) < 2 else [(512, int(sys.argv[1]))]
for c, nmat in cases:
    mats = np.stack([spd(rng, c, 4 * c) for _ in range(nmat)])
    def run():
        try:
            ctx.eigh(mats)
        except WCTNotConverged:
            pass
    run()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(3):
        run()
    ctx.prof_enable(False)
    print('%s C=%d nmat=%d: %.2f ms' % (os.environ.get('TAG', ''), c, nmat, ctx.prof_read()['jacobi']['ms'] / 3), flush=True)
