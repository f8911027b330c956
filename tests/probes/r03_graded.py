"""WCT path on hard spectra at C = 512: graded covariances (eigenvalues over 5-7 decades) at N = 4096 and N = 256 < C,
sweeps used per matrix, time, and error against the NumPy oracle (wct_np semantics)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from wct_tf_amd import _lib
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_features

def graded(rng, n, c, decades, mixing=0.3):
    mix = np.eye(c) + mixing * rng.standard_normal((c, c)) / np.sqrt(c)
    d = 10.0 ** (-np.arange(c) * decades / (c - 1) / 2)
    return np.float32(np.maximum(rng.standard_normal((n, c)) @ mix + 0.3, 0) * d * 3.0)

ctx = Context(0)
rng = np.random.default_rng(3)
c = int(os.environ.get('C', 512))
cases = []
for dec in (3.0, 5.0, 6.0, 7.0):
    cases.append(('graded %.0f dec N=4096' % dec, graded(rng, 4096, c, dec), graded(rng, 4096, c, dec)))
    cases.append(('graded %.0f dec N=256' % dec, graded(rng, 256, c, dec), graded(rng, 256, c, dec)))
cases.append(('graded 6 dec strong mixing N=4096', graded(rng, 4096, c, 6.0, 3.0), graded(rng, 4096, c, 6.0, 3.0)))
cases.append(('synthetic 3 dec N=1024', synthetic_features(1, c, 32, 32, 3.0).reshape(-1, c), synthetic_features(2, c, 32, 32, 3.0).reshape(-1, c)))
cases.append(('synthetic 4 dec N=256', synthetic_features(3, c, 16, 16, 4.0).reshape(-1, c), synthetic_features(4, c, 16, 16, 4.0).reshape(-1, c)))
for name, fc, fs in cases:
    ev = np.linalg.eigvalsh(np.cov(fc.astype(np.float64).T))
    try:
        got, sw = ctx.transform(fc, fs, 0.8, _lib.WCT_NP, return_sweeps=True)
        status = 'ok'
    except _lib.WCTNotConverged as e:
        got, sw, status = None, list(ctx.last_sweeps), 'NOCONV'
    t0 = time.perf_counter()
    for _ in range(3):
        try:
            ctx.transform(fc, fs, 0.8, _lib.WCT_NP)
        except _lib.WCTNotConverged:
            pass
    dt = (time.perf_counter() - t0) / 3
    want = oracle.wct_np(fc.reshape(1, -1, 2, c), fs.reshape(1, -1, 2, c), 0.8).reshape(fc.shape)
    err = np.linalg.norm(got - want) / np.linalg.norm(want) if got is not None else float('nan')
    print('%-36s eig range %.1e..%.1e kept %3d: sweeps %s %s  %.2f ms/call  rel err %.2e' % (
        name, ev.max(), max(ev.min(), 1e-30), (ev > 1e-5).sum(), list(sw), status, 1e3 * dt, err), flush=True)
