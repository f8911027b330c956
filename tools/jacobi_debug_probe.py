"""Why does a solve run out of sweeps?  Prints the solver's final state words (WCT_JACOBI_DEBUG=1) for a few inputs."""
import os, sys
import numpy as np
os.environ['WCT_JACOBI_DEBUG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wct_tf_amd import _lib
from wct_tf_amd.context import Context
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_gpu_ops import _graded_spd
from test_gpu_fuzz import features
ctx = Context(0)
def run(tag, fn):
    print('==', tag, flush=True)
    try:
        fn()
    except Exception as e:
        print('  ->', type(e).__name__, str(e)[:80])
    ctx.sync() if False else None
rng = np.random.default_rng(256)
mats = np.stack([_graded_spd(rng, 256, 3.0), _graded_spd(rng, 256, 1.0, rank=256 // 3)])
for i, a in enumerate(mats):
    ev = np.linalg.eigvalsh(a.astype(np.float64))
    print('mat', i, 'eig max %.3e, smallest 5 %s, #|ev|<1e-5*max: %d' % (ev.max(), np.array2string(ev[:5], precision=2), (np.abs(ev) < 1e-5 * ev.max()).sum()))
run('eigh 256', lambda: ctx.eigh(mats))
rng = np.random.default_rng(0)
scale = 100.0
fc, fs = features(rng, 14, 96, scale), features(rng, 4, 96, scale * 10.0 ** rng.uniform(-1, 1))
run('transform C=96 N=14 scale 1e2', lambda: ctx.transform(fc, fs, 0.5, _lib.WCT_NP))
