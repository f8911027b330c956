"""Round 4 probe: the real-image leg of bench.py with the class breakdown, profiling events on / off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, RELU_TARGETS
ctx = Context(0); ctx.set_weights(synthetic_weights(seed=42))
img = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'gilbert_512.npz'))['image']
c, s = np.ascontiguousarray(img), np.ascontiguousarray(img[:, ::-1])
for prof in (False, True, False, True):
    for _ in range(2):
        ctx.stylize(c, s, RELU_TARGETS, alpha=0.8)
    ctx.prof_reset(); ctx.prof_enable(prof)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.stylize(c, s, RELU_TARGETS, alpha=0.8)
    dt = (time.perf_counter() - t0) / 10
    ctx.prof_enable(False)
    p = ctx.prof_read()
    print('prof %d: %.2f ms per frame; classes %s' % (prof, 1e3 * dt, {k: round(v['ms'] / 10, 2) for k, v in p.items()}), flush=True)
