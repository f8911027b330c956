"""One layer through the reduced-FLOP kernel (algo 2) or the direct one (algo 1), HIP-event timed: for ablation runs of a tuning
build (WCT_WINO_CFG / WCT_WINO_DBG).  args: cin cout H up pool batch algo"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context
cin, cout, h, up, pool, batch, algo = [int(a) for a in sys.argv[1:8]]
ctx = Context(0)
rng = np.random.default_rng(0)
hin = h // 2 if up else h
x = np.maximum(rng.standard_normal((batch, hin, hin, cin)), 0).astype(np.float32)
w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
b = np.zeros(cout, np.float32)
ctx.conv3x3_f16(x, w, b, True, bool(up), bool(pool), algo)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    ctx.conv3x3_f16(x, w, b, True, bool(up), bool(pool), algo)
ctx.prof_enable(False)
pr = ctx.prof_read()
ms = (pr['conv3x3']['ms'] + pr['conv_wino']['ms']) / 5
fl = 2.0 * h * batch * h * 9 * cin * cout
print('%3d->%3d @%3d up=%d pool=%d batch %d algo %d CFG=%s DBG=%s: %.3f ms  %5.0f TFLOP/s (direct FLOPs)' % (
    cin, cout, h, up, pool, batch, algo, os.environ.get('WCT_WINO_CFG', '-'), os.environ.get('WCT_WINO_DBG', '-'), ms, fl / ms / 1e9))
