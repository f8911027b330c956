"""One conv layer, a few launches (for PMC passes).  args: cin cout H up batch"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context
cin, cout, h, up, batch = [int(a) for a in sys.argv[1:6]]
ctx = Context(0)
rng = np.random.default_rng(0)
hin = h // 2 if up else h
x = rng.standard_normal((hin * batch, hin, cin)).astype(np.float32)
w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
b = np.zeros(cout, np.float32)
for _ in range(3):
    ctx.conv3x3(x, w, b, True, bool(up))
