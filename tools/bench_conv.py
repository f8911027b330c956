"""Per-layer timing of the conv3x3 MFMA kernel (HIP-event class timing from the library).
Batch 8 is emulated by an 8x taller image (same tiles, same work)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context

# (cin, cout, H(out), upsample, count per frame)  -- SURVEY.md 8a table
SHAPES = [(64, 64, 512, 0, 5), (64, 64, 512, 1, 5), (64, 128, 256, 0, 5), (128, 128, 256, 0, 4), (128, 128, 256, 1, 3),
          (128, 64, 256, 0, 4), (128, 256, 128, 0, 4), (256, 256, 128, 0, 9), (256, 256, 128, 1, 6), (256, 128, 128, 0, 3),
          (256, 512, 64, 0, 3), (512, 512, 64, 0, 6), (512, 512, 64, 1, 3), (512, 256, 64, 0, 2), (512, 512, 32, 0, 3)]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:                      # only the layers whose narrower side has at most this many channels
    SHAPES = [s for s in SHAPES if min(s[0], s[1]) <= int(sys.argv[2])]
ctx = Context(0)
rng = np.random.default_rng(0)
tot_ms = tot_fl = 0
for cin, cout, h, up, cnt in SHAPES:
    hin = h // 2 if up else h
    x = rng.standard_normal((hin * batch, hin, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    b = np.zeros(cout, np.float32)
    ctx.conv3x3(x, w, b, True, bool(up))
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(3):
        ctx.conv3x3(x, w, b, True, bool(up))
    ctx.prof_enable(False)
    p = ctx.prof_read()['conv3x3']
    ms = p['ms'] / 3
    fl = 2.0 * h * batch * h * 9 * cin * cout
    tot_ms += ms * cnt; tot_fl += fl * cnt
    print('%3d->%3d @%3d up=%d x%d: %.3f ms  %6.0f TFLOP/s' % (cin, cout, h, up, cnt, ms, fl / ms / 1e9), flush=True)
print('weighted total per step: %.2f ms, %.0f TFLOP/s' % (tot_ms, tot_fl / tot_ms / 1e9))
