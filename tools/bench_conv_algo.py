"""Per-layer A-B of the two 3x3 kernels through the pipeline's own entry (wct_conv3x3_f16: fp16 in, fp16 out): algo 1 = direct
implicit GEMM (csrc/conv.hip), algo 2 = Winograd F(2,3) along y (csrc/conv_wino.hip).  HIP-event class timing from the library.
usage: python tools/bench_conv_algo.py [batch=8] [min channels=0]      (WCT_WINO_CFG in a tuning build forces a tile shape)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context

# (cin, cout, H(out), upsample, pool, count per frame)  -- SURVEY.md 8a table
SHAPES = [(64, 64, 512, 0, 0, 5), (64, 64, 512, 1, 0, 5), (64, 128, 256, 0, 0, 5), (128, 128, 256, 0, 1, 4), (128, 128, 256, 1, 0, 3),
          (128, 64, 256, 0, 0, 4), (128, 256, 128, 0, 0, 4), (256, 256, 128, 0, 0, 6), (256, 256, 128, 0, 1, 3), (256, 256, 128, 1, 0, 6),
          (256, 128, 128, 0, 0, 3), (256, 512, 64, 0, 0, 3), (512, 512, 64, 0, 0, 4), (512, 512, 64, 0, 1, 2), (512, 512, 64, 1, 0, 3),
          (512, 256, 64, 0, 0, 2), (512, 512, 32, 0, 0, 3)]
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
minch = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = Context(0)
rng = np.random.default_rng(0)
tot = [0.0, 0.0]
for cin, cout, h, up, pool, cnt in SHAPES:
    if min(cin, cout) < minch:
        continue
    hin = h // 2 if up else h
    x = np.maximum(rng.standard_normal((batch, hin, hin, cin)), 0).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = np.zeros(cout, np.float32)
    ms = []
    for algo in (1, 2):
        ctx.conv3x3_f16(x, w, b, True, bool(up), bool(pool), algo)
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(5):
            ctx.conv3x3_f16(x, w, b, True, bool(up), bool(pool), algo)
        ctx.prof_enable(False)
        pr = ctx.prof_read()
        ms.append((pr['conv3x3']['ms'] + pr['conv_wino']['ms']) / 5)
    fl = 2.0 * h * batch * h * 9 * cin * cout
    tot[0] += ms[0] * cnt; tot[1] += ms[1] * cnt
    print('%3d->%3d @%3d up=%d pool=%d x%d: direct %.3f ms %5.0f TFLOP/s | winograd %.3f ms %5.0f TFLOP/s of the direct FLOPs (%4.0f executed) | x%.2f'
          % (cin, cout, h, up, pool, cnt, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, fl / 1.5 / ms[1] / 1e9, ms[0] / ms[1]), flush=True)
print('weighted per step (these layers, batch %d): direct %.2f ms, winograd %.2f ms' % (batch, tot[0], tot[1]))
