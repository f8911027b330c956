"""Training-step timing at the reference's configuration (batch 8, 256x256 crops, train.py:52,91)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, synthetic_image

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = np.stack([synthetic_image(100 + i, 256, 256) / 255. for i in range(b)]).astype(np.float32)
for relu in ['relu1_1', 'relu2_1', 'relu3_1', 'relu4_1', 'relu5_1']:
    w = synthetic_weights(42, relu_targets=[relu])
    ctx = Context(0)
    ctx.set_weights(w)
    for t in range(1, 3):
        ctx.train_step(relu, x, step=t, learning_rate=1e-4)
    t0 = time.perf_counter()
    n = 5
    for t in range(3, 3 + n):
        r = ctx.train_step(relu, x, step=t, learning_rate=1e-4)
    dt = (time.perf_counter() - t0) / n
    print('%s batch %d: %.1f ms/step  %.1f steps/s  %.0f images/s  (total loss %.4f)' % (relu, b, dt * 1e3, 1 / dt, b / dt, r['total_loss']), flush=True)
    ctx.close()
