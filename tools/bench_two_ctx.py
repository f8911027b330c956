"""Feasibility probe: two contexts (two HIP streams) driven by two host threads on ONE GPU, each stylizing its
own resident batch -- do the conv kernels of one overlap the eigensolver of the other?
usage: bench_two_ctx.py [batch_per_ctx] [steps] [n_ctx]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, synthetic_image

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
LEVELS = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
S = 512
weights = synthetic_weights(seed=42)
lanes = []
for k in range(nctx):
    ctx = Context(0)
    ctx.set_weights(weights)
    c = np.stack([synthetic_image(1000 + k * B + i, S, S) for i in range(B)])
    s = np.stack([synthetic_image(2000 + k * B + i, S, S) for i in range(B)])
    dc, ds, do = ctx.dev_alloc(c.nbytes), ctx.dev_alloc(s.nbytes), ctx.dev_alloc(c.nbytes)
    ctx.h2d(dc, c); ctx.h2d(ds, s)
    lanes.append((ctx, dc, ds, do))

def run(lane, n):
    ctx, dc, ds, do = lane
    for _ in range(n):
        ctx.stylize_batch_dev(dc, S, S, ds, S, S, B, LEVELS, 0.8, do)
    ctx.sync()

for lane in lanes:
    run(lane, 2)
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(lane, steps)) for lane in lanes]
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
print('ctx=%d batch/ctx=%d: %.1f frames/s (%.2f ms per %d frames)' % (nctx, B, nctx * B * steps / dt, 1e3 * dt / steps, nctx * B))
