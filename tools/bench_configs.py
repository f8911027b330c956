"""Latency of the BASELINE.json configurations that are not the headline one, ONE predict() each (wct_stylize: host uint8 in ->
host uint8 out, batch 1), mean of n calls after 3 warm-ups:
  configs[1]  single level relu3_1, 512x512, alpha 0.8
  configs[2]  full 5-level, 512x512, alpha 0.8                     (what bench.py's latency_fps reports)
  configs[4]  full 5-level, 1024x1024 content / 512x512 style, --keep-colors CORAL first, WCT branch and --adain branch
usage: python tools/bench_configs.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, synthetic_image
from wct_tf_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
L5 = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
ctx = Context(0)
ctx.set_weights(synthetic_weights(seed=42))


def timed(label, fn):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    print('%-78s %8.2f ms  %7.1f frames/s' % (label, 1e3 * dt, 1.0 / dt), flush=True)


c512, s512 = synthetic_image(1000, 512, 512), synthetic_image(2000, 512, 512)
c1024 = synthetic_image(1005, 1024, 1024)
timed('configs[1]: relu3_1 only, 512x512, alpha 0.8', lambda: ctx.stylize(c512, s512, ['relu3_1'], alpha=0.8))
timed('configs[2]: 5 levels, 512x512, alpha 0.8', lambda: ctx.stylize(c512, s512, L5, alpha=0.8))
timed('configs[4]: CORAL keep-colors (512 style <- 1024 content colours)', lambda: ops.preserve_colors_np(s512, c1024, ctx=ctx))
scc = ops.preserve_colors_np(s512, c1024, ctx=ctx)
timed('configs[4]: 5 levels, 1024x1024 content / 512x512 style, WCT, alpha 0.8', lambda: ctx.stylize(c1024, scc, L5, alpha=0.8))
timed('configs[4]: the same with --adain', lambda: ctx.stylize(c1024, scc, L5, alpha=0.8, adain=True))
ctx.close()
