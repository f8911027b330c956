"""Probe: does a second context on the SAME GPU hide the eigensolver's latency chains under the other context's convolutions?
One context, 32 pairs per step  vs  two contexts driven by two host threads, 16 pairs each per step (the same 32 pairs of work).
usage (GPU box): python tools/probe/two_ctx_overlap.py [steps]"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, synthetic_image

LEVELS = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
S, ALPHA, STEPS = 512, 0.8, int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda', 0)
weights = synthetic_weights(seed=42)
content = torch.from_numpy(np.stack([synthetic_image(1000 + i, S, S) for i in range(32)])).to(dev)
style = torch.from_numpy(np.stack([synthetic_image(2000 + i, S, S) for i in range(32)])).to(dev)
fb = S * S * 3

def make():
    c = Context(0); c.set_weights(weights); return c

def run(ctx, lo, hi, out, steps):
    for _ in range(steps):
        ctx.stylize_batch_dev(C.c_void_p(content.data_ptr() + lo * fb), S, S, C.c_void_p(style.data_ptr() + lo * fb), S, S, hi - lo, LEVELS, ALPHA,
                              C.c_void_p(out.data_ptr() + lo * fb))
    ctx.sync()

def timed(ctxs, splits, out):
    for c, (lo, hi) in zip(ctxs, splits): run(c, lo, hi, out, 1)          # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c, lo, hi, out, STEPS)) for c, (lo, hi) in zip(ctxs, splits)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3

a, b, c4 = make(), make(), [make(), make()]
o1, o2, o4 = torch.empty_like(content), torch.empty_like(content), torch.empty_like(content)
t1 = timed([a], [(0, 32)], o1)
t2 = timed([a, b], [(0, 16), (16, 32)], o2)
t4 = timed([a, b] + c4, [(0, 8), (8, 16), (16, 24), (24, 32)], o4)
t1b = timed([a], [(0, 32)], o1)
print('one context x 32 pairs: %.2f ms per step (again: %.2f) = %.1f frames/s' % (t1, t1b, 32e3 / min(t1, t1b)))
print('two contexts x 16 pairs, two host threads: %.2f ms per 32 pairs = %.1f frames/s; frames identical: %s' % (t2, 32e3 / t2, bool(torch.equal(o1, o2))))
print('four contexts x 8 pairs: %.2f ms per 32 pairs = %.1f frames/s; frames identical: %s' % (t4, 32e3 / t4, bool(torch.equal(o1, o4))))
