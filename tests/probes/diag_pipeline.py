"""Free-running, level-by-level comparison of the GPU ops with the oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from wct_tf_amd import _lib
from wct_tf_amd.context import Context
from wct_tf_amd.weights import synthetic_weights, synthetic_image, RELU_TARGETS

def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
mode = sys.argv[2] if len(sys.argv) > 2 else 'tf'
w = synthetic_weights(42)
ctx = Context(0); ctx.set_weights(w)
c = synthetic_image(1000, size, size); s = synthetic_image(2000, size, size)
want, levels = oracle.stylize(c, s, w, RELU_TARGETS, alpha=0.8, wct_mode=mode, return_levels=True)
s01 = np.float32(s / 255.)
x = np.float32(c / 255.)
xo = x.copy()
for i, (relu, (fc_o, fs_o, t_o, x_o)) in enumerate(zip(RELU_TARGETS, levels)):
    if i > 0:
        x = np.clip(x, 0, 1)
    fc = ctx.encode(x, relu); fs = ctx.encode(s01, relu)
    C = fc.shape[-1]
    t, sw = ctx.transform(fc.reshape(-1, C), fs.reshape(-1, C), 0.8, _lib.WCT_TF if mode == 'tf' else _lib.WCT_NP, return_sweeps=True)
    t = t.reshape(fc.shape)
    # oracle transform on the GPU's features: isolates transform error from input error
    t_same = (oracle.wct_tf if mode == 'tf' else oracle.wct_np)(fc[None], fs[None], 0.8)[0]
    ev = np.linalg.eigvalsh(np.cov(fc.reshape(-1, C).astype(np.float64).T))[::-1]
    k = int((ev > 1e-5).sum())
    x = ctx.decode(t, relu)
    print('%s: fc vs oracle %.2e  fs %.2e | transform vs oracle-on-same-input %.2e  vs oracle chain %.2e | decoded vs chain %.2e | sweeps %s | k=%d ev around thr %s'
          % (relu, rel(fc, fc_o), rel(fs, fs_o), rel(t, t_same), rel(t, t_o), rel(x, x_o), sw, k, ev[max(0, k - 2):k + 2]), flush=True)
got = ctx.stylize(c, s, RELU_TARGETS, alpha=0.8, wct_mode=mode)
step = np.uint8(np.clip(x, 0, 1) * 255)
d = np.abs(got.astype(int) - want.astype(int)); d2 = np.abs(got.astype(int) - step.astype(int))
print('pipeline vs oracle: mean LSB %.3f max %d ; pipeline vs stepwise GPU: mean %.3f max %d' % (d.mean(), d.max(), d2.mean(), d2.max()))
