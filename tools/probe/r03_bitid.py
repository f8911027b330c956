import numpy as np, sys
sys.path.insert(0, ".")
from wct_tf_amd.context import Context
ctx = Context(0)
rng = np.random.default_rng(5)
out = []
for c in (32, 64, 128, 256, 512):
    x = rng.standard_normal((2, 3 * c, c)).astype(np.float32)
    mats = np.einsum("bnc,bnd->bcd", x, x) / (3 * c)
    ev, vec, sw = ctx.eigh(mats.astype(np.float32), return_sweeps=True)
    out.append((float(np.abs(vec).sum()), float(ev.sum()), list(sw)))
print(out)
