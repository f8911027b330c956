"""What would fp16 feature TAPS cost the transform?  (measured on the CPU with the oracle before building anything.)
For each level of the five-level chain at 512 x 512 (the oracle's own teacher-forced features on the He-normal stand-in weights and
on the contractive net): relative error of the reference transform when its two inputs are rounded to fp16 first,
    || wct(fp16(fc), fp16(fs)) - wct(fc, fs) || / || wct(fc, fs) ||     (wct_tf semantics, alpha 0.8)
-- the error the pipeline would add to every transform if the taps were stored in fp16 (the op-level wct_transform keeps fp32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from oracle.contractive import contractive_weights
from wct_tf_amd.weights import synthetic_weights, synthetic_image, RELU_TARGETS

def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
c, s = synthetic_image(1000, size, size), synthetic_image(2000, size, size)
for name, w in (('He-normal stand-in (seed 42)', synthetic_weights(42)), ('contractive net (seed 7)', contractive_weights(7))):
    _, levels = oracle.stylize(c, s, w, RELU_TARGETS, alpha=0.8, wct_mode='tf', return_levels=True)
    print(name)
    for relu, (fc, fs, t, x) in zip(RELU_TARGETS, levels):
        fc16, fs16 = np.float32(np.float16(fc)), np.float32(np.float16(fs))
        t16 = oracle.wct_tf(fc16, fs16, 0.8)
        ev = np.linalg.eigvalsh(np.cov(np.float64(fc).reshape(-1, fc.shape[-1]).T))
        print('  %s C=%3d N=%6d: input rounding %.2e / %.2e  ->  transform error %.2e   (content eigenvalues %.1e .. %.1e, kept %d of %d)'
              % (relu, fc.shape[-1], fc.shape[-3] * fc.shape[-2], rel(fc16, fc), rel(fs16, fs), rel(t16, t), ev.max(), max(ev.min(), 0), int((ev > 1e-5).sum()), len(ev)), flush=True)
