"""Digest of conv3x3 outputs over a set of shapes (incl. ragged / upsampled / tiny maps) for the tile configuration selected by
WCT_CONV_CFG / WCT_CONV_XCD (tuning build): run once per configuration and compare the digests -- every configuration sums a
pixel's taps and channels in the same order, so they must agree bit for bit.  usage: conv_cfg_check.py [min_cout]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wct_tf_amd.context import Context
min_cout = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(256, 256, 128, 128, 0), (256, 256, 64, 64, 1), (512, 512, 64, 64, 0), (512, 512, 32, 32, 1), (512, 256, 64, 64, 0), (128, 256, 128, 128, 0),
          (256, 512, 37, 53, 0), (512, 512, 2, 2, 0), (512, 512, 16, 48, 0), (64, 256, 50, 18, 1), (512, 512, 31, 17, 1)]
if os.environ.get('CONV_CHECK_TALL'):      # tall images (a batch emulated by height): enough tiles for the strip variants of the narrow layers
    SHAPES = [(64, 64, 4096, 512, 0), (64, 64, 2048, 256, 1), (64, 128, 4096, 256, 0), (128, 128, 4096, 256, 0), (128, 128, 2048, 128, 1),
              (128, 64, 4096, 256, 0), (64, 64, 4099, 515, 0), (64, 64, 2051, 253, 1)]
ctx = Context(0)
rng = np.random.default_rng(5)
h = hashlib.sha256()
for cin, cout, hh, ww, up in SHAPES:
    if cout < min_cout:
        continue
    x = np.maximum(rng.standard_normal((hh, ww, cin)), 0).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    y = ctx.conv3x3(x, w, b, relu=True, upsample=bool(up))
    d = hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()
    h.update(d.encode())
    print('%3d->%3d %3dx%3d up=%d: %s  (mean %.5f)' % (cin, cout, hh, ww, up, d[:12], float(y.mean())))
print('CFG=%s XCD=%s STRIPS=%s digest %s' % (os.environ.get('WCT_CONV_CFG', '-'), os.environ.get('WCT_CONV_XCD', '-'), os.environ.get('WCT_CONV_STRIPS', '-'), h.hexdigest()[:16]))
