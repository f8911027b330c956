"""A well-conditioned synthetic weight set for END-TO-END parity tests (TEST INFRASTRUCTURE ONLY).

Why: the He-normal stand-in weights (wct_tf_amd/weights.py) form a chaotic map -- every zero-bias ReLU layer passes half
of a perturbation's power but only 0.34 of the signal's variance, a relative growth of ~1.2x per layer, ~9x per encoder
pass and ~4x per decoder pass (measured: tests/test_oracle.py) -- so no two implementations, fp64 NumPy included, agree
on the five-level frame.  The reference's pre-trained networks are not chaotic (they were trained to invert each
other, model.py:181-194), but they cannot be downloaded here.  This generator builds the same architecture
(vgg_normalised.py:22-50, model.py:255-298) with layers that are close to isometries on the fluctuating part of their
input, calibrated layer by layer on a seeded image (LSUV-style):
  * 3x3 filters: random, zero-sum (the mean of the input does not propagate), scaled so that the pre-activation
    fluctuation has unit standard deviation on the calibration image;
  * bias = `margin` standard deviations, so the ReLU clips only the far tail and the layer is almost linear;
  * conv1_1 keeps a small bias (its ReLU is what gives relu1_1 more than the 27 directions a linear map of 3x3x3
    patches has);
  * the 3-channel output convs map to 0.5 +- 0.15.
"""
import numpy as np

from . import net_oracle
from wct_tf_amd.weights import ENCODER_CONVS, RELU_TARGETS, decoder_plan, synthetic_image


def _filters(rng, cin, cout):
    w = rng.standard_normal((3, 3, cin, cout)).astype(np.float32)
    if cin > 3:
        w -= w.mean(axis=(0, 1, 2), keepdims=True)
    return w


def contractive_weights(seed=7, margin=2.5, calib_size=96):
    rng = np.random.default_rng(seed)
    pre_w = np.zeros((1, 1, 3, 3), np.float32)
    for c in range(3):
        pre_w[0, 0, 2 - c, c] = 255.0
    pre_b = -np.array([103.939, 116.779, 123.68], np.float32)
    enc = {'preprocess': (pre_w, pre_b)}
    x = net_oracle.conv1x1(np.float32(synthetic_image(seed + 1, calib_size, calib_size) / 255.), pre_w, pre_b)
    feats = {}
    pool_after = {'conv1_2', 'conv2_2', 'conv3_4', 'conv4_4'}
    for name, cin, cout in ENCODER_CONVS:
        w = _filters(rng, cin, cout)
        z = net_oracle.conv3x3_reflect(x, w, np.zeros(cout, np.float32), relu=False)
        w = (w / z.std()).astype(np.float32)
        b = np.full(cout, 0.3 if name == 'conv1_1' else margin, np.float32)
        enc[name] = (w, b)
        x = net_oracle.conv3x3_reflect(x, w, b, relu=True)
        if name.endswith('_1'):
            feats['relu' + name[4:]] = x
        if name in pool_after:
            x = net_oracle.maxpool2x2_same(x)
    dec = {}
    for relu in RELU_TARGETS:
        layers = []
        x = feats[relu]
        plan = decoder_plan(relu)
        nconv = sum(1 for p in plan if p[0] == 'C')
        i = 0
        for kind, cin, cout, act in plan:
            if kind == 'U':
                x = net_oracle.upsample2x_nearest(x)
                continue
            w = _filters(rng, cin, cout)
            z = net_oracle.conv3x3_reflect(x, w, np.zeros(cout, np.float32), relu=False)
            last = i == nconv - 1
            w = (w * ((0.15 if last else 1.0) / z.std())).astype(np.float32)
            b = np.full(cout, 0.5 if last else margin, np.float32)
            layers.append((w, b))
            x = net_oracle.conv3x3_reflect(x, w, b, relu=act)
            i += 1
        dec[relu] = layers
    return {'encoder': enc, 'decoder': dec}
