"""torch-CPU stand-in for the reference's CPU-TensorFlow conv stack (TEST / BASELINE INFRASTRUCTURE ONLY).

BASELINE.md section 3: TensorFlow and Keras cannot be installed here, so "the repo's own NumPy + CPU-TF path" is
timed as the reference's transform in NumPy (oracle.wct_tf / wct_np: restatements pinned bit-for-bit to the
reference's own wct_np at the metric's sizes, tests/test_oracle.py) around a torch-CPU restatement of the Keras
layers the reference builds: reflect pad + 3x3 VALID conv + ReLU (ops.py:12-19), MaxPooling2D(padding='same')
(vgg_normalised.py:42), UpSampling2D (model.py:293), with the same weights and frames.  It is a STAND-IN for
CPU-TF, labelled as such wherever it is reported; bench.py's cpu_baseline leg is its only user besides the tests.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import wct_oracle
from .net_oracle import ENCODER_LAYERS, decoder_layers, preprocess, postprocess


def _t(w_hwio):
    return torch.from_numpy(np.ascontiguousarray(np.transpose(w_hwio, (3, 2, 0, 1))))


class TorchPath(object):
    """Weights converted once (OIHW), then stylize() = one WCT.predict (wct.py:70-106) on the CPU."""

    def __init__(self, weights):
        enc = weights['encoder']
        self.pre = (_t(enc['preprocess'][0]), torch.from_numpy(np.asarray(enc['preprocess'][1], np.float32)))
        self.enc = {n: (_t(w), torch.from_numpy(np.asarray(b, np.float32))) for n, (w, b) in enc.items() if n != 'preprocess'}
        self.dec = {r: [(_t(w), torch.from_numpy(np.asarray(b, np.float32))) for w, b in layers]
                    for r, layers in weights['decoder'].items()}

    @staticmethod
    def _conv(x, w, b, relu):
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, b)
        return F.relu(y) if relu else y

    def encode(self, x, targets):
        want = set(targets)
        deepest = sorted(want)[-1]                          # model.py:60
        x = F.conv2d(x, *self.pre)
        feats = {}
        for layer in ENCODER_LAYERS:
            if layer[0] == 'C':
                x = self._conv(x, *self.enc[layer[1]], True)
                relu = 'relu' + layer[1][4:]
                if relu in want:
                    feats[relu] = x
                if relu == deepest:
                    break
            else:
                x = F.max_pool2d(x, 2, 2, ceil_mode=True)
        return feats

    def decode(self, x, relu_target):
        params = iter(self.dec[relu_target])
        for kind, cin, cout, relu in decoder_layers(relu_target):
            if kind == 'C':
                w, b = next(params)
                x = self._conv(x, w, b, relu)
            else:
                x = F.interpolate(x, scale_factor=2, mode='nearest')
        return x

    def stylize(self, content, style, relu_targets, alpha=1.0, wct_mode='tf', timers=None, transform=None):
        """timers (optional dict): receives 'transform_s', the seconds spent in the NumPy transforms of this call.
        transform (optional callable (fc, fs, alpha) -> [1][H][W][C]): used instead of the restatement -- bench.py passes the
        reference's own lifted function where the reference tree is present"""
        import time
        t_transform = 0.0
        to_t = lambda img: torch.from_numpy(np.float32(preprocess(img)).transpose(2, 0, 1)[None])     # noqa: E731
        to_np = lambda t: t[0].permute(1, 2, 0).numpy()                                                # noqa: E731
        with torch.no_grad():
            sfeat = self.encode(to_t(style), relu_targets)
            x = to_t(content)
            for i, relu in enumerate(relu_targets):
                if i > 0:
                    x = torch.clamp(x, 0, 1)                # model.py:86
                fc = to_np(self.encode(x, [relu])[relu])
                fs = to_np(sfeat[relu])
                t0 = time.time()
                t = (transform or (wct_oracle.wct_tf if wct_mode == 'tf' else wct_oracle.wct_np))(fc, fs, alpha)
                t_transform += time.time() - t0
                x = self.decode(torch.from_numpy(np.ascontiguousarray(t[0].transpose(2, 0, 1)))[None], relu)
        if timers is not None:
            timers['transform_s'] = t_transform
        return postprocess(to_np(x))
