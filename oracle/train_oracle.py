"""TEST INFRASTRUCTURE ONLY.  CPU reference of one decoder training step (model.py:123-223): losses and the
gradients of the decoder's kernels and biases by torch autograd.

Semantics restated from the reference:
  * encoder layers / 'same' max-pool / reflect-pad 3x3 convs / x2 nearest upsample as in oracle/net_oracle.py
    (vgg_normalised.py:22-50, ops.py:12-19, model.py:255-298);
  * content_encoded = enc(x); decoded = dec(content_encoded); decoded_encoded = enc(decoded)     (model.py:135-176)
  * feature_loss = w_f * mse(decoded_encoded, content_encoded); pixel_loss = w_p * mse(decoded, x);
    tv_loss = w_tv * reduce_mean(tf.image.total_variation(decoded))                            (model.py:181-194)
  * only the decoder's variables are trained (model.py:202).

`emulate_fp16=True` rounds the conv operands the way the GPU forward does (fp16 weights and activations, fp32
accumulation, straight-through rounding for the gradient), so that the comparison isolates the backward pass.
"""
import numpy as np
import torch
import torch.nn.functional as F

from wct_tf_amd.weights import ENCODER_CONVS, decoder_plan, RELU_LEVEL


def _ste16(t):
    return t + (t.half().float() - t).detach()


def _conv(x, w_hwio, b, relu, q, store16=False):
    """q: fp16 operands (fp32 accumulate); store16: the output is kept as fp16 on the GPU (every activation but
    the fp32 feature taps and the decoded image), which also decides the max-pool's argmax among near-ties."""
    w = torch.as_tensor(np.ascontiguousarray(np.transpose(w_hwio, (3, 2, 0, 1)))) if not torch.is_tensor(w_hwio) else w_hwio
    if q:
        x, w = _ste16(x), _ste16(w)
    y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, b)
    y = F.relu(y) if relu else y
    return _ste16(y) if store16 else y


def encode(x_nchw, enc, relu_target, q):
    """enc: weights['encoder'] of wct_tf_amd.weights; returns the relu_target feature map (NCHW)."""
    pw, pb = enc['preprocess']
    # 'preprocess': 1x1 conv, HWIO [1][1][3][3] (vgg_normalised.py:25-26)
    x = F.conv2d(x_nchw, torch.as_tensor(np.ascontiguousarray(np.transpose(np.float32(pw), (3, 2, 0, 1)))), torch.as_tensor(np.float32(pb)))
    level = RELU_LEVEL[relu_target]
    prev_block = 1
    for name, cin, cout in ENCODER_CONVS:
        block = int(name[4])
        if block != prev_block:
            x = F.max_pool2d(x, 2, 2, ceil_mode=True)
            prev_block = block
        w, b = enc[name]
        first = name == 'conv1_1'
        # conv1_1 runs at fp32-product accuracy on the GPU (split operands); every later conv on fp16 operands
        tap = name == 'conv%d_1' % level
        x = _conv(x, w, torch.as_tensor(b), True, q and not first, store16=q and not tap)
        if tap:
            return x
    raise ValueError(relu_target)


def decode(feat, dec_params, relu_target, q):
    x = feat
    it = iter(dec_params)
    for kind, cin, cout, relu in decoder_plan(relu_target):
        if kind == 'U':
            x = F.interpolate(x, scale_factor=2, mode='nearest')
        else:
            w, b = next(it)
            x = _conv(x, w, b, bool(relu), q, store16=q and cout != 3)
    return x


def train_losses_and_grads(images_nhwc, weights, relu_target, feature_weight=1.0, pixel_weight=1.0, tv_weight=0.0,
                           emulate_fp16=True):
    """images [B][H][W][3] fp32 in [0,1].  Returns (losses dict, [(dW HWIO, db)] per decoder conv)."""
    q = emulate_fp16
    x = torch.as_tensor(np.ascontiguousarray(np.transpose(np.float32(images_nhwc), (0, 3, 1, 2))))
    params = []
    for w, b in weights['decoder'][relu_target]:
        wt = torch.tensor(np.ascontiguousarray(np.transpose(np.float32(w), (3, 2, 0, 1))), requires_grad=True)
        bt = torch.tensor(np.float32(b), requires_grad=True)
        params.append((wt, bt))
    with torch.no_grad():
        content = encode(x, weights['encoder'], relu_target, q)
    feat_in = content.half().float() if q else content            # the decoder input is stored as fp16 on the GPU
    decoded = decode(feat_in, params, relu_target, q)
    decoded_encoded = encode(decoded, weights['encoder'], relu_target, q)
    feature_loss = feature_weight * torch.mean((decoded_encoded - content) ** 2)
    pixel_loss = pixel_weight * torch.mean((decoded - x) ** 2)
    tv = (decoded[:, :, 1:, :] - decoded[:, :, :-1, :]).abs().sum(dim=(1, 2, 3)) + \
         (decoded[:, :, :, 1:] - decoded[:, :, :, :-1]).abs().sum(dim=(1, 2, 3))
    tv_loss = tv_weight * tv.mean()
    total = feature_loss + pixel_loss + tv_loss
    total.backward()
    grads = [(np.ascontiguousarray(np.transpose(w.grad.numpy(), (2, 3, 1, 0))), b.grad.numpy().copy()) for w, b in params]
    losses = {'feature_loss': float(feature_loss.detach()), 'pixel_loss': float(pixel_loss.detach()),
              'tv_loss': float(torch.as_tensor(tv_loss).detach()), 'total_loss': float(total.detach())}
    return losses, grads
