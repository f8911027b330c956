"""NumPy restatement of the reference encoder / decoder graph (TEST ORACLE ONLY).

The reference delegates these ops to Keras/TensorFlow, which cannot be
installed here, so this file restates their documented semantics; it is
cross-checked against torch-CPU in tests/test_oracle.py ("parity unpinned" by
the reference itself -- it ships no tests).  float32 throughout.

Weights container (shared with the product, see wct_tf_amd/weights.py):
  weights['encoder'][name] = (w_hwio float32 [kh,kw,cin,cout], b float32 [cout])
     names: 'preprocess' (1x1), 'conv1_1', 'conv1_2', 'conv2_1', ... 'conv5_1'
  weights['decoder'][relu] = [(w_hwio, b), ...] in execution order
"""
import numpy as np

from . import wct_oracle

# VGG19-normalised module order as walked by vgg_normalised.py:22-50
# ('C' = reflect-pad + 3x3 valid conv + ReLU named relu<name>, 'P' = 2x2 max-pool
# with padding='same', vgg_normalised.py:42).
ENCODER_LAYERS = [
    ('C', 'conv1_1', 3, 64), ('C', 'conv1_2', 64, 64), ('P', 'pool1'),
    ('C', 'conv2_1', 64, 128), ('C', 'conv2_2', 128, 128), ('P', 'pool2'),
    ('C', 'conv3_1', 128, 256), ('C', 'conv3_2', 256, 256),
    ('C', 'conv3_3', 256, 256), ('C', 'conv3_4', 256, 256), ('P', 'pool3'),
    ('C', 'conv4_1', 256, 512), ('C', 'conv4_2', 512, 512),
    ('C', 'conv4_3', 512, 512), ('C', 'conv4_4', 512, 512), ('P', 'pool4'),
    ('C', 'conv5_1', 512, 512),
]

# model.py:255-277, written as (kind, out_channels)
DECODER_ARCHS = {
    5: [('C', 512), ('U',), ('C', 512), ('C', 512), ('C', 512)],
    4: [('C', 256), ('U',), ('C', 256), ('C', 256), ('C', 256)],
    3: [('C', 128), ('U',), ('C', 128)],
    2: [('C', 64), ('U',)],
    1: [('C', 64)],
}
_RELU_NUM = {'relu1_1': 1, 'relu2_1': 2, 'relu3_1': 3, 'relu4_1': 4, 'relu5_1': 5}
_RELU_CH = {'relu1_1': 64, 'relu2_1': 128, 'relu3_1': 256, 'relu4_1': 512, 'relu5_1': 512}


def decoder_layers(relu_target):
    """Layer list [(kind, cin, cout, relu)] of the decoder for `relu_target`
    (model.py:283-298: walk decoder_num..1, then a final 3-filter conv with no
    activation)."""
    cin = _RELU_CH[relu_target]
    out = []
    for d in range(_RELU_NUM[relu_target], 0, -1):
        for tup in DECODER_ARCHS[d]:
            if tup[0] == 'C':
                out.append(('C', cin, tup[1], True))
                cin = tup[1]
            else:
                out.append(('U', cin, cin, False))
    out.append(('C', cin, 3, False))
    return out


def _pad_reflect(x):
    """1-px REFLECT pad on H and W (edge not repeated), ops.py:12-15."""
    return np.pad(x, ((1, 1), (1, 1), (0, 0)), mode='reflect')


def conv3x3_reflect(x, w_hwio, b, relu=True):
    """Reflect-pad(1) + 3x3 VALID conv + bias (+ReLU); ops.py:17-19,
    vgg_normalised.py:28-40.  x: HxWxCin float32; w: 3x3xCinxCout."""
    x = np.asarray(x, np.float32)
    h, w, cin = x.shape
    cout = w_hwio.shape[3]
    xp = _pad_reflect(x)
    wmat = np.ascontiguousarray(w_hwio, np.float32).reshape(9 * cin, cout)
    out = np.empty((h, w, cout), np.float32)
    rows = max(1, (1 << 22) // max(1, w * 9 * cin))  # bound the im2col strip
    for r0 in range(0, h, rows):
        r1 = min(h, r0 + rows)
        cols = np.empty((r1 - r0, w, 9, cin), np.float32)
        for ky in range(3):
            for kx in range(3):
                cols[:, :, ky * 3 + kx, :] = xp[r0 + ky:r1 + ky, kx:kx + w, :]
        o = cols.reshape(-1, 9 * cin) @ wmat
        out[r0:r1] = o.reshape(r1 - r0, w, cout)
    out += np.asarray(b, np.float32)
    if relu:
        np.maximum(out, 0, out=out)
    return out


_WINO_G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
_WINO_BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)


def wino_layer(cin, cout):
    """The layers the MI355X path runs on its reduced-FLOP kernel (csrc/api.hip::wino_layer; tap layers excepted there)."""
    return cin >= 256 and cout >= 256


def conv3x3_reflect_wino_f16(x, w_hwio, b, relu=True, acc=np.float64):
    """The SAME layer as conv3x3_reflect with the roundings of the MI355X path's reduced-FLOP kernel (csrc/conv_wino.hip:
    Winograd F(2,3) along y, direct along x) -- not a reference mode, a restatement that shows accumulation order only:
    activations rounded to fp16; filters U_f[kx] = sum_ky G[f][ky] g[ky][kx] rounded to fp16; transformed rows T = B^T d
    computed from the fp16 activations and rounded to fp16 (one add per value); products exact, sums in float64;
    y(2r) = M0 + M1 + M2, y(2r+1) = M1 - M2 - M3; bias, ReLU; the result rounded to fp16 (the kernel stores fp16)."""
    x = np.asarray(x, np.float16).astype(acc)
    h, w, cin = x.shape
    cout = w_hwio.shape[3]
    he = h + (h & 1)                                     # an odd last row: its partner row is computed and dropped
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)), mode='reflect')
    if he != h:
        xp = np.concatenate([xp, xp[-1:]], 0)            # (the kernel clamps the row index there; the row only feeds the dropped output)
    U = np.einsum('fk,kxio->fxio', _WINO_G, np.asarray(w_hwio, np.float64)).astype(np.float32).astype(np.float16).astype(acc)
    rows = [xp[i:i + he:2] for i in range(4)]            # padded row 2r + i of every pair-row r
    y = np.empty((he, w, cout), acc)
    M = []
    for f in range(4):
        T = sum(acc(_WINO_BT[f, i]) * rows[i] for i in range(4) if _WINO_BT[f, i] != 0).astype(np.float16).astype(acc)    # [he/2][w+2][cin]
        m = 0
        for kx in range(3):
            m = m + T[:, kx:kx + w, :].reshape(-1, cin) @ U[f, kx]
        M.append(m.reshape(he // 2, w, cout))
    y[0::2] = (M[0] + M[1]) + M[2]
    y[1::2] = (M[1] - M[2]) - M[3]
    y = y[:h] + np.asarray(b, acc)
    if relu:
        y = np.maximum(y, 0)
    return y.astype(np.float16).astype(np.float32)


def conv1x1(x, w_hwio, b):
    """The VGG 'preprocess' 1x1 conv (vgg_normalised.py:25-26,28-38)."""
    cin = w_hwio.shape[2]
    return (np.asarray(x, np.float32).reshape(-1, cin) @ w_hwio.reshape(cin, -1)
            ).reshape(x.shape[0], x.shape[1], -1) + np.asarray(b, np.float32)


def maxpool2x2_same(x):
    """MaxPooling2D(pool 2, stride 2, padding='same') == ceil-mode pooling:
    the odd last row/col pools over the cells that exist (vgg_normalised.py:42)."""
    h, w, c = x.shape
    ho, wo = (h + 1) // 2, (w + 1) // 2
    xp = np.full((ho * 2, wo * 2, c), -np.inf, np.float32)
    xp[:h, :w] = x
    return xp.reshape(ho, 2, wo, 2, c).max(axis=(1, 3))


def upsample2x_nearest(x):
    """UpSampling2D() default: x2 nearest (model.py:293)."""
    return np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)


def _h16(a):
    return np.asarray(a, np.float16).astype(np.float32)


def encode(img01, weights, targets, fp16_storage=False, wino=True):
    """Run the shared VGG on an HxWx3 image in [0,1]; return {relu: features}
    for every relu name in `targets` (model.py:60-75,135-139).

    fp16_storage=True restates the SAME graph with the storage precision of the MI355X path: 3x3
    filters (conv1_1 excepted: it runs on split fp16 pairs, 22 bits) and the activations handed
    from layer to layer are rounded to fp16, sums and the tapped feature maps stay fp32.  It is not
    a reference mode; it lets the stack tests use a tolerance that shows accumulation order only.  wino (with fp16_storage): the
    layers that path runs on its reduced-FLOP kernel (>= 256 channels in and out, not a tap layer) with THAT kernel's roundings
    (conv3x3_reflect_wino_f16)."""
    enc = weights['encoder']
    want = set(targets)
    deepest = sorted(want)[-1]                       # model.py:60
    x = conv1x1(np.asarray(img01, np.float32), *enc['preprocess'])
    feats = {}
    for layer in ENCODER_LAYERS:
        if layer[0] == 'C':
            name = layer[1]
            w = enc[name][0]
            if fp16_storage and name != 'conv1_1':
                w = _h16(w)
            if fp16_storage and wino and wino_layer(layer[2], layer[3]) and not name.endswith('_1'):
                x = conv3x3_reflect_wino_f16(x, enc[name][0], enc[name][1], relu=True, acc=np.float32)
            else:
                x = conv3x3_reflect(x, w, enc[name][1], relu=True)
            relu = 'relu' + name[4:]
            if relu in want:
                feats[relu] = x
            if relu == deepest:
                break
            if fp16_storage:
                x = _h16(x)
        else:
            x = maxpool2x2_same(x)
    return feats


def decode(feat, weights, relu_target, fp16_storage=False, wino=True):
    """Mirror decoder for `relu_target` (model.py:245-304).  fp16_storage: see encode (the decoder's input, every
    filter and every hand-over between layers in fp16; the 3-channel image stays fp32)."""
    x = np.asarray(feat, np.float32)
    if fp16_storage:
        x = _h16(x)
    params = weights['decoder'][relu_target]
    i = 0
    for kind, cin, cout, relu in decoder_layers(relu_target):
        if kind == 'C':
            w, b = params[i]
            i += 1
            if fp16_storage and wino and cout != 3 and wino_layer(cin, cout):
                x = conv3x3_reflect_wino_f16(x, w, b, relu=relu, acc=np.float32)
            else:
                x = conv3x3_reflect(x, _h16(w) if fp16_storage else w, b, relu=relu)
            if fp16_storage and cout != 3:
                x = _h16(x)
        else:
            x = upsample2x_nearest(x)
    return x


def preprocess(image):
    """wct.py:60-64 (batch dim dropped: the oracle works on HxWx3)."""
    return np.asarray(image, np.float64) / 255.


def postprocess(image01):
    """wct.py:66-68: clip to [0,1], *255, truncate to uint8."""
    return np.uint8(np.clip(image01, 0, 1) * 255)


def stylize(content, style, weights, relu_targets, alpha=1.0, adain=False,
            wct_mode='tf', return_levels=False, swap5=False, ss_alpha=0.6, ss_patch_size=3, ss_stride=1,
            fp16_storage=False):
    """One WCT.predict (wct.py:70-106) through the test-mode graph
    (model.py:33-94,123-176): ONE style pass with all taps; levels in
    `relu_targets` order; level i>0 encodes clip(previous decoded, 0, 1)
    (model.py:86); output = last decoded, postprocessed.

    wct_mode 'tf' is what the live graph runs (wct_tf, model.py:154,158);
    'np' swaps in the wct_np semantics the north star names as oracle.
    """
    c01 = np.float32(preprocess(content))
    s01 = np.float32(preprocess(style))
    style_feats = encode(s01, weights, relu_targets, fp16_storage)     # fp16_storage: see encode (not a reference mode)
    x = c01
    levels = []
    for i, relu in enumerate(relu_targets):
        if i > 0:
            x = np.clip(x, 0, 1)
        fc = encode(x, weights, [relu], fp16_storage)[relu]
        fs = style_feats[relu]
        if swap5 and relu == 'relu5_1':          # tf.case priority swap5 > adain > wct (model.py:148-154)
            t = wct_oracle.wct_style_swap(fc, fs, ss_alpha, ss_patch_size, ss_stride)[0]
        elif adain:
            t = wct_oracle.adain(fc, fs, alpha)[0]
        elif wct_mode == 'tf':
            t = wct_oracle.wct_tf(fc, fs, alpha)[0]
        else:
            t = wct_oracle.wct_np(fc, fs, alpha)[0]
        x = decode(t, weights, relu, fp16_storage)
        if return_levels:
            levels.append((fc, fs, t, x))
    out = postprocess(x)
    return (out, levels) if return_levels else out
