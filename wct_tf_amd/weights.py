"""Weight container for the stylize path + synthetic-weight generator.

The reference loads the encoder from `vgg_normalised.t7` (vgg_normalised.py:16,
33-34: OIHW -> HWIO) and the decoders from TF checkpoints (wct.py:46-58).
Neither file exists offline, so the container is format-neutral: plain float32
HWIO arrays + biases, saved/loaded as one .npz.  Real weights drop in by
filling the same dict.

  weights['encoder'][name] = (w[kh,kw,cin,cout], b[cout]); names 'preprocess',
                             'conv1_1' ... 'conv5_1'
  weights['decoder'][relu] = [(w, b), ...] in execution order (model.py:283-298)
"""
import numpy as np

ENCODER_CONVS = [
    ('conv1_1', 3, 64), ('conv1_2', 64, 64),
    ('conv2_1', 64, 128), ('conv2_2', 128, 128),
    ('conv3_1', 128, 256), ('conv3_2', 256, 256), ('conv3_3', 256, 256), ('conv3_4', 256, 256),
    ('conv4_1', 256, 512), ('conv4_2', 512, 512), ('conv4_3', 512, 512), ('conv4_4', 512, 512),
    ('conv5_1', 512, 512),
]
RELU_TARGETS = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
RELU_CHANNELS = {'relu1_1': 64, 'relu2_1': 128, 'relu3_1': 256, 'relu4_1': 512, 'relu5_1': 512}
RELU_LEVEL = {'relu1_1': 1, 'relu2_1': 2, 'relu3_1': 3, 'relu4_1': 4, 'relu5_1': 5}

# model.py:255-277 as conv output widths / 'U' markers
_DEC_ARCH = {5: [512, 'U', 512, 512, 512], 4: [256, 'U', 256, 256, 256],
             3: [128, 'U', 128], 2: [64, 'U'], 1: [64]}


# zero-sum filters only see the variance (0.34) of a rectified unit Gaussian, not
# its second moment (0.5): this gain keeps the activation scale level per layer
_CENTRED_GAIN = 1.21


def decoder_plan(relu_target):
    """[(kind, cin, cout, relu)], kind 'C' | 'U' -- mirrors model.py:283-298."""
    cin = RELU_CHANNELS[relu_target]
    plan = []
    for d in range(RELU_LEVEL[relu_target], 0, -1):
        for item in _DEC_ARCH[d]:
            if item == 'U':
                plan.append(('U', cin, cin, False))
            else:
                plan.append(('C', cin, item, True))
                cin = item
    plan.append(('C', cin, 3, False))
    return plan


def synthetic_weights(seed=42, relu_targets=None):
    """Seeded He-normal stand-ins for the absent pre-trained weights.

    * 'preprocess' has its documented role (vgg_normalised.py:26): x255,
      RGB->BGR, subtract the BGR mean.
    * conv1_1 is He-normal scaled by 1/64 so relu1_1 features have O(1) scale
      (the real network is "normalised" to unit mean activation); deeper
      layers are plain He-normal N(0, 2/(9 cin)), which preserves that scale.
    * each decoder's last conv is scaled so images land inside [0,1] with some
      spread (bias 0.5), otherwise the clip between levels (model.py:86) would
      saturate everything and hide errors.
    Biases are small non-zero normals so the bias path is exercised.
    """
    rng = np.random.default_rng(seed)
    relu_targets = list(relu_targets or RELU_TARGETS)
    enc = {}
    pre_w = np.zeros((1, 1, 3, 3), np.float32)
    for c in range(3):
        pre_w[0, 0, 2 - c, c] = 255.0          # out channel c <- in channel 2-c
    pre_b = -np.array([103.939, 116.779, 123.68], np.float32)
    enc['preprocess'] = (pre_w, pre_b)
    for name, cin, cout in ENCODER_CONVS:
        std = np.sqrt(2.0 / (9 * cin))
        if name == 'conv1_1':
            std /= 64.0
        w = (rng.standard_normal((3, 3, cin, cout)) * std).astype(np.float32)
        if cin > 3:
            # post-ReLU inputs have a large positive mean; zero-sum filters keep
            # it from killing half of the output channels (dead channels give a
            # continuum of near-zero covariance eigenvalues around the 1e-5
            # cut-off of ops.py:68-69, which makes ANY two implementations --
            # including fp32 vs fp64 LAPACK -- disagree)
            w -= w.mean(axis=(0, 1, 2), keepdims=True)
            w *= np.float32(_CENTRED_GAIN)
        b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
        enc[name] = (w, b)
    dec = {}
    for relu in RELU_TARGETS:                    # fixed order => seed-stable
        layers = []
        plan = [p for p in decoder_plan(relu) if p[0] == 'C']
        for i, (_, cin, cout, act) in enumerate(plan):
            std = np.sqrt(2.0 / (9 * cin))
            if i == len(plan) - 1:
                std = 0.25 / np.sqrt(9 * cin)
            w = (rng.standard_normal((3, 3, cin, cout)) * std).astype(np.float32)
            if i > 0:
                w -= w.mean(axis=(0, 1, 2), keepdims=True)
                if i < len(plan) - 1:
                    w *= np.float32(_CENTRED_GAIN)
            b = (rng.standard_normal(cout) * 0.05).astype(np.float32)
            if i == len(plan) - 1:
                b = b + np.float32(0.5)
            layers.append((w, b))
        if relu in relu_targets:
            dec[relu] = layers
    return {'encoder': enc, 'decoder': dec}


def save_weights(path, weights):
    flat = {}
    for name, (w, b) in weights['encoder'].items():
        flat['enc/%s/w' % name] = w
        flat['enc/%s/b' % name] = b
    for relu, layers in weights['decoder'].items():
        for i, (w, b) in enumerate(layers):
            flat['dec/%s/%d/w' % (relu, i)] = w
            flat['dec/%s/%d/b' % (relu, i)] = b
    np.savez(path, **flat)


def load_weights(path):
    z = np.load(path)
    enc, dec = {}, {}
    for key in z.files:
        parts = key.split('/')
        if parts[0] == 'enc' and parts[2] == 'w':
            enc[parts[1]] = (z[key], z['enc/%s/b' % parts[1]])
        elif parts[0] == 'dec' and parts[3] == 'w':
            dec.setdefault(parts[1], {})[int(parts[2])] = (z[key], z['dec/%s/%s/b' % (parts[1], parts[2])])
    dec = {relu: [d[i] for i in sorted(d)] for relu, d in dec.items()}
    return {'encoder': enc, 'decoder': dec}


def synthetic_image(seed, h=512, w=512):
    """Seeded uint8 HxWx3 image: uniform noise smoothed by a 5x5 box blur so
    features are not white noise."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (h + 4, w + 4, 3)).astype(np.float32)
    c = np.cumsum(np.cumsum(np.pad(x, ((1, 0), (1, 0), (0, 0))), axis=0), axis=1)
    box = (c[5:, 5:] - c[:-5, 5:] - c[5:, :-5] + c[:-5, :-5]) / 25.0
    # stretch contrast back out (the blur shrinks the range 5x)
    box = (box - 127.5) * 4.0 + 127.5
    return np.uint8(np.clip(box, 0, 255))


def synthetic_features(seed, c, h, w, decades=3.0, rank=None):
    """Seeded fp32 post-ReLU-like 1xHxWxC features with a non-trivial spectrum:
    max(G.M, 0) with a channel-mixing matrix M and per-channel log-uniform
    scales spanning `decades`; `rank` < C makes the covariance rank-deficient
    (before the ReLU)."""
    rng = np.random.default_rng(seed)
    n = h * w
    k = rank or c
    g = rng.standard_normal((n, k)).astype(np.float32)
    m = rng.standard_normal((k, c)).astype(np.float32) / np.sqrt(k)
    scales = (10.0 ** rng.uniform(-decades / 2, decades / 2, c)).astype(np.float32)
    f = g @ m * scales + rng.uniform(0, 0.5, c).astype(np.float32)
    if rank is None:
        f = np.maximum(f, 0)
    return f.reshape(1, h, w, c).astype(np.float32)


def synthetic_features_exact(seed, c, h, w, decades=2.0):
    """Same construction as synthetic_features, with the channel mixing done in float64 before the cast to
    float32, so that two hosts with different BLAS kernels regenerate the same array to the last bit (up to rare
    round-to-nearest ties): the large golden cases of tests/golden/wct_np_sizes.npz store only a digest of the
    reference's output and rebuild their inputs from the seed."""
    rng = np.random.default_rng(seed)
    n = h * w
    g = rng.standard_normal((n, c))
    m = rng.standard_normal((c, c)) / np.sqrt(c)
    scales = 10.0 ** rng.uniform(-decades / 2, decades / 2, c)
    f = np.maximum(g @ m * scales + rng.uniform(0, 0.5, c), 0)
    return f.reshape(1, h, w, c).astype(np.float32)
