"""The ERROR side of a reduced-FLOP 3x3 convolution, measured on the CPU before a GPU minute is spent.

For every stride-1 layer with >= 256 input and output channels of the 512 x 512 chain (encoder conv3_2-4 / conv4_2-4 / conv5_1 and
the decoder-5 layers, teacher-forced: each layer gets the activations the fp16-storage restatement hands it), the relative error of
the layer's output (after bias and ReLU) against the float64 convolution of the SAME fp16 input with the fp32 filters:

  direct     fp16 filters x fp16 activations, exact accumulation             -- what conv3x3_mfma_kernel computes
  wino-1d    F(2,3) along y, direct along x: filters U[f][kx] = G g[:, kx] rounded to fp16; activation rows T[f] = B^T d computed
             from the fp16 activations and rounded to fp16 (one add per value); 12 products per 2 outputs instead of 18
  wino-2d    F(2x2,3x3): U = G g G^T and V = B^T d B rounded to fp16 (V with ONE rounding, from fp32 intermediate, and with TWO,
             fp16 after each 1-D pass); 16 products per 4 outputs instead of 36

Accumulation is exact (float64) in every arm: the fp32 accumulate of the MFMA adds ~1e-6.  Gate of the review: <= 1.5e-3 per layer.
usage: python tools/probe/winograd_error.py [size=512] [he|contractive|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from oracle import net_oracle as no
from oracle.contractive import contractive_weights
from wct_tf_amd.weights import synthetic_weights, synthetic_image

torch.set_num_threads(8)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def h16(a):
    return np.asarray(a, np.float16).astype(np.float64)


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def conv_valid(xp, w):
    """xp: [H+kh-1, W+kw-1, Cin] float64 (already padded), w: [kh, kw, Cin, Cout] -> [H, W, Cout] float64."""
    t = torch.from_numpy(np.ascontiguousarray(xp.transpose(2, 0, 1)))[None]
    k = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))
    return F.conv2d(t, k)[0].permute(1, 2, 0).numpy()


def layer_errors(x16, w, b):
    """x16: HxWxCin (values exactly representable in fp16), w: 3x3xCinxCout fp32, b: Cout."""
    h, wd, cin = x16.shape
    xp = np.pad(np.float64(x16), ((1, 1), (1, 1), (0, 0)), mode='reflect')
    w64 = np.float64(w)
    fin = lambda y: np.maximum(y + np.float64(b), 0)
    ref = fin(conv_valid(xp, w64))
    out = {}
    out['direct'] = rel(fin(conv_valid(xp, h16(w64))), ref)
    # ---- 1-D F(2,3) along y: pair-row r = output rows 2r, 2r+1 <- padded rows 2r .. 2r+3
    U = h16(np.einsum('fk,kxio->fxio', G, w64))                    # [4][3][cin][cout]
    rows = [xp[i:i + h:2] for i in range(4)]                        # each [h/2, wd+2, cin]: padded row 2r+i
    T = [h16(sum(BT[f, i] * rows[i] for i in range(4))) for f in range(4)]
    M = [conv_valid(T[f], U[f][None]) for f in range(4)]            # 1 x 3 kernels
    y = np.empty((h, wd, w.shape[3]))
    y[0::2] = M[0] + M[1] + M[2]
    y[1::2] = M[1] - M[2] - M[3]
    out['wino-1d'] = rel(fin(y), ref)
    out['wino-1d max|T|/max|x|'] = float(max(np.abs(t).max() for t in T) / np.abs(x16).max())
    # ---- 2-D F(2x2,3x3)
    U2 = h16(np.einsum('fk,gl,klio->fgio', G, G, w64))              # [4][4][cin][cout]
    for tag, two in (('wino-2d (one rounding)', False), ('wino-2d (two roundings)', True)):
        R = [sum(BT[f, i] * rows[i] for i in range(4)) for f in range(4)]          # rows transformed: [h/2, wd+2, cin]
        if two:
            R = [h16(r) for r in R]
        y = np.zeros((h, wd, w.shape[3]))
        Mfg = {}
        for f in range(4):
            cols = [R[f][:, j:j + wd:2] for j in range(4)]          # [h/2, wd/2, cin]: padded column 2c+j
            for g in range(4):
                V = h16(sum(BT[g, j] * cols[j] for j in range(4)))
                Mfg[f, g] = V.reshape(-1, cin) @ U2[f, g]
        for a in range(2):
            for c in range(2):
                acc = 0
                for f in range(4):
                    for g in range(4):
                        if AT[a, f] != 0 and AT[c, g] != 0:
                            acc = acc + AT[a, f] * AT[c, g] * Mfg[f, g]
                y[a::2, c::2] = acc.reshape(h // 2, wd // 2, -1)
        out[tag] = rel(fin(y), ref)
    return out


def chain_inputs(weights, size):
    """(name, x16, w, b) of every wide stride-1 layer, teacher-forced through the fp16-storage restatement."""
    img = np.float32(synthetic_image(1000, size, size)) / np.float32(255)
    enc = weights['encoder']
    x = no.conv1x1(img, *enc['preprocess'])
    out = []
    for layer in no.ENCODER_LAYERS:
        if layer[0] == 'C':
            name = layer[1]
            w, b = enc[name]
            if layer[2] >= 256 and layer[3] >= 256:
                out.append((name, x.copy(), w, b))
            x = no._h16(no.conv3x3_reflect(x, w if name == 'conv1_1' else no._h16(w), b))
        else:
            x = no.maxpool2x2_same(x)
    feat = x
    params = weights['decoder']['relu5_1']
    i = 0
    for kind, cin, cout, relu in no.decoder_layers('relu5_1'):
        if kind == 'C':
            w, b = params[i]
            if cin >= 256 and cout >= 256:
                out.append(('dec5_%d %dx%d %d->%d' % (i, x.shape[0], x.shape[1], cin, cout), x.copy(), w, b))
            i += 1
            x = no._h16(no.conv3x3_reflect(x, no._h16(w), b, relu=relu)) if cout != 3 else x
        else:
            x = no.upsample2x_nearest(x)
    return out


if __name__ == '__main__':
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    which = sys.argv[2] if len(sys.argv) > 2 else 'both'
    sets = []
    if which in ('he', 'both'):
        sets.append(('He-normal stand-in (seed 42)', synthetic_weights(42)))
    if which in ('contractive', 'both'):
        sets.append(('contractive net (seed 7)', contractive_weights(7)))
    for name, w in sets:
        print(name, flush=True)
        for lname, x16, wt, b in chain_inputs(w, size):
            e = layer_errors(x16, wt, b)
            print('  %-28s %4dx%-4d %3d->%3d: direct %.2e | wino-1d %.2e (|T| %.2fx) | wino-2d %.2e / %.2e' % (
                lname, x16.shape[0], x16.shape[1], wt.shape[2], wt.shape[3], e['direct'], e['wino-1d'], e['wino-1d max|T|/max|x|'],
                e['wino-2d (one rounding)'], e['wino-2d (two roundings)']), flush=True)
