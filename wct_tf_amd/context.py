"""Python handle on one wct_ctx (one GPU, one HIP stream) -- the session object
that replaces the reference's tf.Session (wct.py:29-58).  Every method is a
thin ctypes call into libwct_hip.so; no arithmetic of the path happens here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f32, fptr, u8
from .weights import ENCODER_CONVS, RELU_LEVEL, decoder_plan

_LEVEL_C = {1: 64, 2: 128, 3: 256, 4: 512, 5: 512}


def _levels(relu_targets):
    out = []
    for r in relu_targets:
        if isinstance(r, str):
            if r not in RELU_LEVEL:
                raise ValueError('unknown relu target %r' % (r,))
            out.append(RELU_LEVEL[r])
        else:
            out.append(int(r))
    return out


class Context(object):
    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.wct_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.loaded_decoders = set()
        self.encoder_loaded = False

    def close(self):
        if getattr(self, 'h', None):
            self.lib.wct_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.wct_sync(self.h))

    def stream_handle(self):
        """The context's hipStream_t as an integer (torch.cuda.ExternalStream(handle) wraps it)."""
        p = C.c_void_p()
        check(self.lib.wct_get_stream(self.h, C.byref(p)))
        return p.value or 0

    # ---- weights ---------------------------------------------------------
    def set_encoder(self, enc):
        pre_w, pre_b = enc['preprocess']
        pw = f32(np.asarray(pre_w).reshape(3, 3))
        pb = f32(pre_b)
        ws = [f32(enc[name][0]) for name, _, _ in ENCODER_CONVS]
        bs = [f32(enc[name][1]) for name, _, _ in ENCODER_CONVS]
        for (name, cin, cout), w in zip(ENCODER_CONVS, ws):
            if w.shape != (3, 3, cin, cout):
                raise ValueError('%s: expected HWIO %s, got %s' % (name, (3, 3, cin, cout), w.shape))
        check(self.lib.wct_set_encoder(self.h, fptr(pw), fptr(pb), _lib.ptr_array(ws), _lib.ptr_array(bs), len(ws)))
        self.encoder_loaded = True

    def set_decoder(self, relu_target, layers):
        level = _levels([relu_target])[0]
        plan = [p for p in decoder_plan('relu%d_1' % level) if p[0] == 'C']
        if len(layers) != len(plan):
            raise ValueError('decoder relu%d_1 needs %d conv layers, got %d' % (level, len(plan), len(layers)))
        ws = [f32(w) for w, _ in layers]
        bs = [f32(b) for _, b in layers]
        for (_, cin, cout, _), w in zip(plan, ws):
            if w.shape != (3, 3, cin, cout):
                raise ValueError('decoder relu%d_1: expected HWIO %s, got %s' % (level, (3, 3, cin, cout), w.shape))
        check(self.lib.wct_set_decoder(self.h, level, _lib.ptr_array(ws), _lib.ptr_array(bs), len(ws)))
        self.loaded_decoders.add(level)

    def set_weights(self, weights):
        self.set_encoder(weights['encoder'])
        for relu, layers in weights['decoder'].items():
            self.set_decoder(relu, layers)

    # ---- op level ----------------------------------------------------------
    def transform(self, content, style, alpha, mode, eps=-1.0, return_sweeps=False):
        """content [Nc][C], style [Ns][C] float32 -> [Nc][C]"""
        c = f32(content)
        s = f32(style)
        if c.ndim != 2 or s.ndim != 2 or c.shape[1] != s.shape[1]:
            raise ValueError('expected [N][C] feature matrices with equal C')
        out = np.empty_like(c)
        sweeps = (C.c_int * 2)()
        self.last_sweeps = sweeps            # negative entries: that eigensolve failed (the call raises WCTNotConverged)
        check(self.lib.wct_transform(self.h, fptr(c), c.shape[0], fptr(s), s.shape[0], c.shape[1],
                                     float(alpha), int(mode), float(eps), fptr(out), sweeps))
        return (out, list(sweeps)) if return_sweeps else out

    def adain(self, content, style, alpha, epsilon=1e-5):
        c = f32(content)
        s = f32(style)
        out = np.empty_like(c)
        check(self.lib.wct_adain(self.h, fptr(c), c.shape[0], fptr(s), s.shape[0], c.shape[1],
                                 float(alpha), float(epsilon), fptr(out)))
        return out

    def style_swap(self, content, style, alpha, patch_size=3, stride=1, eps=-1.0):
        """content [hc][wc][C], style [hs][ws][C] float32 -> [hc][wc][C] (ops.py:145-278)"""
        c = f32(content)
        s = f32(style)
        out = np.empty_like(c)
        check(self.lib.wct_style_swap(self.h, fptr(c), c.shape[0], c.shape[1], fptr(s), s.shape[0], s.shape[1],
                                      c.shape[2], float(alpha), int(patch_size), int(stride), float(eps), fptr(out)))
        return out

    def set_style_swap(self, ss_alpha=0.6, patch_size=3, stride=1):
        check(self.lib.wct_set_style_swap(self.h, float(ss_alpha), int(patch_size), int(stride)))

    def eigh(self, mats, return_sweeps=False):
        a = f32(mats)
        if a.ndim == 2:
            a = a[None]
        # wct_eigh wants a matrix that is symmetric to the bit (include/wct_hip.h); (a + a^T) / 2 is, and leaves a symmetric
        # input unchanged
        a = np.ascontiguousarray(0.5 * (a + a.transpose(0, 2, 1)), np.float32)
        n, c, _ = a.shape
        evals = np.empty((n, c), np.float32)
        evecs = np.empty((n, c, c), np.float32)
        sweeps = (C.c_int * n)()
        self.last_sweeps = sweeps
        check(self.lib.wct_eigh(self.h, fptr(a), c, n, fptr(evals), fptr(evecs), sweeps))
        return (evals, evecs, list(sweeps)) if return_sweeps else (evals, evecs)

    def conv3x3(self, x, w_hwio, bias, relu=True, upsample=False):
        x = f32(x)
        w = f32(w_hwio)
        b = f32(bias)
        h, wd, cin = x.shape
        cout = w.shape[3]
        s = 2 if upsample else 1
        y = np.empty((h * s, wd * s, cout), np.float32)
        check(self.lib.wct_conv3x3(self.h, fptr(x), h, wd, cin, fptr(w), fptr(b), cout, int(relu), int(upsample), fptr(y)))
        return y

    def conv3x3_f16(self, x, w_hwio, bias, relu=True, upsample=False, pool=False, algo=0):
        """One 3x3 layer as the stylize pipeline runs it (fp16 activations out, optional fused 'same' max-pool) on a batch
        x [B][H][W][Cin] (or [H][W][Cin]); algo 0 = the pipeline's kernel for this shape, 1 = direct, 2 = Winograd F(2,3)."""
        x = f32(x)
        single = x.ndim == 3
        if single:
            x = x[None]
        w = f32(w_hwio)
        b = f32(bias)
        n, h, wd, cin = x.shape
        cout = w.shape[3]
        s = 2 if upsample else 1
        ho, wo = h * s, wd * s
        if pool:
            ho, wo = (ho + 1) // 2, (wo + 1) // 2
        y = np.empty((n, ho, wo, cout), np.float32)
        check(self.lib.wct_conv3x3_f16(self.h, fptr(x), n, h, wd, cin, fptr(w), fptr(b), cout, int(relu), int(upsample), int(pool),
                                       int(algo), fptr(y)))
        return y[0] if single else y

    def maxpool(self, x):
        x = f32(x)
        h, w, c = x.shape
        y = np.empty(((h + 1) // 2, (w + 1) // 2, c), np.float32)
        check(self.lib.wct_maxpool(self.h, fptr(x), h, w, c, fptr(y)))
        return y

    def encode(self, img01, relu_target):
        level = _levels([relu_target])[0]
        x = f32(img01)
        h, w, _ = x.shape
        hh, ww = h, w
        for _ in range(level - 1):
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        feat = np.empty((hh, ww, _LEVEL_C[level]), np.float32)
        check(self.lib.wct_encode(self.h, fptr(x), h, w, level, fptr(feat)))
        return feat

    def decode(self, feat, relu_target):
        level = _levels([relu_target])[0]
        f = f32(feat)
        h, w, _ = f.shape
        s = 1 << (level - 1)
        img = np.empty((h * s, w * s, 3), np.float32)
        check(self.lib.wct_decode(self.h, fptr(f), h, w, level, fptr(img)))
        return img

    def coral_stats(self, img_u8):
        a = u8(img_u8)
        sums = (C.c_double * 9)()
        check(self.lib.wct_coral_stats(self.h, a.ctypes.data_as(_lib._U8), a.shape[0], a.shape[1], sums))
        return np.array(list(sums), np.float64)

    def coral_apply(self, src_u8, m, src_mean, src_std, tgt_mean, tgt_std, want_f64=True, want_u8=True):
        a = u8(src_u8)
        h, w, _ = a.shape
        dd = lambda v: np.ascontiguousarray(v, np.float64).ctypes.data_as(_lib._D)
        arrs = [np.ascontiguousarray(v, np.float64).reshape(-1) for v in (m, src_mean, src_std, tgt_mean, tgt_std)]
        out8 = np.empty((h, w, 3), np.uint8) if want_u8 else None
        out64 = np.empty((h, w, 3), np.float64) if want_f64 else None
        check(self.lib.wct_coral_apply(
            self.h, a.ctypes.data_as(_lib._U8), h, w, *[x.ctypes.data_as(_lib._D) for x in arrs],
            out8.ctypes.data_as(_lib._U8) if want_u8 else None,
            out64.ctypes.data_as(_lib._D) if want_f64 else None))
        return out8, out64

    # ---- the hot path ------------------------------------------------------
    def output_size(self, hc, wc, relu_targets):
        lv = _levels(relu_targets)
        arr = (C.c_int * len(lv))(*lv)
        ho, wo = C.c_int(), C.c_int()
        check(self.lib.wct_output_size(hc, wc, arr, len(lv), C.byref(ho), C.byref(wo)))
        return ho.value, wo.value

    def stylize(self, content, style, relu_targets, alpha=1.0, adain=False, wct_mode='tf', swap5=False):
        """One predict(): HxWx3 images in [0,255] in, uint8 out.  uint8 inputs go to the library as they are (the /255
        runs on the device); anything else is preprocessed exactly as the reference does -- `image / 255.` in float64
        (wct.py:60-64), cast to the float32 the graph's placeholders hold (model.py:43-44) -- and handed over as float32
        images in [0,1] (WCT_FLAG_IMAGES_F32): a float image is NOT rounded to integer levels."""
        content, style = np.asarray(content), np.asarray(style)
        as_f32 = content.dtype != np.uint8 or style.dtype != np.uint8
        if as_f32:
            c = np.ascontiguousarray(np.asarray(content / 255.), np.float32)
            s = np.ascontiguousarray(np.asarray(style / 255.), np.float32)
        else:
            c, s = u8(content), u8(style)
        lv = _levels(relu_targets)
        arr = (C.c_int * len(lv))(*lv)
        ho, wo = self.output_size(c.shape[0], c.shape[1], lv)
        out = np.empty((ho, wo, 3), np.uint8)
        flags = (_lib.FLAG_ADAIN if adain else 0) | (_lib.FLAG_MODE_NP if wct_mode == 'np' else 0) | \
            (_lib.FLAG_SWAP5 if swap5 else 0) | (_lib.FLAG_IMAGES_F32 if as_f32 else 0)
        check(self.lib.wct_stylize(self.h, c.ctypes.data_as(_lib._U8), c.shape[0], c.shape[1],
                                   s.ctypes.data_as(_lib._U8), s.shape[0], s.shape[1], arr, len(lv),
                                   float(alpha), flags, out.ctypes.data_as(_lib._U8)))
        return out

    # device-resident batch (what bench.py times)
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.wct_dev_alloc(self.h, int(nbytes), C.byref(p)))
        return p

    def dev_free(self, p):
        check(self.lib.wct_dev_free(self.h, p))

    def h2d(self, dst, arr):
        a = np.ascontiguousarray(arr)
        check(self.lib.wct_h2d(self.h, dst, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def d2h(self, arr, src):
        assert arr.flags['C_CONTIGUOUS']
        check(self.lib.wct_d2h(self.h, arr.ctypes.data_as(C.c_void_p), src, arr.nbytes))

    def stylize_batch_dev(self, content_dev, hc, wc, style_dev, hs, ws, batch, relu_targets, alpha, out_dev,
                          adain=False, wct_mode='tf', swap5=False, shared_style=False):
        """B pairs resident in HBM.  shared_style: style_dev holds ONE image used for every pair (fixed-style
        video): its encoder pass, statistics and eigensystems run once per call; the frames are bit-identical
        to the ones a replicated style produces."""
        lv = _levels(relu_targets)
        arr = (C.c_int * len(lv))(*lv)
        flags = (_lib.FLAG_ADAIN if adain else 0) | (_lib.FLAG_MODE_NP if wct_mode == 'np' else 0) | \
            (_lib.FLAG_SWAP5 if swap5 else 0) | (_lib.FLAG_STYLE_SHARED if shared_style else 0)
        check(self.lib.wct_stylize_batch_dev(self.h, content_dev, hc, wc, style_dev, hs, ws, batch, arr, len(lv),
                                             float(alpha), flags, out_dev))

    def stylize_batch(self, contents_u8, style_u8, relu_targets, alpha=1.0, adain=False, wct_mode='tf', swap5=False):
        """Host arrays in, host array out: contents [B][H][W][3] uint8 (B <= 32); style either one image
        [Hs][Ws][3] shared by all frames or [B][Hs][Ws][3].  Returns [B][Ho][Wo][3] uint8."""
        c = u8(contents_u8)
        s = u8(style_u8)
        assert c.ndim == 4 and c.shape[3] == 3 and s.ndim in (3, 4) and s.shape[-1] == 3
        shared = s.ndim == 3
        assert shared or s.shape[0] == c.shape[0]
        B, hc, wc = c.shape[:3]
        hs, ws = s.shape[-3], s.shape[-2]
        ho, wo = self.output_size(hc, wc, relu_targets)
        out = np.empty((B, ho, wo, 3), np.uint8)
        dc, ds, do = self.dev_alloc(c.nbytes), self.dev_alloc(s.nbytes), self.dev_alloc(out.nbytes)
        try:
            self.h2d(dc, c)
            self.h2d(ds, s)
            self.stylize_batch_dev(dc, hc, wc, ds, hs, ws, B, relu_targets, alpha, do, adain=adain,
                                   wct_mode=wct_mode, swap5=swap5, shared_style=shared)
            self.sync()
            self.d2h(out, do)
        finally:
            for p in (dc, ds, do):
                self.dev_free(p)
        return out

    # ---- decoder training (model.py:123-223) --------------------------------
    def train_step(self, relu_target, images01, step, learning_rate=1e-4, feature_weight=1.0, pixel_weight=1.0,
                   tv_weight=0.0, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """One Adam step on the decoder of `relu_target`; images01 [B][H][W][3] fp32 in [0,1].
        Returns {'feature_loss', 'pixel_loss', 'tv_loss', 'total_loss'}.  learning_rate=0 only evaluates."""
        level = _levels([relu_target])[0]
        x = f32(images01)
        assert x.ndim == 4 and x.shape[3] == 3
        out = (C.c_float * 4)()
        check(self.lib.wct_train_step(self.h, level, x.ctypes.data_as(_lib._F), x.shape[0], x.shape[1], x.shape[2],
                                      float(feature_weight), float(pixel_weight), float(tv_weight), float(learning_rate),
                                      float(beta1), float(beta2), float(epsilon), int(step), out))
        return {'feature_loss': out[0], 'pixel_loss': out[1], 'tv_loss': out[2], 'total_loss': out[3]}

    def train_grad_buffer(self, relu_target):
        """(device pointer, float count) of the decoder's contiguous gradient buffer (data-parallel all-reduce)."""
        level = _levels([relu_target])[0]
        p, n = C.c_void_p(), C.c_size_t()
        check(self.lib.wct_train_grad_buffer(self.h, level, C.byref(p), C.byref(n)))
        return p.value, n.value

    def train_apply(self, relu_target, step, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """Adam from the gradients currently in the gradient buffer (after train_step(learning_rate=0) and an
        all-reduce): the second half of a data-parallel step."""
        level = _levels([relu_target])[0]
        check(self.lib.wct_train_apply(self.h, level, float(learning_rate), float(beta1), float(beta2), float(epsilon), int(step)))

    def get_decoder(self, relu_target, grads=False):
        """[(w HWIO fp32, b)] of the decoder as it is now on the device (after training steps); with grads=True
        the gradients of the last train_step instead."""
        level = _levels([relu_target])[0]
        out = []
        for i, (_, cin, cout, _) in enumerate(p for p in decoder_plan('relu%d_1' % level) if p[0] == 'C'):
            w = np.empty((3, 3, cin, cout), np.float32)
            b = np.empty(cout, np.float32)
            wp, bp = w.ctypes.data_as(_lib._F), b.ctypes.data_as(_lib._F)
            if grads:
                check(self.lib.wct_get_decoder_layer(self.h, level, i, None, None, wp, bp))
            else:
                check(self.lib.wct_get_decoder_layer(self.h, level, i, wp, bp, None, None))
            out.append((w, b))
        return out

    # ---- measurement -------------------------------------------------------
    def prof_enable(self, on=True):
        check(self.lib.wct_prof_enable(self.h, int(on)))

    def prof_reset(self):
        check(self.lib.wct_prof_reset(self.h))

    def prof_read(self):
        n = len(_lib.PROF_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_longlong * n)()
        fl = (C.c_double * n)()
        by = (C.c_double * n)()
        check(self.lib.wct_prof_read(self.h, ms, cnt, fl, by))
        return {name: {'ms': ms[i], 'launches': cnt[i], 'flops': fl[i], 'bytes': by[i]}
                for i, name in enumerate(_lib.PROF_CLASSES)}

    def eig_stats(self):
        """Eigensolver statistics since the last call: {C: {'matrices', 'sweeps', 'max_sweeps'}} for the covariance
        orders C = 32 * 2^k that were solved (wct_eig_stats; synchronises the stream, cleared on read)."""
        out = (C.c_longlong * 18)()
        check(self.lib.wct_eig_stats(self.h, out))
        return {32 << k: {'matrices': out[3 * k], 'sweeps': out[3 * k + 1], 'max_sweeps': out[3 * k + 2]}
                for k in range(6) if out[3 * k]}


_default = {}


def default_context(device=0):
    """Process-wide context per device (the reference has one session per WCT object;
    the op-level functions in ops.py share this one)."""
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
