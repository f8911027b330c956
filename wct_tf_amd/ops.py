"""Op-level mirror of the reference's ops.py / coral.py / utils.py surface, running
on libwct_hip.so.  Same names, argument meaning and array conventions:

  wct_np(content, style, alpha=0.6, eps=1e-5)   ops.py:92   1xHxWxC in, float32 1xHxWxC out
  wct_tf(content, style, alpha, eps=1e-8)       ops.py:24
  adain(content_features, style_features, alpha, epsilon=1e-5)   ops.py:282
  coral_numpy(source, target)                   coral.py:13 (float64)
  preserve_colors_np(style_rgb, content_rgb)    utils.py:87
"""
import numpy as np

from . import _lib
from .context import default_context


def _flat(feat):
    f = np.squeeze(np.asarray(feat))            # batch must be 1 (ops.py:32-33,98-99)
    if f.ndim != 3:
        raise ValueError('WCT needs a single 1xHxWxC feature map, got shape %s' % (np.shape(feat),))
    h, w, c = f.shape
    return np.ascontiguousarray(f.reshape(h * w, c), np.float32), (h, w, c)


def wct_np(content, style, alpha=0.6, eps=1e-5, ctx=None):
    ctx = ctx or default_context()
    fc, (h, w, c) = _flat(content)
    fs, _ = _flat(style)
    out = ctx.transform(fc, fs, alpha, _lib.WCT_NP, eps)
    return out.reshape(1, h, w, c)


def wct_tf(content, style, alpha, eps=1e-8, ctx=None):
    ctx = ctx or default_context()
    fc, (h, w, c) = _flat(content)
    fs, _ = _flat(style)
    out = ctx.transform(fc, fs, alpha, _lib.WCT_TF, eps)
    return out.reshape(1, h, w, c)


def adain(content_features, style_features, alpha, epsilon=1e-5, ctx=None):
    ctx = ctx or default_context()
    x = np.asarray(content_features, np.float32)
    s = np.asarray(style_features, np.float32)
    if x.ndim == 3:
        x = x[None]
    if s.ndim == 3:
        s = s[None]
    if x.shape[0] != 1 or s.shape[0] != 1:
        raise ValueError('batch must be 1')
    _, h, w, c = x.shape
    out = ctx.adain(x.reshape(h * w, c), s.reshape(-1, c), alpha, epsilon)
    return out.reshape(1, h, w, c)


def wct_style_swap(content, style, alpha, patch_size=3, stride=1, eps=1e-8, ctx=None):
    """ops.py:145: 1xHxWxC content/style encodings -> float32 1xHxWxC"""
    ctx = ctx or default_context()
    c = np.squeeze(np.asarray(content, np.float32))
    s = np.squeeze(np.asarray(style, np.float32))
    if c.ndim != 3 or s.ndim != 3:
        raise ValueError('style swap needs single 1xHxWxC feature maps')
    return ctx.style_swap(c, s, alpha, patch_size, stride, eps)[None]


def _moments_from_sums(sums, npix):
    """mean / population std / (Xn Xn^T + I) of img/255 from the exact integer moments."""
    s1 = sums[:3] / 255.0
    idx = {(0, 0): 3, (0, 1): 4, (0, 2): 5, (1, 1): 6, (1, 2): 7, (2, 2): 8}
    s2 = np.empty((3, 3))
    for (i, j), k in idx.items():
        s2[i, j] = s2[j, i] = sums[k] / (255.0 * 255.0)
    mean = s1 / npix
    cov = s2 - npix * np.outer(mean, mean)           # sum (x-m)(x-m)^T
    std = np.sqrt(np.diag(cov) / npix)                # population std (coral.py:23,27)
    cov_eye = cov / np.outer(std, std) + np.eye(3)    # coral.py:30-31 (not divided by N)
    return mean, std, cov_eye


def _mat_sqrt(x):
    # coral.py:8-11 verbatim in meaning: U sqrt(D) V.T with V = numpy's third SVD output (V^H);
    # the value depends on LAPACK's singular-vector signs, hence LAPACK on the host (see
    # include/wct_hip.h, wct_coral_stats).
    u, d, v = np.linalg.svd(x)
    return u.dot(np.diag(np.sqrt(d))).dot(v.T)


def _coral(source_u8, target_u8, ctx, want_f64, want_u8):
    src = np.ascontiguousarray(source_u8, np.uint8)
    tgt = np.ascontiguousarray(target_u8, np.uint8)
    if src.ndim != 3 or src.shape[2] != 3 or tgt.ndim != 3 or tgt.shape[2] != 3:
        raise ValueError('CORAL needs HxWx3 images')
    ms, ss, cs = _moments_from_sums(ctx.coral_stats(src), src.shape[0] * src.shape[1])
    mt, st, ct = _moments_from_sums(ctx.coral_stats(tgt), tgt.shape[0] * tgt.shape[1])
    m = _mat_sqrt(ct).dot(np.linalg.inv(_mat_sqrt(cs)))          # coral.py:33
    return ctx.coral_apply(src, m, ms, ss, mt, st, want_f64=want_f64, want_u8=want_u8)


def coral_numpy(source, target, ctx=None):
    """source/target: HxWx3 images in [0,1] that are uint8 images / 255 (the only way the
    reference calls it, utils.py:88).  Returns float64 HxWx3."""
    ctx = ctx or default_context()
    s8 = np.rint(np.asarray(source, np.float64) * 255.0)
    t8 = np.rint(np.asarray(target, np.float64) * 255.0)
    if np.abs(s8 / 255.0 - source).max() > 1e-12 or np.abs(t8 / 255.0 - target).max() > 1e-12:
        raise ValueError('coral_numpy on the GPU path takes uint8-valued images / 255 (utils.py:88)')
    return _coral(s8.astype(np.uint8), t8.astype(np.uint8), ctx, True, False)[1]


def preserve_colors_np(style_rgb, content_rgb, ctx=None):
    ctx = ctx or default_context()
    return _coral(style_rgb, content_rgb, ctx, False, True)[0]
