"""NumPy restatement of the reference feature transforms (TEST ORACLE ONLY).

Every function cites the reference lines it follows (paths relative to
/root/reference).  Nothing here is imported by the product path.
"""
import numpy as np


def _flatten_chw(feat):
    """1xHxWxC (or HxWxC) -> (C, H*W), channel-major, as ops.py:98-103 does."""
    f = np.squeeze(feat)
    if f.ndim != 3:
        raise ValueError("expected a single HxWxC feature map")
    h, w, c = f.shape
    return np.transpose(f, (2, 0, 1)).reshape(c, h * w), (h, w, c)


def _unflatten(mat, shape):
    h, w, c = shape
    return np.transpose(mat.reshape(c, h, w), (1, 2, 0))[None]


def wct_np(content, style, alpha=0.6, eps=1e-5, keep=None):
    """Whiten-colour transform, NumPy semantics of ops.py:92-140.

    `keep` = (kc, ks) is NOT a reference argument: it overrides the number of singular values
    kept on the content / style side (the reference keeps those > 1e-5, ops.py:112,125).  Tests
    use it only to enumerate the outcomes that are legitimate when an eigenvalue sits within
    fp32 noise of that threshold (tests/test_gpu_fuzz.py).

    * no eps on the covariance (ops.py:108,121)
    * keep singular values > 1e-5 (ops.py:112,125)
    * D_c = (w+eps)^-1/2 (ops.py:114), D_s = sqrt(w+eps) (ops.py:127)
    * blend alpha*fcs + (1-alpha)*fc : the content mean is NOT restored
      (ops.py:133)
    * result cast to float32 (ops.py:140); arithmetic dtype follows the input.
    """
    fc_full, cshape = _flatten_chw(content)
    fs_full, _ = _flatten_chw(style)
    nc = fc_full.shape[1]
    ns = fs_full.shape[1]

    mc = fc_full.mean(axis=1, keepdims=True)
    fc = fc_full - mc
    cov_c = np.dot(fc, fc.T) / (nc - 1)
    ec, wc, _ = np.linalg.svd(cov_c)
    kc = int((wc > 1e-5).sum()) if keep is None else int(keep[0])
    dc = np.diag((wc[:kc] + eps) ** -0.5)
    whitened = ec[:, :kc].dot(dc).dot(ec[:, :kc].T).dot(fc)

    ms = fs_full.mean(axis=1, keepdims=True)
    fs = fs_full - ms
    cov_s = np.dot(fs, fs.T) / (ns - 1)
    es, ws, _ = np.linalg.svd(cov_s)
    ks = int((ws > 1e-5).sum()) if keep is None else int(keep[1])
    ds = np.sqrt(np.diag(ws[:ks] + eps))
    colored = es[:, :ks].dot(ds).dot(es[:, :ks].T).dot(whitened) + ms

    blended = alpha * colored + (1 - alpha) * fc
    return np.float32(_unflatten(blended, cshape))


def wct_tf(content, style, alpha, eps=1e-8, keep=None, dtype=np.float32):
    """Whiten-colour transform, TensorFlow-graph semantics of ops.py:24-90.

    `keep` = (kc, ks): test-only override of the kept counts, see wct_np (the reference keeps
    singular values > 1e-5, ops.py:68-69).  `dtype`: test-only, np.float64 re-runs the same graph
    in double precision to measure how much of the float32 result is rounding noise.

    * eps*I added to both covariances (ops.py:45,50)
    * keep singular values > 1e-5 (ops.py:68-69)
    * D_c = S^-1/2 with no eps (ops.py:72), D_s = S^1/2 (ops.py:76)
    * blend alpha*fcs + (1-alpha)*(fc + mc) : content mean restored (ops.py:83)
    Computed in float32 like the TF graph.
    """
    f = dtype
    fc_full, cshape = _flatten_chw(np.asarray(content, f))
    fs_full, _ = _flatten_chw(np.asarray(style, f))
    c = fc_full.shape[0]
    nc = fc_full.shape[1]
    ns = fs_full.shape[1]
    eye = np.eye(c, dtype=f)

    mc = fc_full.mean(axis=1, keepdims=True)
    fc = fc_full - mc
    cov_c = np.dot(fc, fc.T) / f(nc - 1.0) + eye * f(eps)
    ms = fs_full.mean(axis=1, keepdims=True)
    fs = fs_full - ms
    cov_s = np.dot(fs, fs.T) / f(ns - 1.0) + eye * f(eps)

    uc, sc, _ = np.linalg.svd(cov_c)
    us, ss, _ = np.linalg.svd(cov_s)
    kc = int((sc > 1e-5).sum()) if keep is None else int(keep[0])
    ks = int((ss > 1e-5).sum()) if keep is None else int(keep[1])

    dc = np.diag(sc[:kc] ** f(-0.5))
    whitened = uc[:, :kc].dot(dc).dot(uc[:, :kc].T).dot(fc)
    ds = np.diag(ss[:ks] ** f(0.5))
    colored = us[:, :ks].dot(ds).dot(us[:, :ks].T).dot(whitened) + ms

    blended = f(alpha) * colored + f(1 - alpha) * (fc + mc)
    return np.float32(_unflatten(blended, cshape))


def adain(content_features, style_features, alpha, epsilon=1e-5):
    """AdaIN, semantics of ops.py:282-294.

    tf.nn.moments over H,W gives the POPULATION variance; batch_normalization is
    (x - mean) * rsqrt(var + eps) * scale + offset with scale = sqrt(style var)
    (no eps on the style side) and offset = style mean; then alpha-blend with
    the un-normalised content features.
    """
    x = np.asarray(content_features, np.float32)
    s = np.asarray(style_features, np.float32)
    if x.ndim == 3:
        x = x[None]
    if s.ndim == 3:
        s = s[None]
    # float64 accumulators, float32 results: NumPy reduces a float32 array over its LEADING axes by plain sequential
    # addition, which loses three digits over the 10^6 pixels of a 1024x1024 relu1_1 map (measured 1.4e-3 on the
    # output) -- an artefact of this restatement, not of tf.nn.moments, whose reductions are tree-shaped
    mu_s = s.mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(np.float32)
    var_s = s.var(axis=(1, 2), keepdims=True, dtype=np.float64).astype(np.float32)
    mu_c = x.mean(axis=(1, 2), keepdims=True, dtype=np.float64).astype(np.float32)
    var_c = x.var(axis=(1, 2), keepdims=True, dtype=np.float64).astype(np.float32)
    inv = 1.0 / np.sqrt(var_c + np.float32(epsilon))
    y = (x - mu_c) * inv * np.sqrt(var_s) + mu_s
    return np.float32(np.float32(alpha) * y + np.float32(1 - alpha) * x)


def mat_sqrt_numpy(x):
    """Matrix square root through the SVD, coral.py:8-11."""
    u, d, vt = np.linalg.svd(x)
    # the reference multiplies by V.T where V is numpy's third return value
    # (already V^H), i.e. it uses vt.T -- reproduced literally (coral.py:10).
    return u.dot(np.diag(np.sqrt(d))).dot(vt.T)


def coral_numpy(source, target):
    """CORAL colour alignment of `source` to `target`, coral.py:13-39.

    Per-channel standardise (population std), cov = X X^T + I (NOT divided by
    N), transfer = sqrt(Ct) . inv(sqrt(Cs)) . Xs, de-normalise with the target
    statistics.  dtype follows the inputs (float64 from `img/255.`).
    """
    c = source.shape[-1]
    src = np.moveaxis(source, -1, 0).reshape(c, -1)
    tgt = np.moveaxis(target, -1, 0).reshape(c, -1)

    src_mean = src.mean(axis=1, keepdims=True)
    src_std = src.std(axis=1, keepdims=True)
    src_n = (src - src_mean) / src_std
    tgt_mean = tgt.mean(axis=1, keepdims=True)
    tgt_std = tgt.std(axis=1, keepdims=True)
    tgt_n = (tgt - tgt_mean) / tgt_std

    cov_s = src_n.dot(src_n.T) + np.eye(c)
    cov_t = tgt_n.dot(tgt_n.T) + np.eye(c)

    xfer = mat_sqrt_numpy(cov_t).dot(np.linalg.inv(mat_sqrt_numpy(cov_s))).dot(src_n)
    out = xfer * tgt_std + tgt_mean
    h, w = source.shape[0], source.shape[1]
    return np.moveaxis(out.reshape(c, h, w), 0, -1)


def preserve_colors_np(style_rgb, content_rgb):
    """utils.py:87-90: CORAL on [0,1] images, clip, truncate to uint8."""
    coraled = coral_numpy(style_rgb / 255., content_rgb / 255.)
    return np.uint8(np.clip(coraled, 0, 1) * 255.)


def style_swap(content, style, patch_size, stride, return_margins=False):
    """Patch swap, semantics of ops.py:220-278 (batch dim dropped: HxWxC in, HxWxC out).

    * every patch_size^2 x C patch of `style` at `stride` (VALID) is a filter;
    * the filters are L2-normalised with tf.nn.l2_normalize(..., dim=3) on the [p,p,C,P] tensor, i.e.
      along the PATCH axis -- each filter element is divided by the norm of that element over all
      patches (ops.py:233; epsilon 1e-12 under the square root) -- reproduced literally;
    * correlation = VALID conv of `content` with the normalised filters at `stride`; argmax over patches
      (first index on ties); the UN-normalised winning patch is pasted back (conv2d_transpose of the
      one-hot map, ops.py:255-259) and overlaps are averaged by the coverage count (ops.py:262-276).
    """
    c = np.asarray(content, np.float32)
    s = np.asarray(style, np.float32)
    p, st = int(patch_size), int(stride)
    hs, ws, ch = s.shape
    rows, cols = (hs - p) // st + 1, (ws - p) // st + 1
    patches = np.empty((rows * cols, p, p, ch), np.float32)
    for r in range(rows):
        for q in range(cols):
            patches[r * cols + q] = s[r * st:r * st + p, q * st:q * st + p, :]
    sq = np.sum(patches.astype(np.float32) ** 2, axis=0, dtype=np.float32)
    normed = patches * (1.0 / np.sqrt(np.maximum(sq, np.float32(1e-12))))[None]
    hc, wc, _ = c.shape
    ho, wo = (hc - p) // st + 1, (wc - p) // st + 1
    cols_c = np.empty((ho * wo, p * p * ch), np.float32)
    for y in range(ho):
        for x in range(wo):
            cols_c[y * wo + x] = c[y * st:y * st + p, x * st:x * st + p, :].reshape(-1)
    enc = cols_c @ normed.reshape(rows * cols, -1).T
    arg = np.argmax(enc, axis=1)
    if return_margins:
        # test infrastructure: how decided every match is -- (best - second best) / |best| per content position [ho][wo]
        top2 = np.sort(np.float64(enc), axis=1)[:, -2:]
        margins = ((top2[:, 1] - top2[:, 0]) / np.maximum(np.abs(top2[:, 1]), 1e-30)).reshape(ho, wo)
    hd, wd = (ho - 1) * st + p, (wo - 1) * st + p
    dec = np.zeros((hd, wd, ch), np.float32)
    cnt = np.zeros((hd, wd, 1), np.float32)
    for y in range(ho):
        for x in range(wo):
            dec[y * st:y * st + p, x * st:x * st + p, :] += patches[arg[y * wo + x]]
            cnt[y * st:y * st + p, x * st:x * st + p, :] += 1
    return (dec / cnt, margins) if return_margins else dec / cnt


def wct_style_swap(content, style, alpha, patch_size=3, stride=1, eps=1e-8, return_margins=False):
    """ops.py:145-218: whiten content and style (S^-1/2, no eps in the gains; eps*I on the covariances),
    style_swap on the whitened maps, colour with the style's S^1/2, add the style mean, blend with the
    un-centred content (ops.py:210)."""
    fc_full, cshape = _flatten_chw(np.asarray(content, np.float32))
    fs_full, sshape = _flatten_chw(np.asarray(style, np.float32))
    c = fc_full.shape[0]
    eye = np.eye(c, dtype=np.float32)
    mc = fc_full.mean(axis=1, keepdims=True)
    fc = fc_full - mc
    cov_c = np.dot(fc, fc.T) / np.float32(fc.shape[1] - 1.0) + eye * np.float32(eps)
    ms = fs_full.mean(axis=1, keepdims=True)
    fs = fs_full - ms
    cov_s = np.dot(fs, fs.T) / np.float32(fs.shape[1] - 1.0) + eye * np.float32(eps)
    uc, sc, _ = np.linalg.svd(cov_c)
    us, ss, _ = np.linalg.svd(cov_s)
    kc = int((sc > 1e-5).sum())
    ks = int((ss > 1e-5).sum())
    wc_mat = uc[:, :kc].dot(np.diag(sc[:kc] ** np.float32(-0.5))).dot(uc[:, :kc].T)
    ws_mat = us[:, :ks].dot(np.diag(ss[:ks] ** np.float32(-0.5))).dot(us[:, :ks].T)
    whiten_c = _unflatten(wc_mat.dot(fc), cshape)[0]
    whiten_s = _unflatten(ws_mat.dot(fs), sshape)[0]
    swapped = style_swap(whiten_c, whiten_s, patch_size, stride, return_margins)
    if return_margins:
        swapped, margins = swapped
    hcw = cshape[0] * cshape[1]
    if swapped.shape[0] * swapped.shape[1] != hcw:
        raise ValueError('style-swap output %s does not match the content map %s: pre-size the content '
                         '(utils.swap_filter_fit, wct.py:84-90)' % (swapped.shape, cshape))
    ssf = swapped.reshape(hcw, c).T
    col = us[:, :ks].dot(np.diag(ss[:ks] ** np.float32(0.5))).dot(us[:, :ks].T)
    fcs = col.dot(ssf) + ms
    blended = np.float32(alpha) * fcs + np.float32(1 - alpha) * (fc + mc)
    out = np.float32(_unflatten(blended, cshape))
    return (out, margins) if return_margins else out
