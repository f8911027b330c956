"""Pin the oracle: against outputs of the reference's own functions
(tests/golden, made by oracle/make_golden.py) and against an independent
torch-CPU implementation for the ops the reference delegates to TF/Keras."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import ROOT, GOLDEN, rel_err, max_rel, check_against_size_digest
from wct_tf_amd.weights import (synthetic_weights, synthetic_features, synthetic_image,
                                decoder_plan)


def _cases(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    names = sorted({k.split('/')[0] for k in z.files})
    return z, names


def test_wct_np_matches_reference_outputs():
    z, names = _cases('wct_np_reference.npz')
    assert len(names) >= 5
    for n in names:
        alpha = float(z[n + '/alpha'])
        args = () if alpha < 0 else (alpha,)
        got = oracle.wct_np(z[n + '/content'], z[n + '/style'], *args)
        ref = z[n + '/out']
        assert got.dtype == np.float32 and got.shape == ref.shape
        # same LAPACK, same op order: should agree to fp32 round-off
        assert rel_err(got, ref) < 1e-5, n
        assert max_rel(got, ref) < 1e-4, n


def test_wct_np_matches_reference_on_defective_and_real_image_features():
    """Dead channels, duplicated channels, 80 %-sparse maps, and features of the reference's sample photo
    (tests/golden/gilbert_96.npz through the synthetic-weight encoder; relu4_1: 144 pixels < 512 channels)."""
    z, names = _cases('wct_np_hard.npz')
    assert len(names) == 5
    for n in names:
        got = oracle.wct_np(z[n + '/content'], z[n + '/style'], float(z[n + '/alpha']))
        ref = z[n + '/out']
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 1e-5 and max_rel(got, ref) < 1e-4, n


def test_wct_np_matches_reference_at_config_sizes():
    """The five WCT shapes of a 512x512 frame -- (512,1024) (512,4096) (256,16384) (128,65536) (64,262144) -- run
    through the reference's own wct_np (oracle/make_golden.py); the fixture holds a digest of each output and the
    inputs are rebuilt from their seeds (in_probe pins the rebuild)."""
    from oracle.make_golden import SIZE_CASES, size_case_inputs, in_probe
    z = np.load(os.path.join(GOLDEN, 'wct_np_sizes.npz'))
    for case in SIZE_CASES:
        fc, fs = size_case_inputs(case)
        assert np.allclose(np.stack([in_probe(fc), in_probe(fs)]), z[case[0] + '/in_probe'], rtol=1e-6)
        out = oracle.wct_np(fc, fs, case[4])
        print(case[0], check_against_size_digest(z, case, out, 1e-5))


def test_wct_np_matches_reference_on_hard_512_channel_spectra():
    """512-channel covariances graded over 5 decades at N = 4096, and N = 256 < C (a 6-decade grading, and
    the relu5_1 shape of a 256x256 input) -- the reference's own wct_np outputs (digests)."""
    from oracle.make_golden import HARD512_CASES, hard512_inputs, in_probe
    z = np.load(os.path.join(GOLDEN, 'wct_np_hard512.npz'))
    for case in HARD512_CASES:
        fc, fs = hard512_inputs(case)
        assert np.allclose(np.stack([in_probe(fc), in_probe(fs)]), z[case[0] + '/in_probe'], rtol=1e-6)
        out = oracle.wct_np(fc, fs, case[4])
        print(case[0], z[case[0] + '/content_eig_max_min_kept'], check_against_size_digest(z, case[:7], out, 1e-5))


def test_wct_np_matches_reference_on_a_512_channel_spectrum_through_the_cutoff():
    """tests/golden/wct_np_cross512.npz: the reference's own wct_np on a 512-channel covariance whose eigenvalues run
    THROUGH the 1e-5 cut-off, ~125 of them within a decade of it (oracle/make_golden.py CROSS512_CASE).  The restatement
    reproduces the reference's output -- including its keep/drop decisions on the borderline modes, which are inside the
    fp32 noise of the SVD -- and the recorded kept counts are the ones its own arithmetic arrives at."""
    from oracle.make_golden import CROSS512_CASE, cross512_inputs, in_probe
    from conftest import check_against_size_digest
    z = np.load(os.path.join(GOLDEN, 'wct_np_cross512.npz'))
    name, c, h, w, alpha = CROSS512_CASE[:5]
    fc, fs = cross512_inputs()
    assert np.allclose(np.stack([in_probe(fc), in_probe(fs)]), z[name + '/in_probe'], rtol=1e-6)
    assert min(z[name + '/within_a_decade_of_cutoff']) >= 20
    out = oracle.wct_np(fc, fs, alpha)
    check_against_size_digest(z, CROSS512_CASE, out, 1e-4)
    kept = tuple(int(k) for k in z[name + '/kept_reference'])
    assert rel_err(oracle.wct_np(fc, fs, alpha, keep=kept), out) < 1e-6           # the default run IS the recorded counts
    # what a borderline mode is worth on this input: 1.4e-4 (content side) / 3.8e-4 (style side) of the output per mode --
    # three style modes off is most of the 1e-3 budget, ten are beyond it
    assert rel_err(oracle.wct_np(fc, fs, alpha, keep=(kept[0], kept[1] + 3)), out) > 4e-4
    assert rel_err(oracle.wct_np(fc, fs, alpha, keep=(kept[0], kept[1] - 10)), out) > 1e-3


def test_wct_tf_is_pinned_by_the_reference_wct_np_without_eps():
    """ops.py:24-90 cannot run (TensorFlow), but ops.py:92-140 can: wct_np(c, s, alpha, eps=0) + (1 - alpha) mc is
    wct_tf up to the 1e-8 that wct_tf adds to the covariance diagonals (ops.py:45,50), i.e. up to 0.5e-8 / lambda_min
    relative in the gains (fixture: oracle/make_golden.py WCT_TF_CASES, run on the reference's own code)."""
    z, names = _cases('wct_tf_reference.npz')
    assert len(names) == 3
    for n in names:
        got = oracle.wct_tf(z[n + '/content'], z[n + '/style'], float(z[n + '/alpha']))
        ref = z[n + '/out']
        bound = 2e-5 + 1e-8 / float(z[n + '/lam_min'].min())
        print(n, 'rel %.2e (bound %.2e)' % (rel_err(got, ref), bound))
        assert got.shape == ref.shape and rel_err(got, ref) < bound and max_rel(got, ref) < 10 * bound, n


def test_contractive_weight_set_is_well_conditioned():
    """The end-to-end GPU test (tests/test_gpu_pipeline.py) runs the five-level chain on oracle.contractive weights;
    this is why it can: one flipped input bit moves the oracle's own output by (almost) nothing, and the oracle with
    this path's fp16 storage stays within a few LSB of the fp32 oracle -- against ~30 LSB for both on the He-normal
    weights (test_five_level_chain_is_chaotic_on_random_weights)."""
    from oracle.contractive import contractive_weights
    w = contractive_weights(7)
    c = synthetic_image(1000, 96, 96)
    s = synthetic_image(2000, 96, 96)
    targets = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
    a = oracle.stylize(c, s, w, targets, alpha=0.8)
    c2 = c.copy()
    c2[48, 48, 0] ^= 1
    b = oracle.stylize(c2, s, w, targets, alpha=0.8)
    a16 = oracle.stylize(c, s, w, targets, alpha=0.8, fp16_storage=True)
    d, d16 = np.abs(a.astype(int) - b.astype(int)), np.abs(a.astype(int) - a16.astype(int))
    print('one-bit flip: mean %.4f max %d LSB; fp16 storage: mean %.3f max %d LSB; image std %.1f' % (d.mean(), d.max(), d16.mean(), d16.max(), a.std()))
    assert d.mean() < 0.1 and d.max() <= 2 and d16.mean() < 3 and a.std() > 15


def test_float_images_are_preprocessed_like_the_reference():
    """wct.py:60-64: predict() divides whatever array it is given by 255 -- a FLOAT image is not rounded to integer
    levels.  The host side of the float path (Context.stylize) must hand over exactly float32(image / 255.)."""
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 255, (5, 7, 3))
    want = np.float32(oracle.preprocess(img))
    assert np.array_equal(np.ascontiguousarray(np.asarray(img / 255.), np.float32), want)
    assert not np.array_equal(np.float32(np.uint8(img) / 255.), want)


def test_coral_matches_reference_outputs():
    z, names = _cases('coral_reference.npz')
    for n in names:
        src, tgt = z[n + '/source'], z[n + '/target']
        got = oracle.coral_numpy(src / 255., tgt / 255.)
        assert rel_err(got, z[n + '/coral']) < 1e-10
        assert np.array_equal(oracle.preserve_colors_np(src, tgt), z[n + '/preserve'])


def test_wct_properties_alpha1_matches_style_statistics():
    # alpha=1, full rank: output mean = style mean, covariance ~= style covariance
    fc = synthetic_features(1, 32, 24, 24, 2.0)
    fs = synthetic_features(2, 32, 20, 28, 2.0)
    for fn in (lambda: oracle.wct_np(fc, fs, 1.0), lambda: oracle.wct_tf(fc, fs, 1.0)):
        out = fn().reshape(-1, 32).astype(np.float64)
        s = fs.reshape(-1, 32).astype(np.float64)
        assert np.allclose(out.mean(0), s.mean(0), atol=1e-3)
        cov_o = np.cov(out.T)
        cov_s = np.cov(s.T)
        assert rel_err(cov_o, cov_s) < 2e-2


def test_wct_tf_vs_np_semantics_differ_only_as_documented():
    fc = synthetic_features(3, 32, 16, 16, 2.0)
    fs = synthetic_features(4, 32, 16, 16, 2.0)
    a = 0.7
    tf_ = oracle.wct_tf(fc, fs, a)
    np_ = oracle.wct_np(fc, fs, a)
    mc = fc.reshape(-1, 32).mean(0)
    # np mode drops (1-alpha)*mc (ops.py:133 vs ops.py:83); eps differences are tiny here
    assert rel_err(tf_ - (1 - a) * mc, np_) < 5e-3


def test_adain_statistics():
    fc = synthetic_features(5, 16, 12, 12, 1.0)
    fs = synthetic_features(6, 16, 10, 14, 1.0)
    out = oracle.adain(fc, fs, 1.0)[0].reshape(-1, 16)
    s = fs.reshape(-1, 16)
    assert np.allclose(out.mean(0), s.mean(0), atol=1e-4)
    assert np.allclose(out.std(0), s.std(0), rtol=2e-2, atol=1e-3)


def _torch_conv(x, w, b, relu):
    t = torch.from_numpy(x.transpose(2, 0, 1)[None])
    t = F.pad(t, (1, 1, 1, 1), mode='reflect')
    wt = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))
    y = F.conv2d(t, wt, torch.from_numpy(b))
    if relu:
        y = F.relu(y)
    return y[0].permute(1, 2, 0).numpy()


def test_conv_pool_upsample_match_torch():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((13, 10, 8)).astype(np.float32)
    w = rng.standard_normal((3, 3, 8, 16)).astype(np.float32)
    b = rng.standard_normal(16).astype(np.float32)
    for relu in (True, False):
        assert rel_err(oracle.conv3x3_reflect(x, w, b, relu), _torch_conv(x, w, b, relu)) < 1e-5
    t = torch.from_numpy(x.transpose(2, 0, 1)[None])
    p = F.max_pool2d(t, 2, 2, ceil_mode=True)[0].permute(1, 2, 0).numpy()
    assert np.array_equal(oracle.maxpool2x2_same(x), p)
    u = F.interpolate(t, scale_factor=2, mode='nearest')[0].permute(1, 2, 0).numpy()
    assert np.array_equal(oracle.upsample2x_nearest(x), u)


def test_decoder_plan_matches_oracle_and_survey_counts():
    counts = {'relu5_1': 13, 'relu4_1': 9, 'relu3_1': 5, 'relu2_1': 3, 'relu1_1': 2}
    for relu, n in counts.items():
        assert decoder_plan(relu) == oracle.decoder_layers(relu)
        assert sum(1 for p in decoder_plan(relu) if p[0] == 'C') == n


def test_encoder_decoder_match_torch_small():
    w = synthetic_weights(relu_targets=['relu3_1'])
    img = synthetic_image(11, 36, 28)          # not a multiple of 4: exercises ceil pooling
    feats = oracle.encode(np.float32(img / 255.), w, ['relu3_1', 'relu1_1'])
    assert feats['relu1_1'].shape == (36, 28, 64)
    assert feats['relu3_1'].shape == (9, 7, 256)
    # independent torch re-run of the same chain
    x = np.float32(img / 255.) @ w['encoder']['preprocess'][0].reshape(3, 3) + w['encoder']['preprocess'][1]
    for kind, *rest in oracle.ENCODER_LAYERS:
        if kind == 'C':
            name = rest[0]
            x = _torch_conv(x.astype(np.float32), *w['encoder'][name], True)
            if name == 'conv3_1':
                break
        else:
            t = torch.from_numpy(x.transpose(2, 0, 1)[None])
            x = F.max_pool2d(t, 2, 2, ceil_mode=True)[0].permute(1, 2, 0).numpy()
    assert rel_err(feats['relu3_1'], x) < 1e-4
    dec = oracle.decode(feats['relu3_1'], w, 'relu3_1')
    assert dec.shape == (36, 28, 3)


def test_stylize_pipeline_shapes_and_chaining():
    targets = ['relu2_1', 'relu1_1']
    w = synthetic_weights(relu_targets=targets)
    c = synthetic_image(1000, 32, 48)
    s = synthetic_image(2000, 40, 24)
    out, levels = oracle.stylize(c, s, w, targets, alpha=0.8, return_levels=True)
    assert out.dtype == np.uint8 and out.shape == (32, 48, 3)
    assert levels[0][0].shape == (16, 24, 128) and levels[0][1].shape == (20, 12, 128)
    out_adain = oracle.stylize(c, s, w, targets, alpha=0.8, adain=True)
    assert out_adain.shape == out.shape and not np.array_equal(out, out_adain)


def test_five_level_chain_is_chaotic_on_random_weights():
    """Conditioning of the end-to-end problem with random (untrained) weights: flipping one
    bit of one content pixel moves the oracle's OWN 5-level output by many LSB.  This is why
    the GPU pipeline tests teacher-force each level instead of comparing the final image."""
    w = synthetic_weights(42)
    c = synthetic_image(1000, 64, 64)
    s = synthetic_image(2000, 64, 64)
    targets = ['relu5_1', 'relu4_1', 'relu3_1', 'relu2_1', 'relu1_1']
    a = oracle.stylize(c, s, w, targets, alpha=0.8)
    c2 = c.copy()
    c2[32, 32, 0] ^= 1
    b = oracle.stylize(c2, s, w, targets, alpha=0.8)
    d = np.abs(a.astype(int) - b.astype(int))
    assert d.mean() > 2.0, d.mean()


def test_style_swap_matches_torch():
    """style_swap restatement vs an independent torch-CPU build of the same graph (conv2d with the
    patch filters, argmax, one-hot, conv_transpose2d, coverage count) -- the reference runs this in TF."""
    rng = np.random.default_rng(2)
    for (hc, wc, hs, ws, ch, p, st) in [(9, 8, 7, 10, 6, 3, 1), (11, 11, 9, 9, 4, 3, 2), (6, 6, 6, 6, 5, 1, 1)]:
        c = rng.standard_normal((hc, wc, ch)).astype(np.float32)
        s = rng.standard_normal((hs, ws, ch)).astype(np.float32)
        got = oracle.style_swap(c, s, p, st)
        ct = torch.from_numpy(c.transpose(2, 0, 1)[None])
        stt = torch.from_numpy(s.transpose(2, 0, 1)[None])
        patches = F.unfold(stt, p, stride=st)[0].T.reshape(-1, ch, p, p)           # [P, C, p, p]
        norm = patches / torch.sqrt(torch.clamp((patches ** 2).sum(0, keepdim=True), min=1e-12))
        enc = F.conv2d(ct, norm, stride=st)
        oh = F.one_hot(enc.argmax(1), enc.shape[1]).permute(0, 3, 1, 2).float()
        dec = F.conv_transpose2d(oh, patches, stride=st)
        cnt = F.conv_transpose2d(oh.sum(1, keepdim=True), torch.ones(1, 1, p, p), stride=st)
        want = (dec / cnt)[0].permute(1, 2, 0).numpy()
        assert got.shape == want.shape
        assert rel_err(got, want) < 1e-5, (hc, wc, p, st)


def test_wct_style_swap_properties():
    fc = synthetic_features(7, 16, 10, 10, 1.0)
    fs = synthetic_features(8, 16, 9, 12, 1.0)
    out = oracle.wct_style_swap(fc, fs, 0.6)
    assert out.shape == fc.shape and out.dtype == np.float32
    assert rel_err(oracle.wct_style_swap(fc, fs, 0.0), fc) < 1e-6        # alpha = 0 returns the content
    # 1x1 patches, stride 1: every whitened content vector is replaced by a whitened style vector, so the
    # coloured result at alpha = 1 consists of actual style feature vectors
    out1 = oracle.wct_style_swap(fc, fs, 1.0, patch_size=1)[0].reshape(-1, 16)
    s_flat = fs[0].reshape(-1, 16)
    d = np.abs(out1[:, None, :] - s_flat[None, :, :]).max(-1).min(-1)
    assert d.max() < 2e-3 * np.abs(s_flat).max()


def test_refresh_model_tracked_rotated_matrix_is_the_error_recomputed_one_is_not():
    """The experiment behind the eigen-stage's refresh (csrc/wct.hip refresh_needed), kept as a regression test of
    the ARGUMENT: a NumPy model of the solver (cyclic Jacobi in float32, eigenvectors rounded to 22 bits like the split-fp16
    products, first-order completion of the spectral functions) on a rank-deficient covariance (C = 96, N = 4) whose rounding-noise
    eigenvalues the reference's absolute cut-off keeps.  With the TRACKED rotated matrix the transform is 3e-3 from the nearest
    exact outcome -- 30x the reference's own float32 error; with the rotated matrix RECOMPUTED from the eigenvectors and the
    untouched covariance it is at 1e-6, whatever the eigenvectors' precision and whatever the sweeps left behind."""
    import runpy
    mod = runpy.run_path(os.path.join(ROOT, 'tools', 'probe', 'diag_nllc.py'))
    r = mod['run']()
    assert 5e-5 < r['reference'] < 2e-4
    for name in ('V fp32', 'V 22-bit', 'V 22-bit tol 1e-6'):
        assert r[(name, 'refreshed')] < 1e-5 < r['reference'], (name, r)
    assert r[('V 22-bit', 'tracked')] > 1e-3 and r[('V 22-bit tol 1e-6', 'tracked')] > 1e-3, r     # more sweeps do not help
    assert r[('V 22-bit', 'no completion')] > 1e-3, r                                             # nor does dropping the completion


def test_winograd_restatement_is_the_same_convolution():
    """oracle.conv3x3_reflect_wino_f16 (the roundings of csrc/conv_wino.hip) on operands for which every rounding is exact -- small
    integers, even filter taps so that G g has no halves -- IS conv3x3_reflect, to the bit: the transform matrices, the pair-row
    bookkeeping (odd heights included) and the output transform are right.  On real-valued operands it stays within the
    per-layer gate of the fp32 convolution (1.5e-3; measured ~3.7e-4)."""
    rng = np.random.default_rng(5)
    for h, w in [(6, 5), (7, 9), (2, 2)]:
        x = rng.integers(-4, 5, (h, w, 8)).astype(np.float32)
        g = (2 * rng.integers(-3, 4, (3, 3, 8, 16))).astype(np.float32)
        b = rng.integers(-2, 3, 16).astype(np.float32)
        for relu in (True, False):
            assert np.array_equal(oracle.conv3x3_reflect_wino_f16(x, g, b, relu), oracle.conv3x3_reflect(x, g, b, relu))
            assert np.array_equal(oracle.conv3x3_reflect_wino_f16(x, g, b, relu, acc=np.float32), oracle.conv3x3_reflect(x, g, b, relu))
    x = np.maximum(rng.standard_normal((12, 10, 64)), 0).astype(np.float32)
    g = (rng.standard_normal((3, 3, 64, 32)) * np.sqrt(2.0 / (9 * 64))).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32) * 0.1
    ref = oracle.conv3x3_reflect(np.float32(np.float16(x)), g, b)
    e = np.linalg.norm(oracle.conv3x3_reflect_wino_f16(x, g, b) - ref) / np.linalg.norm(ref)
    assert e < 1.5e-3, e
