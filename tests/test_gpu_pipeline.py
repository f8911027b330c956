"""GPU parity tests, layer and pipeline level (encoder, decoders, WCT.predict)."""
import os
import numpy as np
import pytest

import oracle
from conftest import rel_err, max_rel
from wct_tf_amd.weights import synthetic_weights, synthetic_image, RELU_TARGETS

pytestmark = pytest.mark.gpu

# fp16 activations through up to 13 stacked convs, against the fp32 oracle: the storage rounding itself is 7.6e-3 at
# relu5_1 of a 512x512 image (oracle fp32 vs oracle with fp16 storage, measured)
ENC_TOL = 1e-2
# Stack tolerances by level (relative L2), ~2.5x the values measured on MI355X at 512x512 (in brackets).  The random
# zero-sum filters amplify a perturbation ~2x per conv (test_oracle.py::test_five_level_chain_is_chaotic_on_random_weights),
# so a deep stack cannot be held to the per-layer figure; the per-layer figure itself (2e-4, every layer at its real size
# on the oracle's own inputs) is test_every_conv_layer_at_full_size_teacher_forced.
#   *_32: against the fp32 oracle;  *_16: against the oracle restated with this path's fp16 storage
ENC_TOL_32 = {1: 1e-5, 2: 1.5e-3, 3: 3e-3, 4: 8e-3, 5: 1.5e-2}      # [4.9e-7, 5.9e-4, 1.1e-3, -, 7.6e-3]
ENC_TOL_16 = {1: 1e-5, 2: 2e-4, 3: 1e-3, 4: 5e-3, 5: 8e-3}          # [4.9e-7, 6.5e-5, 3.8e-4, -, 3.9e-3]
DEC_TOL_32 = {1: 4e-4, 2: 8e-4, 3: 1.5e-3, 4: 5e-3, 5: 8e-3}        # [1.5e-4, 3.0e-4, 5.5e-4, -, 3.3e-3]
DEC_TOL_16 = {1: 2e-5, 2: 8e-5, 3: 4e-4, 4: 2e-3, 5: 4e-3}          # [4.2e-6, 2.3e-5, 1.4e-4, -, 1.7e-3]


@pytest.fixture(scope='module')
def weights():
    return synthetic_weights(seed=42)


@pytest.fixture(scope='module')
def ctx(weights):
    from wct_tf_amd.context import Context
    c = Context(0)
    c.set_weights(weights)
    yield c
    c.close()


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.mark.parametrize('size', [(64, 64), (72, 56), (50, 38)])
def test_encoder_all_levels(ctx, weights, size):
    img = np.float32(synthetic_image(7, *size) / 255.)
    want = oracle.encode(img, weights, RELU_TARGETS)
    for relu in RELU_TARGETS:
        got = ctx.encode(img, relu)
        e = rel_err(got, want[relu])
        print(size, relu, got.shape, 'rel %.2e' % e)
        assert got.shape == want[relu].shape
        assert e < ENC_TOL


@pytest.mark.parametrize('relu,hw', [('relu1_1', (32, 24)), ('relu2_1', (16, 16)), ('relu3_1', (8, 12)),
                                     ('relu4_1', (4, 4)), ('relu5_1', (3, 2))])
def test_decoders(ctx, weights, relu, hw):
    from wct_tf_amd.weights import RELU_CHANNELS
    rng = np.random.default_rng(11)
    feat = np.maximum(rng.standard_normal((hw[0], hw[1], RELU_CHANNELS[relu])), 0).astype(np.float32)
    got = ctx.decode(feat, relu)
    want = oracle.decode(feat, weights, relu)
    e = rel_err(got, want)
    print(relu, got.shape, 'rel %.2e' % e, 'max abs %.3e' % np.abs(got - want).max())
    assert got.shape == want.shape and e < ENC_TOL


def _teacher_forced(ctx, weights, content, style, targets, alpha, mode, adain=False):
    """Run the oracle pipeline; feed each level's oracle inputs to the GPU ops and compare: the encoder on the
    oracle's level input, the transform on the oracle's features, the decoder on the oracle's transformed
    features -- against the fp32 oracle (ENC_TOL) and against the oracle with fp16 storage (ENC_TOL16)."""
    from wct_tf_amd import _lib
    out, levels = oracle.stylize(content, style, weights, targets, alpha=alpha, wct_mode=mode, return_levels=True, adain=adain)
    x_in = np.float32(content / 255.)
    for i, (relu, (fc, fs, t, x)) in enumerate(zip(targets, levels)):
        c = fc.shape[-1]
        if i > 0:
            x_in = np.clip(x_in, 0, 1)
        got_fc = ctx.encode(x_in, relu)
        e_enc = rel_err(got_fc, fc[0] if fc.ndim == 4 else fc)
        e_enc16 = rel_err(got_fc, oracle.encode(x_in, weights, [relu], fp16_storage=True)[relu])
        if adain:
            got_t = ctx.adain(fc.reshape(-1, c), fs.reshape(-1, c), alpha).reshape(t.shape)
        else:
            got_t = ctx.transform(fc.reshape(-1, c), fs.reshape(-1, c), alpha,
                                  _lib.WCT_TF if mode == 'tf' else _lib.WCT_NP).reshape(t.shape)
        e = rel_err(got_t, t)
        got_x = ctx.decode(t, relu)
        e_dec, e_dec16 = rel_err(got_x, x), rel_err(got_x, oracle.decode(t, weights, relu, fp16_storage=True))
        print('%s %s: encoder rel %.2e (fp16-storage oracle %.2e)  transform rel %.2e  decoder rel %.2e (fp16-storage oracle %.2e)'
              % (relu, fc.shape, e_enc, e_enc16, e, e_dec, e_dec16))
        lv = int(relu[4])
        assert e < 1e-3
        assert e_enc < ENC_TOL_32[lv] and e_dec < DEC_TOL_32[lv]
        assert e_enc16 < ENC_TOL_16[lv] and e_dec16 < DEC_TOL_16[lv]
        x_in = x
    return out


def test_pipeline_two_levels_teacher_forced_and_end_to_end(ctx, weights):
    targets = ['relu2_1', 'relu1_1']
    c = synthetic_image(1000, 64, 80)
    s = synthetic_image(2000, 72, 56)
    want = _teacher_forced(ctx, weights, c, s, targets, 0.8, 'tf')
    got = ctx.stylize(c, s, targets, alpha=0.8)
    d = np.abs(got.astype(int) - want.astype(int))
    print('2-level end-to-end: psnr %.1f dB, max LSB %d, mean LSB %.3f' % (psnr(got, want), d.max(), d.mean()))
    assert got.shape == want.shape and psnr(got, want) > 35


def test_pipeline_five_levels_teacher_forced_and_fused_equals_stepwise(ctx, weights):
    """Full relu5_1->relu1_1 chain.  End-to-end equality with the oracle is NOT a usable
    criterion on random weights: the oracle itself moves by ~28 LSB (mean) when ONE bit of
    ONE content pixel flips (tests/test_oracle.py::test_five_level_chain_is_chaotic_on_random_weights),
    so each level is checked on the oracle's own inputs (teacher forcing), and the fused
    wct_stylize call must equal the same GPU ops chained step by step, bit for bit."""
    from wct_tf_amd import _lib
    c = synthetic_image(1000, 128, 128)
    s = synthetic_image(2000, 128, 128)
    for mode in ('tf', 'np'):
        _teacher_forced(ctx, weights, c, s, RELU_TARGETS, 0.8, mode)
        got = ctx.stylize(c, s, RELU_TARGETS, alpha=0.8, wct_mode=mode)
        x = np.float32(c / 255.)
        s01 = np.float32(s / 255.)
        for i, relu in enumerate(RELU_TARGETS):
            if i > 0:
                x = np.clip(x, 0, 1)
            fc, fs = ctx.encode(x, relu), ctx.encode(s01, relu)
            ch = fc.shape[-1]
            t = ctx.transform(fc.reshape(-1, ch), fs.reshape(-1, ch), 0.8,
                              _lib.WCT_TF if mode == 'tf' else _lib.WCT_NP).reshape(fc.shape)
            x = ctx.decode(t, relu)
        step = np.uint8(np.clip(x, 0, 1) * 255)
        assert np.array_equal(got, step), mode


def test_config2_single_level_relu3_512_end_to_end(ctx, weights):
    """BASELINE config 2 as stated: single level relu3_1, 512x512 content and style, alpha 0.8 (model.py:123-176: one
    encoder -> wct -> decoder; covariances 256x256).  One level is well conditioned, so the final uint8 frame is
    compared with the oracle's directly: against the fp32 oracle (the reference's arithmetic) and against the oracle
    restated with this path's fp16 storage."""
    targets = ['relu3_1']
    c = synthetic_image(1000, 512, 512)
    s = synthetic_image(2000, 512, 512)
    _teacher_forced(ctx, weights, c, s, targets, 0.8, 'tf')
    got = ctx.stylize(c, s, targets, alpha=0.8)
    for name, want, min_psnr, max_lsb in (
            ('fp32 oracle', oracle.stylize(c, s, weights, targets, alpha=0.8), 40.0, 12),
            ('fp16-storage oracle', oracle.stylize(c, s, weights, targets, alpha=0.8, fp16_storage=True), 50.0, 3)):
        d = np.abs(got.astype(int) - want.astype(int))
        print('config 2 vs %s: psnr %.1f dB, max LSB %d, mean LSB %.4f, pixels off by > 1 LSB %.5f'
              % (name, psnr(got, want), d.max(), d.mean(), (d > 1).mean()))
        assert got.shape == want.shape == (512, 512, 3)
        assert psnr(got, want) > min_psnr and d.max() <= max_lsb


def test_config1_single_level_relu1_256_alpha1_np_end_to_end(ctx, weights):
    """BASELINE config 1 AS STATED, through the GPU path: single level relu1_1, 256x256 content and style, alpha = 1.0, the
    NumPy `wct()` semantics (ops.py:92-140: eps inside the gains, content mean not restored) -- one conv1_1 encoder pass,
    one 64-channel transform over 65 536 pixels, the one-conv relu1_1 decoder (model.py:123-176, 255-298) and
    wct.py:60-68 at the ends.  The final uint8 frame against oracle.stylize(wct_mode='np') in fp32 (the reference's
    arithmetic) and with this path's fp16 storage; the fused call must also be the chained ops, bit for bit."""
    from wct_tf_amd import _lib
    targets = ['relu1_1']
    c = synthetic_image(1000, 256, 256)
    s = synthetic_image(2000, 256, 256)
    _teacher_forced(ctx, weights, c, s, targets, 1.0, 'np')
    got = ctx.stylize(c, s, targets, alpha=1.0, wct_mode='np')
    fc, fs = ctx.encode(np.float32(c / 255.), 'relu1_1'), ctx.encode(np.float32(s / 255.), 'relu1_1')
    t = ctx.transform(fc.reshape(-1, 64), fs.reshape(-1, 64), 1.0, _lib.WCT_NP).reshape(fc.shape)
    assert np.array_equal(got, np.uint8(np.clip(ctx.decode(t, 'relu1_1'), 0, 1) * 255))
    want_tf = oracle.stylize(c, s, weights, targets, alpha=1.0, wct_mode='tf')
    for name, want, min_psnr, max_lsb in (
            # measured on MI355X (round 5): see profiles/r05_parity_holes.txt
            ('fp32 oracle', oracle.stylize(c, s, weights, targets, alpha=1.0, wct_mode='np'), 45.0, 4),
            ('fp16-storage oracle', oracle.stylize(c, s, weights, targets, alpha=1.0, wct_mode='np', fp16_storage=True), 50.0, 2)):
        d = np.abs(got.astype(int) - want.astype(int))
        print('config 1 vs %s: psnr %.1f dB, max LSB %d, mean LSB %.4f, pixels off by > 1 LSB %.5f, frame std %.1f'
              % (name, psnr(got, want), d.max(), d.mean(), (d > 1).mean(), want.std()))
        assert got.shape == want.shape == (256, 256, 3) and want.std() > 10
        assert psnr(got, want) > min_psnr and d.max() <= max_lsb
    print('config 1: np-mode frame vs the tf-mode oracle %.1f dB (the two semantics differ)' % psnr(got, want_tf))


def test_config3_five_levels_512_end_to_end_on_a_well_conditioned_net():
    """BASELINE config 3 END TO END: full relu5_1 -> relu1_1 chain, 512x512 content and style, alpha 0.8, the final uint8
    frame against oracle.stylize (model.py:78-94: every level encodes clip(previous decoded), wct.py:60-68 at the ends).
    On the He-normal stand-in weights this comparison is meaningless (the map is chaotic: one flipped input bit moves
    the oracle's own frame by ~18 LSB mean at this size); oracle/contractive.py builds the same architecture out of
    near-isometric layers (one flipped bit: 0.007 LSB mean), on which the frame is well defined:
      * against the fp32 oracle -- the reference's arithmetic; this path stores activations in fp16, which alone
        moves the oracle's frame by 1.0 LSB mean / 9 max (oracle fp32 vs oracle fp16-storage, 45.5 dB);
      * against the oracle restated with this path's fp16 storage -- accumulation order and the eigensolver remain."""
    from oracle.contractive import contractive_weights
    from wct_tf_amd.context import Context
    w = contractive_weights(7)
    c = synthetic_image(1000, 512, 512)
    s = synthetic_image(2000, 512, 512)
    cx = Context(0)
    try:
        cx.set_weights(w)
        got = cx.stylize(c, s, RELU_TARGETS, alpha=0.8)
        got_batch = cx.stylize_batch(np.stack([c, synthetic_image(1001, 512, 512)]), np.stack([s, synthetic_image(2001, 512, 512)]),
                                     RELU_TARGETS, alpha=0.8)
    finally:
        cx.close()
    assert np.array_equal(got_batch[0], got)          # the batched entry point (what bench.py times) gives the same frame
    for name, want, min_psnr, max_mean, max_lsb in (
            # measured on MI355X: 45.6 dB / 0.99 LSB mean / 11 LSB max against the fp32 oracle, 44.5 dB / 1.14 / 12 against
            # the fp16-storage restatement (it rounds at slightly different points than the kernels: both sit at the
            # distance the storage precision itself puts between the two oracles)
            ('fp32 oracle', oracle.stylize(c, s, w, RELU_TARGETS, alpha=0.8), 42.0, 1.6, 20),
            ('fp16-storage oracle', oracle.stylize(c, s, w, RELU_TARGETS, alpha=0.8, fp16_storage=True), 42.0, 1.6, 20)):
        d = np.abs(got.astype(int) - want.astype(int))
        print('config 3 end to end vs %s: psnr %.1f dB, max LSB %d, mean LSB %.4f, pixels off by > 1 LSB %.5f, frame std %.1f'
              % (name, psnr(got, want), d.max(), d.mean(), (d > 1).mean(), want.std()))
        assert got.shape == want.shape == (512, 512, 3) and want.std() > 15
        assert psnr(got, want) > min_psnr and d.mean() < max_mean and d.max() <= max_lsb


def _frame_agreement(tag, got, want):
    d = np.abs(got.astype(int) - want.astype(int))
    print('%s: psnr %.1f dB, max LSB %d, mean LSB %.4f, pixels off by > 1 LSB %.5f, frame std %.1f'
          % (tag, psnr(got, want), d.max(), d.mean(), (d > 1).mean(), want.std()))
    return psnr(got, want), d.mean(), d.max()


def test_config3_five_levels_512_end_to_end_np_semantics():
    """The same five-level 512x512 chain with the NumPy `wct()` semantics the north star names (ops.py:92-140: eps inside
    the gains, the content mean handled as wct_np does) instead of the graph's wct_tf -- `wct_mode='np'` through all five
    levels END TO END against oracle.stylize(wct_mode='np') on the contractive net (np mode was
    teacher-forced at 128x128 only)."""
    from oracle.contractive import contractive_weights
    from wct_tf_amd.context import Context
    w = contractive_weights(7)
    c = synthetic_image(1000, 512, 512)
    s = synthetic_image(2000, 512, 512)
    cx = Context(0)
    try:
        cx.set_weights(w)
        got = cx.stylize(c, s, RELU_TARGETS, alpha=0.8, wct_mode='np')
        got_tf = cx.stylize(c, s, RELU_TARGETS, alpha=0.8, wct_mode='tf')
    finally:
        cx.close()
    want = oracle.stylize(c, s, w, RELU_TARGETS, alpha=0.8, wct_mode='np')
    p, mean, mx = _frame_agreement('config 3 end to end, wct_np semantics, vs the fp32 oracle', got, want)
    assert got.shape == want.shape == (512, 512, 3) and want.std() > 15
    assert p > 42.0 and mean < 1.6 and mx <= 20
    # the two semantics are different transforms: the np-mode frame must be the np oracle's, not the tf one's
    assert psnr(got_tf, want) < p - 3


@pytest.mark.parametrize('adain', [False, True], ids=['wct', 'adain'])
def test_config5_chained_end_to_end_1024_content_512_style(adain):
    """BASELINE config 5 CHAINED end to end (stylize.py:85-100): 1024x1024 content, 512x512 style, --keep-colors
    (preserve_colors_np first, coral.py:13-39), then the five levels in the WCT branch and in the --adain branch, final
    uint8 frame against oracle.preserve_colors_np + oracle.stylize on the contractive net (on the He-normal stand-in the
    chained frame is chaotic; see test_config3_...).  VERDICT r3 item 6a: config 5 was teacher-forced only."""
    from oracle.contractive import contractive_weights
    from wct_tf_amd.context import Context
    from wct_tf_amd import ops
    w = contractive_weights(7)
    c = synthetic_image(1005, 1024, 1024)
    s = synthetic_image(2005, 512, 512)
    cx = Context(0)
    try:
        cx.set_weights(w)
        s_cc = ops.preserve_colors_np(s, c, ctx=cx)
        got = cx.stylize(c, s_cc, RELU_TARGETS, alpha=0.8, adain=adain)
    finally:
        cx.close()
    want_cc = oracle.preserve_colors_np(s, c)
    d = np.abs(s_cc.astype(int) - want_cc.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3                     # CORAL: +-1 LSB on a handful of pixels
    want = oracle.stylize(c, want_cc, w, RELU_TARGETS, alpha=0.8, adain=adain)
    p, mean, mx = _frame_agreement('config 5 chained (%s branch) vs the fp32 oracle' % ('adain' if adain else 'wct'), got, want)
    assert got.shape == want.shape == (1024, 1024, 3) and want.std() > 10
    # (round 6: the >= 256-channel layers run on the reduced-FLOP kernel, whose transformed operands are rounded to fp16 once more
    #  than the direct kernel's -- per layer 3.2-3.7e-4 against 2.7-2.9e-4; this frame: 39.3 dB / 2.11 LSB mean in the WCT branch where the
    #  direct-only build measured 41.8 / 1.56, the --adain branch 44.9 / 1.06 where it measured 47.3 / 0.78 -> profiles/r06_wino_layers.txt; asserted with that margin)
    assert p > (43.5 if adain else 38.5) and mean < (1.4 if adain else 2.4) and mx <= 28


def test_pipeline_five_levels_teacher_forced_512(ctx, weights):
    """BASELINE config 3 at its own size (512x512, five levels, alpha 0.8): every level's encoder, transform and
    decoder on the oracle's own level inputs (the chained output itself is chaotic on random weights, see
    test_pipeline_five_levels_teacher_forced_and_fused_equals_stepwise)."""
    c = synthetic_image(1000, 512, 512)
    s = synthetic_image(2000, 512, 512)
    _teacher_forced(ctx, weights, c, s, RELU_TARGETS, 0.8, 'tf')


@pytest.mark.parametrize('size', [(512, 512), (256, 384)])
def test_encoder_decoder_stacks_full_size(ctx, weights, size):
    """Layer stacks at the sizes the metric runs (the tall 32x16 tile path of the 64-channel layers, the 16x16 tile
    path, pools fused into conv epilogues): all five encoder taps and all five decoders against the oracle."""
    img = np.float32(synthetic_image(7, *size) / 255.)
    want = oracle.encode(img, weights, RELU_TARGETS)
    want16 = oracle.encode(img, weights, RELU_TARGETS, fp16_storage=True)
    for relu in RELU_TARGETS:
        got = ctx.encode(img, relu)
        e, e16 = rel_err(got, want[relu]), rel_err(got, want16[relu])
        feat = want[relu]
        dec = ctx.decode(feat, relu)
        d, d16 = rel_err(dec, oracle.decode(feat, weights, relu)), rel_err(dec, oracle.decode(feat, weights, relu, fp16_storage=True))
        print('%s %s: encoder rel %.2e (fp16-storage oracle %.2e); decoder rel %.2e (fp16-storage oracle %.2e)'
              % (size, relu, e, e16, d, d16))
        lv = int(relu[4])
        assert got.shape == want[relu].shape and dec.shape == size + (3,)
        assert e < ENC_TOL_32[lv] and d < DEC_TOL_32[lv] and e16 < ENC_TOL_16[lv] and d16 < DEC_TOL_16[lv]


def test_every_conv_layer_at_full_size_teacher_forced(ctx, weights):
    """Every 3x3 layer of the 512x512 path at its real size -- the 12 encoder convs behind conv1_1 and the 12
    decoder-5 convs ahead of the output conv (all (Cin, Cout, H) combinations of SURVEY 8a's table, both tile
    configurations, the folded upsample) -- each on the ORACLE's input for that layer (fp16-rounded, as stored), so no
    error is inherited: what is measured is one layer's arithmetic.  Tolerance 2e-4 (fp32 accumulation order)."""
    from oracle.net_oracle import ENCODER_LAYERS, decoder_layers
    h16 = lambda a: np.asarray(a, np.float16).astype(np.float32)    # noqa: E731
    img = np.float32(synthetic_image(7, 512, 512) / 255.)
    enc = weights['encoder']
    x = oracle.net_oracle.conv1x1(img, *enc['preprocess'])
    worst = 0.0
    for layer in ENCODER_LAYERS:
        if layer[0] == 'P':
            x = oracle.maxpool2x2_same(x)
            continue
        name = layer[1]
        w, b = enc[name]
        if name != 'conv1_1':
            x = h16(x)
            want = oracle.conv3x3_reflect(x, h16(w), b, relu=True)
            got = ctx.conv3x3(x, w, b, relu=True)
            e = rel_err(got, want)
            worst = max(worst, e)
            print('%s %s -> %d: rel %.2e  max %.2e' % (name, x.shape, w.shape[3], e, max_rel(got, want)))
            assert e < 2e-4 and max_rel(got, want) < 2e-3, name
            x = want
        else:
            x = oracle.conv3x3_reflect(x, w, b, relu=True)
    params = iter(weights['decoder']['relu5_1'])
    up = False
    x = h16(x)
    for kind, cin, cout, relu in decoder_layers('relu5_1'):
        if kind == 'U':
            up = True
            continue
        w, b = next(params)
        if cout == 3:
            break
        xin = oracle.upsample2x_nearest(x) if up else x
        want = oracle.conv3x3_reflect(xin, h16(w), b, relu=relu)
        got = ctx.conv3x3(x, w, b, relu=relu, upsample=up)
        e = rel_err(got, want)
        worst = max(worst, e)
        print('dec5 %s%s -> %d: rel %.2e  max %.2e' % (x.shape, ' x2' if up else '', cout, e, max_rel(got, want)))
        assert e < 2e-4 and max_rel(got, want) < 2e-3
        x, up = h16(want), False
    print('worst layer: %.2e' % worst)


def test_batch32_at_512_equals_single_pairs(ctx, weights):
    """The bench configuration (32 resident pairs of 512x512 per call: batch strides, 32-bit buffer offsets, two
    eigensolver groups): frames of the batched call equal the single-pair call bit for bit."""
    B = 32
    cs = np.stack([synthetic_image(1000 + i, 512, 512) for i in range(B)])
    ss = np.stack([synthetic_image(2000 + i, 512, 512) for i in range(B)])
    dc, ds, do = ctx.dev_alloc(cs.nbytes), ctx.dev_alloc(ss.nbytes), ctx.dev_alloc(cs.nbytes)
    ctx.h2d(dc, cs); ctx.h2d(ds, ss)
    ctx.stylize_batch_dev(dc, 512, 512, ds, 512, 512, B, RELU_TARGETS, 0.8, do)
    ctx.sync()
    outs = np.empty_like(cs)
    ctx.d2h(outs, do)
    for p in (dc, ds, do):
        ctx.dev_free(p)
    for i in (0, 13, 31):
        assert np.array_equal(outs[i], ctx.stylize(cs[i], ss[i], RELU_TARGETS, alpha=0.8)), i
    assert len({outs[i].tobytes() for i in range(B)}) == B


def test_real_image_smoke_gilbert(ctx, weights):
    """SURVEY 8d real-image smoke: crops of the reference's sample photo (samples/gilbert.jpg ->
    tests/golden/gilbert_96.npz, made by oracle/make_golden.py) through three levels, against the oracle; smooth
    natural-image statistics instead of blurred noise."""
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, 'gilbert_96.npz'))
    c, s = z['content'], z['style']
    assert c.shape == s.shape == (96, 96, 3) and c.dtype == np.uint8
    targets = ['relu3_1', 'relu2_1', 'relu1_1']
    want = _teacher_forced(ctx, weights, c, s, targets, 0.8, 'tf')
    got = ctx.stylize(c, s, targets, alpha=0.8)
    d = np.abs(got.astype(int) - want.astype(int))
    print('gilbert 3-level: psnr %.1f dB, max LSB %d, mean LSB %.3f' % (psnr(got, want), d.max(), d.mean()))
    assert psnr(got, want) > 30


def test_pipeline_three_levels_end_to_end(ctx, weights):
    targets = ['relu3_1', 'relu2_1', 'relu1_1']
    c = synthetic_image(1000, 96, 96)
    s = synthetic_image(2000, 96, 96)
    want = oracle.stylize(c, s, weights, targets, alpha=0.8)
    got = ctx.stylize(c, s, targets, alpha=0.8)
    d = np.abs(got.astype(int) - want.astype(int))
    print('3-level end-to-end: psnr %.1f dB, max LSB %d, mean LSB %.3f' % (psnr(got, want), d.max(), d.mean()))
    assert psnr(got, want) > 30


def test_pipeline_adain_and_odd_sizes(ctx, weights):
    targets = ['relu3_1', 'relu1_1']
    c = synthetic_image(1001, 50, 70)           # not multiples of 4: ceil pooling + x2 upsampling grow the frame
    s = synthetic_image(2001, 64, 64)
    want = oracle.stylize(c, s, weights, targets, alpha=0.6, adain=True)
    got = ctx.stylize(c, s, targets, alpha=0.6, adain=True)
    assert got.shape == want.shape == (52, 72, 3)
    print('adain odd: psnr %.1f' % psnr(got, want))
    assert psnr(got, want) > 35


def test_wct_facade_and_batch(ctx, weights):
    from wct_tf_amd import WCT
    targets = ['relu2_1', 'relu1_1']
    model = WCT(checkpoints=None, relu_targets=targets, vgg_path=None, device='/gpu:0', weights=weights)
    c = synthetic_image(1002, 48, 48)
    s = synthetic_image(2002, 48, 48)
    out = model.predict(c, s, alpha=0.8)
    assert out.dtype == np.uint8 and out.shape == (48, 48, 3)
    assert np.array_equal(out, ctx.stylize(c, s, targets, alpha=0.8))
    with pytest.raises(Exception):
        WCT(checkpoints=['/nonexistent'], relu_targets=targets, vgg_path='/nonexistent.npz')
    # batched device-resident entry point == per-pair calls, bit for bit
    B = 3
    cs = np.stack([synthetic_image(1000 + i, 48, 48) for i in range(B)])
    ss = np.stack([synthetic_image(2000 + i, 48, 48) for i in range(B)])
    sess = model.sess
    dc, ds, do = sess.dev_alloc(cs.nbytes), sess.dev_alloc(ss.nbytes), sess.dev_alloc(cs.nbytes)
    sess.h2d(dc, cs)
    sess.h2d(ds, ss)
    sess.stylize_batch_dev(dc, 48, 48, ds, 48, 48, B, targets, 0.8, do)
    sess.sync()
    outs = np.empty_like(cs)
    sess.d2h(outs, do)
    for i in range(B):
        assert np.array_equal(outs[i], model.predict(cs[i], ss[i], alpha=0.8)), i
    for p in (dc, ds, do):
        sess.dev_free(p)


def test_stylize_cli_end_to_end(tmp_path, ctx, weights):
    """python -m wct_tf_amd.stylize with the reference's flags: content x style loop, --keep-colors,
    --passes, --concat, output naming (stylize.py:70-117)."""
    from wct_tf_amd import utils
    from wct_tf_amd.stylize import main
    cdir, sdir, odir = tmp_path / 'c', tmp_path / 's', tmp_path / 'o'
    cdir.mkdir(); sdir.mkdir()
    utils.save_img(str(cdir / 'a.png'), synthetic_image(1, 64, 80))
    utils.save_img(str(cdir / 'b.png'), synthetic_image(2, 64, 64))
    utils.save_img(str(sdir / 'st.png'), synthetic_image(3, 96, 96))
    n = main(['--relu-targets', 'relu2_1', 'relu1_1', '--synthetic-weights', '42', '--content-path', str(cdir),
              '--style-path', str(sdir), '--out-path', str(odir), '--alpha', '0.8', '--style-size', '64',
              '--crop-size', '48', '--keep-colors', '--passes', '2', '--concat'])
    assert n == 2
    out = utils.get_img(str(odir / 'a_st.png'))
    assert out.shape == (64, 64 + 80, 3)
    assert os.path.exists(str(odir / 'b_st.png'))
    # same thing by hand through the op-level API
    from wct_tf_amd import ops
    style = utils.center_crop(utils.resize_to(utils.get_img(str(sdir / 'st.png')), 64), 48)
    content = utils.get_img(str(cdir / 'a.png'))
    style = ops.preserve_colors_np(style, content, ctx=ctx)
    x = ctx.stylize(content, style, ['relu2_1', 'relu1_1'], alpha=0.8)
    x = ctx.stylize(x, style, ['relu2_1', 'relu1_1'], alpha=0.8)
    assert np.array_equal(out[:, 64:], x)


def test_full_size_properties_512(ctx, weights):
    """BASELINE config 3 size (512x512, 5 levels): size-independent properties instead of an oracle run.
    (1) determinism: two runs are bit-identical; (2) batching: the batched device entry point equals the
    single-pair call bit for bit; (3) alpha = 0 makes every transform the identity on the content features,
    so the style image must not matter; (4) per-level WCT at alpha = 1 reproduces the style mean and
    covariance on the GPU's own full-size features (the defining property of the transform)."""
    from wct_tf_amd import _lib
    c = synthetic_image(1000, 512, 512)
    s = synthetic_image(2000, 512, 512)
    s2 = synthetic_image(2001, 512, 512)
    a = ctx.stylize(c, s, RELU_TARGETS, alpha=0.8)
    b = ctx.stylize(c, s, RELU_TARGETS, alpha=0.8)
    assert a.shape == (512, 512, 3) and a.dtype == np.uint8
    assert np.array_equal(a, b)
    cs = np.stack([c, synthetic_image(1001, 512, 512)])
    ss = np.stack([s, s2])
    dc, ds, do = ctx.dev_alloc(cs.nbytes), ctx.dev_alloc(ss.nbytes), ctx.dev_alloc(cs.nbytes)
    ctx.h2d(dc, cs); ctx.h2d(ds, ss)
    ctx.stylize_batch_dev(dc, 512, 512, ds, 512, 512, 2, RELU_TARGETS, 0.8, do)
    ctx.sync()
    outs = np.empty_like(cs)
    ctx.d2h(outs, do)
    for p in (dc, ds, do):
        ctx.dev_free(p)
    assert np.array_equal(outs[0], a)
    z1 = ctx.stylize(c, s, RELU_TARGETS, alpha=0.0)
    z2 = ctx.stylize(c, s2, RELU_TARGETS, alpha=0.0)
    assert np.array_equal(z1, z2)
    s01 = np.float32(s / 255.)
    c01 = np.float32(c / 255.)
    for relu in ('relu5_1', 'relu3_1', 'relu1_1'):
        fc, fs = ctx.encode(c01, relu), ctx.encode(s01, relu)
        ch = fc.shape[-1]
        out, sweeps = ctx.transform(fc.reshape(-1, ch), fs.reshape(-1, ch), 1.0, _lib.WCT_TF, return_sweeps=True)
        fs2 = fs.reshape(-1, ch).astype(np.float64)
        o2 = out.astype(np.float64)
        assert np.abs(o2.mean(0) - fs2.mean(0)).max() < 1e-3 * max(1.0, np.abs(fs2.mean(0)).max())
        cov_o, cov_s = np.cov(o2.T), np.cov(fs2.T)
        err = np.linalg.norm(cov_o - cov_s) / np.linalg.norm(cov_s)
        print(relu, 'sweeps', sweeps, 'cov err %.2e' % err)
        assert err < 5e-3


def test_pipeline_reports_failed_eigensolves_at_sync(ctx, weights):
    """The asynchronous batch entry point cannot return the eigensolver's status itself: the next wct_sync does
    (WCT_STATUS_NOCONV), and the blocking wct_stylize returns it directly; afterwards the context works as before."""
    from wct_tf_amd._lib import WCTNotConverged
    targets = ['relu3_1', 'relu1_1']
    c, s = synthetic_image(1000, 64, 64), synthetic_image(2000, 64, 64)
    good = ctx.stylize(c, s, targets, alpha=0.8)
    os.environ['WCT_JACOBI_MAX_SWEEPS'] = '1'
    try:
        with pytest.raises(WCTNotConverged):
            ctx.stylize(c, s, targets, alpha=0.8)
        cs, ss = np.stack([c, c]), np.stack([s, s])
        dc, ds, do = ctx.dev_alloc(cs.nbytes), ctx.dev_alloc(ss.nbytes), ctx.dev_alloc(cs.nbytes)
        ctx.h2d(dc, cs); ctx.h2d(ds, ss)
        ctx.stylize_batch_dev(dc, 64, 64, ds, 64, 64, 2, targets, 0.8, do)        # enqueues; no status yet
        with pytest.raises(WCTNotConverged):
            ctx.sync()
        ctx.sync()                                                                 # reported once
        for p in (dc, ds, do):
            ctx.dev_free(p)
    finally:
        del os.environ['WCT_JACOBI_MAX_SWEEPS']
    assert np.array_equal(ctx.stylize(c, s, targets, alpha=0.8), good)


def test_stream_handle_orders_foreign_work(ctx, weights):
    """wct_get_stream: a caller's own stream can be ordered behind wct_stylize_batch_dev with events alone (what the
    multi-GPU bench does with the RCCL gather) -- no host sync between the library's kernels and the foreign copy."""
    import ctypes as C
    import torch
    targets = ['relu3_1', 'relu1_1']
    B = 4
    cs = np.stack([synthetic_image(1100 + i, 96, 96) for i in range(B)])
    ss = np.stack([synthetic_image(2100 + i, 96, 96) for i in range(B)])
    dev = torch.device('cuda', 0)
    dc, ds = torch.from_numpy(cs).to(dev), torch.from_numpy(ss).to(dev)
    out = torch.zeros_like(dc)
    torch.cuda.synchronize()
    lib_stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
    side = torch.cuda.Stream(dev)
    ev = torch.cuda.Event()
    ctx.stylize_batch_dev(C.c_void_p(dc.data_ptr()), 96, 96, C.c_void_p(ds.data_ptr()), 96, 96, B, targets, 0.8, C.c_void_p(out.data_ptr()))
    ev.record(lib_stream)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        copy = out.clone()
    side.synchronize()
    ctx.sync()
    for i in range(B):
        assert np.array_equal(copy[i].cpu().numpy(), ctx.stylize(cs[i], ss[i], targets, alpha=0.8)), i


def test_bench_cli_small(tmp_path):
    """bench.py end to end on a small configuration: the JSON contract (metric, value, roofline, eigensolver, latency),
    the strong-scaling mode, and the event protocol of the overlapped gather (forced on the single rank)."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, WCT_BENCH_FORCE_OVERLAP='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--size', '64', '--global-batch', '6', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['scaling'] == 'strong' and line['config']['global_batch'] == 6 and line['n_gpus'] == 1
    assert line['value'] > 0 and line['roofline']['bound'] == 'mfma' and 0 < line['roofline']['frac'] < 1
    assert line['eigensolver']['ms_per_step'] > 0 and line['latency_fps'] > 0
    assert abs(sum(line['breakdown_ms_per_step'].values()) - line['ms_per_step']) < 0.5 * line['ms_per_step']


def test_bench_gpus2_self_launches_two_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher starts two ranks itself (VERDICT r3 missing #1).  Dry run on the one
    GPU of this box: both ranks share it, the exchange goes through gloo; the line says n_gpus = 2 and carries the `strong`
    sub-record (64 pairs per step over the ranks = BASELINE configs[3])."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(WCT_BENCH_BACKEND='gloo', WCT_BENCH_SHARE_GPU='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '64', '--batch', '3', '--steps', '2',
                          '--warmup', '1', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1                                            # rank 0 alone prints
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['config']['global_batch'] == 6
    assert line['strong']['global_batch'] == 64 and line['strong']['pairs_per_gpu_per_step'] == 32 and line['strong']['value'] > 0
    # without the dry-run switch, more ranks than GPUs is refused -- never a silent single-rank run
    env.pop('WCT_BENCH_SHARE_GPU')
    import torch
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--size', '64', '--steps', '1', '--warmup', '0',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and '"n_gpus"' not in out.stdout


def test_two_ranks_gathered_frames_equal_the_single_rank_frames_bit_for_bit(ctx, weights):
    """The N > 1 DEVICE path end to end (VERDICT r4 missing #4): `bench.py --gpus 2 --global-batch 6` at 512x512 -- two ranks
    (sharing the one GPU of this box, exchange through gloo), each stylizing its contiguous shard of six DISTINCT pairs from
    device-resident inputs (shard_range -> stylize_batch_dev at the shard's byte offsets) and ONE gather to rank 0 (gather_frames).
    The digest of the gathered [6][512][512][3] tensor must be the digest of the frames a single rank computes for the same six
    pairs: `bench.py --gpus 1 --global-batch 6`, and stylize_batch called here directly.  A wrong shard offset, a wrong gather
    order or a rank reading another rank's inputs changes it."""
    import hashlib
    import json
    import subprocess
    import sys
    from conftest import ROOT
    n, size = 6, 512
    content = np.stack([synthetic_image(1000 + i, size, size) for i in range(n)])
    style = np.stack([synthetic_image(2000 + i, size, size) for i in range(n)])
    direct = ctx.stylize_batch(content, style, RELU_TARGETS, alpha=0.8)
    assert len({hashlib.sha256(f.tobytes()).hexdigest() for f in direct}) == n          # six different frames
    want = hashlib.sha256(np.ascontiguousarray(direct).tobytes()).hexdigest()
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    digests = {}
    for gpus in (1, 2):
        e = dict(env, WCT_BENCH_BACKEND='gloo', WCT_BENCH_SHARE_GPU='1') if gpus > 1 else env
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--size', str(size), '--global-batch', str(n),
                              '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-latency', '--no-prof'],
                             env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith('{')][-1])
        assert line['n_gpus'] == gpus and line['config']['global_batch'] == n and line['scaling'] == 'strong'
        digests[gpus] = line['frames_sha256']
        if gpus > 1:
            assert line['dist']['world_size'] == 2 and line['dist']['backend'] == 'gloo'
            assert line['dist']['gather_ms_unoverlapped'] > 0 and len(line['dist']['rank_ms_per_step_min_max']) == 2
    print('frames digest: direct %s, bench --gpus 1 %s, bench --gpus 2 %s' % (want[:16], digests[1][:16], digests[2][:16]))
    assert digests[1] == want and digests[2] == want


def test_two_gpus_rccl_gathered_frames_equal_the_single_rank_frames(ctx, weights):
    """The day a box has two GPUs: `bench.py --gpus 2 --global-batch 6` on the REAL backend -- one rank per GPU,
    torch.distributed 'nccl' = RCCL over xGMI, the uint8 gather on its side stream overlapped with the next step -- must report
    backend nccl, world size 2 and the digest of the frames a single rank computes.  SKIPPED (not passed) on one-GPU boxes: RCCL
    refuses two ranks on one device (profiles/r02_dryrun_2ranks_1gpu_rccl_refused.txt), so until such a box appears the overlapped
    RCCL gather has only ever run with one rank."""
    import hashlib
    import json
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs: the nccl (RCCL) backend cannot place two ranks on one device')
    n, size = 6, 512
    content = np.stack([synthetic_image(1000 + i, size, size) for i in range(n)])
    style = np.stack([synthetic_image(2000 + i, size, size) for i in range(n)])
    want = hashlib.sha256(np.ascontiguousarray(ctx.stylize_batch(content, style, RELU_TARGETS, alpha=0.8)).tobytes()).hexdigest()
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'WCT_BENCH_BACKEND', 'WCT_BENCH_SHARE_GPU')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', str(size), '--global-batch', str(n),
                          '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-latency', '--no-prof'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['dist']['backend'] == 'nccl' and line['dist']['world_size'] == 2
    assert line['frames_sha256'] == want


def test_swap5_pipeline(ctx, weights):
    """--swap5: style-swap at relu5_1 (priority over adain), WCT below.  The fused call must equal the
    GPU ops chained by hand; the relu5_1 op is checked against the oracle on the oracle's features."""
    from wct_tf_amd import _lib
    targets = ['relu5_1', 'relu2_1']
    c = synthetic_image(1003, 128, 96)
    s = synthetic_image(2003, 112, 112)
    _, levels = oracle.stylize(c, s, weights, targets, alpha=0.8, swap5=True, ss_alpha=0.6, return_levels=True)
    fc, fs, t, _ = levels[0]
    got_t = ctx.style_swap(fc, fs, 0.6)
    e = rel_err(got_t, t)
    print('swap5 relu5_1 op on oracle features: rel %.2e' % e, fc.shape, fs.shape)
    assert e < 1e-3
    ctx.set_style_swap(0.6, 3, 1)
    got = ctx.stylize(c, s, targets, alpha=0.8, swap5=True, adain=True)       # swap5 wins at relu5_1, adain below
    x = np.float32(c / 255.)
    s01 = np.float32(s / 255.)
    t5 = ctx.style_swap(ctx.encode(x, 'relu5_1'), ctx.encode(s01, 'relu5_1'), 0.6)
    x = np.clip(ctx.decode(t5, 'relu5_1'), 0, 1)
    from wct_tf_amd import ops
    t2 = ops.adain(ctx.encode(x, 'relu2_1'), ctx.encode(s01, 'relu2_1'), 0.8, ctx=ctx)[0]
    x = ctx.decode(t2, 'relu2_1')
    assert np.array_equal(got, np.uint8(np.clip(x, 0, 1) * 255))


def test_config5_1024_content_512_style_adain_keepcolors(ctx, weights):
    """BASELINE config 5: 1024x1024 content / 512x512 style, --keep-colors CORAL first, then the --adain
    branch AND the WCT branch.  Full size, so properties instead of an oracle run: CORAL against the
    float64 oracle within 1 LSB; determinism; AdaIN at alpha = 1 gives every level's output the style's
    per-channel mean/std; the WCT branch reproduces the style covariance on the 1024^2-derived features."""
    from wct_tf_amd import ops, _lib
    content = synthetic_image(1005, 1024, 1024)
    style = synthetic_image(2005, 512, 512)
    style_cc = ops.preserve_colors_np(style, content, ctx=ctx)
    want_cc = oracle.preserve_colors_np(style, content)
    d = np.abs(style_cc.astype(int) - want_cc.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    out_a = ctx.stylize(content, style_cc, RELU_TARGETS, alpha=0.8, adain=True)
    out_b = ctx.stylize(content, style_cc, RELU_TARGETS, alpha=0.8, adain=True)
    assert out_a.shape == (1024, 1024, 3) and np.array_equal(out_a, out_b)
    out_w = ctx.stylize(content, style_cc, RELU_TARGETS, alpha=0.8)
    assert out_w.shape == (1024, 1024, 3) and not np.array_equal(out_w, out_a)
    c01, s01 = np.float32(content / 255.), np.float32(style_cc / 255.)
    for relu in ('relu4_1', 'relu1_1'):
        fc, fs = ctx.encode(c01, relu), ctx.encode(s01, relu)
        ch = fc.shape[-1]
        y = ops.adain(fc, fs, 1.0, ctx=ctx)[0].reshape(-1, ch).astype(np.float64)
        s2 = fs.reshape(-1, ch).astype(np.float64)
        assert np.abs(y.mean(0) - s2.mean(0)).max() < 1e-3 * max(1.0, np.abs(s2.mean(0)).max())
        assert np.abs(y.std(0) - s2.std(0)).max() < 2e-3 * max(1.0, s2.std(0).max())
    fc, fs = ctx.encode(c01, 'relu3_1'), ctx.encode(s01, 'relu3_1')       # Nc = 65536, Ns = 16384
    out = ctx.transform(fc.reshape(-1, 256), fs.reshape(-1, 256), 1.0, _lib.WCT_TF).astype(np.float64)
    s2 = fs.reshape(-1, 256).astype(np.float64)
    assert np.linalg.norm(np.cov(out.T) - np.cov(s2.T)) / np.linalg.norm(np.cov(s2.T)) < 5e-3


@pytest.mark.parametrize('adain', [False, True])
def test_config5_levels_teacher_forced_1024_content_512_style(ctx, weights, adain):
    """BASELINE config 5's SHAPE against the oracle, teacher-forced: content with four times the style's pixels (Nc = 4 Ns at every
    level; since round 6 at 512 x 512 / 256 x 256, see below), five levels, alpha 0.8 -- the WCT
    branch and the --adain branch (ops.py:282-294, stylize.py:85-100), every level's encoder, transform and decoder on
    the oracle's own level inputs."""
    # (round 6, the suite's time budget on the driver -- 681 of its 1200 s in round 5, this test 147 s of them: both branches at
    #  512 / 256, the same 4 : 1 pixel ratio between content and style at every level; the FULL config-5 sizes run end to end,
    #  chained, in both branches in test_config5_chained_end_to_end_1024_content_512_style, the transform at Nc = 1 048 576 in
    #  test_wct_config_sizes and the convolutions at 1024 x 1024 in the chained test's frames)
    size = 512
    content = synthetic_image(1005, size, size)
    style = synthetic_image(2005, size // 2, size // 2)
    want = _teacher_forced(ctx, weights, content, style, RELU_TARGETS, 0.8, 'tf', adain=adain)
    assert want.shape == (size, size, 3)


def test_predict_does_not_quantise_float_images(ctx, weights):
    """wct.py:60-64 divides a float image by 255 as it is; the library's float path (WCT_FLAG_IMAGES_F32) must give the
    frame the oracle gives for the float image -- and not the one it gives for the image rounded down to integers."""
    targets = ['relu2_1', 'relu1_1']
    rng = np.random.default_rng(4)
    c8, s8 = synthetic_image(1000, 64, 80), synthetic_image(2000, 72, 56)
    cf = np.clip(c8 + rng.uniform(0, 0.999, c8.shape), 0, 255)          # same integer part, non-integer values
    sf = np.clip(s8 + rng.uniform(0, 0.999, s8.shape), 0, 255)
    got = ctx.stylize(cf, sf, targets, alpha=0.8)
    want = oracle.stylize(cf, sf, weights, targets, alpha=0.8)
    quantised = oracle.stylize(np.uint8(cf), np.uint8(sf), weights, targets, alpha=0.8)
    d = np.abs(got.astype(int) - want.astype(int))
    dq = np.abs(got.astype(int) - quantised.astype(int))
    print('float predict vs oracle(float): psnr %.1f dB mean %.3f LSB; vs oracle(uint8(image)): psnr %.1f dB mean %.3f LSB'
          % (psnr(got, want), d.mean(), psnr(got, quantised), dq.mean()))
    assert psnr(got, want) > 35 and psnr(got, want) > psnr(got, quantised) + 3
    assert np.array_equal(ctx.stylize(np.float64(c8), np.float32(s8), targets, alpha=0.8), ctx.stylize(c8, s8, targets, alpha=0.8))


def test_abi_error_paths_on_gpu(ctx):
    """Errors cross the ABI as status codes with a message (no exceptions, no silent fallback)."""
    import ctypes as C
    from wct_tf_amd import _lib
    from wct_tf_amd.context import Context
    from wct_tf_amd._lib import WCTHipError
    fresh = Context(0)
    try:
        with pytest.raises(WCTHipError, match='encoder weights not set'):
            fresh.encode(np.zeros((16, 16, 3), np.float32), 'relu1_1')
        with pytest.raises(WCTHipError, match='decoder weights'):
            fresh.decode(np.zeros((4, 4, 64), np.float32), 'relu1_1')
        with pytest.raises(WCTHipError, match='invalid argument'):
            fresh.transform(np.zeros((8, 48), np.float32), np.zeros((8, 48), np.float32), 0.5, _lib.WCT_NP)   # C % 32
        with pytest.raises(ValueError):
            fresh.set_decoder('relu2_1', [(np.zeros((3, 3, 128, 64), np.float32), np.zeros(64, np.float32))])   # 3 layers needed
    finally:
        fresh.close()
    with pytest.raises(WCTHipError):
        Context(99)                                           # no such device


@pytest.mark.parametrize('adain,swap5', [(False, False), (True, False), (False, True)])
def test_shared_style_batch_equals_per_pair(ctx, weights, adain, swap5):
    """WCT_FLAG_STYLE_SHARED (one style for all frames of a batch: the video loop, stylize_video.py:112-121):
    the style encoder pass, statistics and eigensystems run once per call -- and every frame must equal the
    frame the per-pair path produces with that style, bit for bit."""
    targets = ['relu5_1', 'relu3_1', 'relu1_1'] if swap5 else ['relu4_1', 'relu2_1', 'relu1_1']
    B = 5
    frames = np.stack([synthetic_image(300 + i, 64, 48) for i in range(B)])
    style = synthetic_image(400, 56, 72)
    if swap5:
        ctx.set_style_swap(0.6, 3, 1)
    shared = ctx.stylize_batch(frames, style, targets, alpha=0.7, adain=adain, swap5=swap5)
    repl = ctx.stylize_batch(frames, np.stack([style] * B), targets, alpha=0.7, adain=adain, swap5=swap5)
    assert shared.shape == repl.shape == (B, 64, 48, 3)
    assert np.array_equal(shared, repl)
    one = ctx.stylize(frames[3], style, targets, alpha=0.7, adain=adain, swap5=swap5)
    assert np.array_equal(shared[3], one)
    assert len({shared[i].tobytes() for i in range(B)}) == B          # the frames do differ


def test_video_cli_end_to_end(tmp_path):
    """python -m wct_tf_amd.stylize_video on a directory of frames (ffmpeg is absent): frames of two sizes, one
    style, --concat; the written frames equal WCT.predict on the same inputs."""
    from wct_tf_amd import utils
    from wct_tf_amd.stylize_video import main
    from wct_tf_amd.wct import WCT
    targets = ['relu3_1', 'relu1_1']
    in_dir = tmp_path / 'clip'
    in_dir.mkdir()
    frames = [synthetic_image(500 + i, 40, 56) for i in range(5)] + [synthetic_image(510, 32, 32)]
    for i, f in enumerate(frames):
        utils.save_img(str(in_dir / ('frame_%d.png' % (i + 1))), f)
    style = synthetic_image(600, 48, 48)
    utils.save_img(str(tmp_path / 'style.png'), style)
    out_dir = tmp_path / 'out'
    n = main(['--relu-targets'] + targets + ['--in-path', str(in_dir), '--style-path', str(tmp_path / 'style.png'),
              '--out-path', str(out_dir), '--alpha', '0.8', '--synthetic-weights', '42', '--batch', '4', '--concat'])
    assert n == 6
    model = WCT(checkpoints=None, relu_targets=targets, vgg_path=None, weights=synthetic_weights(42, relu_targets=targets))
    for i, f in enumerate(frames):
        got = utils.get_img(str(out_dir / 'clip_style' / ('frame_%d.png' % (i + 1))))
        want = model.predict(f, style, 0.8)
        want = np.hstack([utils._imresize(style, (want.shape[0], want.shape[0])), want])
        assert np.array_equal(got, want), i


def test_wct_facade_from_tf_checkpoint_dirs(tmp_path, weights):
    """WCT(checkpoints=[dir, ...]) with TensorFlow V2 bundles (the reference's own decoder format, wct.py:46-58):
    same frames as with the weights handed over directly."""
    from oracle.tf_ckpt_writer import write_bundle, write_checkpoint_state
    from wct_tf_amd.weights import decoder_plan, save_weights
    from wct_tf_amd.wct import WCT
    targets = ['relu2_1', 'relu1_1']
    dirs = []
    for relu in targets:
        t, count, convs = {}, 0, iter(weights['decoder'][relu])
        for kind, cin, cout, _ in decoder_plan(relu):
            if kind == 'U':
                count += 1
                continue
            w, b = next(convs)
            base = 'encoder_decoder_{r}/decoder_{r}/decoder_model_{r}/{r}_{c}/'.format(r=relu, c=count)
            t[base + 'kernel'], t[base + 'bias'] = np.float32(w), np.float32(b)
            t[base + 'kernel/Adam'] = np.zeros_like(np.float32(w))
            count += 1
        d = tmp_path / ('ckpt_' + relu)
        d.mkdir()
        write_bundle(str(d / 'model.ckpt-15000'), t, block_size=512)
        write_checkpoint_state(str(d), 'model.ckpt-15000')
        dirs.append(str(d))
    vgg = str(tmp_path / 'vgg.npz')
    save_weights(vgg, {'encoder': weights['encoder'], 'decoder': {}})
    content, style = synthetic_image(700, 40, 48), synthetic_image(701, 32, 32)
    a = WCT(checkpoints=dirs, relu_targets=targets, vgg_path=vgg).predict(content, style, 0.6)
    b = WCT(checkpoints=None, relu_targets=targets, vgg_path=None, weights=weights).predict(content, style, 0.6)
    assert np.array_equal(a, b)
    with pytest.raises(Exception, match='No checkpoint found for target relu1_1'):
        WCT(checkpoints=[dirs[0], str(tmp_path)], relu_targets=targets, vgg_path=vgg)


def test_two_contexts_two_threads_and_no_memory_growth(weights):
    """Two contexts (two HIP streams) driven concurrently by two host threads give the frames of a serial run
    (thread-local error/flag state, per-context arenas); and once the arenas have seen the largest shapes,
    further calls at any smaller or equal shape allocate nothing (no growth of device memory)."""
    import threading
    import torch
    from wct_tf_amd.context import Context
    targets = ['relu4_1', 'relu2_1', 'relu1_1']
    ctxs = [Context(0), Context(0)]
    for c in ctxs:
        c.set_weights(weights)
    jobs = [[(synthetic_image(800 + 10 * t + i, 48 + 8 * i, 64), synthetic_image(900 + 10 * t + i, 40, 56)) for i in range(3)]
            for t in range(2)]
    serial = [[ctxs[0].stylize(c, s, targets, alpha=0.8) for c, s in jobs[t]] for t in range(2)]
    got = [None, None]

    def work(t):
        got[t] = [ctxs[t].stylize(c, s, targets, alpha=0.8) for c, s in jobs[t] for _ in range(1)]
    ths = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for t in range(2):
        for a, b in zip(got[t], serial[t]):
            assert np.array_equal(a, b)
    # memory: after the runs above every shape has been seen by ctxs[0]
    ctxs[0].sync(); ctxs[1].sync()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        for c, s in jobs[0] + jobs[1]:
            ctxs[0].stylize(c, s, targets, alpha=0.8)
    ctxs[0].sync()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    for c in ctxs:
        c.close()


def test_epilogue_statistics_equal_the_pass_over_the_features(ctx, weights, tmp_path):
    """The per-channel sums / maxima the transform needs come out of the conv epilogues that write the feature taps
    (16-pixel unit sums, ConvArgs::usum; csrc/conv.hip) whenever the tap's width is a multiple of 16.  The frames must be
    the ones the separate pass over the stored features gives (WCT_FUSE_STATS=0, read once per process: a subprocess), bit
    for bit -- WCT and AdaIN, a batch, and a content whose deeper taps have widths that are not multiples of 16 (mixed)."""
    import subprocess, sys
    rng = np.random.default_rng(5)
    cs = rng.integers(0, 256, (3, 160, 224, 3), dtype=np.uint8)        # tap widths 224, 112, 56, 28, 14: fused at the first two
    st = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)           # 128, 64, 32, 16, 8: fused at four levels
    np.savez(tmp_path / 'in.npz', cs=cs, st=st)
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from wct_tf_amd.context import Context\n"
        "from wct_tf_amd.weights import synthetic_weights, RELU_TARGETS\n"
        "d = np.load(%r)\n"
        "c = Context(0); c.set_weights(synthetic_weights(seed=42))\n"
        "a = c.stylize_batch(d['cs'], d['st'], RELU_TARGETS, alpha=0.8)\n"
        "b = c.stylize_batch(d['cs'], d['st'], RELU_TARGETS, alpha=0.8, adain=True)\n"
        "np.savez(%r, a=a, b=b)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / 'in.npz'),
                                       str(tmp_path / 'out.npz')))
    env = dict(os.environ, WCT_FUSE_STATS='0')
    subprocess.run([sys.executable, '-c', script], check=True, env=env, timeout=600)
    ref = np.load(tmp_path / 'out.npz')
    got_wct = ctx.stylize_batch(cs, st, RELU_TARGETS, alpha=0.8)
    got_adain = ctx.stylize_batch(cs, st, RELU_TARGETS, alpha=0.8, adain=True)
    assert np.array_equal(got_wct, ref['a'])
    assert np.array_equal(got_adain, ref['b'])
    assert len({f.tobytes() for f in got_wct}) == 3


def test_conv1_1_inside_conv1_2s_patch_loader_gives_the_bits_of_the_two_launches(ctx, weights, tmp_path):
    """Encoder passes whose relu1_1 feeds nothing but conv1_2 (every content pass of a level >= 2) compute conv1_1 inside
    conv1_2's patch loader (ConvArgs::img1, csrc/conv.hip: the 64-channel full-resolution map is never written).  The features
    and the frames must be the ones of the two-launch path (WCT_FUSE_CONV1=0, read once per process: a subprocess), bit for
    bit: full tiles, ragged tiles in both directions, images smaller than a tile, the reflected borders, a batch."""
    import subprocess, sys
    rng = np.random.default_rng(11)
    sizes = [(512, 512), (70, 100), (33, 17), (5, 7), (160, 224), (64, 48)]
    imgs = {('i%d' % i): rng.random((h, w, 3), dtype=np.float32) * 1.2 - 0.1 for i, (h, w) in enumerate(sizes)}
    cs = rng.integers(0, 256, (3, 96, 80, 3), dtype=np.uint8)
    st = rng.integers(0, 256, (72, 64, 3), dtype=np.uint8)
    np.savez(tmp_path / 'in.npz', cs=cs, st=st, **imgs)
    body = (
        "d = np.load(IN)\n"
        "c = Context(0); c.set_weights(synthetic_weights(seed=42))\n"
        "out = {}\n"
        "for k in sorted(k for k in d.files if k.startswith('i')):\n"
        "    for lv in ('relu2_1', 'relu3_1'):\n"
        "        if min(d[k].shape[:2]) >= 5 or lv == 'relu2_1':\n"
        "            out[k + lv] = c.encode(d[k], lv)\n"
        "out['frames'] = c.stylize_batch(d['cs'], d['st'], RELU_TARGETS, alpha=0.8)\n")
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from wct_tf_amd.context import Context\n"
        "from wct_tf_amd.weights import synthetic_weights, RELU_TARGETS\n"
        "IN = %r\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / 'in.npz'))
        + body + "np.savez(%r, **out)\n" % str(tmp_path / 'out.npz'))
    env = dict(os.environ, WCT_FUSE_CONV1='0')
    subprocess.run([sys.executable, '-c', script], check=True, env=env, timeout=600)
    ref = np.load(tmp_path / 'out.npz')
    n = 0
    for k in sorted(imgs):
        for lv in ('relu2_1', 'relu3_1'):
            if k + lv in ref.files:
                got = ctx.encode(imgs[k], lv)
                assert got.shape == ref[k + lv].shape
                assert np.array_equal(got, ref[k + lv]), (k, lv, float(np.abs(got - ref[k + lv]).max()))
                assert np.abs(got).max() > 0
                n += 1
    assert n >= 11
    assert np.array_equal(ctx.stylize_batch(cs, st, RELU_TARGETS, alpha=0.8), ref['frames'])


def test_decoder_tail_in_one_launch_gives_the_bits_of_the_two_launches(ctx, weights, tmp_path):
    """Every decoder ends in a 64 -> 64 conv (its input x2-upsampled from relu2_1 up) and the 64 -> 3 output conv (model.py:283-298);
    csrc/conv_tail.hip runs the two as ONE launch with the 64-channel map in LDS.  The decoded images and the
    frames must be the ones of the two-launch path, bit for bit: every decoder, full tiles, ragged tiles in both directions,
    maps smaller than a tile, the reflected borders, a batch.  (The fused launch is NOT the default -- it measured no faster,
    profiles/r06_conv_tail.txt -- so the subprocess is the one that runs it: WCT_FUSE_TAIL=1, read once per process.)"""
    import subprocess, sys
    rng = np.random.default_rng(12)
    feats = {}
    for lv, c, sizes in (('relu1_1', 64, [(48, 32), (37, 29), (5, 3), (16, 16)]), ('relu2_1', 128, [(24, 40), (9, 7), (2, 3)]),
                         ('relu3_1', 256, [(8, 12), (3, 5)]), ('relu4_1', 512, [(6, 4)]), ('relu5_1', 512, [(3, 2)])):
        for i, (h, w) in enumerate(sizes):
            feats['%s/%d' % (lv, i)] = np.maximum(rng.standard_normal((h, w, c)), 0).astype(np.float32)
    cs = rng.integers(0, 256, (3, 96, 80, 3), dtype=np.uint8)
    st = rng.integers(0, 256, (72, 64, 3), dtype=np.uint8)
    np.savez(tmp_path / 'in.npz', cs=cs, st=st, **feats)
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from wct_tf_amd.context import Context\n"
        "from wct_tf_amd.weights import synthetic_weights, RELU_TARGETS\n"
        "d = np.load(%r)\n"
        "c = Context(0); c.set_weights(synthetic_weights(seed=42))\n"
        "out = {k: c.decode(d[k], k.split('/')[0]) for k in d.files if '/' in k}\n"
        "out['frames'] = c.stylize_batch(d['cs'], d['st'], RELU_TARGETS, alpha=0.8)\n"
        "np.savez(%r, **out)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')))
    subprocess.run([sys.executable, '-c', script], check=True, env=dict(os.environ, WCT_FUSE_TAIL='1'), timeout=600)
    ref = np.load(tmp_path / 'out.npz')
    for k, f in feats.items():
        got = ctx.decode(f, k.split('/')[0])
        assert got.shape == ref[k].shape and np.abs(got).max() > 0
        assert np.array_equal(got, ref[k]), (k, float(np.abs(got - ref[k]).max()))
    assert np.array_equal(ctx.stylize_batch(cs, st, RELU_TARGETS, alpha=0.8), ref['frames'])
