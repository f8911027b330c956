"""Randomised GPU parity sweep (hypothesis, derandomised so every run sees the same cases): shapes, ranges and
parameters the hand-written cases do not enumerate -- every call through the C ABI, checked against the CPU oracle.

What the sweep is after:
  * WCT: any C in {32..256 step 32} and C = 512, ragged / tiny pixel counts (N >= 2, N < C included), feature scales from 1e-3
    to 1e3 (the covariance and apply GEMMs split their operands into fp16 hi+lo pairs after a per-matrix
    power-of-two scaling: the result must not depend on the magnitude of the input), both semantics, any alpha;
  * AdaIN: same shapes;
  * conv3x3: any H, W >= 2, channel counts from the path, with and without the folded upsample / ReLU;
  * CORAL: arbitrary image sizes (exact integer moments)."""
import contextlib
import os

import numpy as np
import pytest

hypothesis = pytest.importorskip('hypothesis')
from hypothesis import example, given, settings, strategies as st, HealthCheck, Phase  # noqa: E402

import oracle  # noqa: E402
from conftest import rel_err, max_rel  # noqa: E402
from wct_tf_amd import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
# no shrinking phase: a failing example is reported as drawn (shrinking re-runs the oracle's SVDs for minutes while the GPU
# box idles -- 5 of the 10 minutes of the round-5 lease that found the wide-band excess below)
# WCT_FUZZ_SCALE=k (tools/gpu_fuzz_wide.sh): k times the examples of every sweep, drawn at RANDOM instead of derandomised -- the
# wide runs between rounds that look for what the fixed sequence does not draw (round 6: the noise-above-the-cut-off case)
FUZZ_SCALE = int(os.environ.get('WCT_FUZZ_SCALE', '1'))
COMMON = dict(deadline=None, derandomize='WCT_FUZZ_SCALE' not in os.environ, suppress_health_check=list(HealthCheck),
              phases=(Phase.explicit, Phase.reuse, Phase.generate), database=None)
STATS = {'wct_cases': 0, 'wct_near_cutoff': 0, 'wct_indeterminate': 0, 'wct_wide': 0, 'wct_wide_worst': 0.0}


@pytest.fixture(scope='module')
def ctx():
    from wct_tf_amd.context import Context
    c = Context(0)
    yield c
    c.close()


@contextlib.contextmanager
def _memoised_svd():
    """np.linalg.svd with its results cached by the argument's bytes (test infrastructure: the oracle's keep= sweeps decompose the
    same two covariances at every candidate)."""
    real, cache = np.linalg.svd, {}

    def svd(a, *args, **kw):
        key = (a.shape, a.dtype.str, hash(a.tobytes()), args, tuple(sorted(kw.items())))
        if key not in cache:
            cache[key] = real(a, *args, **kw)
        return cache[key]
    np.linalg.svd = svd
    try:
        yield
    finally:
        np.linalg.svd = real


def features(rng, n, c, scale, mix=True):
    g = rng.standard_normal((n, c))
    if mix:
        g = g @ (rng.standard_normal((c, c)) / np.sqrt(c))
    g = np.maximum(g, 0) * 10.0 ** rng.uniform(-0.7, 0.7, c)
    return np.float32(g * scale)


def _wct_case(ctx, c, hc, wc, hs, ws, alpha, mode, log_scale, seed):
    rng = np.random.default_rng(seed)
    scale = 10.0 ** log_scale
    nc, ns = hc * wc, hs * ws
    fc, fs = features(rng, nc, c, scale), features(rng, ns, c, scale * 10.0 ** rng.uniform(-1, 1))
    fn = oracle.wct_np if mode == 'np' else oracle.wct_tf
    print('wct case: C=%d N=%d/%d (%dx%d, %dx%d) alpha=%r mode=%s log_scale=%r seed=%d' % (c, nc, ns, hc, wc, hs, ws, alpha, mode, log_scale, seed))
    got = ctx.transform(fc, fs, alpha, _lib.WCT_NP if mode == 'np' else _lib.WCT_TF)
    assert np.all(np.isfinite(got))
    # The reference drops eigenvalues <= 1e-5 (absolute, ops.py:68-69 / 112,125).  At tiny feature scales an
    # eigenvalue can sit within fp32 noise of that threshold (these covariances are not graded: the noise of ANY
    # fp32 evaluation, NumPy's included, is ~1e-6 ||A||), and keeping or dropping it are both legitimate readings
    # of the reference.  Criterion, also for those cases: the result must match the oracle for SOME kept count
    # inside the noise band on each side -- and exactly the reference's count when no eigenvalue is in the band.
    eps_cov = 1e-8 if mode == 'tf' else 0.0
    ranges = []
    for x in (fc, fs):
        ev = np.linalg.eigvalsh((np.cov(np.float64(x).T) if x.shape[0] > 1 else np.zeros((c, c))) + eps_cov * np.eye(c))
        band = 2e-6 + 1e-6 * np.abs(ev).max()
        ranges.append((int((ev > 1e-5 + band).sum()), int((ev > 1e-5 - band).sum())))
    near = ranges[0][0] != ranges[0][1] or ranges[1][0] != ranges[1][1]
    STATS['wct_cases'] += 1
    if not near or alpha == 0:
        want = np.asarray(fn(fc.reshape(1, hc, wc, c), fs.reshape(1, hs, ws, c), alpha)).reshape(nc, c)
        assert rel_err(got, want) < 1e-3, (c, nc, ns, alpha, mode, log_scale)
        return
    STATS['wct_near_cutoff'] += 1
    shaped = (fc.reshape(1, hc, wc, c), fs.reshape(1, hs, ws, c))
    # kept counts inside the band: all of them when the band is narrow, its ends and their neighbours when it is wide
    if (ranges[0][1] - ranges[0][0] + 1) * (ranges[1][1] - ranges[1][0] + 1) <= (200 if c < 512 else 16):   # (two SVDs per candidate)
        cand = [list(range(r[0], r[1] + 1)) for r in ranges]
    else:
        cand = [sorted({r[0], min(r[0] + 1, r[1]), max(r[1] - 1, r[0]), r[1]}) for r in ranges]
    wide = max(r[1] - r[0] for r in ranges) >= 8
    errs = {}
    with _memoised_svd():                # (the float32 candidates decompose the same two covariances: once)
        for kc in cand[0]:
            for ks in cand[1]:
                # (wide bands are judged against the float64 outcomes below: the float32 ones are not evaluated)
                errs[(kc, ks)] = float('nan') if wide else rel_err(got, np.asarray(fn(*shaped, alpha, keep=(kc, ks))).reshape(nc, c))
    # A WIDE band is a whole cluster of rounding-noise eigenvalues sitting on the cut-off (N < C pixels at a feature
    # scale whose noise is ~1e-5 or above): every noise direction the reference happens to keep is amplified by up to
    # (1e-5)^-1/2 = 316, and its own output is then rounding noise at the 1e-3..1e-2 level -- measured here as the
    # distance between the oracle in float32 (the reference's arithmetic) and in float64.  No implementation can match
    # what the reference does not determine; the tolerance follows that measured indeterminacy.
    o32 = np.asarray(fn(*shaped, alpha)).reshape(nc, c)
    o64 = np.asarray(fn(np.float64(shaped[0]), np.float64(shaped[1]), alpha, **({'dtype': np.float64} if mode == 'tf' else {}))).reshape(nc, c)
    own = rel_err(o32, o64)
    # o32 vs o64 is ONE draw of that rounding noise: narrow bands are judged against the float32 outcomes with 1e-3 or 4x `own`;
    # wide ones (eight or more noise eigenvalues on the cut-off) against the EXACT outcomes, below
    STATS['wct_indeterminate'] += own > 2.5e-4
    print('near cut-off: C=%d N=%d/%d scale 1e%.1f kept-count band %s: best rel %.2e (reference fp32 vs fp64 on this input: %.2e)'
          % (c, nc, ns, log_scale, ranges, min(errs.values()), own))
    if not wide:
        assert min(errs.values()) < max(1e-3, 4 * own), (c, nc, ns, alpha, mode, log_scale, min(errs.values()), own)
        return
    # WIDE band -- the A-B the advisor asked for (r3) and the review repeated (r4): is the distance above the reference's
    # rounding noise, or this path's own error?  Both fp32 evaluations are measured against the EXACT answer: the oracle in
    # float64 at every kept count of the band.  `ref_noise` = how far the reference's own float32 arithmetic lands from the
    # nearest exact outcome; `gpu_exact` = the same for this path.
    # RESULT (MI355X, round 5, profiles/r05_parity_holes.txt): the excess WAS this path's, not reference noise -- on N << C inputs
    # (rank-deficient covariances whose rounding-noise eigenvalues the absolute cut-off keeps, 8 decades below the norm) the
    # reference's float32 landed 9.5e-5 .. 4.6e-4 from the exact outcome and this path 1.06e-3 .. 2.36e-3.  Cause: the solver
    # tracks the rotated matrix D + E and the eigenvectors V separately (fp32 tile updates; 22-bit products), so V^T A0 V = D + E
    # held only to ~1e-6 ||A||, and the spectral functions' completion took that inconsistency for signal; a kept noise direction
    # has a gain of up to 316.  FIX (csrc/wct.hip refresh_needed): for a matrix with a kept eigenvalue 4 decades below its
    # largest the rotated matrix is RECOMPUTED, E' = V^T A0 V, before the spectral functions -- f(A0) = V f(V^T A0 V) V^T is then
    # exact for orthogonal V.  After it the same cases read 3.5e-5 (was 2.36e-3; reference 4.6e-4), 6.1e-7 (5.2e-4; 4.4e-5),
    # 4.1e-6 (7.2e-4; 1.3e-4): the stated budget holds again -- 1e-3, or 4x what the reference's arithmetic loses on the input.
    STATS['wct_wide'] += 1
    sh64 = (np.float64(shaped[0]), np.float64(shaped[1]))
    kw64 = {'dtype': np.float64} if mode == 'tf' else {}
    # Round 6: the ends of a very wide band are not where this path's OWN kept count lies -- the 7-8e-4 cases of round 5
    # were judged at the wrong outcome.  The exact outcome is now searched over the WHOLE band by coordinate descent from the best
    # end candidate (content count with the style count fixed, then the style count, twice); the two float64 SVDs are computed once
    # (np.linalg.svd memoised for the duration: the oracle stays as it is), so a candidate costs a few small products.
    exact = {}
    with _memoised_svd():
        def ex(k):
            if k not in exact:
                exact[k] = np.asarray(fn(*sh64, alpha, keep=k, **kw64)).reshape(nc, c)
            return exact[k]
        best = min(errs, key=lambda k: rel_err(got, ex(k)))
        for _ in range(2):
            best = min(((kc, best[1]) for kc in range(ranges[0][0], ranges[0][1] + 1)), key=lambda k: rel_err(got, ex(k)))
            best = min(((best[0], ks) for ks in range(ranges[1][0], ranges[1][1] + 1)), key=lambda k: rel_err(got, ex(k)))
    gpu_exact = min(rel_err(got, e) for e in exact.values())
    ref_noise = min(rel_err(o32, e) for e in exact.values())
    print('   wide band: %d exact outcomes visited, this path nearest to kept counts %s' % (len(exact), best))
    STATS['wct_wide_worst'] = max(STATS['wct_wide_worst'], gpu_exact)
    print('   wide band: vs the exact (float64) outcomes of the band: this path %.2e, the reference in float32 %.2e'
          % (gpu_exact, ref_noise))
    assert gpu_exact < max(1e-3, 4 * ref_noise), (c, nc, ns, alpha, mode, log_scale, gpu_exact, ref_noise)


FAILS = []


def _wct_case_collect(*args):
    """the suite: the case, as it is.  Wide runs (WCT_FUZZ_SCALE): a failing case is recorded and the sweep goes on, so that one run reports
    ALL the cases it found (test_wct_random_shapes_report fails on them)"""
    if FUZZ_SCALE == 1:
        return _wct_case(*args)
    try:
        _wct_case(*args)
    except AssertionError as e:
        FAILS.append((args[1:], str(e).splitlines()[0][:200]))
        print('FUZZ-FAIL %r: %s' % FAILS[-1])


# (found by a run of this sweep in round 6 -- the generator's sequence depends on the tests that ran before it in the process -- and
# kept as an explicit case: N << C at the top of the scale range, rounding noise of the covariance five decades above the cut-off)
@example(c=256, hc=12, wc=7, hs=13, ws=6, alpha=0.23694110562368303, mode='np', log_scale=3.0, seed=464496)
@example(c=256, hc=12, wc=7, hs=13, ws=6, alpha=0.9, mode='tf', log_scale=3.0, seed=4)
# (found by the first wide run, tools/gpu_fuzz_wide.sh 8 -- 289 cases, these four over the bound, in the round-5 library as well:
#  a graded spectrum whose worst row the MEAN residual hid (1.35e-3; the stop test now bounds the worst row too), and clusters of
#  eigenvalues a few per cent apart around the cut-off (1.1e-3 .. 1.4e-2; the residual across the cut-off is now measured against
#  the GAP) -- profiles/r06_fuzz_wide.txt)
@example(c=64, hc=12, wc=6, hs=20, ws=10, alpha=0.959527218831276, mode='tf', log_scale=1.2633897218325192, seed=5076692)
@example(c=128, hc=12, wc=25, hs=25, ws=23, alpha=0.8727483594839052, mode='np', log_scale=-1.8471459453454355, seed=38)
@example(c=256, hc=8, wc=19, hs=9, ws=17, alpha=0.8948437019837017, mode='np', log_scale=-1.7328125291679197, seed=176)
@settings(max_examples=30 * FUZZ_SCALE, **COMMON)
@given(c=st.sampled_from([32, 64, 96, 128, 160, 256]), hc=st.integers(2, 26), wc=st.integers(2, 26),
       hs=st.integers(2, 26), ws=st.integers(2, 26),
       alpha=st.floats(0.0, 1.0), mode=st.sampled_from(['np', 'tf']), log_scale=st.floats(-3.0, 3.0),
       seed=st.integers(0, 2 ** 31 - 1))
def test_wct_random_shapes_and_scales(ctx, c, hc, wc, hs, ws, alpha, mode, log_scale, seed):
    _wct_case_collect(ctx, c, hc, wc, hs, ws, alpha, mode, log_scale, seed)


@example(hc=14, wc=31, hs=12, ws=29, alpha=0.9304784057570625, mode='tf', log_scale=-1.0695702389412896, seed=501)
@settings(max_examples=6 * FUZZ_SCALE, **COMMON)
@given(hc=st.integers(2, 40), wc=st.integers(2, 40), hs=st.integers(2, 40), ws=st.integers(2, 40),
       alpha=st.floats(0.0, 1.0), mode=st.sampled_from(['np', 'tf']), log_scale=st.floats(-3.0, 3.0),
       seed=st.integers(0, 2 ** 31 - 1))
def test_wct_random_shapes_and_scales_512_channels(ctx, hc, wc, hs, ws, alpha, mode, log_scale, seed):
    """the same sweep at C = 512, the channel count of two of the five levels (relu4_1, relu5_1): pixel counts from 4 to
    1600 on either side of C, so full-rank and rank-deficient covariances both occur"""
    _wct_case_collect(ctx, 512, hc, wc, hs, ws, alpha, mode, log_scale, seed)




def test_wct_random_shapes_report():
    """Runs after the sweep above: how many of its cases had an eigenvalue inside the noise band of the cut-off
    (those were checked against the band of legitimate outcomes instead of being skipped)."""
    print('WCT sweep: %(wct_cases)d cases, %(wct_near_cutoff)d with an eigenvalue within fp32 noise of the 1e-5 cut-off, '
          '%(wct_indeterminate)d of those with a reference output that is itself rounding noise above 2.5e-4, %(wct_wide)d with a band '
          'of eight or more noise eigenvalues (judged against the float64 outcomes: worst %(wct_wide_worst).2e)' % STATS)
    assert STATS['wct_cases'] >= 36
    assert not FAILS, FAILS


@settings(max_examples=15 * FUZZ_SCALE, **COMMON)
@given(c=st.sampled_from([4, 32, 64, 128, 512]), hc=st.integers(2, 30), wc=st.integers(2, 30),
       hs=st.integers(2, 30), ws=st.integers(2, 30),
       alpha=st.floats(0.0, 1.0), log_scale=st.floats(-2.0, 2.0), seed=st.integers(0, 2 ** 31 - 1))
def test_adain_random_shapes(ctx, c, hc, wc, hs, ws, alpha, log_scale, seed):
    rng = np.random.default_rng(seed)
    nc, ns = hc * wc, hs * ws
    fc, fs = features(rng, nc, c, 10.0 ** log_scale, mix=False), features(rng, ns, c, 10.0 ** log_scale, mix=False)
    want = np.asarray(oracle.adain(fc.reshape(1, hc, wc, c), fs.reshape(1, hs, ws, c), alpha)).reshape(nc, c)
    got = ctx.adain(fc, fs, alpha)
    assert rel_err(got, want) < 1e-4 and max_rel(got, want) < 1e-3


@settings(max_examples=25 * FUZZ_SCALE, **COMMON)
@given(h=st.integers(2, 70), w=st.integers(2, 70), cin=st.sampled_from([64, 128, 256]),
       cout=st.sampled_from([64, 128, 256]), relu=st.booleans(), up=st.booleans(), seed=st.integers(0, 2 ** 31 - 1))
def test_conv3x3_random_shapes(ctx, h, w, cin, cout, relu, up, seed):
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((h, w, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    got = ctx.conv3x3(x, wt, b, relu=relu, upsample=up)
    xin = oracle.upsample2x_nearest(x) if up else x
    h16 = lambda a: np.asarray(a, np.float16).astype(np.float32)   # noqa: E731
    want = oracle.conv3x3_reflect(h16(xin), h16(wt), b, relu)
    assert got.shape == want.shape
    assert rel_err(got, want) < 2e-4 and max_rel(got, want) < 1e-3, (h, w, cin, cout, relu, up)


@settings(max_examples=16 * FUZZ_SCALE, **COMMON)
@given(h=st.integers(2, 44), w=st.integers(2, 44), cin=st.sampled_from([256, 512]), cout=st.sampled_from([256, 512]),
       relu=st.booleans(), up=st.booleans(), pool=st.booleans(), batch=st.integers(1, 3), seed=st.integers(0, 2 ** 31 - 1))
def test_conv3x3_reduced_flop_kernel_random_shapes(ctx, h, w, cin, cout, relu, up, pool, batch, seed):
    """csrc/conv_wino.hip (round 6: Winograd F(2,3) along y, direct along x) through wct_conv3x3_f16 with the kernel forced, on
    sizes the pipeline's layers do not enumerate -- down to 2 x 2 (every row a reflected edge), odd heights (a pair-row with one row
    inside), batches, the folded upsample and the fused ceil-mode pool: against the restatement with its own roundings (1e-4;
    one fp16 ulp on an element), the fp32 oracle (the 1.5e-3 gate of the layer), and the batch must not matter."""
    if pool and not relu:
        return                                    # (the pipeline pools only behind a ReLU; the entry point refuses the combination)
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((batch, h, w, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.1).astype(np.float32)
    got = ctx.conv3x3_f16(x, wt, b, relu=relu, upsample=up, pool=pool, algo=2)
    h16 = lambda a: np.asarray(a, np.float16).astype(np.float32)   # noqa: E731
    for i in range(batch):
        xin = oracle.upsample2x_nearest(x[i]) if up else x[i]
        emu = oracle.conv3x3_reflect_wino_f16(xin, wt, b, relu)
        want32 = oracle.conv3x3_reflect(h16(xin), wt, b, relu)
        if pool:
            emu, want32 = oracle.maxpool2x2_same(emu), oracle.maxpool2x2_same(want32)
        assert got[i].shape == want32.shape, (h, w, cin, cout, relu, up, pool, batch)
        # (a handful of fp16 rounding flips -- up to 2^-10 of an element each -- weigh 1 / sqrt(n) in the norm of an n-element output:
        #  a 2 x 2 image pooled to one pixel has 256 elements, and two flips read 1.2e-4 there)
        assert rel_err(got[i], emu) < 1e-4 + 2e-3 / np.sqrt(emu.size) and max_rel(got[i], emu) < 2e-3, (h, w, cin, cout, relu, up, pool, batch, i)
        assert rel_err(got[i], want32) < 1.5e-3, (h, w, cin, cout, relu, up, pool, batch, i)
    if batch > 1:
        assert np.array_equal(got[0], ctx.conv3x3_f16(x[0], wt, b, relu=relu, upsample=up, pool=pool, algo=2))


@settings(max_examples=10 * FUZZ_SCALE, **COMMON)
@given(c=st.sampled_from([64, 128, 256]), hc=st.integers(5, 18), wc=st.integers(5, 18), hs=st.integers(5, 18), ws=st.integers(5, 18),
       patch=st.sampled_from([1, 3]), alpha=st.floats(0.1, 1.0), seed=st.integers(0, 10 ** 6))
def test_style_swap_random_shapes(ctx, c, hc, wc, hs, ws, patch, alpha, seed):
    """wct_style_swap (ops.py:145-278: whiten both sides, match every content patch to its best style patch by normalised correlation,
    paste, colour) on random map sizes, N < C included.  The match is an argmax: a pixel may differ from the oracle's only inside the
    footprint of a position the oracle itself decides by less than 1e-3 of the correlation (tests/test_gpu_ops.py::test_style_swap)."""
    from wct_tf_amd import ops
    from wct_tf_amd.weights import synthetic_features
    fc = synthetic_features(seed, c, hc, wc, 1.5)
    fs = synthetic_features(seed + 1, c, hs, ws, 1.5)
    want, margins = oracle.wct_style_swap(fc, fs, alpha, patch, 1, return_margins=True)
    got = ops.wct_style_swap(fc, fs, alpha, patch, 1, ctx=ctx)
    assert got.shape == want.shape
    diff = np.abs(got - want).max(-1).reshape(hc, wc) > 1e-3 * np.abs(want).max()
    near = np.zeros(diff.shape, bool)
    for y, x in zip(*np.nonzero(margins < 1e-3)):
        near[y:y + patch, x:x + patch] = True
    assert rel_err(got, want) < 1e-3 or not np.any(diff & ~near), (c, hc, wc, hs, ws, patch, alpha, seed, int((diff & ~near).sum()))


@settings(max_examples=12 * FUZZ_SCALE, **COMMON)
@given(hs=st.integers(1, 90), ws=st.integers(1, 90), ht=st.integers(1, 90), wt=st.integers(1, 90),
       seed=st.integers(0, 2 ** 31 - 1))
def test_coral_random_sizes(ctx, hs, ws, ht, wt, seed):
    from wct_tf_amd.ops import preserve_colors_np
    rng = np.random.default_rng(seed)
    style = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
    content = rng.integers(0, 256, (ht, wt, 3), dtype=np.uint8)
    if hs * ws < 4 or ht * wt < 4:
        return                                    # degenerate covariances: the reference divides by zero / inverts a singular matrix
    want = oracle.preserve_colors_np(style, content)
    got = preserve_colors_np(style, content, ctx=ctx)
    assert got.shape == want.shape
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1


@settings(max_examples=12 * FUZZ_SCALE, **COMMON)
@given(hc=st.integers(16, 90), wc=st.integers(16, 90), hs=st.integers(16, 90), ws=st.integers(16, 90),
       levels=st.lists(st.sampled_from([5, 4, 3, 2, 1]), min_size=1, max_size=4, unique=True),
       alpha=st.floats(0.0, 1.0), mode=st.sampled_from(['tf', 'np']), adain=st.booleans(), seed=st.integers(0, 1000))
def test_fused_pipeline_equals_chained_ops_random(hc, wc, hs, ws, levels, alpha, mode, adain, seed):
    """wct_stylize (one fused call: images resident, fp16 hand-over between the stages, pools fused into the convs)
    against the same GPU ops chained through the layer-level ABI, bit for bit, on random sizes / level subsets."""
    from wct_tf_amd.context import Context
    from wct_tf_amd.weights import synthetic_weights, synthetic_image
    levels = sorted(levels, reverse=True)
    targets = ['relu%d_1' % l for l in levels]
    global _PIPE
    try:
        _PIPE
    except NameError:
        _PIPE = Context(0)
        _PIPE.set_weights(synthetic_weights(42))
    ctx = _PIPE
    c, s = synthetic_image(3000 + seed, hc, wc), synthetic_image(4000 + seed, hs, ws)
    need = (1 << (levels[0] - 1)) + 1                     # the deepest feature map must be >= 2x2 (reflect pad)
    if min(hc, wc, hs, ws) < need:
        with pytest.raises(_lib.WCTHipError, match='too small for relu%d_1' % levels[0]):
            ctx.stylize(c, s, targets, alpha=alpha, wct_mode=mode, adain=adain)
        return
    got = ctx.stylize(c, s, targets, alpha=alpha, wct_mode=mode, adain=adain)
    x, s01 = np.float32(c / 255.), np.float32(s / 255.)
    for i, relu in enumerate(targets):
        if i > 0:
            x = np.clip(x, 0, 1)
        fc, fs = ctx.encode(x, relu), ctx.encode(s01, relu)
        ch = fc.shape[-1]
        if adain:
            t = ctx.adain(fc.reshape(-1, ch), fs.reshape(-1, ch), alpha).reshape(fc.shape)
        else:
            if fc.shape[0] * fc.shape[1] < 2 or fs.shape[0] * fs.shape[1] < 2:
                return                                   # a 1-pixel feature map has no covariance (the reference divides by 0)
            t = ctx.transform(fc.reshape(-1, ch), fs.reshape(-1, ch), alpha,
                              _lib.WCT_TF if mode == 'tf' else _lib.WCT_NP).reshape(fc.shape)
        x = ctx.decode(t, relu)
    want = np.uint8(np.clip(x, 0, 1) * 255)
    assert got.shape == want.shape
    assert np.array_equal(got, want), (hc, wc, hs, ws, levels, alpha, mode, adain)
