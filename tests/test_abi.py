"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, and
exports every symbol include/wct_hip.h declares; host-side logic (descriptors, facade
errors) behaves like the reference's."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def lib():
    from wct_tf_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_all_exported_and_bound(lib):
    from wct_tf_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'wct_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)          # drop comments
    declared = set(re.findall(r'^\s*(?:int|void|const char\*)\s+(wct_[a-z0-9_]+)\s*\(', header, re.M))
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name


def test_errors_cross_the_abi_as_status_codes(lib):
    import ctypes as C
    from wct_tf_amd import _lib
    rc = lib.wct_sync(None)
    assert rc == -2 and b'invalid argument' in lib.wct_last_error()
    ho, wo = C.c_int(), C.c_int()
    lv = (C.c_int * 5)(5, 4, 3, 2, 1)
    assert lib.wct_output_size(500, 500, lv, 5, C.byref(ho), C.byref(wo)) == 0
    assert (ho.value, wo.value) == (512, 512)          # 500 -> ceil-pooled 32 -> x16 = 512
    assert lib.wct_output_size(512, 384, lv, 5, C.byref(ho), C.byref(wo)) == 0
    assert (ho.value, wo.value) == (512, 384)
    bad = (C.c_int * 1)(7)
    assert lib.wct_output_size(64, 64, bad, 1, C.byref(ho), C.byref(wo)) == -2


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from wct_tf_amd.context import Context
    from wct_tf_amd._lib import WCTHipError
    with pytest.raises(WCTHipError):
        Context(0)


def test_model_descriptor_mirrors_reference_attributes():
    from wct_tf_amd import WCTModel
    m = WCTModel(mode='test', relu_targets=['relu3_1', 'relu1_1'])
    for attr in ('content_input', 'style_input', 'alpha', 'swap5', 'ss_alpha', 'use_adain',
                 'decoded_output', 'encoder_decoders', 'vgg_model'):
        assert hasattr(m, attr)
    assert m.deepest_target == 'relu3_1' and len(m.encoder_decoders) == 2
    assert m.encoder_decoders[1].content_input == 'clip(relu3_1.decoded)'
    assert m.encoder_decoders[0].train_op is None and m.encoder_decoders[0].total_loss is None      # model.py:206-208
    t = WCTModel(mode='train', relu_targets=['relu2_1'], batch_size=4, feature_weight=2.0, learning_rate=1e-3, lr_decay=0.1)
    ed = t.encoder_decoders[0]
    assert ed.train_op['call'] == 'wct_train_step' and ed.train_op['level'] == 2
    assert ed.total_loss is not None and ed.decoded_encoded is not None and t.feature_weight == 2.0
    with pytest.raises(RuntimeError):
        m.train_step(None, None, 0)
    with pytest.raises(ValueError):
        WCTModel(mode='predict')


def test_weights_roundtrip(tmp_path):
    from wct_tf_amd.weights import synthetic_weights, save_weights, load_weights
    w = synthetic_weights(relu_targets=['relu2_1'])
    p = str(tmp_path / 'w.npz')
    save_weights(p, w)
    r = load_weights(p)
    assert set(r['encoder']) == set(w['encoder'])
    assert np.array_equal(r['decoder']['relu2_1'][2][0], w['decoder']['relu2_1'][2][0])
    assert np.array_equal(r['encoder']['conv3_1'][1], w['encoder']['conv3_1'][1])


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'wct_tf_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_cli_flags_match_reference_and_utils(tmp_path):
    from wct_tf_amd.stylize import build_parser
    from wct_tf_amd import utils
    flags = {a for act in build_parser()._actions for a in act.option_strings}
    for f in ['--checkpoints', '--relu-targets', '--vgg-path', '--content-path', '--style-path', '--out-path',
              '--keep-colors', '--device', '--style-size', '--crop-size', '--content-size', '--passes', '-r',
              '--random', '--alpha', '--concat', '--adain', '--swap5', '--ss-alpha', '--ss-patch-size', '--ss-stride']:
        assert f in flags, f                                   # stylize.py:16-37
    img = np.uint8(np.random.default_rng(0).integers(0, 256, (60, 90, 3)))
    assert utils.resize_to(img, 30).shape == (30, 45, 3)      # short side -> 30, aspect kept
    assert utils.center_crop(img, 40).shape == (40, 40, 3)
    assert utils.center_crop(img, 80).shape == (80, 80, 3)    # upscales first when too small
    p = str(tmp_path / 'x.png')
    utils.save_img(p, img)
    assert np.array_equal(utils.get_img(p), img)


def test_video_cli_flags_match_reference(tmp_path):
    from wct_tf_amd.stylize_video import build_parser, list_frames
    from wct_tf_amd import utils
    flags = {a for act in build_parser()._actions for a in act.option_strings}
    for f in ['--checkpoints', '--relu-targets', '--vgg-path', '--in-path', '--out-path', '--style-path', '--tmp-dir',
              '--keep-tmp', '--keep-colors', '--style-size', '--crop-size', '--content-size', '--passes', '--device',
              '--alpha', '--concat', '--swap5', '--ss-alpha', '--ss-patch-size', '--ss-stride']:
        assert f in flags, f                                   # stylize_video.py:16-41
    # ffmpeg numbers frames frame_%d.png (stylize_video.py:76): order them numerically, not lexically
    img = np.zeros((4, 4, 3), np.uint8)
    for i in (10, 2, 1, 11):
        utils.save_img(str(tmp_path / ('frame_%d.png' % i)), img)
    assert [os.path.basename(f) for f in list_frames(str(tmp_path))] == \
        ['frame_1.png', 'frame_2.png', 'frame_10.png', 'frame_11.png']


def test_train_cli_flags_match_reference():
    from wct_tf_amd.train import build_parser, torch_decay
    flags = {a for act in build_parser()._actions for a in act.option_strings}
    for f in ['--checkpoint', '--log-path', '--relu-target', '--content-path', '--val-path', '--vgg-path',
              '--feature-weight', '--pixel-weight', '--tv-weight', '--learning-rate', '--lr-decay', '--max-iter',
              '--batch-size', '--save-iter', '--summary-iter', '--max-to-keep']:
        assert f in flags, f                                   # train.py:12-57
    d = build_parser().parse_args(['--checkpoint', 'x', '--relu-target', 'relu3_1'])
    assert (d.feature_weight, d.pixel_weight, d.tv_weight, d.learning_rate, d.lr_decay, d.max_iter, d.batch_size,
            d.save_iter, d.summary_iter, d.max_to_keep) == (1, 1, 0, 1e-4, 0, 16000, 8, 200, 20, 10)
    assert torch_decay(1e-4, 0, 5e-5) == 1e-4 and abs(torch_decay(1e-4, 20000, 5e-5) - 5e-5) < 1e-12   # ops.py:298-309


def test_t7_reader_matches_reference_torchfile():
    """wct_tf_amd/t7.py on tests/golden/tiny_vgg.t7 == what the reference's torchfile.py + the
    vgg_normalised.py:22-34 walk extracted from the same file (fixture made by oracle/make_golden.py)."""
    import os
    from conftest import GOLDEN
    from wct_tf_amd.t7 import load_t7, vgg_weights_from_t7
    z = np.load(os.path.join(GOLDEN, 't7_reference.npz'))
    net = load_t7(os.path.join(GOLDEN, 'tiny_vgg.t7'))
    assert net._typename == b'nn.Sequential'
    for idx, module in enumerate(net.modules):
        assert module._typename == z['typenames/%d' % idx].tobytes()
    enc = vgg_weights_from_t7(os.path.join(GOLDEN, 'tiny_vgg.t7'))
    names = sorted({k.split('/')[0] for k in z.files if not k.startswith('typenames')})
    assert names == sorted(enc) == ['conv1_1', 'conv1_2', 'conv2_1', 'preprocess']
    for n in names:
        assert np.array_equal(enc[n][0], z[n + '/w_hwio']) and enc[n][0].dtype == np.float32
        assert np.array_equal(enc[n][1], z[n + '/b'])
    with pytest.raises(ValueError):
        from wct_tf_amd.t7 import T7Reader
        T7Reader(memoryview(b'\x04\x00\x00\x00\x01\x00'), 8).read()      # truncated


def test_profiling_class_list_matches_the_header():
    """The Python mirror's class names (one per id) and the array length the C ABI declares (WCT_PROF_CLASSES) move together."""
    import re
    from wct_tf_amd import _lib
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'wct_hip.h')).read()
    n = int(re.search(r'#define\s+WCT_PROF_CLASSES\s+(\d+)', hdr).group(1))
    assert n == len(_lib.PROF_CLASSES) == 11
    assert _lib.PROF_CLASSES[0] == 'conv3x3' and _lib.PROF_CLASSES[8] == 'conv12'
    assert len(set(_lib.PROF_CLASSES)) == n
