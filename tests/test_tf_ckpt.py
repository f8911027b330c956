"""wct_tf_amd/tf_ckpt.py (TensorFlow checkpoint V2 bundle reader, no TensorFlow) against bundles written by
oracle/tf_ckpt_writer.py.  There is no TensorFlow here and the reference ships no checkpoint: the reader is
pinned by an independently written writer of the same published format, not by TensorFlow itself."""
import os
import struct

import numpy as np
import pytest

from oracle.tf_ckpt_writer import write_bundle, write_checkpoint_state
from wct_tf_amd import tf_ckpt
from wct_tf_amd.weights import decoder_plan

SCOPE = 'encoder_decoder_{r}/decoder_{r}/decoder_model_{r}/'


def decoder_variables(relu, rng, with_slots=True):
    """Variables as the reference's graph names them: layer '<relu>_<count>', count running over conv AND
    upsampling layers (model.py:283-296), under the name scopes of model.py:123,165."""
    t, want, count = {}, [], 0
    for kind, cin, cout, _ in decoder_plan(relu):
        if kind == 'U':
            count += 1
            continue
        base = SCOPE.format(r=relu) + '%s_%d/' % (relu, count)
        w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        t[base + 'kernel'], t[base + 'bias'] = w, b
        if with_slots:
            t[base + 'kernel/Adam'] = np.zeros_like(w)
            t[base + 'kernel/Adam_1'] = np.zeros_like(w)
        want.append((w, b))
        count += 1
    return t, want


@pytest.mark.parametrize('block_size,shards', [(64, 1), (4096, 1), (300, 3)])
def test_bundle_round_trip(tmp_path, block_size, shards):
    rng = np.random.default_rng(1)
    t, want = decoder_variables('relu2_1', rng)
    other, _ = decoder_variables('relu1_1', rng, with_slots=False)        # a second decoder in the same graph
    t.update(other)
    t['global_step'] = np.array([15000], np.int64)
    t['train_relu2_1/beta1_power'] = np.array([0.9], np.float32)
    d = str(tmp_path)
    write_bundle(os.path.join(d, 'model.ckpt-15000'), t, block_size=block_size, restart_interval=4, num_shards=shards)
    write_checkpoint_state(d, 'model.ckpt-15000')
    assert tf_ckpt.latest_checkpoint(d) == os.path.join(d, 'model.ckpt-15000')
    b = tf_ckpt.Bundle(os.path.join(d, 'model.ckpt-15000'))
    assert b.names() == sorted(t)
    for name in t:
        got = b.tensor(name, verify_crc=t[name].nbytes < 4096)
        assert got.dtype == t[name].dtype and np.array_equal(got, t[name]), name
    layers = tf_ckpt.decoder_weights_from_checkpoint(d, 'relu2_1')        # directory, as --checkpoints takes it
    assert len(layers) == len(want) == 3
    for (w, bias), (w0, b0) in zip(layers, want):
        assert np.array_equal(w, w0) and np.array_equal(bias, b0)
    assert len(tf_ckpt.decoder_weights_from_checkpoint(os.path.join(d, 'model.ckpt-15000'), 'relu1_1')) == 2


def test_twin_decoder_variable_sets(tmp_path):
    """the reference builds every decoder twice (the Conv2D of Conv2DReflect lives in a Lambda, ops.py:17-19;
    build_decoder instantiates it and decoder_model(...) at model.py:171 instantiates it again), so a real checkpoint
    can hold the trained set under .../decoder_model_<relu>/... AND an untrained twin under .../decoder_<relu>/<relu>_N/.
    Both match wct.py:48-49's selection.  The trained set (it owns Adam slots / sits in the decoder_model scope) must
    win whatever the name order; twins that cannot be told apart are an error, not a silent pick."""
    rng = np.random.default_rng(5)
    relu = 'relu2_1'
    t, want = decoder_variables(relu, rng)                       # trained set, with Adam slots
    twin_scope = 'encoder_decoder_{r}/decoder_{r}/'.format(r=relu)
    count = 0
    for kind, cin, cout, _ in decoder_plan(relu):
        if kind == 'U':
            count += 1
            continue
        base = twin_scope + '%s_%d/' % (relu, count)
        t[base + 'kernel'] = rng.standard_normal((3, 3, cin, cout)).astype(np.float32)      # untrained: no slots
        t[base + 'bias'] = np.zeros(cout, np.float32)
        count += 1
    d = str(tmp_path / 'a')
    os.makedirs(d)
    write_bundle(os.path.join(d, 'model.ckpt-9'), t, block_size=256)
    write_checkpoint_state(d, 'model.ckpt-9')
    got = tf_ckpt.decoder_weights_from_checkpoint(d, relu)
    for (w, b), (w0, b0) in zip(got, want):
        assert np.array_equal(w, w0) and np.array_equal(b, b0)
    # without optimiser slots the decoder_model_<relu> scope still identifies the set decoder_model(...) uses
    t2 = {k: v for k, v in t.items() if '/Adam' not in k}
    d2 = str(tmp_path / 'b')
    os.makedirs(d2)
    write_bundle(os.path.join(d2, 'model.ckpt-9'), t2, block_size=256)
    write_checkpoint_state(d2, 'model.ckpt-9')
    got = tf_ckpt.decoder_weights_from_checkpoint(d2, relu)
    assert all(np.array_equal(w, w0) for (w, _), (w0, _) in zip(got, want))
    # two indistinguishable sets: refuse
    t3 = {k.replace('decoder_model_' + relu, 'other_scope'): v for k, v in t2.items()}
    d3 = str(tmp_path / 'c')
    os.makedirs(d3)
    write_bundle(os.path.join(d3, 'model.ckpt-9'), t3, block_size=256)
    write_checkpoint_state(d3, 'model.ckpt-9')
    with pytest.raises(tf_ckpt.CheckpointError, match='stored 2 times'):
        tf_ckpt.decoder_weights_from_checkpoint(d3, relu)


def test_bundle_errors(tmp_path):
    rng = np.random.default_rng(2)
    t, _ = decoder_variables('relu1_1', rng, with_slots=False)
    d = str(tmp_path)
    prefix = os.path.join(d, 'model.ckpt-1')
    write_bundle(prefix, t, block_size=128)
    # the reference's own failure mode: no checkpoint state in the directory (wct.py:58)
    with pytest.raises(Exception, match='No checkpoint found for target relu1_1'):
        tf_ckpt.decoder_weights_from_checkpoint(d, 'relu1_1')
    write_checkpoint_state(d, 'model.ckpt-1')
    with pytest.raises(Exception, match='No variables containing decoder_relu4_1'):
        tf_ckpt.decoder_weights_from_checkpoint(d, 'relu4_1')
    # flipped byte inside the first data block -> block checksum
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[10] ^= 0xFF
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(tf_ckpt.CheckpointError, match='checksum'):
        tf_ckpt.Bundle(prefix)
    raw[10] ^= 0xFF
    raw[-1] ^= 0x01                                                         # magic
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(tf_ckpt.CheckpointError, match='bad magic'):
        tf_ckpt.Bundle(prefix)
    raw[-1] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(raw))
    b = tf_ckpt.Bundle(prefix)
    # corrupt tensor bytes are caught by the per-tensor crc32c when asked for
    name = [n for n in b.names() if n.endswith('bias')][0]
    e = b.entries[name]
    shard = prefix + '.data-00000-of-00001'
    data = bytearray(open(shard, 'rb').read())
    data[e['offset']] ^= 0x40
    open(shard, 'wb').write(bytes(data))
    with pytest.raises(tf_ckpt.CheckpointError, match='tensor checksum'):
        b.tensor(name, verify_crc=True)


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC32C
    assert tf_ckpt.crc32c(b'\x00' * 32) == 0x8A9136AA
    assert tf_ckpt.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert tf_ckpt.crc32c(bytes(range(32))) == 0x46DD794E
    assert tf_ckpt.crc32c(b'123456789') == 0xE3069283


def test_convert_tool_reads_both_reference_formats(tmp_path):
    """wct_tf_amd.convert: Torch7 encoder + TF decoder checkpoints -> the .npz container, which loads back equal."""
    from conftest import GOLDEN
    from wct_tf_amd.convert import main
    from wct_tf_amd.weights import load_weights
    rng = np.random.default_rng(3)
    t, want = decoder_variables('relu1_1', rng)
    d = tmp_path / 'ck'
    d.mkdir()
    write_bundle(str(d / 'model.ckpt-7'), t, block_size=200)
    write_checkpoint_state(str(d), 'model.ckpt-7')
    out = str(tmp_path / 'all.npz')
    main(['--vgg-path', os.path.join(GOLDEN, 'tiny_vgg.t7'), '--checkpoints', str(d), '--relu-targets', 'relu1_1', '--out', out])
    w = load_weights(out)
    assert 'conv1_1' in w['encoder'] and 'preprocess' in w['encoder']
    for (a, b), (a0, b0) in zip(w['decoder']['relu1_1'], want):
        assert np.array_equal(a, a0) and np.array_equal(b, b0)


def test_stylize_cli_rank_shard():
    from wct_tf_amd.stylize import rank_shard
    files = ['f%02d' % i for i in range(10)]
    assert rank_shard(files, {}) == (files, None)
    seen = []
    for r in range(4):
        shard, dev = rank_shard(files, {'WORLD_SIZE': '4', 'RANK': str(r), 'LOCAL_RANK': str(r)})
        assert dev == '/gpu:%d' % r
        seen += shard
    assert seen == files
