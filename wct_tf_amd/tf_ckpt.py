"""Reader for TensorFlow checkpoint V2 bundles (`<prefix>.index` + `<prefix>.data-?????-of-?????`), written
from the published format; no TensorFlow needed.

The reference restores each decoder with `tf.train.Saver(var_list=[v for v in trainable_variables() if
'decoder_<relu>' in v.name])` from `tf.train.get_checkpoint_state(dir)` (wct.py:46-58).  This module does the
same selection on the bundle's keys and returns the decoder's conv layers in graph order, ready for
`wct_tf_amd.weights` / `wct_set_decoder`.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*: a port of LevelDB's table):
  .index  = immutable sorted string table: data blocks, metaindex block, index block, 48-byte footer
            (two block handles, padding, magic 0xdb4775248b80fb57).  A block is a run of prefix-compressed
            entries (varint shared, varint non_shared, varint value_len, key suffix, value), a uint32 array of
            restart offsets and its length; on disk it is followed by 1 byte compression type (0 = none,
            1 = snappy) and a masked crc32c.  Key "" holds BundleHeaderProto, every other key is a tensor
            name holding BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}.
  .data-* = the raw little-endian tensor bytes at [offset, offset+size) of shard `shard_id`.

PARITY STATUS: there is no TensorFlow in this environment and the reference ships no checkpoint, so this
reader is pinned only against bundles produced by oracle/tf_ckpt_writer.py (same published format, written
independently of the reader's code paths: multi-block tables, prefix compression, restarts, crc32c) -- not
against a file written by TensorFlow itself.  DESIGN.md records it as "unpinned by the reference".
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
           10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


class CheckpointError(Exception):
    pass


# ---------------------------------------------------------------------------
# crc32c (Castagnoli), table driven; used for block trailers and, on request, tensor data
# ---------------------------------------------------------------------------
def _make_crc_table():
    poly = 0x82F63B78
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    c = (~crc) & 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return (~c) & 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------
# varints and the three protobuf messages that matter
# ---------------------------------------------------------------------------
def _varint(buf, pos):
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise CheckpointError('varint too long')


def _fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message; value is int or bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise CheckpointError('truncated protobuf field')
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError('unsupported protobuf wire type %d' % wt)
        yield fn, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for fn, _, v in _fields(buf):
        if fn == 2:                                     # repeated Dim dim = 2 { int64 size = 1; string name = 2; }
            size = 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif fn == 3 and v:
            raise CheckpointError('tensor of unknown rank')
    return tuple(dims)


def _parse_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'sliced': False}
    for fn, _, v in _fields(buf):
        if fn == 1:
            e['dtype'] = v
        elif fn == 2:
            e['shape'] = _parse_shape(v)
        elif fn == 3:
            e['shard_id'] = v
        elif fn == 4:
            e['offset'] = _signed64(v)
        elif fn == 5:
            e['size'] = _signed64(v)
        elif fn == 6:
            e['crc32c'] = v
        elif fn == 7:
            e['sliced'] = True
    return e


def _parse_header(buf):
    h = {'num_shards': 1, 'endianness': 0}
    for fn, _, v in _fields(buf):
        if fn == 1:
            h['num_shards'] = v
        elif fn == 2:
            h['endianness'] = v
    return h


# ---------------------------------------------------------------------------
# the table
# ---------------------------------------------------------------------------
def _block_handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def _read_block(data, off, size, verify=True):
    if off + size + 5 > len(data):
        raise CheckpointError('block handle past the end of the index file')
    contents = data[off:off + size]
    ctype = data[off + size]
    if verify:
        stored = struct.unpack_from('<I', data, off + size + 1)[0]
        if stored != masked_crc32c(data[off:off + size + 1]):
            raise CheckpointError('index block checksum mismatch at offset %d' % off)
    if ctype != 0:
        raise CheckpointError('compressed index block (type %d): snappy is not supported' % ctype)
    return contents


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError('block too small')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise CheckpointError('bad restart array')
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise CheckpointError('corrupt block entry')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """All (key, value) pairs of a table file, in order."""
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise CheckpointError('%s: too short for a table footer' % path)
    footer = data[-48:]
    if struct.unpack('<Q', footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError('%s: not a TensorFlow/LevelDB table (bad magic)' % path)
    _, _, pos = _block_handle(footer, 0)                 # metaindex handle (unused)
    ioff, isize, _ = _block_handle(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, bsize, _ = _block_handle(handle, 0)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


# ---------------------------------------------------------------------------
# the bundle
# ---------------------------------------------------------------------------
class Bundle(object):
    """`prefix` as TensorFlow names it: the files are prefix + '.index' and prefix + '.data-00000-of-0000N'."""

    def __init__(self, prefix, verify=True):
        self.prefix = prefix
        index = prefix + '.index'
        if not os.path.exists(index):
            raise CheckpointError('no such checkpoint index: %s' % index)
        self.entries = {}
        self.header = None
        for k, v in read_table(index, verify):
            if k == b'':
                self.header = _parse_header(v)
            else:
                self.entries[k.decode('utf-8')] = _parse_entry(v)
        if self.header is None:
            raise CheckpointError('%s: no bundle header' % index)
        if self.header['endianness'] != 0:
            raise CheckpointError('big-endian bundle')

    def names(self):
        return sorted(self.entries)

    def tensor(self, name, verify_crc=False):
        e = self.entries[name]
        if e['sliced']:
            raise CheckpointError('%s is stored as slices (partitioned variable): not supported' % name)
        if e['dtype'] not in _DTYPES:
            raise CheckpointError('%s: unsupported dtype enum %d' % (name, e['dtype']))
        dt = np.dtype(_DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise CheckpointError('%s: size %d does not match shape %s' % (name, e['size'], e['shape']))
        shard = '%s.data-%05d-of-%05d' % (self.prefix, e['shard_id'], self.header['num_shards'])
        with open(shard, 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise CheckpointError('%s: data shard %s is truncated' % (name, shard))
        if verify_crc and e['crc32c'] is not None and masked_crc32c(raw) != e['crc32c']:
            raise CheckpointError('%s: tensor checksum mismatch' % name)
        return np.frombuffer(raw, dtype=dt.newbyteorder('<')).astype(dt).reshape(e['shape'])


def latest_checkpoint(checkpoint_dir):
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path (wct.py:52): the `checkpoint` text proto."""
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(state):
        return None
    with open(state) as f:
        m = re.search(r'^model_checkpoint_path:\s*"([^"]+)"', f.read(), re.M)
    if not m:
        return None
    path = m.group(1)
    return path if os.path.isabs(path) else os.path.join(checkpoint_dir, path)


_SLOT = re.compile(r'/(Adam(_\d+)?|Momentum|RMSProp(_\d+)?|ExponentialMovingAverage)$')


def decoder_weights_from_checkpoint(checkpoint, relu_target, verify_crc=False):
    """[(kernel HWIO fp32, bias fp32), ...] of the decoder for `relu_target`, in graph order.

    checkpoint: a directory holding a `checkpoint` state file (what --checkpoints takes, stylize.py:17) or a
    bundle prefix.  Variables are selected as wct.py:48-49 does ('decoder_<relu>' in the name; optimiser slots are
    not trainable variables and are skipped); the decoder's layers are named '<relu>_<count>' with `count`
    running over conv AND upsampling layers (model.py:283-296), so layers are ordered by that number."""
    prefix = checkpoint
    if os.path.isdir(checkpoint):
        prefix = latest_checkpoint(checkpoint)
        if prefix is None:
            raise Exception('No checkpoint found for target {} in dir {}'.format(relu_target, checkpoint))
    b = Bundle(prefix)
    tag = 'decoder_' + relu_target
    layer_re = re.compile(r'(?:^|/)%s_(\d+)(?:/|$)' % re.escape(relu_target))
    names = b.names()
    slotted = {_SLOT.sub('', n) for n in names if _SLOT.search(n)}       # variables the optimiser holds moments for
    cands = {}
    for name in names:
        if tag not in name or _SLOT.search(name):
            continue
        leaf = name.rsplit('/', 1)[-1]
        m = layer_re.search(name)
        if leaf not in ('kernel', 'bias') or not m:
            continue
        cands.setdefault((int(m.group(1)), leaf), []).append(name)
    # The reference instantiates each decoder TWICE: Conv2DReflect builds its Conv2D inside a Lambda (ops.py:17-19), once
    # when build_decoder assembles the Keras model and once more when decoder_model(...) is applied (model.py:171), so a
    # real checkpoint can hold a trained set under .../decoder_model_<relu>/... and an untrained twin beside it.  Both
    # match the selection of wct.py:48-49.  Prefer the set the optimiser trained (it owns Adam slots), then the one
    # under the decoder_model_<relu> scope; anything still ambiguous is an error, never a silent pick.
    layers = {}
    model_scope = 'decoder_model_' + relu_target + '/'
    for (idx, leaf), group in cands.items():
        pick = group
        if len(pick) > 1:
            trained = [n for n in pick if n in slotted]
            pick = trained or pick
        if len(pick) > 1:
            scoped = [n for n in pick if model_scope in n]
            pick = scoped or pick
        if len(pick) > 1:
            raise CheckpointError('%s_%d/%s is stored %d times and nothing tells the trained copy apart: %s'
                                  % (relu_target, idx, leaf, len(pick), ', '.join(sorted(pick))))
        layers.setdefault(idx, {})[leaf] = pick[0]
    if not layers:
        raise Exception('No variables containing {} in checkpoint {}'.format(tag, prefix))
    out = []
    for idx in sorted(layers):
        d = layers[idx]
        if 'kernel' not in d or 'bias' not in d:
            raise CheckpointError('layer %s_%d lacks a kernel or a bias' % (relu_target, idx))
        w = np.asarray(b.tensor(d['kernel'], verify_crc), np.float32)
        bias = np.asarray(b.tensor(d['bias'], verify_crc), np.float32)
        if w.ndim != 4 or w.shape[:2] != (3, 3) or bias.shape != (w.shape[3],):
            raise CheckpointError('%s: unexpected conv shapes %s / %s' % (d['kernel'], w.shape, bias.shape))
        out.append((w, bias))
    return out
