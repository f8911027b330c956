"""TEST INFRASTRUCTURE ONLY.  Minimal writer for TensorFlow checkpoint V2 bundles, used to make fixtures for
wct_tf_amd/tf_ckpt.py (no TensorFlow here).  Written from the published format
(tensorflow/core/util/tensor_bundle/tensor_bundle.cc, tensorflow/core/lib/io/table_builder.cc, format.cc,
block_builder.cc) independently of the reader: its own varint/protobuf encoders and a bitwise crc32c.

    write_bundle(prefix, {name: ndarray}, block_size=4096, restart_interval=16)
    write_checkpoint_state(dir, 'model.ckpt-15000')
"""
import os
import struct

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9,
       np.dtype(np.float16): 19}


def _crc32c_bitwise(data):
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _vint(n):
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_varint(num, v):
    return _vint(num << 3) + _vint(v)


def _field_bytes(num, payload):
    return _vint((num << 3) | 2) + _vint(len(payload)) + payload


def _entry_proto(arr, shard, offset, crc):
    shape = b''.join(_field_bytes(2, _field_varint(1, int(d))) for d in arr.shape)
    return (_field_varint(1, _DT[arr.dtype]) + _field_bytes(2, shape) + (_field_varint(3, shard) if shard else b'') +
            (_field_varint(4, offset) if offset else b'') + _field_varint(5, arr.nbytes) +
            _vint((6 << 3) | 5) + struct.pack('<I', crc))


def _header_proto(num_shards):
    version = _field_varint(1, 1)                        # VersionDef.producer = 1
    return _field_varint(1, num_shards) + _field_bytes(3, version)      # endianness LITTLE = 0 is the default


class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.ri = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count < self.ri:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _vint(shared) + _vint(len(key) - shared) + _vint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))

    def empty(self):
        return not self.buf


def _write_block(f, contents):
    off = f.tell()
    trailer_in = contents + b'\x00'                      # kNoCompression
    f.write(contents)
    f.write(b'\x00' + struct.pack('<I', _mask(_crc32c_bitwise(trailer_in))))
    return _vint(off) + _vint(len(contents))             # BlockHandle


def _write_table(path, items, block_size, restart_interval):
    """items: sorted [(key bytes, value bytes)]"""
    with open(path, 'wb') as f:
        index = _BlockBuilder(1)
        blk = _BlockBuilder(restart_interval)
        for k, v in items:
            blk.add(k, v)
            if blk.size() >= block_size:
                index.add(blk.last, _write_block(f, blk.finish()))     # index key: >= last key of the block
                blk = _BlockBuilder(restart_interval)
        if not blk.empty():
            index.add(blk.last, _write_block(f, blk.finish()))
        meta_handle = _write_block(f, _BlockBuilder(1).finish())
        index_handle = _write_block(f, index.finish())
        footer = meta_handle + index_handle
        f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC))


def write_bundle(prefix, tensors, block_size=4096, restart_interval=16, num_shards=1):
    """tensors: {name: ndarray}.  Tensor i goes to shard i % num_shards."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    shards = [open('%s.data-%05d-of-%05d' % (prefix, s, num_shards), 'wb') for s in range(num_shards)]
    items = [(b'', _header_proto(num_shards))]
    for i, name in enumerate(sorted(tensors)):
        arr = np.ascontiguousarray(tensors[name])
        raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
        s = i % num_shards
        off = shards[s].tell()
        shards[s].write(raw)
        items.append((name.encode('utf-8'), _entry_proto(arr, s, off, _mask(_crc32c_bitwise(raw)))))
    for f in shards:
        f.close()
    _write_table(prefix + '.index', sorted(items), block_size, restart_interval)


def write_checkpoint_state(directory, basename):
    with open(os.path.join(directory, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (basename, basename))
