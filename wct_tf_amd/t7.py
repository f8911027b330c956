"""Torch7 `.t7` (binary) reader, written for the one job the reference has for it: pulling the
normalised-VGG19 layer list out of `vgg_normalised.t7` (vgg_normalised.py:16-46, which uses the
third-party torchfile.py with force_8bytes_long=True).  Format facts used (Torch7 File:writeObject):

  object   := int32 tag ; tag 0 nil | 1 number(float64) | 2 string(int32 len + bytes) | 5 boolean(int32)
              | 3 table | 4 torch object   (6/7/8 = functions: not needed, rejected)
  table    := int32 ref-index, int32 n, n x (key object, value object)
  torch    := int32 ref-index, version string ("V <n>" then a class-name string, or just the class name),
              then class payload:
                torch.*Tensor  : int32 ndim, ndim x long size, ndim x long stride, long offset(1-based), storage object
                torch.*Storage : long n, n raw elements
                anything else  : one object (a table of fields)          -- nn.* modules
  long     := 8 bytes (the VGG file was written on a 64-bit box; 4-byte longs via long_size=4)
A ref-index seen before returns the same Python object (shared storages, self references).
"""
import struct

import numpy as np

_STORAGE_DTYPES = {
    b'torch.FloatStorage': np.float32, b'torch.DoubleStorage': np.float64, b'torch.LongStorage': np.int64,
    b'torch.IntStorage': np.int32, b'torch.ShortStorage': np.int16, b'torch.ByteStorage': np.uint8,
    b'torch.CharStorage': np.int8,
}
_TENSOR_TYPES = {k.replace(b'Storage', b'Tensor') for k in _STORAGE_DTYPES}


class T7Object(object):
    """A deserialised Torch class instance (e.g. nn.SpatialConvolution): fields as attributes."""

    def __init__(self, typename, fields):
        self._typename = typename
        self._fields = fields if isinstance(fields, dict) else {}

    def __getattr__(self, name):
        f = self.__dict__.get('_fields', {})
        if name in f:
            return f[name]
        if name.encode() in f:
            return f[name.encode()]
        return None                     # a missing Lua field is nil (the reference relies on this for .name)

    def __repr__(self):
        return 'T7Object(%s)' % self._typename.decode()


class T7Reader(object):
    def __init__(self, data, long_size=8):
        self.d = data
        self.p = 0
        self.long_fmt = '<q' if long_size == 8 else '<i'
        self.long_size = long_size
        self.refs = {}

    def _take(self, n):
        if self.p + n > len(self.d):
            raise ValueError('truncated .t7 file')
        b = self.d[self.p:self.p + n]
        self.p += n
        return b

    def _int(self):
        return struct.unpack('<i', self._take(4))[0]

    def _long(self):
        return struct.unpack(self.long_fmt, self._take(self.long_size))[0]

    def _string(self):
        return bytes(self._take(self._int()))

    def read(self):
        tag = self._int()
        if tag == 0:
            return None
        if tag == 1:
            x = struct.unpack('<d', self._take(8))[0]
            return int(x) if float(x).is_integer() else x
        if tag == 2:
            return self._string()
        if tag == 5:
            return self._int() == 1
        if tag == 3:
            idx = self._int()
            if idx in self.refs:
                return self.refs[idx]
            n = self._int()
            table = {}
            self.refs[idx] = table
            for _ in range(n):
                k = self.read()
                table[k] = self.read()
            # list-like tables (keys 1..n) become Python lists, as Lua arrays are used
            if n > 0 and all(isinstance(k, int) for k in table) and sorted(table) == list(range(1, n + 1)):
                lst = [table[i] for i in range(1, n + 1)]
                self.refs[idx] = lst
                return lst
            return table
        if tag == 4:
            idx = self._int()
            if idx in self.refs:
                return self.refs[idx]
            version = self._string()
            cls = self._string() if version.startswith(b'V ') else version
            if cls in _STORAGE_DTYPES:
                n = self._long()
                dt = np.dtype(_STORAGE_DTYPES[cls])
                arr = np.frombuffer(self._take(n * dt.itemsize), dtype=dt).copy()
                self.refs[idx] = arr
                return arr
            if cls in _TENSOR_TYPES:
                ndim = self._int()
                size = [self._long() for _ in range(ndim)]
                stride = [self._long() for _ in range(ndim)]
                offset = self._long() - 1
                storage = self.read()
                if storage is None or ndim == 0:
                    arr = np.zeros(size, _STORAGE_DTYPES[cls.replace(b'Tensor', b'Storage')])
                else:
                    arr = np.lib.stride_tricks.as_strided(
                        storage[offset:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
                self.refs[idx] = arr
                return arr
            obj = T7Object(cls, None)
            self.refs[idx] = obj
            fields = self.read()
            obj._fields = fields if isinstance(fields, dict) else {}
            return obj
        raise ValueError('unsupported .t7 object tag %d (functions are not needed for weight files)' % tag)


def load_t7(path, long_size=8):
    with open(path, 'rb') as f:
        return T7Reader(memoryview(f.read()), long_size).read()


def vgg_weights_from_t7(path):
    """The encoder half of the weights container from `vgg_normalised.t7`, following
    vgg_normalised.py:22-46: module 0 is the 1x1 'preprocess' conv, convolutions are named conv1_1...,
    weights (O,I,kH,kW) -> HWIO by transpose([2,3,1,0]) (vgg_normalised.py:33)."""
    net = load_t7(path)
    enc = {}
    for idx, module in enumerate(net.modules):
        if module._typename != b'nn.SpatialConvolution':
            continue
        name = 'preprocess' if idx == 0 else (module.name.decode() if module.name is not None else None)
        if name is None:
            raise ValueError('unnamed convolution at module %d' % idx)
        w = np.asarray(module.weight, np.float32).transpose([2, 3, 1, 0])
        enc[name] = (np.ascontiguousarray(w), np.asarray(module.bias, np.float32).copy())
        if name == 'conv5_1':
            break
    return enc
