"""Minimal Torch7 binary WRITER used only to manufacture test fixtures (TEST INFRASTRUCTURE).
It emits the subset a VGG-like nn.Sequential needs, so that both the reference's torchfile.py
(in oracle/make_golden.py) and wct_tf_amd/t7.py can be run on the same file."""
import struct

import numpy as np


class Writer(object):
    def __init__(self):
        self.b = bytearray()
        self.next_ref = 1

    def i32(self, v):
        self.b += struct.pack('<i', v)

    def i64(self, v):
        self.b += struct.pack('<q', v)

    def string(self, s):
        s = s if isinstance(s, bytes) else s.encode()
        self.i32(len(s))
        self.b += s

    def obj(self, o):
        if o is None:
            self.i32(0)
        elif isinstance(o, bool):
            self.i32(5)
            self.i32(1 if o else 0)
        elif isinstance(o, (int, float)):
            self.i32(1)
            self.b += struct.pack('<d', float(o))
        elif isinstance(o, (str, bytes)):
            self.i32(2)
            self.string(o)
        elif isinstance(o, np.ndarray):
            self.tensor(o)
        elif isinstance(o, (list, tuple)):
            self.table({i + 1: v for i, v in enumerate(o)})
        elif isinstance(o, dict) and '_typename' in o:
            self.i32(4)
            self.i32(self._ref())
            self.string('V 1')
            self.string(o['_typename'])
            self.table({k: v for k, v in o.items() if k != '_typename'})
        elif isinstance(o, dict):
            self.table(o)
        else:
            raise TypeError(type(o))

    def _ref(self):
        r = self.next_ref
        self.next_ref += 1
        return r

    def table(self, d):
        self.i32(3)
        self.i32(self._ref())
        self.i32(len(d))
        for k, v in d.items():
            self.obj(k)
            self.obj(v)

    def tensor(self, a):
        a = np.ascontiguousarray(a, np.float32)
        self.i32(4)
        self.i32(self._ref())
        self.string('V 1')
        self.string('torch.FloatTensor')
        self.i32(a.ndim)
        for s in a.shape:
            self.i64(s)
        for s in a.strides:
            self.i64(s // 4)
        self.i64(1)
        self.i32(4)
        self.i32(self._ref())
        self.string('V 1')
        self.string('torch.FloatStorage')
        self.i64(a.size)
        self.b += a.tobytes()


def write_vgg_like(path, convs, seed=0):
    """convs: [(name or None, cin, cout, k)] interleaved with ReflectionPadding/ReLU/MaxPooling like the
    real file; returns the {name: (w_oihw, bias)} that was written."""
    rng = np.random.default_rng(seed)
    modules, truth = [], {}
    for i, (name, cin, cout, k) in enumerate(convs):
        if k == 3:
            modules.append({'_typename': 'nn.SpatialReflectionPadding', 'pad_l': 1, 'pad_r': 1, 'pad_t': 1, 'pad_b': 1})
        w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        m = {'_typename': 'nn.SpatialConvolution', 'nInputPlane': cin, 'nOutputPlane': cout, 'kH': k, 'kW': k,
             'weight': w, 'bias': b, 'train': False}
        if name is not None:
            m['name'] = name
        modules.append(m)
        truth[name if name else 'module%d' % i] = (w, b)
        if k == 3:
            modules.append({'_typename': 'nn.ReLU', 'name': 'relu' + name[4:], 'inplace': True})
        if name in ('conv1_2', 'conv2_2'):
            modules.append({'_typename': 'nn.SpatialMaxPooling', 'name': 'pool' + name[4], 'kW': 2, 'kH': 2, 'dW': 2, 'dH': 2})
    w = Writer()
    w.obj({'_typename': 'nn.Sequential', 'modules': modules, 'train': False})
    with open(path, 'wb') as f:
        f.write(bytes(w.b))
    return truth
