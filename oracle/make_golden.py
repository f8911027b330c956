"""Generate tests/golden/*.npz by EXECUTING the reference's own functions.

Run in the build container (the only place /root/reference exists):
    python -m oracle.make_golden

* `wct_np` is lifted out of /root/reference/ops.py by `ast` (importing ops.py
  fails on `import tensorflow`, ops.py:3) and executed with NumPy only.
* `coral_numpy` / `preserve_colors_np` are imported from the reference's
  coral.py / utils.py as they are.
Nothing from the reference is copied into this repository -- only the numeric
outputs of running it on seeded inputs.  The fixtures pin oracle/wct_oracle.py.
"""
import ast
import os
import sys

import numpy as np

REF = os.environ.get('WCT_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, os.path.dirname(HERE))

from wct_tf_amd.weights import synthetic_features  # noqa: E402  (seeded inputs only)


def lift_function(path, name):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {'np': np}
            exec(compile(mod, path, 'exec'), ns)
            return ns[name]
    raise KeyError(name)


WCT_CASES = [
    # name, C, (hc,wc), (hs,ws), alpha (None -> reference default), decades, rank
    ('c64_default', 64, (20, 20), (16, 24), None, 3.0, None),
    ('c64_alpha1', 64, (24, 16), (24, 16), 1.0, 3.0, None),
    ('c128_alpha08', 128, (16, 16), (20, 12), 0.8, 2.0, None),
    ('c64_rankdef', 64, (6, 6), (5, 7), 0.8, 1.0, 20),
    ('c32_alpha0', 32, (8, 8), (8, 8), 0.0, 2.0, None),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_wct_np = lift_function(os.path.join(REF, 'ops.py'), 'wct_np')
    blob = {}
    for i, (name, c, (hc, wc), (hs, ws), alpha, dec, rank) in enumerate(WCT_CASES):
        content = synthetic_features(100 + i, c, hc, wc, dec, rank)
        style = synthetic_features(200 + i, c, hs, ws, dec, rank)
        out = ref_wct_np(content, style) if alpha is None else ref_wct_np(content, style, alpha)
        blob[name + '/content'] = content
        blob[name + '/style'] = style
        blob[name + '/alpha'] = np.float64(-1.0 if alpha is None else alpha)
        blob[name + '/out'] = out
    np.savez_compressed(os.path.join(OUT, 'wct_np_reference.npz'), **blob)

    sys.path.insert(0, REF)
    import coral as ref_coral          # noqa: E402
    import utils as ref_utils          # noqa: E402
    rng = np.random.default_rng(7)
    blob = {}
    for i, ((hs, ws), (ht, wt)) in enumerate([((24, 20), (32, 28)), ((16, 16), (16, 16))]):
        src = rng.integers(0, 256, (hs, ws, 3)).astype(np.uint8)
        tgt = (rng.integers(0, 256, (ht, wt, 3)) * np.array([1.0, 0.6, 0.3])).astype(np.uint8)
        blob['case%d/source' % i] = src
        blob['case%d/target' % i] = tgt
        blob['case%d/coral' % i] = ref_coral.coral_numpy(src / 255., tgt / 255.)
        blob['case%d/preserve' % i] = ref_utils.preserve_colors_np(src, tgt)
    np.savez_compressed(os.path.join(OUT, 'coral_reference.npz'), **blob)

    # .t7 fixture: a small VGG-shaped nn.Sequential, parsed by the REFERENCE's torchfile.py the way
    # vgg_normalised.py:16-34 does; the arrays it extracts are the golden values for wct_tf_amd/t7.py
    from oracle.t7_writer import write_vgg_like
    import torchfile as ref_torchfile      # noqa: E402  (the reference's vendored reader, executed not copied)
    t7_path = os.path.join(OUT, 'tiny_vgg.t7')
    write_vgg_like(t7_path, [(None, 3, 3, 1), ('conv1_1', 3, 8, 3), ('conv1_2', 8, 8, 3), ('conv2_1', 8, 16, 3)], seed=3)
    t7 = ref_torchfile.load(t7_path, force_8bytes_long=True)
    blob = {}
    for idx, module in enumerate(t7.modules):
        name = module.name.decode() if module.name is not None else None
        if idx == 0:
            name = 'preprocess'
        blob['typenames/%d' % idx] = np.frombuffer(module._typename, dtype=np.uint8)
        if module._typename == b'nn.SpatialConvolution':
            blob['%s/w_hwio' % name] = module.weight.transpose([2, 3, 1, 0])
            blob['%s/b' % name] = module.bias
    np.savez_compressed(os.path.join(OUT, 't7_reference.npz'), **blob)
    print('wrote', os.listdir(OUT))


if __name__ == '__main__':
    main()
