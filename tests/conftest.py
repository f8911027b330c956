import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def rel_err(a, b):
    import numpy as np
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_rel(a, b):
    import numpy as np
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check_against_size_digest(z, case, out, tol):
    """Compare a full-size output [N][C] with the digest the reference left in wct_np_sizes.npz; returns the
    three relative errors (sampled rows, +-1 sketch over all pixels, per-channel mean square)."""
    import numpy as np
    from oracle.make_golden import digest_selectors
    name = case[0]
    rows, signs = digest_selectors(case)
    o = np.asarray(out).reshape(-1, case[1])
    e_rows = rel_err(o[rows], z[name + '/rows'])
    e_sketch = rel_err(signs @ o.astype(np.float64), z[name + '/sketch'])
    e_sq = rel_err((o.astype(np.float64) ** 2).mean(0), z[name + '/sq'])
    e_mean = max_rel(o.astype(np.float64).mean(0), z[name + '/mean'])
    assert e_rows < tol and e_sketch < 2 * tol and e_sq < 2 * tol and e_mean < tol, (name, e_rows, e_sketch, e_sq, e_mean)
    return e_rows, e_sketch, e_sq
