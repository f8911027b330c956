"""NumPy model of the GPU's BLOCK Jacobi (blocks of 32, pair problems of 64, one cross sweep per pair and outer step) with
(a) the cyclic round-robin block pairing the library uses, (b) dynamic pairing: a greedy maximum-weight matching on the
off-block Frobenius norms at every outer step.  Counts outer steps until the strict residual is below the stop threshold."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from jacobi_precondition_proto import strict_r2  # noqa

B = 32

def rot_params(app, aqq, apq):
    rot = np.abs(apq) > 1e-6 * np.sqrt(np.abs(app * aqq))
    tau = 0.5 * (aqq - app)
    h = np.sqrt(tau * tau + apq * apq)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = np.where(rot, np.abs(apq) / (np.abs(tau) + h), 0.0)
    t = np.where((tau >= 0) == (apq >= 0), t, -t)
    c = 1.0 / np.sqrt(1.0 + t * t)
    return c, c * t

def apply_set(S, Q, p, q):
    c, s = rot_params(S[p, p], S[q, q], S[p, q])
    for M in (S, Q):
        Mp, Mq = M[:, p].copy(), M[:, q].copy()
        M[:, p] = c * Mp - s * Mq
        M[:, q] = s * Mp + c * Mq
    Sp, Sq = S[p, :].copy(), S[q, :].copy()
    S[p, :] = c[:, None] * Sp - s[:, None] * Sq
    S[q, :] = s[:, None] * Sp + c[:, None] * Sq

def cross_problem(S):
    """one cross sweep on a 2B x 2B problem: B sets, pair j with B + (j + s) % B"""
    Q = np.eye(2 * B)
    j = np.arange(B)
    for s in range(B):
        apply_set(S, Q, j, B + (j + s) % B)
    return Q

def intra_problem(S):
    """pairs inside each half: round robin on B indices, both halves at once"""
    Q = np.eye(2 * B)
    idx = list(range(B))
    for s in range(B - 1):
        p = np.array([idx[i] for i in range(B // 2)]); q = np.array([idx[B - 1 - i] for i in range(B // 2)])
        pp = np.concatenate([np.minimum(p, q), B + np.minimum(p, q)]); qq = np.concatenate([np.maximum(p, q), B + np.maximum(p, q)])
        apply_set(S, Q, pp, qq)
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return Q

def outer_step(A, pairs, solver):
    n = A.shape[0]
    Qbig = np.zeros((n, n))
    for bi, bj in pairs:
        idx = np.concatenate([np.arange(bi * B, bi * B + B), np.arange(bj * B, bj * B + B)])
        S = A[np.ix_(idx, idx)].copy()
        Qbig[np.ix_(idx, idx)] = solver(S)
    return Qbig.T @ A @ Qbig

def rr_pairs(nblk, step):
    def rr(pos):
        if pos == 0: return 0
        v = pos - 1 + step
        if v >= nblk - 1: v -= nblk - 1
        return v + 1
    return [(rr(g), rr(nblk - 1 - g)) for g in range(nblk // 2)]

def greedy_pairs(A, nblk):
    W = np.zeros((nblk, nblk))
    for i in range(nblk):
        for j in range(i + 1, nblk):
            blk = A[i * B:(i + 1) * B, j * B:(j + 1) * B]
            d = np.sqrt(np.abs(np.outer(np.diag(A)[i * B:(i + 1) * B], np.diag(A)[j * B:(j + 1) * B])))
            W[i, j] = W[j, i] = np.sum((blk / d) ** 2)          # scaled off-block mass (squared cosines)
    free = set(range(nblk)); pairs = []
    order = np.dstack(np.unravel_index(np.argsort(-W, axis=None), W.shape))[0]
    for i, j in order:
        if i < j and i in free and j in free:
            pairs.append((i, j)); free -= {i, j}
    return pairs

def run(A0, mode, tol=1.5e-2, max_steps=150):
    A = A0.astype(np.float64).copy()
    nblk = A.shape[0] // B
    steps = 0; hist = []
    while steps < max_steps:
        # intra step once per nblk - 1 cross steps (as the library does)
        if steps % (nblk - 1) == 0:
            A = outer_step(A, [(2 * g, 2 * g + 1) for g in range(nblk // 2)], intra_problem)
        pairs = rr_pairs(nblk, steps % (nblk - 1)) if mode == 'cyclic' else greedy_pairs(A, nblk)
        A = outer_step(A, pairs, cross_problem)
        steps += 1
        if steps % 5 == 0 or mode == 'cyclic' and steps % (nblk - 1) == 0:
            r2 = strict_r2(A)
            if steps % (nblk - 1) == 0: hist.append('%d:%.1e' % (steps, r2))
            if r2 < tol * tol and (mode != 'cyclic' or steps % (nblk - 1) == 0):
                hist.append('%d:%.1e' % (steps, r2))
                break
    print('%-8s outer steps %3d (= %.1f sweeps of %d)  %s' % (mode, steps, steps / (nblk - 1.0), nblk - 1, ' '.join(hist)), flush=True)

z = np.load(sys.argv[1] if len(sys.argv) > 1 else '/tmp/wct_levels.npz')     # cache written by tests/probes/wct_tol_probe.py
for i, side in ((1, 'fc'), (0, 'fc'), (2, 'fs')):
    f = z['%s%d' % (side, i)]
    C = f.shape[-1]
    X = f.reshape(-1, C).astype(np.float64); X = X - X.mean(0)
    A = (X.T @ X) / (X.shape[0] - 1)
    print('level %d %s C=%d' % (i, side, C), flush=True)
    t0 = time.time()
    run(A, 'cyclic'); run(A, 'dynamic')
    print('  (%.0f s)' % (time.time() - t0), flush=True)

def full_problem(nsw):
    def f(S):
        Q = np.eye(2 * B)
        for _ in range(nsw):
            Q = Q @ intra_problem(S)
            Q = Q @ cross_problem(S)
        return Q
    return f

def run2(A0, solver, label, tol=1.5e-2, max_steps=150):
    A = A0.astype(np.float64).copy()
    nblk = A.shape[0] // B
    steps = 0; hist = []
    while steps < max_steps:
        A = outer_step(A, rr_pairs(nblk, steps % (nblk - 1)), solver)
        steps += 1
        if steps % (nblk - 1) == 0:
            r2 = strict_r2(A); hist.append('%d:%.1e' % (steps, r2))
            if r2 < tol * tol: break
    print('%-28s outer steps %3d  %s' % (label, steps, ' '.join(hist)), flush=True)

print('--- pair problems solved more thoroughly per outer step (cyclic pairing, no separate intra step)')
for i, side in ((1, 'fc'),):
    f = z['%s%d' % (side, i)]
    C = f.shape[-1]
    X = f.reshape(-1, C).astype(np.float64); X = X - X.mean(0)
    A = (X.T @ X) / (X.shape[0] - 1)
    run2(A, full_problem(1), 'intra+cross per pair')
    run2(A, full_problem(2), '2 x (intra+cross) per pair')
    run2(A, full_problem(3), '3 x (intra+cross) per pair')
