"""NumPy prototype: cyclic two-sided Jacobi sweep counts on the level covariances of a 512x512 frame, plain vs
Cholesky-preconditioned (A = L L^T with diagonal pivoting -> A' = L^T L, one step of the Cholesky-LR iteration)."""
import sys, time
import numpy as np

def rr_sets(n):
    """round-robin tournament: n-1 sets of n/2 disjoint pairs"""
    idx = list(range(n))
    sets = []
    for s in range(n - 1):
        p = np.array([idx[i] for i in range(n // 2)])
        q = np.array([idx[n - 1 - i] for i in range(n // 2)])
        sets.append((np.minimum(p, q), np.maximum(p, q)))
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return sets

def strict_r2(A, cut=1e-5):
    d = np.abs(np.diag(A)).astype(np.float64)
    k = d > cut
    E = A.astype(np.float64) - np.diag(np.diag(A).astype(np.float64))
    kk = np.outer(k, k)
    with np.errstate(divide='ignore', invalid='ignore'):
        cos2 = np.where(kk, E * E / np.outer(d, d), 0.0)
    # mixed pairs (one kept, one dropped): angle^2 + contamination
    big = np.maximum.outer(d, d)
    small = np.minimum.outer(d, d)
    mixed = np.where(np.logical_xor.outer(k, k), E * E / (big * big) + np.where(small < cut, 0.01 * E * E / (big * cut), 0.0), 0.0)
    return 0.5 * (cos2.sum() + mixed.sum()) / max(1, k.sum())

def sweep(A, V, sets, dtype):
    for p, q in sets:
        app, aqq, apq = A[p, p], A[q, q], A[p, q]
        rot = np.abs(apq) > 1e-6 * np.sqrt(np.abs(app * aqq)) 
        tau = 0.5 * (aqq - app)
        h = np.sqrt(tau * tau + apq * apq)
        with np.errstate(divide='ignore', invalid='ignore'):
            t = np.where(rot, np.abs(apq) / (np.abs(tau) + h), 0.0)
        t = np.where((tau >= 0) == (apq >= 0), t, -t)
        c = (1.0 / np.sqrt(1.0 + t * t)).astype(dtype)
        s = (c * t).astype(dtype)
        # columns
        Ap, Aq = A[:, p].copy(), A[:, q].copy()
        A[:, p] = c * Ap - s * Aq
        A[:, q] = s * Ap + c * Aq
        # rows
        Ap, Aq = A[p, :].copy(), A[q, :].copy()
        A[p, :] = c[:, None] * Ap - s[:, None] * Aq
        A[q, :] = s[:, None] * Ap + c[:, None] * Aq
        if V is not None:
            Vp, Vq = V[:, p].copy(), V[:, q].copy()
            V[:, p] = c * Vp - s * Vq
            V[:, q] = s * Vp + c * Vq

def run(A0, label, dtype=np.float32, max_sweeps=12, tol=1.5e-2):
    A = A0.astype(dtype).copy()
    n = A.shape[0]
    sets = rr_sets(n)
    out = []
    for sw in range(max_sweeps):
        sweep(A, None, sets, dtype)
        r2 = strict_r2(A)
        out.append(r2)
        if r2 < tol * tol:
            break
    print('%-34s sweeps %2d  r2 per sweep: %s' % (label, len(out), ' '.join('%.1e' % v for v in out)), flush=True)
    return len(out)

def pivoted_cholesky(A):
    A = A.astype(np.float64).copy()
    n = A.shape[0]
    perm = np.arange(n)
    L = np.zeros_like(A)
    d = np.diag(A).copy()
    for k in range(n):
        j = k + int(np.argmax(d[k:]))
        if j != k:
            perm[[k, j]] = perm[[j, k]]
            A[[k, j], :] = A[[j, k], :]; A[:, [k, j]] = A[:, [j, k]]
            L[[k, j], :] = L[[j, k], :]
            d[[k, j]] = d[[j, k]]
        piv = d[k]
        if piv <= 1e-12 * max(d[0], 1e-300):
            break
        L[k, k] = np.sqrt(piv)
        L[k + 1:, k] = (A[k + 1:, k] - L[k + 1:, :k] @ L[k, :k]) / L[k, k]
        d[k + 1:] -= L[k + 1:, k] ** 2
    return L, perm

if __name__ == '__main__':
    z = np.load(sys.argv[1] if len(sys.argv) > 1 else "/tmp/wct_levels.npz")     # level features: tests/probes/wct_tol_probe.py writes this cache
    for i in (0, 1, 2):
        for side in ('fc', 'fs'):
            f = z['%s%d' % (side, i)]
            C = f.shape[-1]
            X = f.reshape(-1, C).astype(np.float64)
            X = X - X.mean(0)
            A = (X.T @ X) / (X.shape[0] - 1)
            ev = np.linalg.eigvalsh(A)
            print('level %d %s: C=%d N=%d  eig max %.3e min %.3e  kept(>1e-5) %d' % (i, side, C, X.shape[0], ev[-1], ev[0], (ev > 1e-5).sum()), flush=True)
            t0 = time.time()
            run(A, 'plain two-sided Jacobi')
            d = np.argsort(-np.diag(A))
            run(A[np.ix_(d, d)], 'diagonal-sorted')
            L, perm = pivoted_cholesky(A)
            A1 = L.T @ L
            run(A1, 'Cholesky-LR x1 (L^T L)')
            L2, _ = pivoted_cholesky(A1)
            run(L2.T @ L2, 'Cholesky-LR x2')
            print('   (%.0f s)' % (time.time() - t0), flush=True)
