"""NumPy model of the DATA ROUTING of the look-ahead block-Jacobi launches (round 3, csrc/wct.hip jacobi_fused_kernel):

  launch L_s = { D(s): pair problems of outer step s,  U(s-1): tile update of outer step s-1 }

D(s) never waits for U(s-1): its 2B x 2B pair problem is assembled from
  * the diagonal blocks of the pair problems D(s-1) wrote (Sbuf: the rotated S images), and
  * ONE off-diagonal B x B block ("crit") it computes itself from the state BEFORE U(s-1) and the rotations Q(s-1):
      crit = Q_g1[:, h1]^T  P_old[tile g1, g2]  Q_g2[:, h2]
U(s-1) reads P_old, writes every tile into P_new (off-diagonal tiles g < h computed + mirrored, diagonal tiles copied
from Sbuf), updates V in place.  This script checks that the routed sequence reproduces the plain sequence
(gather the pair problem from the full matrix, rotate, apply to the whole matrix) to round-off.
"""
import numpy as np

B = 8
M2 = 2 * B


def rr_idx(pos, step, n):
    if pos == 0:
        return 0
    v = pos - 1 + step
    if v >= n - 1:
        v -= n - 1
    return v + 1


def block_pair(g, step, nblk):
    if step < 0:
        return 2 * g, 2 * g + 1
    return rr_idx(g, step, nblk), rr_idx(nblk - 1 - g, step, nblk)


def locate(b, step, nblk):
    """inverse of block_pair: (pair index, half) of block b at outer step `step`"""
    if step < 0:
        return b // 2, b & 1
    if b == 0:
        pos = 0
    else:
        pos = ((b - 1 - step) % (nblk - 1)) + 1
    npair = nblk // 2
    return (pos, 0) if pos < npair else (nblk - 1 - pos, 1)


def pidx(bi, bj):
    return np.concatenate([np.arange(bi * B, bi * B + B), np.arange(bj * B, bj * B + B)])


def solve_pair(S):
    """stand-in for the rotation sets: any orthogonal Q computed from the image (here: its eigenvectors, with a
    deterministic sign/ordering so that both sequences get the same Q from (nearly) the same image)"""
    w, Q = np.linalg.eigh(0.5 * (S + S.T))
    Q = Q * np.sign(Q[np.argmax(np.abs(Q), axis=0), np.arange(Q.shape[1])])
    return Q, Q.T @ S @ Q


def reference(A, V, steps, nblk):
    A = A.copy(); V = V.copy()
    for s in steps:
        Qbig = np.zeros_like(A)
        for g in range(nblk // 2):
            idx = pidx(*block_pair(g, s, nblk))
            Q, _ = solve_pair(A[np.ix_(idx, idx)])
            Qbig[np.ix_(idx, idx)] = Q
        A = Qbig.T @ A @ Qbig
        V = V @ Qbig
    return A, V


def routed(A, V, steps, nblk):
    """one segment: D(first) from the matrix, fused launches, U(last) alone"""
    npair = nblk // 2
    P = [A.copy(), np.full_like(A, np.nan)]
    V = V.copy()
    cur = 0
    Qb = [None, None]; Sb = [None, None]

    def D(s, prev, first, par):
        Qw = np.zeros((npair, M2, M2)); Sw = np.zeros((npair, M2, M2))
        for g in range(npair):
            bi, bj = block_pair(g, s, nblk)
            if first:
                idx = pidx(bi, bj)
                img = P[cur][np.ix_(idx, idx)].copy()
            else:
                Qr, Sr = Qb[par ^ 1], Sb[par ^ 1]
                g1, h1 = locate(bi, prev, nblk); g2, h2 = locate(bj, prev, nblk)
                assert block_pair(g1, prev, nblk)[h1] == bi and block_pair(g2, prev, nblk)[h2] == bj
                img = np.zeros((M2, M2))
                img[:B, :B] = Sr[g1][h1 * B:(h1 + 1) * B, h1 * B:(h1 + 1) * B]
                img[B:, B:] = Sr[g2][h2 * B:(h2 + 1) * B, h2 * B:(h2 + 1) * B]
                if g1 == g2:
                    img[:B, B:] = Sr[g1][h1 * B:(h1 + 1) * B, h2 * B:(h2 + 1) * B]
                else:
                    i1 = pidx(*block_pair(g1, prev, nblk)); i2 = pidx(*block_pair(g2, prev, nblk))
                    X = P[cur][np.ix_(i1, i2)]                       # state BEFORE U(prev)
                    img[:B, B:] = Qr[g1][:, h1 * B:(h1 + 1) * B].T @ X @ Qr[g2][:, h2 * B:(h2 + 1) * B]
                img[B:, :B] = img[:B, B:].T
            Qw[g], Sw[g] = solve_pair(img)
        Qb[par], Sb[par] = Qw, Sw

    def U(s, par):
        nonlocal cur, V
        Qr, Sr = Qb[par], Sb[par]
        old, new = P[cur], P[cur ^ 1]
        new[:] = np.nan
        for g in range(npair):
            ig = pidx(*block_pair(g, s, nblk))
            new[np.ix_(ig, ig)] = Sr[g]                              # diagonal tiles: copies of the rotated images
            for h in range(g + 1, npair):
                ih = pidx(*block_pair(h, s, nblk))
                Y = Qr[g].T @ old[np.ix_(ig, ih)] @ Qr[h]
                new[np.ix_(ig, ih)] = Y
                new[np.ix_(ih, ig)] = Y.T
            V[:, ig] = V[:, ig] @ Qr[g]
        assert not np.isnan(new).any()
        cur ^= 1

    for k, s in enumerate(steps):
        if k == 0:
            D(s, None, True, k & 1)
        else:
            D(s, steps[k - 1], False, k & 1)     # the real launch runs U(steps[k-1]) concurrently: D must not read P[new]
            U(steps[k - 1], (k - 1) & 1)
    U(steps[-1], (len(steps) - 1) & 1)
    return P[cur], V


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    for nblk in (4, 8, 16):
        C = nblk * B
        X = rng.standard_normal((3 * C, C)) * 10.0 ** rng.uniform(-1, 1, C)
        A = X.T @ X / (3 * C)
        V = np.eye(C)
        # a sweep = steps -1, 0 .. nblk-2; a segment may span sweeps (the step before -1 is nblk-2 of the previous sweep)
        sweep = [-1] + list(range(nblk - 1))
        for steps in (sweep, sweep + sweep, sweep[:nblk // 2], sweep[nblk // 2:] + sweep[:3]):
            A1, V1 = reference(A, V, steps, nblk)
            A2, V2 = routed(A, V, steps, nblk)
            ea = np.abs(A1 - A2).max() / np.abs(A1).max(); ev = np.abs(V1 - V2).max()
            print('nblk %2d steps %3d: max |dA|/|A| %.1e  max |dV| %.1e  |A - V^T A0 V| %.1e' %
                  (nblk, len(steps), ea, ev, np.abs(V2.T @ A @ V2 - A2).max() / np.abs(A).max()))
            assert ea < 1e-9 and ev < 1e-7
