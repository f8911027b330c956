"""Data routing of the register-resident pair problem (round 4), proven in NumPy before the HIP kernel is written.

A 64 x 64 pair problem (index blocks a = 0..31, b = 32..63; set s pairs a_k with b_{(k+s) % 32}) is held by G x G lanes,
G = 32 / P, lane (I, Dd) owning the P x P patch of CELLS (k, d), k = P I + i, d = P Dd + j, in SKEWED coordinates
l = (k + d + 1) % 32.  Cell (k, l) holds  S[p_k, p_l], S[p_k, q_l], S[q_k, p_l], S[q_k, q_l]  (p_x = x, q_x = 32 + (x + s) % 32)
and  Q[k, p_l], Q[k, q_l], Q[32 + k, p_l], Q[32 + k, q_l]  (Q gets column rotations only, so its rows never move).
After a set:  pq, Qpq, Qqq come from cell (k, d+1);  qq from (k+1, d);  qp from (k+1, d-1);  pp, Qpp, Qqp stay.
The simulation below executes exactly the per-lane program of the kernel (arrays over the lane index stand for
registers, `slots` / `cs` for the LDS exchange buffers) and is compared with the plain sequence of rotation sets on the
full matrices."""
import numpy as np

NP_ = 32


def rotation(app, aqq, apq):
    """symmetric Schur rotation annihilating apq (the solver's formula, no thresholds)"""
    if apq == 0.0:
        return 1.0, 0.0
    tau = 0.5 * (aqq - app)
    h = np.hypot(tau, apq)
    t = abs(apq) / (abs(tau) + h)
    if (tau >= 0) != (apq >= 0):
        t = -t
    c = 1.0 / np.sqrt(1 + t * t)
    return c, c * t


def reference(S0):
    """the sets applied to full matrices: S <- J^T S J, Q <- Q J, J rotating (p_k, q_k) by the angle that zeroes S[p_k, q_k]"""
    S = S0.copy()
    Q = np.eye(64)
    for s in range(NP_):
        J = np.eye(64)
        for k in range(NP_):
            p, q = k, 32 + (k + s) % 32
            c, sn = rotation(S[p, p], S[q, q], S[p, q])
            # column rotation as the kernel writes it: y_p = c a_p - s a_q, y_q = s a_p + c a_q
            J[p, p] = c; J[q, p] = -sn; J[p, q] = sn; J[q, q] = c
        S = J.T @ S @ J
        Q = Q @ J
    return S, Q


def simulate(S0, P):
    G = NP_ // P
    L = G * G
    t = np.arange(L)
    I, Dd = t % G, t // G
    # registers [i][j][lane]
    z = lambda: np.zeros((P, P, L))
    pp, pq, qp, qq, Qpp, Qpq, Qqp, Qqq = z(), z(), z(), z(), z(), z(), z(), z()
    kk = np.zeros((P, L), int)
    ll = np.zeros((P, P, L), int)
    for i in range(P):
        kk[i] = P * I + i
        for j in range(P):
            ll[i, j] = (kk[i] + P * Dd + j + 1) % 32
            k, l = kk[i], ll[i, j]
            pp[i, j] = S0[k, l]; pq[i, j] = S0[k, 32 + l]; qp[i, j] = S0[32 + k, l]; qq[i, j] = S0[32 + k, 32 + l]
            Qpp[i, j] = (k == l); Qqq[i, j] = (k == l)
    # pivot lanes: Dd == 0; they follow pair k's pivot block in closed form
    piv = Dd == 0
    ppk = np.zeros((P, L)); qqk = np.zeros((P, L)); pqk = np.zeros((P, L))
    cs = np.zeros((2, NP_, 2))
    for i in range(P):
        k = kk[i]
        ppk[i] = S0[k, k]; qqk[i] = S0[32 + k, 32 + k]; pqk[i] = S0[k, 32 + k]
    for lane in np.nonzero(piv)[0]:
        for i in range(P):
            cs[0, kk[i, lane]] = rotation(ppk[i, lane], qqk[i, lane], pqk[i, lane])
    lane_of = lambda Ii, Di: (Di % G) * G + (Ii % G)
    for s in range(NP_):
        cur, nx = s & 1, (s & 1) ^ 1
        # ---- rotate every cell
        npp, npq, nqp, nqq = z(), z(), z(), z()
        for i in range(P):
            ck, sk = cs[cur, kk[i], 0], cs[cur, kk[i], 1]
            for j in range(P):
                cl, sl = cs[cur, ll[i, j], 0], cs[cur, ll[i, j], 1]
                ypp = cl * pp[i, j] - sl * pq[i, j]; ypq = sl * pp[i, j] + cl * pq[i, j]
                yqp = cl * qp[i, j] - sl * qq[i, j]; yqq = sl * qp[i, j] + cl * qq[i, j]
                npp[i, j] = ck * ypp - sk * yqp; npq[i, j] = ck * ypq - sk * yqq
                nqp[i, j] = sk * ypp + ck * yqp; nqq[i, j] = sk * ypq + ck * yqq
                a, b = Qpp[i, j].copy(), Qpq[i, j].copy()
                Qpp[i, j] = cl * a - sl * b; Qpq[i, j] = sl * a + cl * b
                a, b = Qqp[i, j].copy(), Qqq[i, j].copy()
                Qqp[i, j] = cl * a - sl * b; Qqq[i, j] = sl * a + cl * b
        # ---- pivot lanes: next rotation of their P pairs (closed-form diagonals, partner diagonal from pair k+1)
        ppn = np.zeros((P, L)); qqn = np.zeros((P, L))
        for i in range(P):
            c, sn = cs[cur, kk[i], 0], cs[cur, kk[i], 1]
            ppn[i] = c * c * ppk[i] - 2 * c * sn * pqk[i] + sn * sn * qqk[i]
            qqn[i] = sn * sn * ppk[i] + 2 * c * sn * pqk[i] + c * c * qqk[i]
        for i in range(P):
            ppk[i] = ppn[i]
            # partner diagonal of the next set = the q diagonal pair k+1 just produced: same lane for i < P-1, lane I+1 else
            qqk[i] = qqn[i + 1] if i + 1 < P else qqn[0][lane_of(I + 1, Dd)]
            pqk[i] = npq[i, 0]                       # the fresh element of cell (k, l = k+1)
        for lane in np.nonzero(piv)[0]:
            for i in range(P):
                cs[nx, kk[i, lane]] = rotation(ppk[i, lane], qqk[i, lane], pqk[i, lane])
        # ---- exchange: outgoing values sit in per-lane slots, every lane reads its neighbours' slots
        right, down, up_left = lane_of(I, Dd + 1), lane_of(I + 1, Dd), lane_of(I + 1, Dd - 1)
        left = lane_of(I, Dd - 1)
        pp = npp
        n_pq, n_qp, n_qq, n_Qpq, n_Qqq = z(), z(), z(), z(), z()
        for i in range(P):
            for j in range(P):
                if j + 1 < P:
                    n_pq[i, j] = npq[i, j + 1]; n_Qpq[i, j] = Qpq[i, j + 1]; n_Qqq[i, j] = Qqq[i, j + 1]
                else:
                    n_pq[i, j] = npq[i, 0][right]; n_Qpq[i, j] = Qpq[i, 0][right]; n_Qqq[i, j] = Qqq[i, 0][right]
                n_qq[i, j] = nqq[i + 1, j] if i + 1 < P else nqq[0, j][down]
                if i + 1 < P:
                    n_qp[i, j] = nqp[i + 1, j - 1] if j >= 1 else nqp[i + 1, P - 1][left]
                else:
                    n_qp[i, j] = nqp[0, j - 1][down] if j >= 1 else nqp[0, P - 1][up_left]
        pq, qp, qq, Qpq, Qqq = n_pq, n_qp, n_qq, n_Qpq, n_Qqq
    # after 32 sets the arrangement is that of set 0 again
    S = np.zeros((64, 64)); Q = np.zeros((64, 64))
    for i in range(P):
        for j in range(P):
            k, l = kk[i], ll[i, j]
            S[k, l] = pp[i, j]; S[k, 32 + l] = pq[i, j]; S[32 + k, l] = qp[i, j]; S[32 + k, 32 + l] = qq[i, j]
            Q[k, l] = Qpp[i, j]; Q[k, 32 + l] = Qpq[i, j]; Q[32 + k, l] = Qqp[i, j]; Q[32 + k, 32 + l] = Qqq[i, j]
    return S, Q


if __name__ == '__main__':
    rng = np.random.default_rng(0)
    X = rng.standard_normal((200, 64)) * 10.0 ** (-np.arange(64) / 30.0)
    S0 = X.T @ X / 200
    Sr, Qr = reference(S0)
    off = lambda M: np.sqrt((M - np.diag(np.diag(M))) ** 2).sum()
    print('reference: off-diagonal mass of the cross blocks %.3e -> %.3e, |Q^T S0 Q - S| = %.2e' %
          (np.abs(S0[:32, 32:]).sum(), np.abs(Sr[:32, 32:]).sum(), np.abs(Qr.T @ S0 @ Qr - Sr).max()))
    for P in (1, 2, 4):
        S, Q = simulate(S0, P)
        print('P = %d: routed sequence vs plain sequence  |dS| = %.2e  |dQ| = %.2e' % (P, np.abs(S - Sr).max(), np.abs(Q - Qr).max()))
        assert np.abs(S - Sr).max() < 1e-12 and np.abs(Q - Qr).max() < 1e-12
