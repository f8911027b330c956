// The round-4 pair-problem code (wct_tf_amd/csrc/jacobi_dev.h, namespace r4) executed lane by lane on the CPU
// (hip_emul.h) and checked against plain sequential arithmetic in double precision:
//   1. cross step, first mode: S image and rotation matrix of every pair problem == the 32 rotation sets applied to the
//      full 64 x 64 matrices one after the other (the routing of tools/jacobi_patch_proto.py, now on the kernel source);
//   2. tile update U(s): P_new == J^T P J and V_new == V J with J assembled from the kernel's own rotation matrices;
//   3. look-ahead D(s+1): the pair problems assembled from {images of D(s), one MFMA block} == those loaded from P_new;
//   4. the fp16 hi + lo fragments reproduce Q to 2^-20;   5. the intra step (round 5: one wave per 32-wide block, odd-even
//      transposition ordering in registers, the off-diagonal block by two MFMA products): S image and Q == the sequential sets
//      in double, statistics, every index pair of a block met exactly once, Q orthogonal, S_out == Q^T S_in Q.
// Prints one line per check; exit status 0 iff all hold.
#include "hip_emul.h"
namespace emul { thread_local Idx tidx; thread_local Idx bidx; thread_local Block* blk; }
#include <stdio.h>
#include <stdlib.h>
#include "../../wct_tf_amd/csrc/jacobi_dev.h"
#ifndef EMUL_VAR
#define EMUL_VAR 0              // strip layout: 0 = 8 strips on 4 waves (the shipped one), 1 = 16 strips on 8 waves
#endif

static int failures = 0;
static void check(const char* what, double err, double tol) {
  printf("%-86s %.3e (tol %.1e) %s\n", what, err, tol, err <= tol ? "ok" : "FAIL");
  if (!(err <= tol)) ++failures;
}

constexpr int M2 = 64, B = 32, FR = M2 * M2;

// host view of the pairing schedule (block_pair)
static void pair_blocks(int g, int step, int nblk, int& bi, int& bj) { block_pair(g, step, nblk, bi, bj); }
static int pidx(int r, int bi, int bj) { return pair_index<B>(r, bi, bj); }

// rotation matrix out of the fragment-ordered store (qfrag_rc)
static void decode_q(const float* Qo, double* Q) {
  for (int f = 0; f < FR / 4; ++f) {
    int qr, qc;
    qfrag_rc<M2>(f, qr, qc);
    for (int j = 0; j < 4; ++j) Q[(qr + j) * M2 + qc] = Qo[f * 4 + j];
  }
}
static void decode_q16(const half_t* Q16, double* Q) {
  constexpr int NCH = M2 / 32;
  for (int u = 0; u < FR / 4; ++u) {
    const int l16 = u & 63, part = (u >> 6) & 1, cc = (u >> 7) % NCH, mt = (u >> 7) / NCH;
    for (int j = 0; j < 8; ++j) {
      const int k = qfrag16_k<M2>(cc, l16 >> 4, j), col = 16 * mt + (l16 & 15);
      if (part == 0) Q[k * M2 + col] = 0.0;
    }
  }
  for (int u = 0; u < FR / 4; ++u) {
    const int l16 = u & 63, cc = (u >> 7) % NCH, mt = (u >> 7) / NCH;
    for (int j = 0; j < 8; ++j) Q[qfrag16_k<M2>(cc, l16 >> 4, j) * M2 + 16 * mt + (l16 & 15)] += (double)(float)Q16[u * 8 + j];
  }
}

// the solver's rotation in double (thresholds of jacobi_rotation_cs never bind on these inputs)
static void rot(double app, double aqq, double apq, double& c, double& s) {
  if (apq == 0.0) { c = 1; s = 0; return; }
  const double tau = 0.5 * (aqq - app), h = sqrt(tau * tau + apq * apq);
  double t = fabs(apq) / (fabs(tau) + h);
  if ((tau >= 0) != (apq >= 0)) t = -t;
  c = 1.0 / sqrt(1 + t * t); s = c * t;
}
static double ref_offmax = 0;       // largest |a_pq| / sqrt(a_pp a_qq) over the pivots of sets 0 .. 32 of all pair problems
static void reference_cross(const double* S0, double* S, double* Q) {
  std::vector<double> J(FR), T(FR);
  for (int i = 0; i < FR; ++i) { S[i] = S0[i]; Q[i] = (i / M2 == i % M2); }
  for (int s = 0; s <= B; ++s) {
    for (int k = 0; k < B; ++k) {
      const int p = k, q = B + (k + s) % B;
      ref_offmax = std::max(ref_offmax, std::min(1.0, fabs(S[p * M2 + q]) / sqrt(fabs(S[p * M2 + p] * S[q * M2 + q]))));
    }
    if (s == B) break;
    for (int i = 0; i < FR; ++i) J[i] = (i / M2 == i % M2);
    for (int k = 0; k < B; ++k) {
      const int p = k, q = B + (k + s) % B;
      double c, sn;
      rot(S[p * M2 + p], S[q * M2 + q], S[p * M2 + q], c, sn);
      J[p * M2 + p] = c; J[q * M2 + p] = -sn; J[p * M2 + q] = sn; J[q * M2 + q] = c;
    }
    auto mm = [&](const double* X, const double* Y, double* Z, bool xt) {
      for (int i = 0; i < M2; ++i) for (int j = 0; j < M2; ++j) {
        double a = 0; for (int k = 0; k < M2; ++k) a += (xt ? X[k * M2 + i] : X[i * M2 + k]) * Y[k * M2 + j];
        Z[i * M2 + j] = a; }
    };
    mm(S, J.data(), T.data(), false); mm(J.data(), T.data(), S, true);
    mm(Q, J.data(), T.data(), false); memcpy(Q, T.data(), sizeof(double) * FR);
  }
}

// the intra step's ordering (r4::intra_wave): odd-even transposition with exchange on the positions of each 32-wide block -- even
// sets rotate the position pairs (0,1)(2,3).., odd sets (1,2)..(29,30); position p takes s x_p + c x_q, position q takes c x_p - s x_q
static double ref_offmax_intra = 0;
static int intra_pairs_seen = 0;
static void reference_intra(const double* S0, double* S, double* Q) {
  std::vector<double> G(FR), T(FR);
  std::vector<int> seen(FR, 0);
  int idx[2][B];
  for (int h = 0; h < 2; ++h) for (int i = 0; i < B; ++i) idx[h][i] = i;
  for (int i = 0; i < FR; ++i) { S[i] = S0[i]; Q[i] = (i / M2 == i % M2); }
  auto mm = [&](const double* X, const double* Y, double* Z, bool xt) {
    for (int i = 0; i < M2; ++i) for (int j = 0; j < M2; ++j) {
      double a = 0; for (int k = 0; k < M2; ++k) a += (xt ? X[k * M2 + i] : X[i * M2 + k]) * Y[k * M2 + j];
      Z[i * M2 + j] = a; }
  };
  for (int s = 0; s < B; ++s) {
    for (int i = 0; i < FR; ++i) G[i] = (i / M2 == i % M2);
    for (int h = 0; h < 2; ++h)
      for (int x = s & 1; x + 1 < B; x += 2) {
        const int p = h * B + x, q = p + 1;
        double c, sn;
        ref_offmax_intra = std::max(ref_offmax_intra, std::min(1.0, fabs(S[p * M2 + q]) / sqrt(fabs(S[p * M2 + p] * S[q * M2 + q]))));
        rot(S[p * M2 + p], S[q * M2 + q], S[p * M2 + q], c, sn);
        G[p * M2 + p] = sn; G[q * M2 + p] = c; G[p * M2 + q] = c; G[q * M2 + q] = -sn;
        const int ip = idx[h][x], iq = idx[h][x + 1];
        seen[std::min(ip, iq) * B + std::max(ip, iq)] += 1;
        idx[h][x] = iq; idx[h][x + 1] = ip;
      }
    mm(S, G.data(), T.data(), false); mm(G.data(), T.data(), S, true);
    mm(Q, G.data(), T.data(), false); memcpy(Q, T.data(), sizeof(double) * FR);
  }
  intra_pairs_seen = 0;
  for (int i = 0; i < B; ++i) for (int j = i + 1; j < B; ++j) intra_pairs_seen += seen[i * B + j] == 2;    // once in each of the two blocks
}

struct Solver {
  int C, nblk, npair;
  std::vector<float> P[2], V, Qlog[2], Sb[2];
  std::vector<half_t> Q16[2];
  JacobiState st;
  std::vector<float> lds;
  Solver(int C_) : C(C_), nblk(C_ / B), npair(C_ / B / 2) {
    for (int i = 0; i < 2; ++i) { P[i].assign((size_t)C * C, 0.f); Qlog[i].assign((size_t)npair * FR, 0.f); Sb[i].assign((size_t)npair * FR, 0.f); Q16[i].assign((size_t)npair * FR * 2, (half_t)0); }
    V.assign((size_t)C * C, 0.f);
    for (int i = 0; i < C; ++i) V[(size_t)i * C + i] = 1.f;
    memset(&st, 0, sizeof(st));
    st.floor = 0.f; st.seg_stop = 0x7fffffff;
  }
  // a block runs inside exactly the dynamic LDS its launch would get, guard words behind it
  static constexpr int GUARD = 4096;
  void arm(size_t bytes) {
    lds.assign(bytes / 4 + GUARD, __builtin_nanf(""));                 // LDS is not initialised on the device either
    for (int i = 0; i < GUARD; ++i) lds[bytes / 4 + i] = -12345.f;
  }
  void guard_ok(size_t bytes, const char* what) {
    for (int i = 0; i < GUARD; ++i) if (lds[bytes / 4 + i] != -12345.f) { printf("LDS overrun in %s: float %d behind %zu bytes\n", what, i, bytes); exit(2); }
  }
  JacobiFusedArgs args(int cur, int par, int lg_u, int lg_d, int step_d, int step_u, bool has_d, bool has_u, bool first, bool with_v) {
    JacobiFusedArgs a;
    a.Pr = P[cur].data(); a.Pw = P[cur ^ 1].data(); a.V = V.data();
    a.Qr = Qlog[lg_u].data(); a.Qw = Qlog[lg_d].data(); a.Qr16 = Q16[lg_u].data(); a.Qw16 = Q16[lg_d].data();
#ifdef EMUL_U_F16
    a.u_f16 = 1;                                   // the tile update on split fp16 (the batched transform path's)
#endif
    a.Sr = Sb[par].data(); a.Sw = Sb[par ^ 1].data();
    a.st = &st; a.C = C; a.nmat = 1; a.step_d = step_d; a.step_u = step_u; a.has_d = has_d; a.has_u = has_u; a.first = first;
    a.with_v = with_v; a.dbg = 0;
    return a;
  }
  void run_d(const JacobiFusedArgs& a) {
    const size_t bytes = r4::lds_bytes<M2, EMUL_VAR>(1, 0, a.first, a.step_d);
    for (int g = 0; g < npair; ++g) {
      arm(bytes);
      emul::run_block(r4::Lay<EMUL_VAR>::NTD, g, [&](int) { r4::fused_d<M2, EMUL_VAR>(a, 0, g, lds.data()); });
      guard_ok(bytes, "fused_d");
    }
  }
  void run_u(const JacobiFusedArgs& a, bool with_v) {
    const int ntask = npair * (npair - 1) / 2 + npair + (with_v ? npair * npair : 0);
    const size_t bytes = r4::lds_bytes<M2, EMUL_VAR>(0, 1, 0, 0);
    for (int task = 0; task < ntask; ++task) {
      arm(bytes);
      emul::run_block(r4::NT, task, [&](int) { r4::fused_u<M2>(a, 0, task, lds.data()); });
      guard_ok(bytes, "fused_u");
    }
  }
};

int main() {
  const int C = 256;                               // 8 blocks of 32, 4 pair problems
  Solver S(C);
  srand(7);
  {                                                // covariance of graded random features: symmetric positive definite
    const int N = 600;
    std::vector<double> X((size_t)N * C);
    for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) {
      double u = 0; for (int k = 0; k < 6; ++k) u += rand() / (double)RAND_MAX; u -= 3.0;
      X[(size_t)n * C + c] = (u + 0.3 * (n % 7 == c % 7)) * pow(10.0, -1.5 * c / C);
    }
    for (int i = 0; i < C; ++i) for (int j = i; j < C; ++j) {
      double a = 0; for (int n = 0; n < N; ++n) a += X[(size_t)n * C + i] * X[(size_t)n * C + j];
      S.P[0][(size_t)i * C + j] = S.P[0][(size_t)j * C + i] = (float)(a / N);
    }
  }
  const std::vector<float> A0 = S.P[0];
  const int step0 = 2;                             // a cross step in the middle of the schedule
  // ---- launch 1: D(step0), first of a segment
  JacobiFusedArgs a1 = S.args(0, 0, 0, 0, step0, step0 - 1, true, false, true, true);
  S.run_d(a1);
  std::vector<std::vector<double>> Qk(S.npair, std::vector<double>(FR));
  {
    double eS = 0, eQ = 0, e16 = 0, red = 0;
    for (int g = 0; g < S.npair; ++g) {
      int bi, bj;
      pair_blocks(g, step0, S.nblk, bi, bj);
      std::vector<double> S0(FR), Sr(FR), Qr(FR), Q16(FR);
      for (int r = 0; r < M2; ++r) for (int c = 0; c < M2; ++c) S0[r * M2 + c] = A0[(size_t)pidx(r, bi, bj) * C + pidx(c, bi, bj)];
      reference_cross(S0.data(), Sr.data(), Qr.data());
      decode_q(a1.Qw + (size_t)g * FR, Qk[g].data());
      decode_q16(a1.Qw16 + (size_t)g * 2 * FR, Q16.data());
      double nrm = 0, off0 = 0, off1 = 0;
      for (int i = 0; i < FR; ++i) nrm = std::max(nrm, fabs(Sr[i]));
      for (int i = 0; i < FR; ++i) {
        eS = std::max(eS, fabs(a1.Sw[(size_t)g * FR + i] - Sr[i]) / nrm);
        eQ = std::max(eQ, fabs(Qk[g][i] - Qr[i]));
        e16 = std::max(e16, fabs(Q16[i] - Qk[g][i]));
      }
      for (int r = 0; r < B; ++r) for (int c = B; c < M2; ++c) { off0 += S0[r * M2 + c] * S0[r * M2 + c]; off1 += Sr[r * M2 + c] * Sr[r * M2 + c]; }
      red = std::max(red, sqrt(off1 / off0));
    }
    check("cross step (first mode): S image vs the sequential sets, relative to max |S|", eS, 2e-5);
    check("cross step (first mode): rotation matrix Q vs the sequential sets", eQ, 2e-5);
    check("fp16 hi + lo fragments vs the fp32 rotation matrix", e16, 2e-6);
    check("(reference) cross-block mass after / before one cross step", red, 0.7);
    const double offmax = __builtin_bit_cast(float, S.st.offmax), offsig = __builtin_bit_cast(float, S.st.offsig);
    check("convergence statistics: largest relative pivot seen by the sets (offmax) vs the sequential sets", fabs(offmax - ref_offmax) / ref_offmax, 1e-4);
    check("convergence statistics: the same over the significant pairs (floor 0: all of them)", fabs(offsig - ref_offmax) / ref_offmax, 1e-4);
  }
  // ---- launch 2: { D(step0 + 1) by look-ahead, U(step0) with V }
  JacobiFusedArgs a2 = S.args(0, 1, 0, 1, step0 + 1, step0, true, true, false, true);
  S.run_d(a2);                                     // reads P[0] (before U), the images and rotations of launch 1
  S.run_u(a2, true);                               // P[1] = J^T P[0] J, V = V J
  {
    std::vector<double> J((size_t)C * C, 0.0);
    for (int g = 0; g < S.npair; ++g) {
      int bi, bj;
      pair_blocks(g, step0, S.nblk, bi, bj);
      for (int r = 0; r < M2; ++r) for (int c = 0; c < M2; ++c) J[(size_t)pidx(r, bi, bj) * C + pidx(c, bi, bj)] = Qk[g][r * M2 + c];
    }
    std::vector<double> T((size_t)C * C), R((size_t)C * C);
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { double a = 0; for (int k = 0; k < C; ++k) a += (double)A0[(size_t)i * C + k] * J[(size_t)k * C + j]; T[(size_t)i * C + j] = a; }
    for (int i = 0; i < C; ++i) for (int j = 0; j < C; ++j) { double a = 0; for (int k = 0; k < C; ++k) a += J[(size_t)k * C + i] * T[(size_t)k * C + j]; R[(size_t)i * C + j] = a; }
    double nrm = 0, eP = 0, eV = 0;
    for (size_t i = 0; i < (size_t)C * C; ++i) nrm = std::max(nrm, fabs(R[i]));
    for (size_t i = 0; i < (size_t)C * C; ++i) { eP = std::max(eP, fabs(S.P[1][i] - R[i]) / nrm); eV = std::max(eV, fabs(S.V[i] - J[i])); }
    check("tile update: P_new vs J^T P J, relative to max |P|", eP, 5e-6);
    check("tile update: V_new vs V J (V = I before)", eV, 2e-6);
  }
  {
    // the same pair problems loaded from the updated matrix (first mode) must give what the look-ahead blocks gave
    std::vector<float> Sw_la(a2.Sw, a2.Sw + (size_t)S.npair * FR), Qw_la(a2.Qw, a2.Qw + (size_t)S.npair * FR);
    JacobiFusedArgs a3 = S.args(1, 1, 0, 1, step0 + 1, step0, true, false, true, true);
    S.run_d(a3);
    double eS = 0, eQ = 0, nrm = 0;
    for (size_t i = 0; i < (size_t)S.npair * FR; ++i) nrm = std::max(nrm, (double)fabs(a3.Sw[i]));
    for (size_t i = 0; i < (size_t)S.npair * FR; ++i) { eS = std::max(eS, fabs(a3.Sw[i] - Sw_la[i]) / nrm); eQ = std::max(eQ, (double)fabs(a3.Qw[i] - Qw_la[i])); }
    check("look-ahead pair problems vs the same problems loaded from the updated matrix: S", eS, 5e-5);
    check("look-ahead pair problems vs the same problems loaded from the updated matrix: Q", eQ, 5e-4);
  }
  {
    // ---- intra step (step -1), first mode, on the original matrix
    Solver T(C);
    T.P[0] = A0;
    JacobiFusedArgs ai = T.args(0, 0, 0, 0, -1, -2, true, false, true, true);
    T.run_d(ai);
    double eO = 0, eS = 0, dom = 0, eSr = 0, eQr = 0;
    for (int g = 0; g < T.npair; ++g) {
      int bi, bj;
      pair_blocks(g, -1, T.nblk, bi, bj);
      std::vector<double> Q(FR), S0(FR), Tm(FR), R(FR);
      decode_q(ai.Qw + (size_t)g * FR, Q.data());
      for (int r = 0; r < M2; ++r) for (int c = 0; c < M2; ++c) S0[r * M2 + c] = A0[(size_t)pidx(r, bi, bj) * C + pidx(c, bi, bj)];
      for (int i = 0; i < M2; ++i) for (int j = 0; j < M2; ++j) { double a = 0; for (int k = 0; k < M2; ++k) a += Q[k * M2 + i] * Q[k * M2 + j]; eO = std::max(eO, fabs(a - (i == j))); }
      for (int i = 0; i < M2; ++i) for (int j = 0; j < M2; ++j) { double a = 0; for (int k = 0; k < M2; ++k) a += S0[i * M2 + k] * Q[k * M2 + j]; Tm[i * M2 + j] = a; }
      double nrm = 0;
      for (int i = 0; i < M2; ++i) for (int j = 0; j < M2; ++j) { double a = 0; for (int k = 0; k < M2; ++k) a += Q[k * M2 + i] * Tm[k * M2 + j]; R[i * M2 + j] = a; nrm = std::max(nrm, fabs(a)); }
      double off0 = 0, off1 = 0;
      for (int i = 0; i < FR; ++i) eS = std::max(eS, fabs(ai.Sw[(size_t)g * FR + i] - R[i]) / nrm);
      for (int h = 0; h < 2; ++h) for (int r = 0; r < B; ++r) for (int c = 0; c < B; ++c) if (r != c) {
        off0 += S0[(h * B + r) * M2 + h * B + c] * S0[(h * B + r) * M2 + h * B + c];
        off1 += R[(h * B + r) * M2 + h * B + c] * R[(h * B + r) * M2 + h * B + c]; }
      dom = std::max(dom, sqrt(off1 / off0));
      std::vector<double> Sr(FR), Qr(FR);
      reference_intra(S0.data(), Sr.data(), Qr.data());
      for (int i = 0; i < FR; ++i) { eSr = std::max(eSr, fabs(ai.Sw[(size_t)g * FR + i] - Sr[i]) / nrm); eQr = std::max(eQr, fabs(Q[i] - Qr[i])); }
    }
    check("intra step: every index pair of a block meets exactly once in the 32 sets (pairs missing of 496)", 496 - intra_pairs_seen, 0);
    check("intra step: S image vs the sequential odd-even sets, relative to max |S|", eSr, 2e-5);
    check("intra step: rotation matrix Q vs the sequential odd-even sets", eQr, 2e-5);
    {
      const double offmax = __builtin_bit_cast(float, T.st.offmax), offsig = __builtin_bit_cast(float, T.st.offsig);
      check("intra step: largest relative pivot seen by the sets (offmax) vs the sequential sets", fabs(offmax - ref_offmax_intra) / ref_offmax_intra, 1e-4);
      check("intra step: the same over the significant pairs (floor 0: all of them)", fabs(offsig - ref_offmax_intra) / ref_offmax_intra, 1e-4);
    }
    check("intra step: Q^T Q - I", eO, 1e-5);
    check("intra step: S image vs Q^T S_in Q, relative", eS, 2e-5);
    check("intra step: off-diagonal mass inside the two blocks after / before", dom, 0.7);
  }
  printf(failures ? "FAILED (%d)\n" : "all checks passed\n", failures);
  return failures ? 1 : 0;
}
