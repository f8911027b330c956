// Device-side pieces of the batched block-Jacobi eigensolver (K5) that are shared by the kernels of wct.hip and by the
// CPU lane-emulation test (tests/emul): the solver's state words, the pairing schedule, the rotation, the generic
// LDS-image rotation sets, the argument block of the look-ahead launches and the round-4 register-resident pair problem
// (namespace r4).  Device code only: nothing here touches the HIP runtime API.  The includer provides the vector typedefs
// of common.h (wct.hip includes that first; the emulation brings its own prelude, tests/emul/hip_emul.h).
#pragma once
#include <type_traits>
#include <utility>
#ifndef R4_OPAQUE
#define R4_OPAQUE(x) asm volatile("" : "+v"(x))     // the optimiser may not look through the value (no instruction emitted)
#endif
#ifndef JTS
#define JTS(slot) do {} while (0)       // phase stamps of the -DJACOBI_TS build (wct.hip defines the real one)
#endif

struct JacobiState {
  unsigned int offmax;   // max |a_pq|/sqrt(a_pp a_qq) over the pairs rotated this sweep (float bits)
  int done;              // 0 rotating, 1 converged, 2 failed: non-finite input
  int sweeps;
  unsigned int offsig;   // the same maximum over the SIGNIFICANT pairs only (a diagonal above `floor`)
  float floor;           // 1e-4 max|a_ii| of the previous sweep (-> 1e-4 lambda_max): diagonals below it are taken to belong
                         // to the numerical null space of a matrix of this norm (~1700 eps ||A||) when the STATUS is judged
  unsigned int last_sig; // offsig of the last completed sweep (what jacobi_finalize_kernel judges)
  unsigned int dmax;     // max |a_ii| seen by this sweep's pair problems (float bits)
  float r2;              // strict residual measure of the last completed sweep (jacobi_resid_kernel), -1 before the first
  float r2l;             // the lenient one (what the final status is judged by)
  int pad;               // look-ahead path: the buffer (0: A, 1: the second one) that held the matrix when it was declared done
  int seg_stop;          // number of launch segments whose rotations belong to this matrix (INT_MAX while it is still rotating)
  int pad2;              // 1: a diagonal within half a decade of the 1e-5 cut-off at the last residual measurement (first-order completion only)
};
constexpr float JACOBI_SIG_FLOOR = 1e-4f;

__device__ __forceinline__ int rr_idx(int pos, int step, int n) {
  // circle method: position 0 is fixed, the other n-1 rotate
  if (pos == 0) return 0;
  int v = pos - 1 + step;
  if (v >= n - 1) v -= n - 1;
  return v + 1;
}

// blocks (bi, bj) of pair g at outer step `step`; step < 0 is the intra step: neighbours (2g, 2g+1)
__device__ __forceinline__ void block_pair(int g, int step, int nblk, int& bi, int& bj) {
  if (step < 0) { bi = 2 * g; bj = 2 * g + 1; }
  else { bi = rr_idx(g, step, nblk); bj = rr_idx(nblk - 1 - g, step, nblk); }
}

template <int B>
__device__ __forceinline__ int pair_index(int r, int bi, int bj) {
  return r < B ? bi * B + r : bj * B + (r - B);
}

constexpr float JACOBI_ROT_TOL = 1e-6f;    // skip rotations below this relative size
constexpr float JACOBI_CONV_TOL = 1e-2f;   // a sweep that never saw more than this is the last one (quadratic convergence;
                                           // measured: WCT error vs the oracle identical for 2e-3 and 1e-2, 100x worse at 5e-2)
constexpr float JACOBI_FLOOR = 1e-6f;      // both diagonals below this: the pair cannot reach the 1e-5 cut-off

// rotation (c, s) that annihilates a_pq; `off` = pre-rotation relative size (0 if skipped).
// t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (a_qq - a_pp) / (2 a_pq), written as
// t = +-|a_pq| / (|tau| + sqrt(tau^2 + a_pq^2)), tau = (a_qq - a_pp)/2: three transcendentals
// on the dependent chain (sqrt, rcp, rsq) and no division by a_pq.
// `sig` = the same measure if the pair is significant (its larger diagonal is above the matrix' noise floor), else 0:
// pairs inside the numerical null space keep relative off-diagonals of O(1) for ever (every update regenerates
// rounding noise there) without mattering for f(A); they are still rotated, but they do not count as "not converged".
// The same in two parts for the pivot wave of jacobi_cross_sets_pw: the rotation itself is on the serial chain of a set
// (it gates every other wave at the barrier), the convergence statistics are not -- they are evaluated after (c, s) has
// been published, under the latency of that LDS write.
__device__ __forceinline__ bool jacobi_rotation_cs(float app, float aqq, float apq, float& c, float& s) {
  const float den2 = fabsf(app * aqq);
  const float aapq = fabsf(apq);
  const float big = fmaxf(fabsf(app), fabsf(aqq)), small = fminf(fabsf(app), fabsf(aqq));
  const bool live = (fabsf(app) + fabsf(aqq) > JACOBI_FLOOR) & !((small < JACOBI_FLOOR) & (aapq < 1e-6f * big));
  const bool rot = live & (aapq * aapq > JACOBI_ROT_TOL * JACOBI_ROT_TOL * den2) & (aapq > 1e-36f);
  const float tau = 0.5f * (aqq - app);
  const float h = __builtin_amdgcn_sqrtf(tau * tau + apq * apq);
  float t = aapq * __builtin_amdgcn_rcpf(rot ? fabsf(tau) + h : 1.f);
  t = (tau >= 0.f) == (apq >= 0.f) ? t : -t;
  const float n2 = 1.f + t * t;
  float r = __builtin_amdgcn_rsqf(n2);
  r = r * (1.5f - 0.5f * n2 * r * r);            // one Newton step: c^2 + s^2 = 1 to fp32 round-off
  c = rot ? r : 1.f;
  s = rot ? r * t : 0.f;
  return rot;
}
// the same with tan(theta) beside (c, s), for the scaled rotations of the register-resident pair problems (r4::strip_sets):
// t is what the chain of a rotation set waits for (it is published as soon as it exists), (c, s) only feed the pivot lane's
// own closed-form diagonals of the next set
__device__ __forceinline__ bool jacobi_rotation_cst(float app, float aqq, float apq, float& c, float& s, float& tt) {
  const float den2 = fabsf(app * aqq);
  const float aapq = fabsf(apq);
  const float big = fmaxf(fabsf(app), fabsf(aqq)), small = fminf(fabsf(app), fabsf(aqq));
  const bool live = (fabsf(app) + fabsf(aqq) > JACOBI_FLOOR) & !((small < JACOBI_FLOOR) & (aapq < 1e-6f * big));
  const bool rot = live & (aapq * aapq > JACOBI_ROT_TOL * JACOBI_ROT_TOL * den2) & (aapq > 1e-36f);
  const float tau = 0.5f * (aqq - app);
  const float h = __builtin_amdgcn_sqrtf(tau * tau + apq * apq);
  float t = aapq * __builtin_amdgcn_rcpf(rot ? fabsf(tau) + h : 1.f);
  t = (tau >= 0.f) == (apq >= 0.f) ? t : -t;
  tt = rot ? t : 0.f;
  const float n2 = 1.f + t * t;
  float r = __builtin_amdgcn_rsqf(n2);
  r = r * (1.5f - 0.5f * n2 * r * r);            // one Newton step: c^2 + s^2 = 1 to fp32 round-off
  c = rot ? r : 1.f;
  s = rot ? r * t : 0.f;
  return rot;
}
// Round 5: only tan(theta) -- what the chain of a rotation set waits for.  (c, s) follow from it (jacobi_cs_from_t) and feed
// nothing but the pivot lane's own closed-form diagonals of the NEXT set, so r4::strip_sets derives them after the barrier,
// under the latency of that set's first LDS reads, instead of before it with every other wave waiting.
__device__ __forceinline__ bool jacobi_rotation_t(float app, float aqq, float apq, float& tt) {
  const float den2 = fabsf(app * aqq);
  const float aapq = fabsf(apq);
  const float big = fmaxf(fabsf(app), fabsf(aqq)), small = fminf(fabsf(app), fabsf(aqq));
  const bool live = (fabsf(app) + fabsf(aqq) > JACOBI_FLOOR) & !((small < JACOBI_FLOOR) & (aapq < 1e-6f * big));
  const bool rot = live & (aapq * aapq > JACOBI_ROT_TOL * JACOBI_ROT_TOL * den2) & (aapq > 1e-36f);
  const float tau = 0.5f * (aqq - app);
  const float h = __builtin_amdgcn_sqrtf(tau * tau + apq * apq);
  float t = aapq * __builtin_amdgcn_rcpf(rot ? fabsf(tau) + h : 1.f);
  t = (tau >= 0.f) == (apq >= 0.f) ? t : -t;
  tt = rot ? t : 0.f;
  return rot;
}
__device__ __forceinline__ void jacobi_cs_from_t(float tt, float& c, float& s) {
  const float n2 = 1.f + tt * tt;
  float r = __builtin_amdgcn_rsqf(n2);
  r = r * (1.5f - 0.5f * n2 * r * r);            // one Newton step: c^2 + s^2 = 1 to fp32 round-off
  c = tt != 0.f ? r : 1.f;                       // (no rotation: exactly the identity)
  s = r * tt;
}
__device__ __forceinline__ void jacobi_rotation_stats(float app, float aqq, float apq, float floor_m, bool rot, float& off, float& sig) {
  const float den2 = fabsf(app * aqq);
  const float aapq = fabsf(apq);
  const float big = fmaxf(fabsf(app), fabsf(aqq)), small = fminf(fabsf(app), fabsf(aqq));
  const float rel = den2 > 0.f ? fminf(aapq * __builtin_amdgcn_rsqf(den2), 1.f) : 1.f;
  off = rot ? rel : 0.f;
  // branch-free (selects, not exec-mask branches: the pivot wave is the pole of every set)
  const float mixed = fmaxf(aapq * __builtin_amdgcn_rcpf(big), small < 1e-5f ? 0.1f * aapq * __builtin_amdgcn_rsqf(big * 1e-5f) : 0.f);
  const float sig_mixed = big > floor_m ? fminf(off, mixed) : 0.f;
  sig = small > floor_m ? off : sig_mixed;
}

__device__ __forceinline__ void jacobi_rotation(float app, float aqq, float apq, float floor_m, float& c, float& s, float& off, float& sig) {
  const float den2 = fabsf(app * aqq);
  const float aapq = fabsf(apq);
  const float big = fmaxf(fabsf(app), fabsf(aqq)), small = fminf(fabsf(app), fabsf(aqq));
  // (a) both diagonals far below the 1e-5 cut-off: whatever they mix stays dropped;
  // (b) coupling of a kept direction into a noise-level one with a negligible angle
  // bitwise & / | on the predicates: && would become an exec-mask branch around the second half
  const bool live = (fabsf(app) + fabsf(aqq) > JACOBI_FLOOR) & !((small < JACOBI_FLOOR) & (aapq < 1e-6f * big));
  const bool rot = live & (aapq * aapq > JACOBI_ROT_TOL * JACOBI_ROT_TOL * den2) & (aapq > 1e-36f);
  const float rel = den2 > 0.f ? fminf(aapq * __builtin_amdgcn_rsqf(den2), 1.f) : 1.f;
  const float tau = 0.5f * (aqq - app);
  const float h = __builtin_amdgcn_sqrtf(tau * tau + apq * apq);
  float t = aapq * __builtin_amdgcn_rcpf(rot ? fabsf(tau) + h : 1.f);
  t = (tau >= 0.f) == (apq >= 0.f) ? t : -t;
  const float n2 = 1.f + t * t;
  float r = __builtin_amdgcn_rsqf(n2);
  r = r * (1.5f - 0.5f * n2 * r * r);            // one Newton step: c^2 + s^2 = 1 to fp32 round-off
  c = rot ? r : 1.f;
  s = rot ? r * t : 0.f;
  off = rot ? rel : 0.f;
  // both diagonals significant: the cosine; one significant, the other inside the rounding noise (a cosine against
  // it cannot settle): the rotation angle a_pq / big, and -- if the small one is below the reference's 1e-5 cut-off --
  // that it stays there: contamination a_pq^2 / big below 1 % of the cut-off; none significant: does not count
  const float mixed = fmaxf(aapq * __builtin_amdgcn_rcpf(big), small < 1e-5f ? 0.1f * aapq * __builtin_amdgcn_rsqf(big * 1e-5f) : 0.f);
  sig = small > floor_m ? off : (big > floor_m ? fminf(off, mixed) : 0.f);
}

// Rotation sets on an N x N symmetric pair problem (N = 32 or 64: blocks I = 0..N/2-1 and
// J = N/2..N-1) held in LDS, by (N/2)^2 threads t = (k, l), N/2 disjoint pairs per set.
// S and the accumulated rotations Q are interleaved as float2 {S[r][c], Q[r][c]} so thread
// (k, l) moves its 2x2 block of both with four 8-byte LDS reads and writes (rows {p_k, q_k} x
// columns {p_l, q_l}).  Two LDS images ping-pong, so a rotation set costs ONE barrier.  Each
// thread derives rotation(l) from three more reads; rotation(k) is fetched from the lane of
// its own wave that has l == k (ds_bpermute: no LDS round trip, no serial section).
//   SWEEP_CROSS  the (N/2)^2 pairs (i in I, j in J), N/2 sets     -- one outer step
//   SWEEP_INTRA  the pairs inside I and inside J, N/2-1 sets      -- once per outer sweep
// so that one outer sweep visits every pair of the C indices exactly once (a true cyclic
// Jacobi sweep).  All threads of the block call this together (contains __syncthreads()).
// Returns the index of the image that holds the result.
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { SWEEP_CROSS = 1, SWEEP_INTRA = 2 };
template <int MODE, int N>
__device__ __forceinline__ void sweep_pair(int j, int s, int& p, int& q) {
  constexpr int NP = N / 2;
  if (MODE == SWEEP_CROSS) { p = j; q = NP + ((j + s) & (NP - 1)); }
  else { const int base = j & (NP / 2) ? NP : 0, jj = j & (NP / 2 - 1); p = base + rr_idx(jj, s, NP); q = base + rr_idx(NP - 1 - jj, s, NP); }
}
//
// In CROSS mode the three values every thread needs for rotation(l) -- S[p][p], S[q][q], S[p][q] --
// are mirrored in two small compact arrays (DO: diagonal D[2][N] and pair elements O[2][N/2]), kept
// current by the threads that own them; reading them straight out of the float2 image costs three
// 4-way bank-conflicted loads per thread and set (the diagonal has stride 2(N+2) dwords).
// KB = 2x2 blocks per thread (rows k, k + NP/KB, ...; one column pair l): rotation(l) is derived once
// per thread and reused for its KB blocks.
template <int MODE, int N, int KB>
__device__ __forceinline__ int jacobi_sets(f32x2* SQ, float* DO, int t, float floor_m, float& my_off, float& my_sig) {
  constexpr int NP = N / 2, PITCH = N + 1, IMG = N * PITCH, KS = NP / KB;
  constexpr int NSETS = MODE == SWEEP_CROSS ? NP : NP - 1;
  const int kq = t / NP, l = t % NP;
  float* const Dg = DO;                          // [2][N]
  float* const Og = DO + 2 * N;                  // [2][NP]
  if (MODE == SWEEP_CROSS) {
    for (int i = t; i < N; i += NP * KS) Dg[i] = SQ[i * PITCH + i][0];
    for (int i = t; i < NP; i += NP * KS) Og[i] = SQ[i * PITCH + NP + i][0];      // set 0 pairs j with NP + j
    __syncthreads();
  }
  int cur = 0;
  for (int s = 0; s < NSETS; ++s) {
    int pl, ql;
    sweep_pair<MODE, N>(l, s, pl, ql);
    const f32x2* C0 = SQ + cur * IMG;
    f32x2* N0 = SQ + (cur ^ 1) * IMG;
    // every LDS read of the set is issued before anything depends on it
    float lpp, lqq, lpq;
    if (MODE == SWEEP_CROSS) { lpp = Dg[cur * N + pl]; lqq = Dg[cur * N + ql]; lpq = Og[cur * NP + l]; }
    else { lpp = C0[pl * PITCH + pl][0]; lqq = C0[ql * PITCH + ql][0]; lpq = C0[pl * PITCH + ql][0]; }
    int pk[KB], qk[KB];
    f32x2 app[KB], apq[KB], aqp[KB], aqq[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      sweep_pair<MODE, N>(kq + i * KS, s, pk[i], qk[i]);
      app[i] = C0[pk[i] * PITCH + pl]; apq[i] = C0[pk[i] * PITCH + ql];
      aqp[i] = C0[qk[i] * PITCH + pl]; aqq[i] = C0[qk[i] * PITCH + ql];
    }
    float cl, sl, offl, sigl;
    jacobi_rotation(lpp, lqq, lpq, floor_m, cl, sl, offl, sigl);
    my_off = fmaxf(my_off, offl);
    my_sig = fmaxf(my_sig, sigl);
    const int nx = cur ^ 1;
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int k = kq + i * KS;
      const int src_lane = (t & (64 - NP)) | k;      // lane of my wave whose l equals this k
      const float ck = __shfl(cl, src_lane, 64), sk = __shfl(sl, src_lane, 64);
      // S: columns (pair l), then rows (pair k);  Q: columns only
      const float ypp = cl * app[i][0] - sl * apq[i][0], ypq = sl * app[i][0] + cl * apq[i][0];
      const float yqp = cl * aqp[i][0] - sl * aqq[i][0], yqq = sl * aqp[i][0] + cl * aqq[i][0];
      f32x2 npp, npq, nqp, nqq;
      npp[0] = ck * ypp - sk * yqp;  npq[0] = ck * ypq - sk * yqq;
      nqp[0] = sk * ypp + ck * yqp;  nqq[0] = sk * ypq + ck * yqq;
      npp[1] = cl * app[i][1] - sl * apq[i][1];  npq[1] = sl * app[i][1] + cl * apq[i][1];
      nqp[1] = cl * aqp[i][1] - sl * aqq[i][1];  nqq[1] = sl * aqp[i][1] + cl * aqq[i][1];
      N0[pk[i] * PITCH + pl] = npp;  N0[pk[i] * PITCH + ql] = npq;
      N0[qk[i] * PITCH + pl] = nqp;  N0[qk[i] * PITCH + ql] = nqq;
      if (MODE == SWEEP_CROSS) {
        if (k == l) { Dg[nx * N + pk[i]] = npp[0]; Dg[nx * N + qk[i]] = nqq[0]; }
        if (l == ((k + 1) & (NP - 1))) Og[nx * NP + k] = npq[0];     // S[p_k][q_k] of the next set
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  return cur;
}

struct JacobiFusedArgs {
  const float* Pr;      // [nmat][C][C] state the U part reads and the D part takes its look-ahead block from
  float* Pw;            // [nmat][C][C] state the U part writes
  float* V;             // [nmat][C][C] eigenvector accumulation, in place
  const float* Qr; const float* Sr;   // [nmat][npair][M2*M2] rotations / rotated pair problems of step_u
  float* Qw; float* Sw;               // ... written by the D part (step_d)
  const half_t* Qr16; half_t* Qw16;   // the same rotations split into fp16 hi + lo MFMA fragments (V <- V Q; see qfrag16)
  int u_f16 = 0;                      // tile update of the 64-wide block pairs on split fp16 (r4::fused_u) instead of fp32 MFMA
  JacobiState* st;
  int C, nmat;
  int step_d, step_u;   // outer step of the pair problems / of the tile update (= the step before step_d)
  int has_d, has_u;
  int first;            // the D part loads its pair problems straight from Pr (nothing is pending on it)
  int with_v;           // the U part also updates V (otherwise jacobi_vstrip_kernel applies the segment's rotations later)
  int dbg;              // timing experiments (WCT_JACOBI_DBG): 1 U blocks exit at once, 2 D blocks exit at once, 4 no rotation sets
  int mat_major;        // pair-problem blocks are numbered matrix-fastest (XCD locality; jacobi_fused4_kernel)
};

// Rotation matrices are stored in FRAGMENT order (the A operand of v_mfma_f32_16x16x4_f32 for V Q, see
// jacobi_vstrip_kernel): float4 f = (mt * (M2/16) + t) * 64 + lane holds Q[16 t + 4 (lane >> 4) + r][16 mt + (lane & 15)],
// r = 0..3.  Every consumer stages whole tiles, so the order costs the others nothing.
template <int M2>
__device__ __forceinline__ void qfrag_rc(int f, int& row, int& col) {
  constexpr int NTL = M2 / 16;
  const int l = f & 63, t = (f >> 6) % NTL, mt = (f >> 6) / NTL;
  row = 16 * t + 4 * (l >> 4); col = 16 * mt + (l & 15);
}

// inverse of block_pair: pair index and half (0: first block, 1: second) of block b at outer step `step`
__device__ __forceinline__ void block_locate(int b, int step, int nblk, int& g, int& half) {
  if (step < 0) { g = b >> 1; half = b & 1; return; }
  int pos = 0;
  if (b != 0) {
    int v = b - 1 - step;                       // step <= nblk - 2: one wrap is enough
    if (v < 0) v += nblk - 1;
    pos = v + 1;
  }
  const int npair = nblk >> 1;
  if (pos < npair) { g = pos; half = 0; } else { g = nblk - 1 - pos; half = 1; }
}

// V <- V Q runs on the fp16 MFMA pipe with split operands (hi = fp16(x), lo = fp16(x - hi): 22 significand bits; hi*hi +
// hi*lo + lo*hi accumulated in fp32 -- the covariance kernel's scheme): the entries of V and Q are bounded by 1, and three
// v_mfma_f32_16x16x32_f16 replace eight v_mfma_f32_16x16x4_f32 at a sixteenth of their cost each.  The rotation matrices
// are therefore ALSO stored as fp16 fragments: unit u = ((mt * NCH + c) * 2 + part) * 64 + lane (16 bytes = 8 halfs;
// part 0 = hi, 1 = lo; NCH = M2 / 32 K-chunks) holds the A operand of output tile mt and chunk c for lane (m, g) = (lane
// & 15, lane >> 4): element j is Q[k][16 mt + m] with k = 16 (2 c + (j >> 2)) + 4 g + (j & 3).  That k-slot order is
// what makes the accumulator layout of a 16 x 16 tile of V^T (lane (n, g), register r <-> V[n][4 g + r]) the B operand
// of the next product without any data movement: elements 0..3 come from tile 2c, 4..7 from tile 2c + 1.
template <int M2>
__device__ __forceinline__ int qfrag16_k(int c, int g, int j) { return 16 * (2 * c + (j >> 2)) + 4 * g + (j & 3); }

__device__ __forceinline__ void split_f16x8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (half_t)x[j];
    lo[j] = (half_t)(x[j] - (float)hi[j]);
  }
}

// =====================================================================================================================
// Round 4: the cross step of a 64 x 64 pair problem with {S, Q} resident in REGISTERS (256 threads per pair problem).
//
// The round-2/3 kernel (jacobi_cross_sets_pw) keeps the {S, Q} image in LDS: 1024 threads, every one moves its 2 x 2 block
// of both through LDS at every rotation set (4 x ds_read_b64 + 4 x ds_write_b64) and recomputes its addresses -- ~60
// instructions per wave and set of which 24 are the rotation arithmetic; VALU issue and the LDS store path are both
// about half busy (PMC round 3), nothing overlaps them but the second resident block.
// Here (routing proven in tools/jacobi_patch_proto.py: routed sequence == plain sequence to 1e-16 for P = 1, 2, 4):
//   * a lane owns a P x P PATCH of cells (P = 2: 4 cells = 32 floats), cell (k, l) = S[{p_k, q_k}] x [{p_l, q_l}] and
//     Q[{k, 32 + k}] x [{p_l, q_l}] -- Q only takes column rotations, so its ROWS never move and its p columns never
//     leave the lane; S[p_k][p_l] never leaves either;
//   * cells are laid out in SKEWED coordinates (k, d), l = (k + d + 1) mod 32: the set s -> s+1 transition
//     (q_x advances by one) is then a uniform nearest-neighbour shift -- pq, Qpq, Qqq from cell (k, d+1), qq from
//     (k+1, d), qp from (k+1, d-1) -- most of it a register rename inside the patch; only the patch rim, 11 floats
//     per lane and set (6 P - 1) instead of 32, goes through LDS, at addresses that are constants of the lane;
//   * the cells (k, d = 0) produce S[p_k][q_k'] of the next set -- the pivots -- and live in the first 16 lanes of wave 0,
//     which follow (pp, qq) of their two pairs in closed form exactly as the pivot wave of round 2 did and publish (c, s);
//   * one barrier per set (4 waves instead of 16), ping-pong exchange buffers, no address arithmetic in the loop.
// Per set a wave issues 4 cells x 24 rotation instructions + ~25 of exchange and control (was 4 waves x ~60 for the
// same cells), and the LDS instructions drop from 40 (ds_*_b64) per four cells to 12.
// =====================================================================================================================
namespace r4 {
constexpr int NT = 256;                    // threads per pair problem / per update task

// ---------------------------------------------------------------------------------------------------------------------
// Strips: the same register-resident pair problem with an UNEVEN split of the cells, written after the first
// measurements of the 2 x 2 patches (profiles/r04_jacobi_ts.txt): with one wave per SIMD a set costs what the LONGEST
// wave's instruction stream costs, and that was wave 0 -- four cells per lane like everybody else PLUS the pivot work
// (two rotations per pivot lane, closed-form diagonals, statistics): 215 instructions against 115.  Here a lane owns a
// 1 x W strip of cells (k, d0 .. d0 + W - 1): the two half-waves of wave 0 hold the columns d = 0 and d = 1 (W = 1),
// the six half-waves of waves 1-3 five columns each.  Wave 0 rotates ONE cell per lane, its lanes 0..31 hold the 32
// pivot cells (k, 0) and derive ONE rotation each; the convergence statistics of the rotations are evaluated after the
// loop, by all lanes, from a log of the pivot blocks (see SX_LOG_B).  {S, Q} travel as float2 pairs so that the column
// rotation is packed arithmetic (the same (c, s) acts on both halves); inside a strip the set s -> s+1 transition of
// pq / Qpq / Qqq is a register rename (cell j uses the physical pair (j + s) mod W; the loop is unrolled lcm(W, 2) times).
// Exchange per lane and set: 2 W + 3 floats -- qq[0..W) and qp[0..W) to lane k - 1 (qp[W-1] to the next strip's),
// {pq, Qpq, Qqq} of the first column to the previous strip.
// ---------------------------------------------------------------------------------------------------------------------
// Two splits of the 32 columns over half-waves (strips), LAY:
//   0   8 strips on 4 waves (256 threads): 1, 1 | 5 x 6
//   1  16 strips on 8 waves (512 threads): 1, 1 | 3, 3 | 2 x 12 -- two waves per SIMD: a wave's LDS and transcendental
//      latencies hide behind its SIMD partner, and no wave carries more than three cells
template <int LAY> struct Lay;
template <> struct Lay<0> {
  static constexpr int NS = 8, NTD = 256;
  __device__ static constexpr int w(int sg) { return sg < 2 ? 1 : 5; }
  __device__ static constexpr int d0(int sg) { return sg < 2 ? sg : 2 + 5 * (sg - 2); }
  // one exchange buffer: QQ4 f32x4[192] (qq of columns 0..3 of the W = 5 lanes, index t - 64) | QP4 f32x4[192] (qp 0..3) |
  // PQ f32x2[256] ({pq, Qpq} of column 0) | E[3][256] floats (qq of the LAST column, qp of the LAST column, Qqq of column 0).
  // 11 KB a buffer: S image + two buffers + (c, s) = 38.8 KB, so that FOUR blocks of a {D, U} launch share a CU's 160 KB.
  static constexpr int QQ4 = 0, QP4 = 3072, PQ = 6144, E = 8192, BUF = 11264;
};
template <> struct Lay<1> {
  static constexpr int NS = 16, NTD = 512;
  __device__ static constexpr int w(int sg) { return sg < 2 ? 1 : (sg < 4 ? 3 : 2); }
  __device__ static constexpr int d0(int sg) { return sg < 2 ? sg : (sg < 4 ? 2 + 3 * (sg - 2) : 8 + 2 * (sg - 4)); }
  // one exchange buffer: A f32x4[512] (qq of columns 0..2, Qqq of column 0) | B f32x4[512] (qp 0..2 -- the LAST column's is
  // element W - 1 --, pq of column 0) | C float[512] (Qpq of column 0)
  static constexpr int A = 0, B = 8192, C = 16384, BUF = 18432;
};
template <int LAY> struct Xchg {
  static constexpr int CS = 2 * Lay<LAY>::BUF;       // CS[2][32] float2 (c, s)
  static constexpr int DUMMY = CS + 2 * 256;         // absorbs the (c, s) / log stores of the lanes that own no pair
  static constexpr int BYTES = DUMMY + 256 + 16;     // (the (c, s) stores add the ping-pong offset to the dummy address as well)
};
// The S image (64 x 64 floats, in front of the exchange area) is dead while the sets run: it becomes the LOG of the pivot
// blocks, entry (s, k) = float4 (pp, qq, pq, rotated?) the rotation of pair k prepared during set s was derived from.  The
// convergence statistics of those 32 x 32 rotations are evaluated AFTER the loop by all lanes, 1024 / NTD entries each
// (per set on one wave they cost that wave ~310 cycles of every set: three transcendentals and their selects --
// profiles/r04_jacobi_ts.txt -- and made it the pole of the block).
constexpr int SX_LOG_B = 32 * 32 * 16;
// Round 5: pitch of the S image of a cross step, and of the Q image the epilogue reads, in floats.  With the natural pitch
// of 64 a strip lane's gather simg[k * 64 + l] and scatter put the 32 lanes of a half-wave (k = 0..31, one l each) on ONE
// bank -- 32-way conflicts on every one of the 20 loads and 40 stores per lane, ~12 k LDS cycles per pair problem (PMC round
// 4: LDS array 32 % busy, 36 % of it conflicts).  At pitch 68 the bank of (k, l = k + d + 1) is 5 k + d + 1 mod 64: distinct
// over a half-wave; rows stay 16-byte aligned for the float4 copies of the prologue and the epilogue.  The Q image is kept
// TRANSPOSED (QT[c * 68 + r] = Q[r][c]): the strips store it with consecutive lanes on consecutive words, and the epilogue's
// fragment units -- four consecutive ROWS of one column -- are 16-byte loads (they were 12 scalar loads per unit).
constexpr int SP = 68;
constexpr int SIMG_F = 64 * SP;            // floats of one padded 64 x 64 image

// dynamic LDS of one launch of the 256-thread kernel (every block of a launch gets the same amount: the larger of its D and
// U parts).  Kept beside the device code because the CPU emulation (tests/emul) runs the blocks inside EXACTLY this many
// bytes, guard words behind them.
template <int M2, int LAY>
constexpr size_t lds_bytes(int has_d, int has_u, int first, int step_d) {
  constexpr size_t F = sizeof(float);
  size_t need = 0;
  if (has_d) {
    if (step_d < 0) need = (2 * SIMG_F + (M2 / 2) * (M2 / 2 + 4)) * F;                 // intra step: S and Q images, W of the off-diagonal product
    else {
      need = SIMG_F * F + Xchg<LAY>::BYTES;                                            // S image (pitch SP), exchange area
      if (need < 2 * SIMG_F * F) need = 2 * SIMG_F * F;                                // S and Q images of the epilogue
      (void)first;                                                                     // (look-ahead: W sits in the exchange area)
    }
  }
  const size_t u = (M2 * M2 + M2 * (M2 + 4)) * F;                                      // one rotation log (linear), T transposed
  if (has_u && need < u) need = u;
  return need;
}

// A lane's cells: S and Q halves in SEPARATE scalar registers (round 5).  Until round 4 they travelled as float2 {S, Q} pairs so
// that the column rotation was packed arithmetic; a v_pk_fma_f32 costs what two v_fma_f32 cost, and because the S halves of a
// pair are replaced by the rim exchange at every set while the Q halves stay, the pairs had to be re-assembled with 18-22 v_mov
// per set (ISA of round 4: 38 v_pk_fma + 2 v_fma + 22 v_mov per lane and set; now 60 v_fma and no moves).  Spq / Sqq / Qpq / Qqq
// are PHYSICAL slots (cell j uses slot (j + s) mod W in set s).
// (eight separate arrays behind references: as members of one struct they are one stack object, and hipcc kept a slice of it --
//  Spq and the head of Sqp -- in scratch memory)
template <int W> struct Strip { float (&Spp)[W], (&Spq)[W], (&Sqp)[W], (&Sqq)[W], (&Qpp)[W], (&Qpq)[W], (&Qqp)[W], (&Qqq)[W]; };

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>).  The physical-slot arrays of a strip
// are indexed with (j + s) mod W; written as `#pragma unroll` loops those indices are variables until the unroller has run,
// and for W = 2, 3 the arrays were left in scratch memory (hipcc 7.2: 32-40 bytes of scratch per lane, stores and reloads
// in every set); with j a template constant they are registers from the start.
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one unrolled period of the set loop: sets s = 1, 2, .., LCM (mod LCM) of an iteration
template <int LCM, class F, int... Is>
__device__ __forceinline__ void run_period(F& body, std::integer_sequence<int, Is...>) {
  (body(std::integral_constant<int, (Is + 1) % LCM>{}, std::true_type{}), ...);
}

template <int LAY, int W, bool PWAVE, int PRIO = 0>
__device__ __forceinline__ void strip_sets(Strip<W>& R, unsigned char* xb, float* simg, int t, float floor_m, float& my_off, float& my_sig) {
  using L = Lay<LAY>;
  constexpr int LCM = (W % 2) ? 2 * W : W, NS = L::NS, SX_CS = Xchg<LAY>::CS, SX_DUMMY = Xchg<LAY>::DUMMY, SX_BUF = L::BUF;
  static_assert(30 % LCM == 0, "sets 1..30 run as whole unrolled periods");
  const int k = t & 31, sg = t >> 5;
  const bool piv = PWAVE && sg == 0;                                   // lanes 0..31 of wave 0: pair k
  const int dn = sg * 32 + ((k + 1) & 31);                             // lane (k + 1, same strip)
  const int dl = ((sg + NS - 1) % NS) * 32 + ((k + 1) & 31);           // lane (k + 1, previous strip)
  const int rt = ((sg + 1) % NS) * 32 + k;                             // lane (k, next strip)
  const int wprev = L::w((sg + NS - 1) % NS);                          // width of the previous strip
  int a_l[W];
#pragma unroll
  for (int j = 0; j < W; ++j) a_l[j] = SX_CS + ((k + L::d0(sg) + j + 1) & 31) * 8;
  const int a_k = SX_CS + k * 8;
  const int a_csw = piv ? SX_CS + k * 8 : SX_DUMMY;
  // log entry of this lane's pair for the set in progress (pivot lanes; the others store into the dummy slot)
  unsigned char* logw = piv ? reinterpret_cast<unsigned char*>(simg) + k * 16 : xb + SX_DUMMY;
  const int log_step = piv ? 32 * 16 : 0;
  // SCALED rotations (fast Givens): a rotation [x'; y'] = c [1 -t; t 1] [x; y] is applied as x <- x - a y, y <- y + b x -- one
  // fma per output instead of a multiply and an fma -- and its factor c stays pending in a scale rho per INDEX: the stored
  // value of S[i][j] is S[i][j] / (rho_i rho_j), that of Q[r][j] is Q[r][j] / rho_j, and a pair (p, q) publishes
  // (a, b) = (t rho_q / rho_p, t rho_p / rho_q).  Only the pivot lanes know rho (of p_k, and of the q that is with them: it
  // travels with q); they work on TRUE values -- closed-form diagonals, pivot element = rho_p rho_q x stored.  32 cosines
  // >= 2^-1/2 each keep rho within [1.5e-5, 1]; strip_wave multiplies the scales back in when it scatters the cells.
  float ppk = 0.f, qqk = 0.f, pqk = 0.f;                               // pair k's pivot block (pivot lanes), true values
  float tpk = 0.f;                                                     // tan of the pair's rotation of the set in progress (pivot lanes)
  float rp = 1.f, rq = 1.f;                                            // rho of p_k / of the q currently paired with it
  if (PWAVE) {
    __builtin_amdgcn_s_setprio(2);                                     // the chain of a set runs through this wave
    ppk = simg[k * SP + k]; qqk = simg[(32 + k) * SP + 32 + k]; pqk = simg[k * SP + 32 + k];   // set 0 pairs k with 32 + k
    float off, sig, c0, s0;
    jacobi_rotation(ppk, qqk, pqk, floor_m, c0, s0, off, sig);
    if (piv) { my_off = fmaxf(my_off, off); my_sig = fmaxf(my_sig, sig); }
    tpk = s0 * __builtin_amdgcn_rcpf(c0);                              // (no rotation: s = 0)
    f32x2 r; r[0] = tpk; r[1] = tpk;
    *reinterpret_cast<f32x2*>(xb + a_csw) = r;
  } else if (PRIO > 0) {
    __builtin_amdgcn_s_setprio(PRIO);                                  // (tuning builds: WCT_JACOBI_DBG & 8)
  }
  __syncthreads();                                                     // (the S image is free from here on)

  auto take_rim = [&](auto SC) {                                       // the strip's rim for set S out of buffer S & 1
    constexpr int S = decltype(SC)::value, LASTP = (W - 1 + S) % W;
    const unsigned char* b = xb + (S & 1) * SX_BUF;
    if constexpr (LAY == 0) {
      if constexpr (W == 5) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(b + L::QQ4 + (dn - 64) * 16);
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(b + L::QP4 + (dn - 64) * 16);
        const float q5 = *reinterpret_cast<const float*>(b + L::E + dn * 4);
        const float pl = *reinterpret_cast<const float*>(b + L::E + 1024 + dl * 4);
        static_for<W>([&](auto J) {
          constexpr int j = decltype(J)::value;
          // (opaque: four consecutive array elements assigned from one float4 are fused into a 16-byte store, which keeps the
          //  whole array in scratch memory)
          float vq = j < 4 ? q4[j < 4 ? j : 0] : q5, vp = j == 0 ? pl : p4[j > 0 ? j - 1 : 0];
          R4_OPAQUE(vq); R4_OPAQUE(vp);
          R.Sqq[(j + S) % W] = vq;
          R.Sqp[j] = vp;
        });
      } else {
        R.Sqq[0] = *reinterpret_cast<const float*>(b + L::E + dn * 4);
        R.Sqp[0] = *reinterpret_cast<const float*>(b + L::E + 1024 + dl * 4);
      }
      const f32x2 pqin = *reinterpret_cast<const f32x2*>(b + L::PQ + rt * 8);
      R.Spq[LASTP] = pqin[0]; R.Qpq[LASTP] = pqin[1];
      R.Qqq[LASTP] = *reinterpret_cast<const float*>(b + L::E + 2048 + rt * 4);
    } else {
      const float pl = *reinterpret_cast<const float*>(b + L::B + dl * 16 + (wprev - 1) * 4);
      if constexpr (W > 1) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(b + L::A + dn * 16);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + L::B + dn * 16);
        static_for<W>([&](auto J) {
          constexpr int j = decltype(J)::value;
          float vq = a4[j], vp = j == 0 ? pl : b4[j > 0 ? j - 1 : 0];
          R4_OPAQUE(vq); R4_OPAQUE(vp);
          R.Sqq[(j + S) % W] = vq;
          R.Sqp[j] = vp;
        });
      } else {
        R.Sqq[0] = *reinterpret_cast<const float*>(b + L::A + dn * 16);
        R.Sqp[0] = pl;
      }
      R.Spq[LASTP] = *reinterpret_cast<const float*>(b + L::B + rt * 16 + 12);
      R.Qpq[LASTP] = *reinterpret_cast<const float*>(b + L::C + rt * 4);
      R.Qqq[LASTP] = *reinterpret_cast<const float*>(b + L::A + rt * 16 + 12);
    }
  };

#ifdef JACOBI_TS
  unsigned long long busy = 0;
#endif
  auto body = [&](auto SC, auto INC) {
    constexpr int S = decltype(SC)::value, CUR = S & 1, NX = CUR ^ 1;
    constexpr bool IN = decltype(INC)::value;
#ifdef JACOBI_TS
    const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    const f32x2 rk = *reinterpret_cast<const f32x2*>(xb + CUR * 256 + a_k);
    f32x2 rl[W];
    static_for<W>([&](auto J) { constexpr int j = decltype(J)::value; rl[j] = *reinterpret_cast<const f32x2*>(xb + CUR * 256 + a_l[j]); });
    if (IN) take_rim(SC);
    const float ak = rk[0], bk = rk[1];
    float nqq[W], nqp[W];
    static_for<W>([&](auto J) {
      constexpr int j = decltype(J)::value, P = (j + S) % W;
      const float al = rl[j][0], bl = rl[j][1];
      // columns (pair l: x <- x - a y, y <- y + b x) on the S rows p_k, q_k and the Q rows k, 32 + k, then rows (pair k) on the S
      // values: twelve fma per cell, every one of them scalar (see Strip)
      const float yp = __builtin_fmaf(-al, R.Spq[P], R.Spp[j]), yq = __builtin_fmaf(bl, R.Spp[j], R.Spq[P]);
      const float yqp = __builtin_fmaf(-al, R.Sqq[P], R.Sqp[j]), yqq = __builtin_fmaf(bl, R.Sqp[j], R.Sqq[P]);
      const float gp = __builtin_fmaf(-al, R.Qpq[P], R.Qpp[j]), gq = __builtin_fmaf(bl, R.Qpp[j], R.Qpq[P]);
      const float hp = __builtin_fmaf(-al, R.Qqq[P], R.Qqp[j]), hq = __builtin_fmaf(bl, R.Qqp[j], R.Qqq[P]);
      R.Spp[j] = __builtin_fmaf(-ak, yqp, yp);  R.Spq[P] = __builtin_fmaf(-ak, yqq, yq);
      nqp[j] = __builtin_fmaf(bk, yp, yqp);     nqq[j] = __builtin_fmaf(bk, yq, yqq);
      R.Qpp[j] = gp;  R.Qpq[P] = gq;  R.Qqp[j] = hp;  R.Qqq[P] = hq;
    });
    constexpr int P0 = S % W;                                          // physical pair of column 0
    if (PWAVE) {
      // pair k after this set (closed form from registers, true values); partner diagonal of the next set = the q diagonal pair
      // k + 1 has just produced (lane k + 1 of this half-wave), and with it comes that q's scale; next pivot element = this
      // lane's freshly rotated cell (k, 0), scaled back
      float ck, sk;
      jacobi_cs_from_t(tpk, ck, sk);                                   // (of the rotation published BEFORE the last barrier)
      const float c2 = ck * ck, s2 = sk * sk, cs2 = 2.f * ck * sk;
      const float ppn = c2 * ppk - cs2 * pqk + s2 * qqk;
      const float qqn = s2 * ppk + cs2 * pqk + c2 * qqk;
      const float rqn = ck * rq;
      // (a DPP wave shift: lane i reads lane i + 1, the wrap-around lane 31 <- 0 patched with a v_readlane: no LDS round
      // trip on the chain)
      const int qi = __builtin_bit_cast(int, qqn), ri = __builtin_bit_cast(int, rqn);
      const int shl = __builtin_amdgcn_update_dpp(qi, qi, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
      const int shr = __builtin_amdgcn_update_dpp(ri, ri, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
      const int first = __builtin_amdgcn_readlane(qi, 0), firstr = __builtin_amdgcn_readlane(ri, 0);
      const float nb = __builtin_bit_cast(float, k == 31 ? first : shl);
      rp = ck * rp;
      rq = __builtin_bit_cast(float, k == 31 ? firstr : shr);
      // (the ratios of the scales do not wait for the pivot element)
      const float rqp = rq * __builtin_amdgcn_rcpf(rp), rpq = rp * __builtin_amdgcn_rcpf(rq), rpr = rp * rq;
      ppk = ppn; qqk = nb; pqk = rpr * R.Spq[P0];
      const bool rot = jacobi_rotation_t(ppk, qqk, pqk, tpk);
      f32x2 r; r[0] = tpk * rqp; r[1] = tpk * rpq;
      *reinterpret_cast<f32x2*>(xb + NX * 256 + a_csw) = r;
      *reinterpret_cast<f32x4*>(logw) = f32x4{ppk, qqk, pqk, rot ? 1.f : 0.f};
      logw += log_step;
    }
    unsigned char* b = xb + NX * SX_BUF;
    if constexpr (LAY == 0) {
      if constexpr (W == 5) {
        *reinterpret_cast<f32x4*>(b + L::QQ4 + (t - 64) * 16) = f32x4{nqq[0], nqq[1], nqq[2], nqq[3]};
        *reinterpret_cast<f32x4*>(b + L::QP4 + (t - 64) * 16) = f32x4{nqp[0], nqp[1], nqp[2], nqp[3]};
      }
      *reinterpret_cast<float*>(b + L::E + t * 4) = nqq[W - 1];
      *reinterpret_cast<float*>(b + L::E + 1024 + t * 4) = nqp[W - 1];
      *reinterpret_cast<f32x2*>(b + L::PQ + t * 8) = f32x2{R.Spq[P0], R.Qpq[P0]};
      *reinterpret_cast<float*>(b + L::E + 2048 + t * 4) = R.Qqq[P0];
    } else {
      float e0 = R.Spq[P0], e1 = R.Qpq[P0], e2 = R.Qqq[P0];
      R4_OPAQUE(e0); R4_OPAQUE(e1); R4_OPAQUE(e2);
      *reinterpret_cast<f32x4*>(b + L::A + t * 16) = f32x4{nqq[0], W > 1 ? nqq[W > 1 ? 1 : 0] : 0.f, W > 2 ? nqq[W > 2 ? 2 : 0] : 0.f, e2};
      *reinterpret_cast<f32x4*>(b + L::B + t * 16) = f32x4{nqp[0], W > 1 ? nqp[W > 1 ? 1 : 0] : 0.f, W > 2 ? nqp[W > 2 ? 2 : 0] : 0.f, e0};
      *reinterpret_cast<float*>(b + L::C + t * 4) = e1;
    }
#ifdef JACOBI_TS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    busy += __builtin_amdgcn_s_memtime() - ts0;                        // release of the previous barrier -> arrival at this one
#endif
    __syncthreads();
  };
  body(std::integral_constant<int, 0>{}, std::false_type{});           // set 0: the strip is as loaded
#pragma unroll 1
  for (int it = 0; it < 30 / LCM; ++it) run_period<LCM>(body, std::make_integer_sequence<int, LCM>{});     // sets 1 .. 30
  body(std::integral_constant<int, 31 % LCM>{}, std::true_type{});     // set 31
  take_rim(std::integral_constant<int, 32 % LCM>{});                   // the arrangement of "set 32" = that of set 0
  if (PWAVE || PRIO > 0) __builtin_amdgcn_s_setprio(0);
  // the scales of the 64 indices for strip_wave's scatter (the (a, b) slots of the buffer nobody reads any more: the last set
  // read the other one, and a barrier lies between)
  if (piv) {
    *reinterpret_cast<float*>(xb + SX_CS + k * 4) = rp;
    *reinterpret_cast<float*>(xb + SX_CS + (32 + k) * 4) = rq;
  }
  // the statistics of the 32 x 32 logged rotations (the last set's are those of a rotation that is never applied: the
  // state of the pivots after the step), four per lane
#pragma unroll
  for (int i = 0; i < SX_LOG_B / 16 / L::NTD; ++i) {
    const f32x4 e = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(simg) + (t + i * L::NTD) * 16);
    float off, sig;
    jacobi_rotation_stats(e[0], e[1], e[2], floor_m, e[3] != 0.f, off, sig);
    my_off = fmaxf(my_off, off); my_sig = fmaxf(my_sig, sig);
  }
#ifdef JACOBI_TS
  if ((t & 63) == 0 && blockIdx.x < 8192) jac_busy[blockIdx.x * 8 + (t >> 6)] = busy;
#endif
}

// gather -> sets -> scatter for the lanes of one wave (strip width W); contains the barriers of the set loop and one more
template <int LAY, int W, bool PWAVE, int PRIO = 0>
__device__ __forceinline__ void strip_wave(float* simg, float* qimg, unsigned char* xb, int t, float floor_m, float& my_off, float& my_sig) {
  constexpr int B = 32;
  const int k = t & 31, sg = t >> 5;
  float Spp[W], Spq[W], Sqp[W], Sqq[W], Qpp[W], Qpq[W], Qqp[W], Qqq[W];
  Strip<W> R{Spp, Spq, Sqp, Sqq, Qpp, Qpq, Qqp, Qqq};
  int la[W];
  static_for<W>([&](auto J) {
    constexpr int j = decltype(J)::value;
    const int l = (k + Lay<LAY>::d0(sg) + j + 1) & 31;
    la[j] = l;
    const float one = k == l ? 1.f : 0.f;
    R.Spp[j] = simg[k * SP + l];        R.Spq[j] = simg[k * SP + B + l];
    R.Sqp[j] = simg[(B + k) * SP + l];  R.Sqq[j] = simg[(B + k) * SP + B + l];
    R.Qpp[j] = one;  R.Qpq[j] = 0.f;  R.Qqp[j] = 0.f;  R.Qqq[j] = one;
  });
  strip_sets<LAY, W, PWAVE, PRIO>(R, xb, simg, t, floor_m, my_off, my_sig);
  __syncthreads();                                  // every lane has taken its last rim: the exchange area becomes the Q image
  constexpr int LCM = (W % 2) ? 2 * W : W, SF = 32 % LCM;
  // the pending scales of the scaled rotations: S[i][j] = rho_i rho_j x stored, Q[r][j] = rho_j x stored (behind the Q image)
  const float* rho = reinterpret_cast<const float*>(xb + Xchg<LAY>::CS);
  const float rkp = rho[k], rkq = rho[B + k];
  static_for<W>([&](auto J) {
    constexpr int j = decltype(J)::value, P = (j + SF) % W;
    const int l = la[j];
    const float rlp = rho[l], rlq = rho[B + l];
    simg[k * SP + l] = (rkp * rlp) * R.Spp[j];        simg[k * SP + B + l] = (rkp * rlq) * R.Spq[P];
    simg[(B + k) * SP + l] = (rkq * rlp) * R.Sqp[j];  simg[(B + k) * SP + B + l] = (rkq * rlq) * R.Sqq[P];
    // Q transposed: QT[column][row]
    qimg[l * SP + k] = rlp * R.Qpp[j];        qimg[(B + l) * SP + k] = rlq * R.Qpq[P];
    qimg[l * SP + B + k] = rlp * R.Qqp[j];    qimg[(B + l) * SP + B + k] = rlq * R.Qqq[P];
  });
}

// =====================================================================================================================
// Round 5: the INTRA step (the pairs inside the two 32-wide blocks of a pair problem, once per sweep) in REGISTERS, one WAVE per
// block, no barrier and no LDS traffic inside the sets.  Until now it kept the round-2 {S, Q} image in LDS (jacobi_sets: four
// cells per thread, 32 8-byte LDS accesses per thread and set, conflicted column gathers): 158 us per launch at 64 matrices
// against 34 us for a cross step (rocprofv3 per-launch trace, profiles/r05_trace_b32.txt) -- a fifth of a sweep for 6 % of its
// index pairs.
//   * Ordering: ODD-EVEN TRANSPOSITION with exchange (the "caterpillar" ordering): even sets rotate the position pairs
//     (0,1)(2,3)..., odd sets (1,2)(3,4)..(29,30); every rotation also EXCHANGES its two positions, so after 32 sets every pair
//     of the 32 indices has met exactly once (tests/emul checks the cover; NumPy model of the whole sweep: the residual curve
//     of the round-robin ordering, sweep for sweep).  Partners are always NEIGHBOURS: nothing is re-dealt between sets.
//   * Layout: lane (a, b) = (lane >> 3, lane & 7) owns S[4a..4a+3][4b..4b+3] and Q[4a..4a+3][4b..4b+3] (rows of Q: original
//     indices, fixed; columns: positions).  Even sets are lane-local.  In odd sets the pair (4a+3, 4a+4) straddles two lanes:
//     the column (then the row) on either side is fetched from the neighbour lane (ds_bpermute: the LDS crossbar, no memory)
//     and each lane computes its own half of the rotation.
//   * The rotations are derived by the diagonal lanes (a, a) -- two per even set, two per odd set (inner pair and the pair
//     straddling into lane (a+1, a+1)) -- and broadcast by ds_bpermute.
//   * A rotation with exchange: position p takes s x_p + c x_q, position q takes c x_p - s x_q ((c, s) of jacobi_rotation,
//     which annihilates a_pq with x_p' = c x_p - s x_q, x_q' = s x_p + c x_q).  After the 32 sets the positions hold the indices
//     in reverse order; the images are written in POSITION order -- S_out = Q^T S_in Q holds for the Q that is written, and
//     nothing downstream depends on which column of a block carries which eigen-direction.
//   * The off-diagonal block of the pair problem takes no part in the sets: S_AB' = Q_A^T S_AB Q_B by two fp32-MFMA products
//     afterwards (fused_d), exactly how the tile update treats every other tile of the matrix.
// Statistics (offmax / offsig) as jacobi_rotation reports them, diagonal lanes only.
template <int M2>
__device__ __forceinline__ void intra_wave(float* simg, float* qimg, int h, int lane, float floor_m, float& my_off, float& my_sig) {
  constexpr int B = M2 / 2;
  static_assert(B == 32, "8 x 8 lanes of 4 x 4 positions");
  const int a = lane >> 3, b = lane & 7;
  const bool diag = a == b;
  auto bperm = [](int addr, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v))); };
  const int ad_row = (9 * a) << 2, ad_col = (9 * b) << 2;                        // diagonal lanes of my rows / of my columns
  const int ad_rowm = (9 * ((a + 7) & 7)) << 2, ad_colm = (9 * ((b + 7) & 7)) << 2;     // ... of the row / column block before mine
  const int ad_r = ((lane + 1) & 63) << 2, ad_l = ((lane + 63) & 63) << 2;       // lanes (a, b + 1), (a, b - 1)
  const int ad_d = ((lane + 8) & 63) << 2, ad_u = ((lane + 56) & 63) << 2;       // lanes (a + 1, b), (a - 1, b)
  const int ad_dd = ((lane + 9) & 63) << 2;                                      // lane (a + 1, b + 1)
  const bool has_r = b < 7, has_l = b > 0, has_d = a < 7, has_u = a > 0;
  float S[4][4], Q[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(simg + (h * B + 4 * a + i) * SP + h * B + 4 * b);
#pragma unroll
    for (int j = 0; j < 4; ++j) { S[i][j] = v[j]; Q[i][j] = (diag && i == j) ? 1.f : 0.f; }
  }
  // positions (P, P + 1) of this lane, columns: x_P <- s x_P + c x_{P+1}, x_{P+1} <- c x_P - s x_{P+1}
  auto cols = [&](float (&X)[4][4], int P, float c, float s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xp = X[i][P], xq = X[i][P + 1];
      X[i][P] = __builtin_fmaf(s, xp, c * xq);
      X[i][P + 1] = __builtin_fmaf(c, xp, -(s * xq));
    }
  };
  auto rows = [&](float (&X)[4][4], int P, float c, float s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xp = X[P][j], xq = X[P + 1][j];
      X[P][j] = __builtin_fmaf(s, xp, c * xq);
      X[P + 1][j] = __builtin_fmaf(c, xp, -(s * xq));
    }
  };
#pragma unroll 1
  for (int s2 = 0; s2 < B / 2; ++s2) {
    {  // ---- even set: pairs (0,1), (2,3) of every lane's rows and columns
      float c0, s0, c1, s1, o0, g0, o1, g1;
      jacobi_rotation(S[0][0], S[1][1], S[0][1], floor_m, c0, s0, o0, g0);
      jacobi_rotation(S[2][2], S[3][3], S[2][3], floor_m, c1, s1, o1, g1);
      if (diag) { my_off = fmaxf(my_off, fmaxf(o0, o1)); my_sig = fmaxf(my_sig, fmaxf(g0, g1)); }
      const float cc0 = bperm(ad_col, c0), cs0 = bperm(ad_col, s0), cc1 = bperm(ad_col, c1), cs1 = bperm(ad_col, s1);
      const float rc0 = bperm(ad_row, c0), rs0 = bperm(ad_row, s0), rc1 = bperm(ad_row, c1), rs1 = bperm(ad_row, s1);
      cols(S, 0, cc0, cs0); cols(S, 2, cc1, cs1);
      cols(Q, 0, cc0, cs0); cols(Q, 2, cc1, cs1);
      rows(S, 0, rc0, rs0); rows(S, 2, rc1, rs1);
    }
    {  // ---- odd set: pair (1,2) inside the lane, pair (3, 0') across to the next lane
      float ci, si, cx, sx, oi, gi, ox, gx;
      jacobi_rotation(S[1][1], S[2][2], S[1][2], floor_m, ci, si, oi, gi);
      const float qq = bperm(ad_dd, S[0][0]);          // diagonal lanes: S[4a+4][4a+4] of lane (a+1, a+1)
      const float pq = bperm(ad_r, S[3][0]);           // ... and S[4a+3][4a+4] of lane (a, a+1)
      jacobi_rotation(S[3][3], qq, pq, floor_m, cx, sx, ox, gx);
      if (diag) {
        my_off = fmaxf(my_off, has_d ? fmaxf(oi, ox) : oi);
        my_sig = fmaxf(my_sig, has_d ? fmaxf(gi, gx) : gi);
      }
      // columns: inner (c, s) and the straddling pair to the right from lane (b, b), the straddling pair to the left from (b-1, b-1)
      const float cci = bperm(ad_col, ci), csi = bperm(ad_col, si), ccr = bperm(ad_col, cx), csr = bperm(ad_col, sx);
      const float ccl = bperm(ad_colm, cx), csl = bperm(ad_colm, sx);
      const float rci = bperm(ad_row, ci), rsi = bperm(ad_row, si), rcd = bperm(ad_row, cx), rsd = bperm(ad_row, sx);
      const float rcu = bperm(ad_rowm, cx), rsu = bperm(ad_rowm, sx);
      float nr[4], nl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { nr[i] = bperm(ad_r, S[i][0]); nl[i] = bperm(ad_l, S[i][3]); }
      cols(S, 1, cci, csi);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x3 = S[i][3], x0 = S[i][0];
        S[i][3] = has_r ? __builtin_fmaf(csr, x3, ccr * nr[i]) : x3;        // position p of the pair to the right
        S[i][0] = has_l ? __builtin_fmaf(ccl, nl[i], -(csl * x0)) : x0;     // position q of the pair to the left
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { nr[i] = bperm(ad_r, Q[i][0]); nl[i] = bperm(ad_l, Q[i][3]); }
      cols(Q, 1, cci, csi);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x3 = Q[i][3], x0 = Q[i][0];
        Q[i][3] = has_r ? __builtin_fmaf(csr, x3, ccr * nr[i]) : x3;
        Q[i][0] = has_l ? __builtin_fmaf(ccl, nl[i], -(csl * x0)) : x0;
      }
      // rows of S, on the column-rotated values
      float nd[4], nu[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { nd[j] = bperm(ad_d, S[0][j]); nu[j] = bperm(ad_u, S[3][j]); }
      rows(S, 1, rci, rsi);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x3 = S[3][j], x0 = S[0][j];
        S[3][j] = has_d ? __builtin_fmaf(rsd, x3, rcd * nd[j]) : x3;
        S[0][j] = has_u ? __builtin_fmaf(rcu, nu[j], -(rsu * x0)) : x0;
      }
    }
  }
  // images in position order: S row-major (pitch SP), Q transposed (QT[column][row]); 16-byte stores
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = {S[i][0], S[i][1], S[i][2], S[i][3]};
    *reinterpret_cast<f32x4*>(simg + (h * B + 4 * a + i) * SP + h * B + 4 * b) = v;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 v = {Q[0][j], Q[1][j], Q[2][j], Q[3][j]};
    *reinterpret_cast<f32x4*>(qimg + (h * B + 4 * b + j) * SP + h * B + 4 * a) = v;
  }
}

// D part: the pair problem (bi, bj) of matrix m at outer step p.step_d -- 256 threads (jacobi_fused_d: 1024)
// LAY: the strip layout (Lay<LAY>::NTD threads per pair problem)
template <int M2, int LAY>
__device__ __forceinline__ void fused_d(const JacobiFusedArgs& p, int m, int g, float* jsm) {
  static_assert(M2 == 64, "the strip layouts are written for 64 x 64 pair problems");
  constexpr int B = M2 / 2, NW = M2 / 16, FR = M2 * M2, NT = Lay<LAY>::NTD, NWAVES = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, nblk = C / B, npair = nblk / 2;
  int bi, bj;
  block_pair(g, p.step_d, nblk, bi, bj);
  float floor_m = 0.f;
  float my_off = 0.f, my_sig = 0.f, my_dm = 0.f;
  bool finite = true;
  float* Qo = p.Qw + ((size_t)m * npair + g) * FR;
  float* So = p.Sw + ((size_t)m * npair + g) * FR;
  half_t* Qo16 = p.Qw16 + ((size_t)m * npair + g) * (2 * FR);
  float* Simg = jsm;                               // [M2][M2] floats (cross steps)
  if (p.first) {
    if (p.st[m].done) return;
    floor_m = p.st[m].floor;
    JTS(1);
    const float* Am = p.Pr + (size_t)m * C * C;
    {
      // (round 6: 16-byte pieces -- four consecutive columns lie in one block of the pair; 4 loads per lane where 16 scalar ones,
      //  each with its own index arithmetic, stood)
      const int r0 = tid >> 4, c4 = (tid & 15) * 4;
      const float* col = Am + pair_index<B>(c4, bi, bj);
#pragma unroll
      for (int i = 0; i < FR / 4 / NT; ++i) {
        const int r = r0 + (NT / 16) * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(col + (size_t)pair_index<B>(r, bi, bj) * C);
        finite &= (fabsf(v[0]) <= 3.0e38f) & (fabsf(v[1]) <= 3.0e38f) & (fabsf(v[2]) <= 3.0e38f) & (fabsf(v[3]) <= 3.0e38f);
        const int dd = r - c4;
        if (dd >= 0 && dd < 4) my_dm = fmaxf(my_dm, fabsf(dd == 0 ? v[0] : (dd == 1 ? v[1] : (dd == 2 ? v[2] : v[3]))));
        *reinterpret_cast<f32x4*>(Simg + r * SP + c4) = v;
      }
    }
    if (p.step_d < 0)                              // intra step: the Q image starts as zero (its diagonal blocks are written after the sets)
      for (int e = tid; e < SIMG_F; e += NT) jsm[SIMG_F + e] = 0.f;
    __syncthreads();
  } else {
    // ---- look-ahead assembly (cross steps only): the diagonal blocks out of the images D(step_u) left behind, the
    // off-diagonal block Q_g1[:, h1]^T . P_old[tile g1, g2] . Q_g2[:, h2] computed here (see jacobi_fused_d)
    int g1, h1, g2, h2;
    block_locate(bi, p.step_u, nblk, g1, h1);
    block_locate(bj, p.step_u, nblk, g2, h2);
    const float* S1 = p.Sr + ((size_t)m * npair + g1) * FR;
    const float* S2 = p.Sr + ((size_t)m * npair + g2) * FR;
    const bool same = g1 == g2;
    // (round 6: 16-byte pieces.  A lane holds columns c4 .. c4 + 3 of the rows r0 + (NT / 16) i: which HALF the row lies in is a
    //  compile-time property of i, the column half one of the lane -- 4 loads and a handful of address instructions per lane where
    //  16 scalar loads, each with its own selects, stood: the launch's first wait came after 620 instructions)
    constexpr int RP = NT / 16, NV4 = FR / 4 / NT;
    static_assert(B % RP == 0, "a pass of rows lies in one half of the pair problem");
    const int r0 = tid >> 4, c4 = (tid & 15) * 4;
    const bool clo = c4 < B;
    const int ccol = (clo ? h1 : h2) * B + (c4 & (B - 1));
    f32x4 sv[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const bool rlo = RP * i < B;
      const int rr = (rlo ? h1 : h2) * B + ((r0 + RP * i) & (B - 1));
      sv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rlo == clo || same) sv[i] = *reinterpret_cast<const f32x4*>((rlo ? S1 : S2) + rr * M2 + ccol);
    }
    // W = Q_g1[:, h1]^T X (B x M2: 8 MFMA tiles over the waves, the jobs of a wave share their tile row), then crit = W
    // Q_g2[:, h2] (B x B: one tile per wave 0..3).  Operands as in fused_u: k-slot 16 g + 4 lq + s, so the rotation matrices come
    // straight from the logs in their unit order, X from the MIRROR tile (g2, g1) of the symmetric state (four consecutive k of
    // one column = four consecutive floats of a row there), and only W passes through LDS (row-major, pitch 68, in the still
    // idle exchange area): one barrier instead of three, no staging of X / Q1 / Q2.
    constexpr int NJ = (B / 16) * NW / NWAVES, P4 = M2 + 4;
    static_assert(NW % NJ == 0, "the jobs of a wave lie in one tile row of W");
    float* Ws = jsm + SIMG_F;                       // [B][P4]
    const int li = lane & 15, lq = lane >> 4;
    const int trw = (wave * NJ) / NW;               // tile row of this wave's W jobs
    const int trc = wave / (B / 16), tcc = wave % (B / 16);      // its crit tile (waves 0..3)
    f32x4 a4[NW], xb4[NJ][NW], q2[NW];
    if (!same) {
      int b1i, b1j, b2i, b2j;
      block_pair(g1, p.step_u, nblk, b1i, b1j);
      block_pair(g2, p.step_u, nblk, b2i, b2j);
      const float* Pm = p.Pr + (size_t)m * C * C;
      const float* Q1 = p.Qr + ((size_t)m * npair + g1) * FR;
      const float* Q2 = p.Qr + ((size_t)m * npair + g2) * FR;
#pragma unroll
      for (int gg = 0; gg < NW; ++gg) {
        a4[gg] = *reinterpret_cast<const f32x4*>(Q1 + (size_t)(((2 * h1 + trw) * NW + gg) * 64 + lane) * 4);
#pragma unroll
        for (int jb = 0; jb < NJ; ++jb) {
          const int tj = (wave * NJ + jb) % NW;
          xb4[jb][gg] = *reinterpret_cast<const f32x4*>(Pm + (size_t)pair_index<B>(16 * tj + li, b2i, b2j) * C + pair_index<B>(16 * gg + 4 * lq, b1i, b1j));
        }
        q2[gg] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave < (B / 16) * (B / 16))
          q2[gg] = *reinterpret_cast<const f32x4*>(Q2 + (size_t)(((2 * h2 + tcc) * NW + gg) * 64 + lane) * 4);
      }
    }
    if (p.st[m].done) return;                       // (block-uniform)
    floor_m = p.st[m].floor;
    JTS(1);
    f32x4 crit = {0.f, 0.f, 0.f, 0.f};
    if (!same) {
#pragma unroll
      for (int jb = 0; jb < NJ; ++jb) {
        const int tj = (wave * NJ + jb) % NW;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < NW; ++gg)
#pragma unroll
          for (int sx = 0; sx < 4; ++sx) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[gg][sx], xb4[jb][gg][sx], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ws[(16 * trw + 4 * lq + r) * P4 + 16 * tj + li] = acc[r];
      }
      __syncthreads();
      JTS(2);
      if (wave < (B / 16) * (B / 16)) {
#pragma unroll
        for (int gg = 0; gg < NW; ++gg) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(Ws + (16 * trc + li) * P4 + 16 * gg + 4 * lq);
#pragma unroll
          for (int sx = 0; sx < 4; ++sx) crit = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[sx], q2[gg][sx], crit, 0, 0, 0);
        }
      }
      JTS(3);
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int r = r0 + RP * i;
      if ((RP * i < B) == clo || same) {
        const f32x4 v = sv[i];
        *reinterpret_cast<f32x4*>(Simg + r * SP + c4) = v;
        finite &= (fabsf(v[0]) <= 3.0e38f) & (fabsf(v[1]) <= 3.0e38f) & (fabsf(v[2]) <= 3.0e38f) & (fabsf(v[3]) <= 3.0e38f);
        const int dd = r - c4;
        if (dd >= 0 && dd < 4) my_dm = fmaxf(my_dm, fabsf(dd == 0 ? v[0] : (dd == 1 ? v[1] : (dd == 2 ? v[2] : v[3]))));
      }
    }
    if (!same && wave < (B / 16) * (B / 16)) {
      const int tr = wave / (B / 16), tc = wave % (B / 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tr + 4 * lq + r, col = B + 16 * tc + li;
        Simg[row * SP + col] = crit[r];
        Simg[col * SP + row] = crit[r];
        finite &= fabsf(crit[r]) <= 3.0e38f;
      }
    }
    __syncthreads();
  }
  float* Qimg = jsm + SIMG_F;                       // [M2][SP] floats, TRANSPOSED (cross steps: epilogue only)
  JTS(4);
  if (p.step_d >= 0) {
    {
      // ---- strips: gather, 32 sets, scatter, by wave (the branches execute the same barriers)
      unsigned char* xb = reinterpret_cast<unsigned char*>(jsm + SIMG_F);
      const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
      if (wv == 0) strip_wave<LAY, 1, true>(Simg, Qimg, xb, tid, floor_m, my_off, my_sig);
#ifdef WCT_TUNING
      else if (LAY == 0 && (p.dbg & 8)) strip_wave<LAY, 5, false, 1>(Simg, Qimg, xb, tid, floor_m, my_off, my_sig);   // strip waves at priority 1
#endif
      else if (LAY == 0) strip_wave<LAY, 5, false>(Simg, Qimg, xb, tid, floor_m, my_off, my_sig);
      else if (wv == 1) strip_wave<LAY, 3, false>(Simg, Qimg, xb, tid, floor_m, my_off, my_sig);
      else strip_wave<LAY, 2, false>(Simg, Qimg, xb, tid, floor_m, my_off, my_sig);
      JTS(5);
    }
  } else {
    // ---- intra step (always the first launch of a segment): waves 0 / 1 diagonalise the two 32 x 32 diagonal blocks in
    // registers (intra_wave); the off-diagonal block follows by two fp32-MFMA products, S_AB' = Q_A^T S_AB Q_B
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wv < 2) intra_wave<M2>(Simg, Qimg, wv, lane, floor_m, my_off, my_sig);
    JTS(5);
    __syncthreads();
    constexpr int WP = B + 4;
    float* Ws = jsm + 2 * SIMG_F;                   // [B][WP]
    const int li = lane & 15, lq = lane >> 4;
    const int ti = (wv >> 1) & 1, tj = wv & 1;      // 16 x 16 tile of the 32 x 32 block (waves 0..3)
    if (wv < 4) {                                   // W = Q_A^T S_AB: A operand QT rows (the Q image is transposed), B operand rows of S_AB
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < B / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Qimg[(16 * ti + li) * SP + 4 * kk + lq], Simg[(4 * kk + lq) * SP + B + 16 * tj + li], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ws[(16 * ti + 4 * lq + r) * WP + 16 * tj + li] = acc[r];
    }
    __syncthreads();
    if (wv < 4) {                                   // S_AB' = W Q_B, mirrored into S_BA'
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < B / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[(16 * ti + li) * WP + 4 * kk + lq], Qimg[(B + 16 * tj + li) * SP + B + 4 * kk + lq], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ti + 4 * lq + r, col = B + 16 * tj + li;
        Simg[row * SP + col] = acc[r];
        Simg[col * SP + row] = acc[r];
        finite &= fabsf(acc[r]) <= 3.0e38f;
      }
    }
  }
  __syncthreads();
  {
    constexpr int NCH = M2 / 32;
#pragma unroll
    for (int i = 0; i < FR / 4 / NT; ++i) {
      const int f = tid + i * NT;
      *reinterpret_cast<f32x4*>(So + (size_t)f * 4) = *reinterpret_cast<const f32x4*>(Simg + (f >> 4) * SP + (f & 15) * 4);
      int qr, qc;
      qfrag_rc<M2>(f, qr, qc);                      // fragment order (see qfrag_rc): rows qr .. qr + 3 of column qc
      *reinterpret_cast<f32x4*>(Qo + (size_t)f * 4) = *reinterpret_cast<const f32x4*>(Qimg + qc * SP + qr);
      // fp16 hi / lo fragment unit f: rows qfrag16_k(cc, g, 0..3) and (.., 4..7) are two runs of four consecutive rows
      const int l16 = f & 63, part = (f >> 6) & 1, cc = (f >> 7) % NCH, mt = (f >> 7) / NCH;
      const float* qcol = Qimg + (16 * mt + (l16 & 15)) * SP;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(qcol + qfrag16_k<M2>(cc, l16 >> 4, 0));
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(qcol + qfrag16_k<M2>(cc, l16 >> 4, 4));
      const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      half8 hi, lo;
      split_f16x8(x, hi, lo);
      *reinterpret_cast<half8*>(Qo16 + (size_t)f * 8) = part ? lo : hi;
    }
  }
  if (!finite) my_off = __builtin_inff();
  for (int o = 32; o > 0; o >>= 1) {
    my_off = fmaxf(my_off, __shfl_xor(my_off, o, 64));
    my_sig = fmaxf(my_sig, __shfl_xor(my_sig, o, 64));
    my_dm = fmaxf(my_dm, __shfl_xor(my_dm, o, 64));
  }
  if ((tid & 63) == 0) {
    if (my_off > 0.f) atomicMax(&p.st[m].offmax, __float_as_uint(my_off));
    if (my_sig > 0.f) atomicMax(&p.st[m].offsig, __float_as_uint(my_sig));
    if (my_dm > 0.f && my_dm < 3.0e38f) atomicMax(&p.st[m].dmax, __float_as_uint(my_dm));
  }
  JTS(6);
#ifdef JACOBI_TS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  JTS(7);
#endif
}

// U part: one task of the tile update (see jacobi_fused_u) by 256 threads: a wave owns a 16-row strip of the 64 x 64 tile
// (four 16 x 16 MFMA tiles).
template <int M2>
__device__ __forceinline__ void fused_u(const JacobiFusedArgs& p, int m, int task, float* jsm) {
  constexpr int B = M2 / 2, NW = M2 / 16, FR = M2 * M2, NV = FR / 4 / NT;
  static_assert(NT / 64 == NW, "one 16-row strip per wave");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid >= NT) return;                         // launched beside 512-thread pair problems: waves 4..7 have no part (before any barrier)
  const int C = p.C, nblk = C / B, npair = nblk / 2;
  const int n_off = npair * (npair - 1) / 2;
  const size_t cc = (size_t)C * C;
  if (task >= n_off && task < n_off + npair) {            // diagonal tile g: the image D(step_u) left behind
    const int g = task - n_off;
    int gi, gj;
    block_pair(g, p.step_u, nblk, gi, gj);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e4 = (tid + i * NT) * 4, lr = e4 / M2, lc = e4 % M2;
      const f32x4 v = *reinterpret_cast<const f32x4*>(p.Sr + ((size_t)m * npair + g) * FR + e4);
      *reinterpret_cast<f32x4*>(p.Pw + m * cc + (size_t)pair_index<B>(lr, gi, gj) * C + pair_index<B>(lc, gi, gj)) = v;
    }
    return;
  }
  const bool is_v = task >= n_off;               // (V tasks exist only in launches with_v)
  int g, h;
  if (is_v) { const int t = task - n_off - npair; g = t / npair; h = t % npair; }       // g = M2-row block of V
  else { int t = task; g = 0; while (t >= npair - 1 - g) { t -= npair - 1 - g; ++g; } h = g + 1 + t; }
  int hi, hj, gi = 0, gj = 0;
  block_pair(h, p.step_u, nblk, hi, hj);
  if (!is_v) block_pair(g, p.step_u, nblk, gi, gj);
  // Operands by the shortest way (round 4): the MFMA k-slot of lane-quarter lq in step (gg, s) is k = 16 gg + 4 lq + s (any
  // order of k serves, as long as both operands use it), so that a lane's four consecutive steps take four CONSECUTIVE k:
  //   * X: the wave's own 16-row strip, four 16-byte loads per lane straight from the matrix -- no LDS;
  //   * Q_g (second product, A operand): the rotation log is stored in exactly this unit order (qfrag_rc: unit f = rows
  //     16 t + 4 (l >> 4) .. + 3 of column 16 mt + (l & 15)) -- the wave's quarter, four units per lane, straight from the log;
  //   * Q_h (first product, B operand): every wave needs all of it: a LINEAR copy of the log in LDS, read back one unit at a time;
  //   * T = X Q_h goes through LDS transposed (T^t[j][k], pitch 68: 16-byte writes from the accumulators, 16-byte reads).
  // 40 LDS instructions per thread and task instead of 270, two barriers instead of three; 33.8 KB (four blocks of a {D, U}
  // launch per CU).  A V task reads its strip from V directly and uses no LDS at all.
  constexpr int P4 = M2 + 4;
  const int ti = wave, li = lane & 15, lq = lane >> 4;
  f32x4 acc[NW];
#pragma unroll
  for (int tj = 0; tj < NW; ++tj) acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (is_v) {
    constexpr int NCH = M2 / 32;
    const half_t* q16 = p.Qr16 + ((size_t)m * npair + h) * (2 * FR);
    float* Vm = p.V + m * cc;
    const float* Vrow = Vm + (size_t)(g * M2 + 16 * ti + li) * C;
    f32x4 vx[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)               // qfrag16_k(c, lq, 4 hf .. 4 hf + 3): four consecutive columns of one block
        vx[c][hf] = *reinterpret_cast<const f32x4*>(Vrow + pair_index<B>(qfrag16_k<M2>(c, lq, 4 * hf), hi, hj));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = vx[c][j >> 2][j & 3];
      half8 bh, bl;
      split_f16x8(x, bh, bl);
#pragma unroll
      for (int tj = 0; tj < NW; ++tj) {
        const half8 ah = *reinterpret_cast<const half8*>(q16 + ((size_t)((tj * NCH + c) * 2 + 0) * 64 + lane) * 8);
        const half8 al = *reinterpret_cast<const half8*>(q16 + ((size_t)((tj * NCH + c) * 2 + 1) * 64 + lane) * 8);
        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[tj], 0, 0, 0);
        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[tj], 0, 0, 0);
        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[tj], 0, 0, 0);
      }
    }
#pragma unroll
    for (int tj = 0; tj < NW; ++tj)
      *reinterpret_cast<f32x4*>(Vm + (size_t)(g * M2 + 16 * ti + li) * C + pair_index<B>(16 * tj + 4 * lq, hi, hj)) = acc[tj];
    return;
  }
  if (p.u_f16) {
  // Round 6: the update of an off-diagonal tile, Y = Q_g^T (X Q_h), on the fp16 MFMA pipe with split operands instead of 2 x 64
  // v_mfma_f32_16x16x4_f32 per wave -- 48 v_mfma_f32_16x16x32_f16 at a sixteenth of the cost each.  At 64 matrices the tile
  // update was the throughput-bound half of a {D, U} launch (profiles/r05_du_split.txt: 19 of 52.6 us).
  //   * every operand is split as hi = fp16(x), los = fp16((x - hi) 2^12) -- the lo half SCALED, its products in accumulators of
  //     their own that join at 2^-12: 22 significand bits RELATIVE to each entry down to 2^-14, an absolute 2^-36 below.  (The
  //     first version took the rotation matrices from the V pass's fp16 log, whose unscaled lo halves resolve an absolute 2^-25
  //     of entries bounded by 1: the eigensolver and transform tests passed, style-swap on a rank-deficient 512-channel
  //     covariance -- smallest kept eigenvalue 6e-5 against a norm of ~1e2 -- lost 1.5e-2: a graded matrix lives on the RELATIVE
  //     accuracy of the small couplings in Q.  The rotation matrices are therefore split here, from the fp32 log.)
  //   * scales are powers of two and follow the GRADING of the matrix: product 1 scales every ROW of X by its own s_i (|X_i s_i|
  //     <= 2^11; the row's four lanes agree on it by two shuffles), product 2 every COLUMN of T by its own s'_c (column maxima
  //     over the four waves' strips through LDS).  (One scale per 64 x 64 tile was the second version: the fuzz sweep's N << C
  //     case at C = 256 went from 2.8e-5 to 1.6e-3 of the exact outcome -- entries near the 1e-5 cut-off sat 2^-36 x 1e2 = 1.5e-9
  //     under a tile maximum of the large directions, 1.5e-4 of themselves per update.)
  //   * product 1, T = X Q_h: A = the wave's 16-row strip of X split in registers, B = Q_h's fragments, split once per block
  //     into LDS (fragment (mt, c) of lane l = the fp32 log's units (mt, 2c), (mt, 2c + 1) of lane l: qfrag16_k); the accumulator
  //     layout (lane (n, g), register r <-> T[16 ti + 4 g + r][16 tj + n]) IS one half of lane (n, g)'s slot of product 2's B
  //     fragment (elements 4 (ti & 1) + r of fragment (tj, ti >> 1)), so T goes to LDS as split fragments with 8-byte stores from
  //     the lane that computed it -- no transpose;
  //   * product 2, Y = Q_g^T T: A = the wave's quarter of Q_g (its fp32 units, split in registers), B = T's fragments from LDS;
  //     the output mapping is the fp32 path's (lane: four consecutive rows of one column), Y = acc / s'_c exactly.
  // 33.8 KB of LDS as before, three barriers (two before).
  // RESULT (profiles/r06_tile_update_f16.txt, same box A-B): eigensolver 11.47 -> 10.83 ms per 32-pair step, 8.15 -> 7.5 at 16, 6.63
  // -> 6.37 at 8; wct_eigh to full convergence as accurate as with the fp32 update (eigenvalues 2.8e-5, whitening matrix 1.3e-5 vs
  // 1.1e-5 of float64's), every transform test, golden and fuzz case inside its budget; the WHITENING the transform applies
  // (tools/probe/r06_whitening_accuracy.py: alpha = 1 against an identity-covariance style) is at 3.1e-4 / 3.3e-4 / 4.5e-5 of
  // float64's where the fp32 update gives 2.8e-4 / 6.3e-5 / 4.5e-5 -- the same band, set by the residual-based stop and the
  // second-order completion, not by the update.  Style-swap's patch matching is an ARGMAX over correlations of whitened features
  // and flips matches the oracle decides by less than ~4 x that error; its one-pair solve therefore keeps the fp32 update
  // (JacobiFusedArgs::u_f16 = 0: launch_style_swap and wct_eigh), the batched transform path takes this one.
  constexpr int NCH = M2 / 32;
  constexpr float LO_UP = 4096.f, LO_DOWN = 1.f / 4096.f;
  typedef _Float16 half4v __attribute__((ext_vector_type(4)));
  auto split4 = [](const f32x4& v, float scale, half4v& hi, half4v& los) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = v[r] * scale;
      hi[r] = (half_t)x;
      los[r] = (half_t)((x - (float)hi[r]) * LO_UP);
    }
  };
  auto cat = [](const half4v& a, const half4v& b) { return half8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; };
  auto pow2_scale = [](float mx, float& up, float& down) {        // up = 2^(11 - e), down = 1 / up, for mx = f 2^e (f in [0.5, 1))
    up = 1.f; down = 1.f;
    if (mx > 0.f && mx < 3.0e38f) {
      int e;
      frexpf(mx, &e);
      e = e < -100 ? -100 : e;                                    // (denormal dust: the scale stays finite)
      up = ldexpf(1.f, 11 - e);
      down = ldexpf(1.f, e - 11);
    }
  };
  half_t* Qh16 = reinterpret_cast<half_t*>(jsm);                 // [(mt * NCH + c) * 2 + part][64][8]: Q_h's fragments, part 0 = hi, 1 = los
  half_t* T16 = Qh16 + 2 * FR;                                    // the same layout: T with its columns scaled
  float* cmax = reinterpret_cast<float*>(T16 + 2 * FR);           // [4 waves][M2]: column maxima of the waves' strips of T
  f32x4 xv[NCH][2];
  half8 gqh[NCH], gqs[NCH];                                       // Q_g, columns 16 ti ..: A fragments (hi | los) of the two chunks
  float srow, srow_inv;                                           // the scale of X's row 16 ti + li
  {
    const float* Xrow = p.Pr + m * cc + (size_t)pair_index<B>(16 * ti + li, gi, gj) * C;
    const float* Qg = p.Qr + ((size_t)m * npair + g) * FR;
    const float* Qhg = p.Qr + ((size_t)m * npair + h) * FR;
    f32x4 gv[NW], hv[NV];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        xv[c][hf] = *reinterpret_cast<const f32x4*>(Xrow + pair_index<B>(qfrag16_k<M2>(c, lq, 4 * hf), hi, hj));
        gv[2 * c + hf] = *reinterpret_cast<const f32x4*>(Qg + (size_t)((ti * NW + 2 * c + hf) * 64 + lane) * 4);
      }
#pragma unroll
    for (int i = 0; i < NV; ++i) hv[i] = *reinterpret_cast<const f32x4*>(Qhg + (size_t)(tid + i * NT) * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {            // fp32 unit u = (mt * NW + t) * 64 + l -> elements 4 (t & 1) .. + 3 of fragment (mt, t >> 1), lane l
      const int u = tid + i * NT, l = u & 63, t = (u >> 6) % NW, mt = (u >> 6) / NW;
      half4v qh, qs;
      split4(hv[i], 1.f, qh, qs);
      half_t* slot = Qh16 + ((size_t)((mt * NCH + (t >> 1)) * 2) * 64 + l) * 8 + 4 * (t & 1);
      *reinterpret_cast<half4v*>(slot) = qh;
      *reinterpret_cast<half4v*>(slot + 64 * 8) = qs;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      half4v a0, s0, a1, s1;
      split4(gv[2 * c], 1.f, a0, s0);
      split4(gv[2 * c + 1], 1.f, a1, s1);
      gqh[c] = cat(a0, a1);
      gqs[c] = cat(s0, s1);
    }
    float mx = 0.f;                           // the row's maximum: its 64 entries sit in the four lanes li, li + 16, li + 32, li + 48
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(xv[c][hf][j]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    pow2_scale(mx, srow, srow_inv);
  }
  __syncthreads();
  f32x4 accs[NW];                                                 // the products with a scaled lo half
#pragma unroll
  for (int tj = 0; tj < NW; ++tj) accs[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {                                 // diag(s) T = (diag(s) X) Q_h
    half4v a0, s0, a1, s1;
    split4(xv[c][0], srow, a0, s0);
    split4(xv[c][1], srow, a1, s1);
    const half8 ah = cat(a0, a1), as = cat(s0, s1);
#pragma unroll
    for (int tj = 0; tj < NW; ++tj) {
      const half8 bh = *reinterpret_cast<const half8*>(Qh16 + ((size_t)((tj * NCH + c) * 2 + 0) * 64 + lane) * 8);
      const half8 bs = *reinterpret_cast<const half8*>(Qh16 + ((size_t)((tj * NCH + c) * 2 + 1) * 64 + lane) * 8);
      accs[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as, bh, accs[tj], 0, 0, 0);
      accs[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bs, accs[tj], 0, 0, 0);
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[tj], 0, 0, 0);
    }
  }
  // the lane holds T[16 ti + 4 lq + r][16 tj + li] x (the scale of row 4 lq + r of the strip, which lane 4 lq + r knows)
  float rinv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) rinv[r] = __shfl(srow_inv, 4 * lq + r, 64);
#pragma unroll
  for (int tj = 0; tj < NW; ++tj) {
    float cm = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[tj][r] = (acc[tj][r] + accs[tj][r] * LO_DOWN) * rinv[r];
      cm = fmaxf(cm, fabsf(acc[tj][r]));
    }
    cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
    cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
    if (lq == 0) cmax[wave * M2 + 16 * tj + li] = cm;
  }
  __syncthreads();
  float cup[NW], cdown[NW];                                       // the scales of the columns 16 tj + li
#pragma unroll
  for (int tj = 0; tj < NW; ++tj) {
    const int col = 16 * tj + li;
    const float mx = fmaxf(fmaxf(cmax[col], cmax[M2 + col]), fmaxf(cmax[2 * M2 + col], cmax[3 * M2 + col]));
    pow2_scale(mx, cup[tj], cdown[tj]);
    half4v th, ts;                          // elements 4 (ti & 1) + r of this lane's own slot of B fragment (tj, ti >> 1)
    split4(acc[tj], cup[tj], th, ts);
    half_t* slot = T16 + ((size_t)((tj * NCH + (ti >> 1)) * 2) * 64 + lane) * 8 + 4 * (ti & 1);
    *reinterpret_cast<half4v*>(slot) = th;
    *reinterpret_cast<half4v*>(slot + 64 * 8) = ts;
    acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
    accs[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCH; ++c)             // Y diag(s') = Q_g^T (T diag(s')): rows 16 ti .. of Y (the columns 16 ti .. of Q_g)
#pragma unroll
    for (int tj = 0; tj < NW; ++tj) {
      const half8 bh = *reinterpret_cast<const half8*>(T16 + ((size_t)((tj * NCH + c) * 2 + 0) * 64 + lane) * 8);
      const half8 bs = *reinterpret_cast<const half8*>(T16 + ((size_t)((tj * NCH + c) * 2 + 1) * 64 + lane) * 8);
      accs[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gqs[c], bh, accs[tj], 0, 0, 0);
      accs[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gqh[c], bs, accs[tj], 0, 0, 0);
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gqh[c], bh, acc[tj], 0, 0, 0);
    }
#pragma unroll
  for (int tj = 0; tj < NW; ++tj)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[tj][r] = (acc[tj][r] + accs[tj][r] * LO_DOWN) * cdown[tj];
  } else {
  float* Qh = jsm;                                  // [FR] the log of pair h, unit order
  float* Tt = jsm + FR;                             // [M2][P4] T transposed
  f32x4 xa[NW], ga[NW];
  {
    const float* Xrow = p.Pr + m * cc + (size_t)pair_index<B>(16 * ti + li, gi, gj) * C;
    const float* Qg = p.Qr + ((size_t)m * npair + g) * FR;
    const float* Qhg = p.Qr + ((size_t)m * npair + h) * FR;
    f32x4 hv[NV];
#pragma unroll
    for (int gg = 0; gg < NW; ++gg) {
      xa[gg] = *reinterpret_cast<const f32x4*>(Xrow + pair_index<B>(16 * gg + 4 * lq, hi, hj));
      ga[gg] = *reinterpret_cast<const f32x4*>(Qg + (size_t)((ti * NW + gg) * 64 + lane) * 4);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) hv[i] = *reinterpret_cast<const f32x4*>(Qhg + (size_t)(tid + i * NT) * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(Qh + (tid + i * NT) * 4) = hv[i];
  }
  __syncthreads();
#pragma unroll
  for (int gg = 0; gg < NW; ++gg) {         // T = X Qh
    f32x4 b4[NW];
#pragma unroll
    for (int tj = 0; tj < NW; ++tj) b4[tj] = *reinterpret_cast<const f32x4*>(Qh + ((tj * NW + gg) * 64 + lane) * 4);
#pragma unroll
    for (int sx = 0; sx < 4; ++sx)
#pragma unroll
      for (int tj = 0; tj < NW; ++tj)
        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gg][sx], b4[tj][sx], acc[tj], 0, 0, 0);
  }
#pragma unroll
  for (int tj = 0; tj < NW; ++tj) {         // the lane holds T[16 ti + 4 lq .. + 3][16 tj + li]
    *reinterpret_cast<f32x4*>(Tt + (16 * tj + li) * P4 + 16 * ti + 4 * lq) = acc[tj];
    acc[tj] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
#pragma unroll
  for (int gg = 0; gg < NW; ++gg) {         // Y = Qg^T T
    f32x4 b4[NW];
#pragma unroll
    for (int tj = 0; tj < NW; ++tj) b4[tj] = *reinterpret_cast<const f32x4*>(Tt + (16 * tj + li) * P4 + 16 * gg + 4 * lq);
#pragma unroll
    for (int sx = 0; sx < 4; ++sx)
#pragma unroll
      for (int tj = 0; tj < NW; ++tj)
        acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[gg][sx], b4[tj][sx], acc[tj], 0, 0, 0);
  }
  }
  float* Pw = p.Pw + m * cc;
#pragma unroll
  for (int tj = 0; tj < NW; ++tj) {
    const int col = pair_index<B>(16 * tj + li, hi, hj);
#pragma unroll
    for (int r = 0; r < 4; ++r) Pw[(size_t)pair_index<B>(16 * ti + 4 * lq + r, gi, gj) * C + col] = acc[tj][r];
    // mirror tile (h, g): this lane's four rows are four consecutive columns there
    *reinterpret_cast<f32x4*>(Pw + (size_t)col * C + pair_index<B>(16 * ti + 4 * lq, gi, gj)) = acc[tj];
  }
}
}  // namespace r4
