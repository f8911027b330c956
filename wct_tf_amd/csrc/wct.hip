// Whiten-colour transform (and AdaIN) for gfx950, replacing ops.py:24-140,282-294.
//
// Feature maps stay pixel-major (NHWC flattened: X[n][c], fp32) so none of the
// reference's four transposes exist.  Per transform:
//   K3  colstats      per-channel mean (two-stage, deterministic)
//   K4  cov           C x C covariance of the centred features, split-K; operands split into fp16
//                     hi+lo pairs (22 bits) on v_mfma_f32_32x32x16_f16, mean subtracted at load
//   K5  jacobi        batched two-sided block-Jacobi eigensolver (content+style together)
//   K6  tbuild        T = E_s f_s(L_s) E_s^T . E_c f_c(L_c) E_c^T with the 1e-5 cut-off
//   K7  apply         out = (x - mc) M^T + b,  M = alpha T + (1-alpha) I  (one GEMM,
//                     blend and re-centring folded into M and b)
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

// A launcher that returns void (the per-launch helpers of the eigensolver's trains) notes a refused launch here, with the kernel's
// name, the moment it happens; the train's driver returns it (launch_rc_take) instead of a generic failure at a later sync.
static thread_local int launch_rc_sticky = WCT_OK;
#define LAUNCH_NOTE(what) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess && launch_rc_sticky == WCT_OK) { \
    wct_set_error("launch of %s refused: %s (%s:%d)", what, hipGetErrorString(e_), __FILE__, __LINE__); launch_rc_sticky = WCT_ERR_HIP; } } while (0)
static inline int launch_rc_take() { const int r = launch_rc_sticky; launch_rc_sticky = WCT_OK; return r; }

// ---------------------------------------------------------------------------
// K3: per-channel sums over the pixel axis
// ---------------------------------------------------------------------------
// grid (nslab, 2P): matrix m = 2*pair + side (0 content, 1 style); slab s reduces rows
// [s*rows_per_slab, ...) of X_m[N_side][C]
// With a shared style (one style image for every pair of the batch: video) only pair 0's style matrix
// (matrix 1) is computed; the style matrices of the other pairs are skipped and their consumers read pair 0's.
__device__ __host__ __forceinline__ bool skip_style_mat(int mat, int shared_style) {
  return shared_style && (mat & 1) && mat > 1;
}

struct StatArgs {
  const float* x[2];     // content base [P][Nc][C], style base [P][Ns][C]
  int n[2];
  const float* u[2];     // per side: unit sums [P][ceil(n/16)][C] left by the conv epilogue that wrote x (ConvArgs::usum), or null
  const unsigned* umax[2];   // with u: [P][UMAX_SLOTS] bit patterns whose maximum is the largest value of each map
  const float* mean;     // [2P][C] or null; if set, accumulate (x-mean)^2 instead of x
  float* partial;        // [2P][nslab][C]
  float* absmax;         // [2P][nslab] max |x| of the slab (first pass only) or null
  int C, nslab;
  int shared_style;
};

// First pass (mean == null): the sum runs over UNITS of 16 consecutive rows, each added up in the fixed tree of
// unit_row_sum (common.h), then over the units of the slab in a fixed order.  A conv epilogue that wrote the features can
// hand the unit sums over (u): the 8 GB pass over the features of a 32-pair step shrinks to a pass over 1/16 of them, and
// the result is the same bit for bit whether the features come from the pipeline or from the caller (op-level entry
// points, widths that are not a multiple of 16).
__global__ __launch_bounds__(256) void colsum_kernel(StatArgs p) {
  __shared__ f32x4 red[256];
  const int mat = blockIdx.y, slab = blockIdx.x;
  if (skip_style_mat(mat, p.shared_style)) return;
  const int b = mat & 1, pair = mat >> 1;
  const int C = p.C, cq = C / 4;
  const int nrp = 256 / cq;                  // rows / units handled in parallel (C <= 1024)
  const int tid = threadIdx.x;
  const int rp = tid / cq, c4 = tid % cq;
  const int N = p.n[b];
  const float* x = p.x[b] + (size_t)pair * N * C;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float amax = 0.f;
  if (p.mean) {
    const int rows_per_slab = (N + p.nslab - 1) / p.nslab;
    const int r0 = slab * rows_per_slab;
    const int r1 = min(N, r0 + rows_per_slab);
    const f32x4 m = *reinterpret_cast<const f32x4*>(p.mean + mat * C + c4 * 4);
    if (rp < nrp)
      for (int r = r0 + rp; r < r1; r += nrp) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * C + c4 * 4);
        v -= m; acc += v * v;
      }
  } else {
    const int units = (N + 15) >> 4;
    const int ups = (units + p.nslab - 1) / p.nslab;
    const int u0 = slab * ups, u1 = min(units, u0 + ups);
    const float* U = p.u[b] ? p.u[b] + (size_t)pair * units * C : nullptr;
    if (rp < nrp)
      for (int u = u0 + rp; u < u1; u += nrp) {
        if (U) {
          acc += *reinterpret_cast<const f32x4*>(U + (size_t)u * C + c4 * 4);
        } else {
          f32x4 t[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = u * 16 + i;
            t[i] = *reinterpret_cast<const f32x4*>(x + (size_t)min(r, N - 1) * C + c4 * 4);
            if (r >= N) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};            // ragged last unit: + 0 is exact
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(t[i][0]), fabsf(t[i][1]))), fmaxf(fabsf(t[i][2]), fabsf(t[i][3])));
          }
#pragma unroll
          for (int w = 1; w < 16; w <<= 1)
#pragma unroll
            for (int i = 0; i < 16; i += 2 * w) t[i] += t[i + w];
          acc += t[0];
        }
      }
    if (U) amax = __builtin_bit_cast(float, p.umax[b][pair * UMAX_SLOTS + (tid & (UMAX_SLOTS - 1))]);   // block max below
  }
  red[tid] = acc;
  __syncthreads();
  if (rp == 0) {
    for (int j = 1; j < nrp; ++j) acc += red[j * cq + c4];
    *reinterpret_cast<f32x4*>(p.partial + ((size_t)mat * p.nslab + slab) * C + c4 * 4) = acc;
  }
  if (p.absmax) {                            // block max (max is order-independent: deterministic)
    __syncthreads();
    float* redf = reinterpret_cast<float*>(red);
    redf[tid] = amax;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (tid < st) redf[tid] = fmaxf(redf[tid], redf[tid + st]);
      __syncthreads();
    }
    if (tid == 0) p.absmax[(size_t)mat * p.nslab + slab] = redf[0];
  }
}

// scale[m] = 2^k with 2 * max|x| * 2^k in [8192, 16384): the centred features |x - mean| <= 2 max|x| then
// sit well inside the fp16 range, whatever the range of the fp32 input (1 if the input is all zero) -- computed by block 0 of
// colsum_finish_kernel (round 5; it was a launch of its own, cov_scale_kernel)
// out[m][c] = sum_slab partial / denom_side; with absmax / scale given, block 0 of a matrix also does cov_scale_kernel's job
// (round 5: one launch less per level)
__global__ void colsum_finish_kernel(const float* partial, float* out, int C, int nslab, float d0, float d1, int shared_style,
                                     const float* absmax = nullptr, float* scale = nullptr) {
  const int mat = blockIdx.y;
  if (skip_style_mat(mat, shared_style)) return;
  if (scale && blockIdx.x == 0 && threadIdx.x < 64) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nslab; i += 64) m = fmaxf(m, absmax[(size_t)mat * nslab + i]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) {
      float sc = 1.f;
      if (m > 0.f && m < 1e30f) {
        int e;
        frexpf(2.f * m, &e);                   // 2m = f * 2^e, f in [0.5, 1)
        sc = ldexpf(1.f, 14 - e);
      }
      scale[mat] = sc;
    }
  }
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // eight independent partial sums keep eight loads in flight (a single dependent chain of up to 256 L2
  // round trips made this trivial kernel take 65 us); the order is fixed, so the result is reproducible
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* pp = partial + (size_t)mat * nslab * C + c;
  int i = 0;
  for (; i + 8 <= nslab; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += pp[(size_t)(i + j) * C];
  }
  for (; i < nslab; ++i) a[i & 7] += pp[(size_t)i * C];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  out[mat * C + c] = s / ((mat & 1) == 0 ? d0 : d1);
}

// ---------------------------------------------------------------------------
// generic fp32 GEMM tile on v_mfma_f32_32x32x2_f32
//   D[m][n] = sum_k A(m,k) B(k,n)
// A element (m,k): a_kmajor ? A[k*lda+m] : A[m*lda+k];  B element (k,n): b_kmajor ? B[k*ldb+n] : B[n*ldb+k]
// ---------------------------------------------------------------------------
constexpr int GK = 16;

// one operand tile (GK x BX, k-major in LDS) moves global -> registers -> LDS in two phases so the
// loads of K-step t+1 are in flight while the MFMAs of step t run
template <int BX>
struct GemmStage {
  static constexpr int NV = GK * BX / 4 / 256;      // float4 per thread
  f32x4 v[NV];
  // kmajor: element (k, x) at src[k*ld + x];  else element (x, k) at src[x*ld + k]
  __device__ __forceinline__ void load(const float* src, int ld, bool kmajor, int k0, int kend, int x0, int X,
                                       const float* sub_x, const float* sub_k, const float* scale_k, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int item = tid + i * 256;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (kmajor) {
        const int k = item / (BX / 4), x4 = item % (BX / 4);
        const int gk = k0 + k, gx = x0 + x4 * 4;
        if (gk < kend && gx < X) {
          t = *reinterpret_cast<const f32x4*>(src + (size_t)gk * ld + gx);
          if (sub_x) t -= *reinterpret_cast<const f32x4*>(sub_x + gx);
        }
      } else {
        const int x = item / (GK / 4), k4 = item % (GK / 4);
        const int gx = x0 + x, gk = k0 + k4 * 4;
        if (gx < X && gk < kend) {          // K and ksplit are multiples of 4
          t = *reinterpret_cast<const f32x4*>(src + (size_t)gx * ld + gk);
          if (sub_k) t -= *reinterpret_cast<const f32x4*>(sub_k + gk);
          if (scale_k) t *= *reinterpret_cast<const f32x4*>(scale_k + gk);
        }
      }
      v[i] = t;
    }
  }
  __device__ __forceinline__ void store(float* lds /* [GK][BX+4] */, bool kmajor, int tid) const {
    constexpr int P = BX + 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int item = tid + i * 256;
      if (kmajor) {
        const int k = item / (BX / 4), x4 = item % (BX / 4);
        *reinterpret_cast<f32x4*>(lds + k * P + x4 * 4) = v[i];
      } else {
        const int x = item / (GK / 4), k4 = item % (GK / 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[(k4 * 4 + j) * P + x] = v[i][j];
      }
    }
  }
};

// Round 5: does the rotated matrix D + E the solver hands over need a REFRESH, E' = V^T A0 V recomputed from the eigenvectors
// and the untouched covariance?  The solver tracks the rotated matrix (fp32 tile updates, ~80 of them per element) and V (22-bit
// products, ~200 block rotations) separately, so V^T A0 V = D + E holds only to ~1e-6 ||A||.  The spectral functions take f(A0) =
// V f(D + E) V^T with E to first / second order: an inconsistency of 1e-6 ||A|| in E is harmless while the kept eigenvalues are
// within a few decades of the norm, and is the whole error (1e-3 .. 2e-3 of the transform, tests/test_gpu_fuzz.py wide bands) once
// kept eigenvalues sit 4+ decades below it -- rank-deficient covariances whose rounding noise the absolute 1e-5 cut-off keeps, gain
// up to 316.  With E' the identity f(A0) = V f(V^T A0 V) V^T is exact for orthogonal V whatever the sweeps left behind (NumPy model of
// the failing case: 3.3e-3 with the tracked E, 1.3e-6 with E', V in 22 bits either way).  Cost: two C^3 products per matrix that
// needs it, none for the others (the blocks of a batch whose predicate is false exit at once).
// Predicate (from the tracked diagonal = eigenvalue estimates): an eigenvalue that is kept, or within half a decade below the
// cut-off, and below 1e-4 of the largest.
constexpr float REFRESH_RATIO = 1e-4f;
__device__ __forceinline__ bool refresh_needed(const float* Am, int C, int tid) {     // all 256 threads of a block; contains barriers
  float dmax = 0.f, dmin = 3.0e38f;
  for (int i = tid; i < C; i += 256) {
    const float d = Am[(size_t)i * C + i];
    dmax = fmaxf(dmax, fabsf(d));
    // kept, or within half a decade below the cut-off (spectral_add2_kernel's `near` band: its side of the cut-off is not settled)
    if (d > 3.3e-6f) dmin = fminf(dmin, d);
  }
  for (int o = 32; o > 0; o >>= 1) { dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64)); dmin = fminf(dmin, __shfl_xor(dmin, o, 64)); }
  __shared__ float red[2][4];
  if ((tid & 63) == 0) { red[0][tid >> 6] = dmax; red[1][tid >> 6] = dmin; }
  __syncthreads();
  dmax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  dmin = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
  return dmin < 3.0e38f && dmin < REFRESH_RATIO * dmax;
}

template <int BM, int BN>
// (round 5: four blocks per CU -- 126 registers with the accumulators in VGPRs, 144 with AGPRs before: the tail's products
//  3.86 -> 3.81 ms per 32-pair step, bit-identical; a K-stage of 32 instead of 16 loses 0.4 ms: profiles/r05_gemm_variants.txt)
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(GemmArgs p) {
  constexpr int TM = BM / 64, TN = BN / 64;       // 32x32 MFMA tiles per wave (2x2 waves)
  constexpr int PA = BM + 4, PB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[GK * PA];
  __shared__ __attribute__((aligned(16))) float Bs[GK * PB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int batch = blockIdx.z / p.nsplit, split = blockIdx.z % p.nsplit;
  if (p.skip_shared && skip_style_mat(batch, 1)) return;
  if (p.mask_in && !p.mask_in[batch]) return;
  if (p.mask_diag) {                             // (uniform per block)
    const bool need = refresh_needed(p.mask_diag + batch * p.s_mask, p.M, tid);
    if (p.mask_out && blockIdx.x == 0 && blockIdx.y == 0 && split == 0 && tid == 0) p.mask_out[batch] = need ? 1 : 0;
    if (!need) return;
  }
  const int kbeg = split * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);
  if (p.A_odd) p.A = (batch & 1) ? p.A_odd + (batch >> 1) * p.sA_odd : p.A + (batch >> 1) * p.sA;
  else p.A += batch * p.sA;
  p.B += batch * p.sB;
  if (p.a_sub_m) p.a_sub_m += batch * p.s_sub_m;
  if (p.b_sub_n) p.b_sub_n += batch * p.s_sub_n;
  if (p.a_sub_k) p.a_sub_k += batch * p.s_sub_k;
  if (p.a_scale_k) p.a_scale_k += batch * p.s_scale_k;
  if (p.bias_n) p.bias_n += batch * p.s_bias;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  GemmStage<BM> sa;
  GemmStage<BN> sb;
  if (kbeg < kend) {
    sa.load(p.A, p.lda, p.a_kmajor, kbeg, kend, m0, p.M, p.a_sub_m, p.a_sub_k, p.a_scale_k, tid);
    sb.load(p.B, p.ldb, p.b_kmajor, kbeg, kend, n0, p.N, p.b_sub_n, nullptr, nullptr, tid);
  }
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    sa.store(As, p.a_kmajor, tid);
    sb.store(Bs, p.b_kmajor, tid);
    __syncthreads();
    if (k0 + GK < kend) {
      sa.load(p.A, p.lda, p.a_kmajor, k0 + GK, kend, m0, p.M, p.a_sub_m, p.a_sub_k, p.a_scale_k, tid);
      sb.load(p.B, p.ldb, p.b_kmajor, k0 + GK, kend, n0, p.N, p.b_sub_n, nullptr, nullptr, tid);
    }
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      float a[TM], b[TN];
      const int kr = kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kr * PA + (wm * TM + i) * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kr * PB + (wn * TN + j) * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: reg r of a tile = row (r&3)+8*(r>>2)+4*(lane>>5), col lane&31
  float* o32 = p.out32 ? p.out32 + batch * p.s_out + (size_t)split * p.out_split_stride : nullptr;
  half_t* o16 = p.out16 ? p.out16 + batch * p.s_out : nullptr;
  float biasv[TN];                               // fetched before the first store (see conv_epilogue_t)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
    biasv[j] = (p.bias_n && gn < p.N) ? p.bias_n[gn] : 0.f;
  }
  // ... and handed to the store loop as plain register values: hipcc otherwise re-issues `s_waitcnt vmcnt(0)` at the first
  // use in every predicated block, and each of those waits for all stores before it (64 serial round trips per thread)
#pragma unroll
  for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(biasv[j]));
  float av = 0.f;                                // blend epilogue: max |M| of this lane's elements
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
      const float bias = biasv[j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < p.M && gn < p.N) {
          float v = acc[i][j][r] + bias;
          if (p.blend) {                         // (uniform) what blend_matrix_kernel did to the stored T, element by element
            v = p.alpha * acc[i][j][r];
            if (gm == gn) v += 1.f - p.alpha;
            av = fmaxf(av, fabsf(v));
          }
          if (o32) o32[(size_t)gm * p.ldo + gn] = v;
          if (o16) o16[(size_t)gm * p.ldo + gn] = (half_t)v;
        }
      }
    }
  if (p.blend) {                                 // one atomic per wave; the maximum does not depend on the order
    for (int o = 32; o > 0; o >>= 1) av = fmaxf(av, __shfl_xor(av, o, 64));
    if (lane == 0 && av < 1e30f) atomicMax(p.mabs + batch, __float_as_uint(av));
  }
}

int launch_gemm(GemmArgs g, int nsplit, int nbatch, hipStream_t s) {
  g.nsplit = nsplit;
  const int gz = nsplit * nbatch;
  // few large tiles leave most of the chip idle when the batch is small (one pair's 512 x 512 products are 16 tiles of
  // 128 x 128 with a K loop of 512: 45 us apiece, 0.9 ms of a batch-1 frame); 64 x 64 tiles give four times the blocks.
  // Every output element is the same k-ordered fma chain under either tiling: the results are bit-identical.
  const bool small_grid = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * gz < 256;
  if (g.M >= 128 && g.N >= 128 && !small_grid) {
    dim3 grid(cdiv(g.N, 128), cdiv(g.M, 128), gz);
    hipLaunchKernelGGL((gemm_f32_kernel<128, 128>), grid, dim3(256), 0, s, g);
  } else if (g.M >= 128 && !small_grid) {
    dim3 grid(cdiv(g.N, 64), cdiv(g.M, 128), gz);
    hipLaunchKernelGGL((gemm_f32_kernel<128, 64>), grid, dim3(256), 0, s, g);
  } else {
    dim3 grid(cdiv(g.N, 64), cdiv(g.M, 64), gz);
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64>), grid, dim3(256), 0, s, g);
  }
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// K4: covariance partials  S[i][j] = sum_n (x[n][i]-m_i)(x[n][j]-m_j) s^2  on the fp16 MFMA pipe with
// split operands.  Every centred, scaled fp32 value v is split as v = hi + lo, hi = fp16(v),
// lo = fp16(v - hi) (the subtraction is exact): 22 significand bits, and hi*hi + hi*lo + lo*hi is
// accumulated in fp32 by three v_mfma_f32_32x32x16_f16 (the dropped lo*lo term is 2^-22 relative).  That
// is fp32-product accuracy at 3/16 of the fp32-MFMA time (v_mfma_f32_32x32x2_f32: 64 cycles for K=2).
// s is a power of two per matrix (colsum_finish_kernel) so no fp32 input can leave the fp16 range.
// Only tiles on or above the diagonal are computed (cov_finish_kernel mirrors); a diagonal tile stages
// its operand once.  Block = BT x BT tile, 256 threads = 2x2 waves; K-stage = 32 pixels.
// LDS operand image: [channel][32 k] fp16 = 64-B rows, 16-B pieces XOR-swizzled as in the conv kernel.
// ---------------------------------------------------------------------------
struct CovArgs {
  const float* x[2];     // content base [P][Nc][C], style base [P][Ns][C]
  int n[2];
  const float* mean;     // [2P][C]
  const float* scale;    // [2P]
  float* partial;        // [2P][nsplit][C][C]
  int C, ksplit, nsplit, ntile;   // ntile = tiles per side
  int shared_style;
};

template <int BT>
__global__ __launch_bounds__(256, 2) void cov_f16x2_kernel(CovArgs p) {
  constexpr int TM = BT / 64;                  // 32x32 MFMA tiles per wave and side
  constexpr int KPT = BT * 32 / 256;           // k values staged per thread and operand (16 or 8)
  constexpr int NPC = KPT / 8;                 // 16-B pieces per thread and operand half
  __shared__ __attribute__((aligned(16))) unsigned char lds[4][BT * 64];   // A hi, A lo, B hi, B lo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // upper-triangular tile index -> (ti <= tj)
  int ti = 0, rem = blockIdx.x;
  while (rem >= p.ntile - ti) { rem -= p.ntile - ti; ++ti; }
  const int tj = ti + rem;
  const bool diag = ti == tj;
  const int m0 = ti * BT, n0 = tj * BT;
  const int mat = blockIdx.z, split = blockIdx.y;
  if (skip_style_mat(mat, p.shared_style)) return;
  const int side = mat & 1, pair = mat >> 1;
  const int N = p.n[side], C = p.C;
  const float* x = p.x[side] + (size_t)pair * N * C;
  const int kbeg = split * p.ksplit;
  const int kend = min(N, kbeg + p.ksplit);
  const float sc = p.scale[mat];

  // staging role: channel c of the tile, k-group kg (wave-uniform).  Loads go through a buffer resource
  // (32-bit lane offset + scalar row offset, rows past N read 0, no branches around the loads).  Rows past
  // the slice end get a zero scale (scalar select), so they contribute exactly 0; channels past C (ragged
  // tile) produce values that the store mask drops.
  const int c = tid % BT;
  const int kg = __builtin_amdgcn_readfirstlane(tid / BT);
  const float mean_a = p.mean[mat * C + min(m0 + c, C - 1)];
  const float mean_b = p.mean[mat * C + min(n0 + c, C - 1)];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((size_t)N * C * 4), 0x00020000);
  const int voff_a = min(m0 + c, C - 1) * 4, voff_b = min(n0 + c, C - 1) * 4;

  f32x16 acc[TM][TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load = [&](float (&r)[KPT], int voff, int k0) {
#pragma unroll
    for (int j = 0; j < KPT; ++j)
      r[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (k0 + kg * KPT + j) * C * 4, 0));
  };
  // split (x - mean) * s into fp16 hi + lo and park the 16-B pieces in the swizzled LDS image
  // (`tail`: the stage straddles the slice end; rows past it get a zero scale -- a uniform select that only
  //  the last stage of a slice pays for)
  auto split_store = [&](const float (&r)[KPT], float mean, int k0, bool tail, unsigned char* hi, unsigned char* lo) {
#pragma unroll
    for (int q = 0; q < NPC; ++q) {
      half8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float s_in = (!tail || k0 + kg * KPT + q * 8 + j < kend) ? sc : 0.f;
        const float v = (r[q * 8 + j] - mean) * s_in;
        h[j] = (half_t)v;
        l[j] = (half_t)(v - (float)h[j]);
      }
      const int chunk = kg * NPC + q;          // 16-B piece (8 k values) within the 64-B row
      const int off = (c * 4 + (chunk ^ ((c >> 2) & 3))) * 16;
      *reinterpret_cast<half8*>(hi + off) = h;
      *reinterpret_cast<half8*>(lo + off) = l;
    }
  };
  auto mma_stage = [&](const unsigned char* bh, const unsigned char* bl) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      half8 ah[TM], al[TM], bhf[TM], blf[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + (lane & 31);
        const int off = (r * 4 + (chunk ^ ((r >> 2) & 3))) * 16;
        ah[i] = *reinterpret_cast<const half8*>(lds[0] + off);
        al[i] = *reinterpret_cast<const half8*>(lds[1] + off);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int r = (wn * TM + j) * 32 + (lane & 31);
        const int off = (r * 4 + (chunk ^ ((r >> 2) & 3))) * 16;
        bhf[j] = *reinterpret_cast<const half8*>(bh + off);
        blf[j] = *reinterpret_cast<const half8*>(bl + off);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bhf[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], blf[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bhf[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  float ra[KPT], rb[KPT];
  if (diag) {                                  // one operand: the tile is its own transpose partner
    if (kbeg < kend) load(ra, voff_a, kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      if (k0 + 32 <= kend) split_store(ra, mean_a, k0, false, lds[0], lds[1]);
      else split_store(ra, mean_a, k0, true, lds[0], lds[1]);
      __syncthreads();
      if (k0 + 32 < kend) load(ra, voff_a, k0 + 32);
      mma_stage(lds[0], lds[1]);
      __syncthreads();
    }
  } else {
    if (kbeg < kend) { load(ra, voff_a, kbeg); load(rb, voff_b, kbeg); }
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      if (k0 + 32 <= kend) {
        split_store(ra, mean_a, k0, false, lds[0], lds[1]);
        split_store(rb, mean_b, k0, false, lds[2], lds[3]);
      } else {
        split_store(ra, mean_a, k0, true, lds[0], lds[1]);
        split_store(rb, mean_b, k0, true, lds[2], lds[3]);
      }
      __syncthreads();
      if (k0 + 32 < kend) { load(ra, voff_a, k0 + 32); load(rb, voff_b, k0 + 32); }
      mma_stage(lds[2], lds[3]);
      __syncthreads();
    }
  }

  // reg r of a tile = row (r&3)+8*(r>>2)+4*(lane>>5), col lane&31
  float* out = p.partial + ((size_t)mat * p.nsplit + split) * C * C;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int gn = n0 + (wn * TM + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < C && gn < C) out[(size_t)gm * C + gn] = acc[i][j][r];
      }
    }
}

// cov[m] = sum_split partial / (scale_m^2 (N_m - 1)) + eps I; entries below the diagonal tiles are the
// mirror of the computed upper tiles (BT = tile side of the partials)
// (round 4: in 64 x 64 tiles -- a tile below the diagonal of the BT grid reads its mirror tile's rows, coalesced, and turns
//  them in LDS; element by element the lower triangle walked columns of every partial.  Same sums in the same order.)
// grid (C / 64, C / 64, 2P)
__global__ __launch_bounds__(256) void cov_finish_kernel(const float* partial, const float* scale, float* cov, int C, int nsplit, int BT,
                                                         float inv0, float inv1, float eps, int shared_style, float* cov0 = nullptr) {
  __shared__ float tt[64][65];
  const int mat = blockIdx.z;             // 2*pair + side
  if (skip_style_mat(mat, shared_style)) return;
  const size_t cc = (size_t)C * C;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const bool mirror = r0 / BT > c0 / BT;  // (uniform: BT is a multiple of 64)
  const float* pb = partial + (size_t)mat * nsplit * cc;
  const float sc = scale[mat];
  const float f = ((mat & 1) == 0 ? inv0 : inv1) / (sc * sc);
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // direct: element (r0 + 4 ty + i, c0 + 4 tx ..); mirror: element (c0 + 4 ty + i, r0 + 4 tx ..) of the upper triangle
    const int r = (mirror ? c0 : r0) + ty * 4 + i, c = (mirror ? r0 : c0) + tx * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (r < C && c < C)
      for (int k = 0; k < nsplit; ++k) sum += *reinterpret_cast<const f32x4*>(pb + (size_t)k * cc + (size_t)r * C + c);
    acc[i] = sum;
  }
  if (mirror) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) tt[ty * 4 + i][tx * 4 + j] = acc[i][j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = tt[tx * 4 + j][ty * 4 + i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty * 4 + i, c = c0 + tx * 4;
    if (r >= C || c >= C) continue;
    f32x4 v = acc[i] * f;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (r == c + j) v[j] += eps;
    *reinterpret_cast<f32x4*>(cov + (size_t)mat * cc + (size_t)r * C + c) = v;
    if (cov0) *reinterpret_cast<f32x4*>(cov0 + (size_t)mat * cc + (size_t)r * C + c) = v;     // (the copy the solver does not rotate: refresh_needed)
  }
}

// ---------------------------------------------------------------------------
// K5: batched symmetric eigensolver -- two-sided block Jacobi.
//   Column blocks of B = M2/2 are paired round-robin; each pair's M2 x M2 diagonal
//   problem gets one cyclic Jacobi sweep in LDS (jacobi_diag_kernel), then every
//   M2 x M2 tile of A (and of the eigenvector matrix V) is updated as Qg^T A_gh Qh on
//   fp32 MFMA (jacobi_update_kernel).  Two launches per step, C/B-1 steps per sweep.
//   A per-matrix `done` flag turns later launches into no-ops.
// ---------------------------------------------------------------------------
#ifdef JACOBI_TS
__device__ unsigned long long jac_ts[8192 * 10];
__device__ unsigned long long jac_busy[8192 * 8];      // strip_sets: cycles each wave spent between the barriers of its 32 sets
#define JTS(slot) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { jac_ts[blockIdx.x * 10 + (slot)] = __builtin_amdgcn_s_memtime(); if ((slot) == 0) jac_ts[blockIdx.x * 10 + 8] = wall_clock64(); if ((slot) == 7) jac_ts[blockIdx.x * 10 + 9] = wall_clock64(); } } while (0)
#else
#define JTS(slot) do {} while (0)
#endif
#include "jacobi_dev.h"
#ifdef WCT_TUNING
#define JDBG(p) ((p).dbg)       // WCT_JACOBI_DBG timing experiments (results invalid): tuning builds only
#else
#define JDBG(p) 0
#endif



// The cross sweep with ONE wave carrying the rotation parameters ("pivot wave"), the default since round 2.
// In jacobi_cross_sets every thread derives rotation(l) itself (~35 instructions with three transcendentals on a
// dependent chain, 32x redundant) and the owners mirror the pivots into the D/O arrays.  Here:
//   * threads are numbered along the diagonals of the (k, l) grid: k = t % NP, d = t / NP, l = k + d + 1 (mod NP), so the
//     NP threads (k, k+1) that own S[p_k][q_k'] -- the element pair k annihilates in the NEXT set -- are the first NP
//     lanes of wave 0;
//   * those lanes keep pair k's pivot block (pp, qq, pq) in registers: pp and qq follow in closed form from the
//     rotation they computed for this set (c^2 pp - 2cs pq + s^2 qq, ...), the q diagonal of pair k+1 arrives by one
//     lane shuffle, pq is their own freshly rotated element; they derive the next rotation and publish (c, s) in a
//     ping-pong LDS array; every other wave reads (c_l, s_l), (c_k, s_k) -- two 8-byte loads -- and only rotates;
//   * within a set every element of the {S, Q} image is read and written by exactly one thread, so the image is
//     updated in place (one image instead of two); with PITCH = N the 8-byte accesses of a diagonal are conflict-free.
// Per set a bulk wave issues ~45 instructions instead of ~100, and the dependent chain of a set is the pivot wave's.
// exp_mode: timing experiments of a -DWCT_TUNING build (WCT_JACOBI_DBG: results INVALID); constant 0 in the product, so
// none of their predicates exists in the shipped kernels
template <int N>
__device__ __forceinline__ void jacobi_cross_sets_pw(unsigned char* sq, unsigned char* csb, int t, float floor_m, float& my_off, float& my_sig, int exp_mode_in = 0) {
#ifdef WCT_TUNING
  const int exp_mode = exp_mode_in;
#else
  constexpr int exp_mode = 0;
#endif
  constexpr int NP = N / 2, ROWB = N * 8;                            // bytes per image row
  constexpr int CSB = NP * 8, DUMMY = 2 * CSB;                       // csb layout: CS[2][NP] float2 (c, s), dummy float2
  const int k = t & (NP - 1), d = t / NP;
  const int l = (k + d + 1) & (NP - 1);
  const bool pwave = __builtin_amdgcn_readfirstlane(t >> 6) == 0;    // wave-uniform: the wave holding the lanes (k, k+1)
  const bool piv = d == 0;
  const int nb_lane = (t & 63 & ~(NP - 1)) | ((k + 1) & (NP - 1));   // lane of pair k+1
  float ppk = 0.f, qqk = 0.f, pqk = 0.f, ck = 1.f, sk = 0.f;         // pair k's pivot block and rotation (pivot wave only)
  if (pwave) {
    __builtin_amdgcn_s_setprio(2);                                   // the chain of a set runs through this wave
    const f32x2* SQ = reinterpret_cast<const f32x2*>(sq);
    ppk = SQ[k * N + k][0]; qqk = SQ[(NP + k) * N + NP + k][0]; pqk = SQ[k * N + NP + k][0];   // set 0 pairs k with NP + k
    float off, sig;
    jacobi_rotation(ppk, qqk, pqk, floor_m, ck, sk, off, sig);
    if (piv) { my_off = fmaxf(my_off, off); my_sig = fmaxf(my_sig, sig); }
    f32x2 r; r[0] = ck; r[1] = sk;
    *reinterpret_cast<f32x2*>(csb + (piv ? k * 8 : DUMMY)) = r;
  }
  __syncthreads();
  const int row_pk = k * ROWB, col_pl = l * 8;
  const int a_pp = row_pk + col_pl;
  int qk = NP + k, ql = NP + l;                                      // q index of set s: NP + ((j + s) & (NP - 1))
  const int cs_l = l * 8, cs_k = k * 8;
  const int cs_w0 = piv ? k * 8 : DUMMY, cs_w1 = piv ? CSB + k * 8 : DUMMY;   // the dummy slot absorbs the lanes that own no pair
  auto ld2 = [&](const unsigned char* base, int off) { return *reinterpret_cast<const f32x2*>(base + off); };
  auto body = [&](auto CURC) {
    constexpr int CUR = decltype(CURC)::value, NX = CUR ^ 1;
    const f32x2 rl = ld2(csb + CUR * CSB, cs_l), rk = ld2(csb + CUR * CSB, cs_k);
    const int qlb = ql * 8, qkb = qk * ROWB;
    const int a_pq = row_pk + qlb, a_qp = qkb + col_pl, a_qq = qkb + qlb;
    const f32x2 app = ld2(sq, a_pp), apq = ld2(sq, a_pq), aqp = ld2(sq, a_qp), aqq = ld2(sq, a_qq);
    const float cl = rl[0], sl = rl[1], ckk = rk[0], skk = rk[1];
    // columns (pair l) on the {S, Q} pairs as packed operations -- the same (c_l, s_l) acts on both halves of a
    // float2, so v_pk_mul / v_pk_fma take the scalars broadcast and need no operand assembly --, then rows (pair k) on
    // the S halves alone.  (Written element by element the same arithmetic came out of hipcc as 12 packed operations
    // fed by ~30 v_mov: the set loop is VALU-issue bound, ablation in profiles/r03_jacobi_set_ablation.txt.)
    const f32x2 ypp = cl * app - sl * apq, ypq = sl * app + cl * apq;
    const f32x2 yqp = cl * aqp - sl * aqq, yqq = sl * aqp + cl * aqq;
    f32x2 npp = ypp, npq = ypq, nqp = yqp, nqq = yqq;
    npp[0] = ckk * ypp[0] - skk * yqp[0];  npq[0] = ckk * ypq[0] - skk * yqq[0];
    nqp[0] = skk * ypp[0] + ckk * yqp[0];  nqq[0] = skk * ypq[0] + ckk * yqq[0];
    if (pwave) {
      // pair k after this set's rotation (closed form from registers), the next set's partner diagonal from the
      // lane of pair k+1 -- a DPP wave shift (lane i reads lane i+1), the wrap-around lane NP-1 <- 0 patched with a
      // v_readlane: no LDS round trip on the chain (round 2 used ds_bpermute here) --, the next pivot element from
      // this thread's own block
      const float c2 = ck * ck, s2 = sk * sk, cs2 = 2.f * ck * sk;
      const float ppn = c2 * ppk - cs2 * pqk + s2 * qqk;
      const float qqn = s2 * ppk + cs2 * pqk + c2 * qqk;
      const int qqn_i = __builtin_bit_cast(int, qqn);
      const int shl = __builtin_amdgcn_update_dpp(qqn_i, qqn_i, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
      const int first = __builtin_amdgcn_readlane(qqn_i, 0);
      const float qq_next = __builtin_bit_cast(float, k == NP - 1 ? first : shl);
      ppk = ppn; qqk = qq_next; pqk = npq[0];
      bool rot = false;
      if (!(exp_mode & 16)) rot = jacobi_rotation_cs(ppk, qqk, pqk, ck, sk);     // (timing experiment: no rotation on the chain)
      f32x2 r; r[0] = ck; r[1] = sk;
      *reinterpret_cast<f32x2*>(csb + (NX ? cs_w1 : cs_w0)) = r;
      float off, sig;
      jacobi_rotation_stats(ppk, qqk, pqk, floor_m, rot, off, sig);
      if (piv) { my_off = fmaxf(my_off, off); my_sig = fmaxf(my_sig, sig); }
    }
    if (!(exp_mode & 64)) {                                          // (timing experiment: no image stores)
    *reinterpret_cast<f32x2*>(sq + a_pp) = npp;  *reinterpret_cast<f32x2*>(sq + a_pq) = npq;
    *reinterpret_cast<f32x2*>(sq + a_qp) = nqp;  *reinterpret_cast<f32x2*>(sq + a_qq) = nqq;
    } else { my_off += npp[0] + npq[1] + nqp[0] + nqq[1]; }
    qk = NP | ((qk + 1) & (NP - 1));
    ql = NP | ((ql + 1) & (NP - 1));
    if (!(exp_mode & 32)) __syncthreads();                           // (timing experiment: no barrier per set)
  };
#pragma unroll 1
  for (int s = 0; s < NP; s += 2) {
    body(std::integral_constant<int, 0>{});
    body(std::integral_constant<int, 1>{});
  }
}

template <int M2>
static size_t jacobi_diag_lds(int step, bool pw) {
  if (pw && step >= 0) return (size_t)M2 * M2 * sizeof(f32x2) + (size_t)(M2 + 2) * sizeof(f32x2);   // image, CS[2][M2/2], dummy
  return (size_t)2 * M2 * (M2 + 1) * sizeof(f32x2) + 4 * M2 * sizeof(float);                         // images, D[2][M2], O[2][M2/2], dummy[M2]
}

// ---------------------------------------------------------------------------
// K5, look-ahead launches (round 3).  The serial chain of a solve used to be  D(s) -> U(s) -> D(s+1) -> ...  (pair
// problems, then the tile update that needs their rotations, then the next pair problems that need the updated
// tiles): two dependent launches per outer step, the latency-bound pair kernel idle while the tile update runs and
// vice versa.  Here ONE launch carries  { D(s), U(s-1) }:
//   * D(s) does not wait for U(s-1).  Its 2B x 2B pair problem (blocks bi, bj) is assembled from the rotated images
//     D(s-1) left behind (Sbuf: they ARE the diagonal tiles after U(s-1)) and ONE off-diagonal B x B block it
//     computes itself -- crit = Q_g1[:, h1]^T . P_old[tile g1, g2] . Q_g2[:, h2], with (g1, h1) / (g2, h2) the pair and
//     half that held bi / bj at step s-1 -- from the matrix state BEFORE U(s-1) and the rotations of step s-1;
//   * U(s-1) reads P_old and writes every tile into the other buffer P_new (off-diagonal tiles g < h computed and
//     mirrored, diagonal tiles copied from Sbuf), so D(s) can read P_old while it runs; V is updated in place.
// Both parts use the whole block (M2/2)^2 threads = (M2/16)^2 waves, one 16x16 output tile of v_mfma_f32_16x16x4_f32
// per wave.  The chain of a solve becomes D -> D -> D ...; the tile updates run beside it.
// Data routing checked against the plain sequence in tools/jacobi_lookahead_proto.py.
// ---------------------------------------------------------------------------
// Phase timing build (-DJACOBI_TS, tools/r03_jacobi_ts.sh): lane 0 of every pair-problem block stamps s_memtime at its
// phase boundaries; the launcher prints per-launch means.  Compiled out of the product.

template <int M2>
static size_t jacobi_fused_lds(int has_d, int has_u, int first, int step_d) {
  constexpr int B = M2 / 2;
  size_t need = 0;
  if (has_d) {
    if (step_d < 0) need = jacobi_diag_lds<M2>(-1, true);                               // intra sets: two images
    else {
      need = jacobi_diag_lds<M2>(0, true);
      if (!first) need = std::max(need, (size_t)(M2 * (M2 + 1) + 2 * M2 * (B + 1) + B * (M2 + 1)) * sizeof(float));
    }
  }
  if (has_u) need = std::max(need, (size_t)3 * M2 * (M2 + 1) * sizeof(float));
  return need;
}

template <int M2>
__device__ __forceinline__ void jacobi_fused_d(const JacobiFusedArgs& p, int m, int g, float* jsm) {
  constexpr int B = M2 / 2, NT = B * B, KB = 1, NW = M2 / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, nblk = C / B, npair = nblk / 2;
  f32x2* SQ = reinterpret_cast<f32x2*>(jsm);
  int bi, bj;
  block_pair(g, p.step_d, nblk, bi, bj);
  // the matrix' state words (done, floor) are read AFTER the block's data loads have been issued: a done matrix costs a few
  // wasted loads, a live one no longer waits a memory round trip before it requests anything (round 3 phase stamps)
  float floor_m = 0.f;
  float my_off = 0.f, my_sig = 0.f, my_dm = 0.f;
  bool finite = true;
  float* Qo = p.Qw + ((size_t)m * npair + g) * (M2 * M2);
  float* So = p.Sw + ((size_t)m * npair + g) * (M2 * M2);
  if (p.first) {
    if (p.st[m].done || (JDBG(p) & 2)) return;
    floor_m = p.st[m].floor;
    JTS(1);
    const float* Am = p.Pr + (size_t)m * C * C;
    const int PITCH = p.step_d >= 0 ? M2 : M2 + 1;
    for (int e = tid; e < M2 * M2; e += NT) {
      const int r = e / M2, c = e % M2;
      f32x2 v;
      v[0] = Am[(size_t)pair_index<B>(r, bi, bj) * C + pair_index<B>(c, bi, bj)];
      v[1] = r == c ? 1.f : 0.f;
      finite &= fabsf(v[0]) <= 3.0e38f;
      if (r == c) my_dm = fmaxf(my_dm, fabsf(v[0]));
      SQ[r * PITCH + c] = v;
    }
    __syncthreads();
  } else {
    // ---- look-ahead assembly (cross steps only: a segment never starts behind an intra step)
    int g1, h1, g2, h2;
    block_locate(bi, p.step_u, nblk, g1, h1);
    block_locate(bj, p.step_u, nblk, g2, h2);
    const float* S1 = p.Sr + ((size_t)m * npair + g1) * (M2 * M2);
    const float* S2 = p.Sr + ((size_t)m * npair + g2) * (M2 * M2);
    const bool same = g1 == g2;
    // this thread's four image elements that come out of the previous images (requested before the staging loads)
    float sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * NT, r = e / M2, c = e % M2;
      const bool rlo = r < B, clo = c < B;
      const float* src = rlo ? S1 : S2;
      const int rr = (rlo ? h1 : h2) * B + (rlo ? r : r - B);
      const int cc = (clo ? h1 : h2) * B + (clo ? c : c - B);
      sv[i] = (rlo == clo || same) ? src[rr * M2 + cc] : 0.f;
    }
    float* Xs = jsm;                                // [M2][M2 + 1]  tile (g1, g2) of the state before U(step_u)
    float* Q1s = Xs + M2 * (M2 + 1);                // [M2][B + 1]   columns h1 of Q_g1
    float* Q2s = Q1s + M2 * (B + 1);                // [M2][B + 1]   columns h2 of Q_g2
    float* Ws = Q2s + M2 * (B + 1);                 // [B][M2 + 1]   Q_g1[:, h1]^T X
    f32x4 xv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
    const int se = tid * 4, sr = se / M2, sc = se % M2;
    float* Qdst = nullptr;
    if (!same) {
      int b1i, b1j, b2i, b2j;
      block_pair(g1, p.step_u, nblk, b1i, b1j);
      block_pair(g2, p.step_u, nblk, b2i, b2j);
      const float* Pm = p.Pr + (size_t)m * C * C;
      xv = *reinterpret_cast<const f32x4*>(Pm + (size_t)pair_index<B>(sr, b1i, b1j) * C + pair_index<B>(sc, b2i, b2j));
      // Q columns: M2 x B floats per side = NT / 2 float4 (the fragments mt of that half); first half of the block
      // fetches Q_g1, second half Q_g2
      const bool one = tid < NT / 2;
      const int t2 = one ? tid : tid - NT / 2;
      const int f = (one ? h1 : h2) * (NT / 2) + t2;
      int qr, qc;
      qfrag_rc<M2>(f, qr, qc);
      qc -= (one ? h1 : h2) * B;
      qv = *reinterpret_cast<const f32x4*>(p.Qr + ((size_t)m * npair + (one ? g1 : g2)) * (M2 * M2) + (size_t)f * 4);
      Qdst = (one ? Q1s : Q2s) + qr * (B + 1) + qc;
    }
    if (p.st[m].done || (JDBG(p) & 2)) return;        // (block-uniform)
    floor_m = p.st[m].floor;
    JTS(1);
    f32x4 crit = {0.f, 0.f, 0.f, 0.f};
    if (!same) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { Xs[sr * (M2 + 1) + sc + j] = xv[j]; Qdst[j * (B + 1)] = qv[j]; }
      __syncthreads();
      JTS(2);
      const int li = lane & 15, lq = lane >> 4;
      if (wave < (B / 16) * NW) {                   // W = Q_g1[:, h1]^T X   (B x M2)
        const int tr = wave / NW, tj = wave % NW;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int kk = 0; kk < M2; kk += 4)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Q1s[(kk + lq) * (B + 1) + 16 * tr + li], Xs[(kk + lq) * (M2 + 1) + 16 * tj + li], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ws[(16 * tr + 4 * lq + r) * (M2 + 1) + 16 * tj + li] = acc[r];
      }
      __syncthreads();
      if (wave < (B / 16) * (B / 16)) {             // crit = W Q_g2[:, h2]   (B x B)
        const int tr = wave / (B / 16), tc = wave % (B / 16);
#pragma unroll 4
        for (int kk = 0; kk < M2; kk += 4)
          crit = __builtin_amdgcn_mfma_f32_16x16x4f32(Ws[(16 * tr + li) * (M2 + 1) + kk + lq], Q2s[(kk + lq) * (B + 1) + 16 * tc + li], crit, 0, 0, 0);
      }
      __syncthreads();                              // the staging area becomes the image
      JTS(3);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * NT, r = e / M2, c = e % M2;
      if ((r < B) == (c < B) || same) {
        f32x2 v; v[0] = sv[i]; v[1] = r == c ? 1.f : 0.f;
        SQ[e] = v;
        finite &= fabsf(sv[i]) <= 3.0e38f;
        if (r == c) my_dm = fmaxf(my_dm, fabsf(sv[i]));
      }
    }
    if (!same && wave < (B / 16) * (B / 16)) {
      const int tr = wave / (B / 16), tc = wave % (B / 16), li = lane & 15, lq = lane >> 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tr + 4 * lq + r, col = B + 16 * tc + li;
        f32x2 v; v[0] = crit[r]; v[1] = 0.f;
        SQ[row * M2 + col] = v;
        SQ[col * M2 + row] = v;
        finite &= fabsf(crit[r]) <= 3.0e38f;
      }
    }
    __syncthreads();
  }
  JTS(4);
  if (p.step_d >= 0) {
    if (!(JDBG(p) & 4)) jacobi_cross_sets_pw<M2>(reinterpret_cast<unsigned char*>(SQ), reinterpret_cast<unsigned char*>(jsm + 2 * M2 * M2), tid, floor_m, my_off, my_sig, JDBG(p));
    JTS(5);
    for (int e = tid; e < M2 * M2; e += NT) So[e] = SQ[e][0];
    int qr, qc;
    qfrag_rc<M2>(tid, qr, qc);                      // NT float4 = one tile, fragment order
    f32x4 q;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = SQ[(qr + j) * M2 + qc][1];
    *reinterpret_cast<f32x4*>(Qo + (size_t)tid * 4) = q;
    {  // fp16 hi / lo fragment unit `tid`
      constexpr int NCH = M2 / 32;
      const int l16 = tid & 63, part = (tid >> 6) & 1, cc = (tid >> 7) % NCH, mt = (tid >> 7) / NCH;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = SQ[qfrag16_k<M2>(cc, l16 >> 4, j) * M2 + 16 * mt + (l16 & 15)][1];
      half8 hi, lo;
      split_f16x8(x, hi, lo);
      *reinterpret_cast<half8*>(p.Qw16 + ((size_t)m * npair + g) * (2 * M2 * M2) + (size_t)tid * 8) = part ? lo : hi;
    }
  } else {
    constexpr int PITCH = M2 + 1;
    float* DO = jsm + 4 * M2 * PITCH;
    const int cur = jacobi_sets<SWEEP_INTRA, M2, KB>(SQ, DO, tid, floor_m, my_off, my_sig);
    JTS(5);
    for (int e = tid; e < M2 * M2; e += NT) So[e] = SQ[cur * M2 * PITCH + (e / M2) * PITCH + (e % M2)][0];
    int qr, qc;
    qfrag_rc<M2>(tid, qr, qc);
    f32x4 q;
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = SQ[cur * M2 * PITCH + (qr + j) * PITCH + qc][1];
    *reinterpret_cast<f32x4*>(Qo + (size_t)tid * 4) = q;
    {
      constexpr int NCH = M2 / 32;
      const int l16 = tid & 63, part = (tid >> 6) & 1, cc = (tid >> 7) % NCH, mt = (tid >> 7) / NCH;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = SQ[cur * M2 * PITCH + qfrag16_k<M2>(cc, l16 >> 4, j) * PITCH + 16 * mt + (l16 & 15)][1];
      half8 hi, lo;
      split_f16x8(x, hi, lo);
      *reinterpret_cast<half8*>(p.Qw16 + ((size_t)m * npair + g) * (2 * M2 * M2) + (size_t)tid * 8) = part ? lo : hi;
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

// tasks of one matrix: [off-diagonal tiles g < h][diagonal tile copies][V tiles (row block, column pair)]
template <int M2>
__device__ __forceinline__ void jacobi_fused_u(const JacobiFusedArgs& p, int m, int task, float* jsm) {
  constexpr int B = M2 / 2, PITCH = M2 + 1, NW = M2 / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.C, nblk = C / B, npair = nblk / 2;
  const int n_off = npair * (npair - 1) / 2;
  const size_t cc = (size_t)C * C;
  const int e4 = tid * 4, lr = e4 / M2, lc = e4 % M2;     // this thread's float4 of a staged tile
  if (task >= n_off && task < n_off + npair) {            // diagonal tile g: the image D(step_u) left behind
    const int g = task - n_off;
    int gi, gj;
    block_pair(g, p.step_u, nblk, gi, gj);
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.Sr + ((size_t)m * npair + g) * (M2 * M2) + e4);
    *reinterpret_cast<f32x4*>(p.Pw + m * cc + (size_t)pair_index<B>(lr, gi, gj) * C + pair_index<B>(lc, gi, gj)) = v;
    return;
  }
  const bool is_v = task >= n_off;               // (V tasks exist only in launches with_v)
  int g, h;
  if (is_v) { const int t = task - n_off - npair; g = t / npair; h = t % npair; }       // g = M2-row block of V
  else { int t = task; g = 0; while (t >= npair - 1 - g) { t -= npair - 1 - g; ++g; } h = g + 1 + t; }
  int hi, hj, gi = 0, gj = 0;
  block_pair(h, p.step_u, nblk, hi, hj);
  if (!is_v) block_pair(g, p.step_u, nblk, gi, gj);
  float* Xs = jsm;
  float* Qhs = Xs + M2 * PITCH;
  float* Qgs = Qhs + M2 * PITCH;
  {
    const float* X = is_v ? p.V + m * cc : p.Pr + m * cc;
    const int gr = is_v ? g * M2 + lr : pair_index<B>(lr, gi, gj);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(X + (size_t)gr * C + pair_index<B>(lc, hi, hj));
    const f32x4 hv = *reinterpret_cast<const f32x4*>(p.Qr + ((size_t)m * npair + h) * (M2 * M2) + e4);
    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
    if (!is_v) gv = *reinterpret_cast<const f32x4*>(p.Qr + ((size_t)m * npair + g) * (M2 * M2) + e4);
    int qr, qc;
    qfrag_rc<M2>(tid, qr, qc);                      // the rotations arrive in fragment order
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Xs[lr * PITCH + lc + j] = xv[j];
      Qhs[(qr + j) * PITCH + qc] = hv[j];
      if (!is_v) Qgs[(qr + j) * PITCH + qc] = gv[j];
    }
  }
  __syncthreads();
  const int ti = wave / NW, tj = wave % NW, li = lane & 15, lq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (is_v) {
    // (V Qh)^T = Qh^T V^T tile (column tile tj, rows 16 ti ..) exactly as jacobi_vstrip_kernel computes it -- the same
    // split-fp16 operands, k-slots and order of the MFMAs -- so V comes out bit-identical whichever kernel updates it
    constexpr int NCH = M2 / 32;
    const half_t* q16 = p.Qr16 + ((size_t)m * npair + h) * (2 * M2 * M2);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = Xs[(16 * ti + li) * PITCH + qfrag16_k<M2>(c, lq, j)];
      half8 bh, bl;
      split_f16x8(x, bh, bl);
      const half8 ah = *reinterpret_cast<const half8*>(q16 + ((size_t)((tj * NCH + c) * 2 + 0) * 64 + lane) * 8);
      const half8 al = *reinterpret_cast<const half8*>(q16 + ((size_t)((tj * NCH + c) * 2 + 1) * 64 + lane) * 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int kk = 0; kk < M2; kk += 4)      // T = X Qh
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Xs[(16 * ti + li) * PITCH + kk + lq], Qhs[(kk + lq) * PITCH + 16 * tj + li], acc, 0, 0, 0);
  }
  if (!is_v) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) Xs[(16 * ti + 4 * lq + r) * PITCH + 16 * tj + li] = acc[r];
    __syncthreads();
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kk = 0; kk < M2; kk += 4)      // Y = Qg^T T
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Qgs[(kk + lq) * PITCH + 16 * ti + li], Xs[(kk + lq) * PITCH + 16 * tj + li], acc, 0, 0, 0);
    float* Pw = p.Pw + m * cc;
    const int col = pair_index<B>(16 * tj + li, hi, hj);
#pragma unroll
    for (int r = 0; r < 4; ++r) Pw[(size_t)pair_index<B>(16 * ti + 4 * lq + r, gi, gj) * C + col] = acc[r];
    // mirror tile (h, g): this lane's four rows are four consecutive columns there
    *reinterpret_cast<f32x4*>(Pw + (size_t)col * C + pair_index<B>(16 * ti + 4 * lq, gi, gj)) = acc;
  } else {
    // lane (n, q), register r = (V Qh)[row 16 ti + n][column 16 tj + 4 q + r]: four consecutive columns
    float* Vm = p.V + m * cc;
    *reinterpret_cast<f32x4*>(Vm + (size_t)(g * M2 + 16 * ti + li) * C + pair_index<B>(16 * tj + 4 * lq, hi, hj)) = acc;
  }
}

// grid: [nmat * npair pair problems (if has_d)] [ntask * nmat update tasks, task-major (if has_u)]
template <int M2>
__global__ __launch_bounds__((M2 / 2) * (M2 / 2), M2 == 64 ? 8 : 2) void jacobi_fused_kernel(JacobiFusedArgs p) {
  extern __shared__ __attribute__((aligned(16))) float jsm[];
  constexpr int B = M2 / 2;
  const int npair = p.C / B / 2;
  const int n_d = p.has_d ? p.nmat * npair : 0;
  int b = blockIdx.x;
  if (b < n_d) {
    const int m = b / npair, g = b % npair;
    JTS(0);
    jacobi_fused_d<M2>(p, m, g, jsm);               // (checks the matrix' done flag itself, after issuing its loads)
  } else {
    b -= n_d;
    const int task = b / p.nmat, m = b % p.nmat;
    if (p.st[m].done || (JDBG(p) & 1)) return;
    jacobi_fused_u<M2>(p, m, task, jsm);
  }
}

namespace r4 {
template <int M2, int LAY>
static size_t fused_lds(int has_d, int has_u, int first, int step_d) {
  static_assert(lds_bytes<64, 0>(1, 1, 0, 0) * 4 <= 160 * 1024, "four blocks of a {D, U} launch per CU");
  return lds_bytes<M2, LAY>(has_d, has_u, first, step_d);
}
}  // namespace r4


// grid: [nmat * npair pair problems (if has_d)] [ntask * nmat update tasks, task-major (if has_u)]; 256 threads
template <int M2, int LAY>
__global__ __launch_bounds__(r4::Lay<LAY>::NTD, 4) void jacobi_fused4_kernel(JacobiFusedArgs p) {
  extern __shared__ __attribute__((aligned(16))) float jsm[];
  constexpr int B = M2 / 2;
  const int npair = p.C / B / 2;
  const int n_d = p.has_d ? p.nmat * npair : 0;
  int b = blockIdx.x;
  if (b < n_d) {
    // (p.mat_major: matrix index fastest, like the update tasks behind them -- with a multiple of 8 matrices every block that
    //  touches matrix m, in this launch and the next, runs on XCD m % 8 and finds the images, logs and tiles in its L2)
    const int m = p.mat_major ? b % p.nmat : b / npair, g = p.mat_major ? b / p.nmat : b % npair;
    JTS(0);
    if (JDBG(p) & 2) return;                 // (tuning builds, timing experiment: the pair problems exit at once -- results invalid)
    r4::fused_d<M2, LAY>(p, m, g, jsm);
  } else {
    b -= n_d;
    const int task = b / p.nmat, m = b % p.nmat;
    if (p.st[m].done || (JDBG(p) & 1)) return;   // (JDBG 1: the tile updates exit at once)
    r4::fused_u<M2>(p, m, task, jsm);
  }
}

// ---------------------------------------------------------------------------
// V <- V Q for all the rotations of one launch segment, with V resident in REGISTERS (round 3).
// Updating V inside the tile update streams the whole of V through the chip at every outer step (64 % of the update's
// traffic: 4 MB per 512-channel matrix and step), although nothing in the solver reads V.  Here a wave owns 16 rows of V
// -- C/16 tiles of 16x16 held as MFMA accumulators, C/4 registers per lane -- and applies the rotation matrices of all
// the steps of a segment (the Q log the pair problems wrote) without V leaving the registers:
//   (V Q_h)^T = Q_h^T V^T is computed tile by tile with v_mfma_f32_16x16x4_f32; a 16x16 tile of V^T in the C/D layout
//   (lane (n, q), register r  <->  V[row n][tile column 4 q + r]) is ALREADY the B operand of the k-step that covers the
//   columns {4 q + r : q = 0..3}, so the old tiles feed the MFMAs straight from the accumulator registers, the results
//   arrive in the same layout, and the only operand that is loaded is Q (fragment order: one 16-byte LDS read per
//   four MFMAs; the tile of a pair is staged once per block, double-buffered).
// Register indices must be static: the strip is kept in POSITION order of the round-robin pairing (pair g = positions g
// and NBLK-1-g; the intra step pairs positions 2g, 2g+1, natural order = the order of step 0); advancing one outer step
// moves position pos+1 to pos (pos >= 1) and position 1 to NBLK-1 -- a static register rotation.
// V traffic: one read and one write per SEGMENT instead of per step.  Runs on a side stream behind the segment, beside
// the next segment's pair problems (MFMA pipe and registers here, LDS and VALU there).
// ---------------------------------------------------------------------------
struct VStripArgs {
  float* V; const half_t* Qlog;     // Qlog: [slot][nmat][npair][2*M2*M2] fp16 hi/lo fragments (qfrag16), slot = step - step_begin
  const JacobiState* st;
  int C, nmat, step_begin, step_end, seg;
  int dbg;       // timing experiment (WCT_JACOBI_DBG & 128): return at once
  int mat_major; // grid (nmat, strips) instead of (strips, nmat): see vstrip_launch
};

// Round 5: the rotation log reaches the block by LDS-DMA into a RING of VS_RING tiles (buffer_load_dwordx4 ... lds, 1 KiB per wave
// instruction, no registers), VS_RING - 1 tiles ahead of the one in use.  Until round 4 a tile was requested four pairs ahead into
// 64 VGPRs and parked in a two-tile LDS buffer: with ~2 us per access (the log streams from HBM: 64 MB per segment at 64 matrices)
// and ~0.2 us of MFMA work per tile a block waited nine tenths of its life, and while it waits it holds a whole CU (414 registers
// per lane: no pair-problem or update wave fits beside it) -- the V pass cost the solve a quarter of its time at batch 32 although
// it runs on its own stream.  Protocol per tile q: s_waitcnt vmcnt((VS_RING - 2) x NDMA) retires this wave's share of tile q (the
// counter is in order), the barrier makes it everybody's and retires slot (q - 1) % VS_RING, whose refill (tile q + VS_RING - 1)
// is issued at once.  Past the end of the segment the refills re-fetch the last tile into slots nobody reads: the outstanding
// count stays constant and so does the wait.
// Round 5, last change: a block of the pass used to hold its CU alone (128 KB of ring, 398 registers per lane), and the launch that
// opens the next segment waited for it (per-launch trace: 86-117 us for that launch against 25-35 alone).  With a ring of FIVE tiles
// (80 KB) and the fragments of a tile requested one K-chunk ahead instead of all sixteen at once (364 registers) two pair-problem /
// update blocks fit beside it on every CU: eigensolver 12.05 -> 11.55 ms per 32-pair step, 8.44 -> 7.9 at 16 pairs, frames identical
// (profiles/r05_vstrip_coresident.txt: rings of 4 / 5 / 6 / 7 / 8 tiles = 11.86 / 11.55 / 11.60 / 11.78 / 11.88).
constexpr int VS_RING = 5;
template <int M2, int W>
constexpr size_t vstrip_lds_bytes() { return (size_t)VS_RING * M2 * M2 * sizeof(float); }

template <int M2, int NBLK, int W>
__global__ __launch_bounds__(W * 64) void jacobi_vstrip_kernel(VStripArgs p) {
  constexpr int B = M2 / 2, TB = B / 16, NCH = M2 / 32, NPAIR = NBLK / 2, FR = M2 * M2, C = NBLK * B;
  constexpr int TILE_BYTES = FR * 4;                       // fp16 hi / lo fragments of one rotation matrix
  constexpr int NDMA = TILE_BYTES / 1024 / W;              // 1-KiB DMA pieces per wave and tile
  constexpr bool FRAGS_AHEAD = W <= 4;                     // one wave per SIMD: registers to hold a tile's 16 fragments at once
  static_assert(NDMA >= 1 && NDMA * W * 1024 == TILE_BYTES, "a tile is a whole number of 1-KiB pieces per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char vs_ring[];
  const int m = p.mat_major ? blockIdx.x : blockIdx.y;
  if (p.st[m].seg_stop <= p.seg || JDBG(p)) return;   // no rotations of this segment belong to the matrix (done before it began)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int row0 = ((p.mat_major ? blockIdx.y : blockIdx.x) * W + wave) * 16;
  float* Vm = p.V + (size_t)m * C * C + (size_t)(row0 + li) * C + 4 * lq;
  const size_t slot_stride = (size_t)p.nmat * NPAIR * FR;
  const float* qbase = reinterpret_cast<const float*>(p.Qlog) + (size_t)m * NPAIR * FR;   // a tile of fp16 hi/lo fragments = FR * 4 bytes too
  // tile q of the segment: step = step_begin + q / NPAIR, pair = q % NPAIR
  const int ntile = (p.step_end - p.step_begin) * NPAIR;
  const __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)qbase, 0, 0x7FFFFFFF, 0x00020000);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma = [&](int q) {                                  // (uniform) tile min(q, ntile - 1) -> ring slot q % VS_RING
    const int qq = q < ntile ? q : ntile - 1;
    const unsigned soff = (unsigned)(((size_t)(qq / NPAIR) * slot_stride + (size_t)(qq % NPAIR) * FR) * sizeof(float));
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int piece = wave_u + i * W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, reinterpret_cast<__attribute__((address_space(3))) void*>(
          (__attribute__((address_space(3))) unsigned char*)vs_ring + (q % VS_RING) * TILE_BYTES + piece * 1024), 16, lane * 16, soff + piece * 1024, 0, 0);
    }
  };
  f32x4 v[NBLK * TB];
  auto block_at = [&](int pos, int step) { return step < 0 ? pos : rr_idx(pos, step, NBLK); };
#pragma unroll
  for (int pos = 0; pos < NBLK; ++pos) {
    const int blk = block_at(pos, p.step_begin);
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) v[pos * TB + tb] = *reinterpret_cast<const f32x4*>(Vm + blk * B + tb * 16);
  }
  // (the strip's loads are OLDER than the DMAs on the in-order counter: the first tile's wait then covers the strip and tile 0 only)
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q0 = 0; q0 < VS_RING - 1; ++q0) dma(q0);       // tiles 0 .. VS_RING - 2
  int q = 0;
  // one pair (the g-th of its step: g is a compile-time constant once the loops below are unrolled, and so are the
  // positions pa / pb and the register sets g % PD, (g + 1) % PD): tiles x = positions pa, pb;
  // out[mt] = sum over chunks of the three split-operand MFMAs.  The strip must stay in registers.
#define VSTRIP_PAIR(pa, pb, g)                                                                                       \
  {                                                                                                                  \
    /* this wave's pieces of tile q have landed; lgkmcnt(0): its ds_reads of tile q - 1 have RETIRED before the barrier behind which    \
       another wave refills that slot (the fragments were consumed by the previous pair's MFMAs, the wait is free) */       \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((VS_RING - 2) * NDMA) : "memory");                           \
    __builtin_amdgcn_s_barrier();                          /* ... everybody's; and nobody reads tile q - 1 any more */ \
    asm volatile("" ::: "memory");                                                                                   \
    dma(q + VS_RING - 1);                                                                                            \
    const half_t* qb = reinterpret_cast<const half_t*>(vs_ring + (q % VS_RING) * TILE_BYTES);                        \
    f32x4 out[2 * TB];                                                                                               \
    _Pragma("unroll") for (int mt = 0; mt < 2 * TB; ++mt) out[mt] = f32x4{0.f, 0.f, 0.f, 0.f};                      \
    /* The fragments of a K-chunk are requested together, one chunk ahead of the MFMAs that use them (round 5: hipcc \
       read them one at a time, each ds_read_b128 followed by s_waitcnt lgkmcnt(0) and its MFMAs -- sixteen exposed   \
       LDS round trips per pair, and the three MFMAs of an accumulator back to back: 100 us of a block's 115).  Per   \
       accumulator the order of the six products is unchanged (chunk 0: lo.hi, hi.lo, hi.hi; chunk 1 likewise). */   \
    half8 fa[NCH][2 * TB][2];                                                                                        \
    _Pragma("unroll") for (int mt = 0; mt < 2 * TB; ++mt)                                                            \
      _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                         \
        fa[0][mt][part] = *reinterpret_cast<const half8*>(qb + (((mt * NCH + 0) * 2 + part) * 64 + lane) * 8);       \
    if (FRAGS_AHEAD) __builtin_amdgcn_sched_barrier(0);   /* (two waves per SIMD: 256 registers -- let the compiler read just in time) */ \
    half8 bh[NCH], bl[NCH];                                                                                          \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                                \
      /* tiles 2c, 2c+1 of the pair's 2 TB tiles: positions pa (tiles 0..TB-1) and pb (TB..2TB-1) */                 \
      const f32x4 x0 = 2 * c < TB ? v[(pa) * TB + 2 * c] : v[(pb) * TB + 2 * c - TB];                                \
      const f32x4 x1 = 2 * c + 1 < TB ? v[(pa) * TB + 2 * c + 1] : v[(pb) * TB + 2 * c + 1 - TB];                    \
      const float xx[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};                                  \
      split_f16x8(xx, bh[c], bl[c]);                                                                                 \
    }                                                                                                                \
    if (FRAGS_AHEAD) __builtin_amdgcn_sched_barrier(0);                                                              \
    _Pragma("unroll") for (int c = 0; c < NCH; ++c) {                                                                \
      if (c + 1 < NCH) {                                   /* the next chunk's fragments under this chunk's MFMAs */ \
        _Pragma("unroll") for (int mt = 0; mt < 2 * TB; ++mt)                                                        \
          _Pragma("unroll") for (int part = 0; part < 2; ++part)                                                     \
            fa[c + 1][mt][part] = *reinterpret_cast<const half8*>(qb + (((mt * NCH + c + 1) * 2 + part) * 64 + lane) * 8); \
      }                                                                                                              \
      _Pragma("unroll") for (int mt = 0; mt < 2 * TB; ++mt) out[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[c][mt][1], bh[c], out[mt], 0, 0, 0); \
      _Pragma("unroll") for (int mt = 0; mt < 2 * TB; ++mt) out[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[c][mt][0], bl[c], out[mt], 0, 0, 0); \
      _Pragma("unroll") for (int mt = 0; mt < 2 * TB; ++mt) out[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[c][mt][0], bh[c], out[mt], 0, 0, 0); \
      if (FRAGS_AHEAD) __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                                                \
    if (FRAGS_AHEAD) __builtin_amdgcn_sched_barrier(0);                                                              \
    _Pragma("unroll") for (int t = 0; t < TB; ++t) { v[(pa) * TB + t] = out[t]; v[(pb) * TB + t] = out[TB + t]; }    \
    ++q;                                                                                                             \
  }
  // (the loops over g are spelled out with compile-time g: `#pragma unroll` alone leaves g a variable inside the macro)
  auto step_pairs = [&](auto INTRA) {
    constexpr bool intra = decltype(INTRA)::value;
#define VP(G) if constexpr ((G) < NPAIR) { if constexpr (intra) VSTRIP_PAIR(2 * (G), 2 * (G) + 1, G) else VSTRIP_PAIR(G, NBLK - 1 - (G), G) }
    VP(0) VP(1) VP(2) VP(3) VP(4) VP(5) VP(6) VP(7)
#undef VP
  };
#pragma unroll 1
  for (int step = p.step_begin; step < p.step_end; ++step) {
    if (step < 0) {
      step_pairs(std::true_type{});          // natural order
    } else {
      step_pairs(std::false_type{});
      if (step + 1 < p.step_end && NBLK > 2) {        // positions of the next step: pos <- pos + 1, NBLK - 1 <- 1
        f32x4 keep[TB];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) keep[tb] = v[1 * TB + tb];
#pragma unroll
        for (int pos = 1; pos < NBLK - 1; ++pos)
#pragma unroll
          for (int tb = 0; tb < TB; ++tb) v[pos * TB + tb] = v[(pos + 1) * TB + tb];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) v[(NBLK - 1) * TB + tb] = keep[tb];
      }
    }
  }
#undef VSTRIP_PAIR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills past the end are still landing in this block's LDS
  const int last = p.step_end - 1;
#pragma unroll
  for (int pos = 0; pos < NBLK; ++pos) {
    const int blk = block_at(pos, last);
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) *reinterpret_cast<f32x4*>(Vm + blk * B + tb * 16) = v[pos * TB + tb];
  }
}

// after a solve: matrices whose final state sits in the second buffer are copied back into A
__global__ __launch_bounds__(256) void jacobi_gather_kernel(float* A, const float* P1, const JacobiState* st, int C, int cur_final) {
  const int m = blockIdx.y;
  const int buf = st[m].done ? st[m].pad : cur_final;
  if (buf == 0) return;
  const size_t cc = (size_t)C * C, n4 = cc / 4;
  const f32x4* src = reinterpret_cast<const f32x4*>(P1 + m * cc);
  f32x4* dst = reinterpret_cast<f32x4*>(A + m * cc);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

constexpr int JACOBI_ROWSUM_MAX = 1024;                // words of row sums per matrix (C <= 1024; see JACOBI_ROW_K below)
// mat0 = index of the group's first matrix in the 2P batch; skipped style matrices start out `done`
__global__ __launch_bounds__(256) void jacobi_init_kernel(const float* A, float* V, JacobiState* st, int C, int mat0, int shared_style, unsigned* rowsum) {
  __shared__ float red[4];
  const int m = blockIdx.y;
  const bool skip = skip_style_mat(mat0 + m, shared_style);
  const size_t cc = (size_t)C * C;
  if (!skip)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cc; i += (size_t)gridDim.x * blockDim.x)
      V[(size_t)m * cc + i] = (i / C == i % C) ? 1.f : 0.f;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < C; i += 256) rowsum[(size_t)m * JACOBI_ROWSUM_MAX + i] = 0u;      // (jacobi_resid_kernel adds, jacobi_check_kernel clears)
    float mx = 0.f;                          // NaN diagonals drop out of fmaxf; the pair kernels flag them
    if (!skip) for (int i = threadIdx.x; i < C; i += 256) mx = fmaxf(mx, fabsf(A[(size_t)m * cc + (size_t)i * C + i]));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      JacobiState z;
      z.offmax = 0u; z.done = skip ? 1 : 0; z.sweeps = 0; z.offsig = 0u;
      z.floor = JACOBI_SIG_FLOOR * fminf(mx, 3.0e38f); z.last_sig = 0u; z.dmax = 0u; z.r2 = -1.f; z.r2l = -1.f; z.pad = 0;
      z.seg_stop = skip ? 0 : 0x7fffffff; z.pad2 = 0;
      st[m] = z;
    }
  }
}

// State of a matrix AFTER a sweep, measured on the matrix itself: means of squared cosines e^2 / (d_p d_q) of the
// off-diagonal residual E.  A squared cosine is at once the relative perturbation E still causes in the smaller
// eigenvalue of the pair (what decides on which side of the reference's 1e-5 cut-off it falls, and how accurate its
// gain is) and the size of the first-order term F o E of a spectral function against f(D) (spectral_matrix_kernel).
//   STRICT (decides when to stop; what the reference's spectral functions need): both diagonals above the cut-off:
//     the squared cosine; one above, one below: the squared angle e^2 / big^2 plus the contamination of the dropped
//     direction, 0.01 e^2 / (big 1e-5) (it has to stay below the cut-off); both below: nothing (both are dropped).
//     Normalised by the number of kept diagonals.
//   LENIENT (judges the status when the sweep budget runs out): the same with "inside the rounding noise of a matrix
//     of this norm" (below `floor`) in the place of "dropped": cosines against noise diagonals never settle -- N < C
//     pixels at a feature scale whose noise exceeds 1e-5 -- and do not matter (the content has no energy there).
// partial[m][chunk][0..3] = strict sum (half the sum over the ordered pairs of the chunk's rows), kept diagonals,
// lenient sum, significant diagonals.  Fixed thread -> element map and reduction order: bit-reproducible, so the
// sweep count -- and with it every output bit -- does not depend on scheduling.
// mixed_w: weight of the kept-x-dropped couplings in the STRICT measure.  The second-order completion covers the block of
// kept eigenvalues only, so when the stop threshold is raised for it (4e-2 instead of 1.5e-2) the couplings across the
// cut-off keep their old bound: their squared measure is weighted by (4 / 1.5)^2.
// Round 4: the measurement walks 64 x 64 TILES of the upper triangle (the matrix is symmetric: an off-diagonal tile counts for
// its mirror image as well, half the bytes and half the arithmetic of the row-chunk walk it replaces; the two diagonal segments
// of a tile are 128 gathered values instead of the whole diagonal per block).  partial[m][tile][0..5] = strict kept-x-kept sum,
// strict kept-x-dropped sum (weighted in jacobi_check_kernel: the `near` flag needs the whole diagonal), kept diagonals, lenient
// sum, significant diagonals, near flag; jacobi_check_kernel (one wave per matrix) sums them over the tiles with a fixed lane ->
// tile map and a fixed tree.  (Letting the block that finishes a matrix last evaluate the test -- a counter and a device-scope
// fence per block -- saves the second launch and cost 2.5 ms per 32-pair step: the fence writes the XCD's L2 back, and the
// tiles the update has just written are still dirty in it.)
constexpr int JACOBI_RESID_T = 64;
constexpr int JACOBI_RESID_MAXTILES = 136;             // C <= 1024: 16 x 17 / 2
constexpr int JACOBI_RESID_STRIDE = 8;                 // floats per (matrix, tile)
// Round 6: the stop test also bounds the WORST ROW.  r2 is a mean over the rows of a row's sum of squared cosines; the completion's
// error in ONE eigen-direction is set by that direction's own row, and a mean over C rows hides a row C times above it -- graded
// spectra do that (the rows of the smallest eigenvalues settle last): C = 64, 72 pixels, eigenvalues 4.5e3 .. 3.5e-2 stopped at
// sweep 5 with the mean under the threshold and the transform 1.35e-3 off; one sweep later 4e-6 (wide fuzz run,
// profiles/r06_fuzz_wide.txt).  Row sums: every tile adds its 64 row and 64 column sums into rowsum[m][C] as 2^-20 fixed point
// with integer atomics (associative: the sums, and with them the sweep counts and every output bit, do not depend on the order),
// jacobi_check_kernel takes the maximum, clears the words, and a matrix is done only if max_p rowsum_p / 2 < JACOBI_ROW_K tol^2 too.
// (a contribution is capped at 32 = 2^25 words: at most 2 x 16 tiles add to a row, so a word cannot wrap)
constexpr float JACOBI_ROW_K = 8.f;
constexpr float JACOBI_ROW_FIX = 1048576.f;            // 2^20
struct JacobiCheckArgs {
  int mid;                   // 1: the test in the middle of a sweep (jacobi_check_mid)
  float tol_max, tol_fn;
  int buf, segs, lenient_from;
  float row_k;               // JACOBI_ROW_K (tuning builds: WCT_JACOBI_ROW_K)
};

// done: 0 = still rotating, 1 = converged, 2 = failed (a non-finite element reached a pair problem).
// Converged = the sweep saw no rotated pair above tol_max (the classical test: the matrix was already diagonal to
// tol_max BEFORE the sweep), or -- tol_fn > 0 -- the residual measured after the sweep is below tol_fn: the matrix
// function built from this state with the first-order completion is then accurate to O(tol_fn^2).
// lenient_from: from this many completed sweeps on, a matrix whose SIGNIFICANT pairs were all below tol_max in the sweep
// is done as well (what jacobi_finalize_kernel accepts when the budget runs out): the noise-level pairs of a
// rank-deficient matrix never settle, and without this such a matrix always burns the whole sweep budget.
__device__ __forceinline__ void jacobi_check_end(JacobiState& s, const float (&v)[5], const JacobiCheckArgs& a) {
  s.sweeps += 1;
  const float r2 = v[0] / fmaxf(v[1], 1.f);
  const bool rows_ok = v[4] < a.row_k * a.tol_fn * a.tol_fn;
  s.r2 = r2;
  s.r2l = v[2] / fmaxf(v[3], 1.f);
  const unsigned bits = s.offmax;                       // max of non-negative floats as bit patterns; >= 0x7f800000: inf / NaN
  if (bits >= 0x7f800000u || !(r2 < 3.0e38f)) s.done = 2;
  else if (__uint_as_float(bits) < a.tol_max || (a.tol_fn > 0.f && r2 < a.tol_fn * a.tol_fn && rows_ok)) s.done = 1;
  else if (s.sweeps >= a.lenient_from && __uint_as_float(s.offsig) < a.tol_max) s.done = 1;
  if (s.done) { s.pad = a.buf; s.seg_stop = a.segs; }
  s.last_sig = s.offsig;
  s.floor = fmaxf(s.floor, JACOBI_SIG_FLOOR * __uint_as_float(s.dmax));
  s.offmax = 0u;
  s.offsig = 0u;
  s.dmax = 0u;
}
// The same residual test in the MIDDLE of a sweep (tol_fn callers only): near the end the residual halves every third
// of a sweep (tools/jacobi_block_order_proto.py), so a matrix that a full sweep would take 10x below the stop threshold
// is usually below it half a sweep earlier; its remaining launches of the sweep turn into no-ops.  Touches nothing but
// `done`, r2 / r2l and the sweep count (the half sweep counts as one).
__device__ __forceinline__ void jacobi_check_mid(JacobiState& s, const float (&v)[5], const JacobiCheckArgs& a) {
  const float r2 = v[0] / fmaxf(v[1], 1.f);
  if (s.offmax < 0x7f800000u && r2 < a.tol_fn * a.tol_fn && v[4] < a.row_k * a.tol_fn * a.tol_fn) {       // finite so far and converged
    s.r2 = r2;
    s.r2l = v[2] / fmaxf(v[3], 1.f);
    s.last_sig = s.offsig;
    s.sweeps += 1;
    s.done = 1;
    s.pad = a.buf;
    s.seg_stop = a.segs;
  }
}

// grid (tiles of the upper triangle, matrices)
__global__ __launch_bounds__(256) void jacobi_resid_kernel(const float* A, const JacobiState* st, float* partial, int C, unsigned* rowsum, float mixed_w) {
  constexpr int T = JACOBI_RESID_T;
  __shared__ float dr[T], dc[T], ir[T], ic[T];
  __shared__ float red[5][4];
  __shared__ float colred[4][T];
  const int m = blockIdx.y, tid = threadIdx.x;
  if (st[m].done) return;                                // (jacobi_check_kernel skips the matrix as well)
  const int ntr = (C + T - 1) / T;
  int ti = 0, t = blockIdx.x;
  while (t >= ntr - ti) { t -= ntr - ti; ++ti; }
  const int tj = ti + t;
  const bool diag = ti == tj;
  const float* Am = A + (size_t)m * C * C;
  const float floor_m = st[m].floor;
  if (tid < 2 * T) {
    const int i = (tid < T ? ti * T + tid : tj * T + tid - T);
    const float d = i < C ? fabsf(Am[(size_t)i * C + i]) : 0.f;
    (tid < T ? dr : dc)[tid & (T - 1)] = d;
    (tid < T ? ir : ic)[tid & (T - 1)] = d > 0.f ? 1.f / d : 0.f;
  }
  __syncthreads();
  // v: kept x kept, kept x dropped, kept diagonals, lenient, significant diagonals; near flag beside them
  float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int near = 0;
  if (diag && tid < T && ti * T + tid < C) {
    const float d = dr[tid];
    v[2] = d > 1e-5f ? 1.f : 0.f;
    v[4] = d > floor_m ? 1.f : 0.f;
    // an eigenvalue within half a decade of the cut-off: its kept / dropped side can still change, and a pair of the kept
    // block may really be a pair across the cut-off -- such a matrix keeps the old bound on ALL its couplings
    near = (d > 3.3e-6f) & (d < 3e-5f);
  }
  // an ordered pair (p, q) weighs 1/2: an off-diagonal tile stands for its mirror image too
  const float w = diag ? 0.5f : 1.f;
  const int c4 = (tid & 15) * 4, gq = tj * T + c4;
  float sr[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};     // this thread's part of its 4 rows' / 4 columns' strict sums
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = (tid >> 4) + 16 * k, gp = ti * T + r;
    if (gp < C && gq < C) {                              // (C is a multiple of 32: a group of four columns is inside or outside)
      const f32x4 e4 = *reinterpret_cast<const f32x4*>(Am + (size_t)gp * C + gq);
      const float dp = dr[r], ip = ir[r];
      const bool kp = dp > 1e-5f, sp = dp > floor_m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dq = dc[c4 + j], iq = ic[c4 + j];
        const bool kq = dq > 1e-5f, sq = dq > floor_m;
        const bool bigp = dp >= dq;
        const float small = bigp ? dq : dp, ibig = bigp ? ip : iq;
        const float e2 = (diag && r == c4 + j) ? 0.f : w * e4[j] * e4[j];
        const float cos2 = e2 * ip * iq;
        const float mixed = e2 * ibig * (ibig + (small < 1e-5f ? 0.01f * 1e5f : 0.f));
        // Round 6: ACROSS the cut-off f jumps, and what the residual does there is rotate the kept direction into the dropped one by
        // e / (d_k - d_d), not e / d_k: with a cluster of eigenvalues around 1e-5 (spacing of a few per cent: 8 of a 128-channel
        // covariance within +-20 % in the wide fuzz run, profiles/r06_fuzz_wide.txt) the old measure let the sweeps stop at angles of
        // 0.2 .. 0.4 between neighbours on either side, 1.1e-3 .. 1.4e-2 of the transform.  The gap is floored at 1e-3 of the larger
        // diagonal (closer pairs float32 cannot tell apart: either side is the reference's).
        const float big = bigp ? dp : dq;
        const float igap = __builtin_amdgcn_rcpf(fmaxf(big - small, 1e-3f * big));
        const float across = e2 * (igap * igap + ibig * (small < 1e-5f ? 0.01f * 1e5f : 0.f));
        v[0] += (kp & kq) ? cos2 : 0.f;
        v[1] += (kp ^ kq) ? across : 0.f;
        const float se = (kp & kq) ? cos2 : ((kp ^ kq) ? mixed_w * across : 0.f);
        sr[k] += se;
        sc[j] += se;
        v[3] += (sp & sq) ? cos2 : ((sp | sq) ? mixed : 0.f);
      }
    }
  }
  const int near_any = __syncthreads_or(near);
  // row sums: the 16 lanes of a row group hold a row's 64 columns; column sums: 4 rows per thread, 4 row groups per wave, 4 waves.
  // (a diagonal tile weighs an ordered pair 1/2 and holds both orders: its row and column sums add up to the whole)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float x = sr[k];
    for (int o = 1; o < 16; o <<= 1) x += __shfl_xor(x, o, 64);
    const int gp = ti * T + (tid >> 4) + 16 * k;
    if ((tid & 15) == 0 && gp < C) atomicAdd(rowsum + (size_t)m * JACOBI_ROWSUM_MAX + gp, (unsigned)(fminf(x, 32.f) * JACOBI_ROW_FIX + 0.5f));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = sc[j];
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    if ((tid & 63) < 16) colred[tid >> 6][c4 + j] = x;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
    if ((tid & 63) == 0) red[j][tid >> 6] = v[j];
  }
  __syncthreads();
  if (tid < T && tj * T + tid < C) {
    const float x = (colred[0][tid] + colred[1][tid]) + (colred[2][tid] + colred[3][tid]);
    atomicAdd(rowsum + (size_t)m * JACOBI_ROWSUM_MAX + tj * T + tid, (unsigned)(fminf(x, 32.f) * JACOBI_ROW_FIX + 0.5f));
  }
  float* out = partial + ((size_t)m * JACOBI_RESID_MAXTILES + blockIdx.x) * JACOBI_RESID_STRIDE;
  if (tid < 5) out[tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
  if (tid == 5) out[5] = near_any ? 1.f : 0.f;
}

// grid (matrices), one wave: the measurement's sums over the tiles, then the test (ck.mid: the one in the middle of a sweep)
__global__ __launch_bounds__(64) void jacobi_check_kernel(JacobiState* st, const float* partial, int ntile, float mixed_w, JacobiCheckArgs ck,
                                                          int* done_host /* this group's mapped host words, or null */, unsigned* rowsum, int C) {
  const int m = blockIdx.x, tid = threadIdx.x;
  if (st[m].done) { if (tid == 0 && done_host) done_host[m] = st[m].done; return; }
  unsigned rmax = 0u;                                   // the worst row's sum (fixed point), and the words cleared for the next measurement
  for (int p = tid; p < C; p += 64) {
    unsigned* w = rowsum + (size_t)m * JACOBI_ROWSUM_MAX + p;
    rmax = max(rmax, *w);
    *w = 0u;
  }
  for (int o = 32; o > 0; o >>= 1) rmax = max(rmax, (unsigned)__shfl_xor((int)rmax, o, 64));
  float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* pm = partial + (size_t)m * JACOBI_RESID_MAXTILES * JACOBI_RESID_STRIDE;
  for (int tt = tid; tt < ntile; tt += 64)
#pragma unroll
    for (int j = 0; j < 6; ++j) a[j] += pm[tt * JACOBI_RESID_STRIDE + j];
#pragma unroll
  for (int j = 0; j < 6; ++j)
    for (int o = 32; o > 0; o >>= 1) a[j] += __shfl_xor(a[j], o, 64);
  if (tid == 0) {
    const bool near_m = a[5] > 0.f;
    // remembered for jacobi_finalize_kernel: such a matrix has only the first-order completion to pay for a late acceptance
    st[m].pad2 = near_m ? 1 : 0;
    const float vv[5] = {(near_m ? mixed_w : 1.f) * a[0] + mixed_w * a[1], a[2], a[3], a[4], 0.5f * (float)rmax * (1.f / JACOBI_ROW_FIX)};
    if (ck.mid) jacobi_check_mid(st[m], vv, ck); else jacobi_check_end(st[m], vv, ck);
    if (done_host) done_host[m] = st[m].done;
  }
}

// End of a solve.  A matrix still rotating after the last allowed sweep has FAILED only if its last sweep still saw a
// significant pair above the tolerance; noise-level pairs alone (numerical null space of a rank-deficient covariance:
// N < C pixels, dead or duplicated channels) never settle and do not matter.
// sweeps_out[m] = sweeps used if matrix m is good, -sweeps if it failed to converge, -1000 - sweeps for non-finite
// input; fail[0] += matrices not converged, fail[1] += non-finite ones.
// fail: this group's slot of the caller's status words (host memory mapped into the device, one slot per stream
// group so that plain read-modify-writes of one thread suffice), read by the caller after its next stream sync --
// a failed solve is never silently dropped.
// stats (or null): this group's slot of the caller's solver statistics (host memory mapped into the device):
// [0] matrices solved, [1] sum of their sweeps, [2] the largest sweep count -- wct_eig_stats reads and clears them
__global__ void jacobi_finalize_kernel(const JacobiState* st, int* sweeps_out, int nmat, float tol_max, float tol_fn, volatile int* fail, int debug,
                                       volatile int* stats = nullptr) {
  const int m = threadIdx.x;
  int d = m < nmat ? st[m].done : 1;
  if (debug && m < nmat)
    printf("jacobi m=%d done=%d sweeps=%d last_sig=%.3e r2=%.3e r2l=%.3e floor=%.3e\n", m, st[m].done, st[m].sweeps, __uint_as_float(st[m].last_sig), st[m].r2, st[m].r2l, st[m].floor);
  // out of sweeps: good enough after all if the last sweep's significant pairs were below tol_max, or if the lenient
  // residual is within 4x of the target (second-order error 16x the target's: still inside the 1e-3 budget)
  // (second-order completion: error ~ r^3, and its stop threshold is already 4e-2 -- twice that is the most the 1e-3 budget takes)
  // -- except for a matrix with an eigenvalue near the cut-off (pad2, jacobi_resid_kernel): it keeps the first-order completion
  // and with it the first-order bound 4 x 1.5e-2
  const bool near_cut = m < nmat && st[m].pad2 != 0;
  const float tol_l = tol_fn > 0.f ? (tol_fn > 2e-2f ? (near_cut ? 4.f * 1.5e-2f : 2.f * tol_fn) : 4.f * tol_fn) : tol_max;
  if (d == 0 && (__uint_as_float(st[m].last_sig) < tol_max || (st[m].r2l >= 0.f && st[m].r2l < tol_l * tol_l))) d = 1;
  if (m < nmat && sweeps_out) sweeps_out[m] = d == 1 ? st[m].sweeps : (d == 2 ? -1000 - st[m].sweeps : -st[m].sweeps);
  const int n_open = __builtin_popcountll(__ballot(d == 0)), n_nan = __builtin_popcountll(__ballot(d == 2));
  if (m == 0 && fail) {
    if (n_open) fail[0] = fail[0] + n_open;
    if (n_nan) fail[1] = fail[1] + n_nan;
  }
  if (stats) {
    int sw = m < nmat ? st[m].sweeps : 0;                  // 0: a matrix that was never rotated (shared style)
    int cnt = sw > 0 ? 1 : 0, mx = sw;
    for (int o = 32; o > 0; o >>= 1) { sw += __shfl_xor(sw, o, 64); cnt += __shfl_xor(cnt, o, 64); mx = max(mx, __shfl_xor(mx, o, 64)); }
    if (m == 0) { stats[0] = stats[0] + cnt; stats[1] = stats[1] + sw; if (mx > stats[2]) stats[2] = mx; }
  }
}

constexpr int JACOBI_MAX_SWEEPS = 16;   // round 3: 12 -> 16 (graded rank-deficient 512-channel spectra used 10-12; a failure stays loud)
// WCT_JACOBI_MAX_SWEEPS (read at every solve): LOWERS the sweep budget so that the non-convergence path can be tested.  It
// cannot raise it: a larger budget moves `lenient_from` and with it the sweep counts -- the output bits -- of slowly
// converging matrices; only a tuning build accepts values above the compiled budget.
static int jacobi_max_sweeps() {
  const char* e = getenv("WCT_JACOBI_MAX_SWEEPS");
  const int n = e ? atoi(e) : JACOBI_MAX_SWEEPS;
#ifdef WCT_TUNING
  constexpr int cap = 30;
#else
  constexpr int cap = JACOBI_MAX_SWEEPS;
#endif
  return n < 1 ? 1 : (n > cap ? cap : n);
}

size_t jacobi_workspace_bytes(int C, int nmat) {
  // per matrix: the rotation log of two launch segments (2 x C/B steps x C/M2 tiles x M2^2 = 4 C^2 floats for both block
  // widths) and the same again as fp16 hi/lo fragments, two generations of rotated pair problems (2 x C x M2 <= 128 C), the second matrix buffer (C^2); then state
  // words and residual partials
  return (size_t)nmat * ((size_t)9 * C * C + (size_t)128 * C) * sizeof(float) + 1024 +
         (size_t)nmat * (sizeof(JacobiState) + JACOBI_RESID_MAXTILES * JACOBI_RESID_STRIDE * sizeof(float) + JACOBI_ROWSUM_MAX * sizeof(unsigned));
}

// One group = a set of matrices on its own stream (the two halves of a batch run as two groups so
// that one half's latency-bound pair problems hide under the other half's chip-wide tile update).
struct JacobiGroup {
  float* A; float* V; int nmat; float* Qbuf; JacobiState* st; float* resid; hipStream_t stream; int* sweeps_out;
  half_t* Qlog16[2];                         // the same rotation logs as fp16 hi/lo fragments (V <- V Q)
  float* Qlog[2]; float* Sb[2]; float* P[2]; // look-ahead path: rotation logs of two segments, rotated pair problems of two
                                             // consecutive steps, the two matrix buffers (P[0] = A)
  int cur, par, lg, segs;                    // buffer holding the matrices; generation of the last pair problems; log in use;
                                             // segments completed
  int* done_host = nullptr;                  // the group's slot of JacobiHost::flags_dev (jacobi_check_kernel), or null
  int u_f16 = 0;                             // tile updates on split fp16 (the batched transform path) or fp32 MFMA (wct_eigh, style-swap)
  int vstrip;                                // V is updated per segment by jacobi_vstrip_kernel on the side stream `vs`
  hipStream_t vs; hipEvent_t ev_seg, ev_v[2];// main -> side (segment enqueued), side -> main (log buffer free again)
  bool v_busy[2];
  int mat0, shared_style;      // position in a WCT batch (skip_style_mat); 0, 0 for a plain batch
  float tol_fn;                // > 0: also stop on the measured residual (callers that complete f(A) to first order)
  int* fail;                   // device view of this group's slot [2] of the caller's status words, or null
  int* stats;                  // device view of this group's statistics slot [3] for matrices of this size class, or null
};

// per host thread (= per ctx user): pinned copies of the groups' convergence flags and one event per group
// (events belong to the device that was current when they were created: one set per device, so that a thread driving
//  contexts on several GPUs never records an event of GPU 0 on a stream of GPU 1 -- ADVICE r2)
// flags: the `done` words of the groups' matrices, host memory MAPPED into the device ([4 groups][64]): jacobi_check_kernel stores
// them there itself, and the host reads them after the sweep's event -- up to round 5 a device-to-host copy of the JacobiState
// array per group and sweep carried them (45 copy kernels of ~5 us per step in the solver's launch trains, at every batch size)
struct JacobiHost { int* flags; int* flags_dev; hipEvent_t ev[4]; hipStream_t vs[4]; hipEvent_t ev_seg[4], ev_v[4][2]; };
static JacobiHost* jacobi_host() {
  constexpr int MAXDEV = 16;
  static thread_local JacobiHost hs[MAXDEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  JacobiHost& h = hs[dev];
  if (!h.flags) {
    if (hipHostMalloc((void**)&h.flags, 4 * 64 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h.flags_dev, h.flags, 0) != hipSuccess) { h.flags = nullptr; return nullptr; }
    bool ok = true;
    // (tuning builds: WCT_JACOBI_VPRIO = 1 / 2 puts the V-pass streams at the lowest / highest stream priority.  Measured, either
    //  way: the eigensolver 12.2 -> 22 ms per 32-pair step, nothing at 8 pairs where the V pass does not run -- profiles/r05_vprio.txt)
    static const int vprio = tune_int("WCT_JACOBI_VPRIO", 0);
    int prio_lo = 0, prio_hi = 0;
    if (vprio && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) prio_lo = prio_hi = 0;
    const int prio = vprio == 1 ? prio_lo : vprio == 2 ? prio_hi : 0;
    for (int g = 0; g < 4 && ok; ++g) {
      ok = hipEventCreateWithFlags(&h.ev[g], hipEventDisableTiming) == hipSuccess &&
           (vprio ? hipStreamCreateWithPriority(&h.vs[g], hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(&h.vs[g], hipStreamNonBlocking)) == hipSuccess &&
           hipEventCreateWithFlags(&h.ev_seg[g], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&h.ev_v[g][0], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&h.ev_v[g][1], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { (void)hipHostFree(h.flags); h.flags = nullptr; return nullptr; }
  }
  return &h;
}

// Pair-problem kernel of the 64-wide block pairs: 1 = strips over 4 waves (256 threads; the product), 2 = strips over 8 waves
// (512 threads), 0 = the round-3 kernel (LDS-resident {S, Q} image, 1024 threads).  A-B switch of a tuning build
// (WCT_JACOBI_R4); the product build compiles mode 1 only.  Measured (profiles/r04_jacobi_ab.txt), eigensolver ms per step at
// batch 32 / 8 / 1: mode 1 14.4 / 7.6 / 6.0, mode 2 16.2 / 7.7 / 6.1 (same per-SIMD instruction load, more exchange
// instructions), mode 0 17.1 / 8.0 / 6.4.
static int jacobi_r4_mode() {
  static const int on = tune_int("WCT_JACOBI_R4", 1);
  return on;
}

// ---- look-ahead orchestration -------------------------------------------------------------------------------------
// launch { D(step_d) writing log slot step_d - seg_begin, U(step_u) reading slot step_u - seg_begin }
template <int M2>
static void jacobi_fused_launch(JacobiGroup& G, int C, int seg_begin, bool has_d, int step_d, bool has_u, int step_u, bool first) {
  constexpr int B = M2 / 2, NT = B * B;
  const int npair = C / B / 2;
  const int ntask = npair * (npair - 1) / 2 + npair + (G.vstrip ? 0 : npair * npair);
  const size_t slot = (size_t)G.nmat * npair * M2 * M2;
  JacobiFusedArgs a;
  a.Pr = G.P[G.cur]; a.Pw = G.P[G.cur ^ 1]; a.V = G.V;
  a.Qr = G.Qlog[G.lg] + (size_t)(step_u - seg_begin) * slot; a.Qw = G.Qlog[G.lg] + (size_t)(step_d - seg_begin) * slot;
  a.Qr16 = G.Qlog16[G.lg] + (size_t)(step_u - seg_begin) * slot * 2; a.Qw16 = G.Qlog16[G.lg] + (size_t)(step_d - seg_begin) * slot * 2;
  a.Sr = G.Sb[G.par]; a.Sw = G.Sb[G.par ^ 1];
  a.st = G.st; a.C = C; a.nmat = G.nmat; a.step_d = step_d; a.step_u = step_u;
  a.has_d = has_d; a.has_u = has_u; a.first = first; a.with_v = !G.vstrip; a.u_f16 = G.u_f16;
  static const int xcd = tune_int("WCT_JACOBI_XCD", 1);
  a.mat_major = (xcd & 2) && M2 == 64 && G.nmat % 8 == 0;
  static const int dbg = tune_int("WCT_JACOBI_DBG", 0);
  a.dbg = dbg;
  const unsigned grid = (has_d ? G.nmat * npair : 0) + (has_u ? G.nmat * ntask : 0);
  // round 4 (M2 = 64): pair problems resident in registers (namespace r4, csrc/jacobi_dev.h); the 32-wide block pairs of
  // C <= 128 keep the round-3 kernel
  const int r4m = M2 == 64 ? jacobi_r4_mode() : 0;
  if constexpr (M2 == 64) {
#ifdef WCT_TUNING
    if (r4m == 2) hipLaunchKernelGGL((jacobi_fused4_kernel<M2, 1>), dim3(grid), dim3(r4::Lay<1>::NTD), (r4::fused_lds<M2, 1>(has_d, has_u, first, step_d)), G.stream, a);
    else if (r4m == 0) hipLaunchKernelGGL((jacobi_fused_kernel<M2>), dim3(grid), dim3(NT), jacobi_fused_lds<M2>(has_d, has_u, first, step_d), G.stream, a);
    else
#endif
    hipLaunchKernelGGL((jacobi_fused4_kernel<M2, 0>), dim3(grid), dim3(r4::Lay<0>::NTD), (r4::fused_lds<M2, 0>(has_d, has_u, first, step_d)), G.stream, a);
    LAUNCH_NOTE("jacobi_fused4_kernel");
  } else {
    hipLaunchKernelGGL((jacobi_fused_kernel<M2>), dim3(grid), dim3(NT), jacobi_fused_lds<M2>(has_d, has_u, first, step_d), G.stream, a);
    LAUNCH_NOTE("jacobi_fused_kernel");
  }
#ifdef JACOBI_TS
  static const int ts_intra = getenv("WCT_TS_INTRA") != nullptr;      // (-DJACOBI_TS builds only) summarise the intra steps instead
  if (has_d && M2 == 64 && (ts_intra ? step_d < 0 : (!first && step_d >= 0))) {
    static int nlaunch = 0, last_nmat = 0;
    static double sum[8] = {0}, span = 0;
    if (G.nmat != last_nmat) { nlaunch = 0; span = 0; for (double& v : sum) v = 0; last_nmat = G.nmat; }
    (void)hipStreamSynchronize(G.stream);
    static unsigned long long host[8192 * 10];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(jac_ts), sizeof(host));
    const int nb = G.nmat * npair < 8192 ? G.nmat * npair : 8192;
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < nb; ++i) {
      const unsigned long long* t = host + i * 10;
      for (int k = 0; k < 7; ++k) sum[k] += (double)(t[k + 1] - t[k]) / nb;
      if (t[8] < lo) lo = t[8];
      if (t[9] > hi) hi = t[9];
    }
    if (nlaunch == 0) {
      int occ = 0;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, jacobi_fused4_kernel<64, 0>, r4::Lay<0>::NTD, r4::fused_lds<64, 0>(has_d, has_u, first, step_d));
      printf("jacobi_ts: occupancy API says %d blocks of %d threads per CU with %zu B of LDS\n", occ, r4::Lay<0>::NTD, r4::fused_lds<64, 0>(has_d, has_u, first, step_d));
    }
    span += (double)(hi - lo);
    if (nlaunch == 40) {
      std::vector<unsigned long long> st0, en0;
      for (int i = 0; i < nb; ++i) { st0.push_back(host[i * 10 + 8] - lo); en0.push_back(host[i * 10 + 9] - lo); }
      std::sort(st0.begin(), st0.end()); std::sort(en0.begin(), en0.end());
      printf("jacobi_ts nmat %d launch 40: block start offsets (10 ns ticks) p0 %llu p25 %llu p50 %llu p75 %llu p90 %llu p100 %llu | end offsets p0 %llu p50 %llu p100 %llu\n", G.nmat,
             st0[0], st0[nb / 4], st0[nb / 2], st0[3 * nb / 4], st0[9 * nb / 10], st0[nb - 1], en0[0], en0[nb / 2], en0[nb - 1]);
    }
    if (r4m) {
      static unsigned long long hb[8192 * 8];
      static double bs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (nlaunch == 0) for (double& v : bs) v = 0;
      (void)hipMemcpyFromSymbol(hb, HIP_SYMBOL(jac_busy), sizeof(hb));
      for (int i = 0; i < nb; ++i) for (int w = 0; w < 8; ++w) bs[w] += (double)hb[i * 8 + w] / nb;
      if ((nlaunch + 1) % 32 == 0)
        printf("jacobi_ts nmat %d: cycles between barrier release and next arrival, summed over the 32 sets, by wave: pivots %.0f | strips %.0f %.0f %.0f %.0f %.0f %.0f %.0f\n",
               G.nmat, bs[0] / (nlaunch + 1), bs[1] / (nlaunch + 1), bs[2] / (nlaunch + 1), bs[3] / (nlaunch + 1), bs[4] / (nlaunch + 1), bs[5] / (nlaunch + 1), bs[6] / (nlaunch + 1), bs[7] / (nlaunch + 1));
    }
    if (++nlaunch % 32 == 0) {
      printf("jacobi_ts nmat %d (%d launches): state %.0f | loads->LDS %.0f | crit %.0f | image %.0f | sets %.0f | stores issued %.0f | drained %.0f | first start -> last end %.0f wall-clock ticks (100 MHz) (others: s_memtime ticks)\n",
             G.nmat, nlaunch, sum[0] / nlaunch, sum[1] / nlaunch, sum[2] / nlaunch, sum[3] / nlaunch, sum[4] / nlaunch, sum[5] / nlaunch, sum[6] / nlaunch, span / nlaunch);
      fflush(stdout);
    }
  }
#endif
  if (has_d) G.par ^= 1;
  if (has_u) G.cur ^= 1;
}

// which (block width, block count) combinations jacobi_vstrip_kernel is instantiated for
template <int M2>
static bool vstrip_supported(int C) {
  const int nblk = C / (M2 / 2);
  return M2 == 64 ? (nblk == 8 || nblk == 16) : (nblk == 2 || nblk == 4 || nblk == 8);
}
template <int M2>
static void vstrip_launch(const JacobiGroup& G, int C, int step_begin, int step_end, hipStream_t s) {
  const int nblk = C / (M2 / 2);
  VStripArgs a;
  a.V = G.V; a.Qlog = G.Qlog16[G.lg]; a.st = G.st; a.C = C; a.nmat = G.nmat; a.step_begin = step_begin; a.step_end = step_end; a.seg = G.segs;
  static const int dbg = tune_int("WCT_JACOBI_DBG", 0);
  a.dbg = dbg & 128;
  // Round 5: blocks are handed to the 8 XCDs round-robin by their linear index.  All the row strips of a matrix stream the SAME
  // rotation log (1 MB per 512-channel matrix and segment); with the strip index running fastest they sat on 8 different XCDs
  // and every one of the 8 L2s fetched every log from HBM.  With the MATRIX index running fastest (and a multiple of 8 matrices)
  // the strips of matrix m all run on XCD m % 8 and share its L2.  WCT_JACOBI_XCD=0 (tuning builds): the old order.
  static const int xcd = tune_int("WCT_JACOBI_XCD", 1);
  a.mat_major = xcd && G.nmat % 8 == 0;
#define VSTRIP_CASE(m2, nb, w) \
  if (M2 == m2 && nblk == nb) {                                                                                      \
    const size_t vs_lds = (vstrip_lds_bytes<m2, w>());                                                               \
    /* (every launch: cheap, idempotent, no per-device flag to race on -- ADVICE r5; a refusal is reported here) */   \
    if (vs_lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi_vstrip_kernel<m2, nb, w>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vs_lds) != hipSuccess) \
      LAUNCH_NOTE("jacobi_vstrip_kernel (hipFuncSetAttribute: dynamic LDS above 64 KiB)");                            \
    hipLaunchKernelGGL((jacobi_vstrip_kernel<m2, nb, w>), a.mat_major ? dim3(G.nmat, C / 16 / w) : dim3(C / 16 / w, G.nmat), dim3(w * 64), vs_lds, s, a); \
    LAUNCH_NOTE("jacobi_vstrip_kernel");                                                                             \
  }
  VSTRIP_CASE(64, 16, 4) VSTRIP_CASE(64, 8, 4) VSTRIP_CASE(32, 8, 4) VSTRIP_CASE(32, 4, 4) VSTRIP_CASE(32, 2, 2)
#undef VSTRIP_CASE
}

// before the first launch of a segment: its log buffer must not be in use by a V pass any more
static int jacobi_segment_begin(JacobiGroup* grp, int ngrp) {
  for (int g = 0; g < ngrp; ++g) {
    JacobiGroup& G = grp[g];
    if (G.vstrip && G.v_busy[G.lg]) { HIP_TRY(hipStreamWaitEvent(G.stream, G.ev_v[G.lg], 0)); G.v_busy[G.lg] = false; }
  }
  return WCT_OK;
}
// after the last launch of a segment (steps [step_begin, step_end)): hand its log to the V pass, switch logs
template <int M2>
static int jacobi_segment_end(JacobiGroup* grp, int ngrp, int C, int step_begin, int step_end) {
  for (int g = 0; g < ngrp; ++g) {
    JacobiGroup& G = grp[g];
    if (G.vstrip) {
      HIP_TRY(hipEventRecord(G.ev_seg, G.stream));
      HIP_TRY(hipStreamWaitEvent(G.vs, G.ev_seg, 0));
      vstrip_launch<M2>(G, C, step_begin, step_end, G.vs);
      HIP_TRY(hipEventRecord(G.ev_v[G.lg], G.vs));
      G.v_busy[G.lg] = true;
      G.lg ^= 1;
    }
    G.segs += 1;
  }
  return WCT_OK;
}

// steps [step_begin, step_end) of one sweep as look-ahead launches: D(begin) | {D(s), U(s-1)} ... | U(end-1); afterwards
// the matrices are complete in P[cur].  Launches of the groups are interleaved.  [lo, hi) restricts the launches that
// are enqueued by this call to those of index lo..hi-1 (index = step of the D part; hi = step_end is the closing U).
template <int M2>
static void jacobi_enqueue_segment(JacobiGroup* grp, int ngrp, int C, int step_begin, int step_end, int lo, int hi) {
  for (int step = lo; step < hi; ++step)
    for (int g = 0; g < ngrp; ++g)
      jacobi_fused_launch<M2>(grp[g], C, step_begin, step < step_end, step, step > step_begin, step - 1, step == step_begin);
}

static float jacobi_row_k() {
  static const float k = tune_float("WCT_JACOBI_ROW_K", JACOBI_ROW_K);
  return k;
}

// residual measurement of the group's matrices (in P[cur]), then the test `ck`
static void jacobi_measure(JacobiGroup& G, int C, const JacobiCheckArgs& ck) {
  const int ntr = (C + JACOBI_RESID_T - 1) / JACOBI_RESID_T;
  const int ntile = ntr * (ntr + 1) / 2;
  const float mixed_w = G.tol_fn > 2e-2f ? 7.1f : 1.f;
  unsigned* rowsum = reinterpret_cast<unsigned*>(G.resid + (size_t)G.nmat * JACOBI_RESID_MAXTILES * JACOBI_RESID_STRIDE);
  hipLaunchKernelGGL(jacobi_resid_kernel, dim3(ntile, G.nmat), dim3(256), 0, G.stream, G.P[G.cur], G.st, G.resid, C, rowsum, mixed_w);
  hipLaunchKernelGGL(jacobi_check_kernel, dim3(G.nmat), dim3(64), 0, G.stream, G.st, G.resid, ntile, mixed_w, ck, G.done_host, rowsum, C);
  LAUNCH_NOTE("jacobi_resid_kernel / jacobi_check_kernel");
}

template <int M2>
static int jacobi_run_groups_fused(JacobiGroup* grp, int ngrp, int C) {
  constexpr int B = M2 / 2;
  const int nblk = C / B;
  const int half = -1 + nblk / 2;                 // steps [-1, half) | [half, nblk - 1)
  const int max_sweeps = jacobi_max_sweeps();
  static const float conv_tol = tune_float("WCT_JACOBI_CONV_TOL", JACOBI_CONV_TOL);
  static const int mid_env = tune_int("WCT_JACOBI_MID", 3);
  // V in registers, per segment (jacobi_vstrip_kernel): 0 never, 1 from WCT_JACOBI_VSTRIP_MIN matrices per group on, 2 always
  static const int vs_env = tune_int("WCT_JACOBI_VSTRIP", 1);
  static const int vs_min = tune_int("WCT_JACOBI_VSTRIP_MIN", 24);
  const int mid_from = nblk >= 8 ? mid_env : -1;
  JacobiHost* host = jacobi_host();
  for (int g = 0; g < ngrp; ++g) {
    JacobiGroup& G = grp[g];
    G.cur = 0; G.par = 0; G.lg = 0; G.segs = 0; G.v_busy[0] = G.v_busy[1] = false;
    G.vstrip = host && g < 4 && vstrip_supported<M2>(C) && (vs_env == 2 || (vs_env == 1 && G.nmat >= vs_min && C >= 256));
    if (G.vstrip) { G.vs = host->vs[g]; G.ev_seg = host->ev_seg[g]; G.ev_v[0] = host->ev_v[g][0]; G.ev_v[1] = host->ev_v[g][1]; }
    G.done_host = host && g < 4 ? host->flags_dev + g * 64 : nullptr;
    hipLaunchKernelGGL(jacobi_init_kernel, dim3(64, G.nmat), dim3(256), 0, G.stream, G.A, G.V, G.st, C, G.mat0, G.shared_style,
                       reinterpret_cast<unsigned*>(G.resid + (size_t)G.nmat * JACOBI_RESID_MAXTILES * JACOBI_RESID_STRIDE));
    HIP_TRY(hipGetLastError());
  }
  bool pending = false;
  int rc;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    bool mid = false;
    if (mid_from >= 0 && sweep >= mid_from)
      for (int g = 0; g < ngrp; ++g) mid = mid || grp[g].tol_fn > 0.f;
    // segments of this sweep: [-1, half) + [half, nblk - 1) around a residual test, or the whole sweep in one
    const int end1 = mid ? half : nblk - 1;
    if ((rc = jacobi_segment_begin(grp, ngrp))) return rc;
    // the flags of the previous sweep are looked at once half a sweep of launches is enqueued (the GPU never idles)
    jacobi_enqueue_segment<M2>(grp, ngrp, C, -1, end1, -1, half);
    if (pending) {
      bool all = true;
      for (int g = 0; g < ngrp; ++g) {
        HIP_TRY(hipEventSynchronize(host->ev[g]));
        for (int m = 0; m < grp[g].nmat; ++m) all = all && host->flags[g * 64 + m] != 0;
      }
      pending = false;
      if (all) break;          // every matrix was done before this sweep began: its launches were no-ops
    }
    jacobi_enqueue_segment<M2>(grp, ngrp, C, -1, end1, half, end1 + 1);
    if ((rc = jacobi_segment_end<M2>(grp, ngrp, C, -1, end1))) return rc;
    if (mid) {
      for (int g = 0; g < ngrp; ++g)
        if (grp[g].tol_fn > 0.f) {
          jacobi_measure(grp[g], C, JacobiCheckArgs{1, conv_tol, grp[g].tol_fn, grp[g].cur, grp[g].segs, 1 << 30, jacobi_row_k()});
        }
      if ((rc = jacobi_segment_begin(grp, ngrp))) return rc;
      jacobi_enqueue_segment<M2>(grp, ngrp, C, half, nblk - 1, half, nblk);
      if ((rc = jacobi_segment_end<M2>(grp, ngrp, C, half, nblk - 1))) return rc;
    }
    for (int g = 0; g < ngrp; ++g) {
      jacobi_measure(grp[g], C, JacobiCheckArgs{0, conv_tol, grp[g].tol_fn, grp[g].cur, grp[g].segs, max_sweeps - 3, jacobi_row_k()});
    }
    if (host && sweep >= 2 && sweep + 1 < max_sweeps) {
      for (int g = 0; g < ngrp; ++g) {
        HIP_TRY(hipEventRecord(host->ev[g], grp[g].stream));      // (behind the sweep's jacobi_check_kernel, which stored the flags)
      }
      pending = true;
    }
  }
  for (int g = 0; g < ngrp; ++g) {
    JacobiGroup& G = grp[g];
    for (int l = 0; l < 2; ++l)            // the V passes still in flight belong to this solve
      if (G.vstrip && G.v_busy[l]) { HIP_TRY(hipStreamWaitEvent(G.stream, G.ev_v[l], 0)); G.v_busy[l] = false; }
    hipLaunchKernelGGL(jacobi_gather_kernel, dim3(32, G.nmat), dim3(256), 0, G.stream, G.A, G.P[1], G.st, C, G.cur);
    HIP_TRY(hipGetLastError());
    if (G.sweeps_out || G.fail || G.stats)
      hipLaunchKernelGGL(jacobi_finalize_kernel, dim3(1), dim3(64), 0, G.stream, G.st, G.sweeps_out, G.nmat, conv_tol, G.tol_fn, G.fail, tune_set("WCT_JACOBI_DEBUG"), G.stats);
      HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipGetLastError());
  return launch_rc_take();           // a launch one of the train's void helpers saw refused (named there)
}

static int jacobi_make_group(JacobiGroup* G, float* A, float* V, int C, int nmat, void* workspace, size_t workspace_bytes,
                             int* sweeps_out, int* fail, hipStream_t s, int* stats = nullptr) {
  ARG_CHECK(C % 32 == 0 && C >= 32 && C <= 1024 && nmat >= 1 && nmat <= 64);
  ARG_CHECK(workspace_bytes >= jacobi_workspace_bytes(C, nmat));
  const size_t cc = (size_t)C * C;
  const size_t qbytes = (size_t)nmat * (9 * cc + (size_t)128 * C) * sizeof(float);
  G->A = A; G->V = V; G->nmat = nmat; G->Qbuf = reinterpret_cast<float*>(workspace);
  G->Qlog[0] = G->Qbuf; G->Qlog[1] = G->Qbuf + 2 * cc * nmat;
  G->Sb[0] = G->Qbuf + 4 * cc * nmat; G->Sb[1] = G->Sb[0] + (size_t)64 * C * nmat;
  G->P[0] = A; G->P[1] = G->Sb[1] + (size_t)64 * C * nmat; G->cur = 0; G->par = 0; G->lg = 0; G->segs = 0;
  G->Qlog16[0] = reinterpret_cast<half_t*>(G->P[1] + cc * nmat); G->Qlog16[1] = G->Qlog16[0] + 4 * cc * nmat;
  G->vstrip = 0; G->vs = nullptr; G->v_busy[0] = G->v_busy[1] = false;
  G->st = reinterpret_cast<JacobiState*>(reinterpret_cast<char*>(workspace) + ((qbytes + 255) / 256) * 256);
  G->resid = reinterpret_cast<float*>(reinterpret_cast<char*>(G->st) + (((size_t)nmat * sizeof(JacobiState) + 255) / 256) * 256);
  G->tol_fn = 0.f;
  G->stream = s; G->sweeps_out = sweeps_out; G->fail = fail; G->stats = stats;
  G->mat0 = 0; G->shared_style = 0;
  return WCT_OK;
}

static int jacobi_dispatch(JacobiGroup* grp, int ngrp, int C) {
  // block pairs of 64 indices (32-column blocks) by default; WCT_JACOBI_M2=32 selects 16-column blocks
  static const int force32 = tune_int("WCT_JACOBI_M2", 0) == 32;
  static const int force64 = tune_int("WCT_JACOBI_M2", 0) == 64;
  // measured (16 matrices): 64-wide pairs win from C = 256 up (half the tile traffic), 32-wide below
  // (the round-2 two-launch steps -- separate pair-problem and tile-update kernels -- were deleted in round 4: the
  //  look-ahead launches have been the only path since round 3)
  const bool m64 = !force32 && C % 64 == 0 && (C >= 256 || force64);
  return m64 ? jacobi_run_groups_fused<64>(grp, ngrp, C) : jacobi_run_groups_fused<32>(grp, ngrp, C);
}

// statistics slot of stream group g for matrices of order C inside the caller's status words: after the 8 failure
// words come [4 groups][6 size classes][3] counters (size class = log2(C / 32), capped)
static int* jacobi_stats_slot(int* eig_fail, int g, int C) {
  if (!eig_fail) return nullptr;
  int cls = 0;
  for (int c = C / 32; c > 1 && cls < 5; c >>= 1) ++cls;
  return eig_fail + 8 + (g * 6 + cls) * 3;
}

int launch_jacobi_eigh(float* A, float* V, int C, int nmat, void* workspace, size_t workspace_bytes,
                       int* sweeps_done_dev, int* eig_fail, hipStream_t s) {
  JacobiGroup G;
  int rc = jacobi_make_group(&G, A, V, C, nmat, workspace, workspace_bytes, sweeps_done_dev, eig_fail, s, jacobi_stats_slot(eig_fail, 0, C));
  if (rc) return rc;
  return jacobi_dispatch(&G, 1, C);
}

// ---------------------------------------------------------------------------
// K6: spectral functions with the reference cut-off
// ---------------------------------------------------------------------------
// First-order completion of the matrix function (Daleckii-Krein): the sweeps stop with A = D + E, E a small
// off-diagonal residual (|e_pq| up to ~tol sqrt(d_p d_q)), and V^T A0 V = D + E holds to round-off, so
//   f(A0) = V f(D + E) V^T = V (f(D) + F o E) V^T + O(|E|^2),  F_pq = (f(d_p) - f(d_q)) / (d_p - d_q).
// Using G = f(D) + F o E instead of f(D) squares the error the residual leaves in the transform (measured on the
// level features of a 512x512 frame: 2.7e-3 -> see DESIGN) and lets the sweeps stop a whole sweep earlier.
// f is the reference's spectral function INCLUDING its cut-off (ops.py:68-77 / 112-127): f(l) = 0 for l <= 1e-5,
// else l^-1/2 | l^1/2 (wct_tf), (l + eps)^-1/2 | (l + eps)^1/2 (wct_np).  The divided differences of l^+-1/2 have
// closed forms without cancellation: -1 / (sa sb (sa + sb)) and 1 / (sa + sb) with sa = sqrt(a), sb = sqrt(b);
// across the cut-off (one eigenvalue kept, one dropped) F_pq = f(d_kept) / (d_kept - d_dropped).
// `kind`: 0 whitening gain l^-1/2, 1 colouring gain l^1/2.   G [nmat][C][C], matrix m of A / G at stride `stride`.
// Round 6: the expansion is one in the squared cosines e^2 / (d_p d_q), and it is only used where it converges.  The sweeps may
// stop with pairs that are NOT resolved: the rounding-noise diagonals of a rank-deficient covariance (N < C pixels) never settle
// against each other (the lenient rule of jacobi_check_end lets such a matrix go), and when the feature scale puts that noise
// above the 1e-5 cut-off -- 1e-7 ||cov|| ~ 1 at features of ~1e3 -- those diagonals are KEPT, with squared cosines of order one
// and more among them.  The second divided differences then grow like f(a) e^2 / (2 d_a d_b) per term: measured on
// C = 256, 84 / 78 pixels, features ~1e3 (tools/probe/r06_fuzz_case.py, profiles/r06_noise_block.txt) the transform was off by
// 0.14 .. 11.5 (six seeds of six).  A pair with e^2 > d_p d_q / 4 takes no part in either order of the completion: f of its two
// diagonals stands alone (0.02 .. 0.15 on the same six; the rest is spectral_cut's, below).  On a matrix that met the stop test
// no pair is near that bound (the test bounds the mean over the rows of a row's SUM of squared cosines by 1.6e-3), so every
// converged result keeps its bits.
__device__ __forceinline__ bool pair_resolved(float dp, float dq, float e) { return e * e <= 0.25f * dp * dq; }

// The cut-off of ONE matrix (round 6).  The reference drops eigenvalues <= 1e-5 (ops.py:68-69 / 112,125).  A covariance is positive
// semi-definite in exact arithmetic, so a NEGATIVE diagonal of the rotated matrix is rounding noise and nothing else, and the most
// negative one, -r, measures the noise of this evaluation on this matrix (N < C pixels: C - N + 1 directions of exact zeros come
// out as a cluster symmetric around 0, radius r ~ 1.5e-7 ||cov||).  A positive diagonal no larger than 2 r is the same noise: it is
// dropped like the exact zero it stands for -- cut = max(1e-5, 2 r).  Below a feature scale of ~10 (r < 5e-6) this is the
// reference's cut-off unchanged; above it the noise directions, which the absolute 1e-5 would keep with gains (d + eps)^-1/2
// that dwarf the signal's, no longer enter.  Why it is needed (profiles/r06_noise_block.txt: C = 256, 84 / 78 pixels, features
// ~1e3, six seeds; error of the transform against the float64 oracle, the reference's own float32 evaluation 1.3e-4 .. 7e-4):
// diagonal gains only 3.5e-3 .. 1.9e-2, with the first-order completion 5e-2 .. 1.3, with the second-order one 0.35 .. 16.
// The kept count stays inside the band the parity tests accept as the reference's (tests/test_gpu_fuzz.py: eigenvalues within
// 2e-6 + 1e-6 ||cov|| of 1e-5 may fall on either side in float32).  Block-wide (256 threads), `red`: 4 floats of LDS.
constexpr float SPECTRAL_NOISE_CUT = 2.f;
__device__ __forceinline__ float spectral_cut(const float* Am, int C, int tid, float* red) {
  float mn = 0.f;
  for (int i = tid; i < C; i += 256) mn = fminf(mn, Am[(size_t)i * C + i]);
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = mn;
  __syncthreads();
  mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
  return fmaxf(1e-5f, -SPECTRAL_NOISE_CUT * mn);
}

__device__ __forceinline__ float spectral_entry(float dp, float dq, float e, bool diag, int kind, float shift, float cut) {
  const bool kp = dp > cut, kq = dq > cut;
  if (diag) return kp ? (kind == 0 ? 1.f / sqrtf(dp + shift) : sqrtf(dp + shift)) : 0.f;
  if (!kp && !kq) return 0.f;
  if (kp && kq) {
    if (!pair_resolved(dp, dq, e)) return 0.f;
    const float sa = sqrtf(dp + shift), sb = sqrtf(dq + shift);
    return kind == 0 ? -e / (sa * sb * (sa + sb)) : e / (sa + sb);
  }
  const float dk = kp ? dp : dq, dd = kp ? dq : dp;
  const float fk = kind == 0 ? 1.f / sqrtf(dk + shift) : sqrtf(dk + shift);
  // across the cut-off f jumps, and the first-order term e f_k / (d_k - d_d) is only the expansion of
  // (f_k - 0) sin(theta) cos(theta) for a small rotation angle theta ~ e / gap: it can never exceed f_k / 2.  A residual
  // that is not small against the gap (two eigenvalues within a per cent of 1e-5) is clamped to that bound
  return e * fk / fmaxf(dk - dd, 2.f * fabsf(e));
}

// The three elementwise passes over a matrix in 64 x 64 TILES (round 4).  Written element by element (round 3) every
// output read two diagonal entries (a gather with stride C + 1: one cache line per thread) and the mirrored entry
// (a column walk), ~1.4 ms per 32-pair step for three kernels that move 100 MB each.  A block now parks the two
// diagonal segments of its tile and the MIRROR tile in LDS (rows loaded coalesced, read transposed), so every global
// access is a coalesced 16-byte row piece.  Per element the arithmetic is what it was, in the same order.
struct SpectralTile {
  float dp[64], dq[64];          // diagonal entries of the tile's rows / columns
  float mt[64][65];              // mirror tile: mt[c][r] = A[q0 + c][p0 + r]
};
__device__ __forceinline__ void spectral_tile_load(SpectralTile& t, const float* Am, int C, int p0, int q0, int tid) {
  if (tid < 64) t.dp[tid] = p0 + tid < C ? Am[(size_t)(p0 + tid) * C + p0 + tid] : 0.f;
  else if (tid < 128) t.dq[tid - 64] = q0 + tid - 64 < C ? Am[(size_t)(q0 + tid - 64) * C + q0 + tid - 64] : 0.f;
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = q0 + ty * 4 + i, c = p0 + tx * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < C && c < C) v = *reinterpret_cast<const f32x4*>(Am + (size_t)r * C + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) t.mt[ty * 4 + i][tx * 4 + j] = v[j];
  }
}

// grid (C / 64, C / 64, nbatch): tile rows p0 = 64 blockIdx.y, columns q0 = 64 blockIdx.x
__global__ __launch_bounds__(256) void spectral_matrix_kernel(const float* A, float* G, int C, size_t stride, int kind, float shift, int correct) {
  __shared__ SpectralTile t;
  const int m = blockIdx.z, p0 = blockIdx.y * 64, q0 = blockIdx.x * 64, tid = threadIdx.x;
  __shared__ float cutred[4];
  const float* Am = A + m * stride;
  float* Gm = G + m * stride;
  const float cut = spectral_cut(Am, C, tid, cutred);
  spectral_tile_load(t, Am, C, p0, q0, tid);
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = ty * 4 + i, p = p0 + pl, q = q0 + tx * 4;
    if (p >= C || q >= C) continue;
    const f32x4 a = *reinterpret_cast<const f32x4*>(Am + (size_t)p * C + q);
    f32x4 g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // the two triangles agree to round-off; their mean keeps G exactly symmetric
      const float e = correct ? 0.5f * (a[j] + t.mt[tx * 4 + j][pl]) : 0.f;
      g[j] = spectral_entry(t.dp[pl], t.dq[tx * 4 + j], e, p == q + j, kind, shift, cut);
    }
    *reinterpret_cast<f32x4*>(Gm + (size_t)p * C + q) = g;
  }
}

// Second-order term of the same expansion (round 3):  (L2)_pq = sum_k f[d_p, d_k, d_q] E_pk E_kq  with the second divided
// differences of the spectral functions, which have closed forms without cancellation (sa = sqrt(a + shift) ...):
//   l^+1/2:  f[a,b,c] = -1 / ((sa+sb)(sb+sc)(sa+sc))
//   l^-1/2:  f[a,b,c] = (sa+sb+sc) / (sa sb sc (sa+sb)(sb+sc)(sa+sc))
// and both SEPARATE into matrix products of elementwise-scaled copies of E:  with N_pk = E_pk / (s_p + s_k),
// R_pk = E_pk / s_k, P_pk = N_pk / s_k,
//   l^+1/2:  L2 = -(N N) o 1/(s_p + s_q)
//   l^-1/2:  L2 = ( sym(R N) + (s_p + s_q)/2 (P N) ) o 1/(s_p s_q (s_p + s_q))      (P N = N diag(1/s) N is symmetric)
// (checked against an exact eigendecomposition in NumPy: the error of f(D + E) drops from ~r^2 to ~r^3).  Only the block
// of kept eigenvalues takes part: a dropped direction's couplings are bounded by sqrt(1e-5 d_p) tol and enter squared.
// It lets the sweeps stop one sweep earlier for the same transform error (profiles/r03_eig_calibration.txt).
__global__ __launch_bounds__(256) void spectral_prep2_kernel(const float* A, float* N, float* R, float* Pm, int C, size_t stride, int kind, float shift) {
  __shared__ SpectralTile t;
  const int m = blockIdx.z, p0 = blockIdx.y * 64, q0 = blockIdx.x * 64, tid = threadIdx.x;
  const size_t cc = (size_t)C * C;
  __shared__ float cutred[4];
  const float* Am = A + m * stride;
  const float cut = spectral_cut(Am, C, tid, cutred);
  spectral_tile_load(t, Am, C, p0, q0, tid);
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = ty * 4 + i, p = p0 + pl, k0 = q0 + tx * 4;
    if (p >= C || k0 >= C) continue;
    const f32x4 a = *reinterpret_cast<const f32x4*>(Am + (size_t)p * C + k0);
    const float dp = t.dp[pl];
    f32x4 n4 = {0.f, 0.f, 0.f, 0.f}, r4 = n4, p4 = n4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dk = t.dq[tx * 4 + j];
      const float e = 0.5f * (a[j] + t.mt[tx * 4 + j][pl]);
      if (p != k0 + j && dp > cut && dk > cut && pair_resolved(dp, dk, e)) {
        const float sp = sqrtf(dp + shift), sk = sqrtf(dk + shift);
        n4[j] = e / (sp + sk);
        r4[j] = e / sk;
        p4[j] = n4[j] / sk;
      }
    }
    const size_t o = m * cc + (size_t)p * C + k0;
    *reinterpret_cast<f32x4*>(N + o) = n4;
    if (kind == 0) { *reinterpret_cast<f32x4*>(R + o) = r4; *reinterpret_cast<f32x4*>(Pm + o) = p4; }
  }
}

__global__ __launch_bounds__(256) void spectral_add2_kernel(const float* A, float* G, const float* X1, const float* X2, int C, size_t stride, int kind, float shift) {
  __shared__ float dps[64], dqs[64];
  __shared__ float x1t[64][65], x2t[64][65];       // mirror tiles of X1 (kind 0 only), X2
  const int m = blockIdx.z, p0 = blockIdx.y * 64, q0 = blockIdx.x * 64, tid = threadIdx.x;
  const size_t cc = (size_t)C * C;
  const float* Am = A + m * stride;
  float* Gm = G + m * stride;
  // a matrix with an eigenvalue within half a decade of the cut-off keeps the first-order completion and the old stop
  // threshold (jacobi_resid_kernel weighs its residual accordingly): around the cut-off sit the noise directions of
  // rank-deficient covariances, whose mutual couplings never become small, and a second-order sum over them is noise
  int near = 0;
  for (int i = tid; i < C; i += 256) {
    const float d = fabsf(Am[(size_t)i * C + i]);
    near |= (d > 3.3e-6f) & (d < 3e-5f);
  }
  if (__syncthreads_or(near)) return;
  __shared__ float cutred[4];
  const float cut = spectral_cut(Am, C, tid, cutred);
  if (tid < 64) dps[tid] = p0 + tid < C ? Am[(size_t)(p0 + tid) * C + p0 + tid] : 0.f;
  else if (tid < 128) dqs[tid - 64] = q0 + tid - 64 < C ? Am[(size_t)(q0 + tid - 64) * C + q0 + tid - 64] : 0.f;
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = q0 + ty * 4 + i, c = p0 + tx * 4;
    f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1;
    if (r < C && c < C) {
      v2 = *reinterpret_cast<const f32x4*>(X2 + m * cc + (size_t)r * C + c);
      if (kind == 0) v1 = *reinterpret_cast<const f32x4*>(X1 + m * cc + (size_t)r * C + c);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { x2t[ty * 4 + i][tx * 4 + j] = v2[j]; x1t[ty * 4 + i][tx * 4 + j] = v1[j]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = ty * 4 + i, p = p0 + pl, q = q0 + tx * 4;
    if (p >= C || q >= C) continue;
    const size_t o = (size_t)p * C + q;
    const f32x4 x2 = *reinterpret_cast<const f32x4*>(X2 + m * cc + o);
    f32x4 x1 = {0.f, 0.f, 0.f, 0.f};
    if (kind == 0) x1 = *reinterpret_cast<const f32x4*>(X1 + m * cc + o);
    f32x4 g = *reinterpret_cast<const f32x4*>(Gm + o);
    const float dp = dps[pl];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dq = dqs[tx * 4 + j];
      if (!(dp > cut && dq > cut)) continue;
      const float sp = sqrtf(dp + shift), sq = sqrtf(dq + shift);
      float l2;
      if (kind == 1) l2 = -0.5f * (x2[j] + x2t[tx * 4 + j][pl]) / (sp + sq);
      else l2 = (0.5f * (x1[j] + x1t[tx * 4 + j][pl]) + 0.25f * (sp + sq) * (x2[j] + x2t[tx * 4 + j][pl])) / (sp * sq * (sp + sq));
      g[j] += l2;
    }
    *reinterpret_cast<f32x4*>(Gm + o) = g;
  }
}

// ---- round 5: the transform tail in MERGED launches.  Until round 4 every level ran the chain spectral_matrix -> prep2 ->
// products -> add2 -> V G -> (V G) V^T twice, content side then style side (13 launches), then T = Tcs Tw, a memset, the blend
// and the apply: 17 launches per level, most of them a few microseconds of work on half the matrices.  Here one launch of each
// kind covers all 2P matrices of the level (matrix m = 2 pair + side; kind = m & 1: 0 whitening, 1 colouring), the first-order
// matrix and the second-order operands come out of ONE pass over the tile, the blend rides on the epilogue of T = Tcs Tw
// (gemm_f32_kernel), and the pass that opens the chain also clears mabs and writes the bias vector: 8 launches per level.
// Element by element the arithmetic is the one of the kernels above, in the same order: the outputs are the same bits.
struct SpecAllArgs {
  const float* A;            // [2P][C][C] rotated covariances (diagonal = eigenvalues, off-diagonal = residual)
  float* G;                  // [2P][C][C]
  float *N, *R, *Pm;         // N [2P][C][C]; R, Pm [P][C][C] (whitening side only)
  const float *X1, *X2;      // add2: X1 [P][C][C], X2 [2P][C][C]
  int C; float shift; int correct, second, shared_style;
  unsigned* mabs; const float* mean; float* bias; float alpha; int mode;     // per-pair housekeeping of the opening pass
};

__global__ __launch_bounds__(256) void spectral_open_all_kernel(SpecAllArgs a) {
  __shared__ SpectralTile t;
  const int m = blockIdx.z, p0 = blockIdx.y * 64, q0 = blockIdx.x * 64, tid = threadIdx.x;
  const int C = a.C, kind = m & 1, pair = m >> 1;
  const size_t cc = (size_t)C * C;
  if (kind == 0 && blockIdx.x == 0 && blockIdx.y == 0) {
    // what blend_matrix_kernel did besides the blend: bias = alpha ms (+ (1 - alpha) mc in tf mode); and mabs starts at 0
    if (tid == 0) a.mabs[pair] = 0u;
    const float* mp = a.mean + (size_t)pair * 2 * C;
    const float* ms = a.mean + (size_t)(a.shared_style ? 0 : pair) * 2 * C + C;
    for (int i = tid; i < C; i += 256) {
      float b = a.alpha * ms[i];
      if (a.mode == WCT_MODE_TF) b += (1.f - a.alpha) * mp[i];
      a.bias[pair * C + i] = b;
    }
  }
  if (skip_style_mat(m, a.shared_style)) return;
  __shared__ float cutred[4];
  const float* Am = a.A + m * cc;
  const float cut = spectral_cut(Am, C, tid, cutred);
  spectral_tile_load(t, Am, C, p0, q0, tid);
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = ty * 4 + i, p = p0 + pl, q = q0 + tx * 4;
    if (p >= C || q >= C) continue;
    const f32x4 av = *reinterpret_cast<const f32x4*>(Am + (size_t)p * C + q);
    const float dp = t.dp[pl];
    f32x4 g, n4 = {0.f, 0.f, 0.f, 0.f}, r4 = n4, p4 = n4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dk = t.dq[tx * 4 + j];
      const float em = 0.5f * (av[j] + t.mt[tx * 4 + j][pl]);      // the two triangles agree to round-off; their mean keeps G symmetric
      g[j] = spectral_entry(dp, dk, a.correct ? em : 0.f, p == q + j, kind, a.shift, cut);
      if (a.second && p != q + j && dp > cut && dk > cut && pair_resolved(dp, dk, em)) {
        const float sp = sqrtf(dp + a.shift), sk = sqrtf(dk + a.shift);
        n4[j] = em / (sp + sk);
        r4[j] = em / sk;
        p4[j] = n4[j] / sk;
      }
    }
    const size_t o = (size_t)p * C + q;
    *reinterpret_cast<f32x4*>(a.G + m * cc + o) = g;
    if (a.second) {
      *reinterpret_cast<f32x4*>(a.N + m * cc + o) = n4;
      if (kind == 0) { *reinterpret_cast<f32x4*>(a.R + pair * cc + o) = r4; *reinterpret_cast<f32x4*>(a.Pm + pair * cc + o) = p4; }
    }
  }
}

__global__ __launch_bounds__(256) void spectral_add2_all_kernel(SpecAllArgs a) {
  __shared__ float dps[64], dqs[64];
  __shared__ float x1t[64][65], x2t[64][65];       // mirror tiles of X1 (kind 0 only), X2
  const int m = blockIdx.z, p0 = blockIdx.y * 64, q0 = blockIdx.x * 64, tid = threadIdx.x;
  if (skip_style_mat(m, a.shared_style)) return;
  const int C = a.C, kind = m & 1;
  const size_t cc = (size_t)C * C;
  const float* Am = a.A + m * cc;
  float* Gm = a.G + m * cc;
  const float* X2 = a.X2 + m * cc;
  const float* X1 = a.X1 + (size_t)(m >> 1) * cc;
  int near = 0;                                     // (see spectral_add2_kernel)
  for (int i = tid; i < C; i += 256) {
    const float d = fabsf(Am[(size_t)i * C + i]);
    near |= (d > 3.3e-6f) & (d < 3e-5f);
  }
  if (__syncthreads_or(near)) return;
  __shared__ float cutred[4];
  const float cut = spectral_cut(Am, C, tid, cutred);
  if (tid < 64) dps[tid] = p0 + tid < C ? Am[(size_t)(p0 + tid) * C + p0 + tid] : 0.f;
  else if (tid < 128) dqs[tid - 64] = q0 + tid - 64 < C ? Am[(size_t)(q0 + tid - 64) * C + q0 + tid - 64] : 0.f;
  const int ty = tid >> 4, tx = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = q0 + ty * 4 + i, c = p0 + tx * 4;
    f32x4 v1 = {0.f, 0.f, 0.f, 0.f}, v2 = v1;
    if (r < C && c < C) {
      v2 = *reinterpret_cast<const f32x4*>(X2 + (size_t)r * C + c);
      if (kind == 0) v1 = *reinterpret_cast<const f32x4*>(X1 + (size_t)r * C + c);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { x2t[ty * 4 + i][tx * 4 + j] = v2[j]; x1t[ty * 4 + i][tx * 4 + j] = v1[j]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pl = ty * 4 + i, p = p0 + pl, q = q0 + tx * 4;
    if (p >= C || q >= C) continue;
    const size_t o = (size_t)p * C + q;
    const f32x4 x2 = *reinterpret_cast<const f32x4*>(X2 + o);
    f32x4 x1 = {0.f, 0.f, 0.f, 0.f};
    if (kind == 0) x1 = *reinterpret_cast<const f32x4*>(X1 + o);
    f32x4 g = *reinterpret_cast<const f32x4*>(Gm + o);
    const float dp = dps[pl];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float dq = dqs[tx * 4 + j];
      if (!(dp > cut && dq > cut)) continue;
      const float sp = sqrtf(dp + a.shift), sq = sqrtf(dq + a.shift);
      float l2;
      if (kind == 1) l2 = -0.5f * (x2[j] + x2t[tx * 4 + j][pl]) / (sp + sq);
      else l2 = (0.5f * (x1[j] + x1t[tx * 4 + j][pl]) + 0.25f * (sp + sq) * (x2[j] + x2t[tx * 4 + j][pl])) / (sp * sq * (sp + sq));
      g[j] += l2;
    }
    *reinterpret_cast<f32x4*>(Gm + o) = g;
  }
}

// 0: spectral functions of the diagonal only, 1: + first-order completion, 2 (default): + second-order completion
static int eig_correct_enabled() {
  static const int on = tune_int("WCT_EIG_CORRECT", 2);
  return on;
}
// residual at which the WCT path stops sweeping.  Calibrated on the level features of a 512x512 frame
// (profiles/r02_eig_calibration.txt): with r2 the strict measure at the stop, the transform error is ~0.55 sqrt(r2)
// without the first-order completion and ~0.8 r2 (+ ~3e-5 from the other stages) with it, so 1.5e-2 bounds the
// completed transform's error by ~1.8e-4 -- five times inside the 1e-3 budget.  0 without the completion.
constexpr float JACOBI_TOL_FN = 1.5e-2f;
constexpr float JACOBI_TOL_FN2 = 4e-2f;            // with the second-order completion (profiles/r03_eig_calibration.txt: the transform error
                                                   // at 4e-2 with it, 3.1e-5 .. 1.3e-4, is what 1.5e-2 gave without it, one sweep later)
static float jacobi_tol_fn() {
  static const float t = tune_float("WCT_JACOBI_TOL_FN", eig_correct_enabled() >= 2 ? JACOBI_TOL_FN2 : (eig_correct_enabled() ? JACOBI_TOL_FN : 0.f));
  return t;          // (tuning builds: an explicit WCT_JACOBI_TOL_FN also applies without the completion -- calibration runs)
}

// out[b] = V[b] G[b] V[b]^T for nbatch matrices (strides in elements); X: scratch of the same shape as G
// scratch2: 5 * nbatch * C * C floats for the second-order completion (or null: first order at most)
static int launch_spectral_function(const float* A, const float* V, float* G, float* X, float* out, int C, int nbatch,
                                    size_t stride, size_t out_stride, int kind, float shift, hipStream_t s, float* scratch2 = nullptr) {
  const size_t cc = (size_t)C * C;
  const dim3 tiles(cdiv(C, 64), cdiv(C, 64), nbatch);
  hipLaunchKernelGGL(spectral_matrix_kernel, tiles, dim3(256), 0, s, A, G, C, stride, kind, shift, eig_correct_enabled());
  HIP_TRY(hipGetLastError());
  if (scratch2 && eig_correct_enabled() >= 2) {
    float* N = scratch2; float* R = N + nbatch * cc; float* Pm = R + nbatch * cc; float* X1 = Pm + nbatch * cc; float* X2 = X1 + nbatch * cc;
    hipLaunchKernelGGL(spectral_prep2_kernel, tiles, dim3(256), 0, s, A, N, R, Pm, C, stride, kind, shift);
    HIP_TRY(hipGetLastError());
    GemmArgs a = {};   // X2 = (kind 1: N, kind 0: P) N
    a.A = kind == 1 ? N : Pm; a.lda = C; a.a_kmajor = 0; a.B = N; a.ldb = C; a.b_kmajor = 1; a.sA = a.sB = cc;
    a.M = C; a.N = C; a.K = C; a.ksplit = C; a.out32 = X2; a.ldo = C; a.s_out = cc;
    int rc2 = launch_gemm(a, 1, nbatch, s);
    if (rc2) return rc2;
    if (kind == 0) {
      a.A = R; a.out32 = X1;   // X1 = R N
      if ((rc2 = launch_gemm(a, 1, nbatch, s))) return rc2;
    }
    hipLaunchKernelGGL(spectral_add2_kernel, tiles, dim3(256), 0, s, A, G, X1, X2, C, stride, kind, shift);
    HIP_TRY(hipGetLastError());
  }
  GemmArgs g = {};   // X = V G
  g.A = V; g.lda = C; g.a_kmajor = 0; g.B = G; g.ldb = C; g.b_kmajor = 1; g.sA = g.sB = stride;
  g.M = C; g.N = C; g.K = C; g.ksplit = C; g.out32 = X; g.ldo = C; g.s_out = stride;
  int rc = launch_gemm(g, 1, nbatch, s);
  if (rc) return rc;
  GemmArgs h = {};   // out = X V^T
  h.A = X; h.lda = C; h.a_kmajor = 0; h.B = V; h.ldb = C; h.b_kmajor = 0; h.sA = h.sB = stride;
  h.M = C; h.N = C; h.K = C; h.ksplit = C; h.out32 = out; h.ldo = C; h.s_out = out_stride;
  return launch_gemm(h, 1, nbatch, s);
}

// (the blend M = alpha T + (1 - alpha) I, max |M| and the bias vector: gemm_f32_kernel's blend epilogue and
// spectral_open_all_kernel since round 5)


// ---------------------------------------------------------------------------
// K7: apply  out[n][j] = sum_k (x[n][k] - mc[k]) M[j][k] + b[j]   (ops.py:73-83 with the blend folded into M, b)
// Same split-operand scheme as the covariance: (x - mc) s_x and M s_M are split into fp16 hi + lo and
// multiplied by three v_mfma_f32_32x32x16_f16 with fp32 accumulation; s_x, s_M are powers of two.
// D[channel][pixel] (A = M rows, B = pixels) so that a lane ends up with 4 consecutive channels of one
// pixel, as in the conv epilogue.  Block = BC channels x BP pixels, 256 threads = 2x2 waves, K-stage 32.
// ---------------------------------------------------------------------------
struct ApplyArgs {
  const float* x; int N; int C;     // content features [P][N][C]
  const float* mean;                // [2P][C]: content mean of pair p at 2p
  const float* M; const float* bias;  // [P][C][C], [P][C]
  const float* xscale;              // [2P]: content scale of pair p at 2p
  const unsigned* mabs;             // [P] max |M| (float bits)
  half_t* out16; float* out32;      // [P][N][C], either may be null
};

template <int BC, int BP>
__global__ __launch_bounds__(256, 2) void apply_f16x2_kernel(ApplyArgs p) {
  constexpr int TM = BC / 64, TN = BP / 64;
  constexpr int MI = BC * 4 / 256, XI = BP * 4 / 256;    // 16-B (8 k) pieces per thread and stage
  __shared__ __attribute__((aligned(16))) unsigned char lm[2][BC * 64];   // M hi, lo
  __shared__ __attribute__((aligned(16))) unsigned char lx[2][BP * 64];   // x hi, lo
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int pair = blockIdx.z;
  const int n0 = blockIdx.x * BP, c0 = blockIdx.y * BC;
  const int N = p.N, C = p.C;
  const float* x = p.x + (size_t)pair * N * C;
  const float* M = p.M + (size_t)pair * C * C;
  const float* mean = p.mean + (size_t)pair * 2 * C;
  const float sx = p.xscale[2 * pair];
  float sM = 1.f;
  {
    const float m = __uint_as_float(p.mabs[pair]);
    if (m > 0.f) { int e; frexpf(m, &e); sM = ldexpf(1.f, 14 - e); }
  }

  // staging pointers (rows clamped: out-of-range rows are computed on valid data and dropped at the store)
  const float* mp[MI]; const float* xp[XI];
  int moff[MI], xoff[XI], xk[XI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int item = tid + i * 256, row = item >> 2, kq = item & 3;
    mp[i] = M + (size_t)min(c0 + row, C - 1) * C + kq * 8;
    moff[i] = (row * 4 + (kq ^ ((row >> 2) & 3))) * 16;
  }
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int item = tid + i * 256, row = item >> 2, kq = item & 3;
    xp[i] = x + (size_t)min(n0 + row, N - 1) * C + kq * 8;
    xoff[i] = (row * 4 + (kq ^ ((row >> 2) & 3))) * 16;
    xk[i] = kq * 8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 rm[MI][2], rx[XI][2];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      rm[i][0] = *reinterpret_cast<const f32x4*>(mp[i] + k0);
      rm[i][1] = *reinterpret_cast<const f32x4*>(mp[i] + k0 + 4);
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      rx[i][0] = *reinterpret_cast<const f32x4*>(xp[i] + k0) - *reinterpret_cast<const f32x4*>(mean + k0 + xk[i]);
      rx[i][1] = *reinterpret_cast<const f32x4*>(xp[i] + k0 + 4) - *reinterpret_cast<const f32x4*>(mean + k0 + xk[i] + 4);
    }
  };
  auto split_store = [&](const f32x4 (&r)[2], float sc, unsigned char* hi, unsigned char* lo, int off) {
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = r[j >> 2][j & 3] * sc;
      h[j] = (half_t)v;
      l[j] = (half_t)(v - (float)h[j]);
    }
    *reinterpret_cast<half8*>(hi + off) = h;
    *reinterpret_cast<half8*>(lo + off) = l;
  };

  load(0);
  for (int k0 = 0; k0 < C; k0 += 32) {
#pragma unroll
    for (int i = 0; i < MI; ++i) split_store(rm[i], sM, lm[0], lm[1], moff[i]);
#pragma unroll
    for (int i = 0; i < XI; ++i) split_store(rx[i], sx, lx[0], lx[1], xoff[i]);
    __syncthreads();
    if (k0 + 32 < C) load(k0 + 32);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + (lane & 31);
        const int off = (r * 4 + (chunk ^ ((r >> 2) & 3))) * 16;
        ah[i] = *reinterpret_cast<const half8*>(lm[0] + off);
        al[i] = *reinterpret_cast<const half8*>(lm[1] + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + (lane & 31);
        const int off = (r * 4 + (chunk ^ ((r >> 2) & 3))) * 16;
        bh[j] = *reinterpret_cast<const half8*>(lx[0] + off);
        bl[j] = *reinterpret_cast<const half8*>(lx[1] + off);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // epilogue: reg r of a tile = channel (r&3)+8*(r>>2)+4*(lane>>5), pixel lane&31
  const float inv = 1.f / (sx * sM);
  const float* bias = p.bias + (size_t)pair * C;
  const int kgrp = lane >> 5;
  // this lane's bias values, fetched before the first store (a load issued between the stores waits with vmcnt(0),
  // i.e. for every store issued so far: see conv_epilogue_t)
  f32x4 bvs[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int co = c0 + (wm * TM + i) * 32 + 8 * rq + 4 * kgrp;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      bvs[i][rq] = co < C ? *reinterpret_cast<const f32x4*>(bias + co) : z;
    }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    const bool n_ok = n < N;
    const size_t row = ((size_t)pair * N + (n_ok ? n : 0)) * C;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int cb = c0 + (wm * TM + i) * 32;
      unsigned pk[4][2];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int co = cb + 8 * rq + 4 * kgrp;
        const f32x4 bv = bvs[i][rq];
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][rq * 4 + q] * inv + bv[q];
        if (p.out32 && n_ok && co < C) *reinterpret_cast<f32x4*>(p.out32 + row + co) = v;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 lo = {(half_t)v[0], (half_t)v[1]}, hi = {(half_t)v[2], (half_t)v[3]};
        pk[rq][0] = __builtin_bit_cast(unsigned, lo);
        pk[rq][1] = __builtin_bit_cast(unsigned, hi);
      }
      if (p.out16) {
        // pair register quads across the two half-waves (v_permlane32_swap): 16-B stores of 8 channels
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          auto sxw = __builtin_amdgcn_permlane32_swap(pk[2 * m][0], pk[2 * m + 1][0], false, false);
          auto syw = __builtin_amdgcn_permlane32_swap(pk[2 * m][1], pk[2 * m + 1][1], false, false);
          u32x4 o = {sxw[0], syw[0], sxw[1], syw[1]};
          const int co = cb + 16 * m + 8 * kgrp;
          if (n_ok && co < C) *reinterpret_cast<u32x4*>(p.out16 + row + co) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// workspace carving (P independent content/style pairs per call)
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct WctCarve {
  float *mean, *var, *stat_partial, *absmax, *scale, *cov_partial, *A, *A0, *V, *d, *G, *X, *S2, *Tw, *Tcs, *T, *M, *bias;
  unsigned* mabs; int* refresh;
  void* jacobi_ws; size_t jacobi_bytes;
  int nslab, nsplit, ksplit;
  size_t total;
};

static int wct_nslab(int N) { int s = cdiv(N, 64); return s < 1 ? 1 : (s > 256 ? 256 : s); }

static void cov_split(int C, int Nmax, int P, int* nsplit, int* ksplit) {
  const int nt = C >= 128 ? cdiv(C, 128) : 1;
  const int tiles = nt * (nt + 1) / 2 * P;     // tiles on or above the diagonal
  // slices per matrix: x2 matrices per pair.  A single pair gets >= 64 blocks (its covariance is a few tens
  // of microseconds either way); a finer split only multiplies the partial-sum traffic of a large batch
  int want = 32 / tiles;
  if (want < 1) want = 1;
  int ks = cdiv(cdiv(Nmax, want), 32) * 32;    // multiple of the kernel's K-stage
  if (ks < 256) ks = 256;
  *ksplit = ks;
  *nsplit = cdiv(Nmax, ks);
}

static WctCarve carve(void* base, int C, int Nc, int Ns, int P) {
  WctCarve w;
  const int Nmax = Nc > Ns ? Nc : Ns;
  w.nslab = wct_nslab(Nmax);
  cov_split(C, Nmax, 1, &w.nsplit, &w.ksplit);   // independent of P: a pair's result must not depend on its batch
  size_t off = 0;
  char* b = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) { void* p = b ? b + off : nullptr; off += align_up(bytes); return p; };
  const size_t cc = (size_t)C * C * sizeof(float);
  w.mean = (float*)take((size_t)2 * P * C * sizeof(float));
  w.var = (float*)take((size_t)2 * P * C * sizeof(float));
  w.stat_partial = (float*)take((size_t)2 * P * w.nslab * C * sizeof(float));
  w.absmax = (float*)take((size_t)2 * P * w.nslab * sizeof(float));
  w.scale = (float*)take((size_t)2 * P * sizeof(float));
  w.cov_partial = (float*)take((size_t)2 * P * w.nsplit * cc);
  w.A = (float*)take(2 * P * cc);
  w.A0 = (float*)take(2 * P * cc);          // the covariances as computed (the solver rotates w.A in place)
  w.refresh = (int*)take((size_t)2 * P * sizeof(int));
  w.V = (float*)take(2 * P * cc);
  w.d = (float*)take((size_t)2 * P * C * sizeof(float));
  w.G = (float*)take(2 * P * cc);
  w.X = (float*)take(2 * P * cc);
  w.S2 = (float*)take(7 * P * cc);          // second-order operands of a level, all matrices at once: N [2P], X2 [2P], R, Pm, X1 [P]
  w.Tw = (float*)take(2 * P * cc);          // [2P][C][C] interleaved: whitening matrix of pair p at 2p, colouring matrix at 2p + 1
  w.Tcs = w.Tw + (size_t)C * C;             // (style-swap path: one pair, Tw and Tcs side by side)
  w.T = (float*)take(P * cc);
  w.M = (float*)take(P * cc);
  w.bias = (float*)take((size_t)P * C * sizeof(float));
  w.mabs = (unsigned*)take((size_t)P * sizeof(unsigned));
  w.jacobi_bytes = jacobi_workspace_bytes(C, 2 * P) + 4 * (1024 + 64 * sizeof(JacobiState));   // up to 4 groups
  w.jacobi_ws = take(w.jacobi_bytes);
  w.total = off;
  return w;
}

size_t wct_workspace_bytes(int C, int Nc, int Ns, int P) {
  return carve(nullptr, C < 32 ? 32 : C, Nc, Ns, P).total;
}

static int launch_means(const float* content, int Nc, const float* style, int Ns, int C, int P,
                        const WctCarve& w, bool with_var, int shared_style, hipStream_t s, const WctFeatStats* fs = nullptr) {
  StatArgs sa;
  sa.x[0] = content; sa.x[1] = style; sa.n[0] = Nc; sa.n[1] = Ns;
  for (int b = 0; b < 2; ++b) {
    sa.u[b] = fs && fs->umax[b] ? fs->u[b] : nullptr;
    sa.umax[b] = sa.u[b] ? fs->umax[b] : nullptr;
  }
  sa.mean = nullptr; sa.partial = w.stat_partial; sa.absmax = w.absmax; sa.C = C; sa.nslab = w.nslab; sa.shared_style = shared_style;
  hipLaunchKernelGGL(colsum_kernel, dim3(w.nslab, 2 * P), dim3(256), 0, s, sa);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(C, 256), 2 * P), dim3(256), 0, s, w.stat_partial, w.mean, C, w.nslab, (float)Nc, (float)Ns, shared_style,
                     (const float*)w.absmax, w.scale);
  if (with_var) {
    sa.mean = w.mean; sa.absmax = nullptr;
    hipLaunchKernelGGL(colsum_kernel, dim3(w.nslab, 2 * P), dim3(256), 0, s, sa);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(C, 256), 2 * P), dim3(256), 0, s, w.stat_partial, w.var, C, w.nslab, (float)Nc, (float)Ns, shared_style,
                       (const float*)nullptr, (float*)nullptr);
  }
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// refresh (see refresh_needed): X = A0 V, then A <- V^T X, for the matrices whose kept spectrum reaches 4 decades below their
// norm; the others' blocks exit at once.  Before ANY spectral function of the tracked (A, V): launch_wct's apply stage and the
// three of launch_style_swap (relu5_1, C = 512: every input of 352 x 352 or smaller has N < C -- the rank-deficient case).
// always: every matrix, whatever its spectrum (measured for style-swap when the solver's tile update moved to split fp16: the
// 3 of 900 patch matches that flipped on the 32 x 32 test case flipped with the rotated matrix recomputed as well -- it is V, not
// the tracked matrix, that carries the difference; the switch stays for experiments).
static int launch_refresh(const WctCarve& w, int C, int P, int shared_style, hipStream_t s, bool always = false) {
  const size_t cc = (size_t)C * C;
  int rc;
  GemmArgs r1 = {};
  r1.A = w.A0; r1.lda = C; r1.a_kmajor = 0; r1.B = w.V; r1.ldb = C; r1.b_kmajor = 1; r1.sA = r1.sB = cc; r1.skip_shared = shared_style;
  r1.M = C; r1.N = C; r1.K = C; r1.ksplit = C; r1.out32 = w.X; r1.ldo = C; r1.s_out = cc;
  if (!always) { r1.mask_diag = w.A; r1.s_mask = cc; r1.mask_out = w.refresh; }
  if ((rc = launch_gemm(r1, 1, 2 * P, s))) return rc;
  GemmArgs r2 = {};
  r2.A = w.V; r2.lda = C; r2.a_kmajor = 1; r2.B = w.X; r2.ldb = C; r2.b_kmajor = 1; r2.sA = r2.sB = cc; r2.skip_shared = shared_style;
  r2.M = C; r2.N = C; r2.K = C; r2.ksplit = C; r2.out32 = w.A; r2.ldo = C; r2.s_out = cc;
  if (!always) r2.mask_in = w.refresh;
  rc = launch_gemm(r2, 1, 2 * P, s);
#ifdef WCT_TUNING
  // WCT_REFRESH_STATS=1 (tuning builds): how often the predicate fires -- the flags are read back after every call (a stream
  // sync: measurement only) and the running totals printed, per channel count
  if (!rc && !always && tune_set("WCT_REFRESH_STATS")) {
    static long fired[5] = {0, 0, 0, 0, 0}, seen[5] = {0, 0, 0, 0, 0};
    int flags[64];
    if (hipMemcpyAsync(flags, w.refresh, (size_t)2 * P * sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
      const int cls = C >= 512 ? 4 : (C >= 256 ? 3 : (C >= 128 ? 2 : (C >= 64 ? 1 : 0)));
      for (int m = 0; m < 2 * P; ++m) if (!skip_style_mat(m, shared_style)) { seen[cls] += 1; fired[cls] += flags[m] != 0; }
      fprintf(stderr, "refresh fired (running totals) C<=32 %ld/%ld | 64 %ld/%ld | 128 %ld/%ld | 256 %ld/%ld | 512+ %ld/%ld\n",
              fired[0], seen[0], fired[1], seen[1], fired[2], seen[2], fired[3], seen[3], fired[4], seen[4]);
    }
  }
#endif
  return rc;
}

int launch_wct(const float* content, int Nc, const float* style, int Ns, int C, int P, float alpha, int mode, float eps_in,
               half_t* out16, float* out32, void* workspace, size_t workspace_bytes, int* sweeps_dev,
               int stages, hipStream_t s, const hipStream_t* side, int nside, hipEvent_t ev_fork,
               const hipEvent_t* ev_join, int shared_style, int* eig_fail, const WctFeatStats* stats) {
  ARG_CHECK(C % 32 == 0 && C >= 32 && C <= 1024 && Nc >= 2 && Ns >= 2 && P >= 1 && P <= 32);
  ARG_CHECK(mode == WCT_MODE_NP || mode == WCT_MODE_TF);
  // the covariance kernel addresses one feature map through a buffer resource with 32-bit byte offsets
  ARG_CHECK((size_t)Nc * C * 4 < ((size_t)1 << 31) && (size_t)Ns * C * 4 < ((size_t)1 << 31));
  WctCarve w = carve(workspace, C, Nc, Ns, P);
  ARG_CHECK(workspace_bytes >= w.total);
  int rc;
  const size_t cc = (size_t)C * C;
  if (stages & WCT_STAGE_COV) {
  if ((rc = launch_means(content, Nc, style, Ns, C, P, w, false, shared_style, s, stats))) return rc;

  // covariance partials: matrix 2p+side, side 0 = content, 1 = style; slices past a side's N write zeros
  const int BT = C >= 128 ? 128 : 64;
  {
    CovArgs ca;
    ca.x[0] = content; ca.x[1] = style; ca.n[0] = Nc; ca.n[1] = Ns;
    ca.mean = w.mean; ca.scale = w.scale; ca.partial = w.cov_partial;
    ca.C = C; ca.ksplit = w.ksplit; ca.nsplit = w.nsplit; ca.ntile = cdiv(C, BT); ca.shared_style = shared_style;
    dim3 grid(ca.ntile * (ca.ntile + 1) / 2, w.nsplit, 2 * P);
    if (BT == 128) hipLaunchKernelGGL((cov_f16x2_kernel<128>), grid, dim3(256), 0, s, ca);
    else hipLaunchKernelGGL((cov_f16x2_kernel<64>), grid, dim3(256), 0, s, ca);
    HIP_TRY(hipGetLastError());
  }
  // eps_in < 0 selects the reference defaults: 1e-8 on the covariance diagonal for wct_tf
  // (ops.py:24,45,50), 1e-5 inside the spectral gains for wct_np (ops.py:92,114,127)
  const float eps_user = eps_in >= 0.f ? eps_in : (mode == WCT_MODE_TF ? 1e-8f : 1e-5f);
  const float eps = mode == WCT_MODE_TF ? eps_user : 0.f;
  hipLaunchKernelGGL(cov_finish_kernel, dim3(cdiv(C, 64), cdiv(C, 64), 2 * P), dim3(256), 0, s,
                     w.cov_partial, w.scale, w.A, C, w.nsplit, BT, 1.f / (float)(Nc - 1), 1.f / (float)(Ns - 1), eps, shared_style, w.A0);
  HIP_TRY(hipGetLastError());
  }
  if (stages & WCT_STAGE_EIG) {
    if (nside > 0 && P >= 2) {
      // The eigensolver alternates a latency-bound kernel on a few workgroups (pair problems) with a
      // chip-wide tile update.  Splitting the batch into groups on separate streams lets one group's
      // pair problems hide under the other groups' tile updates.
      // measured per 5-level step with the pivot-wave pair kernel (round 2, profiles/r02_eig_groups.txt), Jacobi ms:
      // batch 32: 1 group 31.6, 2: 30.0, 3: 28.4, 4: 28.0; batch 16: 19.9 / 19.5 / - / 18.2; batch 8: 15.6 / 15.5 / 15.2 /
      // 15.6; batch 4: 13.2 / 13.5 / - / 14.1; batch 2: 11.8 / 12.1 / - / 12.9 -- many matrices: one group's pair
      // problems hide under the other groups' tile updates; few: every extra stream only adds launch traffic
      // round 3: with the look-ahead launches a group's pair problems already run beside its own tile update, and more
      // groups only split the chip (Jacobi ms per step, 1 / 2 / 4 groups: batch 32: 20.6 / 21.0 / 22.5, batch 8: 9.3 / 11.9 / 10.0)
      static const int fused_on = tune_int("WCT_JACOBI_FUSED", 1);
      int ngrp = fused_on ? 1 : (P >= 12 ? 4 : (P >= 6 ? 2 : 1));
      if (ngrp > nside + 1) ngrp = nside + 1;
      static const int force_ngrp = tune_int("WCT_EIG_NGRP", 0);   // tuning switch
      if (force_ngrp >= 1 && force_ngrp <= 4 && force_ngrp <= nside + 1) ngrp = force_ngrp;
      JacobiGroup grp[4];
      HIP_TRY(hipEventRecord(ev_fork, s));
      size_t off = 0;
      int m0 = 0;
      for (int g = 0; g < ngrp; ++g) {
        const int n = (2 * P * (g + 1)) / ngrp - (2 * P * g) / ngrp;
        const size_t bytes = (jacobi_workspace_bytes(C, n) + 255) & ~(size_t)255;
        hipStream_t sg = g == 0 ? s : side[g - 1];
        if (g > 0) HIP_TRY(hipStreamWaitEvent(sg, ev_fork, 0));
        if ((rc = jacobi_make_group(&grp[g], w.A + (size_t)m0 * cc, w.V + (size_t)m0 * cc, C, n, (char*)w.jacobi_ws + off,
                                    bytes, sweeps_dev ? sweeps_dev + m0 : nullptr, eig_fail ? eig_fail + 2 * g : nullptr, sg, jacobi_stats_slot(eig_fail, g, C)))) return rc;
        grp[g].mat0 = m0; grp[g].shared_style = shared_style; grp[g].tol_fn = jacobi_tol_fn();
        grp[g].u_f16 = (stages & WCT_STAGE_EIG_FP32UPDATE) ? 0 : 1;
        off += bytes; m0 += n;
      }
      if ((rc = jacobi_dispatch(grp, ngrp, C))) return rc;
      for (int g = 1; g < ngrp; ++g) {
        HIP_TRY(hipEventRecord(ev_join[g - 1], side[g - 1]));
        HIP_TRY(hipStreamWaitEvent(s, ev_join[g - 1], 0));
      }
    } else {
      JacobiGroup G;
      if ((rc = jacobi_make_group(&G, w.A, w.V, C, 2 * P, w.jacobi_ws, w.jacobi_bytes, sweeps_dev, eig_fail, s, jacobi_stats_slot(eig_fail, 0, C)))) return rc;
      G.shared_style = shared_style; G.tol_fn = jacobi_tol_fn();
      G.u_f16 = (stages & WCT_STAGE_EIG_FP32UPDATE) ? 0 : 1;
      if ((rc = jacobi_dispatch(&G, 1, C))) return rc;
    }
  }
  if (!(stages & WCT_STAGE_APPLY)) return WCT_OK;

  {
    // Tw = Vc f_c(Ac) Vc^T, Tcs = Vs f_s(As) Vs^T with the first- and second-order completion of f on the residual
    // off-diagonals; the wct_np semantics shift the kept eigenvalues by eps inside the gains (ops.py:114,127), wct_tf does not.
    // One launch of each kind over all 2P matrices of the level (see spectral_open_all_kernel).
    const float shift = mode == WCT_MODE_NP ? (eps_in >= 0.f ? eps_in : 1e-5f) : 0.f;
    const int second = eig_correct_enabled() >= 2;
    const dim3 tiles(cdiv(C, 64), cdiv(C, 64), 2 * P);
    if ((rc = launch_refresh(w, C, P, shared_style, s))) return rc;
    SpecAllArgs sa = {};
    sa.A = w.A; sa.G = w.G; sa.C = C; sa.shift = shift; sa.correct = eig_correct_enabled(); sa.second = second; sa.shared_style = shared_style;
    sa.N = w.S2; float* X2 = sa.N + 2 * P * cc; sa.R = X2 + 2 * P * cc; sa.Pm = sa.R + P * cc; float* X1 = sa.Pm + P * cc;
    sa.X1 = X1; sa.X2 = X2;
    sa.mabs = w.mabs; sa.mean = w.mean; sa.bias = w.bias; sa.alpha = alpha; sa.mode = mode;
    hipLaunchKernelGGL(spectral_open_all_kernel, tiles, dim3(256), 0, s, sa);
    HIP_TRY(hipGetLastError());
    if (second) {
      GemmArgs a = {};   // X2[m] = (colouring: N[m], whitening: Pm[pair]) . N[m]
      a.A = sa.Pm; a.sA = cc; a.A_odd = sa.N + cc; a.sA_odd = 2 * cc; a.lda = C; a.a_kmajor = 0;
      a.B = sa.N; a.ldb = C; a.b_kmajor = 1; a.sB = cc; a.skip_shared = shared_style;
      a.M = C; a.N = C; a.K = C; a.ksplit = C; a.out32 = X2; a.ldo = C; a.s_out = cc;
      if ((rc = launch_gemm(a, 1, 2 * P, s))) return rc;
      GemmArgs b = {};   // X1[pair] = R[pair] . N[2 pair]   (whitening side only)
      b.A = sa.R; b.sA = cc; b.lda = C; b.a_kmajor = 0; b.B = sa.N; b.ldb = C; b.b_kmajor = 1; b.sB = 2 * cc;
      b.M = C; b.N = C; b.K = C; b.ksplit = C; b.out32 = X1; b.ldo = C; b.s_out = cc;
      if ((rc = launch_gemm(b, 1, P, s))) return rc;
      hipLaunchKernelGGL(spectral_add2_all_kernel, tiles, dim3(256), 0, s, sa);
      HIP_TRY(hipGetLastError());
    }
    GemmArgs g = {};   // X[m] = V[m] G[m]
    g.A = w.V; g.lda = C; g.a_kmajor = 0; g.B = w.G; g.ldb = C; g.b_kmajor = 1; g.sA = g.sB = cc; g.skip_shared = shared_style;
    g.M = C; g.N = C; g.K = C; g.ksplit = C; g.out32 = w.X; g.ldo = C; g.s_out = cc;
    if ((rc = launch_gemm(g, 1, 2 * P, s))) return rc;
    GemmArgs h = {};   // Tw / Tcs [m] = X[m] V[m]^T
    h.A = w.X; h.lda = C; h.a_kmajor = 0; h.B = w.V; h.ldb = C; h.b_kmajor = 0; h.sA = h.sB = cc; h.skip_shared = shared_style;
    h.M = C; h.N = C; h.K = C; h.ksplit = C; h.out32 = w.Tw; h.ldo = C; h.s_out = cc;
    if ((rc = launch_gemm(h, 1, 2 * P, s))) return rc;
  }
  {
    GemmArgs g = {};   // M = alpha (Tcs . Tw) + (1 - alpha) I, max |M| -> mabs: the blend in the product's epilogue
    g.A = w.Tcs; g.lda = C; g.a_kmajor = 0; g.B = w.Tw; g.ldb = C; g.b_kmajor = 1; g.sA = shared_style ? 0 : 2 * cc; g.sB = 2 * cc;
    g.M = C; g.N = C; g.K = C; g.ksplit = C; g.out32 = w.M; g.ldo = C; g.s_out = cc;
    g.blend = 1; g.alpha = alpha; g.mabs = w.mabs;
    if ((rc = launch_gemm(g, 1, P, s))) return rc;
  }
  {  // out[n][j] = sum_k (x[n][k]-mc[k]) M[j][k] + bias[j]
    ApplyArgs a;
    a.x = content; a.N = Nc; a.C = C; a.mean = w.mean; a.M = w.M; a.bias = w.bias;
    a.xscale = w.scale; a.mabs = w.mabs; a.out16 = out16; a.out32 = out32;
    if (C >= 128) hipLaunchKernelGGL((apply_f16x2_kernel<128, 128>), dim3(cdiv(Nc, 128), cdiv(C, 128), P), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((apply_f16x2_kernel<64, 128>), dim3(cdiv(Nc, 128), cdiv(C, 64), P), dim3(256), 0, s, a);
  }
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// AdaIN (ops.py:282-294): out = alpha*((x-mu_c)*rsqrt(var_c+eps)*sqrt(var_s)+mu_s) + (1-alpha)*x
// ---------------------------------------------------------------------------
__global__ void adain_apply_kernel(const float* x, size_t n4_per_pair, int C, const float* mean, const float* var,
                                   float alpha, float eps, half_t* out16, float* out32, int shared_style) {
  const int pair = blockIdx.y;
  const int cq = C / 4;
  const float* mp = mean + (size_t)pair * 2 * C;
  const float* vp = var + (size_t)pair * 2 * C;
  const float* ms = mean + (size_t)(shared_style ? 0 : pair) * 2 * C + C;   // style moments
  const float* vs = var + (size_t)(shared_style ? 0 : pair) * 2 * C + C;
  const size_t base = (size_t)pair * n4_per_pair;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4_per_pair; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + (base + i) * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float inv = 1.f / sqrtf(vp[c + j] + eps);
      const float y = (v[j] - mp[c + j]) * inv * sqrtf(vs[c + j]) + ms[c + j];
      o[j] = alpha * y + (1.f - alpha) * v[j];
    }
    if (out32) *reinterpret_cast<f32x4*>(out32 + (base + i) * 4) = o;
    if (out16) {
      half4 h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (half_t)o[j];
      *reinterpret_cast<half4*>(out16 + (base + i) * 4) = h;
    }
  }
}

int launch_adain(const float* content, int Nc, const float* style, int Ns, int C, int P, float alpha, float eps,
                 half_t* out16, float* out32, void* workspace, size_t workspace_bytes, hipStream_t s, int shared_style,
                 const WctFeatStats* stats) {
  ARG_CHECK(C % 4 == 0 && C <= 1024 && Nc >= 1 && Ns >= 1 && P >= 1 && P <= 32);
  WctCarve w = carve(workspace, C < 32 ? 32 : C, Nc, Ns, P);
  ARG_CHECK(workspace_bytes >= w.total);
  int rc;
  if ((rc = launch_means(content, Nc, style, Ns, C, P, w, true, shared_style, s, stats))) return rc;
  const size_t n4 = (size_t)Nc * C / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adain_apply_kernel, dim3((unsigned)blocks, P), dim3(256), 0, s, content, n4, C, w.mean, w.var, alpha, eps, out16, out32, shared_style);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// Style-swap at relu5_1 (ops.py:145-278, `--swap5`): whiten content and style, replace every
// content patch by its best-correlated (un-normalised) style patch, colour with the style.
// One content/style pair per call; the batch loop is in api.hip.
// ---------------------------------------------------------------------------
// dst[m][(i*p + j)*C + c] = src[(y*st + i)][(x*st + j)][c],  m = y*wo + x   (tf.extract_image_patches, VALID)
__global__ void im2col_kernel(const float* src, float* dst, int w, int C, int p, int st, int ho, int wo) {
  const int c4n = C / 4;
  const size_t total = (size_t)ho * wo * p * p * c4n;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    size_t t = e / c4n;
    const int j = (int)(t % p); t /= p;
    const int i = (int)(t % p); t /= p;
    const int x = (int)(t % wo), y = (int)(t / wo);
    reinterpret_cast<f32x4*>(dst)[e] =
        *reinterpret_cast<const f32x4*>(src + ((size_t)(y * st + i) * w + (x * st + j)) * C + c4 * 4);
  }
}

// inv[k] = rsqrt(max(sum_n B[n][k]^2, 1e-12))   -- tf.nn.l2_normalize(dim=3): over the PATCH axis (ops.py:233)
__global__ void patch_axis_inv_norm_kernel(const float* B, float* inv, int n, int K) {
  const int k4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k4 * 4 >= K) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < n; ++r) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(B + (size_t)r * K + k4 * 4);
    s += v * v;
  }
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = 1.f / sqrtf(fmaxf(s[j], 1e-12f));
  *reinterpret_cast<f32x4*>(inv + k4 * 4) = o;
}

// idx[m] = first argmax_n E[m][n]; one wave per row
__global__ __launch_bounds__(64) void row_argmax_kernel(const float* E, int* idx, int n) {
  const int m = blockIdx.x, lane = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < n; j += 64) {
    const float v = E[(size_t)m * n + j];
    if (v > best) { best = v; bi = j; }          // ascending j per lane keeps the first maximum
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  // a row of NaN / -inf correlations (non-finite features) leaves no winner: keep the gather in bounds
  if (lane == 0) idx[m] = (bi >= 0 && bi < n) ? bi : 0;
}

// overlap-add of the winning patches divided by the coverage count (ops.py:255-276), as a gather
__global__ void swap_reconstruct_kernel(const float* patches /* [Pn][p*p*C] */, const int* idx, float* out,
                                        int h, int w, int C, int p, int st, int ho, int wo) {
  const int c4n = C / 4;
  const size_t total = (size_t)h * w * c4n;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % c4n);
    const size_t pix = e / c4n;
    const int x = (int)(pix % w), y = (int)(pix / w);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    for (int i = 0; i < p; ++i) {
      const int yy = y - i;
      if (yy < 0 || yy % st) continue;
      const int py = yy / st;
      if (py >= ho) continue;
      for (int j = 0; j < p; ++j) {
        const int xx = x - j;
        if (xx < 0 || xx % st) continue;
        const int px = xx / st;
        if (px >= wo) continue;
        const int n = idx[py * wo + px];
        acc += *reinterpret_cast<const f32x4*>(patches + ((size_t)n * p * p + i * p + j) * C + c4 * 4);
        cnt += 1.f;
      }
    }
    reinterpret_cast<f32x4*>(out)[e] = acc / cnt;
  }
}

// out = alpha * (col + ms) + (1 - alpha) * x      (ops.py:210-213)
__global__ void swap_blend_kernel(const float* col, const float* x, const float* ms, size_t n4, int C, float alpha,
                                  half_t* out16, float* out32) {
  const int cq = C / 4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const f32x4 a = reinterpret_cast<const f32x4*>(col)[i], b = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 m = *reinterpret_cast<const f32x4*>(ms + c);
    const f32x4 o = alpha * (a + m) + (1.f - alpha) * b;
    if (out32) reinterpret_cast<f32x4*>(out32)[i] = o;
    if (out16) {
      half4 h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (half_t)o[j];
      reinterpret_cast<half4*>(out16)[i] = h;
    }
  }
}

struct SwapCarve {
  float *d3, *Wc, *Ws, *Ac, *Bs, *inv, *E, *ss, *col;
  int* idx;
  size_t total;
};

static SwapCarve swap_carve(void* base, int C, int hc, int wc, int hs, int ws, int p, int st) {
  SwapCarve w;
  const size_t K = (size_t)p * p * C;
  const size_t Mo = (size_t)((hc - p) / st + 1) * ((wc - p) / st + 1);
  const size_t Pn = (size_t)((hs - p) / st + 1) * ((ws - p) / st + 1);
  size_t off = 0;
  char* b = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) { void* q = b ? b + off : nullptr; off += align_up(bytes); return q; };
  w.d3 = (float*)take(3 * (size_t)C * 4);
  w.Wc = (float*)take((size_t)hc * wc * C * 4);
  w.Ws = (float*)take((size_t)hs * ws * C * 4);
  w.Ac = (float*)take(Mo * K * 4);
  w.Bs = (float*)take(Pn * K * 4);
  w.inv = (float*)take(K * 4);
  w.E = (float*)take(Mo * Pn * 4);
  w.idx = (int*)take(Mo * 4);
  w.ss = (float*)take((size_t)hc * wc * C * 4);
  w.col = (float*)take((size_t)hc * wc * C * 4);
  w.total = off;
  return w;
}

size_t style_swap_workspace_bytes(int C, int hc, int wc, int hs, int ws, int p, int st) {
  return wct_workspace_bytes(C, hc * wc, hs * ws, 1) + swap_carve(nullptr, C, hc, wc, hs, ws, p, st).total + 1024;
}

static inline unsigned ew_grid(size_t n) { size_t g = (n + 255) / 256; return (unsigned)(g > 4096 ? 4096 : (g ? g : 1)); }

int launch_style_swap(const float* content, int hc, int wc, const float* style, int hs, int ws, int C,
                      float alpha, int patch, int stride, float eps, half_t* out16, float* out32,
                      void* workspace, size_t workspace_bytes, hipStream_t s, int* eig_fail) {
  ARG_CHECK(C % 32 == 0 && C >= 32 && C <= 1024 && patch >= 1 && stride >= 1);
  ARG_CHECK(hc >= patch && wc >= patch && hs >= patch && ws >= patch);
  const int ho = (hc - patch) / stride + 1, wo = (wc - patch) / stride + 1;
  const int rows = (hs - patch) / stride + 1, cols = (ws - patch) / stride + 1;
  if ((ho - 1) * stride + patch != hc || (wo - 1) * stride + patch != wc) {
    wct_set_error("style-swap with patch %d stride %d maps a %dx%d feature map to %dx%d: pre-size the content "
                  "(swap_filter_fit / center_crop_to, wct.py:84-90)", patch, stride, hc, wc,
                  (ho - 1) * stride + patch, (wo - 1) * stride + patch);
    return WCT_ERR_ARG;
  }
  const int Nc = hc * wc, Ns = hs * ws, Mo = ho * wo, Pn = rows * cols, K = patch * patch * C;
  const size_t wct_bytes = align_up(wct_workspace_bytes(C, Nc, Ns, 1));
  ARG_CHECK(workspace_bytes >= style_swap_workspace_bytes(C, hc, wc, hs, ws, patch, stride));
  WctCarve w = carve(workspace, C, Nc, Ns, 1);
  SwapCarve sw = swap_carve((char*)workspace + wct_bytes, C, hc, wc, hs, ws, patch, stride);
  int rc;
  // statistics, covariances (+eps I), eigendecompositions: the same stages as wct_tf
  if ((rc = launch_wct(content, Nc, style, Ns, C, 1, alpha, WCT_MODE_TF, eps, nullptr, nullptr, workspace, wct_bytes,
                       nullptr, WCT_STAGE_COV | WCT_STAGE_EIG | WCT_STAGE_EIG_FP32UPDATE, s, nullptr, 0, nullptr, nullptr, 0, eig_fail))) return rc;
  const size_t cc = (size_t)C * C;
  // content whitening, style whitening, style colouring: S^-1/2 | S^1/2 over the kept singular values, no eps in the
  // gains (ops.py:187-189,197-198,208-209), with the first-order completion on the solver's residual
  if ((rc = launch_refresh(w, C, 1, 0, s))) return rc;
  if ((rc = launch_spectral_function(w.A, w.V, w.G, w.X, w.Tw, C, 1, 2 * cc, cc, 0, 0.f, s, w.S2))) return rc;
  if ((rc = launch_spectral_function(w.A + cc, w.V + cc, w.G + cc, w.X + cc, w.Tcs, C, 1, 2 * cc, cc, 0, 0.f, s, w.S2))) return rc;
  if ((rc = launch_spectral_function(w.A + cc, w.V + cc, w.G + cc, w.X + cc, w.T, C, 1, 2 * cc, cc, 1, 0.f, s, w.S2))) return rc;
  auto apply = [&](const float* X, int N, const float* mean, const float* T, float* out) {   // out = (X - mean) T^T
    GemmArgs g = {};
    g.A = X; g.lda = C; g.a_kmajor = 0; g.a_sub_k = mean; g.B = T; g.ldb = C; g.b_kmajor = 0;
    g.M = N; g.N = C; g.K = C; g.ksplit = C; g.out32 = out; g.ldo = C;
    return launch_gemm(g, 1, 1, s);
  };
  if ((rc = apply(content, Nc, w.mean, w.Tw, sw.Wc))) return rc;
  if ((rc = apply(style, Ns, w.mean + C, w.Tcs, sw.Ws))) return rc;
  hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid((size_t)Mo * K / 4)), dim3(256), 0, s, sw.Wc, sw.Ac, wc, C, patch, stride, ho, wo);
  hipLaunchKernelGGL(im2col_kernel, dim3(ew_grid((size_t)Pn * K / 4)), dim3(256), 0, s, sw.Ws, sw.Bs, ws, C, patch, stride, rows, cols);
  hipLaunchKernelGGL(patch_axis_inv_norm_kernel, dim3(cdiv(K / 4, 64)), dim3(64), 0, s, sw.Bs, sw.inv, Pn, K);
  HIP_TRY(hipGetLastError());
  {  // E[m][n] = sum_k Ac[m][k] inv[k] Bs[n][k]
    GemmArgs g = {};
    g.A = sw.Ac; g.lda = K; g.a_kmajor = 0; g.a_scale_k = sw.inv; g.B = sw.Bs; g.ldb = K; g.b_kmajor = 0;
    g.M = Mo; g.N = Pn; g.K = K; g.ksplit = K; g.out32 = sw.E; g.ldo = Pn;
    if ((rc = launch_gemm(g, 1, 1, s))) return rc;
  }
  hipLaunchKernelGGL(row_argmax_kernel, dim3(Mo), dim3(64), 0, s, sw.E, sw.idx, Pn);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(swap_reconstruct_kernel, dim3(ew_grid((size_t)Nc * C / 4)), dim3(256), 0, s, sw.Bs, sw.idx, sw.ss,
                     hc, wc, C, patch, stride, ho, wo);
  HIP_TRY(hipGetLastError());
  {  // col = ss . Tcol^T
    GemmArgs g = {};
    g.A = sw.ss; g.lda = C; g.a_kmajor = 0; g.B = w.T; g.ldb = C; g.b_kmajor = 0;
    g.M = Nc; g.N = C; g.K = C; g.ksplit = C; g.out32 = sw.col; g.ldo = C;
    if ((rc = launch_gemm(g, 1, 1, s))) return rc;
  }
  hipLaunchKernelGGL(swap_blend_kernel, dim3(ew_grid((size_t)Nc * C / 4)), dim3(256), 0, s, sw.col, content, w.mean + C,
                     (size_t)Nc * C / 4, C, alpha, out16, out32);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
