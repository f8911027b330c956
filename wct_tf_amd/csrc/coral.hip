// CORAL colour preservation (coral.py:13-39, utils.py:87-90) -- the per-pixel parts.
// Both kernels are HBM-bound byte streams: 3 B/pixel in for the moments, 3 B in +
// 3 B (uint8) or 24 B (float64) out for the apply.
#include "common.h"

// exact integer moments: sum x_c (3) and sum x_i x_j (6), per block then reduced
__global__ __launch_bounds__(256) void coral_stats_kernel(const uint8_t* img, size_t npix, unsigned long long* partial) {
  unsigned long long acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned r = img[i * 3], g = img[i * 3 + 1], b = img[i * 3 + 2];
    acc[0] += r; acc[1] += g; acc[2] += b;
    acc[3] += r * r; acc[4] += r * g; acc[5] += r * b;
    acc[6] += g * g; acc[7] += g * b; acc[8] += b * b;
  }
  __shared__ unsigned long long red[4][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    unsigned long long v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int k = threadIdx.x;
    partial[(size_t)blockIdx.x * 9 + k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  }
}

__global__ void coral_stats_finish_kernel(const unsigned long long* partial, int nblocks, unsigned long long* out9) {
  const int k = threadIdx.x;
  if (k >= 9) return;
  unsigned long long s = 0;
  for (int i = 0; i < nblocks; ++i) s += partial[(size_t)i * 9 + k];
  out9[k] = s;
}

int launch_coral_stats(const uint8_t* img, size_t npix, unsigned long long* partial, int nblocks,
                       unsigned long long* out9, hipStream_t s) {
  hipLaunchKernelGGL(coral_stats_kernel, dim3(nblocks), dim3(256), 0, s, img, npix, partial);
  hipLaunchKernelGGL(coral_stats_finish_kernel, dim3(1), dim3(64), 0, s, partial, nblocks, out9);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

__global__ void coral_apply_kernel(const uint8_t* src, size_t npix, CoralApplyArgs a, uint8_t* out_u8, double* out_f64) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    double xn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xn[c] = ((double)src[i * 3 + c] / 255.0 - a.src_mean[c]) / a.src_std[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double y = a.M[c * 3 + 0] * xn[0];
      y += a.M[c * 3 + 1] * xn[1];
      y += a.M[c * 3 + 2] * xn[2];
      y = y * a.tgt_std[c] + a.tgt_mean[c];
      if (out_f64) out_f64[i * 3 + c] = y;
      if (out_u8) {
        double v = fmin(fmax(y, 0.0), 1.0) * 255.0;
        out_u8[i * 3 + c] = (uint8_t)v;
      }
    }
  }
}

int launch_coral_apply(const uint8_t* src, size_t npix, const CoralApplyArgs& a, uint8_t* out_u8,
                       double* out_f64, hipStream_t s) {
  size_t blocks = (npix + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(coral_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, npix, a, out_u8, out_f64);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
