// Decoder training (model.py:178-223, train.py): the backward pass and the optimiser for gfx950.
//
// The forward pass of a training step reuses the inference kernels (fp16 activations, fp32 accumulate) and
// keeps every activation.  The backward pass runs in fp32:
//   dgrad  dXp[p'][ci] = sum_{tap,co} dY[p' - tap][co] W[tap][ci][co]     im2col("full", zero pad) x GEMM
//          followed by the adjoint of the 1-px REFLECT pad (border rows/columns fold back inside)
//   wgrad  dW[tap][ci][co] = sum_p Xp[p + tap][ci] dY[p][co]              im2col(reflect pad) ^T x GEMM, split-K
// both on a split-operand fp16-MFMA GEMM (gemm_f16x2_kernel below).  ReLU masks come from the saved activations, the
// x2 nearest upsample and the ceil-mode 2x2 max-pool have their adjoints here, as have the three losses of
// model.py:181-194 (feature MSE, pixel MSE, total variation) and Adam (tf.train.AdamOptimizer, model.py:199).
// Every reduction is two-stage with a fixed order: a training step is bit-reproducible.
#include "common.h"

static inline int tr_blocks(size_t n, int per = 256, int cap = 8192) {
  size_t b = (n + per - 1) / per;
  return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}

__device__ __forceinline__ int tr_reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// ---------------------------------------------------------------------------
// im2col of an fp16 activation with the conv's own 1-px REFLECT pad (and the folded x2 upsample):
// col[b][y][x][(tap, ci)] = act[b][refl(y+ky-1)][refl(x+kx-1)][ci]     (H, W = conv output = padded-input size - 2)
// ---------------------------------------------------------------------------
__global__ void im2col_act_kernel(const half_t* x, float* col, int B, int H, int W, int C, int upsample) {
  const int Hin = upsample ? H / 2 : H, Win = upsample ? W / 2 : W;
  const size_t total = (size_t)B * H * W * 9 * (C / 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    size_t t = i / (C / 4);
    const int tap = (int)(t % 9); t /= 9;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int b = (int)(t / H);
    int iy = tr_reflect(yy + tap / 3 - 1, H), ix = tr_reflect(xx + tap % 3 - 1, W);
    if (upsample) { iy >>= 1; ix >>= 1; }
    const half4 v = *reinterpret_cast<const half4*>(x + (((size_t)b * Hin + iy) * Win + ix) * C + c4 * 4);
    f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    *reinterpret_cast<f32x4*>(col + i * 4) = o;
  }
}
int launch_im2col_act(const half_t* x, float* col, int B, int H, int W, int C, int upsample, hipStream_t s) {
  ARG_CHECK(C % 4 == 0 && H >= 2 && W >= 2);
  hipLaunchKernelGGL(im2col_act_kernel, dim3(tr_blocks((size_t)B * H * W * 9 * (C / 4))), dim3(256), 0, s, x, col, B, H, W, C, upsample);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// "full" im2col of a gradient for the data gradient: over the PADDED output grid (H+2) x (W+2),
// col[b][y'][x'][(tap, co)] = g[b][y' - ky][x' - kx][co]  or 0 outside
__global__ void im2col_grad_kernel(const float* g, float* col, int B, int H, int W, int C) {
  const int Hp = H + 2, Wp = W + 2;
  const size_t total = (size_t)B * Hp * Wp * 9 * (C / 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    size_t t = i / (C / 4);
    const int tap = (int)(t % 9); t /= 9;
    const int xp = (int)(t % Wp); t /= Wp;
    const int yp = (int)(t % Hp);
    const int b = (int)(t / Hp);
    const int y = yp - tap / 3, x = xp - tap % 3;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < H && x >= 0 && x < W) o = *reinterpret_cast<const f32x4*>(g + (((size_t)b * H + y) * W + x) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(col + i * 4) = o;
  }
}
int launch_im2col_grad(const float* g, float* col, int B, int H, int W, int C, hipStream_t s) {
  ARG_CHECK(C % 4 == 0);
  hipLaunchKernelGGL(im2col_grad_kernel, dim3(tr_blocks((size_t)B * (H + 2) * (W + 2) * 9 * (C / 4))), dim3(256), 0, s, g, col, B, H, W, C);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// adjoint of the 1-px REFLECT pad: gp [B][H+2][W+2][C] -> g [B][H][W][C]; padded row 0 mirrors row 1 of the
// image, padded row H+1 mirrors row H-2 (same for columns)
__global__ void reflect_fold_kernel(const float* gp, float* g, int B, int H, int W, int C) {
  const int Wp = W + 2, Hp = H + 2;
  const size_t total = (size_t)B * H * W * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + 1; if (y == 1) ys[ny++] = 0; if (y == H - 2) ys[ny++] = H + 1;
    xs[nx++] = x + 1; if (x == 1) xs[nx++] = 0; if (x == W - 2) xs[nx++] = W + 1;
    float acc = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int e = 0; e < nx; ++e) acc += gp[(((size_t)b * Hp + ys[a]) * Wp + xs[e]) * C + c];
    g[i] = acc;
  }
}
int launch_reflect_fold(const float* gp, float* g, int B, int H, int W, int C, hipStream_t s) {
  ARG_CHECK(H >= 2 && W >= 2);
  hipLaunchKernelGGL(reflect_fold_kernel, dim3(tr_blocks((size_t)B * H * W * C)), dim3(256), 0, s, gp, g, B, H, W, C);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ReLU backward: g *= (act > 0); the activation is the saved (post-ReLU) output, fp16 or fp32
__global__ void relu_mask16_kernel(float* g, const half_t* act, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (!((float)act[i] > 0.f)) g[i] = 0.f;
}
__global__ void relu_mask32_kernel(float* g, const float* act, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (!(act[i] > 0.f)) g[i] = 0.f;
}
int launch_relu_mask16(float* g, const half_t* act, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(relu_mask16_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, g, act, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_relu_mask32(float* g, const float* act, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(relu_mask32_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, g, act, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// adjoint of UpSampling2D x2 nearest: gs[y][x] = sum of the 2x2 block of gb
__global__ void upsample_adjoint_kernel(const float* gb, float* gs, int B, int h, int w, int C) {
  const size_t total = (size_t)B * h * w * C;
  const int W2 = 2 * w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    const float* p = gb + (((size_t)b * 2 * h + 2 * y) * W2 + 2 * x) * C + c;
    gs[i] = (p[0] + p[C]) + (p[(size_t)W2 * C] + p[(size_t)W2 * C + C]);
  }
}
int launch_upsample_adjoint(const float* gb, float* gs, int B, int h, int w, int C, hipStream_t s) {
  hipLaunchKernelGGL(upsample_adjoint_kernel, dim3(tr_blocks((size_t)B * h * w * C)), dim3(256), 0, s, gb, gs, B, h, w, C);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// adjoint of the 2x2/2 'same' max-pool: the gradient of a pooled cell goes to the first maximum of its window
// (row-major, as tf.nn.max_pool's gradient picks it); `pre` is the pool's fp16 input
__global__ void maxpool_adjoint_kernel(const half_t* pre, const float* gpool, float* gpre, int B, int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const size_t total = (size_t)B * Ho * Wo * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    size_t t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float best = -3.0e38f; int arg = -1;
    size_t idx[4]; bool ok[4];
    for (int d = 0; d < 4; ++d) {
      const int y = 2 * oy + (d >> 1), x = 2 * ox + (d & 1);
      ok[d] = y < H && x < W;
      idx[d] = (((size_t)b * H + (ok[d] ? y : 0)) * W + (ok[d] ? x : 0)) * C + c;
      if (ok[d]) { const float v = (float)pre[idx[d]]; if (v > best) { best = v; arg = d; } }
    }
    const float gv = gpool[i];
    for (int d = 0; d < 4; ++d) if (ok[d]) gpre[idx[d]] = d == arg ? gv : 0.f;
  }
}
int launch_maxpool_adjoint(const half_t* pre, const float* gpool, float* gpre, int B, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(maxpool_adjoint_kernel, dim3(tr_blocks((size_t)B * ((H + 1) / 2) * ((W + 1) / 2) * C)), dim3(256), 0, s, pre, gpool, gpre, B, H, W, C);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// two-stage sums (fixed order): partial[blockIdx.x] = sum over the block's strided elements
// ---------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void block_sum_store(double v, double* partial) {
  __shared__ double red[256];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
// MSE: loss = weight * mean((a-b)^2); grad (w.r.t. a) = 2 weight / n * (a - b), written or accumulated
__global__ void mse_kernel(const float* a, const float* b, size_t n, float gscale, float* grad, int accumulate, double* partial) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    acc += (double)d * d;
    if (grad) grad[i] = (accumulate ? grad[i] : 0.f) + gscale * d;
  }
  block_sum_store<void>(acc, partial);
}
// total variation (tf.image.total_variation): sum |x[y+1]-x[y]| + |x[x+1]-x[x]| per image; grad accumulated
__global__ void tv_kernel(const float* x, int B, int H, int W, int C, float gscale, float* grad, double* partial) {
  const size_t total = (size_t)B * H * W * C;
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i / C;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const float v = x[i];
    float gsum = 0.f;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    if (yy + 1 < H) { const float d = x[i + (size_t)W * C] - v; acc += fabsf(d); gsum -= sgn(d); }
    if (xx + 1 < W) { const float d = x[i + C] - v; acc += fabsf(d); gsum -= sgn(d); }
    if (yy > 0) gsum += sgn(v - x[i - (size_t)W * C]);
    if (xx > 0) gsum += sgn(v - x[i - C]);
    if (grad) grad[i] += gscale * gsum;
  }
  block_sum_store<void>(acc, partial);
}
__global__ void finish_sum_kernel(const double* partial, int n, double scale, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    *out = (float)(s * scale);
  }
}
int launch_mse(const float* a, const float* b, size_t n, float weight, float* grad, int accumulate, double* partial /* [1024] */,
               float* loss_out, hipStream_t s) {
  const int nb = tr_blocks(n, 256, 1024);
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, s, a, b, n, 2.f * weight / (float)n, grad, accumulate, partial);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)weight / (double)n, loss_out);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_tv(const float* x, int B, int H, int W, int C, float weight, float* grad, double* partial, float* loss_out, hipStream_t s) {
  const int nb = tr_blocks((size_t)B * H * W * C, 256, 1024);
  // loss = weight * mean_b(tv_b) = weight / B * sum
  hipLaunchKernelGGL(tv_kernel, dim3(nb), dim3(256), 0, s, x, B, H, W, C, weight / (float)B, weight != 0.f ? grad : nullptr, partial);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(64), 0, s, partial, nb, (double)weight / (double)B, loss_out);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// bias gradient: column sums of g [rows][C] in two fixed-order stages; split-K reduction of wgrad partials
// ---------------------------------------------------------------------------
__global__ void colsum_rows_kernel(const float* g, size_t rows, int C, int nslab, float* partial) {
  // 256 threads = rg row groups x min(C, 256) channels; the row groups of a slab are added in a fixed order
  __shared__ float red[256];
  const int slab = blockIdx.x;
  const size_t per = (rows + nslab - 1) / nslab;
  const size_t r0 = slab * per, r1 = r0 + per < rows ? r0 + per : rows;
  const int cw = C < 256 ? C : 256, rg = 256 / cw;
  const int c_in = threadIdx.x % cw, grp = threadIdx.x / cw;
  for (int c0 = 0; c0 < C; c0 += cw) {
    const int c = c0 + c_in;
    float acc = 0.f;
    if (grp < rg && c < C)
      for (size_t r = r0 + grp; r < r1; r += rg) acc += g[r * C + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (grp == 0 && c < C) {
      for (int k = 1; k < rg; ++k) acc += red[k * cw + c_in];
      partial[(size_t)slab * C + c] = acc;
    }
    __syncthreads();
  }
}
__global__ void reduce_slabs_kernel(const float* partial, size_t n, int nslab, float* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += partial[(size_t)k * n + i];
    out[i] = s;
  }
}
int launch_bias_grad(const float* g, size_t rows, int C, float* partial /* [256][C] */, float* out, hipStream_t s) {
  const int nslab = rows < 256 ? (int)rows : 256;
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(nslab), dim3(256), 0, s, g, rows, C, nslab, partial);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(tr_blocks(C)), dim3(256), 0, s, partial, (size_t)C, nslab, out);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_reduce_slabs(const float* partial, size_t n, int nslab, float* out, hipStream_t s) {
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, partial, n, nslab, out);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// Adam (tf.train.AdamOptimizer): lr_t = lr sqrt(1-b2^t)/(1-b1^t); w -= lr_t m / (sqrt(v) + eps)
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* w, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2, float eps) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    w[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}
int launch_adam(float* w, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2, float eps, hipStream_t s) {
  hipLaunchKernelGGL(adam_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, w, m, v, g, n, lr_t, b1, b2, eps);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// weight re-layouts on the device (after every optimiser step)
// ---------------------------------------------------------------------------
// fp32 HWIO -> fp16 MFMA A-fragments [cout/32][tap][cin/16][lane][8]  (the layout of api.hip::pack_conv)
__global__ void pack_conv_frag_kernel(const float* w, half_t* frag, int cin, int cout) {
  const size_t total = (size_t)cout * 9 * cin;
  const int c16 = cin / 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    size_t f = i >> 9;                                  // fragment index ((T*9 + tap)*c16 + k16)
    const int k16 = (int)(f % c16); f /= c16;
    const int tap = (int)(f % 9);
    const int T = (int)(f / 9);
    const int co = T * 32 + (lane & 31), ci = k16 * 16 + (lane >> 5) * 8 + j;
    frag[i] = (half_t)w[((size_t)tap * cin + ci) * cout + co];
  }
}
// fp32 HWIO -> the Winograd fragments of conv_wino.hip, [cout/32][f*3+kx][cin/16][lane][8] of U_f[kx] = sum_ky G[f][ky] g[ky][kx]
// (api.hip::pack_conv computes the same values on the host: the sum in double, one rounding to fp16)
__global__ void pack_conv_wino_frag_kernel(const float* w, half_t* frag, int cin, int cout) {
  const size_t total = (size_t)cout * 12 * cin;
  const int c16 = cin / 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    size_t fr = i >> 9;                                 // fragment index ((T*12 + f*3 + kx)*c16 + k16)
    const int k16 = (int)(fr % c16); fr /= c16;
    const int t12 = (int)(fr % 12);
    const int T = (int)(fr / 12);
    const int f = t12 / 3, kx = t12 % 3;
    const int co = T * 32 + (lane & 31), ci = k16 * 16 + (lane >> 5) * 8 + j;
    const double g0 = w[((size_t)(0 * 3 + kx) * cin + ci) * cout + co], g1 = w[((size_t)(1 * 3 + kx) * cin + ci) * cout + co],
                 g2 = w[((size_t)(2 * 3 + kx) * cin + ci) * cout + co];
    const double u = f == 0 ? g0 : f == 1 ? 0.5 * g0 + 0.5 * g1 + 0.5 * g2 : f == 2 ? 0.5 * g0 - 0.5 * g1 + 0.5 * g2 : g2;
    frag[i] = (half_t)(float)u;
  }
}
// 64 -> 3 output conv: A-fragments [k-step 4][lane][8] with row tap*3+co (27 of 32), see ConvLastArgs
__global__ void pack_last_frag_kernel(const float* w, half_t* frag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * 64 * 8) return;
  const int j = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
  const int row = lane & 31, cin = ks * 16 + (lane >> 5) * 8 + j;
  frag[i] = row < 27 ? (half_t)w[((size_t)(row / 3) * 64 + cin) * 3 + row % 3] : (half_t)0.f;
}
// HWIO [tap][ci][co] -> [(tap, co)][ci]: the B operand of the data-gradient GEMM; co runs to cout_pad >= cout
// (zero rows), so that K = 9 cout_pad is a multiple of 4 for the 3-channel output conv
__global__ void transpose_w_kernel(const float* w, float* wt, int cin, int cout, int cout_pad) {
  const size_t total = (size_t)9 * cin * cout_pad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    size_t t = i / cin;
    const int co = (int)(t % cout_pad);
    const int tap = (int)(t / cout_pad);
    wt[i] = co < cout ? w[((size_t)tap * cin + ci) * cout + co] : 0.f;
  }
}
// [n][3] -> [n][4] with a zero fourth channel
__global__ void pad3to4_kernel(const float* x, float* y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 o = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
    *reinterpret_cast<f32x4*>(y + i * 4) = o;
  }
}
int launch_pad3to4(const float* x, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(pad3to4_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, x, y, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_pack_conv_frag(const float* w, half_t* frag, int cin, int cout, hipStream_t s) {
  ARG_CHECK(cin % 16 == 0 && cout % 32 == 0);
  hipLaunchKernelGGL(pack_conv_frag_kernel, dim3(tr_blocks((size_t)cout * 9 * cin)), dim3(256), 0, s, w, frag, cin, cout);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_pack_conv_wino_frag(const float* w, half_t* frag, int cin, int cout, hipStream_t s) {
  ARG_CHECK(cin % 16 == 0 && cout % 32 == 0);
  hipLaunchKernelGGL(pack_conv_wino_frag_kernel, dim3(tr_blocks((size_t)cout * 12 * cin)), dim3(256), 0, s, w, frag, cin, cout);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_pack_last_frag(const float* w, half_t* frag, hipStream_t s) {
  hipLaunchKernelGGL(pack_last_frag_kernel, dim3(8), dim3(256), 0, s, w, frag);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_transpose_w(const float* w, float* wt, int cin, int cout, int cout_pad, hipStream_t s) {
  hipLaunchKernelGGL(transpose_w_kernel, dim3(tr_blocks((size_t)9 * cin * cout_pad)), dim3(256), 0, s, w, wt, cin, cout, cout_pad);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// conv1_1 (folded preprocess) data gradient: 64 -> 3.  g [B][H][W][64] masked; wf = folded weights [27][64]
// (tap*3 + c major); output over the PADDED grid gp [B][H+2][W+2][3] (fold it with launch_reflect_fold)
__global__ void conv_first_dgrad_kernel(const float* g, const float* wf, float* gp, int B, int H, int W) {
  const int Hp = H + 2, Wp = W + 2;
  const size_t total = (size_t)B * Hp * Wp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    const int xp = (int)(t % Wp); t /= Wp;
    const int yp = (int)(t % Hp);
    const int b = (int)(t / Hp);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int y = yp - tap / 3, x = xp - tap % 3;
      if (y < 0 || y >= H || x < 0 || x >= W) continue;
      const float* gr = g + (((size_t)b * H + y) * W + x) * 64;
      const float* w0 = wf + (tap * 3 + 0) * 64, *w1 = wf + (tap * 3 + 1) * 64, *w2 = wf + (tap * 3 + 2) * 64;
      for (int co = 0; co < 64; ++co) { const float gv = gr[co]; a0 += gv * w0[co]; a1 += gv * w1[co]; a2 += gv * w2[co]; }
    }
    gp[i * 3 + 0] = a0; gp[i * 3 + 1] = a1; gp[i * 3 + 2] = a2;
  }
}
int launch_conv_first_dgrad(const float* g, const float* wf, float* gp, int B, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(conv_first_dgrad_kernel, dim3(tr_blocks((size_t)B * (H + 2) * (W + 2), 256, 65535)), dim3(256), 0, s, g, wf, gp, B, H, W);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// Split-operand GEMM for the backward pass: D[m][n] = sum_k A(m,k) B(k,n) / (sa sb) on v_mfma_f32_32x32x16_f16.
// fp32 operands are scaled by a power of two and split into fp16 hi + lo at staging (22 significand bits, as in
// the covariance / apply kernels of wct.hip); an operand whose values are exact fp16 numbers (saved activations)
// skips its lo half.  A(m,k) = AKM ? A[k*lda+m] : A[m*lda+k];  B(k,n) = BKM ? B[k*ldb+n] : B[n*ldb+k].
// Block = 128 x BN tile, 256 threads = 2x2 waves, K-stage 32, split-K over blockIdx.z (partials reduced in a fixed
// order by the caller).  The data gradient runs <0,1> (rows of the im2col'ed gradient x transposed weights), the
// weight gradient <1,1> (im2col'ed activations^T x gradient).
// ---------------------------------------------------------------------------
struct SplitGemmArgs {
  const float* A; int lda; const float* B; int ldb;
  int M, N, K, ksplit;
  const float* a_scale; const float* b_scale;    // device scalars (power of two) or null = 1
  int a_exact16, b_exact16;                       // operand values are exact fp16 numbers: no lo half
  float* out; int ldo; size_t out_split_stride;
};

template <int BX, bool KM>
struct SplitStage {
  // one operand tile: BX rows x 32 k.  KM (k-major source): thread = (row, k-group) with 32*BX/256 strided scalar
  // loads, coalesced over the rows; row-major source: thread = 16-B pieces of 8 consecutive k
  static constexpr int N = KM ? BX * 32 / 256 : (BX * 4 / 256) * 8;
  float v[N];
  __device__ __forceinline__ void load(const float* src, int ld, int x0, int X, int k0, int kend, int tid) {
    if (KM) {
      const int r = tid % BX, kg = tid / BX;
      const bool ok = x0 + r < X;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int k = k0 + kg * N + j;
        v[j] = (ok && k < kend) ? src[(size_t)k * ld + x0 + r] : 0.f;
      }
    } else {
#pragma unroll
      for (int i = 0; i < N / 8; ++i) {
        const int item = tid + i * 256, row = item >> 2, kq = item & 3;
        const bool ok = x0 + row < X;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = k0 + kq * 8 + h * 4;
          f32x4 t = {0.f, 0.f, 0.f, 0.f};
          if (ok && k < kend) t = *reinterpret_cast<const f32x4*>(src + (size_t)(x0 + row) * ld + k);   // K % 4 == 0
#pragma unroll
          for (int j = 0; j < 4; ++j) v[i * 8 + h * 4 + j] = t[j];
        }
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* hi, unsigned char* lo, float sc, bool exact16, int tid) const {
#pragma unroll
    for (int q = 0; q < N / 8; ++q) {
      int row, chunk;
      if (KM) { row = tid % BX; chunk = (tid / BX) * (N / 8) + q; }
      else { const int item = tid + q * 256; row = item >> 2; chunk = item & 3; }
      half8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = v[q * 8 + j] * sc;
        h[j] = (half_t)x;
        l[j] = (half_t)(x - (float)h[j]);
      }
      const int off = (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16;
      *reinterpret_cast<half8*>(hi + off) = h;
      if (!exact16) *reinterpret_cast<half8*>(lo + off) = l;
    }
  }
};

template <int BN, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_f16x2_kernel(SplitGemmArgs p) {
  constexpr int BM = 128, TM = 2, TN = BN / 64;
  __shared__ __attribute__((aligned(16))) unsigned char la[2][BM * 64];
  __shared__ __attribute__((aligned(16))) unsigned char lb[2][BN * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int split = blockIdx.z;
  const int kbeg = split * p.ksplit, kend = min(p.K, kbeg + p.ksplit);
  const float sa = p.a_scale ? *p.a_scale : 1.f, sb = p.b_scale ? *p.b_scale : 1.f;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  SplitStage<BM, AKM> sA;
  SplitStage<BN, BKM> sB;
  if (kbeg < kend) { sA.load(p.A, p.lda, m0, p.M, kbeg, kend, tid); sB.load(p.B, p.ldb, n0, p.N, kbeg, kend, tid); }
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    sA.store(la[0], la[1], sa, p.a_exact16, tid);
    sB.store(lb[0], lb[1], sb, p.b_exact16, tid);
    __syncthreads();
    if (k0 + 32 < kend) { sA.load(p.A, p.lda, m0, p.M, k0 + 32, kend, tid); sB.load(p.B, p.ldb, n0, p.N, k0 + 32, kend, tid); }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 2 + (lane >> 5);
      half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + (lane & 31);
        const int off = (r * 4 + (chunk ^ ((r >> 2) & 3))) * 16;
        ah[i] = *reinterpret_cast<const half8*>(la[0] + off);
        if (!p.a_exact16) al[i] = *reinterpret_cast<const half8*>(la[1] + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + (lane & 31);
        const int off = (r * 4 + (chunk ^ ((r >> 2) & 3))) * 16;
        bh[j] = *reinterpret_cast<const half8*>(lb[0] + off);
        if (!p.b_exact16) bl[j] = *reinterpret_cast<const half8*>(lb[1] + off);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (!p.a_exact16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          if (!p.b_exact16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  const float inv = 1.f / (sa * sb);
  float* out = p.out + (size_t)split * p.out_split_stride;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (gm < p.M && gn < p.N) out[(size_t)gm * p.ldo + gn] = acc[i][j][r] * inv;
      }
    }
}

template <bool AKM, bool BKM>
static int launch_split_gemm(const SplitGemmArgs& a, int nsplit, hipStream_t s) {
  if (a.N > 64) {
    dim3 grid(cdiv(a.N, 128), cdiv(a.M, 128), nsplit);
    hipLaunchKernelGGL((gemm_f16x2_kernel<128, AKM, BKM>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid(cdiv(a.N, 64), cdiv(a.M, 128), nsplit);
    hipLaunchKernelGGL((gemm_f16x2_kernel<64, AKM, BKM>), grid, dim3(256), 0, s, a);
  }
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// power-of-two scale that brings max |x| to [8192, 16384) (1 for an all-zero / non-finite input); the max of
// non-negative floats is the max of their bit patterns, so the atomics commute
__global__ void absmax_bits_kernel(const float* x, size_t n4, unsigned* bits) {
  __shared__ float red[4];
  float m = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m < 3e38f) atomicMax(bits, __float_as_uint(m));          // one atomic per block
  }
}
__global__ void scale_from_bits_kernel(const unsigned* bits, float* scale) {
  const float m = __uint_as_float(*bits);
  float sc = 1.f;
  if (m > 0.f && m < 1e30f) { int e; frexpf(m, &e); sc = ldexpf(1.f, 14 - e); }
  *scale = sc;
}
// scratch: 2 dwords (bits, scale); n must be a multiple of 4 (every gradient tensor here is)
int launch_pow2_scale(const float* x, size_t n, void* scratch, hipStream_t s) {
  ARG_CHECK(n % 4 == 0);
  unsigned* bits = reinterpret_cast<unsigned*>(scratch);
  float* scale = reinterpret_cast<float*>(scratch) + 1;
  HIP_TRY(hipMemsetAsync(bits, 0, sizeof(unsigned), s));
  hipLaunchKernelGGL(absmax_bits_kernel, dim3(tr_blocks(n / 4, 256 * 8, 512)), dim3(256), 0, s, x, n / 4, bits);
  hipLaunchKernelGGL(scale_from_bits_kernel, dim3(1), dim3(1), 0, s, bits, scale);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// the two GEMMs of a conv layer's backward pass (scratch: 2 dwords for the gradient's power-of-two scale)
// ---------------------------------------------------------------------------
// data gradient w.r.t. the conv's (unpadded, possibly upsampled) input: gin [B][H][W][cin]
//   g [B][H][W][cout] (already masked), wt = transpose_w(W) [(tap,co)][cin]; col / gp are workspaces
int launch_conv_dgrad(const float* g, const float* wt, int B, int H, int W, int cin, int cout,
                      float* col /* B*(H+2)*(W+2)*9*cout */, float* gp /* B*(H+2)*(W+2)*cin */, float* gin,
                      void* scratch, hipStream_t s) {
  int rc;
  const float* gscale = reinterpret_cast<const float*>(scratch) + 1;       // launch_pow2_scale(g) ran on this scratch
  if ((rc = launch_im2col_grad(g, col, B, H, W, cout, s))) return rc;
  SplitGemmArgs a = {};
  const int Mp = B * (H + 2) * (W + 2), K = 9 * cout;
  a.A = col; a.lda = K; a.B = wt; a.ldb = cin;
  a.M = Mp; a.N = cin; a.K = K; a.ksplit = (K + 31) / 32 * 32;
  a.a_scale = gscale; a.b_scale = nullptr; a.a_exact16 = 0; a.b_exact16 = 0;
  a.out = gp; a.ldo = cin; a.out_split_stride = 0;
  if ((rc = launch_split_gemm<false, true>(a, 1, s))) return rc;
  return launch_reflect_fold(gp, gin, B, H, W, cin, s);
}
// weight gradient dW [9*cin][cout] (HWIO): x16 is the conv's fp16 input activation (before the folded upsample);
// g has row pitch ldg >= cout (the 3-channel gradient comes padded to 4)
int launch_conv_wgrad(const half_t* x16, int upsample, const float* g, int ldg, int B, int H, int W, int cin, int cout,
                      float* col /* B*H*W*9*cin */, float* partial /* nsplit*9*cin*cout */, int nsplit, float* dw,
                      void* scratch, hipStream_t s) {
  int rc;
  const float* gscale = reinterpret_cast<const float*>(scratch) + 1;       // launch_pow2_scale(g) ran on this scratch
  const int P = B * H * W;
  if ((rc = launch_im2col_act(x16, col, B, H, W, cin, upsample, s))) return rc;
  SplitGemmArgs a = {};
  a.A = col; a.lda = 9 * cin; a.B = g; a.ldb = ldg;
  a.M = 9 * cin; a.N = cout; a.K = P;
  a.ksplit = ((P + nsplit - 1) / nsplit + 31) / 32 * 32;
  a.a_scale = nullptr; a.b_scale = gscale; a.a_exact16 = 1; a.b_exact16 = 0;      // activations are exact fp16 values
  a.out = partial; a.ldo = cout; a.out_split_stride = (size_t)9 * cin * cout;
  const int ns = (P + a.ksplit - 1) / a.ksplit;
  if ((rc = launch_split_gemm<true, true>(a, ns, s))) return rc;
  return launch_reduce_slabs(partial, (size_t)9 * cin * cout, ns, dw, s);
}
int conv_wgrad_splits(int B, int H, int W) {
  const long P = (long)B * H * W;
  long n = P / 2048;
  return (int)(n < 1 ? 1 : (n > 128 ? 128 : n));
}
