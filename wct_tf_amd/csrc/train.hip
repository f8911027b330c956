// Decoder training (model.py:178-223, train.py): the backward pass and the optimiser for gfx950.
//
// The forward pass of a training step reuses the inference kernels (fp16 activations, fp32 accumulate) and
// keeps every activation.  The backward pass runs in fp32:
//   dgrad  dXp[p'][ci] = sum_{tap,co} dY[p' - tap][co] W[tap][ci][co]     im2col("full", zero pad) x GEMM
//          followed by the adjoint of the 1-px REFLECT pad (border rows/columns fold back inside)
//   wgrad  dW[tap][ci][co] = sum_p Xp[p + tap][ci] dY[p][co]              im2col(reflect pad) ^T x GEMM, split-K
// both on the fp32 MFMA GEMM of wct.hip (gemm_f32_kernel).  ReLU masks come from the saved activations, the
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
  const int slab = blockIdx.x;
  const size_t per = (rows + nslab - 1) / nslab;
  const size_t r0 = slab * per, r1 = r0 + per < rows ? r0 + per : rows;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = 0.f;
    for (size_t r = r0; r < r1; ++r) acc += g[r * C + c];
    partial[(size_t)slab * C + c] = acc;
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

// ---------------------------------------------------------------------------
// the two GEMMs
// ---------------------------------------------------------------------------
// data gradient w.r.t. the conv's (unpadded, possibly upsampled) input: gin [B][H][W][cin]
//   g [B][H][W][cout] (already masked), wt = transpose_w(W) [(tap,co)][cin]; col / gp are workspaces
int launch_conv_dgrad(const float* g, const float* wt, int B, int H, int W, int cin, int cout,
                      float* col /* B*(H+2)*(W+2)*9*cout */, float* gp /* B*(H+2)*(W+2)*cin */, float* gin, hipStream_t s) {
  int rc;
  if ((rc = launch_im2col_grad(g, col, B, H, W, cout, s))) return rc;
  GemmArgs a = {};
  const int Mp = B * (H + 2) * (W + 2), K = 9 * cout;
  a.A = col; a.lda = K; a.a_kmajor = 0;
  a.B = wt; a.ldb = cin; a.b_kmajor = 1;
  a.M = Mp; a.N = cin; a.K = K; a.ksplit = K;
  a.out32 = gp; a.ldo = cin;
  if ((rc = launch_gemm(a, 1, 1, s))) return rc;
  return launch_reflect_fold(gp, gin, B, H, W, cin, s);
}
// weight gradient dW [9*cin][cout] (HWIO): x16 is the conv's fp16 input activation (before the folded upsample)
int launch_conv_wgrad(const half_t* x16, int upsample, const float* g, int ldg, int B, int H, int W, int cin, int cout,
                      float* col /* B*H*W*9*cin */, float* partial /* nsplit*9*cin*cout */, int nsplit, float* dw, hipStream_t s) {
  int rc;
  if ((rc = launch_im2col_act(x16, col, B, H, W, cin, upsample, s))) return rc;
  const int P = B * H * W;
  GemmArgs a = {};
  a.A = col; a.lda = 9 * cin; a.a_kmajor = 1;
  a.B = g; a.ldb = ldg; a.b_kmajor = 1;          // ldg >= cout (the 3-channel gradient comes padded to 4)
  a.M = 9 * cin; a.N = cout; a.K = P;
  a.ksplit = ((P + nsplit - 1) / nsplit + 15) / 16 * 16;
  a.out32 = partial; a.ldo = cout; a.out_split_stride = (size_t)9 * cin * cout;
  const int ns = (P + a.ksplit - 1) / a.ksplit;
  if ((rc = launch_gemm(a, ns, 1, s))) return rc;
  return launch_reduce_slabs(partial, (size_t)9 * cin * cout, ns, dw, s);
}
int conv_wgrad_splits(int B, int H, int W) {
  const long P = (long)B * H * W;
  long n = P / 2048;
  return (int)(n < 1 ? 1 : (n > 128 ? 128 : n));
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
