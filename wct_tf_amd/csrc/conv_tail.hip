// The tail of every decoder as ONE kernel (gfx950 / CDNA4): the last 64 -> 64 conv (reflect pad, ReLU; its input x2-upsampled in
// the decoders of relu2_1 .. relu5_1) and the 64 -> 3 output conv without activation (model.py:283-298, ops.py:12-19).
//
// Two launches -- conv3x3_mfma_kernel writing the 64-channel full-resolution map, conv_last_kernel reading it back -- moved
// 2 x 1.07 GB per 32-pair step and decoder through HBM for nothing (conv_last 1.2 ms per step at 4.6 TB/s, and the
// same bytes as store traffic inside the 64 -> 64 launches).  Here a block computes, for a 16 x 16 tile of the image, the
// 64-channel activations on the 18 x 18 halo patch (1.27 x the tile's own MFMA work), keeps them in LDS and applies the output
// conv to them: the map never exists in memory.
//
// Arithmetic = the two launches', operation for operation, so the frames are the same bits (WCT_FUSE_TAIL=0 is the test hook):
//   * mid[q] = fp16(ReLU(bias + sum)) with the sum in conv3x3_mfma_kernel's order (K-chunk of 32 channels, tap, k-step of 16;
//     v_mfma_f32_32x32x16_f16, bias added last, ReLU on the rounded value), evaluated AT THE REFLECTED POSITION of every halo
//     pixel -- what the output conv's reflect pad would have read from the stored map;
//   * out = conv_last_kernel's two stages: P[tap * 3 + co][q] = sum_c W[tap][c][co] mid[q][c] on the MFMA pipe (k-steps 0..3),
//     then bias + the nine shifted partials in tap order.
//
// One block = 256 threads = 4 waves = (output-channel tile ct, half of the 11 pixel tiles): a wave streams ONE channel tile's
// weights (36 KB per block and wave, as conv3x3_mfma_kernel<32,64,4,1> does per 128 pixels).  LDS: the x patch of a K-chunk
// (20 x 20 pixels x 64 B, single buffer -- the second chunk waits in registers) overlaid later by the mid patch (324 x 128 B,
// 16-byte pieces XOR-swizzled by the pixel), and the partials (324 x 28 floats): 77.8 KB, two blocks per CU.
#include "common.h"

namespace {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int XW = 20;                 // x patch: 20 x 20 pixels (tile + 2 on every side)
constexpr int MW = 18, MQ = MW * MW;   // mid patch: 18 x 18 = 324 pixels = 11 MFMA pixel tiles (the last one partial)
constexpr int NTILE = (MQ + 31) / 32;
constexpr int PP = 28;                 // partials pitch per patch pixel (27 used)
constexpr int XP_BYTES = XW * XW * 64, MID_BYTES = MQ * 128, PART_OFF = MID_BYTES, LDS_BYTES = PART_OFF + MQ * PP * 4;
static_assert(XP_BYTES <= MID_BYTES, "the mid patch overlays the x patch");

__device__ __forceinline__ int reflect_idx(int i, int n) {      // 1-px REFLECT padding (conv.hip); ragged tiles clamp
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  i = i < 0 ? 0 : i;
  return i >= n ? n - 1 : i;
}

__global__ __launch_bounds__(256, 2) void conv_tail_kernel(ConvTailArgs p, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave & 1, half = wave >> 1;                  // this wave's channel tile of the 64 -> 64 conv, and its pixel tiles T = half, half + 2, ..
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int y0 = ty * 16, x0 = tx * 16;
  const int Hin = p.upsample ? p.H / 2 : p.H, Win = p.upsample ? p.W / 2 : p.W;
  const half_t* xb = p.x + (size_t)b * Hin * Win * 64;
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7FFFFFFF, 0x00020000);

  // ---- x patch loader: window pixel (r, s) = image pixel (reflect(y0 - 2 + r), reflect(x0 - 2 + s)), 4 pieces of 8 channels per chunk
  constexpr int XITEMS = XW * XW * 4, XPER = (XITEMS + 255) / 256;
  unsigned xsrc[XPER], xdst[XPER];
#pragma unroll
  for (int i = 0; i < XPER; ++i) {
    const int item = tid + i * 256, ok = item < XITEMS;
    const int pix = ok ? item >> 2 : 0, piece = item & 3;
    const int r = pix / XW, sx = pix - r * XW;
    int iy = reflect_idx(y0 - 2 + r, p.H), ix = reflect_idx(x0 - 2 + sx, p.W);
    if (p.upsample) { iy >>= 1; ix >>= 1; }
    xsrc[i] = (unsigned)((iy * Win + ix) * 64 + piece * 8) * 2;
    xdst[i] = ok ? ((r * XW + sx) * 4 + (piece ^ ((sx >> 2) & 3))) * 16 : 0x7FFFFFF0u;     // (idle lanes: skipped at the store)
  }
  u32x4 xr[XPER];
  auto load_x = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < XPER; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xsrc[i], chunk * 64, 0);
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < XPER; ++i)
      if (tid + i * 256 < XITEMS) *reinterpret_cast<u32x4*>(smem + xdst[i]) = xr[i];
  };

  // ---- this wave's pixel tiles of the mid patch: q = 32 T + (lane & 31), read at the REFLECTED position of the halo pixel
  constexpr int MYT = (NTILE + 1) / 2;                        // 6 (half 0) / 5 (half 1)
  const int kgrp = lane >> 5;
  int roff[MYT][3][2];                                        // [tile][kx][ks]: byte offset of piece (ks, kgrp) of window pixel (row wy - 1, column wx - 1 + kx)
  int qs[MYT];
#pragma unroll
  for (int i = 0; i < MYT; ++i) {
    const int T = half + 2 * i;
    const int q = T * 32 + (lane & 31);
    qs[i] = q;
    const int qq = q < MQ ? q : MQ - 1;
    const int py = qq / MW, px = qq - py * MW;
    // window coordinates of the reflected position: image (ay, ax) -> window (ay - (y0 - 2), ax - (x0 - 2))
    const int wy = reflect_idx(y0 - 1 + py, p.H) - (y0 - 2), wx = reflect_idx(x0 - 1 + px, p.W) - (x0 - 2);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int sx = wx - 1 + kx;                            // the pixel's window column decides the swizzle
        roff[i][kx][ks] = ((wy - 1) * XW + sx) * 64 + (((ks * 2 + kgrp) ^ ((sx >> 2) & 3)) * 16);
      }
  }
  auto b_addr = [&](int i, int ky, int kx, int ks) { return roff[i][kx][ks] + ky * XW * 64; };   // (the row is an immediate)

  // weights of this wave's channel tile: fragment (ct, tap, k16) = 1 KiB at ((ct * 9 + tap) * 4 + k16) * 1024 bytes
  const unsigned wlane = (unsigned)(ct * 9 * 4 * 1024 + lane * 16);
  f32x16 acc[MYT];
#pragma unroll
  for (int i = 0; i < MYT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  load_x(0);
  half8 wf[2][2];                                             // [tap parity][ks]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wf[0][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane + ks * 1024, 0, 0));
  store_x();
  __syncthreads();
  load_x(1);                                                  // the second K-chunk waits in registers
#pragma unroll
  for (int chunk = 0; chunk < 2; ++chunk) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int t = chunk * 9 + tap, ky = tap / 3, kx = tap % 3;
      {                                                        // the next tap's weights (past the end: re-read, unused)
        const int nt = tap == 8 ? 0 : tap + 1, nc = tap == 8 ? (chunk == 0 ? 1 : chunk) : chunk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          wf[(t + 1) & 1][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane + ks * 1024, (nt * 4 + nc * 2) * 1024, 0));
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        half8 bq[MYT];
#pragma unroll
        for (int i = 0; i < MYT; ++i)
          if (i < MYT - 1 || half == 0) bq[i] = *reinterpret_cast<const half8*>(smem + b_addr(i, ky, kx, ks));
#pragma unroll
        for (int i = 0; i < MYT; ++i)
          if (i < MYT - 1 || half == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t & 1][ks], bq[i], acc[i], 0, 0, 0);
      }
    }
    if (chunk == 0) {
      __syncthreads();                                         // every wave done with the first chunk's patch
      store_x();
      __syncthreads();
    }
  }
  __syncthreads();                                             // the x patch is dead: the mid patch takes its place

  // ---- mid patch: bias last, one rounding to fp16, ReLU on the rounded pair (conv_epilogue_t).  acc register r of a tile holds
  // channel 32 ct + (r & 3) + 8 (r >> 2) + 4 kgrp of pixel q = 32 T + (lane & 31); mid[q] = 8 pieces of 16 B, piece j at j ^ (q & 7)
  {
    f32x4 bv[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) bv[rq] = *reinterpret_cast<const f32x4*>(p.bias + ct * 32 + 8 * rq + 4 * kgrp);
    const h2 zero2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
    for (int i = 0; i < MYT; ++i) {
      if (!(i < MYT - 1 || half == 0)) continue;
      const int q = qs[i];
      if (q < MQ) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[i][rq * 4 + j] + bv[rq][j];
          h2 lo = {(half_t)v[0], (half_t)v[1]}, hi = {(half_t)v[2], (half_t)v[3]};
          lo = __builtin_elementwise_max(lo, zero2);
          hi = __builtin_elementwise_max(hi, zero2);
          // channels c0 = 32 ct + 8 rq + 4 kgrp .. + 3: piece c0 >> 3 = 4 ct + rq, its half kgrp
          const int piece = 4 * ct + rq;
          *reinterpret_cast<u32x2*>(smem + q * 128 + ((piece ^ (q & 7)) * 16) + kgrp * 8) =
              u32x2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
        }
      }
    }
  }
  __syncthreads();

  // ---- output conv, stage 1 (conv_last_kernel): partials P[tap * 3 + co][q] over the 64 channels; tiles wave, wave + 4, wave + 8
  {
    half8 lw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) lw[ks] = *reinterpret_cast<const half8*>(p.wlast + (ks * 64 + lane) * 8);
    float* part = reinterpret_cast<float*>(smem + PART_OFF);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int T = wave + 4 * i;
      if (T >= NTILE) break;
      const int q = T * 32 + (lane & 31), qq = q < MQ ? q : MQ - 1;
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 bq = *reinterpret_cast<const half8*>(smem + qq * 128 + (((ks * 2 + kgrp) ^ (qq & 7)) * 16));
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(lw[ks], bq, a, 0, 0, 0);
      }
      if (q < MQ) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {                         // register r = row (r & 3) + 8 (r >> 2) + 4 kgrp
          const int row = (r & 3) + 8 * (r >> 2) + 4 * kgrp;
          if (row < 27) part[q * PP + row] = a[r];
        }
      }
    }
  }
  __syncthreads();
  // ---- stage 2: bias + the nine shifted partials, tap order
  {
    const float* part = reinterpret_cast<const float*>(smem + PART_OFF);
    const int ly = tid >> 4, lx = tid & 15;
    const int oy = y0 + ly, ox = x0 + lx;
    float a0 = p.blast[0], a1 = p.blast[1], a2 = p.blast[2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const float* pq = part + ((ly + ky) * MW + lx + kx) * PP + tap * 3;
      a0 += pq[0]; a1 += pq[1]; a2 += pq[2];
    }
    if (oy < p.H && ox < p.W) {
      float* o = p.y + (((size_t)b * p.H + oy) * p.W + ox) * 3;
      o[0] = a0; o[1] = a1; o[2] = a2;
    }
  }
}
}  // namespace

int launch_conv_tail(const ConvTailArgs& a, hipStream_t s) {
  ARG_CHECK(a.H > 1 && a.W > 1 && a.B > 0 && (!a.upsample || (a.H % 2 == 0 && a.W % 2 == 0)));
  ARG_CHECK((size_t)a.H * a.W * 64 * 2 < ((size_t)1 << 31));
  const int tiles_x = cdiv(a.W, 16), tiles_y = cdiv(a.H, 16);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipLaunchKernelGGL(conv_tail_kernel, dim3(tiles_x * tiles_y, a.B), dim3(256), LDS_BYTES, s, a, tiles_x);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
