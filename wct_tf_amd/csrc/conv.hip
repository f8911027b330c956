// Convolution stack of the stylize path, hand-written for gfx950 (CDNA4).
//
// Replaces what the reference delegates to cuDNN through Keras:
//   pad_reflect + Conv2D 3x3 'valid' + bias (+ReLU)  (ops.py:12-19, vgg_normalised.py:28-40)
//   UpSampling2D x2 nearest folded into the loader      (model.py:293)
//   MaxPooling2D(padding='same')                        (vgg_normalised.py:42)
//   the 1x1 'preprocess' conv folded into conv1_1       (vgg_normalised.py:25-26)
//   final 64->3 conv without activation                 (model.py:298)
//
// Layout: activations NHWC fp16, weights fp16 packed as MFMA A-fragments [Cout/32][tap][Cin/16][lane][8], fp32 accumulate
// on v_mfma_f32_32x32x16_f16.  The implicit GEMM is D[cout][pixel] =
// sum_{tap,cin} W[cout][tap,cin] * X[pixel+tap][cin]  (A = weights, B = pixels),
// so every lane ends up holding 4 consecutive output channels of one pixel and
// the epilogue stores 8 B (fp16) / 16 B (fp32) per lane.
#include "common.h"
#include <stdio.h>

// ---------------------------------------------------------------------------
// generic 3x3 conv, Cin % 32 == 0, Cout % BN == 0
// ---------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Phase timing build (-DCONV_TS, tools/conv_phase_timing.sh): lane 0 of every block stamps s_memtime at its phase
// boundaries; the launcher prints the per-launch means.  Compiled out of the product.
#ifdef CONV_TS
__device__ unsigned long long conv_ts[16384 * 10];
#define TS(slot) do { if (ts_on && threadIdx.x == 0) { const unsigned fl_ = blockIdx.x + gridDim.x * blockIdx.y; if (fl_ < 16384) conv_ts[fl_ * 10 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TS(slot) do {} while (0)
#endif
constexpr int TW = 16;          // tile width in pixels (one MFMA B-fragment = 2 rows x 16)
constexpr int PITCH = 20;       // patch row pitch in pixels (18 used; multiple of 4 keeps the swizzle aligned)
constexpr int BK = 32;          // input channels per K-chunk

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // 1-px REFLECT padding (edge not repeated): -1 -> 1, n -> n-2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  // ragged tiles can run further out; clamp (those lanes are masked at the store)
  i = i < 0 ? 0 : i;
  return i >= n ? n - 1 : i;
}

// epilogue shared by the conv kernels: bias, ReLU, (2x2 max-pool), store
// acc register r of a 32x32 tile holds channel (r&3) + 8*(r>>2) + 4*(lane>>5) of pixel lane&31 (= tile row
// (lane&31)>>4, column lane&15).  POOL: the 2x2/2 'same' max-pool that follows conv1_2 / conv2_2 / conv3_4 / conv4_4
// (vgg_normalised.py:42) is taken here: an MFMA pixel tile is 2 rows x 16 columns starting at even coordinates, so
// every pooling window lies inside one tile: max with lane^1 (column pair) and lane^16 (row pair); cells outside the
// image are 0, neutral after the ReLU (ceil-mode edge).  OUT32: the fp32 feature tap (and the fp16 copy if p.y16).
//
// Written for instruction count (round 2): the round-1 epilogue tested p.pool / p.y32 / p.y16 / `inside` inside its
// innermost loops and rebuilt 64-bit addresses per store -- ~2000 instructions and ~100 branches per wave, 10-13
// thousand cycles per tile measured with s_memtime (profiles/r02_conv_phase_timing.txt: 30 % of a 64-channel tile's
// residency).  Here the three variants are compile-time, the stores are buffer stores whose per-lane offset is
// pushed out of range for pixels outside the image (the hardware drops them: no exec-mask branches), offsets within
// a pixel are instruction immediates, ReLU runs on the packed fp16 pairs (rounding is monotonic and exact at 0, so
// max-after-round == round-after-max), and the bias comes from a copy the block parked in LDS at its start (a bias
// load from global memory between stores waits with vmcnt(0), i.e. for every store issued so far -- the round-1
// epilogue made 8 MT serial store round trips per tile -- and even hoisted above the stores its L2 latency sat on the
// critical path).  The bias is still added LAST, after the sum over taps and channels: starting the accumulators
// from it saved the 128 adds but moved fp32 roundings, and with them a few fp16 roundings of the activations.
template <int MT, int NT, bool POOL, bool OUT32>
__device__ __forceinline__ void conv_epilogue_t(const ConvArgs& p, f32x16 (&acc)[NT][MT], int b, int y0, int x0, int n0,
                                                int wm, int wn, int lane, const unsigned char* bias_lds) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  constexpr int OOB = (int)0x80000000u;      // beyond num_records of every per-image buffer (< 2^31 bytes, checked at launch)
  const int frag_px = lane & 15;
  const int frag_py = (lane & 31) >> 4;
  const int kgrp = lane >> 5;
  const int Ho = POOL ? (p.H + 1) / 2 : p.H, Wo = POOL ? (p.W + 1) / 2 : p.W;
  const unsigned img_elems = (unsigned)Ho * Wo * p.Cout;
  // one buffer per image and precision; a null output gets an empty buffer: all its stores are dropped
  const __amdgpu_buffer_rsrc_t r16 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.y16 ? p.y16 + (size_t)b * img_elems : nullptr), 0, p.y16 ? img_elems * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t r32 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(OUT32 && p.y32 ? p.y32 + (size_t)b * img_elems : nullptr), 0, OUT32 && p.y32 ? img_elems * 4 : 0, 0x00020000);
  f32x4 bvs[NT][4];                          // this lane's bias values, from the block's LDS copy
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
      bvs[nt][rq] = *reinterpret_cast<const f32x4*>(bias_lds + ((wn * NT + nt) * 32 + 8 * rq + 4 * kgrp) * 4);
  const float lo1 = p.relu ? 0.f : -__builtin_inff();
  const h2 lo2 = {(half_t)lo1, (half_t)lo1};
  const int ox = x0 + frag_px;
  const int chan = n0 + wn * NT * 32;        // first channel of this wave; + nt*32 (+ 8 rq / 16 m) are immediates
  // feature statistics (ConvArgs::usum / umax): per run of 16 pixels the channel sums, per image the largest value
  const bool stats = OUT32 && p.usum != nullptr;
  float* const us = stats ? p.usum + (size_t)b * (p.H * (p.W >> 4)) * p.Cout + chan + 4 * kgrp : nullptr;
  float vmax = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int oy = y0 + (wm * MT + mt) * 2 + frag_py;
    const bool in_img = (oy < p.H) & (ox < p.W);
    const bool inside = POOL ? (in_img & (frag_py == 0) & ((frag_px & 1) == 0)) : in_img;
    const int pix = POOL ? (oy >> 1) * Wo + (ox >> 1) : oy * p.W + ox;
    const int off16 = inside ? (pix * p.Cout + chan + 8 * kgrp) * 2 : OOB;
    const int off32 = inside ? (pix * p.Cout + chan + 4 * kgrp) * 4 : OOB;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      unsigned pk[4][2];                      // fp16 x4 of each register quad, packed
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][rq * 4 + j] + bvs[nt][rq][j];
        if (OUT32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], lo1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r32, off32 + (nt * 32 + 8 * rq) * 4, 0, 0);
          if (stats) {                         // uniform; W % 16 == 0: a tile row is a run of 16 pixels of the flattened map
            f32x4 t = v;
            unit_row_sum4(t);
            if (in_img) {
              vmax = fmaxf(fmaxf(vmax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
              if (frag_px == 0)
                *reinterpret_cast<f32x4*>(us + (size_t)(oy * (p.W >> 4) + (x0 >> 4)) * p.Cout + nt * 32 + 8 * rq) = t;
            }
          }
        }
        h2 lo = {(half_t)v[0], (half_t)v[1]}, hi = {(half_t)v[2], (half_t)v[3]};
        if (!OUT32) {
          lo = __builtin_elementwise_max(lo, lo2);
          hi = __builtin_elementwise_max(hi, lo2);
        }
        if (POOL) {
          // column pair by DPP quad_perm(1,0,3,2); row pair (lane ^ 16) by v_permlane16_swap of the value with itself: one
          // result holds the even row in both rows of lanes, the other the odd row -- their max is the pair's in every lane.
          // (was ds_swizzle xor 16: an LDS round trip per value, 64 of them in a row per wave)
          const h2 zero = {(half_t)0.f, (half_t)0.f};
          h2 q[2] = {in_img ? lo : zero, in_img ? hi : zero};
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            unsigned t = __builtin_bit_cast(unsigned, q[d]);
            const h2 o = __builtin_bit_cast(h2, __builtin_amdgcn_update_dpp(0, (int)t, 0xB1, 0xF, 0xF, true));
            q[d] = __builtin_elementwise_max(q[d], o);
            t = __builtin_bit_cast(unsigned, q[d]);
            const auto rows = __builtin_amdgcn_permlane16_swap(t, t, false, false);
            q[d] = __builtin_elementwise_max(__builtin_bit_cast(h2, (unsigned)rows[0]), __builtin_bit_cast(h2, (unsigned)rows[1]));
          }
          lo = q[0]; hi = q[1];
        }
        pk[rq][0] = __builtin_bit_cast(unsigned, lo);
        pk[rq][1] = __builtin_bit_cast(unsigned, hi);
      }
      // a lane holds channels 8rq+4kgrp..+3 of its pixel: the two half-waves own interleaved 8-byte pieces.  One
      // v_permlane32_swap per dword pairs quad 2m with quad 2m+1 so that the lower half stores channels 16m..16m+7
      // and the upper half 16m+8..16m+15: 16-byte stores, half as many.
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * m][0], pk[2 * m + 1][0], false, false);
        auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * m][1], pk[2 * m + 1][1], false, false);
        u32x4 o = {sx[0], sy[0], sx[1], sy[1]};
        __builtin_amdgcn_raw_buffer_store_b128(o, r16, off16 + (nt * 32 + 16 * m) * 2, 0, 0);
      }
    }
  }
  // values are >= 0 (ReLU): the bit patterns order like the values
  if (stats) umax_merge(p.umax + b * UMAX_SLOTS, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), vmax, lane);
}

template <int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[NT][MT], int b, int y0, int x0, int n0,
                                              int wm, int wn, int lane, const unsigned char* bias_lds) {
  if (p.pool) conv_epilogue_t<MT, NT, true, false>(p, acc, b, y0, x0, n0, wm, wn, lane, bias_lds);
  else if (p.y32) conv_epilogue_t<MT, NT, false, true>(p, acc, b, y0, x0, n0, wm, wn, lane, bias_lds);
  else conv_epilogue_t<MT, NT, false, false>(p, acc, b, y0, x0, n0, wm, wn, lane, bias_lds);
}

// ---------------------------------------------------------------------------
// Implicit GEMM per 256-thread block: a TH x 16 pixel tile x BN output channels, K = 9 taps x Cin in
// chunks of 32 input channels.
//  * pixels (MFMA B operand): the (TH+2) x 18 halo patch of a K-chunk is staged through LDS in 16-B
//    pieces with an XOR swizzle (conflict-free ds_read_b128); a wave reads one k-step of fragments
//    (8 MFMAs) ahead, with immediate offsets only (taps are compile-time in the 18-tap unrolled body);
//  * weights (A operand): pre-packed as MFMA fragments, every wave streams its own fragments straight
//    from global memory (L2-resident: all blocks read the same weights) into VGPRs with one fully
//    coalesced 1-KiB buffer_load_dwordx4 per fragment, a full tap (16 MFMAs) ahead.  No weight traffic
//    through LDS, no ds_write pass, no per-tap barrier: the waves of a block only meet when the patch
//    changes, every 9 taps.  (The first version staged weights through LDS with a barrier per tap:
//    771 TFLOP/s over the layer mix vs 943 for this one.)
//  * __launch_bounds__(256, 2): two blocks resident per CU cover each other's chunk boundaries.
// ---------------------------------------------------------------------------
// WLDS (round 5, tuning builds: the block DESIGN 8.2 of rounds 3-4 asked for): BOTH operands of the implicit GEMM come out of
// LDS.  The weights of a STAGE -- the three taps of one kernel row ky of the current K-chunk, BN x 3 x 32 fp16 = 48 KB for the
// 256-channel block, already in MFMA A-fragment order in global memory -- are copied global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB = one fragment per wave instruction, no registers, lane-linear on both sides), one whole stage
// ahead into the other of two buffers; a wave takes its fragments with ds_read_b128 at immediate offsets one tap ahead.  Every
// weight byte enters the CU once per 256 pixels x 256 channels (the register path streams it once per wave: 2 x per 256 x 128).
// Protocol: the barrier that opens the LAST tap of stage S (a) publishes stage S + 1 (issued a stage earlier; __syncthreads
// drains the DMA queue because LDS-DMA sits on vmcnt) before the first fragment of its first tap is prefetched, and (b) retires
// buffer S % 2 for every wave (its last fragment read was a tap earlier), so the DMA of stage S + 2 follows at once.
// STRIPS (round 5): the generic loader walks a strip of horizontally adjacent tiles as well, and the FIRST patch of the next tile
// is copied global -> LDS by LDS-DMA (no registers) under the current tile's epilogue.  A 64-channel layer has two K-chunks per
// tile: its first patch (7 k cycles of exposed latency, profiles/r02_conv_phase_timing.txt) and its epilogue were 45 % of a tile.
template <int TH, int BN, int WM, int WN, bool FUSE1 = false, bool WLDS = false, bool STRIPS = false>
__global__ __launch_bounds__(256, ((TH / 2) / WM) * ((BN / 32) / WN) > 8 ? 1 : 2) void conv3x3_mfma_kernel(ConvArgs p, int tiles_x, int tiles_y, int n_tiles, int strip) {
  constexpr bool LOOP = FUSE1 || STRIPS;           // the block walks `strip` tiles
  static_assert(!(FUSE1 && STRIPS) && !(WLDS && STRIPS), "one loader variant at a time");
  constexpr int PH = TH + 2;
  constexpr int MT = (TH / 2) / WM;
  constexpr int NT = (BN / 32) / WN;
  constexpr int PATCH_ITEMS = PH * 18 * 4;
  constexpr int PATCH_PER_THREAD = (PATCH_ITEMS + 255) / 256;
  constexpr int PATCH_BYTES = PH * PITCH * 64;
  // one wave per SIMD (the 4 x 4-tile block) has no partner block to cover a chunk boundary: its patch is double-buffered --
  // the next chunk's patch is parked in the other buffer under the last tap's MFMAs, and the boundary is ONE barrier
  constexpr bool DB = MT * NT > 8;
  constexpr int NBUF = DB ? 2 : 1;
  constexpr int DUMP_OFF = PATCH_BYTES;          // a buffer = patch + the dump slots of the idle loader lanes
  constexpr int BUF_STRIDE = PATCH_BYTES + 4096;
  constexpr int BIAS_OFF = NBUF * BUF_STRIDE;
  // FUSE1 (see the comment above make_patch below): the image patch behind the bias -- (TH + 4) x 20 pixels x 3 floats, row ry
  // = image row reflect(y0 - 2 + ry), then 4 zero floats (the target of the padded k = 27..31)
  // -- every image value as the packed pair {fp16 hi, fp16 lo} of conv_first_kernel's split; 16 zero words behind it (the reads
  // of the padded k = 27..31 may run that far); conv1_1's weight fragments (8 KB) and bias
  constexpr int IMG_OFF = BIAS_OFF + BN * 4, IMG_H = TH + 4, IMG_W = 20, IMG_ROW = IMG_W * 3, IMG_ZERO = IMG_OFF + IMG_H * IMG_ROW * 4;
  constexpr int W1_OFF = IMG_ZERO + 64, B1_OFF = W1_OFF + 8192;
  // -- and the landing area of the NEXT tile's image patch (global -> LDS directly, no registers: 9 dwords per thread, planar
  // [pixel slot i][channel][thread] because the hardware writes lane l of a wave at base + 4 l; the 12-byte form of the
  // instruction did not land the pixels at base + 12 l -- parity test red -- and was worth 2.5 % of this kernel)
  constexpr int STG_OFF = B1_OFF + 256;
  // WLDS: two weight-stage buffers behind the bias (1-KiB aligned): [cout tile BN/32][tap of the row 3][k-step 2][lane 64][16 B]
  constexpr int WST_BYTES = (BN / 32) * 3 * 2 * 1024;
  constexpr int WST_OFF = ((BIAS_OFF + BN * 4 + 1023) / 1024) * 1024;
  static_assert(!WLDS || !FUSE1, "WLDS: generic loader only");
  static_assert(!FUSE1 || (BN == 64 && !(((TH / 2) / WM) * ((BN / 32) / WN) > 8)), "FUSE1: 64 output channels, single patch buffer");
  // tap at which the next K-chunk's patch loads are issued (their registers are live from there to the
  // chunk boundary only); the tall tile has no registers to spare and loads at the boundary
  constexpr int PF_TAP = TH <= 16 ? 6 : 9;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const patch_lds = smem;

  [[maybe_unused]] bool ts_on = true;          // (-DCONV_TS) a strip block stamps its SECOND tile: entry = the end of the first
  TS(8);
  int bid = blockIdx.x;
  int ntile = bid % n_tiles;
  bid /= n_tiles;
  if (!FUSE1 && p.xcd_map) {
    // (tuning switch) blocks reach the 8 XCDs round-robin by their linear index; with the channel tile running fastest the
    // n_tiles blocks that read the SAME pixel patch sat on n_tiles different XCDs (every L2 fetched every patch).  Here the
    // 8 consecutive indices of a group work on 8 different pixel tiles and the channel tile advances with the group:
    // index = (pixel group * n_tiles + channel tile) * 8 + pixel in group -- the blocks of a pixel tile share an XCD.
    const int lin = (int)blockIdx.x, g8 = lin >> 3;
    ntile = g8 % n_tiles;
    bid = (g8 / n_tiles) * 8 + (lin & 7);
  }
  // FUSE1: a block walks a strip of `strip` horizontally adjacent tiles (the next tile's image patch is fetched under the
  // epilogue of the current one); blockIdx.x counts strips
  const int strips_x = LOOP ? (tiles_x + strip - 1) / strip : tiles_x;
  const int tx = LOOP ? (bid % strips_x) * strip : bid % strips_x;
  const int ty = bid / strips_x;
  const int b = blockIdx.y;
  int y0 = ty * TH;
  int x0 = tx * TW;
  int strip_i = 0;
  const int n0 = ntile * BN;
  int tid_ = threadIdx.x;

next_tile:   // (FUSE1: the strip loop -- the whole body; otherwise passed once)
  // (thread index and tile row are made opaque per iteration: hoisted out of the strip loop, everything that depends on the lane
  //  and y0 only -- fragment addresses, reflected rows, patch and store offsets -- stayed live across the tile's epilogue and
  //  27 registers were spilled; recomputing them costs a few dozen instructions per tile)
  if (LOOP) asm volatile("" : "+v"(tid_), "+s"(y0));
  ts_on = !FUSE1 || strip < 2 || strip_i == 1;
  if (FUSE1 && strip >= 2) TS(8);
  const int tid = tid_;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int Hin = p.upsample ? p.H / 2 : p.H;
  const int Win = p.upsample ? p.W / 2 : p.W;
  const half_t* xb = p.x + (size_t)b * Hin * Win * p.Cin;

  unsigned patch_src[PATCH_PER_THREAD];      // byte offsets from xb / smem (32-bit: SGPR base + VGPR offset addressing)
  unsigned patch_dst[PATCH_PER_THREAD];
#pragma unroll
  for (int i = 0; i < PATCH_PER_THREAD; ++i) {
    int item = tid + i * 256;
    int valid = item < PATCH_ITEMS;
    int pix = valid ? item >> 2 : 0;
    int chunk = item & 3;
    int py = pix / 18, px = pix - py * 18;
    int iy = reflect_idx(y0 - 1 + py, p.H);
    int ix = reflect_idx(x0 - 1 + px, p.W);
    if (p.upsample) { iy >>= 1; ix >>= 1; }
    patch_src[i] = valid ? ((iy * Win + ix) * p.Cin + chunk * 8) * 2 : 0;
    patch_dst[i] = valid ? ((py * PITCH + px) * 4 + (chunk ^ ((px >> 2) & 3))) * 16 : DUMP_OFF + tid * 16;
  }

  const int frag_px = lane & 15;
  const int frag_py = (lane & 31) >> 4;
  const int kgrp = lane >> 5;

  // patch read bases per (kx, k-step): pixel (row wm*MT*2 + frag_py, column frag_px + kx); the (mt, ky) row
  // offset is an immediate in the unrolled loop, so a fragment read is one ds_read_b128 with no address math
  int rd_base[3][2];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int px = frag_px + kx;
      rd_base[kx][ks] = ((wm * MT * 2 + frag_py) * PITCH + px) * 64 + (((ks * 2 + kgrp) ^ ((px >> 2) & 3)) * 16);
    }

  // per-lane weight fragment bases: fragment (cout tile, tap, cin16) = 64 lanes x 16 B, contiguous
  const int c16 = p.Cin >> 4;
  unsigned wfrag[NT];                        // byte offsets from p.w
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    wfrag[nt] = (((n0 >> 5) + wn * NT + nt) * 9 * c16 * 512 + lane * 8) * 2;
  // buffer addressing: SGPR resource + 32-bit lane offset (VGPR, loop-invariant) + SGPR offset (tap / chunk):
  // no per-load address arithmetic and no 64-bit address registers
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7FFFFFFF, 0x00020000);

  // STRIPS: chunk 0 of the patch of the tile at xt, global -> LDS.  One wave instruction fills 64 consecutive 16-byte slots of
  // the [PH][PITCH] x 4 slot image (lane-linear destination); the XOR swizzle of patch_dst is applied on the SOURCE side
  // (slot s of pixel px takes channel piece s ^ ((px >> 2) & 3)), the two padding pixels of a row re-read pixel 17, and the
  // lanes past the image (half an instruction) land in the dump area behind it.
  auto dma_patch0 = [&](int xt) {
    constexpr int NINS = (PH * PITCH * 4 + 63) / 64;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll 1                         // (rolled: unrolled, the eleven lane offsets live at once next to the accumulators spilled 62 registers)
    for (int i0 = 0; i0 < (NINS + 3) / 4; ++i0) {
      const int i = wave_u + i0 * 4;
      if (i < NINS) {                                          // (uniform)
        const int d = i * 64 + lane, ps = d >> 2, slot = d & 3;
        int py = ps / PITCH;
        const int px = ps - py * PITCH;
        py = py < PH ? py : PH - 1;
        int iy = reflect_idx(y0 - 1 + py, p.H), ix = reflect_idx(xt - 1 + (px < 18 ? px : 17), p.W);
        if (p.upsample) { iy >>= 1; ix >>= 1; }
        const int piece = slot ^ ((px >> 2) & 3);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, reinterpret_cast<__attribute__((address_space(3))) void*>(
            (__attribute__((address_space(3))) unsigned char*)smem + i * 1024), 16, ((iy * Win + ix) * p.Cin + piece * 8) * 2, 0, 0, 0);
      }
    }
  };
  // WLDS: this wave's share of a stage: fragments f = wave, wave + 4, ... of the (BN / 32) * 6; fragment f = (ct * 3 + tap3) * 2 + ks
  // = 1 KiB at p.w + ((n0 / 32 + ct) * 9 * c16 + (3 ky + tap3) * c16 + 2 chunk + ks) * 1024 bytes
  auto dma_stage = [&](int stage) {           // stage = chunk * 3 + ky (uniform)
    constexpr int NF = (BN / 32) * 6;
    const int chunk = stage / 3, ky = stage - chunk * 3, bufw = stage & 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);       // (M0 and the scalar offset must be SGPRs: no waterfall loops)
#pragma unroll
    for (int i = 0; i < NF / 4; ++i) {
      const int f = wave_u + i * 4, ks = f & 1, t3 = (f >> 1) % 3, ct = (f >> 1) / 3;
      // buffer form: SGPR resource + one loop-invariant lane offset + a scalar fragment offset -- no per-fragment VGPR address
      const int soff = ((((n0 >> 5) + ct) * 9 + ky * 3 + t3) * c16 + chunk * 2 + ks) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, reinterpret_cast<__attribute__((address_space(3))) void*>(
          (__attribute__((address_space(3))) unsigned char*)smem + WST_OFF + bufw * WST_BYTES + f * 1024), 16, lane * 16, soff, 0, 0);
    }
  };
  f32x16 acc[NT][MT];
  // the block's bias values -> LDS (read back by the epilogue; visible after the first patch barrier)
  if (tid < BN / 4 && strip_i == 0)
    *reinterpret_cast<f32x4*>(smem + BIAS_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(p.bias + n0 + tid * 4);

  const int n_chunks = p.Cin / BK;           // even: Cin is a multiple of 64 on this path
  u32x4 patch_regs[PATCH_PER_THREAD];
  half8 wf[2][NT][2];                        // ping-pong weight fragments, loaded a full tap ahead
  half8 wa[2][NT];                           // WLDS: ping-pong weight fragments by k-step, read out of LDS one k-step ahead
  auto read_wfrag = [&](half8 (&dst)[NT], int stage, int t3, int ks) {     // the fragments of k-step ks of tap (stage, t3)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      dst[nt] = *reinterpret_cast<const half8*>(smem + WST_OFF + (stage & 1) * WST_BYTES +
                                                ((((wn * NT + nt) * 3 + t3) * 2 + ks) * 64 + lane) * 16);
  };
  auto mma_group_w = [&](half8 (&w)[NT], half8 (&bq)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nt], bq[mt], acc[nt][mt], 0, 0, 0);
  };
  half8 bf[2][MT];                           // ping-pong pixel fragments, read one k-step (8 MFMAs) ahead

  auto read_group = [&](half8 (&dst)[MT], int ky, int kx, int ks, int buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      dst[mt] = *reinterpret_cast<const half8*>(patch_lds + rd_base[kx][ks] + (mt * 2 + ky) * PITCH * 64 + buf * BUF_STRIDE);
  };
  auto mma_group = [&](half8 (&w)[NT][2], half8 (&bq)[MT], int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nt][ks], bq[mt], acc[nt][mt], 0, 0, 0);
  };
  auto load_patch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < PATCH_PER_THREAD; ++i)
      patch_regs[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, patch_src[i], chunk * BK * 2, 0);
  };
  auto store_patch = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PATCH_PER_THREAD; ++i)
      *reinterpret_cast<u32x4*>(smem + patch_dst[i] + buf * BUF_STRIDE) = patch_regs[i];
  };

  // FUSE1: the K-chunk's patch is COMPUTED -- relu1_1 = ReLU(conv1_1(image)) at the patch's (reflected) positions, channels
  // 32 chunk .. + 31 -- with conv_first_kernel's arithmetic (split fp16 operands, k = ky * 9 + kx * 3 + c, the three MFMAs per
  // k-step in its order, bias added last, one rounding to fp16; the ReLU on the rounded pair, which is the same value): bit for
  // bit what conv_first_kernel would have stored and the loader read back.
  //  * pixel tiles (32 MFMA columns) without a division: PH / 2 "main" tiles = patch rows (2m, 2m + 1) x columns 0..15, and
  //    ceil(PH / 16) "edge" tiles = 16 rows x columns 16, 17; tile T = 4 t + wave.
  //  * the activation of patch pixel (py, px) is the one at (ay, ax) = (reflect(y0 - 1 + py), reflect(x0 - 1 + px)); its 27
  //    inputs are three runs of 9 words in the image patch: rows ay - y0 + 1 + ky, from column 3 (ax - x0 + 1).
  //  * a lane needs the 8 values k = 16 ks + 8 kgrp + j.  For kgrp = 0 these sit at the word offsets j | R+7, R+8, 2R .. 2R+5
  //    (R = IMG_ROW); for kgrp = 1 at the SAME offsets plus 8 (j = 0 of ks 0; j = 2 of ks 1) or plus R - 1 (the others) -- so
  //    two per-lane bases and immediate offsets, no address arithmetic per read.  (k = 27..31, kgrp 1: the weights are 0, the
  //    reads land on finite neighbours or on the zero words.)
  auto make_patch = [&](int chunk) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const unsigned char* w1 = smem + W1_OFF + chunk * 4096;
    half8 wh[2], wl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      wh[ks] = *reinterpret_cast<const half8*>(w1 + ((ks * 2 + 0) * 64 + lane) * 16);
      wl[ks] = *reinterpret_cast<const half8*>(w1 + ((ks * 2 + 1) * 64 + lane) * 16);
    }
    const int l = lane & 31, kg = lane >> 5;
    f32x4 b1[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) b1[rq] = *reinterpret_cast<const f32x4*>(smem + B1_OFF + (chunk * 32 + 8 * rq + 4 * kg) * 4);
    constexpr int NMAIN = PH / 2, NEDGE = (PH + 15) / 16, NT1 = (NMAIN + NEDGE + 3) / 4, R = IMG_ROW;
    const int pxm = l & 15, pxe = 16 + (l & 1);
    const int colm = IMG_OFF + (reflect_idx(x0 - 1 + pxm, p.W) - x0 + 1) * 12 - (y0 - 1) * R * 4;
    const int cole = IMG_OFF + (reflect_idx(x0 - 1 + pxe, p.W) - x0 + 1) * 12 - (y0 - 1) * R * 4;
    const int dstm = pxm * 64 + kg * 8, dste = pxe * 64 + kg * 8, swm = (pxm >> 2) & 3;     // (columns 16, 17: swizzle 0)
#pragma unroll 1                     // (unrolled twice the kernel spills 30 registers)
    for (int t = 0; t < NT1; ++t) {
      const int T = t * 4 + wave;                                  // wave-uniform
      const bool edge = T >= NMAIN;
      const int py = edge ? (T - NMAIN) * 16 + (l >> 1) : 2 * T + (l >> 4);
      const bool valid = py < PH && T < NMAIN + NEDGE;
      const int pyc = py < PH ? py : PH - 1;
      const int base = (edge ? cole : colm) + reflect_idx(y0 - 1 + pyc, p.H) * (R * 4);
      const unsigned char* b8 = smem + base + kg * 32;
      const unsigned char* bR = smem + base + kg * (R - 1) * 4;
      unsigned w[2][8];
      w[0][0] = *reinterpret_cast<const unsigned*>(b8);
#pragma unroll
      for (int j = 1; j < 8; ++j) w[0][j] = *reinterpret_cast<const unsigned*>(bR + j * 4);
      w[1][0] = *reinterpret_cast<const unsigned*>(bR + (R + 7) * 4);
      w[1][1] = *reinterpret_cast<const unsigned*>(bR + (R + 8) * 4);
#pragma unroll
      for (int j = 2; j < 8; ++j) w[1][j] = *reinterpret_cast<const unsigned*>(b8 + (2 * R + j - 2) * 4);
      f32x16 a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned uh[4], ul[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
          uh[i2] = __builtin_amdgcn_perm(w[ks][2 * i2 + 1], w[ks][2 * i2], 0x05040100u);   // the two hi halves
          ul[i2] = __builtin_amdgcn_perm(w[ks][2 * i2 + 1], w[ks][2 * i2], 0x07060302u);   // the two lo halves
        }
        const half8 bh = __builtin_bit_cast(half8, u32x4{uh[0], uh[1], uh[2], uh[3]});
        const half8 bl = __builtin_bit_cast(half8, u32x4{ul[0], ul[1], ul[2], ul[3]});
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], bh, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bl, a1, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], bh, a1, 0, 0, 0);
      }
      // the lane holds channels 8 rq + 4 kgrp .. + 3 of its pixel: half of the 16-byte piece rq
      const int dst = valid ? pyc * (PITCH * 64) + (edge ? dste : dstm) : DUMP_OFF + tid * 16;
      const int sw = valid && !edge ? swm : 0;
      const h2v zero2 = {(half_t)0.f, (half_t)0.f};
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        h2v h0 = {(half_t)(a1[rq * 4 + 0] + b1[rq][0]), (half_t)(a1[rq * 4 + 1] + b1[rq][1])};
        h2v h1 = {(half_t)(a1[rq * 4 + 2] + b1[rq][2]), (half_t)(a1[rq * 4 + 3] + b1[rq][3])};
        h0 = __builtin_elementwise_max(h0, zero2);
        h1 = __builtin_elementwise_max(h1, zero2);
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(smem + dst + (valid ? (rq ^ sw) * 16 : 0)) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
      }
    }
  };

  // FUSE1: image patch of the tile at xt -> landing area (asynchronous; vmcnt) / landing area -> image patch as split pairs
  constexpr int IMG_NP = IMG_H * IMG_W, IMG_PER = (IMG_NP + 255) / 256;
  auto fetch_img = [&](int xt) {
    const float* ib = p.img1 + (size_t)b * p.H * p.W * 3;
#pragma unroll
    for (int i = 0; i < IMG_PER; ++i) {
      const int e = tid + i * 256, ee = e < IMG_NP ? e : IMG_NP - 1;
      const int ry = ee / IMG_W, rx = ee - ry * IMG_W;
      const float* s = ib + ((size_t)reflect_idx(y0 - 2 + ry, p.H) * p.W + reflect_idx(xt - 2 + rx, p.W)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        __builtin_amdgcn_global_load_lds(s + c, reinterpret_cast<__attribute__((address_space(3))) void*>(
            (__attribute__((address_space(3))) unsigned char*)smem + STG_OFF + ((i * 3 + c) * 256 + wave * 64) * 4), 4, 0, 0);
    }
  };
  auto park_img = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's words have landed (nobody else reads them)
#pragma unroll
    for (int i = 0; i < IMG_PER; ++i) {
      const int e = tid + i * 256;
      if (e < IMG_NP) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float raw = *reinterpret_cast<const float*>(smem + STG_OFF + ((i * 3 + c) * 256 + tid) * 4);
          const float v = p.clamp01 ? fminf(fmaxf(raw, 0.f), 1.f) : raw;
          const half_t hi = (half_t)v, lo = (half_t)(v - (float)hi);
          *reinterpret_cast<unsigned*>(smem + IMG_OFF + (e * 3 + c) * 4) =
              (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
        }
      }
    }
  };
  if (FUSE1 && strip_i == 0) {
    fetch_img(x0);
    // conv1_1's weight fragments (8 KB = 256 x 32 B) and bias, once per block
    const u32x4 wa = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.w1frag) + tid * 32);
    const u32x4 wb = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.w1frag) + tid * 32 + 16);
    *reinterpret_cast<u32x4*>(smem + W1_OFF + tid * 32) = wa;
    *reinterpret_cast<u32x4*>(smem + W1_OFF + tid * 32 + 16) = wb;
    if (tid < 16) {
      *reinterpret_cast<f32x4*>(smem + B1_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(p.bias1 + tid * 4);
      *reinterpret_cast<unsigned*>(smem + IMG_ZERO + tid * 4) = 0u;
    }
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.f;
  TS(1);
  if (FUSE1) {
    // (the image patch in LDS was last read by the previous tile's second make_patch, which ends in a barrier)
    park_img();
    TS(9);
  } else if (STRIPS) {
    // the first patch of EVERY tile of a strip comes by LDS-DMA: the first tile's here, the others' under the previous epilogue
    // (one code path: a register-path first patch beside it left part of patch_regs in scratch memory)
    if (strip_i == 0) dma_patch0(x0);
  } else {
    load_patch(0);
  }
  if (WLDS) {
    dma_stage(0);
    if (n_chunks * 3 > 1) dma_stage(1);
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[0][nt][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wfrag[nt] + ks * 1024, 0, 0));
  }
  if (FUSE1) {
    __syncthreads();
    make_patch(0);
  } else if (STRIPS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the patch (and everything older) have landed
  } else {
    store_patch(0);
  }
  __syncthreads();                           // (WLDS: drains the DMA queue as well -- stages 0 and 1 are in LDS for every wave)
  TS(2);
  if (WLDS) read_wfrag(wa[0], 0, 0, 0);
  read_group(bf[0], 0, 0, 0, 0);

#pragma unroll 1
  for (int pair = 0; pair < n_chunks; pair += 2) {
#pragma unroll
    for (int t = 0; t < 18; ++t) {
      const int tap = t % 9, chunk_i = pair + t / 9;
      const int ky = tap / 3, kx = tap % 3;
      const int ntap = (tap + 1) % 9, nky = ntap / 3, nkx = ntap % 3;
      const bool more = chunk_i + 1 < n_chunks;
      const int buf = DB ? t / 9 : 0;            // compile-time: an immediate in the fragment reads
      if constexpr (WLDS) {
        // ---- both operands out of LDS.  wa[0] / bf[0] hold the fragments of this tap's first k-step (read during the previous
        // tap), the second k-step's go out now; under ITS MFMAs the first k-step of the next tap is read -- across a stage
        // boundary (kx = 2) behind the barrier that publishes the next stage's weights (and, at a chunk boundary, the next
        // patch, parked in the other buffer just before: ONE barrier per stage, none per chunk).
        const int stage = chunk_i * 3 + ky;
        const bool last = stage + 1 >= n_chunks * 3;           // (uniform)
        if (tap == PF_TAP && more) load_patch(chunk_i + 1);
        read_group(bf[1], ky, kx, 1, buf);
        read_wfrag(wa[1], stage, kx, 1);
        if (DB && tap == 8 && more) store_patch(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_group_w(wa[0], bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (kx == 2) {
          if (!last) {
            __syncthreads();                                   // stage + 1 has landed for everybody; nobody reads this stage's buffer again
            if (stage + 2 < n_chunks * 3) dma_stage(stage + 2);
            read_wfrag(wa[0], stage + 1, 0, 0);
            if (!DB && tap == 8) {                             // single patch buffer: it is free now (every wave's reads have returned)
              store_patch(0);
              __syncthreads();
            }
            read_group(bf[0], ky == 2 ? 0 : ky + 1, 0, 0, ky == 2 ? (DB ? buf ^ 1 : 0) : buf);
          }
        } else {
          read_wfrag(wa[0], stage, kx + 1, 0);
          read_group(bf[0], ky, kx + 1, 0, buf);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_group_w(wa[1], bf[1]);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      // 1) next tap's weight fragments (past the very end: re-read, unused); the next patch leaves PF_TAP early
      {
        const int nchunk = tap == 8 ? (more ? chunk_i + 1 : chunk_i) : chunk_i;
        const int wtap = (ntap * c16 + nchunk * 2) * 1024;   // uniform: the SGPR offset operand
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            wf[(t + 1) & 1][nt][ks] =
                __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wfrag[nt] + ks * 1024, wtap, 0));
      }
      if (!FUSE1 && tap == PF_TAP && more) load_patch(chunk_i + 1);
      // 2) second k-step's pixels go out, first k-step's MFMAs run on fragments read during the previous tap
      read_group(bf[1], ky, kx, 1, buf);
      if (DB && tap == 8 && more) store_patch(buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_group(wf[t & 1], bf[0], 0);
      __builtin_amdgcn_sched_barrier(0);
      // 3) next tap's first k-step goes out under the second k-step's MFMAs; at a chunk boundary the
      //    patch is replaced first (every wave done reading it)
      if (tap != 8) {
        read_group(bf[0], nky, nkx, 0, buf);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(wf[t & 1], bf[1], 1);
      } else {
        mma_group(wf[t & 1], bf[1], 1);
        if (more) {
          __syncthreads();
          if (FUSE1) {
            TS(3);
            make_patch(chunk_i + 1);
            __syncthreads();
            TS(4);
          } else if (!DB) {
            if (PF_TAP > 8) load_patch(chunk_i + 1);
            store_patch(0);
            __syncthreads();
          }
        }
        read_group(bf[0], 0, 0, 0, DB ? buf ^ 1 : 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  TS(5);
  const bool more_tiles = LOOP && strip_i + 1 < strip && tx + strip_i + 1 < tiles_x;      // uniform
  if (FUSE1 && more_tiles) fetch_img(x0 + TW);        // arrives under the epilogue (whose stores sit on the same in-order counter)
  if (STRIPS && more_tiles) {
    __syncthreads();                         // every wave has taken its last fragments out of the patch
    dma_patch0(x0 + TW);
  }
  TS(0);
  conv_epilogue<MT, NT>(p, acc, b, y0, x0, n0, wm, wn, lane, smem + BIAS_OFF);
  TS(6);
  if (more_tiles) {
    TS(7);
    ++strip_i;
    x0 += TW;
    goto next_tile;
  }
#ifdef CONV_TS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TS(7);
#endif
}

template <int TH, int BN, int WM, int WN, bool FUSE1 = false, bool WLDS = false, bool STRIPS = false>
static int launch_conv_cfg(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH);
  const int n_tiles = a.Cout / BN;
  constexpr int PH = TH + 2;
  constexpr int NBUF = ((TH / 2) / WM) * ((BN / 32) / WN) > 8 ? 2 : 1;
  size_t lds = (size_t)NBUF * (PH * PITCH * 64 + 4096) + BN * sizeof(float)      // NBUF x (patch, dump slots), bias
               + (FUSE1 ? (size_t)(TH + 4) * 20 * 3 * 4 + 64 + 8192 + 256 + 9 * 1024 : 0);  // image patch, zero words, conv1_1 weights and bias, landing area
  if (WLDS) lds = ((lds + 1023) / 1024) * 1024 + 2 * (size_t)(BN / 32) * 3 * 2 * 1024;      // two weight-stage buffers
  // FUSE1: strips of 4 / 2 tiles per block while that leaves the chip >= 4 blocks per resident slot (512 slots)
  const long tiles = (long)tiles_x * tiles_y * a.B;
  static const int strip_force = tune_int("WCT_FUSE1_STRIP", 0);   // tuning switch
  const int strip = !(FUSE1 || STRIPS) ? 1 : strip_force ? strip_force : tiles >= 8192 ? 4 : tiles >= 4096 ? 2 : 1;
  dim3 grid(((FUSE1 || STRIPS) ? cdiv(tiles_x, strip) : tiles_x) * tiles_y * n_tiles, a.B);
  ConvArgs ax = a;
  static const int xcd = tune_int("WCT_CONV_XCD", 0);     // tuning switch: XCD-aware tile order (needs a multiple of 8 pixel tiles per image)
  ax.xcd_map = !FUSE1 && !STRIPS && xcd && (tiles_x * tiles_y) % 8 == 0;
  if (lds > 64 * 1024) {
    // more dynamic LDS than the 64 KiB a runtime may enforce by default (gfx950 has 160 KiB per CU): say so, once per device
    // and instantiation; a refusal surfaces here and not at a later synchronisation
    static bool raised[16] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 16 && !raised[dev]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_mfma_kernel<TH, BN, WM, WN, FUSE1, WLDS, STRIPS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      raised[dev] = true;
    }
  }
  hipLaunchKernelGGL((conv3x3_mfma_kernel<TH, BN, WM, WN, FUSE1, WLDS, STRIPS>), grid, dim3(256), lds, s, ax, tiles_x, tiles_y, n_tiles, strip);
#ifdef CONV_TS
  if (a.B >= 8) {
    static int nlaunch = 0;
    ++nlaunch;
    (void)hipStreamSynchronize(s);
    static unsigned long long host[16384 * 10];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(conv_ts), sizeof(host));
    const int nb = (int)(grid.x * grid.y < 16384 ? grid.x * grid.y : 16384);
    double sum[6] = {0}, mid = 0;   // slots: 8 entry, 1 addresses ready, 2 first patch in LDS, (FUSE1: 3-4 the second chunk's patch,) 5 taps done, 6 epilogue issued, 7 stores drained
    for (int i = 0; i < nb; ++i) {
      const unsigned long long* h = host + (size_t)i * 10;
      sum[0] += (double)(h[1] - h[8]); sum[1] += (double)(h[2] - h[1]); sum[2] += (double)(h[5] - h[2]);
      sum[3] += (double)(h[6] - h[5]); sum[4] += (double)(h[7] - h[6]); sum[5] += (double)(h[7] - h[8]);
      if (FUSE1) mid += (double)(h[4] - h[3]);
    }
    if (FUSE1) {
      double park = 0, fetch = 0;
      for (int i = 0; i < nb; ++i) { const unsigned long long* h = host + (size_t)i * 10; park += (double)(h[9] - h[1]); fetch += (double)(h[0] - h[5]); }
      fprintf(stderr, "TS (conv1_1 in the loader, strip %d: the second tile) second patch %.0f of the mainloop cycles; image patch parked %.0f of the load0 cycles; next fetch issued %.0f of the epilogue cycles\n",
              strip, mid / nb, park / nb, fetch / nb);
    }
    fprintf(stderr, "TS launch %d <%d,%d,%d,%d> Cin %d Cout %d H %d up %d pool %d y32 %d blocks %d: setup %.0f load0 %.0f mainloop %.0f epilogue %.0f drain %.0f total %.0f\n",
            nlaunch, TH, BN, WM, WN, a.Cin, a.Cout, a.H, a.upsample, a.pool, a.y32 != nullptr, nb,
            sum[0] / nb, sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, sum[5] / nb);
  }
#endif
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

int launch_conv3x3(const ConvArgs& a, hipStream_t s) {
  ARG_CHECK(a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.H > 1 && a.W > 1 && a.B > 0);
  ARG_CHECK(!a.upsample || (a.H % 2 == 0 && a.W % 2 == 0));
  // one image's activations are addressed with 32-bit byte offsets (buffer loads)
  ARG_CHECK((size_t)a.H * a.W * a.Cin * 2 < ((size_t)1 << 31));
  ARG_CHECK((size_t)a.H * a.W * a.Cout * (a.y32 ? 4 : 2) < ((size_t)1 << 31));     // per-image output buffers (epilogue)
  ARG_CHECK(!a.usum || (a.y32 && a.umax && a.relu && a.W % 16 == 0));
  ARG_CHECK(!a.pool || (a.relu && !a.y32));            // the fused pool relies on post-ReLU values (>= 0) at ragged edges
  // the reduced-FLOP kernel, where the caller packed its filters for it (api.hip: the wide layers) and the epilogue is one it has
  if (conv3x3_wino_takes(a)) return launch_conv3x3_wino(a, s);
  // pick the largest tile that still gives the chip >= ~2 blocks per CU (two are resident per CU)
  if (a.img1) {                                           // conv1_1 inside the patch loader: the 512-pixel x 64-channel block only
    ARG_CHECK(a.Cin == 64 && a.Cout == 64 && !a.upsample && a.w1frag && a.bias1);
    return launch_conv_cfg<32, 64, 4, 1, true>(a, s);
  }
  const long px16 = (long)cdiv(a.W, TW) * cdiv(a.H, 16) * a.B;
  const long px32 = (long)cdiv(a.W, TW) * cdiv(a.H, 32) * a.B;
  static const int force = tune_int("WCT_CONV_CFG", 0);   // tuning switch
  if (force == 1 && a.Cout % 128 == 0) return launch_conv_cfg<16, 128, 2, 2>(a, s);
  if (force == 2) return launch_conv_cfg<32, 64, 4, 1>(a, s);
  if (force == 3) return launch_conv_cfg<16, 64, 4, 1>(a, s);
  if (force == 4) return launch_conv_cfg<8, 64, 2, 2>(a, s);
  // (round 3: a 256-pixel x 256-channel block, 4 x 4 MFMA tiles per wave with the accumulators in AGPRs, ONE wave per SIMD --
  //  half the LDS fragment reads per MFMA.  As it stands it ties the 128-channel block: 512->512 @64 1156 vs 1136 TFLOP/s
  //  (1175 vs 1143 with the double-buffered patch, DB), 256->256 @128 1057-1068 vs 1066-1085, worse where the grid gets
  //  small (512->512 @32: 381 vs 590); without a partner block the first patch and the epilogue of every tile are exposed.
  //  Kept as a switch: the starting point of DESIGN 8.2)
#ifdef WCT_TUNING      // (2 VGPR spills: not instantiated in the product build -- VERDICT r4)
  if (force == 5 && a.Cout % 256 == 0) return launch_conv_cfg<16, 256, 2, 2>(a, s);
  // (round 5: the same block with BOTH operands through LDS -- weights by LDS-DMA a stage ahead; see the kernel's WLDS note)
  if (force == 6 && a.Cout % 256 == 0) return launch_conv_cfg<16, 256, 2, 2, false, true>(a, s);
  if (force == 7 && a.Cout % 256 == 0 && px16 * (a.Cout / 256) >= 256) return launch_conv_cfg<16, 256, 2, 2, false, true>(a, s);
  // ... and on the shipped 256-pixel x 128-channel block: 76 KB of LDS, still two blocks per CU
  if (force == 8 && a.Cout % 128 == 0 && px16 * (a.Cout / 128) >= 512) return launch_conv_cfg<16, 128, 2, 2, false, true>(a, s);
#endif
  // (round 2: <32,64,2,2> and <16,128,1,4> -- the waves of a block split the output channels instead of the pixels,
  //  halving / removing the redundant weight streams -- measured 593 vs 601 TFLOP/s on 64->64 @512^2 and 5-8 % slower on
  //  the wide layers, profiles/r02_conv_cfg_sweep.txt: the weight streams are not what bounds these layers; removed)
#ifdef WCT_TUNING
  // (round 5, tuning builds: WCT_CONV_STRIPS=1 -- strips with the next tile's first patch by LDS-DMA under the epilogue, on the layers
  //  with few K-chunks per tile and many tiles.  Bit-identical, 10 % SLOWER on 64 -> 64 @512 (0.194 -> 0.215 ms at batch 8,
  //  profiles/r05_conv_strips.txt): the strip serialises a tile's store drain with the next tile's start, which two independent
  //  blocks per CU overlap for free)
  static const int strips_on = tune_int("WCT_CONV_STRIPS", 0);
  if (strips_on && a.Cin <= 128) {
    if (a.Cout % 128 == 0 && px16 * (a.Cout / 128) >= 512 && px16 >= 4096) return launch_conv_cfg<16, 128, 2, 2, false, false, true>(a, s);
    if (!(a.Cout % 128 == 0 && px16 * (a.Cout / 128) >= 512) && px32 * (a.Cout / 64) >= 512 && px32 >= 4096)
      return launch_conv_cfg<32, 64, 4, 1, false, false, true>(a, s);
  }
#endif
  if (a.Cout % 128 == 0 && px16 * (a.Cout / 128) >= 512) return launch_conv_cfg<16, 128, 2, 2>(a, s);
  if (px32 * (a.Cout / 64) >= 512) return launch_conv_cfg<32, 64, 4, 1>(a, s);      // 512 px x 64 ch
  if (a.Cout % 128 == 0 && px16 * (a.Cout / 128) >= 256) return launch_conv_cfg<16, 128, 2, 2>(a, s);
  if (px16 * (a.Cout / 64) >= 256) return launch_conv_cfg<16, 64, 4, 1>(a, s);
  return launch_conv_cfg<8, 64, 2, 2>(a, s);
}

// ---------------------------------------------------------------------------
// conv1_1: 3 -> 64 with the folded preprocess, fp32 image in.  The layer is all output: 64 channels
// written per 3 read.  K = 27 (padded to 32) runs on the MFMA pipe at fp32-product accuracy: image
// values and weights are split into fp16 hi + lo pairs (22 significand bits), hi*hi + hi*lo + lo*hi
// accumulate in fp32 (three v_mfma_f32_32x32x16_f16 per k-step; the VALU version spent 1728 FMAs per
// thread and was VALU-bound at 1.4 TB/s).  The MFMA layout leaves a lane with 4 channels of one
// pixel, which would store 32 B per pixel and instruction; the results therefore go through a
// per-wave LDS transpose and leave as whole pixel rows (256 B fp32 / 128 B fp16): every store
// instruction writes 1 KiB of contiguous memory.
// k = ky*9 + kx*3 + c: a patch row [px][3] holds the 9 values of a ky contiguously.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_kernel(ConvFirstArgs p, int strips_x) {
  // One block walks a strip of CF_TILES horizontally adjacent 16x16 pixel tiles (x 64 channels); wave w owns pixel
  // rows 4w..4w+3 of a tile (two 2x16 MFMA pixel tiles).  The next tile's 18x18x3 halo patch is fetched into
  // registers while the current one is computed, and the per-block set-up (weight fragments, bias, fragment
  // addresses) is paid once per strip.
  constexpr int CF_TILES = 4, PATCH_N = 18 * 18 * 3, PER = (PATCH_N + 255) / 256;
  __shared__ float patch[PATCH_N + 4];         // [972] stays 0: the target of the padded k = 27..31
  __shared__ __attribute__((aligned(16))) unsigned char tr[4][32 * 256];   // per wave: 32 pixels x 64 ch fp32
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sx = blockIdx.x % strips_x, ty = blockIdx.x / strips_x;
  const int b = blockIdx.y;
  const int y0 = ty * 16;
  const float* xb = p.x + (size_t)b * p.H * p.W * 3;
  // this thread's patch elements: (row offset, column in the patch, channel) never change along the strip
  int e_row[PER], e_px[PER], e_c[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + i * 256;
    const int pix = e < PATCH_N ? e / 3 : 0;
    e_c[i] = e < PATCH_N ? e - pix * 3 : 0;
    const int py = pix / 18;
    e_px[i] = pix - py * 18;
    e_row[i] = reflect_idx(y0 - 1 + py, p.H) * p.W;
  }
  float pre[PER];
  auto fetch = [&](int x0) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      float v = xb[((size_t)e_row[i] + reflect_idx(x0 - 1 + e_px[i], p.W)) * 3 + e_c[i]];
      pre[i] = p.clamp01 ? fminf(fmaxf(v, 0.f), 1.f) : v;
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i) if (tid + i * 256 < PATCH_N) patch[tid + i * 256] = pre[i];
  };
  if (tid < 4) patch[PATCH_N + tid] = 0.f;
  fetch(sx * CF_TILES * 16);
  // weight fragments [cout tile][k-step][hi/lo][lane][8]: 8 coalesced 1-KiB loads, register-resident
  half8 wh[2][2], wl[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      wh[t][ks] = *reinterpret_cast<const half8*>(p.wfrag + (((t * 2 + ks) * 2 + 0) * 64 + lane) * 8);
      wl[t][ks] = *reinterpret_cast<const half8*>(p.wfrag + (((t * 2 + ks) * 2 + 1) * 64 + lane) * 8);
    }
  const int frag_px = lane & 15, frag_py = (lane & 31) >> 4, kgrp = lane >> 5;
  // byte address of patch element k of this lane's pixel in tile row pair 0 of the wave (mt adds an immediate)
  int addr[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = ks * 16 + kgrp * 8 + j;
      const int ky = k / 9, kp = k - ky * 9;
      addr[ks][j] = k < 27 ? (((wave * 4 + frag_py + ky) * 18 + frag_px) * 3 + kp) * 4 : 972 * 4;
    }
  f32x4 bias[2][4];                            // bias of this lane's channels (nt, rq)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) bias[nt][rq] = *reinterpret_cast<const f32x4*>(p.bias + nt * 32 + 8 * rq + 4 * kgrp);
  park();
  __syncthreads();
  const unsigned char* pb = reinterpret_cast<const unsigned char*>(patch);
  unsigned char* const wt = tr[wave];
  float vmax = 0.f;
  for (int tile = 0; tile < CF_TILES; ++tile) {
  const int x0 = (sx * CF_TILES + tile) * 16;
  if (x0 >= p.W) break;                        // uniform: the strip runs past the image
  const bool next = tile + 1 < CF_TILES && x0 + 16 < p.W;
  if (next) fetch(x0 + 16);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 bh, bl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // the padded-k slot must not move with mt: its address is absolute
        const int off = (ks * 16 + kgrp * 8 + j < 27) ? mt * 2 * 54 * 4 : 0;
        const float v = *reinterpret_cast<const float*>(pb + addr[ks][j] + off);
        bh[j] = (half_t)v;
        bl[j] = (half_t)(v - (float)bh[j]);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[nt][ks], bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt][ks], bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[nt][ks], bh, acc[nt], 0, 0, 0);
      }
    }
    // transpose through LDS: pixel row = 64 fp32 = 16 pieces of 16 B, piece index XOR (pixel & 15)
    const int px = lane & 31;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[nt][rq * 4 + j] + bias[nt][rq][j], 0.f);
        const int piece = nt * 8 + 2 * rq + kgrp;
        *reinterpret_cast<f32x4*>(wt + (px * 16 + (piece ^ (px & 15))) * 16) = v;
        if (p.usum) {                          // uniform; statistics of the fp32 tap as in conv_epilogue_t
          f32x4 t = v;
          unit_row_sum4(t);
          const int sy = y0 + wave * 4 + mt * 2 + frag_py;
          if (sy < p.H) {
            vmax = fmaxf(fmaxf(vmax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            if (frag_px == 0)
              *reinterpret_cast<f32x4*>(p.usum + ((size_t)b * (p.H * (p.W >> 4)) + sy * (p.W >> 4) + (x0 >> 4)) * 64 + piece * 4) = t;
          }
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int oy0 = y0 + wave * 4 + mt * 2;
    if (p.y32) {
      const int piece = lane & 15;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it * 4 + (lane >> 4);
        const int oy = oy0 + (q >> 4), ox = x0 + (q & 15);
        const f32x4 v = *reinterpret_cast<const f32x4*>(wt + (q * 16 + (piece ^ (q & 15))) * 16);
        if (oy < p.H && ox < p.W) *reinterpret_cast<f32x4*>(p.y32 + (((size_t)b * p.H + oy) * p.W + ox) * 64 + piece * 4) = v;
      }
    }
    if (p.y16) {
      const int c8 = lane & 7;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 8 + (lane >> 3);
        const int oy = oy0 + (q >> 4), ox = x0 + (q & 15);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wt + (q * 16 + ((2 * c8) ^ (q & 15))) * 16);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wt + (q * 16 + ((2 * c8 + 1) ^ (q & 15))) * 16);
        half8 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = (half_t)v0[j]; h[4 + j] = (half_t)v1[j]; }
        if (oy < p.H && ox < p.W) *reinterpret_cast<half8*>(p.y16 + (((size_t)b * p.H + oy) * p.W + ox) * 64 + c8 * 8) = h;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();           // the region is rewritten by the next pixel tile
  }
  if (next) {                                  // swap in the next tile's patch once every wave has read this one
    __syncthreads();
    park();
    __syncthreads();
  }
  }
  if (p.usum) umax_merge(p.umax + b * UMAX_SLOTS, (int)blockIdx.x * 4 + wave, vmax, lane);
}

int launch_conv_first(const ConvFirstArgs& a, hipStream_t s) {
  ARG_CHECK(a.H > 1 && a.W > 1 && a.B > 0);
  ARG_CHECK(!a.usum || (a.y32 && a.umax && a.W % 16 == 0));
  const int strips_x = cdiv(a.W, 64), tiles_y = cdiv(a.H, 16);
  hipLaunchKernelGGL(conv_first_kernel, dim3(strips_x * tiles_y, a.B), dim3(256), 0, s, a, strips_x);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// decoder output conv: 64 -> 3, no activation, fp32 out.  Three output channels would waste the MFMA
// tile, so the taps are moved to the output side: stage 1 computes, for every pixel q of the 18x18
// halo patch, the 27 partial products  P[tap*3+co][q] = sum_c W[tap][c][co] x[q][c]  -- a 32 x 64 x 324
// GEMM on v_mfma_f32_32x32x16_f16 (27 of 32 rows used) whose pixel fragments come straight from global
// memory (each activation is read once, no LDS patch); stage 2 adds the 9 shifted partials per output
// pixel out of LDS (27 floats per pixel instead of 72 16-B patch reads and 864 dot products).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_last_kernel(ConvLastArgs p, int tiles_x) {
  constexpr int PP = 33;                                   // partials pitch per patch pixel (odd: conflict-free)
  __shared__ float part[18 * 18 * PP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int b = blockIdx.y;
  const int y0 = ty * 16, x0 = tx * 16;
  const half_t* xb = p.x + (size_t)b * p.H * p.W * 64;
  half8 wf[4];                                             // A fragments: rows tap*3+co, k = 16 ks + 8 (lane>>5) ..
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const half8*>(p.wfrag + (ks * 64 + lane) * 8);
  // 324 patch pixels = 11 tiles of 32 (the last one partial), tiles wave, wave + 4, wave + 8 per wave.  Round 5: ALL of a wave's
  // activation loads (up to 3 tiles x 4 x 16 B per lane) are issued before the first MFMA -- the kernel is a read stream (3.7 TB/s
  // with one tile's loads in flight per wave: a memory latency per tile) with three blocks per CU to hide it behind.
  half8 bf[3][4];
  int qs[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tile = wave + 4 * i;                         // (wave-uniform; tile 11 does not exist: wave 3 has two tiles)
    const int q = tile * 32 + (lane & 31);
    qs[i] = q;
    const int qq = q < 324 ? q : 323;
    const int py = qq / 18, px = qq - py * 18;
    const int iy = reflect_idx(y0 - 1 + py, p.H), ix = reflect_idx(x0 - 1 + px, p.W);
    const half_t* src = xb + ((size_t)iy * p.W + ix) * 64 + (lane >> 5) * 8;
    if (tile < 11) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bf[i][ks] = *reinterpret_cast<const half8*>(src + ks * 16);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (wave + 4 * i >= 11) break;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], bf[i][ks], acc, 0, 0, 0);
    if (qs[i] < 324) {
#pragma unroll
      for (int r = 0; r < 16; ++r)                         // register r = row (r&3) + 8 (r>>2) + 4 (lane>>5)
        part[qs[i] * PP + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[r];
    }
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int oy = y0 + ly, ox = x0 + lx;
  float a0 = p.bias[0], a1 = p.bias[1], a2 = p.bias[2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const float* pq = part + ((ly + ky) * 18 + lx + kx) * PP + tap * 3;
    a0 += pq[0]; a1 += pq[1]; a2 += pq[2];
  }
  if (oy < p.H && ox < p.W) {
    float* o = p.y + (((size_t)b * p.H + oy) * p.W + ox) * 3;
    o[0] = a0; o[1] = a1; o[2] = a2;
  }
}

int launch_conv_last(const ConvLastArgs& a, hipStream_t s) {
  ARG_CHECK(a.H > 1 && a.W > 1 && a.B > 0);
  const int tiles_x = cdiv(a.W, 16), tiles_y = cdiv(a.H, 16);
  hipLaunchKernelGGL(conv_last_kernel, dim3(tiles_x * tiles_y, a.B), dim3(256), 0, s, a, tiles_x);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// 2x2/2 max-pool, 'same' (= ceil mode): the odd last row/col pools what exists
// ---------------------------------------------------------------------------
__global__ void maxpool_kernel(const half_t* x, half_t* y, int B, int H, int W, int C, int Ho, int Wo) {
  const int c8 = C / 8;
  size_t total = (size_t)B * Ho * Wo * c8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c8);
    size_t t = i / c8;
    int ox = (int)(t % Wo); t /= Wo;
    int oy = (int)(t % Ho);
    int b = (int)(t / Ho);
    const half_t* xb = x + (size_t)b * H * W * C;
    int iy = oy * 2, ix = ox * 2;
    half8 m = *reinterpret_cast<const half8*>(xb + ((size_t)iy * W + ix) * C + c * 8);
    if (ix + 1 < W) m = __builtin_elementwise_max(m, *reinterpret_cast<const half8*>(xb + ((size_t)iy * W + ix + 1) * C + c * 8));
    if (iy + 1 < H) {
      m = __builtin_elementwise_max(m, *reinterpret_cast<const half8*>(xb + ((size_t)(iy + 1) * W + ix) * C + c * 8));
      if (ix + 1 < W) m = __builtin_elementwise_max(m, *reinterpret_cast<const half8*>(xb + ((size_t)(iy + 1) * W + ix + 1) * C + c * 8));
    }
    *reinterpret_cast<half8*>(y + i * 8) = m;
  }
}

int launch_maxpool2x2(const half_t* x, half_t* y, int B, int H, int W, int C, hipStream_t s) {
  ARG_CHECK(C % 8 == 0);
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  size_t total = (size_t)B * Ho * Wo * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, x, y, B, H, W, C, Ho, Wo);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// dtype / range conversions at the path boundary (wct.py:60-68)
// ---------------------------------------------------------------------------
__global__ void u8_to_f32_kernel(const uint8_t* x, float* y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = (float)((double)x[i] / 255.0);    // image / 255. in float64, then the fp32 feed cast
}
__global__ void f32_to_u8_kernel(const float* x, uint8_t* y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = fminf(fmaxf(x[i], 0.f), 1.f) * 255.f;
    y[i] = (uint8_t)v;                       // np.uint8() truncates
  }
}
__global__ void f32_to_f16_kernel(const float* x, half_t* y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (half_t)x[i];
}
__global__ void f16_to_f32_kernel(const half_t* x, float* y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}

static inline int ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b ? b : 1));
}
int launch_u8_to_f32(const uint8_t* x, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_f32_to_u8(const float* x, uint8_t* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_u8_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_f32_to_f16(const float* x, half_t* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
int launch_f16_to_f32(const half_t* x, float* y, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, n);
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
