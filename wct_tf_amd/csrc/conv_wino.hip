// Reduced-FLOP 3x3 convolution for the wide layers of the stylize path (gfx950 / CDNA4): Winograd F(2,3) along y, direct along x.
//
// Replaces, like conv.hip, what the reference hands to cuDNN through Keras -- pad_reflect + Conv2D 3x3 'valid' + bias + ReLU
// (ops.py:12-19, vgg_normalised.py:28-40, model.py:291), with the x2 nearest upsample in the loader (model.py:293) and the
// 'same' 2x2 max-pool in the epilogue (vgg_normalised.py:42) -- and cuDNN's algorithm selection for 3x3 / stride 1 includes
// Winograd.  Two output rows (2r, 2r+1) of a column take 4 transformed input rows instead of 2 x 3:
//     T0 = d0 - d2   T1 = d1 + d2   T2 = d2 - d1   T3 = d1 - d3          (d_i = padded input row 2r + i, per pixel and channel)
//     U_f[kx] = sum_ky G[f][ky] g[ky][kx],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]     (per kx, cin, cout; packed at upload)
//     M_f = sum_{kx, cin} U_f[kx] T_f[x + kx]                                                (the MFMA work: 12 products per 2 outputs
//     y(2r) = M0 + M1 + M2      y(2r+1) = M1 - M2 - M3                                        instead of 18: 1.5 x fewer MFMAs)
// The 2-D form F(2x2,3x3) would save 2.25 x but needs 16 fp32 accumulators per 4 outputs: 64 K accumulators per CU bound a
// workgroup to 64 tiles x 64 channels, whose operands (weights 32 KB + patch 10 KB per 16 input channels and 512 MFMA cycles) ask
// the L2 for ~83 B/clk/CU -- more than a CU can take.  The 1-D form keeps 2 accumulators per output: the block is the 256 pixels
// x 128 channels of the direct kernel, ONE block per CU with its 256 accumulator registers in AGPRs.
//
// Error side, measured before anything was built (tools/probe/winograd_error.py, profiles/r06_winograd_error.txt): T is rounded
// to fp16 once more than the direct path's operands; per layer 3.0e-4 .. 6.0e-4 of the output against 2.0e-4 .. 4.0e-4 direct.
//
// Layout in LDS: the transformed patch of a K-chunk (32 input channels), rows (pair-row, f) x 18 pixels (pitch 20) x 64 B with
// the XOR swizzle of conv.hip, double-buffered; the loader takes the four raw rows of a (pair-row, pixel, 16-byte piece) item
// straight from global memory into registers, transforms them with packed fp16 adds and stores four pieces.  MFMA operands as in
// conv.hip: A = weights, pre-packed fragments [Cout/32][f*3+kx][Cin/16][lane][8] streamed from global memory (L2) two taps ahead;
// B = pixels, one ds_read_b128 per fragment a tap ahead; a wave owns MT pixel tiles (2 pair-rows x 16 columns) x NT channel tiles
// x 4 row frequencies.  One LDS-only barrier per K-chunk (s_waitcnt lgkmcnt(0) + s_barrier: the weight loads stay in flight).
#include "common.h"
#include <algorithm>

#include <stdio.h>
namespace {
// Phase timing build (-DWINO_TS): lane 0 of every wave stamps s_memtime at its phase boundaries; the launcher prints the means.
#ifdef WINO_TS
__device__ unsigned long long wino_ts[8192 * 4 * 8];
#define WTS(slot) do { if ((threadIdx.x & 63) == 0) { const unsigned fl_ = blockIdx.x + gridDim.x * blockIdx.y; if (fl_ < 8192) wino_ts[(fl_ * 4 + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define WTS(slot) do {} while (0)
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int TW = 16;          // tile width in pixels
constexpr int PITCH = 20;       // patch row pitch in pixels (18 used)
constexpr int BK = 32;          // input channels per K-chunk

// a - b on eight packed fp16 values: v_pk_add_f16 with the neg modifiers on the second operand (hipcc expands the vector
// subtraction into v_sub_f16 + v_sub_f16_sdwa + v_pack_b32_f16 per pair: three instructions for one)
__device__ __forceinline__ half8 pk_sub(half8 a, half8 b) {
  u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b), r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned o;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(ua[i]), "v"(ub[i]));
    r[i] = o;
  }
  return __builtin_bit_cast(half8, r);
}

__device__ __forceinline__ int reflect_idx(int i, int n) {      // 1-px REFLECT padding; ragged tiles clamp (masked at the store)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  i = i < 0 ? 0 : i;
  return i >= n ? n - 1 : i;
}

// IL: the schedule for ONE wave per SIMD (the 16-row tiles: 256 accumulator registers per lane in AGPRs).  Nothing but the wave's
// own instruction order hides a latency there, so every MFMA is followed by its share of the tap's other work in SOURCE order,
// pinned by a sched_barrier: the weight loads of tap t + 3, the pixel reads of tap t + 1, and the staging of the next chunk's
// patch spread over the taps (raw rows requested slot by slot at taps 2.., transformed and parked slot by slot at taps 7.., one
// barrier after tap 10) -- no branch in the loop body: the last chunk stages a patch nobody reads.
template <int TH, int BN, int WM, int WN, bool IL>
__global__ __launch_bounds__(256, TH >= 16 ? 1 : 2)
void conv3x3_wino_kernel(ConvArgs p, int tiles_x, int n_tiles, int dbg_arg) {
#ifdef WCT_TUNING
  const int dbg = dbg_arg;      // ablation switches (tuning builds, WCT_WINO_DBG): 1 no weight loads, 2 no patch staging, 4 no MFMAs, 8 no fragment reads, 16 no epilogue
#else
  constexpr int dbg = 0;
#endif
  constexpr int PR = TH / 2;                 // pair-rows of the tile
  constexpr int MT = (PR / 2) / WM;          // MFMA pixel tiles (2 pair-rows x 16 columns) per wave
  constexpr int NT = (BN / 32) / WN;
  constexpr int G = 256 / PR;                // loader threads per pair-row
  constexpr int SL = (72 + G - 1) / G;       // (pixel, piece) slots per loader thread: 18 x 4 per pair-row
  constexpr int PATCH_BYTES = PR * 4 * PITCH * 64;
  constexpr int DUMP_OFF = 2 * PATCH_BYTES;  // IL: where the idle lanes of the last loader slot put their four pieces (8 KB)
  constexpr int PF_TAP = 5;                  // tap at which the next chunk's raw rows are requested; they are parked at tap 10
  static_assert(PR % 2 == 0 && (PR / 2) % WM == 0 && 256 % PR == 0 && WM * WN == 4, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  WTS(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  const int ntile = bid % n_tiles;
  bid /= n_tiles;
  const int tx = bid % tiles_x, ty = bid / tiles_x;
  const int b = blockIdx.y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = ntile * BN;

  const int Hin = p.upsample ? p.H / 2 : p.H;
  const int Win = p.upsample ? p.W / 2 : p.W;
  const half_t* xb = p.x + (size_t)b * Hin * Win * p.Cin;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_wino, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7FFFFFFF, 0x00020000);

  // ---- loader: thread (lpr, ls) takes the slots ls, ls + G, .. of pair-row lpr
  const int lpr = tid / G, ls = tid % G;
  unsigned row_off[4], col_off[SL], dst_off[SL];
  bool slot_ok[SL];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int iy = reflect_idx(y0 + 2 * lpr - 1 + i, p.H);
    if (p.upsample) iy >>= 1;
    row_off[i] = (unsigned)(iy * Win) * p.Cin * 2;
  }
#pragma unroll
  for (int j = 0; j < SL; ++j) {
    const int slot = ls + j * G;
    slot_ok[j] = slot < 72;
    const int px = slot_ok[j] ? slot >> 2 : 0, piece = slot & 3;
    int ix = reflect_idx(x0 - 1 + px, p.W);
    if (p.upsample) ix >>= 1;
    col_off[j] = (unsigned)(ix * p.Cin + piece * 8) * 2;
    dst_off[j] = ((lpr * 4 * PITCH + px) * 4 + (piece ^ ((px >> 2) & 3))) * 16;
    if (IL && !slot_ok[j]) dst_off[j] = DUMP_OFF + tid * 16;        // (no exec-mask branch in the interleaved loop)
  }
  u32x4 raw[SL][4];
  auto load_rows = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < SL; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        raw[j][i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, row_off[i] + col_off[j], chunk * BK * 2, 0);
  };
  auto park_rows = [&](int buf) {             // T = B^T d on packed fp16 pairs, four pieces per slot
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const half8 d0 = __builtin_bit_cast(half8, raw[j][0]), d1 = __builtin_bit_cast(half8, raw[j][1]);
      const half8 d2 = __builtin_bit_cast(half8, raw[j][2]), d3 = __builtin_bit_cast(half8, raw[j][3]);
      const half8 t0 = pk_sub(d0, d2), t1 = d1 + d2, t2 = pk_sub(d2, d1), t3 = pk_sub(d1, d3);
      if (slot_ok[j]) {
        unsigned char* d = smem + buf * PATCH_BYTES + dst_off[j];
        *reinterpret_cast<half8*>(d) = t0;
        *reinterpret_cast<half8*>(d + PITCH * 64) = t1;
        *reinterpret_cast<half8*>(d + 2 * PITCH * 64) = t2;
        *reinterpret_cast<half8*>(d + 3 * PITCH * 64) = t3;
      }
    }
  };

  // ---- fragment addressing
  const int frag_px = lane & 15, frag_pr = (lane & 31) >> 4, kgrp = lane >> 5;
  int rd_base[3][2];            // pixel (pair-row wm*MT*2 + frag_pr, column frag_px + kx), k-step ks; (mt, f) rows are immediates
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int px = frag_px + kx;
      rd_base[kx][ks] = (((wm * MT * 2 + frag_pr) * 4) * PITCH + px) * 64 + (((ks * 2 + kgrp) ^ ((px >> 2) & 3)) * 16);
    }
  const int c16 = p.Cin >> 4;
  unsigned wfrag[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wfrag[nt] = (((n0 >> 5) + wn * NT + nt) * 12 * c16 * 512 + lane * 8) * 2;

  // the bias rides in M1, the one frequency that enters both output rows with +1 (y(2r) = M0 + M1 + M2, y(2r+1) = M1 - M2 - M3):
  // acc register r of a tile holds channel (r & 3) + 8 (r >> 2) + 4 kgrp
  f32x16 acc[4][NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x4 bv[4];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) bv[rq] = *reinterpret_cast<const f32x4*>(p.bias + n0 + (wn * NT + nt) * 32 + 8 * rq + 4 * kgrp);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][nt][mt][r] = f == 1 ? bv[r >> 2][r & 3] : 0.f;
  }

  const int n_chunks = p.Cin / BK;             // even
  half8 wf[3][NT][2];                          // weight fragments of three taps: in use, next, the one after
  half8 bf[IL ? 3 : 2][2][MT];                 // pixel fragments of two taps (both k-steps); IL: three -- read two taps ahead
  auto load_w = [&](half8 (&dst)[NT][2], int tap12, int chunk) {
    const int soff = (tap12 * c16 + chunk * 2) * 1024;          // uniform
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        dst[nt][ks] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wfrag[nt] + ks * 1024, soff, 0));
  };
  auto read_b = [&](half8 (&dst)[2][MT], int f, int kx, int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        dst[ks][mt] = *reinterpret_cast<const half8*>(smem + buf * PATCH_BYTES + rd_base[kx][ks] + (mt * 8 + f) * PITCH * 64);
  };

  if constexpr (IL) {
    constexpr int NM = 2 * MT * NT;              // MFMAs of a tap
    constexpr int NR = 2 * NT + 2 * MT;          // its regular fillers: weight loads (tap t + 3), pixel reads (tap t + 1)
#ifndef WINO_LT0
#define WINO_LT0 1
#define WINO_PT0 6
#endif
    constexpr int LT0 = WINO_LT0, PT0 = WINO_PT0;              // staging: slot s is requested at tap LT0 + s, parked at tap PT0 + s; barrier after tap 9
    static_assert(LT0 + SL <= PT0 && PT0 + SL <= 10, "staging schedule");
#ifndef WINO_WRING
#define WINO_WRING 4
#endif
#ifndef WINO_ABL
#define WINO_ABL 0
#endif
#ifndef WINO_RAW_AUX
#define WINO_RAW_AUX 0         // cache policy of the raw-row loads (experiment: 1 sc0, 2 nt, 16 sc1)
#endif
    constexpr int ABL = WINO_ABL;                // (compile-time ablation of the interleaved loop: 1 no weight loads, 2 no raw-row loads, 4 no MFMAs, 8 no pixel reads, 16 no park, 32 raw rows of chunk 0 always, 64 raw-row loads of one cache line per instruction)
    constexpr int WR = WINO_WRING, WA = WR - 1;  // weight fragments of WR taps: loaded WA taps ahead (WR divides 24)
    static_assert(24 % WR == 0, "ring");
    half8 wq[WR][NT][2];
    half8 tq[4];                                 // the transformed pieces of the slot being parked
    load_rows(0);
#pragma unroll
    for (int i = 0; i < WA; ++i) load_w(wq[i], i, 0);
    // (first patch: every slot valid or dumped -- the plain park with the predicate replaced by the dump address)
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const half8 d0 = __builtin_bit_cast(half8, raw[j][0]), d1 = __builtin_bit_cast(half8, raw[j][1]);
      const half8 d2 = __builtin_bit_cast(half8, raw[j][2]), d3 = __builtin_bit_cast(half8, raw[j][3]);
      unsigned char* d = smem + dst_off[j];
      *reinterpret_cast<half8*>(d) = pk_sub(d0, d2);
      *reinterpret_cast<half8*>(d + PITCH * 64) = d1 + d2;
      *reinterpret_cast<half8*>(d + 2 * PITCH * 64) = pk_sub(d2, d1);
      *reinterpret_cast<half8*>(d + 3 * PITCH * 64) = pk_sub(d1, d3);
    }
    WTS(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    WTS(2);
    read_b(bf[0], 0, 0, 0);
    read_b(bf[1], 0, 1, 0);
#pragma unroll 1
    for (int pair = 0; pair < n_chunks; pair += 2) {
#pragma unroll
      for (int t = 0; t < 24; ++t) {
        const int tt = t % 12, chunk_i = pair + t / 12, buf = t / 12;
        const int f = tt / 3, kx = tt % 3;
        const int chunk_n = chunk_i + 1 < n_chunks ? chunk_i + 1 : chunk_i;     // uniform; the last chunk re-stages itself
        const int t3 = (tt + WA) % 12, c3 = tt + WA >= 12 ? chunk_n : chunk_i;
        const int t1 = (tt + 2) % 12, buf1 = tt >= 10 ? buf ^ 1 : buf;        // (the pixel reads run two taps ahead)
        const int soff3 = (t3 * c16 + c3 * 2) * 1024;
#pragma unroll
        for (int j = 0; j < NM; ++j) {
          const int ks = j / (MT * NT), mt = (j / NT) % MT, nt = j % NT;
          if (!(ABL & 4)) acc[f][nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[t % WR][nt][ks], bf[t % 3][ks][mt], acc[f][nt][mt], 0, 0, 0);
          else asm volatile("" :: "v"(wq[t % WR][nt][ks]), "v"(bf[t % 3][ks][mt]));
#pragma unroll
          for (int k = j * NR / NM; k < (j + 1) * NR / NM; ++k) {
            if (k < 2 * NT) {
              if (!(ABL & 1)) wq[(t + WA) % WR][k / 2][k % 2] =
                  __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wfrag[k / 2] + (k % 2) * 1024, soff3, 0));
            } else {
              const int kk = k - 2 * NT, ks1 = kk / MT, mt1 = kk % MT;
              if (!(ABL & 8)) bf[(t + 2) % 3][ks1][mt1] = *reinterpret_cast<const half8*>(smem + buf1 * PATCH_BYTES + rd_base[t1 % 3][ks1] + (mt1 * 8 + t1 / 3) * PITCH * 64);
            }
          }
          if (tt >= LT0 && tt < LT0 + SL) {            // one slot's four raw rows: a load behind each of the first four MFMAs
            const int sl = tt - LT0;
            if (j < 4 && !(ABL & 2)) raw[sl][j] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (ABL & 64) ? (lane & 3) * 16 + j * 64 + sl * 256 : row_off[j] + col_off[sl], (ABL & 32) ? 0 : chunk_n * BK * 2, WINO_RAW_AUX);
          }
          if (tt >= PT0 && tt < PT0 + SL && !(ABL & 16)) {            // one slot parked: a transformed piece, then its store, per pair of MFMAs
            const int sl = tt - PT0;
            {                                            // 8 items over the tap's MFMAs
#pragma unroll
              for (int it = j * 8 / NM; it < (j + 1) * 8 / NM; ++it) {
                const int fq = it >> 1;
                if ((it & 1) == 0) {
                  const half8 da = __builtin_bit_cast(half8, raw[sl][fq == 0 ? 0 : fq == 1 ? 1 : fq == 2 ? 2 : 1]);
                  const half8 db = __builtin_bit_cast(half8, raw[sl][fq == 0 ? 2 : fq == 1 ? 2 : fq == 2 ? 1 : 3]);
                  tq[fq] = fq == 1 ? da + db : pk_sub(da, db);
                } else {
                  *reinterpret_cast<half8*>(smem + (buf ^ 1) * PATCH_BYTES * (slot_ok[sl] ? 1 : 0) + dst_off[sl] + fq * PITCH * 64) = tq[fq];
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (tt == 9) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    WTS(3);
  } else {
  load_rows(0);
  load_w(wf[0], 0, 0);
  load_w(wf[1], 1, 0);
  park_rows(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  read_b(bf[0], 0, 0, 0);

#pragma unroll 1
  for (int pair = 0; pair < n_chunks; pair += 2) {
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      const int tt = t % 12, chunk_i = pair + t / 12, buf = t / 12;
      const int f = tt / 3, kx = tt % 3;
      const bool more = chunk_i + 1 < n_chunks;                 // uniform
      // 1) the weights of tap t + 2 (past the very end: re-read, unused)
      {
        const int t2 = (tt + 2) % 12;
        const int c2 = tt + 2 >= 12 ? (more ? chunk_i + 1 : chunk_i) : chunk_i;
        if (!(dbg & 1)) load_w(wf[(t + 2) % 3], t2, c2);
      }
      if (tt == PF_TAP && more && !(dbg & 2)) load_rows(chunk_i + 1);
      // 2) the pixels of tap t + 1; across the chunk boundary from the other buffer (published by the barrier of tap 10)
      if (!(dbg & 8)) {
        if (tt != 11) read_b(bf[(t + 1) & 1], (tt + 1) / 3, (tt + 1) % 3, buf);
        else if (more) read_b(bf[(t + 1) & 1], 0, 0, buf ^ 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      // 3) this tap's MFMAs
      if (!(dbg & 4))
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[f][nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t % 3][nt][ks], bf[t & 1][ks][mt], acc[f][nt][mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // 4) the next chunk's patch: parked in the other buffer, published for tap 11's reads
      if (tt == 10 && more) {
        if (!(dbg & 2)) park_rows(buf ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }

  }     // !IL
  if (dbg & 16) return;
  // ---- epilogue: y(2r) = M0 + M1 + M2, y(2r+1) = M1 - M2 - M3 (the bias came in with M1), ReLU on the rounded pairs, (2x2 max-pool), stores.
  // acc register r of a tile holds channel (r & 3) + 8 (r >> 2) + 4 kgrp of pixel (pair-row frag_pr, column frag_px).
  constexpr int OOB = (int)0x80000000u;
  const int Ho = p.pool ? (p.H + 1) / 2 : p.H, Wo = p.pool ? (p.W + 1) / 2 : p.W;
  const unsigned img_elems = (unsigned)Ho * Wo * p.Cout;
  const __amdgpu_buffer_rsrc_t r16 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y16 + (size_t)b * img_elems), 0, img_elems * 2, 0x00020000);
  const float lo1 = p.relu ? 0.f : -__builtin_inff();
  const h2 lo2 = {(half_t)lo1, (half_t)lo1};
  const h2 zero2 = {(half_t)0.f, (half_t)0.f};
  const int ox = x0 + frag_px;
  const int chan = n0 + wn * NT * 32;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int oy = y0 + ((wm * MT + mt) * 2 + frag_pr) * 2;        // the even row of this lane's pair
    const bool in0 = (oy < p.H) & (ox < p.W), in1 = (oy + 1 < p.H) & (ox < p.W);
    int off[2];
    if (p.pool) {
      off[0] = in0 && (frag_px & 1) == 0 ? (((oy >> 1) * Wo + (ox >> 1)) * p.Cout + chan + 8 * kgrp) * 2 : OOB;
      off[1] = OOB;
    } else {
      off[0] = in0 ? ((oy * p.W + ox) * p.Cout + chan + 8 * kgrp) * 2 : OOB;
      off[1] = in1 ? (((oy + 1) * p.W + ox) * p.Cout + chan + 8 * kgrp) * 2 : OOB;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      unsigned pk[2][4][2];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 e, o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = rq * 4 + j;
          const float m0 = acc[0][nt][mt][r], m1 = acc[1][nt][mt][r], m2 = acc[2][nt][mt][r], m3 = acc[3][nt][mt][r];
          e[j] = (m0 + m1) + m2;
          o[j] = (m1 - m2) - m3;
        }
        h2 e0 = {(half_t)e[0], (half_t)e[1]}, e1 = {(half_t)e[2], (half_t)e[3]};
        h2 o0 = {(half_t)o[0], (half_t)o[1]}, o1 = {(half_t)o[2], (half_t)o[3]};
        e0 = __builtin_elementwise_max(e0, lo2); e1 = __builtin_elementwise_max(e1, lo2);
        o0 = __builtin_elementwise_max(o0, lo2); o1 = __builtin_elementwise_max(o1, lo2);
        if (p.pool) {
          // the row pair is this lane's own (e, o); the column pair sits in lane ^ 1 (DPP quad_perm(1,0,3,2)); cells outside
          // the image are 0, neutral after the ReLU (ceil-mode edge)
          h2 q[2] = {__builtin_elementwise_max(in0 ? e0 : zero2, in1 ? o0 : zero2), __builtin_elementwise_max(in0 ? e1 : zero2, in1 ? o1 : zero2)};
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const unsigned tq = __builtin_bit_cast(unsigned, q[d]);
            const h2 nb = __builtin_bit_cast(h2, __builtin_amdgcn_update_dpp(0, (int)tq, 0xB1, 0xF, 0xF, true));
            q[d] = __builtin_elementwise_max(q[d], nb);
          }
          e0 = q[0]; e1 = q[1];
        }
        pk[0][rq][0] = __builtin_bit_cast(unsigned, e0); pk[0][rq][1] = __builtin_bit_cast(unsigned, e1);
        pk[1][rq][0] = __builtin_bit_cast(unsigned, o0); pk[1][rq][1] = __builtin_bit_cast(unsigned, o1);
      }
      // the two half-waves own interleaved 8-byte pieces: one v_permlane32_swap per dword pairs quad 2m with quad 2m+1 so that
      // the lower half stores channels 16m..16m+7 and the upper half 16m+8..16m+15 -- 16-byte stores (conv.hip's epilogue)
#pragma unroll
      for (int par = 0; par < 2; ++par) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          auto sx = __builtin_amdgcn_permlane32_swap(pk[par][2 * m][0], pk[par][2 * m + 1][0], false, false);
          auto sy = __builtin_amdgcn_permlane32_swap(pk[par][2 * m][1], pk[par][2 * m + 1][1], false, false);
          u32x4 ov = {sx[0], sy[0], sx[1], sy[1]};
          __builtin_amdgcn_raw_buffer_store_b128(ov, r16, off[par] + (nt * 32 + 16 * m) * 2, 0, 0);
        }
      }
    }
  }
  WTS(4);
#ifdef WINO_TS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WTS(5);
#endif
}

template <int TH, int BN, int WM, int WN, bool IL = false>
int launch_wino_cfg(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = cdiv(a.W, TW), tiles_y = cdiv(a.H, TH), n_tiles = a.Cout / BN;
  const size_t lds = 2 * (size_t)(TH / 2) * 4 * PITCH * 64 + (IL ? 8192 : 0);
  static const int dbg = tune_int("WCT_WINO_DBG", 0);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<TH, BN, WM, WN, IL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((conv3x3_wino_kernel<TH, BN, WM, WN, IL>), dim3(tiles_x * tiles_y * n_tiles, a.B), dim3(256), lds, s, a, tiles_x, n_tiles, dbg);
#ifdef WINO_TS
  if (IL) {
    (void)hipStreamSynchronize(s);
    static unsigned long long host[8192 * 4 * 8];
    (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(wino_ts), sizeof(host));
    const int nb = (int)std::min<long>((long)tiles_x * tiles_y * n_tiles * a.B, 8192);
    double sum[5] = {0}; unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nb * 4; ++i) {
      const unsigned long long* h = host + (size_t)i * 8;
      sum[0] += (double)(h[1] - h[0]); sum[1] += (double)(h[2] - h[1]); sum[2] += (double)(h[3] - h[2]);
      sum[3] += (double)(h[4] - h[3]); sum[4] += (double)(h[5] - h[4]);
      t0 = std::min(t0, h[0]); t1 = std::max(t1, h[5]);
    }
    fprintf(stderr, "WTS <%d,%d,%d,%d> Cin %d Cout %d H %d B %d blocks %d: prologue (entry -> first patch parked) %.0f, barrier %.0f, mainloop %.0f (%d chunks: %.0f per chunk), epilogue %.0f, store drain %.0f; first entry -> last exit %.0f cycles\n",
            TH, BN, WM, WN, a.Cin, a.Cout, a.H, a.B, nb, sum[0] / (nb * 4), sum[1] / (nb * 4), sum[2] / (nb * 4), a.Cin / 32, sum[2] / (nb * 4) / (a.Cin / 32),
            sum[3] / (nb * 4), sum[4] / (nb * 4), (double)(t1 - t0));
  }
#endif
  HIP_TRY(hipGetLastError());
  return WCT_OK;
}
}  // namespace

// The layers this kernel takes (launch_conv3x3 asks): fp16 output only (no fp32 tap, no statistics, no conv1_1 in the loader).
// Every tile shape accumulates in the same order (chunk, f, kx, k-step), so the choice below -- made from the grid size, like
// launch_conv3x3's -- does not change a bit of the output: batch 32 == single pair holds.
bool conv3x3_wino_takes(const ConvArgs& a) {
  return a.w_wino != nullptr && a.y16 != nullptr && a.y32 == nullptr && a.usum == nullptr && a.img1 == nullptr &&
         a.Cin % 64 == 0 && a.Cout % 64 == 0 && (!a.pool || a.relu);
}

int launch_conv3x3_wino(const ConvArgs& a, hipStream_t s) {
  ARG_CHECK(conv3x3_wino_takes(a) && a.H > 1 && a.W > 1 && a.B > 0);
  ARG_CHECK(!a.upsample || (a.H % 2 == 0 && a.W % 2 == 0));
  ARG_CHECK((size_t)a.H * a.W * a.Cin * 2 < ((size_t)1 << 31));
  ARG_CHECK((size_t)a.H * a.W * a.Cout * 2 < ((size_t)1 << 31));
  const long px16 = (long)cdiv(a.W, TW) * cdiv(a.H, 16) * a.B;
  static const int force = tune_int("WCT_WINO_CFG", 0);   // tuning switch
  if (force == 1 && a.Cout % 128 == 0) return launch_wino_cfg<16, 128, 2, 2>(a, s);
  if (force == 2) return launch_wino_cfg<16, 64, 2, 2>(a, s);
  if (force == 3) return launch_wino_cfg<8, 64, 2, 2>(a, s);
  if (force == 4 && a.Cout % 128 == 0) return launch_wino_cfg<8, 128, 1, 4>(a, s);
  if (force == 5 && a.Cout % 128 == 0) return launch_wino_cfg<16, 128, 1, 4>(a, s);
  if (force == 6 && a.Cout % 128 == 0) return launch_wino_cfg<16, 128, 1, 4, true>(a, s);
  if (force == 7 && a.Cout % 128 == 0) return launch_wino_cfg<16, 128, 2, 2, true>(a, s);
  // one 256-pixel x 128-channel block per CU with the interleaved schedule where that fills the chip; below, blocks of 8 rows at
  // two per CU (the second block covers the first one's prologue and epilogue)
  const long px8 = (long)cdiv(a.W, TW) * cdiv(a.H, 8) * a.B;
  if (a.Cout % 128 == 0 && px16 * (a.Cout / 128) >= 256) return launch_wino_cfg<16, 128, 1, 4, true>(a, s);
  if (a.Cout % 128 == 0 && px8 * (a.Cout / 128) >= 256) return launch_wino_cfg<8, 128, 1, 4>(a, s);
  return launch_wino_cfg<8, 64, 2, 2>(a, s);
}
