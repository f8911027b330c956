// Shared declarations for libwct_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WCT_OK 0
#define WCT_ERR_HIP -1
#define WCT_ERR_ARG -2
#define WCT_ERR_STATE -3
#define WCT_ERR_NOMEM -4
#define WCT_ERR_NOCONV -5

void wct_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      wct_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                    __FILE__, __LINE__);                                       \
      return WCT_ERR_HIP;                                                      \
    }                                                                          \
  } while (0)

#define ARG_CHECK(cond)                                                        \
  do {                                                                         \
    if (!(cond)) {                                                             \
      wct_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__);\
      return WCT_ERR_ARG;                                                      \
    }                                                                          \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// A-B / tuning switches.  In the product build they are constants: a stray environment variable must never change the
// numerics of a library whose contract is bit-reproducible output.  Only a -DWCT_TUNING build (tools/, experiments on the
// GPU box) reads them from the environment.  THREE documented TEST hooks stay live in every build, all safe by
// construction: WCT_JACOBI_MAX_SWEEPS can only LOWER the sweep budget (clamped to the compiled one; the solve then fails
// loudly, never silently); WCT_FUSE_STATS=0, WCT_FUSE_CONV1=0 and WCT_FUSE_TAIL=1 each select a path whose output is
// bit-identical (asserted by tests/test_gpu_pipeline.py); WCT_WINOGRAD=0 keeps every 3x3 layer on the direct kernel (the
// round-5 arithmetic: other roundings, same tolerances).  grep getenv: these five and nothing else outside #ifdef WCT_TUNING.
#ifdef WCT_TUNING
#include <stdlib.h>
static inline int tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline float tune_float(const char* name, float dflt) { const char* e = getenv(name); return e ? (float)atof(e) : dflt; }
static inline bool tune_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline int tune_int(const char*, int dflt) { return dflt; }
static inline float tune_float(const char*, float dflt) { return dflt; }
static inline bool tune_set(const char*) { return false; }
#endif

// ---- conv.hip -------------------------------------------------------------
struct ConvArgs {
  const half_t* x;     // [B][Hin][Win][Cin] fp16 NHWC
  const half_t* w;     // [Cout][9][Cin] fp16 (tap-major, K contiguous)
  const float* bias;   // [Cout]
  half_t* y16;         // [B][H][W][Cout] fp16 or nullptr
  float* y32;          // [B][H][W][Cout] fp32 or nullptr
  int B, H, W;         // output size (== input size, or 2x input if upsample)
  int Cin, Cout;
  int upsample;        // input is the x2 nearest upsample of x (model.py:293)
  int relu;
  int pool;            // fuse the following 2x2/2 'same' max-pool: outputs are [(H+1)/2][(W+1)/2][Cout]
  int xcd_map = 0;     // (tuning switch WCT_CONV_XCD) tile order that keeps the blocks of a pixel tile on one XCD (conv.hip)
  // feature statistics from the fp32 epilogue (null: off; need y32, relu, W % 16 == 0): usum [B][H*W/16][Cout] = the sum of
  // every run of 16 consecutive pixels (unit_row_sum's fixed tree -- what colsum_kernel computes from the stored features),
  // umax [B][UMAX_SLOTS] = bit patterns whose maximum is the largest value of the image (>= 0 after the ReLU), merged with
  // atomicMax: zero it before the launch
  float* usum;
  unsigned* umax;
  // conv1_1 fused into the patch loader (conv1_2 of an encoder pass whose relu1_1 nobody else reads; Cin = Cout = 64): the
  // block computes its halo patch of relu1_1 activations from the IMAGE instead of reading them -- x is unused, img1 is the
  // [B][H][W][3] fp32 image and (w1frag, bias1, clamp01) are conv1_1's ConvFirstArgs.  null img1: off.
  const float* img1 = nullptr;
  const half_t* w1frag = nullptr;
  const float* bias1 = nullptr;
  int clamp01 = 0;
  // the layer's filters transformed for the reduced-FLOP kernel (conv_wino.hip): fp16 MFMA A-fragments
  // [Cout/32][f*3+kx][Cin/16][lane][8] of U_f[kx] = sum_ky G[f][ky] g[ky][kx].  null: the direct kernel only.
  const half_t* w_wino = nullptr;
};
// The maxima of an image are merged into UMAX_SLOTS words (two cache lines) instead of one: tens of thousands of
// wavefronts bumping ONE word -- or 32 words of one cache line -- serialise in a single L2 channel (measured: +94 us per
// launch); a wave also reads its slot first and skips the atomic when it cannot raise it (values only grow).
constexpr int UMAX_SLOTS = 64;
__device__ __forceinline__ void umax_merge(unsigned* umax_img, int slot, float vmax, int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  if (lane == 0) {
    unsigned* w = umax_img + (slot & (UMAX_SLOTS - 1));
    const unsigned bits = __builtin_bit_cast(unsigned, vmax);
    if (bits > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, bits);
  }
}
int launch_conv3x3(const ConvArgs& a, hipStream_t s);
// conv_wino.hip: Winograd F(2,3) along y x direct along x -- 12 MFMA products per 2 outputs instead of 18 (fp16 output only)
bool conv3x3_wino_takes(const ConvArgs& a);
int launch_conv3x3_wino(const ConvArgs& a, hipStream_t s);

struct ConvFirstArgs {   // 3 -> 64 with the 1x1 'preprocess' folded in
  const float* x;      // [B][H][W][3] fp32 image in [0,1]
  const half_t* wfrag; // folded weights as fp16 hi/lo MFMA A-fragments [cout/32][k-step 2][hi,lo][lane][8], k = tap*3+cin (27 -> 32)
  const float* bias;   // [64]
  half_t* y16;
  float* y32;
  int B, H, W;
  int clamp01;         // clip(x,0,1) at load (model.py:86)
  float* usum;         // as in ConvArgs (with y32; W % 16 == 0)
  unsigned* umax;
};

// Sums over the 16 lanes of a DPP row (= the 16 pixels of one row of an MFMA pixel tile), the same bits in every
// lane: ((p0+p1)+(p2+p3) + (p4+p5)+(p6+p7)) + (the same of p8..p15).  colsum_kernel adds 16 consecutive feature rows in
// exactly this tree, so statistics taken in a conv epilogue and statistics taken from the stored features agree bit for bit.
// Four values at once (the clang DPP builtin costs a v_mov_b32_dpp + v_add_f32 per step; written out, a step is one
// v_add_f32_dpp, and interleaving the four chains covers the VALU-write -> DPP-read wait states; the leading s_nop
// covers the writer of the inputs).  dst = dpp(src0) + src1 with all three the same register.
__device__ __forceinline__ void unit_row_sum4(f32x4& t) {
  float a = t[0], b = t[1], c = t[2], d = t[3];
#define WCT_DPP_STEP(ctrl)                                                             \
  "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"       \
  "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"       \
  "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"       \
  "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  asm volatile("s_nop 1\n"
               WCT_DPP_STEP("quad_perm:[1,0,3,2]")
               WCT_DPP_STEP("quad_perm:[2,3,0,1]")
               WCT_DPP_STEP("row_half_mirror")
               WCT_DPP_STEP("row_mirror")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef WCT_DPP_STEP
  t[0] = a; t[1] = b; t[2] = c; t[3] = d;
}
int launch_conv_first(const ConvFirstArgs& a, hipStream_t s);

struct ConvLastArgs {    // 64 -> 3, no activation (model.py:298)
  const half_t* x;     // [B][H][W][64]
  const half_t* wfrag; // fp16 MFMA A-fragments [k-step 4][lane][8]: row tap*3+cout (27 of 32), k = cin
  const float* bias;   // [3]
  float* y;            // [B][H][W][3] fp32, unclipped
  int B, H, W;
};
int launch_conv_last(const ConvLastArgs& a, hipStream_t s);

// conv_tail.hip: the last 64 -> 64 conv of a decoder (ReLU; its input x2-upsampled if `upsample`) and the 64 -> 3 output conv in one
// kernel -- the 64-channel full-resolution map stays in LDS.  The bits of launch_conv3x3 + launch_conv_last.
struct ConvTailArgs {
  const half_t* x;      // [B][Hin][Win][64] fp16 (Hin = H / 2 if upsample)
  const half_t* w;      // the 64 -> 64 conv's fragments (ConvArgs::w)
  const float* bias;    // [64]
  const half_t* wlast;  // the output conv's fragments (ConvLastArgs::wfrag)
  const float* blast;   // [3]
  float* y;             // [B][H][W][3] fp32, unclipped
  int B, H, W, upsample;
};
int launch_conv_tail(const ConvTailArgs& a, hipStream_t s);

int launch_maxpool2x2(const half_t* x, half_t* y, int B, int H, int W, int C, hipStream_t s);
int launch_u8_to_f32(const uint8_t* x, float* y, size_t n, hipStream_t s);       // /255 (wct.py:64)
int launch_f32_to_u8(const float* x, uint8_t* y, size_t n, hipStream_t s);       // uint8(clip*255) (wct.py:68)
int launch_f32_to_f16(const float* x, half_t* y, size_t n, hipStream_t s);
int launch_f16_to_f32(const half_t* x, float* y, size_t n, hipStream_t s);

// ---- wct.hip --------------------------------------------------------------
// generic fp32 GEMM on v_mfma_f32_32x32x2_f32:  D[m][n] = sum_k A(m,k) B(k,n)
// A element (m,k): a_kmajor ? A[k*lda+m] : A[m*lda+k];  B element (k,n): b_kmajor ? B[k*ldb+n] : B[n*ldb+k]
struct GemmArgs {
  const float* A; int lda; int a_kmajor;
  const float* B; int ldb; int b_kmajor;
  const float* a_sub_m;   // subtract vector indexed by m from A (cov centring)       or null
  const float* b_sub_n;   // subtract vector indexed by n from B                      or null
  const float* a_sub_k;   // subtract vector indexed by k from A (apply centring)     or null
  const float* a_scale_k; // scale A by vector indexed by k (E diag(d))               or null
  int M, N, K;
  int ksplit;             // K elements per blockIdx.z slice (multiple of 16)
  float* out32; half_t* out16; int ldo;
  size_t out_split_stride;  // elements between K-slices of out32 (split-K partials)
  const float* bias_n;    // added per output column or null
  // batching: blockIdx.z = batch * nsplit + split; strides in elements (0 = shared)
  int nsplit;
  size_t sA, sB, s_sub_m, s_sub_n, s_sub_k, s_scale_k, s_out, s_bias;
  // round 5 (the transform tail in merged launches over the 2P matrices of a level, batch b = matrix 2 * pair + side):
  const float* A_odd;     // if set: A of an odd batch b is A_odd + (b >> 1) * sA_odd, that of an even one A + (b >> 1) * sA
  size_t sA_odd;
  int skip_shared;        // batches b with skip_style_mat(b, 1) have nothing to do (one style for all pairs: WCT_FLAG_STYLE_SHARED)
  // blend epilogue (T = Tcs Tw -> M = alpha T + (1 - alpha) I, ops.py:83 folded into the apply matrix): the store carries the
  // blend and the block merges max |M| into mabs[batch] (bit patterns of non-negative floats; zeroed by an earlier kernel)
  int blend; float alpha; unsigned* mabs;
  // refresh products (round 5, launch_wct): run for the batches whose matrix needs it, a no-op for the others.  mask_diag: the
  // [nbatch][M][M] matrices whose DIAGONAL decides (refresh_needed, csrc/wct.hip); every block of a batch evaluates it the same way
  // and block (0, 0) stores it in mask_out[batch].  mask_in: a mask stored by an earlier launch.
  const float* mask_diag; size_t s_mask; int* mask_out; const int* mask_in;
};

int launch_gemm(GemmArgs g, int nsplit, int nbatch, hipStream_t s);

// Feature matrices are pixel-major: X[n][c] (NHWC flattened), fp32.
enum { WCT_MODE_NP = 0, WCT_MODE_TF = 1 };

// Statistics a conv epilogue took while it wrote a feature map (ConvArgs::usum / umax): [0] content, [1] style; a side
// whose u is null is summed from the features themselves (same bits either way, colsum_kernel).
struct WctFeatStats { const float* u[2]; const unsigned* umax[2]; };

// P independent whiten-colour transforms on `s`:  out = blend(T (x - mc) + ms)
// content [P][Nc][C], style [P][Ns][C]; out16/out32 [P][Nc][C] (either may be null).
size_t wct_workspace_bytes(int C, int Nc, int Ns, int P);
int launch_wct(const float* content, int Nc, const float* style, int Ns, int C, int P,
               float alpha, int mode, float eps /* <0: reference default */, half_t* out16, float* out32,
               void* workspace, size_t workspace_bytes, int* sweeps_dev, int stages, hipStream_t s,
               const hipStream_t* side /* optional extra streams: the eigenproblems are split over 1+nside */,
               int nside, hipEvent_t ev_fork, const hipEvent_t* ev_join,
               int shared_style /* style holds ONE feature map used by all P pairs: its statistics and
                                   eigensystem are computed once */,
               int* eig_fail /* device-visible status words [4 stream groups][2] (not converged, non-finite), bumped by the
                                eigensolver, or null */,
               const struct WctFeatStats* stats = nullptr /* unit sums / maxima a conv epilogue left beside the features */);
// WCT_STAGE_EIG_FP32UPDATE (with WCT_STAGE_EIG): the eigensolver's tile updates on fp32 MFMA instead of split fp16 -- style-swap, whose
// patch matching is an argmax over the whitened features (csrc/jacobi_dev.h r4::fused_u)
enum { WCT_STAGE_COV = 1, WCT_STAGE_EIG = 2, WCT_STAGE_APPLY = 4, WCT_STAGE_ALL = 7, WCT_STAGE_EIG_FP32UPDATE = 8 };
int launch_adain(const float* content, int Nc, const float* style, int Ns, int C, int P,
                 float alpha, float eps, half_t* out16, float* out32,
                 void* workspace, size_t workspace_bytes, hipStream_t s, int shared_style,
                 const struct WctFeatStats* stats = nullptr);
// Symmetric eigensolver (batched): A [nmat][C][C] is overwritten (diag -> eigenvalues),
// V [nmat][C][C] gets eigenvectors in columns.  C multiple of 32, 32 <= C <= 1024.
// sweeps_done_dev[m]: sweeps used (> 0) if matrix m converged, -sweeps if it was still rotating when the sweep
// budget ran out, <= -1000 for non-finite input; eig_fail as in launch_wct.
int launch_jacobi_eigh(float* A, float* V, int C, int nmat, void* workspace,
                       size_t workspace_bytes, int* sweeps_done_dev, int* eig_fail, hipStream_t s);
size_t jacobi_workspace_bytes(int C, int nmat);

// Style-swap at relu5_1 (ops.py:145-278): one pair, content [hc*wc][C], style [hs*ws][C] fp32.
size_t style_swap_workspace_bytes(int C, int hc, int wc, int hs, int ws, int patch, int stride);
int launch_style_swap(const float* content, int hc, int wc, const float* style, int hs, int ws, int C,
                      float alpha, int patch, int stride, float eps, half_t* out16, float* out32,
                      void* workspace, size_t workspace_bytes, hipStream_t s, int* eig_fail);

// ---- train.hip ------------------------------------------------------------
int launch_im2col_act(const half_t* x, float* col, int B, int H, int W, int C, int upsample, hipStream_t s);
int launch_im2col_grad(const float* g, float* col, int B, int H, int W, int C, hipStream_t s);
int launch_reflect_fold(const float* gp, float* g, int B, int H, int W, int C, hipStream_t s);
int launch_relu_mask16(float* g, const half_t* act, size_t n, hipStream_t s);
int launch_relu_mask32(float* g, const float* act, size_t n, hipStream_t s);
int launch_upsample_adjoint(const float* gb, float* gs, int B, int h, int w, int C, hipStream_t s);
int launch_maxpool_adjoint(const half_t* pre, const float* gpool, float* gpre, int B, int H, int W, int C, hipStream_t s);
int launch_mse(const float* a, const float* b, size_t n, float weight, float* grad, int accumulate, double* partial,
               float* loss_out, hipStream_t s);
int launch_tv(const float* x, int B, int H, int W, int C, float weight, float* grad, double* partial, float* loss_out, hipStream_t s);
int launch_bias_grad(const float* g, size_t rows, int C, float* partial, float* out, hipStream_t s);
int launch_reduce_slabs(const float* partial, size_t n, int nslab, float* out, hipStream_t s);
int launch_adam(float* w, float* m, float* v, const float* g, size_t n, float lr_t, float b1, float b2, float eps, hipStream_t s);
int launch_pack_conv_frag(const float* w, half_t* frag, int cin, int cout, hipStream_t s);
int launch_pack_conv_wino_frag(const float* w, half_t* frag, int cin, int cout, hipStream_t s);
int launch_pack_last_frag(const float* w, half_t* frag, hipStream_t s);
int launch_transpose_w(const float* w, float* wt, int cin, int cout, int cout_pad, hipStream_t s);
int launch_pad3to4(const float* x, float* y, size_t n, hipStream_t s);
int launch_conv_dgrad(const float* g, const float* wt, int B, int H, int W, int cin, int cout,
                      float* col, float* gp, float* gin, void* scratch, hipStream_t s);
int launch_conv_wgrad(const half_t* x16, int upsample, const float* g, int ldg, int B, int H, int W, int cin, int cout,
                      float* col, float* partial, int nsplit, float* dw, void* scratch, hipStream_t s);
int conv_wgrad_splits(int B, int H, int W);
int launch_pow2_scale(const float* x, size_t n, void* scratch, hipStream_t s);   // scratch[1] = 2^k with max|x| 2^k in [8192,16384)
int launch_conv_first_dgrad(const float* g, const float* wf, float* gp, int B, int H, int W, hipStream_t s);

// ---- coral.hip ------------------------------------------------------------
struct CoralApplyArgs {
  double M[9], src_mean[3], src_std[3], tgt_mean[3], tgt_std[3];
};
int launch_coral_stats(const uint8_t* img, size_t npix, unsigned long long* partial, int nblocks,
                       unsigned long long* out9, hipStream_t s);
int launch_coral_apply(const uint8_t* src, size_t npix, const CoralApplyArgs& a, uint8_t* out_u8,
                       double* out_f64, hipStream_t s);
