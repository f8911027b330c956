/* libwct_hip.so -- C ABI of the MI355X (gfx950) stylize hot path.
 *
 * The reference (eridgd/WCT-TF) has no FFI: its boundary is the Python API that
 * drives one TensorFlow session.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference tree).  INTEGRATION.md
 * shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative wct_status on failure;
 *     wct_last_error() returns a thread-local message.  No exceptions cross the ABI.
 *   - one wct_ctx = one GPU + one HIP stream; calls on a ctx are serialised
 *     (the reference's predict() is blocking and not re-entrant, wct.py:70-106);
 *     distinct contexts are independent (one per process/GPU for multi-GPU).
 *   - "host" pointers are caller-owned and only read/written during the call;
 *     "dev" pointers are device memory obtained from wct_dev_alloc (or any
 *     hipMalloc'ed / torch CUDA pointer on the same device).
 *   - images: uint8 HxWx3 RGB, row-major (wct.py:60-68).  features: float32
 *     NHWC with batch 1, i.e. [H*W][C] pixel-major (ops.py:32-33 squeeze).
 *   - weights: float32 HWIO, exactly what vgg_normalised.py:33 / Keras Conv2D hold.
 */
#ifndef WCT_HIP_H
#define WCT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wct_ctx wct_ctx;

enum wct_status { WCT_STATUS_OK = 0, WCT_STATUS_HIP = -1, WCT_STATUS_ARG = -2,
                  WCT_STATUS_STATE = -3, WCT_STATUS_NOMEM = -4,
                  /* An eigendecomposition behind the call (the stand-in for tf.svd / np.linalg.svd, ops.py:53-65,110,123)
                   * did not converge within its sweep budget (16 sweeps; 4-6 are typical at C = 512, 7-10 on graded
                   * rank-deficient spectra; the test hook WCT_JACOBI_MAX_SWEEPS can only lower it), or met NaN/Inf
                   * in a covariance.  The outputs of the call ARE written (best effort, as LAPACK does with info > 0)
                   * but must not be trusted; the reference's own worry at this spot is ops.py:57-65.  Reported by the
                   * blocking calls themselves and, for wct_stylize_batch_dev (asynchronous), by the next wct_sync. */
                  WCT_STATUS_NOCONV = -5 };

/* transform semantics: wct_np (ops.py:92-140) or the live-graph wct_tf (ops.py:24-90) */
enum wct_mode { WCT_NP = 0, WCT_TF = 1 };

/* flags for wct_stylize* */
enum wct_flags {
  WCT_FLAG_ADAIN = 1,      /* --adain: AdaIN at every level instead of WCT (model.py:148-158) */
  WCT_FLAG_MODE_NP = 2,    /* use wct_np semantics instead of the graph's wct_tf */
  WCT_FLAG_SWAP5 = 4,      /* --swap5: style-swap at relu5_1 (wins over ADAIN there, model.py:148-152) */
  WCT_FLAG_STYLE_SHARED = 8, /* wct_stylize_batch_dev only: `style` is ONE image shared by all B pairs (stylize_video.py
                              keeps one style for every frame but re-runs it per frame, stylize_video.py:88-106);
                              the style pass, statistics and eigensystems run once per call, results are identical */
  WCT_FLAG_IMAGES_F32 = 16  /* content / style point at float32 images already in [0,1] (WCT.preprocess applied by the
                              caller: a FLOAT input of predict() is divided by 255 without rounding, wct.py:60-64)
                              instead of uint8 ones; the output stays uint8 */
};

/* ---- lifecycle: replaces WCT.__init__'s tf.Session setup (wct.py:29-44) ---- */
int  wct_create(int device, wct_ctx** out);
void wct_destroy(wct_ctx* ctx);
const char* wct_last_error(void);
int  wct_sync(wct_ctx* ctx);                       /* block until the ctx stream is idle; WCT_STATUS_NOCONV if an
                                                       eigensolve of the work just completed failed (see above) */
int  wct_device_count(int* n);
/* The ctx's HIP stream (a hipStream_t), so that a caller can order its own device work -- e.g. the RCCL gather of the
 * finished frames, multi-GPU runs -- behind wct_stylize_batch_dev without a host sync (record an event on it, wait for
 * the event on the other stream).  The reference has no counterpart: its session is blocking (wct.py:97-104). */
int  wct_get_stream(wct_ctx* ctx, void** stream_out);

/* ---- weights: replace vgg_from_t7 (vgg_normalised.py:10-55) and the per-decoder
 * Saver.restore (wct.py:46-58).  The library copies, folds the 1x1 'preprocess'
 * into conv1_1 and repacks to its fp16 MFMA-fragment layout [Cout/32][tap][Cin/16][lane][8].
 *   pre_w [3][3] (in,out) and pre_b [3]: the 1x1 preprocess conv;
 *   w[i] HWIO 3x3 and b[i] for conv1_1, conv1_2, conv2_1, conv2_2, conv3_1..3_4,
 *   conv4_1..4_4, conv5_1 (13 layers). */
int wct_set_encoder(wct_ctx* ctx, const float* pre_w, const float* pre_b,
                    const float* const* w, const float* const* b, int n_layers);
/* decoder for relu<level>_1, level 1..5; conv layers in execution order
 * (model.py:283-298): 2, 3, 5, 9, 13 layers for level 1..5. */
int wct_set_decoder(wct_ctx* ctx, int level, const float* const* w, const float* const* b,
                    int n_layers);

/* ---- op level (host pointers), for parity tests -------------------------------- */
/* wct_np / wct_tf (ops.py:24-140): content [Nc][C], style [Ns][C], out [Nc][C].
 * eps: the reference functions' `eps` argument (wct_np: added inside the spectral gains,
 * default 1e-5; wct_tf: added to the covariance diagonal, default 1e-8); eps < 0 = default.
 * Cut-off: the reference keeps eigenvalues > 1e-5 (ops.py:68-69 / 112,125).  So does this path, with one refinement for
 * covariances whose float32 rounding noise is ABOVE that absolute threshold (N < C pixels at feature scales from ~10 up: the
 * exact zeros come out as +-1.5e-7 ||cov||): a covariance is positive semi-definite, its most negative computed eigenvalue -r
 * measures that noise, and eigenvalues <= max(1e-5, 2 r) are dropped -- the outcome of the reference's formula in exact
 * arithmetic, inside the band of kept counts its own float32 evaluation can land on (csrc/wct.hip spectral_cut). */
int wct_transform(wct_ctx* ctx, const float* content, int Nc, const float* style, int Ns,
                  int C, float alpha, int mode, float eps, float* out,
                  int* sweeps_out /* [2] (content, style) or NULL: Jacobi sweeps used, > 0 when converged;
                                     -sweeps when still rotating after the sweep budget; <= -1000 for non-finite
                                     input.  Any negative entry also makes the call return WCT_STATUS_NOCONV */);
/* adain (ops.py:282-294), epsilon as in the reference signature */
int wct_adain(wct_ctx* ctx, const float* content, int Nc, const float* style, int Ns,
              int C, float alpha, float epsilon, float* out);
/* wct_style_swap (ops.py:145-278): content [hc*wc][C], style [hs*ws][C], out [hc*wc][C]; `alpha` is the
 * reference's ss_alpha; eps < 0 = its default 1e-8.  (hc, wc) must survive the patch/stride round trip
 * (utils.swap_filter_fit, wct.py:84-90) -- always true for stride 1. */
int wct_style_swap(wct_ctx* ctx, const float* content, int hc, int wc, const float* style, int hs, int ws,
                   int C, float alpha, int patch_size, int stride, float eps, float* out);
/* style-swap settings used by wct_stylize* when WCT_FLAG_SWAP5 is set: WCT(ss_patch_size, ss_stride)
 * (wct.py:17-18) and predict(ss_alpha) (wct.py:70).  Defaults 0.6 / 3 / 1 (stylize.py:34-37). */
int wct_set_style_swap(wct_ctx* ctx, float ss_alpha, int patch_size, int stride);
/* symmetric eigendecomposition used in place of tf.svd / np.linalg.svd (ops.py:53-55,110,123):
 * A [nmat][C][C] in; evals [nmat][C], evecs [nmat][C][C] (columns) out.  The UPPER triangle of A (a[i][j],
 * i <= j) is authoritative: the entry point mirrors it into the lower one on its staged copy before the
 * solve (the solver reads an element from whichever triangle is contiguous for the kernel at hand), so a
 * matrix that is symmetric only to round-off, or upper-only data, is solved as that symmetric matrix. */
int wct_eigh(wct_ctx* ctx, const float* A, int C, int nmat, float* evals, float* evecs,
             int* sweeps_out /* [nmat] or NULL; same contract as wct_transform's */);
/* Conv2DReflect (ops.py:17-19): x [H][W][Cin] fp32, w HWIO, y [Ho][Wo][Cout] fp32;
 * upsample!=0 applies UpSampling2D x2 first (model.py:293). fp16 operands, fp32 accumulate.
 * Cin and Cout must be multiples of 64 (every 3x3 layer of the path but conv1_1 / the output conv). */
int wct_conv3x3(wct_ctx* ctx, const float* x, int H, int W, int Cin, const float* w_hwio,
                const float* bias, int Cout, int relu, int upsample, float* y);
/* The same layer as the stylize pipeline runs it (Conv2DReflect, ops.py:17-19; cuDNN picks the algorithm for Keras' Conv2D,
 * vgg_normalised.py:35-40 / model.py:291): a batch x [B][H][W][Cin], fp16 activations out (returned as fp32), optionally with the
 * following MaxPooling2D(padding='same') fused (vgg_normalised.py:42; needs relu).  algo 0: the kernel the pipeline uses for
 * this layer shape; 1: the direct implicit-GEMM kernel; 2: the reduced-FLOP kernel (Winograd F(2,3) along y, csrc/conv_wino.hip).
 * y [B][Ho][Wo][Cout]. */
int wct_conv3x3_f16(wct_ctx* ctx, const float* x, int B, int H, int W, int Cin, const float* w_hwio,
                    const float* bias, int Cout, int relu, int upsample, int pool, int algo, float* y);
/* MaxPooling2D(padding='same') (vgg_normalised.py:42): y [(H+1)/2][(W+1)/2][C] */
int wct_maxpool(wct_ctx* ctx, const float* x, int H, int W, int C, float* y);
/* encoder to relu<level>_1 (model.py:135-139): img01 [H][W][3] in [0,1]; feat [h][w][C] */
int wct_encode(wct_ctx* ctx, const float* img01, int H, int W, int level, float* feat);
/* decoder for relu<level>_1 (model.py:245-304): feat [h][w][C]; img [h*2^(l-1)][w*2^(l-1)][3] */
int wct_decode(wct_ctx* ctx, const float* feat, int h, int w, int level, float* img);
/* coral_numpy / preserve_colors_np (coral.py:13-39, utils.py:87-90), the O(pixels) parts:
 *   wct_coral_stats: exact integer moments of a uint8 image on the GPU:
 *     sums[0..2] = sum_c x, sums[3..8] = sum x_i x_j for (i,j) = 00,01,02,11,12,22.
 *   wct_coral_apply: out = (M ((x/255 - src_mean)/src_std)) * tgt_std + tgt_mean in float64,
 *     and its clip/x255/truncate uint8 image (utils.py:89).  Either output may be NULL.
 * The 3x3 step between them (coral.py:30-33: X X^T + I, matSqrt through the SVD, inverse) stays on
 * the host in the caller: the reference's matSqrt multiplies U sqrt(D) by U -- not U^T -- so its
 * value is defined by LAPACK's singular-vector signs, and only LAPACK reproduces that. */
int wct_coral_stats(wct_ctx* ctx, const uint8_t* img, int H, int W, double sums[9]);
int wct_coral_apply(wct_ctx* ctx, const uint8_t* src, int H, int W, const double M[9],
                    const double src_mean[3], const double src_std[3],
                    const double tgt_mean[3], const double tgt_std[3],
                    uint8_t* out_u8, double* out_f64);

/* ---- the hot path: WCT.predict (wct.py:70-106) -------------------------------------
 * levels: relu levels in pipeline order, e.g. {5,4,3,2,1}.  Output size: wct_output_size. */
/* Images must keep a feature map of at least 2x2 at the deepest level (every conv reflect-pads by one pixel, and
 * tf.pad REFLECT refuses a 1-pixel map just the same): H, W >= 2^(level-1) + 1, else WCT_STATUS_ARG. */
int wct_output_size(int Hc, int Wc, const int* levels, int n_levels, int* Ho, int* Wo);
int wct_stylize(wct_ctx* ctx, const uint8_t* content, int Hc, int Wc,
                const uint8_t* style, int Hs, int Ws,
                const int* levels, int n_levels, float alpha, unsigned flags,
                uint8_t* out);
/* batched, device-resident variant: B independent pairs (same sizes), content [B][Hc][Wc][3],
 * style [B][Hs][Ws][3], out [B][Ho][Wo][3], all device pointers; asynchronous on the ctx
 * stream (call wct_sync).  This is what bench.py times. */
int wct_stylize_batch_dev(wct_ctx* ctx, const uint8_t* content_dev, int Hc, int Wc,
                          const uint8_t* style_dev, int Hs, int Ws, int B,
                          const int* levels, int n_levels, float alpha, unsigned flags,
                          uint8_t* out_dev);

/* ---- decoder training (model.py:123-223, train.py:129-196) ---------------------------
 * One optimiser step of the decoder for relu<level>_1 (the encoder is frozen, model.py:202):
 *   F = enc(x); D = dec(F); F' = enc(D);
 *   loss = feature_weight * mse(F', F) + pixel_weight * mse(D, x) + tv_weight * mean_b(total_variation(D))
 *   Adam(lr, beta1, beta2, eps) on the decoder's kernels and biases (tf.train.AdamOptimizer, model.py:199).
 * images: host fp32 [B][H][W][3] in [0,1] (train.py:72-83), H and W multiples of 2^(level-1).
 * step: 1-based step number (Adam bias correction); lr: the already decayed rate (torch_decay, model.py:17-19);
 * lr == 0 computes losses and gradients without touching the weights.  losses_out[4] = feature, pixel, tv, total.
 * Forward in the inference precision (fp16 activations, fp32 accumulate), backward and optimiser in fp32. */
int wct_train_step(wct_ctx* ctx, int level, const float* images, int B, int H, int W,
                   float feature_weight, float pixel_weight, float tv_weight,
                   float lr, float beta1, float beta2, float eps, int step, float* losses_out);
/* conv `layer` (0-based, the 3-channel output conv last) of the decoder for relu<level>_1: its current fp32
 * weights [3][3][Cin][Cout] / bias, and the gradients of the last wct_train_step.  Any pointer may be NULL.
 * This is what a checkpoint writer (tf.train.Saver.save, train.py:183-185) reads. */
int wct_get_decoder_layer(wct_ctx* ctx, int level, int layer, float* w_hwio, float* bias,
                          float* grad_w, float* grad_b);

/* Data-parallel training (one process per GPU): every rank runs wct_train_step(lr = 0) on its own shard of the
 * batch, the ranks average the gradient buffer below with ONE all-reduce (RCCL), then every rank applies the same
 * Adam step.  *grad_dev: one contiguous device buffer of *count floats with the gradients of all layers (the
 * layout is private and identical on every rank).  train_step(lr=0) + train_apply(lr) == train_step(lr), bit for bit. */
int wct_train_grad_buffer(wct_ctx* ctx, int level, float** grad_dev, size_t* count);
int wct_train_apply(wct_ctx* ctx, int level, float lr, float beta1, float beta2, float eps, int step);

/* ---- device memory helpers (thin wrappers so callers need no HIP binding) ----------- */
int wct_dev_alloc(wct_ctx* ctx, size_t bytes, void** out);
int wct_dev_free(wct_ctx* ctx, void* p);
int wct_h2d(wct_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int wct_d2h(wct_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- measurement: per-kernel-class HIP-event timing on the ctx stream ----------------
 * class ids: 0 conv3x3 (MFMA), 1 conv_first, 2 conv_last, 3 pool, 4 wct stats+cov,
 * 5 jacobi eigensolver, 6 wct tbuild+apply, 7 other, 8 conv12 (conv1_1 + conv1_2 + pool in one
 * launch: the content passes of the levels >= 2), 9 conv_wino (the 3x3 launches the reduced-FLOP kernel takes -- csrc/conv_wino.hip,
 * the >= 256-channel layers without a feature tap; flops are the DIRECT convolution's, the kernel executes 2/3 of them), 10
 * conv_tail (the last 64 -> 64 conv of a decoder + the 64 -> 3 output conv in one launch, csrc/conv_tail.hip; flops of the two
 * layers, the halo recomputation not counted).  When enabled every launch group is bracketed by hipEventRecord on the ctx stream;
 * wct_prof_read syncs and accumulates. */
#define WCT_PROF_CLASSES 11
int wct_prof_enable(wct_ctx* ctx, int on);
int wct_prof_reset(wct_ctx* ctx);
int wct_prof_read(wct_ctx* ctx, double ms[WCT_PROF_CLASSES], long long launches[WCT_PROF_CLASSES],
                  double flops[WCT_PROF_CLASSES], double bytes[WCT_PROF_CLASSES]);
/* Eigensolver statistics since the last call (the decompositions that replace tf.svd / np.linalg.svd, ops.py:53-55,
 * 110,123), per size class k = 0..5 (covariances of order 32 * 2^k): out[3k] = matrices solved, out[3k+1] = sum of
 * the sweeps they took, out[3k+2] = the largest sweep count.  Synchronises the ctx stream; cleared on read. */
int wct_eig_stats(wct_ctx* ctx, long long out[18]);

#ifdef __cplusplus
}
#endif
#endif /* WCT_HIP_H */
