// C ABI of libwct_hip.so (see include/wct_hip.h): context, weight upload, and the
// host-side orchestration of one WCT.predict (wct.py:70-106) as a chain of HIP
// kernel launches on a single stream.  Nothing here touches the CPU oracle.
#include "common.h"
#include "../../include/wct_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void wct_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* wct_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct ConvLayer {
  half_t* w = nullptr;   // [cout][9][cin]
  float* b = nullptr;
  half_t* ww = nullptr;  // the filters transformed for conv_wino.hip (ConvArgs::w_wino), or null: the direct kernel only
  int cin = 0, cout = 0;
};

struct PlanStep { char kind; int cin, cout, relu; };   // 'C' or 'U'

struct Decoder {
  bool loaded = false;
  std::vector<PlanStep> plan;
  std::vector<ConvLayer> convs;    // every conv but the last
  half_t* last_w = nullptr;        // MFMA A-fragments of the 64->3 conv (see ConvLastArgs::wfrag)
  float* last_b = nullptr;
  // training (wct_train_step): fp32 master copies (HWIO), Adam moments and the last step's gradients, one entry
  // per conv of the plan (the output conv last)
  std::vector<float*> w32, b32, mw, vw, mb, vb, gw, gb;
  std::vector<int> cin, cout;
  float* grad_arena = nullptr;     // gw / gb point into ONE contiguous buffer: a single all-reduce covers a decoder
  size_t grad_count = 0;
};

struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
constexpr int WCT_EIG_WORDS = 8 + 4 * 6 * 3;

struct wct_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side[3] = {nullptr, nullptr, nullptr};   // extra streams: each level's eigenproblems run in up to 4 groups
  int nside = 3;
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  bool enc_loaded = false;
  half_t* first_w = nullptr;       // folded conv1_1 as fp16 hi/lo MFMA fragments (ConvFirstArgs::wfrag)
  float* first_b = nullptr;
  ConvLayer enc[12];               // conv1_2 .. conv5_1
  float* first_w32 = nullptr;      // folded conv1_1 weights fp32 [27][64] (data gradient of the feature loss)
  float* enc_wt[12] = {nullptr};   // encoder weights transposed [(tap, cout)][cin] fp32 (B operand of the dgrad GEMM)
  DevBuf train_ws;
  Decoder dec[6];
  DevBuf act[2], feat_c, feat_s[6], img_c, img_s, img_t[2], wct_out, wct_ws, stage[4];
  DevBuf usum_c, usum_s[6], umax;     // feature statistics from the tap epilogues (ConvArgs::usum / umax); umax: [7][32][UMAX_SLOTS] words

  int* eig_fail = nullptr;         // pinned host memory mapped into the device: [4 stream groups][2] eigenproblems that did
                                   // (then [4 groups][6 size classes][3] solver statistics: matrices, sweeps, max sweeps)
  int* eig_fail_dev = nullptr;     // not converge / had non-finite input -- bumped by jacobi_finalize_kernel
  float ss_alpha = 0.6f;           // style-swap settings (stylize.py:34-37 defaults)
  int ss_patch = 3, ss_stride = 1;
  bool prof = false;
  bool no_wino = false;            // the training forward keeps every layer on the direct kernel (its backward assumes it)
  std::vector<ProfRec> recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
  double prof_ms[WCT_PROF_CLASSES] = {0};
  long long prof_n[WCT_PROF_CLASSES] = {0};
  double prof_flops[WCT_PROF_CLASSES] = {0};
  double prof_bytes[WCT_PROF_CLASSES] = {0};
};

static const int LEVEL_C[6] = {0, 64, 128, 256, 512, 512};
static const int ENC_CIN[12] = {64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512};
static const int ENC_COUT[12] = {64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512};

static int ensure(wct_ctx* c, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes) return WCT_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (b.p) HIP_TRY(hipFree(b.p));
  b.p = nullptr; b.cap = 0;
  size_t want = bytes + bytes / 8 + 4096;
  HIP_TRY(hipMalloc(&b.p, want));
  b.cap = want;
  return WCT_OK;
}

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

struct ProfScope {
  wct_ctx* c; ProfRec r; bool on;
  ProfScope(wct_ctx* ctx, int cls, double flops, double bytes) : c(ctx), on(ctx->prof) {
    if (!on) return;
    if (c->free_events.empty()) {
      hipEventCreate(&r.a); hipEventCreate(&r.b);
    } else {
      r.a = c->free_events.back().first; r.b = c->free_events.back().second;
      c->free_events.pop_back();
    }
    r.cls = cls; r.flops = flops; r.bytes = bytes;
    hipEventRecord(r.a, c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    hipEventRecord(r.b, c->stream);
    c->recs.push_back(r);
  }
};

extern "C" int wct_device_count(int* n) {
  ARG_CHECK(n != nullptr);
  HIP_TRY(hipGetDeviceCount(n));
  return WCT_OK;
}

extern "C" int wct_create(int device, wct_ctx** out) {
  ARG_CHECK(out != nullptr);
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (n <= 0 || device < 0 || device >= n) {
    wct_set_error("no such HIP device %d (count %d)", device, n);
    return WCT_ERR_ARG;
  }
  HIP_TRY(hipSetDevice(device));
  wct_ctx* c = new wct_ctx();
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    wct_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
    delete c;
    return WCT_ERR_HIP;
  }
  bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < 3 && ok; ++i)
    ok = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    wct_set_error("hipStreamCreate/hipEventCreate failed");
    delete c;
    return WCT_ERR_HIP;
  }
  if (ok) ok = hipHostMalloc((void**)&c->eig_fail, WCT_EIG_WORDS * sizeof(int), hipHostMallocMapped) == hipSuccess &&
               hipHostGetDevicePointer((void**)&c->eig_fail_dev, c->eig_fail, 0) == hipSuccess;
  if (!ok) {
    wct_set_error("hipHostMalloc (mapped status words) failed");
    delete c;
    return WCT_ERR_HIP;
  }
  for (int i = 0; i < WCT_EIG_WORDS; ++i) c->eig_fail[i] = 0;
  { const int n = tune_int("WCT_EIG_GROUPS", -1); if (n >= 0) c->nside = n < 1 ? 0 : (n > 4 ? 3 : n - 1); }
  *out = c;
  return WCT_OK;
}

static void free_layer(ConvLayer& l) {
  if (l.w) hipFree(l.w);
  if (l.b) hipFree(l.b);
  if (l.ww) hipFree(l.ww);
  l = ConvLayer();
}
static void free_decoder(Decoder& d) {
  for (auto& l : d.convs) free_layer(l);
  for (auto* v : {&d.w32, &d.b32, &d.mw, &d.vw, &d.mb, &d.vb})
    for (float* p : *v) if (p) hipFree(p);
  if (d.grad_arena) hipFree(d.grad_arena);
  if (d.last_w) hipFree(d.last_w);
  if (d.last_b) hipFree(d.last_b);
  d = Decoder();
}

extern "C" void wct_destroy(wct_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->first_w) hipFree(c->first_w);
  if (c->first_b) hipFree(c->first_b);
  if (c->first_w32) hipFree(c->first_w32);
  for (int i = 0; i < 12; ++i) if (c->enc_wt[i]) hipFree(c->enc_wt[i]);
  if (c->train_ws.p) hipFree(c->train_ws.p);
  for (auto& l : c->enc) free_layer(l);
  for (auto& d : c->dec) free_decoder(d);
  DevBuf* bufs[] = {&c->act[0], &c->act[1], &c->feat_c, &c->img_c, &c->img_s, &c->img_t[0], &c->img_t[1],
                    &c->wct_out, &c->wct_ws, &c->stage[0], &c->stage[1], &c->stage[2], &c->stage[3]};
  for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
  for (auto& b : c->feat_s) if (b.p) hipFree(b.p);
  for (auto& b : c->usum_s) if (b.p) hipFree(b.p);
  if (c->usum_c.p) hipFree(c->usum_c.p);
  if (c->umax.p) hipFree(c->umax.p);
  for (auto& r : c->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  for (auto& e : c->free_events) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  if (c->eig_fail) hipHostFree(c->eig_fail);
  hipEventDestroy(c->ev_fork);
  for (int i = 0; i < 3; ++i) { hipStreamSynchronize(c->side[i]); hipEventDestroy(c->ev_join[i]); hipStreamDestroy(c->side[i]); }
  hipStreamDestroy(c->stream);
  delete c;
}

// After a stream sync: did any eigensolve since the last check fail?  The frames / features of the call are still
// written (best effort, like LAPACK's "did not converge" info > 0), but the status is loud: WCT_STATUS_NOCONV.
static int eig_status(wct_ctx* c) {
  int n_open = 0, n_nan = 0;
  for (int g = 0; g < 4; ++g) { n_open += c->eig_fail[2 * g]; n_nan += c->eig_fail[2 * g + 1]; }
  if (!n_open && !n_nan) return WCT_OK;
  for (int i = 0; i < 8; ++i) c->eig_fail[i] = 0;
  wct_set_error("eigensolver: %d covariance matri%s still rotating after the sweep budget, %d with non-finite entries "
                "(NaN/Inf features?); the outputs of this call are not reliable", n_open, n_open == 1 ? "x" : "ces", n_nan);
  return WCT_ERR_NOCONV;
}

// At the START of a blocking call: failures still pending in the status words were raised by EARLIER asynchronous work
// (wct_stylize_batch_dev calls that nobody followed with wct_sync).  They are reported now -- loudly, but named for what
// they are, and before this call runs, so that its own status and sweep counts are never mixed with them.
static int eig_stale(wct_ctx* c) {
  HIP_TRY(hipStreamSynchronize(c->stream));
  int n_open = 0, n_nan = 0;
  for (int g = 0; g < 4; ++g) { n_open += c->eig_fail[2 * g]; n_nan += c->eig_fail[2 * g + 1]; }
  if (!n_open && !n_nan) return WCT_OK;
  for (int i = 0; i < 8; ++i) c->eig_fail[i] = 0;
  wct_set_error("eigensolver failures of an EARLIER asynchronous wct_stylize_batch_dev call that was not followed by wct_sync "
                "(%d matri%s still rotating after the sweep budget, %d with non-finite entries): the frames of that call are "
                "not reliable; the present call was not started", n_open, n_open == 1 ? "x" : "ces", n_nan);
  return WCT_ERR_NOCONV;
}

extern "C" int wct_sync(wct_ctx* c) {
  ARG_CHECK(c != nullptr);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return eig_status(c);
}

// Solver statistics since the last call (after a stream sync), per size class k (matrices of order 32 * 2^k, k = 0..5):
// out[3k] = matrices solved, out[3k+1] = sum of their sweeps, out[3k+2] = the largest sweep count.  Cleared on read.
extern "C" int wct_eig_stats(wct_ctx* c, long long out[18]) {
  ARG_CHECK(c && out);
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 18; ++k) out[k] = 0;
  for (int g = 0; g < 4; ++g)
    for (int k = 0; k < 6; ++k) {
      int* s = c->eig_fail + 8 + (g * 6 + k) * 3;
      out[3 * k] += s[0]; out[3 * k + 1] += s[1];
      if (s[2] > out[3 * k + 2]) out[3 * k + 2] = s[2];
      s[0] = s[1] = s[2] = 0;
    }
  return WCT_OK;
}

extern "C" int wct_get_stream(wct_ctx* c, void** stream_out) {
  ARG_CHECK(c && stream_out);
  *stream_out = (void*)c->stream;
  return WCT_OK;
}

extern "C" int wct_dev_alloc(wct_ctx* c, size_t bytes, void** out) {
  ARG_CHECK(c && out && bytes > 0);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMalloc(out, bytes));
  return WCT_OK;
}
extern "C" int wct_dev_free(wct_ctx* c, void* p) {
  ARG_CHECK(c != nullptr);
  if (p) HIP_TRY(hipFree(p));
  return WCT_OK;
}
extern "C" int wct_h2d(wct_ctx* c, void* dst, const void* src, size_t bytes) {
  ARG_CHECK(c && dst && src);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return WCT_OK;
}
extern "C" int wct_d2h(wct_ctx* c, void* dst, const void* src, size_t bytes) {
  ARG_CHECK(c && dst && src);
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
extern "C" int wct_prof_enable(wct_ctx* c, int on) {
  ARG_CHECK(c != nullptr);
  c->prof = on != 0;
  return WCT_OK;
}
static int prof_drain(wct_ctx* c) {
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (auto& r : c->recs) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
    c->prof_ms[r.cls] += ms;
    c->prof_n[r.cls] += 1;
    c->prof_flops[r.cls] += r.flops;
    c->prof_bytes[r.cls] += r.bytes;
    c->free_events.push_back({r.a, r.b});
  }
  c->recs.clear();
  return WCT_OK;
}
extern "C" int wct_prof_reset(wct_ctx* c) {
  ARG_CHECK(c != nullptr);
  TRY(prof_drain(c));
  for (int i = 0; i < WCT_PROF_CLASSES; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; c->prof_flops[i] = 0; c->prof_bytes[i] = 0; }
  return WCT_OK;
}
extern "C" int wct_prof_read(wct_ctx* c, double ms[WCT_PROF_CLASSES], long long launches[WCT_PROF_CLASSES],
                             double flops[WCT_PROF_CLASSES], double bytes[WCT_PROF_CLASSES]) {
  ARG_CHECK(c != nullptr);
  TRY(prof_drain(c));
  for (int i = 0; i < WCT_PROF_CLASSES; ++i) {
    if (ms) ms[i] = c->prof_ms[i];
    if (launches) launches[i] = c->prof_n[i];
    if (flops) flops[i] = c->prof_flops[i];
    if (bytes) bytes[i] = c->prof_bytes[i];
  }
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------
static int upload(wct_ctx* c, const void* host, size_t bytes, void** dev) {
  HIP_TRY(hipMalloc(dev, bytes));
  HIP_TRY(hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return WCT_OK;
}

// Which layers also get their filters packed for the reduced-FLOP kernel (conv_wino.hip).  Decided by the layer's channel counts
// alone -- never by the batch or the image size: a frame must not depend on the batch it is computed in.  WCT_WINOGRAD=0 is a TEST
// hook (read once per process, like WCT_FUSE_CONV1): every layer on the direct kernel, the round-5 bits.  A -DWCT_TUNING build
// takes the channel threshold from WCT_WINO_MIN.
static bool wino_layer(int cin, int cout) {
  static const int on = getenv("WCT_WINOGRAD") ? atoi(getenv("WCT_WINOGRAD")) : 1;
  static const int min_ch = tune_int("WCT_WINO_MIN", 256);
  return on && cin >= min_ch && cout >= min_ch && cin % 64 == 0 && cout % 64 == 0;
}

// HWIO fp32 -> fp16 MFMA A-fragments [cout/32][tap][cin/16][lane][8]: lane l of a fragment holds output
// channel 32*T + (l & 31), input channels 16*k16 + 8*(l >> 5) .. +7 (the v_mfma_f32_32x32x16_f16 A layout),
// so a wave fetches a fragment with ONE fully coalesced 1-KiB load
static int pack_conv(wct_ctx* c, const float* w_hwio, const float* b, int cin, int cout, ConvLayer* out) {
  free_layer(*out);
  std::vector<half_t> packed((size_t)cout * 9 * cin);
  const int c16 = cin / 16;
  for (int tap = 0; tap < 9; ++tap)
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < cout; ++co) {
        const int lane = (co & 31) + 32 * ((ci >> 3) & 1);
        const size_t frag = ((size_t)(co >> 5) * 9 + tap) * c16 + (ci >> 4);
        packed[(frag * 64 + lane) * 8 + (ci & 7)] = (half_t)w_hwio[((size_t)tap * cin + ci) * cout + co];
      }
  TRY(upload(c, packed.data(), packed.size() * sizeof(half_t), (void**)&out->w));
  TRY(upload(c, b, (size_t)cout * sizeof(float), (void**)&out->b));
  out->cin = cin; out->cout = cout;
  if (wino_layer(cin, cout)) {
    // U_f[kx] = sum_ky G[f][ky] g[ky][kx] (Winograd F(2,3) along y), in double, rounded to fp16 once; fragments
    // [cout/32][f*3+kx][cin/16][lane][8] like the direct ones
    static const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    std::vector<half_t> pw((size_t)cout * 12 * cin);
    for (int f = 0; f < 4; ++f)
      for (int kx = 0; kx < 3; ++kx)
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < cout; ++co) {
            double u = 0;
            for (int ky = 0; ky < 3; ++ky) u += G[f][ky] * (double)w_hwio[((size_t)(ky * 3 + kx) * cin + ci) * cout + co];
            const int lane = (co & 31) + 32 * ((ci >> 3) & 1);
            const size_t frag = ((size_t)(co >> 5) * 12 + f * 3 + kx) * c16 + (ci >> 4);
            pw[(frag * 64 + lane) * 8 + (ci & 7)] = (half_t)(float)u;
          }
    TRY(upload(c, pw.data(), pw.size() * sizeof(half_t), (void**)&out->ww));
  }
  return WCT_OK;
}

extern "C" int wct_set_encoder(wct_ctx* c, const float* pre_w, const float* pre_b,
                               const float* const* w, const float* const* b, int n_layers) {
  ARG_CHECK(c && pre_w && pre_b && w && b && n_layers == 13);
  HIP_TRY(hipSetDevice(c->device));
  // fold the pointwise 'preprocess' conv (vgg_normalised.py:25-26) into conv1_1: a pointwise
  // op commutes with the reflect pad, so  conv1_1(pad(P x + pb)) = conv1_1'(pad(x)) exactly.
  std::vector<float> fw(27 * 64), fb(64);
  const float* w1 = w[0];   // [3][3][3][64]
  for (int co = 0; co < 64; ++co) {
    double acc = b[0][co];
    for (int tap = 0; tap < 9; ++tap)
      for (int cp = 0; cp < 3; ++cp) acc += (double)pre_b[cp] * w1[(tap * 3 + cp) * 64 + co];
    fb[co] = (float)acc;
  }
  for (int tap = 0; tap < 9; ++tap)
    for (int ci = 0; ci < 3; ++ci)
      for (int co = 0; co < 64; ++co) {
        double acc = 0;
        for (int cp = 0; cp < 3; ++cp) acc += (double)pre_w[ci * 3 + cp] * w1[(tap * 3 + cp) * 64 + co];
        fw[(tap * 3 + ci) * 64 + co] = (float)acc;
      }
  if (c->first_w) { hipFree(c->first_w); c->first_w = nullptr; }
  if (c->first_b) { hipFree(c->first_b); c->first_b = nullptr; }
  // hi/lo split fragments: lane l of fragment (t, ks) holds channel 32t + (l&31), k = 16ks + 8(l>>5) .. +7
  std::vector<half_t> frag(2 * 2 * 2 * 64 * 8);
  for (int t = 0; t < 2; ++t)
    for (int ks = 0; ks < 2; ++ks)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int co = t * 32 + (l & 31), k = ks * 16 + (l >> 5) * 8 + j;
          const float wv = k < 27 ? fw[k * 64 + co] : 0.f;
          const half_t hi = (half_t)wv;
          const half_t lo = (half_t)(wv - (float)hi);
          frag[((((t * 2 + ks) * 2 + 0) * 64) + l) * 8 + j] = hi;
          frag[((((t * 2 + ks) * 2 + 1) * 64) + l) * 8 + j] = lo;
        }
  TRY(upload(c, frag.data(), frag.size() * sizeof(half_t), (void**)&c->first_w));
  TRY(upload(c, fb.data(), fb.size() * sizeof(float), (void**)&c->first_b));
  if (c->first_w32) { hipFree(c->first_w32); c->first_w32 = nullptr; }
  TRY(upload(c, fw.data(), fw.size() * sizeof(float), (void**)&c->first_w32));
  for (int i = 0; i < 12; ++i) {
    TRY(pack_conv(c, w[i + 1], b[i + 1], ENC_CIN[i], ENC_COUT[i], &c->enc[i]));
    const int cin = ENC_CIN[i], cout = ENC_COUT[i];
    std::vector<float> wt((size_t)9 * cin * cout);
    for (int tap = 0; tap < 9; ++tap)
      for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co) wt[((size_t)tap * cout + co) * cin + ci] = w[i + 1][((size_t)tap * cin + ci) * cout + co];
    if (c->enc_wt[i]) { hipFree(c->enc_wt[i]); c->enc_wt[i] = nullptr; }
    TRY(upload(c, wt.data(), wt.size() * sizeof(float), (void**)&c->enc_wt[i]));
  }
  c->enc_loaded = true;
  return WCT_OK;
}

static std::vector<PlanStep> decoder_plan(int level) {
  // model.py:255-277 walked from `level` down to 1, then the 3-filter output conv (model.py:298)
  static const int arch5[] = {512, -1, 512, 512, 512, 0};
  static const int arch4[] = {256, -1, 256, 256, 256, 0};
  static const int arch3[] = {128, -1, 128, 0};
  static const int arch2[] = {64, -1, 0};
  static const int arch1[] = {64, 0};
  static const int* archs[6] = {nullptr, arch1, arch2, arch3, arch4, arch5};
  std::vector<PlanStep> plan;
  int cin = LEVEL_C[level];
  for (int d = level; d >= 1; --d)
    for (const int* a = archs[d]; *a != 0; ++a) {
      if (*a < 0) plan.push_back({'U', cin, cin, 0});
      else { plan.push_back({'C', cin, *a, 1}); cin = *a; }
    }
  plan.push_back({'C', cin, 3, 0});
  return plan;
}

extern "C" int wct_set_decoder(wct_ctx* c, int level, const float* const* w, const float* const* b, int n_layers) {
  ARG_CHECK(c && w && b && level >= 1 && level <= 5);
  HIP_TRY(hipSetDevice(c->device));
  Decoder& d = c->dec[level];
  free_decoder(d);
  d.plan = decoder_plan(level);
  int nconv = 0;
  for (auto& s : d.plan) nconv += s.kind == 'C';
  if (n_layers != nconv) {
    wct_set_error("decoder relu%d_1 needs %d conv layers, got %d", level, nconv, n_layers);
    return WCT_ERR_ARG;
  }
  int i = 0;
  for (auto& s : d.plan) {
    if (s.kind != 'C') continue;
    if (s.cout == 3) {
      // HWIO [3][3][64][3] = [(tap*64+cin)][3] -> A fragments of the 32 x 64 matrix with row tap*3+cout:
      // lane l of k-step ks holds row l&31, cin = 16 ks + 8 (l>>5) .. +7
      std::vector<half_t> w16((size_t)4 * 64 * 8);
      for (int ks = 0; ks < 4; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) {
            const int row = l & 31, cin = ks * 16 + (l >> 5) * 8 + j;
            const int tap = row / 3, co = row % 3;
            w16[((size_t)ks * 64 + l) * 8 + j] = row < 27 ? (half_t)w[i][((size_t)tap * 64 + cin) * 3 + co] : (half_t)0.f;
          }
      TRY(upload(c, w16.data(), w16.size() * sizeof(half_t), (void**)&d.last_w));
      TRY(upload(c, b[i], 3 * sizeof(float), (void**)&d.last_b));
    } else {
      d.convs.emplace_back();
      TRY(pack_conv(c, w[i], b[i], s.cin, s.cout, &d.convs.back()));
    }
    {
      float *w32 = nullptr, *b32 = nullptr;
      TRY(upload(c, w[i], (size_t)9 * s.cin * s.cout * sizeof(float), (void**)&w32));
      TRY(upload(c, b[i], (size_t)s.cout * sizeof(float), (void**)&b32));
      d.w32.push_back(w32); d.b32.push_back(b32); d.cin.push_back(s.cin); d.cout.push_back(s.cout);
    }
    ++i;
  }
  d.loaded = true;
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// encoder / decoder drivers (device pointers, batch B)
// ---------------------------------------------------------------------------
static void level_dims(int H, int W, int level, int* h, int* w) {
  for (int l = 1; l < level; ++l) { H = (H + 1) / 2; W = (W + 1) / 2; }   // 'same' pooling = ceil
  *h = H; *w = W;
}

// every 3x3 conv reflect-pads its input by one pixel, which needs a map of at least 2x2 (tf.pad REFLECT refuses a
// padding >= the dimension just the same): the feature map of the deepest level must still be 2x2
static int check_min_size(const char* what, int H, int W, int level) {
  int h, w;
  level_dims(H, W, level, &h, &w);
  if (h < 2 || w < 2) {
    wct_set_error("%s %dx%d is too small for relu%d_1: its %dx%d feature map cannot be reflect-padded (need >= %dx%d pixels)",
                  what, H, W, level, h, w, (1 << (level - 1)) + 1, (1 << (level - 1)) + 1);
    return WCT_ERR_ARG;
  }
  return WCT_OK;
}


static int run_conv(wct_ctx* c, const ConvLayer& l, const half_t* x, half_t* y16, float* y32,
                    int B, int H, int W, int upsample, int relu, int pool = 0, float* usum = nullptr, unsigned* umax = nullptr,
                    bool tap_layer = false) {
  ConvArgs a;
  a.x = x; a.w = l.w; a.bias = l.b; a.y16 = y16; a.y32 = y32; a.usum = usum; a.umax = umax;
  // a layer whose output CAN be tapped (conv2_1 .. conv5_1) stays on the direct kernel in every pass, tapped or not: the kernel
  // of a layer must not depend on which levels the caller asked for (fused == chained, bit for bit)
  a.w_wino = c->no_wino || tap_layer ? nullptr : l.ww;
  a.B = B; a.H = H; a.W = W; a.Cin = l.cin; a.Cout = l.cout; a.upsample = upsample; a.relu = relu; a.pool = pool;
  const double px = (double)B * H * W;
  const double in_px = upsample ? px / 4 : px;
  const double out_px = pool ? (double)B * ((H + 1) / 2) * ((W + 1) / 2) : px;
  // class 9: the launches the reduced-FLOP kernel takes; flops = the DIRECT convolution's (the kernel executes 2/3 of them)
  ProfScope ps(c, conv3x3_wino_takes(a) ? 9 : 0, 2.0 * px * 9 * l.cin * l.cout,
               in_px * l.cin * 2 + out_px * l.cout * ((y16 ? 2 : 0) + (y32 ? 4 : 0)) + 9.0 * l.cin * l.cout * 2);
  return launch_conv3x3(a, c->stream);
}

// img: [B][H][W][3] fp32 device.  taps32[l] (l=1..5): fp32 feature output for relu<l>_1 or null.
// usum / umax (optional, per level like taps32): statistics of the tap, taken by the epilogue that writes it
static int run_encoder(wct_ctx* c, const float* img, int B, int H, int W, int clamp01, int deepest,
                       float* const taps32[6], float* const* usum = nullptr, unsigned* const* umax = nullptr) {
  if (!c->enc_loaded) { wct_set_error("encoder weights not set (wct_set_encoder)"); return WCT_ERR_STATE; }
  ARG_CHECK(deepest >= 1 && deepest <= 5 && H >= 2 && W >= 2);
  const size_t act_bytes = (size_t)B * H * W * 64 * sizeof(half_t);
  TRY(ensure(c, c->act[0], act_bytes));
  TRY(ensure(c, c->act[1], act_bytes));
  half_t* cur = (half_t*)c->act[0].p;
  half_t* nxt = (half_t*)c->act[1].p;
  // relu1_1 is read by conv1_2 only (no tap on it: every content pass of a level >= 2): conv1_1 moves into conv1_2's patch
  // loader (ConvArgs::img1) -- the 64-channel full-resolution map is neither written nor read, and the bits are the same.
  // (WCT_FUSE_CONV1=0: the two launches; test hook, read once per process)
  static const int fuse_conv1_env = getenv("WCT_FUSE_CONV1") ? atoi(getenv("WCT_FUSE_CONV1")) : 1;
  const bool fuse_conv1 = fuse_conv1_env && deepest > 1 && !taps32[1];
  if (!fuse_conv1) {
    ConvFirstArgs a;
    a.x = img; a.wfrag = c->first_w; a.bias = c->first_b;
    a.y16 = deepest > 1 ? cur : nullptr; a.y32 = taps32[1];
    a.B = B; a.H = H; a.W = W; a.clamp01 = clamp01;
    a.usum = usum && taps32[1] ? usum[1] : nullptr; a.umax = a.usum ? umax[1] : nullptr;
    const double px = (double)B * H * W;
    ProfScope ps(c, 1, 2.0 * px * 27 * 64, px * (12 + 64 * ((a.y16 ? 2 : 0) + (a.y32 ? 4 : 0))));
    TRY(launch_conv_first(a, c->stream));
  }
  if (deepest == 1) return WCT_OK;
  // enc[i]: level whose relu*_1 it produces (0 = none), and whether a 'same' max-pool follows it
  // (conv1_2, conv2_2, conv3_4, conv4_4 -- their outputs feed nothing but the pool, so it is fused
  // into their epilogue; WCT_FUSE_POOL=0 runs the separate pool kernel instead)
  static const int seq_tap[12] = {0, 2, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5};
  static const int pool_after[12] = {1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0};
  static const int fuse_pool = tune_int("WCT_FUSE_POOL", 1);
  int h = H, w = W;
  for (int i = 0; i < 12; ++i) {
    const ConvLayer& l = c->enc[i];
    const int tap = seq_tap[i];
    const bool last = tap == deepest;
    const bool fuse = pool_after[i] && fuse_pool;
    float* const us = tap && taps32[tap] && usum ? usum[tap] : nullptr;
    if (i == 0 && fuse_conv1) {
      ConvArgs a;
      a.x = nullptr; a.w = l.w; a.bias = l.b; a.y16 = nxt; a.y32 = nullptr; a.usum = nullptr; a.umax = nullptr;
      a.B = B; a.H = h; a.W = w; a.Cin = 64; a.Cout = 64; a.upsample = 0; a.relu = 1; a.pool = fuse;
      a.img1 = img; a.w1frag = c->first_w; a.bias1 = c->first_b; a.clamp01 = clamp01;
      const double px = (double)B * h * w, out_px = fuse ? (double)B * ((h + 1) / 2) * ((w + 1) / 2) : px;
      ProfScope ps(c, 8, 2.0 * px * (27 * 64 + 9 * 64 * 64), px * 12 + out_px * 64 * 2 + 9.0 * 64 * 64 * 2);
      TRY(launch_conv3x3(a, c->stream));
    } else
    TRY(run_conv(c, l, cur, last ? nullptr : nxt, tap ? taps32[tap] : nullptr, B, h, w, 0, 1, fuse, us, us ? umax[tap] : nullptr, tap != 0));
    half_t* t = cur; cur = nxt; nxt = t;
    if (last) break;
    if (pool_after[i]) {
      if (!fuse) {
        ProfScope ps(c, 3, 0, (double)B * h * w * l.cout * 2 * 1.25);
        TRY(launch_maxpool2x2(cur, nxt, B, h, w, l.cout, c->stream));
        t = cur; cur = nxt; nxt = t;
      }
      h = (h + 1) / 2; w = (w + 1) / 2;
    }
  }
  return WCT_OK;
}

// feat16: [B][h][w][C] fp16 device (may alias neither act buffer). img_out: [B][h*2^(l-1)][..][3] fp32
static int run_decoder(wct_ctx* c, int level, const half_t* feat16, int B, int h, int w, float* img_out) {
  Decoder& d = c->dec[level];
  if (!d.loaded) { wct_set_error("decoder weights for relu%d_1 not set (wct_set_decoder)", level); return WCT_ERR_STATE; }
  const int scale = 1 << (level - 1);
  const size_t act_bytes = (size_t)B * h * scale * w * scale * 64 * sizeof(half_t);
  const size_t deep_bytes = (size_t)B * h * w * LEVEL_C[level] * sizeof(half_t) * 4;   // after the first upsample, <= 4x
  TRY(ensure(c, c->act[0], act_bytes > deep_bytes ? act_bytes : deep_bytes));
  TRY(ensure(c, c->act[1], act_bytes > deep_bytes ? act_bytes : deep_bytes));
  const half_t* cur = feat16;
  half_t* bufs[2] = {(half_t*)c->act[0].p, (half_t*)c->act[1].p};
  int which = 0, ci = 0, up = 0;
  // WCT_FUSE_TAIL=1 (test hook, read once per process): the last 64 -> 64 conv and the 64 -> 3 output conv as ONE launch
  // (csrc/conv_tail.hip): the 64-channel full-resolution map is neither written nor read, and the bits are the same.  NOT the
  // default: measured on one box (profiles/r06_conv_tail.txt) the fused launch takes what the two launches take (+0.25 ms per
  // 32-pair step) -- the 64 -> 64 layer is bound by MFMA work and per-tile fixed costs, not by the bytes the fusion removes, and
  // the 18 x 18 halo costs 1.37 x its MFMA work.
  static const int fuse_tail_env = getenv("WCT_FUSE_TAIL") ? atoi(getenv("WCT_FUSE_TAIL")) : 0;
  const size_t nsteps = d.plan.size();
  for (size_t si = 0; si < nsteps; ++si) {
    const PlanStep& s = d.plan[si];
    if (s.kind == 'U') { up = 1; h *= 2; w *= 2; continue; }
    if (fuse_tail_env && !c->no_wino /* (the training forward keeps every activation) */ && s.cin == 64 && s.cout == 64 &&
        si + 1 < nsteps && d.plan[si + 1].kind == 'C' && d.plan[si + 1].cout == 3) {
      const ConvLayer& l = d.convs[ci++];
      ConvTailArgs a;
      a.x = cur; a.w = l.w; a.bias = l.b; a.wlast = d.last_w; a.blast = d.last_b; a.y = img_out; a.B = B; a.H = h; a.W = w; a.upsample = up;
      const double px = (double)B * h * w;
      ProfScope ps(c, 10, 2.0 * px * (9 * 64 * 64 + 576 * 3), px * ((up ? 32 : 128) + 12) + 9.0 * 64 * 64 * 2);
      TRY(launch_conv_tail(a, c->stream));
      break;
    }
    if (s.cout == 3) {
      ARG_CHECK(up == 0);
      ConvLastArgs a;
      a.x = cur; a.wfrag = d.last_w; a.bias = d.last_b; a.y = img_out; a.B = B; a.H = h; a.W = w;
      const double px = (double)B * h * w;
      ProfScope ps(c, 2, 2.0 * px * 576 * 3, px * (128 + 12));
      TRY(launch_conv_last(a, c->stream));
    } else {
      half_t* out = bufs[which];
      TRY(run_conv(c, d.convs[ci++], cur, out, nullptr, B, h, w, up, 1));
      cur = out; which ^= 1; up = 0;
    }
  }
  return WCT_OK;
}

static int run_transform(wct_ctx* c, const float* fc, int Nc, const float* fs, int Ns, int C, int P,
                         float alpha, unsigned flags, float eps, half_t* out16, float* out32, int* sweeps_dev,
                         const WctFeatStats* st = nullptr) {
  const int shared = (flags & WCT_FLAG_STYLE_SHARED) ? 1 : 0;
  const size_t ws = wct_workspace_bytes(C, Nc, Ns, P);
  TRY(ensure(c, c->wct_ws, ws));
  if (flags & WCT_FLAG_ADAIN) {
    ProfScope ps(c, 7, 0, (double)P * (2.0 * Nc + 2.0 * Ns) * C * 4 + (double)P * Nc * C * 6);
    return launch_adain(fc, Nc, fs, Ns, C, P, alpha, 1e-5f, out16, out32, c->wct_ws.p, c->wct_ws.cap, c->stream, shared, st);
  }
  const int mode = (flags & WCT_FLAG_MODE_NP) ? WCT_MODE_NP : WCT_MODE_TF;
  {
    ProfScope ps(c, 4, (double)P * 2.0 * C * C * ((double)Nc + Ns), (double)P * 2.0 * ((double)Nc + Ns) * C * 4);
    TRY(launch_wct(fc, Nc, fs, Ns, C, P, alpha, mode, eps, out16, out32, c->wct_ws.p, c->wct_ws.cap, sweeps_dev, WCT_STAGE_COV, c->stream, nullptr, 0, nullptr, nullptr, shared, c->eig_fail_dev, st));
  }
  {
    ProfScope ps(c, 5, 0, 0);
    TRY(launch_wct(fc, Nc, fs, Ns, C, P, alpha, mode, eps, out16, out32, c->wct_ws.p, c->wct_ws.cap, sweeps_dev, WCT_STAGE_EIG, c->stream, c->side, c->nside, c->ev_fork, c->ev_join, shared, c->eig_fail_dev));
  }
  ProfScope ps(c, 6, (double)P * (2.0 * C * C * Nc + 6.0 * C * C * C), (double)P * Nc * C * (4 + (out16 ? 2 : 0) + (out32 ? 4 : 0)));
  return launch_wct(fc, Nc, fs, Ns, C, P, alpha, mode, eps, out16, out32, c->wct_ws.p, c->wct_ws.cap, sweeps_dev, WCT_STAGE_APPLY, c->stream, nullptr, 0, nullptr, nullptr, shared, c->eig_fail_dev);
}

// ---------------------------------------------------------------------------
// op-level host entry points
// ---------------------------------------------------------------------------
static int stage_in(wct_ctx* c, int slot, const void* host, size_t bytes, void** dev) {
  TRY(ensure(c, c->stage[slot], bytes));
  HIP_TRY(hipMemcpyAsync(c->stage[slot].p, host, bytes, hipMemcpyHostToDevice, c->stream));
  *dev = c->stage[slot].p;
  return WCT_OK;
}
static int fetch(wct_ctx* c, void* host, const void* dev, size_t bytes) {
  HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return WCT_OK;
}

extern "C" int wct_transform(wct_ctx* c, const float* content, int Nc, const float* style, int Ns, int C,
                             float alpha, int mode, float eps, float* out, int* sweeps_out) {
  ARG_CHECK(c && content && style && out && (mode == WCT_NP || mode == WCT_TF));
  HIP_TRY(hipSetDevice(c->device));
  TRY(eig_stale(c));
  void *dc, *ds;
  TRY(stage_in(c, 0, content, (size_t)Nc * C * 4, &dc));
  TRY(stage_in(c, 1, style, (size_t)Ns * C * 4, &ds));
  TRY(ensure(c, c->stage[2], (size_t)Nc * C * 4));
  TRY(ensure(c, c->stage[3], 256));
  TRY(run_transform(c, (float*)dc, Nc, (float*)ds, Ns, C, 1, alpha, mode == WCT_NP ? WCT_FLAG_MODE_NP : 0, eps,
                    nullptr, (float*)c->stage[2].p, (int*)c->stage[3].p));
  TRY(fetch(c, out, c->stage[2].p, (size_t)Nc * C * 4));
  if (sweeps_out) TRY(fetch(c, sweeps_out, c->stage[3].p, 2 * sizeof(int)));
  return eig_status(c);
}

extern "C" int wct_adain(wct_ctx* c, const float* content, int Nc, const float* style, int Ns, int C,
                         float alpha, float epsilon, float* out) {
  ARG_CHECK(c && content && style && out);
  HIP_TRY(hipSetDevice(c->device));
  void *dc, *ds;
  TRY(stage_in(c, 0, content, (size_t)Nc * C * 4, &dc));
  TRY(stage_in(c, 1, style, (size_t)Ns * C * 4, &ds));
  TRY(ensure(c, c->stage[2], (size_t)Nc * C * 4));
  TRY(ensure(c, c->wct_ws, wct_workspace_bytes(C, Nc, Ns, 1)));
  TRY(launch_adain((float*)dc, Nc, (float*)ds, Ns, C, 1, alpha, epsilon, nullptr, (float*)c->stage[2].p,
                   c->wct_ws.p, c->wct_ws.cap, c->stream, 0));
  return fetch(c, out, c->stage[2].p, (size_t)Nc * C * 4);
}

extern "C" int wct_set_style_swap(wct_ctx* c, float ss_alpha, int patch_size, int stride) {
  ARG_CHECK(c && patch_size >= 1 && stride >= 1);
  c->ss_alpha = ss_alpha; c->ss_patch = patch_size; c->ss_stride = stride;
  return WCT_OK;
}

extern "C" int wct_style_swap(wct_ctx* c, const float* content, int hc, int wc, const float* style, int hs, int ws,
                              int C, float alpha, int patch_size, int stride, float eps, float* out) {
  ARG_CHECK(c && content && style && out && hc > 0 && wc > 0 && hs > 0 && ws > 0);
  HIP_TRY(hipSetDevice(c->device));
  TRY(eig_stale(c));
  void *dc, *ds;
  TRY(stage_in(c, 0, content, (size_t)hc * wc * C * 4, &dc));
  TRY(stage_in(c, 1, style, (size_t)hs * ws * C * 4, &ds));
  TRY(ensure(c, c->stage[2], (size_t)hc * wc * C * 4));
  ARG_CHECK(patch_size >= 1 && stride >= 1 && hc >= patch_size && wc >= patch_size && hs >= patch_size && ws >= patch_size);
  TRY(ensure(c, c->wct_ws, style_swap_workspace_bytes(C, hc, wc, hs, ws, patch_size, stride)));
  TRY(launch_style_swap((float*)dc, hc, wc, (float*)ds, hs, ws, C, alpha, patch_size, stride, eps, nullptr,
                        (float*)c->stage[2].p, c->wct_ws.p, c->wct_ws.cap, c->stream, c->eig_fail_dev));
  TRY(fetch(c, out, c->stage[2].p, (size_t)hc * wc * C * 4));
  return eig_status(c);
}

__global__ void extract_diag_kernel(const float* A, float* d, int C) {
  const int m = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < C) d[m * C + k] = A[(size_t)m * C * C + (size_t)k * C + k];
}

// wct_eigh's input contract: the UPPER triangle is authoritative.  The solver reads an element from whichever triangle is
// contiguous for the kernel at hand, so the staged copy is mirrored first (a C caller may pass upper-only data or a matrix
// that is symmetric only to round-off -- ADVICE r4).  In tiles: coalesced rows in, transposed through LDS out.
__global__ __launch_bounds__(256) void mirror_upper_kernel(float* A, int C) {
  __shared__ float t[32][33];
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj < bi) return;                                     // tile (bi, bj) of the upper triangle -> tile (bj, bi)
  float* Am = A + (size_t)blockIdx.z * C * C;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int i = bi * 32 + r, j = bj * 32 + tx;
    t[r][tx] = (i < C && j < C) ? Am[(size_t)i * C + j] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = bj * 32 + r, j = bi * 32 + tx;           // element (i, j) of the lower side = upper (j, i)
    if (i < C && j < C && i > j) Am[(size_t)i * C + j] = t[tx][r];
  }
}

extern "C" int wct_eigh(wct_ctx* c, const float* A, int C, int nmat, float* evals, float* evecs, int* sweeps_out) {
  ARG_CHECK(c && A && evals && evecs && nmat >= 1 && nmat <= 64);
  HIP_TRY(hipSetDevice(c->device));
  TRY(eig_stale(c));
  const size_t mb = (size_t)nmat * C * C * 4;
  void* dA;
  TRY(stage_in(c, 0, A, mb, &dA));
  hipLaunchKernelGGL(mirror_upper_kernel, dim3(cdiv(C, 32), cdiv(C, 32), nmat), dim3(256), 0, c->stream, (float*)dA, C);
  TRY(ensure(c, c->stage[1], mb));
  TRY(ensure(c, c->stage[2], (size_t)nmat * C * 4 + 1024));
  TRY(ensure(c, c->wct_ws, jacobi_workspace_bytes(C, nmat)));
  int* sw = (int*)((char*)c->stage[2].p + (size_t)nmat * C * 4);
  {
    ProfScope ps(c, 5, 0, 0);
    TRY(launch_jacobi_eigh((float*)dA, (float*)c->stage[1].p, C, nmat, c->wct_ws.p, c->wct_ws.cap, sw, c->eig_fail_dev, c->stream));
  }
  hipLaunchKernelGGL(extract_diag_kernel, dim3(cdiv(C, 256), nmat), dim3(256), 0, c->stream, (float*)dA, (float*)c->stage[2].p, C);
  TRY(fetch(c, evals, c->stage[2].p, (size_t)nmat * C * 4));
  TRY(fetch(c, evecs, c->stage[1].p, mb));
  if (sweeps_out) TRY(fetch(c, sweeps_out, sw, nmat * sizeof(int)));
  return eig_status(c);
}

extern "C" int wct_conv3x3(wct_ctx* c, const float* x, int H, int W, int Cin, const float* w_hwio, const float* bias,
                           int Cout, int relu, int upsample, float* y) {
  ARG_CHECK(c && x && w_hwio && bias && y);
  HIP_TRY(hipSetDevice(c->device));
  const int Ho = upsample ? 2 * H : H, Wo = upsample ? 2 * W : W;
  ConvLayer l;
  TRY(pack_conv(c, w_hwio, bias, Cin, Cout, &l));
  void* dx;
  int rc = stage_in(c, 0, x, (size_t)H * W * Cin * 4, &dx);
  if (!rc) rc = ensure(c, c->stage[1], (size_t)H * W * Cin * 2);
  if (!rc) rc = ensure(c, c->stage[2], (size_t)Ho * Wo * Cout * 4);
  if (!rc) rc = launch_f32_to_f16((float*)dx, (half_t*)c->stage[1].p, (size_t)H * W * Cin, c->stream);
  if (!rc) rc = run_conv(c, l, (half_t*)c->stage[1].p, nullptr, (float*)c->stage[2].p, 1, Ho, Wo, upsample, relu);
  if (!rc) rc = fetch(c, y, c->stage[2].p, (size_t)Ho * Wo * Cout * 4);
  hipStreamSynchronize(c->stream);
  free_layer(l);
  return rc;
}

// One layer as the PIPELINE runs it: fp16 activations in, fp16 out (returned as fp32), optionally with the fused 2x2 'same'
// max-pool.  algo 0: the pipeline's choice for this layer shape, 1: the direct kernel, 2: the reduced-FLOP kernel (conv_wino.hip;
// an error if the shape is not one it takes).  x: [B][H][W][Cin] fp32 (rounded to fp16 on the device like wct_conv3x3).
extern "C" int wct_conv3x3_f16(wct_ctx* c, const float* x, int B, int H, int W, int Cin, const float* w_hwio, const float* bias,
                               int Cout, int relu, int upsample, int pool, int algo, float* y) {
  ARG_CHECK(c && x && w_hwio && bias && y && B >= 1 && algo >= 0 && algo <= 2);
  HIP_TRY(hipSetDevice(c->device));
  const int Hc = upsample ? 2 * H : H, Wc = upsample ? 2 * W : W;
  const int Ho = pool ? (Hc + 1) / 2 : Hc, Wo = pool ? (Wc + 1) / 2 : Wc;
  ConvLayer l;
  TRY(pack_conv(c, w_hwio, bias, Cin, Cout, &l));
  int rc = WCT_OK;
  if (algo == 2 && !l.ww) {
    // a shape the policy leaves on the direct kernel: pack its Winograd fragments on the device for this call
    float* w32 = nullptr;
    rc = upload(c, w_hwio, (size_t)9 * Cin * Cout * sizeof(float), (void**)&w32);
    if (!rc && hipMalloc((void**)&l.ww, (size_t)Cout * 12 * Cin * sizeof(half_t)) != hipSuccess) { wct_set_error("hipMalloc failed"); rc = WCT_ERR_NOMEM; }
    if (!rc) rc = launch_pack_conv_wino_frag(w32, l.ww, Cin, Cout, c->stream);
    hipStreamSynchronize(c->stream);
    if (w32) hipFree(w32);
  }
  const bool keep = c->no_wino;
  c->no_wino = algo == 1;
  void* dx;
  const size_t nin = (size_t)B * H * W * Cin, nout = (size_t)B * Ho * Wo * Cout;
  if (!rc) rc = stage_in(c, 0, x, nin * 4, &dx);
  if (!rc) rc = ensure(c, c->stage[1], nin * 2);
  if (!rc) rc = ensure(c, c->stage[2], nout * 2);
  if (!rc) rc = ensure(c, c->stage[3], nout * 4);
  if (!rc) rc = launch_f32_to_f16((float*)dx, (half_t*)c->stage[1].p, nin, c->stream);
  if (!rc) {
    ConvArgs a;
    a.x = (half_t*)c->stage[1].p; a.w = l.w; a.bias = l.b; a.y16 = (half_t*)c->stage[2].p; a.y32 = nullptr; a.usum = nullptr; a.umax = nullptr;
    a.B = B; a.H = Hc; a.W = Wc; a.Cin = Cin; a.Cout = Cout; a.upsample = upsample; a.relu = relu; a.pool = pool;
    a.w_wino = c->no_wino ? nullptr : l.ww;
    if (algo == 2 && !conv3x3_wino_takes(a)) { wct_set_error("wct_conv3x3_f16: algo 2 does not take this layer"); rc = WCT_ERR_ARG; }
    if (!rc) {
      const double px = (double)B * Hc * Wc;
      ProfScope ps(c, 0, 2.0 * px * 9 * Cin * Cout, (double)nin * 2 + (double)nout * 2 + 9.0 * Cin * Cout * 2);
      rc = launch_conv3x3(a, c->stream);
    }
  }
  c->no_wino = keep;
  if (!rc) rc = launch_f16_to_f32((half_t*)c->stage[2].p, (float*)c->stage[3].p, nout, c->stream);
  if (!rc) rc = fetch(c, y, c->stage[3].p, nout * 4);
  hipStreamSynchronize(c->stream);
  free_layer(l);
  return rc;
}

extern "C" int wct_maxpool(wct_ctx* c, const float* x, int H, int W, int C, float* y) {
  ARG_CHECK(c && x && y && C % 8 == 0);
  HIP_TRY(hipSetDevice(c->device));
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  void* dx;
  TRY(stage_in(c, 0, x, (size_t)H * W * C * 4, &dx));
  TRY(ensure(c, c->stage[1], (size_t)H * W * C * 2));
  TRY(ensure(c, c->stage[2], (size_t)Ho * Wo * C * 2));
  TRY(ensure(c, c->stage[3], (size_t)Ho * Wo * C * 4));
  TRY(launch_f32_to_f16((float*)dx, (half_t*)c->stage[1].p, (size_t)H * W * C, c->stream));
  TRY(launch_maxpool2x2((half_t*)c->stage[1].p, (half_t*)c->stage[2].p, 1, H, W, C, c->stream));
  TRY(launch_f16_to_f32((half_t*)c->stage[2].p, (float*)c->stage[3].p, (size_t)Ho * Wo * C, c->stream));
  return fetch(c, y, c->stage[3].p, (size_t)Ho * Wo * C * 4);
}

extern "C" int wct_encode(wct_ctx* c, const float* img01, int H, int W, int level, float* feat) {
  ARG_CHECK(c && img01 && feat && level >= 1 && level <= 5);
  HIP_TRY(hipSetDevice(c->device));
  TRY(check_min_size("image", H, W, level));
  int h, w;
  level_dims(H, W, level, &h, &w);
  void* dimg;
  TRY(stage_in(c, 0, img01, (size_t)H * W * 3 * 4, &dimg));
  const size_t fb = (size_t)h * w * LEVEL_C[level] * 4;
  TRY(ensure(c, c->feat_c, fb));
  float* taps[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  taps[level] = (float*)c->feat_c.p;
  TRY(run_encoder(c, (float*)dimg, 1, H, W, 0, level, taps));
  return fetch(c, feat, c->feat_c.p, fb);
}

extern "C" int wct_decode(wct_ctx* c, const float* feat, int h, int w, int level, float* img) {
  ARG_CHECK(c && feat && img && level >= 1 && level <= 5);
  HIP_TRY(hipSetDevice(c->device));
  const int C = LEVEL_C[level], scale = 1 << (level - 1);
  void* df;
  TRY(stage_in(c, 0, feat, (size_t)h * w * C * 4, &df));
  TRY(ensure(c, c->wct_out, (size_t)h * w * C * 2));
  TRY(launch_f32_to_f16((float*)df, (half_t*)c->wct_out.p, (size_t)h * w * C, c->stream));
  const size_t ib = (size_t)h * scale * w * scale * 3 * 4;
  TRY(ensure(c, c->img_t[0], ib));
  TRY(run_decoder(c, level, (half_t*)c->wct_out.p, 1, h, w, (float*)c->img_t[0].p));
  return fetch(c, img, c->img_t[0].p, ib);
}

extern "C" int wct_coral_stats(wct_ctx* c, const uint8_t* img, int H, int W, double sums[9]) {
  ARG_CHECK(c && img && sums && H > 0 && W > 0);
  HIP_TRY(hipSetDevice(c->device));
  const size_t npix = (size_t)H * W;
  void* dimg;
  TRY(stage_in(c, 0, img, npix * 3, &dimg));
  int nblocks = (int)((npix + 255) / 256);
  if (nblocks > 1024) nblocks = 1024;
  TRY(ensure(c, c->stage[1], (size_t)(nblocks + 1) * 9 * sizeof(unsigned long long)));
  unsigned long long* partial = (unsigned long long*)c->stage[1].p;
  unsigned long long* out9 = partial + (size_t)nblocks * 9;
  TRY(launch_coral_stats((uint8_t*)dimg, npix, partial, nblocks, out9, c->stream));
  unsigned long long host9[9];
  TRY(fetch(c, host9, out9, sizeof(host9)));
  for (int k = 0; k < 9; ++k) sums[k] = (double)host9[k];     // exact: < 2^53
  return WCT_OK;
}

extern "C" int wct_coral_apply(wct_ctx* c, const uint8_t* src, int H, int W, const double M[9],
                               const double src_mean[3], const double src_std[3], const double tgt_mean[3],
                               const double tgt_std[3], uint8_t* out_u8, double* out_f64) {
  ARG_CHECK(c && src && M && src_mean && src_std && tgt_mean && tgt_std && (out_u8 || out_f64) && H > 0 && W > 0);
  HIP_TRY(hipSetDevice(c->device));
  const size_t npix = (size_t)H * W;
  CoralApplyArgs a;
  for (int i = 0; i < 9; ++i) a.M[i] = M[i];
  for (int i = 0; i < 3; ++i) { a.src_mean[i] = src_mean[i]; a.src_std[i] = src_std[i]; a.tgt_mean[i] = tgt_mean[i]; a.tgt_std[i] = tgt_std[i]; }
  void* dsrc;
  TRY(stage_in(c, 0, src, npix * 3, &dsrc));
  if (out_u8) TRY(ensure(c, c->stage[1], npix * 3));
  if (out_f64) TRY(ensure(c, c->stage[2], npix * 3 * sizeof(double)));
  TRY(launch_coral_apply((uint8_t*)dsrc, npix, a, out_u8 ? (uint8_t*)c->stage[1].p : nullptr,
                         out_f64 ? (double*)c->stage[2].p : nullptr, c->stream));
  if (out_u8) TRY(fetch(c, out_u8, c->stage[1].p, npix * 3));
  if (out_f64) TRY(fetch(c, out_f64, c->stage[2].p, npix * 3 * sizeof(double)));
  return WCT_OK;
}

// ---------------------------------------------------------------------------
// the hot path
// ---------------------------------------------------------------------------
extern "C" int wct_output_size(int Hc, int Wc, const int* levels, int n_levels, int* Ho, int* Wo) {
  ARG_CHECK(levels && Ho && Wo && n_levels >= 1 && Hc >= 2 && Wc >= 2);
  int H = Hc, W = Wc;
  for (int i = 0; i < n_levels; ++i) {
    ARG_CHECK(levels[i] >= 1 && levels[i] <= 5);
    TRY(check_min_size("content", H, W, levels[i]));
    int h, w;
    level_dims(H, W, levels[i], &h, &w);
    H = h << (levels[i] - 1); W = w << (levels[i] - 1);     // ceil pooling then x2 upsampling
  }
  *Ho = H; *Wo = W;
  return WCT_OK;
}

extern "C" int wct_stylize_batch_dev(wct_ctx* c, const uint8_t* content, int Hc, int Wc, const uint8_t* style,
                                     int Hs, int Ws, int B, const int* levels, int n_levels, float alpha,
                                     unsigned flags, uint8_t* out) {
  ARG_CHECK(c && content && style && out && levels && n_levels >= 1 && n_levels <= 16 && B >= 1 && B <= 32);
  HIP_TRY(hipSetDevice(c->device));
  int deepest = 0;
  for (int i = 0; i < n_levels; ++i) {
    ARG_CHECK(levels[i] >= 1 && levels[i] <= 5);
    if (levels[i] > deepest) deepest = levels[i];
    if (!c->dec[levels[i]].loaded) { wct_set_error("decoder weights for relu%d_1 not set", levels[i]); return WCT_ERR_STATE; }
  }
  int Ho, Wo;
  TRY(wct_output_size(Hc, Wc, levels, n_levels, &Ho, &Wo));
  TRY(check_min_size("style", Hs, Ws, deepest));

  // WCT_FLAG_STYLE_SHARED: `style` is ONE image for all B pairs (a video with a fixed style); its encoder pass,
  // statistics and eigensystems are computed once per call instead of once per pair
  const int shared = (flags & WCT_FLAG_STYLE_SHARED) ? 1 : 0;
  const int Bs = shared ? 1 : B;
  // images to fp32 in [0,1] (wct.py:60-64)
  const size_t nc = (size_t)B * Hc * Wc * 3, ns = (size_t)Bs * Hs * Ws * 3;
  const float* img_c = reinterpret_cast<const float*>(content);
  const float* img_s = reinterpret_cast<const float*>(style);
  if (!(flags & WCT_FLAG_IMAGES_F32)) {
    TRY(ensure(c, c->img_c, nc * 4));
    TRY(ensure(c, c->img_s, ns * 4));
    ProfScope ps(c, 7, 0, (double)(nc + ns) * 5);
    TRY(launch_u8_to_f32(content, (float*)c->img_c.p, nc, c->stream));
    TRY(launch_u8_to_f32(style, (float*)c->img_s.p, ns, c->stream));
    img_c = (const float*)c->img_c.p; img_s = (const float*)c->img_s.p;
  }
  // The per-channel sums and the largest value of every tap come out of the epilogue that writes it (16-pixel unit sums,
  // ConvArgs::usum): the transform's statistics pass then reads 1/16 of the feature bytes.  Needs a feature width that is
  // a multiple of 16; other widths (and WCT_FUSE_STATS=0) take the sums from the stored features -- the same bits.
  static const int fuse_stats = getenv("WCT_FUSE_STATS") ? atoi(getenv("WCT_FUSE_STATS")) : 1;
  const bool want_stats = fuse_stats != 0;
  constexpr size_t UROW = 32 * UMAX_SLOTS;                           // words per row: 32 images
  TRY(ensure(c, c->umax, 7 * UROW * sizeof(unsigned)));
  unsigned* const umax_c = (unsigned*)c->umax.p;                    // row 0: content (per level), rows 1..5: style levels
  if (want_stats) HIP_TRY(hipMemsetAsync(c->umax.p, 0, 7 * UROW * sizeof(unsigned), c->stream));
  // ONE style pass with a tap per requested level (model.py:69-75)
  float* taps[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* us_s[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned* um_s[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n_levels; ++i) {
    const int l = levels[i];
    int h, w;
    level_dims(Hs, Ws, l, &h, &w);
    TRY(ensure(c, c->feat_s[l], (size_t)Bs * h * w * LEVEL_C[l] * 4));
    taps[l] = (float*)c->feat_s[l].p;
    if (want_stats && w % 16 == 0) {
      TRY(ensure(c, c->usum_s[l], (size_t)Bs * h * (w / 16) * LEVEL_C[l] * 4));
      us_s[l] = (float*)c->usum_s[l].p; um_s[l] = umax_c + UROW * l;
    }
  }
  TRY(run_encoder(c, img_s, Bs, Hs, Ws, 0, deepest, taps, us_s, um_s));

  const float* cur = img_c;
  int H = Hc, W = Wc;
  for (int i = 0; i < n_levels; ++i) {
    const int l = levels[i], C = LEVEL_C[l];
    int h, w, hs, ws;
    level_dims(H, W, l, &h, &w);
    level_dims(Hs, Ws, l, &hs, &ws);
    TRY(ensure(c, c->feat_c, (size_t)B * h * w * C * 4));
    float* ctaps[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ctaps[l] = (float*)c->feat_c.p;
    float* us_c[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned* um_c[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (want_stats && w % 16 == 0) {
      TRY(ensure(c, c->usum_c, (size_t)B * h * (w / 16) * C * 4));
      us_c[l] = (float*)c->usum_c.p; um_c[l] = umax_c;
      if (i > 0) HIP_TRY(hipMemsetAsync(umax_c, 0, UROW * sizeof(unsigned), c->stream));
    }
    WctFeatStats st;
    st.u[0] = us_c[l]; st.umax[0] = um_c[l]; st.u[1] = us_s[l]; st.umax[1] = um_s[l];
    // level i>0 encodes clip(previous decoded, 0, 1) (model.py:86): the clamp is in the conv1_1 loader
    TRY(run_encoder(c, cur, B, H, W, i > 0, l, ctaps, us_c, um_c));
    TRY(ensure(c, c->wct_out, (size_t)B * h * w * C * 2));
    if (l == 5 && (flags & WCT_FLAG_SWAP5)) {
      // tf.case priority at relu5_1: swap5 > adain > wct (model.py:148-154); pairs one at a time
      ARG_CHECK(h >= c->ss_patch && w >= c->ss_patch && hs >= c->ss_patch && ws >= c->ss_patch);
      TRY(ensure(c, c->wct_ws, style_swap_workspace_bytes(C, h, w, hs, ws, c->ss_patch, c->ss_stride)));
      ProfScope ps(c, 7, 0, 0);
      for (int b = 0; b < B; ++b)
        TRY(launch_style_swap((float*)c->feat_c.p + (size_t)b * h * w * C, h, w,
                              (float*)c->feat_s[l].p + (size_t)(shared ? 0 : b) * hs * ws * C, hs, ws, C, c->ss_alpha, c->ss_patch,
                              c->ss_stride, -1.f, (half_t*)c->wct_out.p + (size_t)b * h * w * C, nullptr,
                              c->wct_ws.p, c->wct_ws.cap, c->stream, c->eig_fail_dev));
    } else
    TRY(run_transform(c, (float*)c->feat_c.p, h * w, (float*)c->feat_s[l].p, hs * ws, C, B, alpha, flags, -1.f,
                      (half_t*)c->wct_out.p, nullptr, nullptr, &st));
    const int scale = 1 << (l - 1);
    const int H2 = h * scale, W2 = w * scale;
    DevBuf& dst = c->img_t[i & 1];
    TRY(ensure(c, dst, (size_t)B * H2 * W2 * 3 * 4));
    TRY(run_decoder(c, l, (half_t*)c->wct_out.p, B, h, w, (float*)dst.p));
    cur = (float*)dst.p; H = H2; W = W2;
  }
  {
    ProfScope ps(c, 7, 0, (double)B * H * W * 3 * 5);
    TRY(launch_f32_to_u8(cur, out, (size_t)B * H * W * 3, c->stream));     // wct.py:66-68
  }
  return WCT_OK;
}

extern "C" int wct_stylize(wct_ctx* c, const uint8_t* content, int Hc, int Wc, const uint8_t* style, int Hs, int Ws,
                           const int* levels, int n_levels, float alpha, unsigned flags, uint8_t* out) {
  ARG_CHECK(c && content && style && out);
  HIP_TRY(hipSetDevice(c->device));
  TRY(eig_stale(c));
  int Ho, Wo;
  TRY(wct_output_size(Hc, Wc, levels, n_levels, &Ho, &Wo));
  void *dc, *ds;
  const size_t px = (flags & WCT_FLAG_IMAGES_F32) ? sizeof(float) : 1;       // bytes per colour sample of the inputs
  TRY(stage_in(c, 0, content, (size_t)Hc * Wc * 3 * px, &dc));
  TRY(stage_in(c, 1, style, (size_t)Hs * Ws * 3 * px, &ds));
  TRY(ensure(c, c->stage[2], (size_t)Ho * Wo * 3));
  TRY(wct_stylize_batch_dev(c, (uint8_t*)dc, Hc, Wc, (uint8_t*)ds, Hs, Ws, 1, levels, n_levels, alpha, flags,
                            (uint8_t*)c->stage[2].p));
  TRY(fetch(c, out, c->stage[2].p, (size_t)Ho * Wo * 3));
  return eig_status(c);
}

// ---------------------------------------------------------------------------
// decoder training (model.py:123-223, train.py:129-196): one optimiser step
// ---------------------------------------------------------------------------
namespace {
struct TrainArena {
  char* base; size_t off;
  template <typename T> T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (n * sizeof(T) + 255) & ~(size_t)255;
    return p;
  }
};
struct EncStep { int idx; int h, w; half_t* out; half_t* pooled; int ph, pw; };      // idx into enc[]; (h, w) conv dims
struct DecStep { int conv; int up; int h, w; const half_t* in; half_t* out; };       // (h, w) = conv output dims
}  // namespace

// Adam moments (zero) and the gradient arena of a decoder, allocated on first use.  Gradients of all layers live in
// one contiguous buffer ([w0][b0][w1][b1]..., every piece 64-float aligned) so that data-parallel training needs a
// single all-reduce per step (wct_train_grad_buffer).
static int ensure_train_state(wct_ctx* c, Decoder& d) {
  if (!d.mw.empty()) return WCT_OK;
  const int nconv = (int)d.w32.size();
  size_t total = 0;
  auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
  for (int i = 0; i < nconv; ++i) total += pad((size_t)9 * d.cin[i] * d.cout[i]) + pad((size_t)d.cout[i]);
  HIP_TRY(hipMalloc((void**)&d.grad_arena, total * sizeof(float)));
  HIP_TRY(hipMemsetAsync(d.grad_arena, 0, total * sizeof(float), c->stream));
  d.grad_count = total;
  size_t off = 0;
  for (int i = 0; i < nconv; ++i) {
    const size_t nw = (size_t)9 * d.cin[i] * d.cout[i], nb = (size_t)d.cout[i];
    float* p[4] = {nullptr, nullptr, nullptr, nullptr};
    const size_t bytes[4] = {nw * 4, nw * 4, nb * 4, nb * 4};
    for (int k = 0; k < 4; ++k) { HIP_TRY(hipMalloc((void**)&p[k], bytes[k])); HIP_TRY(hipMemsetAsync(p[k], 0, bytes[k], c->stream)); }
    d.mw.push_back(p[0]); d.vw.push_back(p[1]); d.mb.push_back(p[2]); d.vb.push_back(p[3]);
    d.gw.push_back(d.grad_arena + off); off += pad(nw);
    d.gb.push_back(d.grad_arena + off); off += pad(nb);
  }
  return WCT_OK;
}

// Adam on every conv of the decoder from the gradients currently in the arena, then refresh the fp16 forward weights
static int apply_adam(wct_ctx* c, Decoder& d, float lr, float beta1, float beta2, float eps, int step) {
  hipStream_t s = c->stream;
  const float lr_t = lr * sqrtf(1.f - powf(beta2, (float)step)) / (1.f - powf(beta1, (float)step));
  int ci = 0;
  for (int i = 0; i < (int)d.w32.size(); ++i) {
    const size_t nw = (size_t)9 * d.cin[i] * d.cout[i];
    TRY(launch_adam(d.w32[i], d.mw[i], d.vw[i], d.gw[i], nw, lr_t, beta1, beta2, eps, s));
    TRY(launch_adam(d.b32[i], d.mb[i], d.vb[i], d.gb[i], (size_t)d.cout[i], lr_t, beta1, beta2, eps, s));
    if (d.cout[i] == 3) {
      TRY(launch_pack_last_frag(d.w32[i], d.last_w, s));
      HIP_TRY(hipMemcpyAsync(d.last_b, d.b32[i], 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    } else {
      ConvLayer& l = d.convs[ci++];
      TRY(launch_pack_conv_frag(d.w32[i], l.w, l.cin, l.cout, s));
      if (l.ww) TRY(launch_pack_conv_wino_frag(d.w32[i], l.ww, l.cin, l.cout, s));
      HIP_TRY(hipMemcpyAsync(l.b, d.b32[i], (size_t)l.cout * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
  }
  return WCT_OK;
}

// images: host fp32 [B][H][W][3] in [0,1] (train.py:72-83).  losses_out[4] = feature, pixel, tv, total (host).
// step = 1-based optimiser step (Adam bias correction); lr is the already decayed learning rate (model.py:17-19).
extern "C" int wct_train_step(wct_ctx* c, int level, const float* images, int B, int H, int W,
                              float feature_weight, float pixel_weight, float tv_weight,
                              float lr, float beta1, float beta2, float eps, int step, float* losses_out) {
  ARG_CHECK(c && images && losses_out && level >= 1 && level <= 5 && B >= 1 && B <= 64 && step >= 1);
  HIP_TRY(hipSetDevice(c->device));
  if (!c->enc_loaded) { wct_set_error("encoder weights not set (wct_set_encoder)"); return WCT_ERR_STATE; }
  // every forward layer of a training step on the direct kernel: the backward pass differentiates THAT arithmetic (fp16-rounded
  // operands, exact products), and the gradient tests hold it to 1e-4
  struct DirectOnly { wct_ctx* c; DirectOnly(wct_ctx* x) : c(x) { c->no_wino = true; } ~DirectOnly() { c->no_wino = false; } } direct_only(c);
  Decoder& d = c->dec[level];
  if (!d.loaded) { wct_set_error("decoder weights for relu%d_1 not set (wct_set_decoder)", level); return WCT_ERR_STATE; }
  const int scale = 1 << (level - 1);
  ARG_CHECK(H % scale == 0 && W % scale == 0 && H / scale >= 2 && W / scale >= 2);   // the decoder returns exactly H x W
  hipStream_t s = c->stream;
  const int C = LEVEL_C[level];
  const int h = H / scale, w = W / scale;
  const int nconv = (int)d.w32.size();
  static const int seq_tap[12] = {0, 2, 0, 3, 0, 0, 0, 4, 0, 0, 0, 5};
  static const int pool_after[12] = {1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0};
  int tap_idx = -1;
  for (int i = 0; i < 12 && level > 1; ++i) if (seq_tap[i] == level) tap_idx = i;

  // optimiser state, allocated on first use
  TRY(ensure_train_state(c, d));

  // ---- carve the workspace (first pass sizes, second pass pointers)
  std::vector<EncStep> enc_steps;
  std::vector<DecStep> dec_steps;
  float *X = nullptr, *F = nullptr, *Fp = nullptr, *D = nullptr, *g0 = nullptr, *g1 = nullptr, *col = nullptr, *gp = nullptr,
        *partial = nullptr, *wt = nullptr, *dloss = nullptr;
  half_t *A0 = nullptr, *e0 = nullptr;
  double* lpart = nullptr;
  size_t total = 0;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) TRY(ensure(c, c->train_ws, total));
    TrainArena ar = {pass ? (char*)c->train_ws.p : nullptr, 0};
    enc_steps.clear(); dec_steps.clear();
    size_t gmax = (size_t)B * H * W * 64, colmax = 0, gpmax = 0, partmax = 256 * 512, wtmax = 0;
    X = ar.take<float>((size_t)B * H * W * 3);
    F = ar.take<float>((size_t)B * h * w * C);
    Fp = ar.take<float>((size_t)B * h * w * C);
    A0 = ar.take<half_t>((size_t)B * h * w * C);
    D = ar.take<float>((size_t)B * H * W * 3);
    // decoder activations
    {
      int hh = h, ww = w, up = 0, ci = 0;
      const half_t* cur = A0;
      for (auto& st : d.plan) {
        if (st.kind == 'U') { up = 1; hh *= 2; ww *= 2; continue; }
        half_t* out = st.cout == 3 ? nullptr : ar.take<half_t>((size_t)B * hh * ww * st.cout);
        dec_steps.push_back({ci, up, hh, ww, cur, out});
        const size_t px = (size_t)B * hh * ww, pxp = (size_t)B * (hh + 2) * (ww + 2);
        colmax = std::max(colmax, std::max(px * 9 * st.cin, pxp * 9 * st.cout));
        gpmax = std::max(gpmax, pxp * st.cin);
        gmax = std::max(gmax, std::max(px * st.cin, px * (size_t)std::max(st.cout, 4)));
        partmax = std::max(partmax, (size_t)conv_wgrad_splits(B, hh, ww) * 9 * st.cin * st.cout);
        wtmax = std::max(wtmax, (size_t)9 * st.cin * st.cout);
        cur = out; up = 0; ++ci;
      }
    }
    // encoder pass over the decoded image, every activation kept
    e0 = ar.take<half_t>((size_t)B * H * W * 64);
    {
      int hh = H, ww = W;
      colmax = std::max(colmax, (size_t)B * (hh + 2) * (ww + 2) * 9 * 64);
      gpmax = std::max(gpmax, (size_t)B * (hh + 2) * (ww + 2) * 64);
      for (int i = 0; i <= tap_idx; ++i) {
        const int cout = ENC_COUT[i], cin = ENC_CIN[i];
        EncStep es = {i, hh, ww, nullptr, nullptr, 0, 0};
        if (i != tap_idx) es.out = ar.take<half_t>((size_t)B * hh * ww * cout);
        const size_t pxp = (size_t)B * (hh + 2) * (ww + 2);
        colmax = std::max(colmax, pxp * 9 * cout);
        gpmax = std::max(gpmax, pxp * cin);
        gmax = std::max(gmax, (size_t)B * hh * ww * std::max(cin, cout));
        if (pool_after[i] && i != tap_idx) {
          es.ph = (hh + 1) / 2; es.pw = (ww + 1) / 2;
          es.pooled = ar.take<half_t>((size_t)B * es.ph * es.pw * cout);
          hh = es.ph; ww = es.pw;
        }
        enc_steps.push_back(es);
      }
    }
    g0 = ar.take<float>(gmax); g1 = ar.take<float>(gmax);
    col = ar.take<float>(colmax); gp = ar.take<float>(gpmax);
    partial = ar.take<float>(partmax); wt = ar.take<float>(wtmax);
    lpart = ar.take<double>(1024); dloss = ar.take<float>(16);      // 4 losses + scratch of the backward GEMMs
    total = ar.off;
  }

  // ---- forward
  HIP_TRY(hipMemcpyAsync(X, images, (size_t)B * H * W * 3 * sizeof(float), hipMemcpyHostToDevice, s));
  {
    float* taps[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    taps[level] = F;
    TRY(run_encoder(c, X, B, H, W, 0, level, taps));                       // content features (the target of the feature loss)
  }
  TRY(launch_f32_to_f16(F, A0, (size_t)B * h * w * C, s));
  for (auto& ds : dec_steps) {
    const int cin = d.cin[ds.conv], cout = d.cout[ds.conv];
    if (cout == 3) {
      ConvLastArgs a;
      a.x = ds.in; a.wfrag = d.last_w; a.bias = d.last_b; a.y = D; a.B = B; a.H = ds.h; a.W = ds.w;
      TRY(launch_conv_last(a, s));
    } else {
      TRY(run_conv(c, d.convs[ds.conv], ds.in, ds.out, nullptr, B, ds.h, ds.w, ds.up, 1));
    }
    (void)cin;
  }
  {  // encoder over the decoded image (not clipped in training: model.py:176), pools as separate kernels
    ConvFirstArgs a;
    a.x = D; a.wfrag = c->first_w; a.bias = c->first_b; a.y16 = level > 1 ? e0 : nullptr; a.y32 = level == 1 ? Fp : nullptr;
    a.B = B; a.H = H; a.W = W; a.clamp01 = 0; a.usum = nullptr; a.umax = nullptr;
    TRY(launch_conv_first(a, s));
    const half_t* cur = e0;
    for (auto& es : enc_steps) {
      const bool last = es.idx == tap_idx;
      TRY(run_conv(c, c->enc[es.idx], cur, last ? nullptr : es.out, last ? Fp : nullptr, B, es.h, es.w, 0, 1));
      cur = es.out;
      if (es.pooled) { TRY(launch_maxpool2x2(es.out, es.pooled, B, es.h, es.w, ENC_COUT[es.idx], s)); cur = es.pooled; }
    }
  }

  // ---- losses and the gradient w.r.t. the decoded image
  const size_t nF = (size_t)B * h * w * C, nD = (size_t)B * H * W * 3;
  float* g = g0; float* gother = g1;
  TRY(launch_mse(Fp, F, nF, feature_weight, g, 0, lpart, dloss + 0, s));     // g = d feature_loss / d F'
  TRY(launch_relu_mask32(g, Fp, nF, s));
  if (level == 1) {
    TRY(launch_conv_first_dgrad(g, c->first_w32, gp, B, H, W, s));
  } else {
    for (int k = (int)enc_steps.size() - 1; k >= 0; --k) {
      const EncStep& es = enc_steps[k];
      const int cin = ENC_CIN[es.idx], cout = ENC_COUT[es.idx];
      TRY(launch_pow2_scale(g, (size_t)B * es.h * es.w * cout, dloss + 8, s));
      TRY(launch_conv_dgrad(g, c->enc_wt[es.idx], B, es.h, es.w, cin, cout, col, gp, gother, dloss + 8, s));   // w.r.t. the conv input
      std::swap(g, gother);
      if (k > 0) {
        const EncStep& pv = enc_steps[k - 1];
        if (pv.pooled) {
          TRY(launch_maxpool_adjoint(pv.out, g, gother, B, pv.h, pv.w, ENC_COUT[pv.idx], s));
          std::swap(g, gother);
        }
        TRY(launch_relu_mask16(g, pv.out, (size_t)B * pv.h * pv.w * ENC_COUT[pv.idx], s));
      } else {
        TRY(launch_relu_mask16(g, e0, (size_t)B * H * W * 64, s));
      }
    }
    TRY(launch_conv_first_dgrad(g, c->first_w32, gp, B, H, W, s));
  }
  TRY(launch_reflect_fold(gp, gother, B, H, W, 3, s));                       // d feature_loss / d D
  g = gother; gother = (g == g0) ? g1 : g0;
  TRY(launch_mse(D, X, nD, pixel_weight, g, 1, lpart, dloss + 1, s));        // += d pixel_loss / d D
  TRY(launch_tv(D, B, H, W, 3, tv_weight, g, lpart, dloss + 2, s));           // += d tv_loss / d D

  // ---- decoder backward: weight / bias gradients, data gradients down to the second conv
  for (int k = (int)dec_steps.size() - 1; k >= 0; --k) {
    const DecStep& ds = dec_steps[k];
    const int cin = d.cin[ds.conv], cout = d.cout[ds.conv];
    const size_t px = (size_t)B * ds.h * ds.w;
    if (cout != 3) TRY(launch_relu_mask16(g, ds.out, px * cout, s));
    TRY(launch_bias_grad(g, px, cout, partial, d.gb[ds.conv], s));
    int cpad = cout;
    if (cout == 3) {           // the GEMM operands want 16-byte rows: pad the 3-channel gradient to 4
      TRY(launch_pad3to4(g, gother, px, s));
      std::swap(g, gother);
      cpad = 4;
    }
    TRY(launch_pow2_scale(g, px * cpad, dloss + 8, s));                  // one scale per gradient tensor, both GEMMs use it
    TRY(launch_conv_wgrad(ds.in, ds.up, g, cpad, B, ds.h, ds.w, cin, cout, col, partial, conv_wgrad_splits(B, ds.h, ds.w), d.gw[ds.conv], dloss + 8, s));
    if (k > 0) {
      TRY(launch_transpose_w(d.w32[ds.conv], wt, cin, cout, cpad, s));
      TRY(launch_conv_dgrad(g, wt, B, ds.h, ds.w, cin, cpad, col, gp, gother, dloss + 8, s));
      std::swap(g, gother);
      if (ds.up) {
        TRY(launch_upsample_adjoint(g, gother, B, ds.h / 2, ds.w / 2, cin, s));
        std::swap(g, gother);
      }
    }
  }

  // ---- Adam (model.py:199: beta1 0.9, beta2 0.999 in the reference), then refresh the fp16 forward weights
  if (lr != 0.f) TRY(apply_adam(c, d, lr, beta1, beta2, eps, step));
  float hl[4];
  HIP_TRY(hipMemcpyAsync(hl, dloss, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  losses_out[0] = hl[0]; losses_out[1] = hl[1]; losses_out[2] = hl[2]; losses_out[3] = hl[0] + hl[1] + hl[2];
  return WCT_OK;
}

// the decoder's current fp32 weights (after training steps) and the gradients of the last wct_train_step,
// conv `layer` of the plan (the output conv is the last); any pointer may be NULL
extern "C" int wct_get_decoder_layer(wct_ctx* c, int level, int layer, float* w_hwio, float* bias, float* grad_w, float* grad_b) {
  ARG_CHECK(c && level >= 1 && level <= 5);
  HIP_TRY(hipSetDevice(c->device));
  Decoder& d = c->dec[level];
  if (!d.loaded) { wct_set_error("decoder weights for relu%d_1 not set (wct_set_decoder)", level); return WCT_ERR_STATE; }
  ARG_CHECK(layer >= 0 && layer < (int)d.w32.size());
  const size_t nw = (size_t)9 * d.cin[layer] * d.cout[layer] * sizeof(float), nb = (size_t)d.cout[layer] * sizeof(float);
  if ((grad_w || grad_b) && d.gw.empty()) { wct_set_error("no training step has run for relu%d_1", level); return WCT_ERR_STATE; }
  if (w_hwio) HIP_TRY(hipMemcpyAsync(w_hwio, d.w32[layer], nw, hipMemcpyDeviceToHost, c->stream));
  if (bias) HIP_TRY(hipMemcpyAsync(bias, d.b32[layer], nb, hipMemcpyDeviceToHost, c->stream));
  if (grad_w) HIP_TRY(hipMemcpyAsync(grad_w, d.gw[layer], nw, hipMemcpyDeviceToHost, c->stream));
  if (grad_b) HIP_TRY(hipMemcpyAsync(grad_b, d.gb[layer], nb, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return WCT_OK;
}

// Data-parallel training: run wct_train_step with lr = 0 on every rank's shard, all-reduce (average) the gradient
// buffer this returns over RCCL, then wct_train_apply on every rank.  *grad_dev is ONE contiguous device buffer of
// *count floats holding the gradients of all layers of the decoder (the layout is private; it is the same on every
// rank).  The single-process sequence train_step(lr=0) + train_apply(lr) equals train_step(lr) bit for bit.
extern "C" int wct_train_grad_buffer(wct_ctx* c, int level, float** grad_dev, size_t* count) {
  ARG_CHECK(c && grad_dev && count && level >= 1 && level <= 5);
  HIP_TRY(hipSetDevice(c->device));
  Decoder& d = c->dec[level];
  if (!d.loaded) { wct_set_error("decoder weights for relu%d_1 not set (wct_set_decoder)", level); return WCT_ERR_STATE; }
  TRY(ensure_train_state(c, d));
  HIP_TRY(hipStreamSynchronize(c->stream));
  *grad_dev = d.grad_arena; *count = d.grad_count;
  return WCT_OK;
}
extern "C" int wct_train_apply(wct_ctx* c, int level, float lr, float beta1, float beta2, float eps, int step) {
  ARG_CHECK(c && level >= 1 && level <= 5 && step >= 1);
  HIP_TRY(hipSetDevice(c->device));
  Decoder& d = c->dec[level];
  if (!d.loaded || d.mw.empty()) { wct_set_error("no gradients for relu%d_1: run wct_train_step first", level); return WCT_ERR_STATE; }
  TRY(apply_adam(c, d, lr, beta1, beta2, eps, step));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return WCT_OK;
}
