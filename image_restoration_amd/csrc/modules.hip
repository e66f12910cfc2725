// Module-level entry points: MDTA (Attention.forward/backward, Restormer.py:99-132) and GDFN
// (FeedForward.forward/backward, Restormer.py:76-93) as fixed sequences of the three kernel
// archetypes (pointwise MFMA GEMM, pixel-axis Gram, depthwise stencil) plus the c x c glue.
// Host code only: carves the caller's saved/workspace blobs and enqueues kernels on the given stream.
#include <stdlib.h>

#include "internal.h"

namespace mi {

static size_t fbytes(size_t n) { return align_up(n * sizeof(float), 256); }
static size_t tbytes(size_t n, int dt) { return align_up(n * dtype_size(dt), 256); }
template <typename A, typename B> static size_t max2(A a, B b) { return (size_t)a > (size_t)b ? (size_t)a : (size_t)b; }

// plain 1x1 conv: y[B,M,N] = W[M,K] x[B,K,N] (+bias) (+res);  transposed: W given as [K,M] used as its transpose
static mi_pw_desc conv1x1(const void* x, int K, const float* w, bool transposed, int w_ld, const float* bias,
                          const void* res, void* y, int M, int B, int64_t N, int dtype) {
  mi_pw_desc d;
  memset(&d, 0, sizeof(d));
  d.x1 = x; d.x1_bs = (int64_t)K * N; d.k1 = K;
  d.w = w;
  if (transposed) { d.w_sm = 1; d.w_sk = w_ld; } else { d.w_sm = w_ld; d.w_sk = 1; }
  d.bias = bias;
  d.r = res; d.r_bs = (int64_t)M * N;
  d.y = y; d.y_bs = (int64_t)M * N;
  d.m = M; d.n = N; d.batch = B; d.groups = 1; d.dtype = dtype;
  return d;
}

// ------------------------------------------------------------------ weight gradient beside input gradient
// The weight-gradient Gram (dW = dY X^T) and the input-gradient GEMM (dX = W^T dY) of one 1x1 conv both stream dY.  Run one
// after the other over a 32-image batch, the second reads dY from HBM again (2 GB at C = 96: nothing survives in the 256 MiB
// Infinity Cache).  Launched side by side on two streams they walk dY in the same order, and whichever runs behind finds the
// other's lines in the Infinity Cache.  MI_CO_STREAM=1 enables it (A/B switch); the side stream and its events are created on
// first use (i.e. in an eager warm-up step, before any graph capture) and joined back before the entry point returns.
struct CoStream {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
static bool co_stream_enabled() {
  const char* e = MI_ENV(MI_CO_STREAM);
  return e && atoi(e) == 1;
}
static CoStream* co_stream() {
  static thread_local CoStream cs[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  CoStream& c = cs[dev];
  if (!c.side) {
    if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &c;
}
// fork: the side stream waits for everything enqueued on `st` so far; returns the stream to launch the side work on
static hipStream_t co_fork(hipStream_t st) {
  if (!co_stream_enabled()) return st;
  CoStream* c = co_stream();
  if (!c) return st;
  if (hipEventRecord(c->fork, st) != hipSuccess || hipStreamWaitEvent(c->side, c->fork, 0) != hipSuccess) return st;
  return c->side;
}
static int co_join(hipStream_t side, hipStream_t st) {
  if (side == st) return MI_OK;
  CoStream* c = co_stream();
  MI_CHECK_ARG(c != nullptr, "co_join: side stream lost");
  MI_CHECK_HIP(hipEventRecord(c->join, side));
  MI_CHECK_HIP(hipStreamWaitEvent(st, c->join, 0));
  return MI_OK;
}

// ------------------------------------------------------------------ attention core (shared by MDTA and cross-MDTA)
// q, k, v are channel slices of NCHW tensors: base pointer + batch stride (elements); heads are contiguous channel
// groups of c = C/heads rows (Restormer.py:117-119 'b (head c) h w').
struct QkvView { const void* q; int64_t q_bs; const void* k; int64_t k_bs; const void* v; int64_t v_bs; };
struct AttnDims { int B, C, heads, dtype; int64_t N; };
struct AttnSaved { float* A; float* P; float* nrm; float* M; void* Mb; void* Mtb; };   // Mb / Mtb: bf16 M_b and M_b^T (mi_pw_desc.w_b16)
struct AttnScratch {
  float* graw; float* ss; float* dM; float* dwo_part; float* dtemp_part; float* wd; float* attn_scr; void* wdb;
  void* gram_ws; void* pw_ws; void* cs_ws;
};

static void attn_saved_carve(Carver& cv, const AttnDims& d, AttnSaved* r) {
  const size_t C = d.C, B = d.B, c = C / d.heads, Z = B * d.heads;
  r->A = cv.take<float>(fbytes(Z * c * c));
  r->P = cv.take<float>(fbytes(Z * c * c));
  r->nrm = cv.take<float>(fbytes(Z * 2 * c));
  r->M = cv.take<float>(fbytes(B * C * C));
  r->Mb = cv.take(B * C * C * 2);
  r->Mtb = cv.take(B * C * C * 2);
}

static mi_gram_desc attn_qk_gram(const AttnDims& d, const QkvView& v, float* graw, float* ss) {
  const int c = d.C / d.heads;
  mi_gram_desc g;
  memset(&g, 0, sizeof(g));
  g.a = v.q; g.a_bs = v.q_bs; g.a_gs = (int64_t)c * d.N; g.ma = c;
  g.b = v.k; g.b_bs = v.k_bs; g.b_gs = (int64_t)c * d.N; g.mb = c;
  g.n = d.N; g.batch = d.B; g.groups = d.heads; g.dtype = d.dtype;
  g.out = graw; g.out_ld = c; g.out_zs = (int64_t)c * c; g.sumsq = ss;
  return g;
}
static mi_gram_desc attn_dm_gram(const AttnDims& d, const void* dout, const QkvView& v, float* dM) {
  mi_gram_desc g;
  memset(&g, 0, sizeof(g));
  g.a = dout; g.a_bs = (int64_t)d.C * d.N; g.ma = d.C;
  g.b = v.v; g.b_bs = v.v_bs; g.mb = d.C;
  g.n = d.N; g.batch = d.B; g.groups = 1; g.dtype = d.dtype;
  g.out = dM; g.out_ld = d.C; g.out_zs = (int64_t)d.C * d.C;
  return g;
}
static mi_gram_desc wgrad_gram(const void* dy, int m, const void* x, int k, int B, int64_t N, int dtype, float* out,
                               int accumulate) {
  mi_gram_desc g;
  memset(&g, 0, sizeof(g));
  g.a = dy; g.a_bs = (int64_t)m * N; g.ma = m;
  g.b = x; g.b_bs = (int64_t)k * N; g.mb = k;
  g.n = N; g.batch = B; g.groups = 1; g.dtype = dtype;
  g.sum_batch = 1; g.accumulate = accumulate; g.out = out; g.out_ld = k; g.out_zs = 0;
  return g;
}
// the grouped per-image GEMM(s) of the q/k gradients over the stacked operand [k; q] (two K-panels of c rows each); weights
// [Z][2c][2c] (attn_bwd_finish_kernel): rows 0..c-1 give dq, rows c..2c-1 give dk.  rows = 2c with a second output: both in one pass.
static mi_pw_desc attn_dqk_desc(const AttnDims& d, const QkvView& v, const float* w, int row0, int rows, void* y, int64_t y_bs,
                                const void* wb = nullptr) {
  const int c = d.C / d.heads;
  mi_pw_desc dd;
  memset(&dd, 0, sizeof(dd));
  dd.x1 = v.k; dd.x1_bs = v.k_bs; dd.x1_gs = (int64_t)c * d.N; dd.k1 = c;
  dd.x2 = v.q; dd.x2_bs = v.q_bs; dd.x2_gs = (int64_t)c * d.N; dd.k2 = c;
  dd.w = w + (int64_t)row0 * 2 * c; dd.w_bs = (int64_t)d.heads * 2 * c * 2 * c; dd.w_gs = (int64_t)2 * c * 2 * c; dd.w_sm = 2 * c; dd.w_sk = 1;
  if (wb) { dd.w_b16 = (const char*)wb + (int64_t)row0 * 2 * c * 2; dd.w_b16_sm = 2 * c; }
  dd.y = y; dd.y_bs = y_bs; dd.y_gs = (int64_t)c * d.N;
  dd.m = rows; dd.n = d.N; dd.batch = d.B; dd.groups = d.heads; dd.dtype = d.dtype;
  return dd;
}
static mi_pw_desc attn_dqk_merged(const AttnDims& d, const QkvView& v, const float* w, void* dq, int64_t dq_bs, void* dk, int64_t dk_bs,
                                  const void* wb = nullptr) {
  const int c = d.C / d.heads;
  mi_pw_desc dd = attn_dqk_desc(d, v, w, 0, 2 * c, dq, dq_bs, wb);
  dd.y_split = c; dd.y2 = dk; dd.y2_bs = dk_bs; dd.y2_gs = (int64_t)c * d.N;
  return dd;
}

static void attn_scratch_carve(Carver& cv, const AttnDims& d, AttnScratch* w) {
  const size_t C = d.C, B = d.B, c = C / d.heads, Z = B * d.heads;
  w->graw = cv.take<float>(fbytes(Z * c * c));
  w->ss = cv.take<float>(fbytes(Z * 2 * c));
  w->dM = cv.take<float>(fbytes(B * C * C));
  w->dwo_part = cv.take<float>(fbytes(B * C * C));
  w->dtemp_part = cv.take<float>(fbytes(Z));
  w->wd = cv.take<float>(fbytes(Z * 2 * c * 2 * c));     // [Z][2c][2c]: rows of dq, then rows of dk, over the stacked [k; q]
  w->wdb = cv.take(Z * 2 * c * 2 * c * 2);               // the same in bf16
  w->attn_scr = cv.take<float>(fbytes(attn_bwd_scratch_floats((int)B, (int)C, d.heads)));
  QkvView fake{(void*)256, 0, (void*)256, 0, (void*)256, 0};
  mi_gram_desc g1 = attn_qk_gram(d, fake, (float*)256, (float*)256);
  mi_gram_desc g2 = attn_dm_gram(d, (void*)256, fake, (float*)256);
  w->gram_ws = nullptr; w->pw_ws = nullptr; w->cs_ws = nullptr;  // sized by the caller together with its own GEMMs
  (void)g1; (void)g2;
}
static size_t attn_gram_ws_bytes(const AttnDims& d) {
  QkvView fake{(void*)256, 0, (void*)256, 0, (void*)256, 0};
  mi_gram_desc g1 = attn_qk_gram(d, fake, (float*)256, (float*)256);
  mi_gram_desc g2 = attn_dm_gram(d, (void*)256, fake, (float*)256);
  return max2(mi_gram_workspace(&g1), mi_gram_workspace(&g2));
}
static size_t attn_pw_ws_bytes(const AttnDims& d) {
  mi_pw_desc b = conv1x1((void*)256, d.C, (const float*)256, false, d.C, nullptr, nullptr, (void*)256, d.C, d.B, d.N, d.dtype);
  b.w_bs = (int64_t)d.C * d.C;  // per-image C x C (M_b and its transpose)
  QkvView fake{(void*)256, 0, (void*)256, 0, (void*)256, 0};
  mi_pw_desc q = attn_dqk_desc(d, fake, (const float*)256, 0, d.C / d.heads, (void*)256, 0);
  mi_pw_desc q2 = attn_dqk_merged(d, fake, (const float*)256, (void*)256, 0, (void*)256, 0);
  return max2(max2(mi_pw_gemm_workspace(&b), mi_pw_gemm_workspace(&q)), mi_pw_gemm_workspace(&q2));
}

// out = (residual?) + project_out(softmax(temperature * q^ k^T) v)        Restormer.py:121-131
static int attn_core_fwd(const AttnDims& d, const QkvView& v, const float* temperature, const float* proj_w,
                         const float* proj_b, const void* residual, void* out, const AttnSaved& sv, const AttnScratch& w,
                         void* stream, const mi_f8_scales* f8 = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  mi_gram_desc g = attn_qk_gram(d, v, w.graw, w.ss);
  MI_TRY(mi_gram(&g, w.gram_ws, stream));
  MI_TRY(launch_attn_fold(w.graw, w.ss, temperature, proj_w, sv.P, sv.A, sv.nrm, sv.M, d.B, d.C, d.heads, st, sv.Mb, sv.Mtb));
  mi_pw_desc d2 = conv1x1(v.v, d.C, sv.M, false, d.C, proj_b, residual, out, d.C, d.B, d.N, d.dtype);
  d2.x1_bs = v.v_bs;
  d2.w_bs = (int64_t)d.C * d.C;
  d2.w_b16 = sv.Mb; d2.w_b16_sm = d.C;
  if (f8) { d2.f8 = 1; d2.f8_sx = f8->x2; d2.f8_sw = f8->w2; }
  return mi_pw_gemm(&d2, w.pw_ws, stream);
}

// given dout: gradients w.r.t. q, k, v (written to the given channel slices) and temperature / project_out params
static int attn_core_bwd(const AttnDims& d, const QkvView& v, const void* dout, void* dq, int64_t dq_bs, void* dk,
                         int64_t dk_bs, void* dv, int64_t dv_bs, const AttnSaved& sv, const float* temperature,
                         const float* proj_w, float* g_temperature, float* g_proj_w, float* g_proj_b, int acc,
                         const AttnScratch& w, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int B = d.B, C = d.C, hd = d.heads;
  if (g_proj_b) MI_TRY(launch_chan_sum(dout, g_proj_b, B, C, d.N, d.dtype, acc, w.cs_ws, st));
  mi_gram_desc g1 = attn_dm_gram(d, dout, v, w.dM);  // dM_b = dY V^T
  MI_TRY(mi_gram(&g1, w.gram_ws, stream));
  float* dwo_part = w.dwo_part;
  float* dtemp_part = w.dtemp_part;
  if (acc) {                 // parameter gradients accumulated in place: partials may wait for mi_deferred_flush (common.h)
    float* a1 = deferred_take((size_t)B * C * C, st);
    float* a2 = a1 ? deferred_take((size_t)B * hd, st) : nullptr;
    if (a1 && a2) { dwo_part = a1; dtemp_part = a2; }
  }
  MI_TRY(launch_attn_bwd_small(w.dM, sv.A, sv.P, sv.nrm, temperature, proj_w, dwo_part, dtemp_part, w.wd,
                               w.attn_scr, B, C, hd, st, w.wdb));
  MI_TRY(launch_reduce_rows(dwo_part, g_proj_w, B, (int64_t)C * C, (int64_t)C * C, acc, 1.0f, st));
  MI_TRY(launch_reduce_rows(dtemp_part, g_temperature, B, hd, hd, acc, 1.0f, st));
  // dq = G1 k + D1 q ; dk = G1^T q + D2 k   (grouped over heads, per-image weights)
  // both in one pass over q and k where the GEMM form can write two outputs (bf16 wave-owned forms), else one GEMM each
  const int c = C / hd;
  mi_pw_desc dd = attn_dqk_merged(d, v, w.wd, dq, dq_bs, dk, dk_bs, w.wdb);
  if (mi_pw_gemm_split_ok(&dd) && !MI_ENV(MI_ATTN_DQK_SPLIT)) {
    MI_TRY(mi_pw_gemm(&dd, w.pw_ws, stream));
  } else {
    dd = attn_dqk_desc(d, v, w.wd, 0, c, dq, dq_bs, w.wdb);
    MI_TRY(mi_pw_gemm(&dd, w.pw_ws, stream));
    dd = attn_dqk_desc(d, v, w.wd, c, c, dk, dk_bs, w.wdb);
    MI_TRY(mi_pw_gemm(&dd, w.pw_ws, stream));
  }
  // dv = M_b^T dY
  mi_pw_desc dvd = conv1x1(dout, C, sv.M, true, C, nullptr, nullptr, dv, C, B, d.N, d.dtype);
  dvd.w_bs = (int64_t)C * C;
  dvd.y_bs = dv_bs;
  dvd.w_b16 = sv.Mtb; dvd.w_b16_sm = C;                  // M_b^T, written row-major by attn_fold
  return mi_pw_gemm(&dvd, w.pw_ws, stream);
}

// ------------------------------------------------------------------ MDTA
struct MdtaSaved { void* qkv0; void* qkv; AttnSaved at; size_t bytes; };
static AttnDims mdta_dims(const mi_mdta_shape* s) { return AttnDims{s->B, s->C, s->heads, s->dtype, (int64_t)s->H * s->W}; }
static MdtaSaved mdta_saved_layout(const mi_mdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B;
  Carver cv(base);
  MdtaSaved r;
  r.qkv0 = cv.take(tbytes(B * 3 * C * N, s->dtype));
  r.qkv = cv.take(tbytes(B * 3 * C * N, s->dtype));
  attn_saved_carve(cv, mdta_dims(s), &r.at);
  r.bytes = cv.off;
  return r;
}
static QkvView mdta_view(const mi_mdta_shape* s, const void* qkv) {
  const int64_t N = (int64_t)s->H * s->W, bs = 3 * (int64_t)s->C * N;
  const size_t es = dtype_size(s->dtype), plane = (size_t)s->C * N * es;
  return QkvView{qkv, bs, (const char*)qkv + plane, bs, (const char*)qkv + 2 * plane, bs};
}

struct MdtaWs { AttnScratch at; MdtaSaved inf; void* dqkv; void* dqkv0; void* dw_ws; size_t bytes; };
static MdtaWs mdta_ws_layout(const mi_mdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B;
  const AttnDims d = mdta_dims(s);
  Carver cv(base);
  MdtaWs w;
  attn_scratch_carve(cv, d, &w.at);
  mi_gram_desc g3 = wgrad_gram((void*)256, 3 * (int)C, (void*)256, (int)C, (int)B, (int64_t)N, s->dtype, (float*)256, 0);
  w.at.gram_ws = cv.take(max2(attn_gram_ws_bytes(d), mi_gram_workspace(&g3)));
  {  // weight-pack scratch of the largest 1x1 GEMM of this module
    mi_pw_desc a = conv1x1((void*)256, (int)C, (const float*)256, false, (int)C, nullptr, nullptr, (void*)256, 3 * (int)C,
                           (int)B, (int64_t)N, s->dtype);
    mi_pw_desc e = conv1x1((void*)256, 3 * (int)C, (const float*)256, true, (int)C, nullptr, nullptr, (void*)256, (int)C,
                           (int)B, (int64_t)N, s->dtype);  // input gradient: W_qkv^T
    w.at.pw_ws = cv.take(max2(max2(mi_pw_gemm_workspace(&a), mi_pw_gemm_workspace(&e)), attn_pw_ws_bytes(d)));
  }
  w.dw_ws = cv.take(mi_dwconv_bwd_workspace((int)B, 3 * (int)C, s->H, s->W, s->ks));
  w.at.cs_ws = cv.take(chan_sum_workspace(3 * (int)C, (int64_t)N));
  // big activation-sized buffers last: forward(inference) and backward never run concurrently on one blob
  size_t mark = cv.off;
  w.inf = mdta_saved_layout(s, base ? (char*)base + mark : nullptr);
  w.dqkv = base ? (char*)base + mark : nullptr;
  w.dqkv0 = base ? (char*)base + mark + tbytes(B * 3 * C * N, s->dtype) : nullptr;
  size_t bwd_big = 2 * tbytes(B * 3 * C * N, s->dtype);
  w.bytes = mark + max2(w.inf.bytes, bwd_big);
  return w;
}

static int mdta_check(const mi_mdta_shape* s) {
  MI_CHECK_ARG(s, "mdta: null shape");
  MI_CHECK_ARG(s->B > 0 && s->C > 0 && s->heads > 0 && s->H > 0 && s->W > 0, "mdta: bad shape");
  MI_CHECK_ARG(s->C % s->heads == 0, "mdta: C=%d not divisible by heads=%d", s->C, s->heads);
  MI_CHECK_ARG(s->dtype == MI_F32 || s->dtype == MI_BF16, "mdta: bad dtype %d", s->dtype);
  MI_CHECK_ARG(s->ks == 3 || s->ks == 5 || s->ks == 7, "mdta: bad depthwise kernel size %d", s->ks);
  return MI_OK;
}

// ------------------------------------------------------------------ cross-MDTA (q from x, k/v from y)
struct XmdtaSaved { void* q0; void* q; void* kv0; void* kv; AttnSaved at; size_t bytes; };
static AttnDims xmdta_dims(const mi_xmdta_shape* s) { return AttnDims{s->B, s->C, s->heads, s->dtype, (int64_t)s->H * s->W}; }
static XmdtaSaved xmdta_saved_layout(const mi_xmdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B;
  Carver cv(base);
  XmdtaSaved r;
  r.q0 = cv.take(tbytes(B * C * N, s->dtype));
  r.q = cv.take(tbytes(B * C * N, s->dtype));
  r.kv0 = cv.take(tbytes(B * 2 * C * N, s->dtype));
  r.kv = cv.take(tbytes(B * 2 * C * N, s->dtype));
  attn_saved_carve(cv, xmdta_dims(s), &r.at);
  r.bytes = cv.off;
  return r;
}
static QkvView xmdta_view(const mi_xmdta_shape* s, const void* q, const void* kv) {
  const int64_t N = (int64_t)s->H * s->W;
  const size_t plane = (size_t)s->C * N * dtype_size(s->dtype);
  return QkvView{q, (int64_t)s->C * N, kv, 2 * (int64_t)s->C * N, (const char*)kv + plane, 2 * (int64_t)s->C * N};
}
struct XmdtaWs { AttnScratch at; XmdtaSaved inf; void* dq; void* dq0; void* dkv; void* dkv0; void* dw_ws; size_t bytes; };
static XmdtaWs xmdta_ws_layout(const mi_xmdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B;
  const AttnDims d = xmdta_dims(s);
  Carver cv(base);
  XmdtaWs w;
  attn_scratch_carve(cv, d, &w.at);
  mi_gram_desc g3 = wgrad_gram((void*)256, 2 * (int)C, (void*)256, (int)C, (int)B, (int64_t)N, s->dtype, (float*)256, 0);
  w.at.gram_ws = cv.take(max2(attn_gram_ws_bytes(d), mi_gram_workspace(&g3)));
  {
    mi_pw_desc a = conv1x1((void*)256, (int)C, (const float*)256, false, (int)C, nullptr, nullptr, (void*)256, 2 * (int)C,
                           (int)B, (int64_t)N, s->dtype);
    mi_pw_desc e = conv1x1((void*)256, 2 * (int)C, (const float*)256, true, (int)C, nullptr, nullptr, (void*)256, (int)C,
                           (int)B, (int64_t)N, s->dtype);
    w.at.pw_ws = cv.take(max2(max2(mi_pw_gemm_workspace(&a), mi_pw_gemm_workspace(&e)), attn_pw_ws_bytes(d)));
  }
  const int ksm = s->ks_q > s->ks_kv ? s->ks_q : s->ks_kv;
  w.dw_ws = cv.take(mi_dwconv_bwd_workspace((int)B, 2 * (int)C, s->H, s->W, ksm));
  w.at.cs_ws = cv.take(chan_sum_workspace(2 * (int)C, (int64_t)N));
  size_t mark = cv.off;
  w.inf = xmdta_saved_layout(s, base ? (char*)base + mark : nullptr);
  Carver big(base ? (char*)base + mark : nullptr);
  w.dq = big.take(tbytes(B * C * N, s->dtype));
  w.dq0 = big.take(tbytes(B * C * N, s->dtype));
  w.dkv = big.take(tbytes(B * 2 * C * N, s->dtype));
  w.dkv0 = big.take(tbytes(B * 2 * C * N, s->dtype));
  w.bytes = mark + max2(w.inf.bytes, big.off);
  return w;
}
static int xmdta_check(const mi_xmdta_shape* s) {
  MI_CHECK_ARG(s, "xmdta: null shape");
  MI_CHECK_ARG(s->B > 0 && s->C > 0 && s->heads > 0 && s->H > 0 && s->W > 0, "xmdta: bad shape");
  MI_CHECK_ARG(s->C % s->heads == 0, "xmdta: C=%d not divisible by heads=%d", s->C, s->heads);
  MI_CHECK_ARG(s->dtype == MI_F32 || s->dtype == MI_BF16, "xmdta: bad dtype %d", s->dtype);
  MI_CHECK_ARG((s->ks_q == 3 || s->ks_q == 5 || s->ks_q == 7) && (s->ks_kv == 3 || s->ks_kv == 5 || s->ks_kv == 7),
               "xmdta: bad depthwise kernel sizes %d/%d", s->ks_q, s->ks_kv);
  return MI_OK;
}

// ------------------------------------------------------------------ GDFN
// Saved for backward: the conv input h0 (2h planes) and the gate output g (h planes).  The conv OUTPUT h1 (2h planes, 40%
// of the GDFN's saved bytes) is stored only where the backward kernel cannot recompute it from h0 (5x5 / 7x7, odd row
// widths; mi_dwconv_gate_recompute_ok) or when MI_GDFN_STORE_Y=1 asks for the old behaviour.  Beyond the Infinity Cache
// (bs 32, C=254 at 256^2) forward + backward of the depthwise stage take 376 + 685 us this way against 540 + 741 us with a
// stored y; the recomputing backward needed packed-fp32 math for that (975 us without: VALU-bound).
struct GdfnSaved { void* h0; void* h1; void* g; size_t bytes; };
static bool gdfn_recompute(const mi_gdfn_shape* s) {
  // the layout switch travels in the shape (flags bit 0), never in the environment: forward and backward of one call pair
  // must carve the saved blob identically
  return !(s->flags & 1) && mi_dwconv_gate_recompute_ok(s->H, s->W, s->ks) != 0;
}
static GdfnSaved gdfn_saved_layout(const mi_gdfn_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, B = s->B, h = s->hidden;
  Carver cv(base);
  GdfnSaved r;
  r.h0 = cv.take(tbytes(B * 2 * h * N, s->dtype));
  r.h1 = gdfn_recompute(s) ? nullptr : cv.take(tbytes(B * 2 * h * N, s->dtype));
  r.g = cv.take(tbytes(B * h * N, s->dtype));
  r.bytes = cv.off;
  return r;
}
struct GdfnWs { void* gram_ws; void* pw_ws; void* dw_ws; void* cs_ws; GdfnSaved inf; void* dg; void* dh0; size_t bytes; };
static GdfnWs gdfn_ws_layout(const mi_gdfn_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, B = s->B, h = s->hidden, C = s->C;
  Carver cv(base);
  GdfnWs w;
  mi_gram_desc g1 = wgrad_gram((void*)256, (int)C, (void*)256, (int)h, (int)B, (int64_t)N, s->dtype, (float*)256, 0);
  mi_gram_desc g2 = wgrad_gram((void*)256, 2 * (int)h, (void*)256, (int)C, (int)B, (int64_t)N, s->dtype, (float*)256, 0);
  w.gram_ws = cv.take(max2(mi_gram_workspace(&g1), mi_gram_workspace(&g2)));
  {
    mi_pw_desc a = conv1x1((void*)256, (int)C, (const float*)256, false, (int)C, nullptr, nullptr, (void*)256, 2 * (int)h,
                           (int)B, (int64_t)N, s->dtype);
    mi_pw_desc b = conv1x1((void*)256, 2 * (int)h, (const float*)256, true, (int)C, nullptr, nullptr, (void*)256, (int)C,
                           (int)B, (int64_t)N, s->dtype);
    w.pw_ws = cv.take(max2(mi_pw_gemm_workspace(&a), mi_pw_gemm_workspace(&b)));
  }
  w.dw_ws = cv.take(mi_dwconv_bwd_workspace((int)B, 2 * (int)h, s->H, s->W, s->ks));
  w.cs_ws = cv.take(chan_sum_workspace(2 * (int)h > (int)C ? 2 * (int)h : (int)C, (int64_t)N));
  size_t mark = cv.off;
  w.inf = gdfn_saved_layout(s, base ? (char*)base + mark : nullptr);
  w.dg = base ? (char*)base + mark : nullptr;
  w.dh0 = base ? (char*)base + mark + tbytes(B * h * N, s->dtype) : nullptr;
  size_t bwd_big = tbytes(B * h * N, s->dtype) + tbytes(B * 2 * h * N, s->dtype);
  w.bytes = mark + max2(w.inf.bytes, bwd_big);
  return w;
}
static int gdfn_check(const mi_gdfn_shape* s) {
  MI_CHECK_ARG(s, "gdfn: null shape");
  MI_CHECK_ARG(s->B > 0 && s->C > 0 && s->hidden > 0 && s->H > 0 && s->W > 0, "gdfn: bad shape");
  MI_CHECK_ARG(s->dtype == MI_F32 || s->dtype == MI_BF16, "gdfn: bad dtype %d", s->dtype);
  MI_CHECK_ARG(s->ks == 3 || s->ks == 5 || s->ks == 7, "gdfn: bad depthwise kernel size %d", s->ks);
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" size_t mi_mdta_saved_bytes(const mi_mdta_shape* s) {
  if (mdta_check(s) != MI_OK) return 0;
  return mdta_saved_layout(s, nullptr).bytes;
}
extern "C" size_t mi_mdta_workspace(const mi_mdta_shape* s) {
  if (mdta_check(s) != MI_OK) return 0;
  return mdta_ws_layout(s, nullptr).bytes;
}

// LayerNorm in front of a 1x1 conv, applied inside the GEMM as the X tile is loaded (mi_pw_desc.ln_*): the normalised tensor
// never exists in HBM.  ln == nullptr: x is already normalised.
static void ln_head_apply(mi_pw_desc* d, const mi_ln_head* ln) {
  if (!ln) return;
  d->ln_w = ln->w; d->ln_b = ln->with_bias ? ln->b : nullptr; d->ln_mean = ln->mean; d->ln_rstd = ln->rstd;
  d->ln_mode = ln->with_bias ? 1 : 2;
}
static int ln_head_check(const mi_ln_head* ln, const char* who) {
  MI_CHECK_ARG(ln && ln->w && (!ln->with_bias || ln->b) && ((ln->mean == nullptr) == (ln->rstd == nullptr)),
               "%s: LayerNorm head needs w (and b for WithBias); mean / rstd both or neither", who);
  return MI_OK;
}
static int mdta_fwd_impl(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* residual, void* out,
                         void* saved, void* ws, void* stream, const mi_ln_head* ln, const mi_f8_scales* f8 = nullptr) {
  MI_TRY(mdta_check(s));
  MI_CHECK_ARG(p && x && out && ws, "mdta_fwd: null pointer");
  MI_CHECK_ARG(p->temperature && p->qkv_w && p->dw_w && p->proj_w, "mdta_fwd: null parameter");
  const int B = s->B, C = s->C, dt = s->dtype;
  const int64_t N = (int64_t)s->H * s->W;
  MdtaWs w = mdta_ws_layout(s, ws);
  MdtaSaved sv = saved ? mdta_saved_layout(s, saved) : w.inf;
  // qkv0 = qkv(x);  qkv = dw(qkv0)                               Restormer.py:114
  mi_pw_desc d1 = conv1x1(x, C, p->qkv_w, false, C, p->qkv_b, nullptr, sv.qkv0, 3 * C, B, N, dt);
  ln_head_apply(&d1, ln);
  if (f8) { d1.f8 = 1; d1.f8_sx = f8->x1; d1.f8_sw = f8->w1; }
  MI_TRY(mi_pw_gemm(&d1, w.at.pw_ws, stream));
  MI_TRY(mi_dwconv_fwd(sv.qkv0, p->dw_w, p->dw_b, sv.qkv, B, 3 * C, s->H, s->W, s->ks, dt, stream));
  return attn_core_fwd(mdta_dims(s), mdta_view(s, sv.qkv), p->temperature, p->proj_w, p->proj_b, residual, out, sv.at, w.at,
                       stream, f8);
}
extern "C" int mi_mdta_fwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* residual, void* out,
                           void* saved, void* ws, void* stream) {
  return mdta_fwd_impl(s, p, x, residual, out, saved, ws, stream, nullptr);
}
extern "C" int mi_mdta_fwd_ln_ok(const mi_mdta_shape* s) {
  if (mdta_check(s) != MI_OK) return 0;
  mi_pw_desc d = conv1x1((void*)256, s->C, (const float*)256, false, s->C, nullptr, nullptr, (void*)256, 3 * s->C, s->B,
                         (int64_t)s->H * s->W, s->dtype);
  return mi_pw_gemm_ln_ok(&d);
}
extern "C" int mi_mdta_fwd_ln(const mi_mdta_shape* s, const mi_mdta_params* p, const mi_ln_head* ln, const void* x,
                              const void* residual, void* out, void* saved, void* ws, void* stream) {
  MI_TRY(ln_head_check(ln, "mdta_fwd_ln"));
  MI_CHECK_ARG(mi_mdta_fwd_ln_ok(s), "mdta_fwd_ln: shape not covered (bf16, C <= 128, H*W %% 64 == 0)");
  return mdta_fwd_impl(s, p, x, residual, out, saved, ws, stream, ln);
}

// fp8 MFMA operands in both projections (inference: nothing is saved).  ln may be NULL (x is then the LayerNorm output).
static int f8_check(const mi_f8_scales* f, const char* who) {
  MI_CHECK_ARG(f && f->x1 > 0.f && f->w1 > 0.f && f->x2 > 0.f && f->w2 > 0.f, "%s: fp8 scales must be positive", who);
  return MI_OK;
}
extern "C" int mi_mdta_fwd_f8_ok(const mi_mdta_shape* s, int with_ln) {
  if (mdta_check(s) != MI_OK || s->dtype != MI_BF16) return 0;
  const int64_t N = (int64_t)s->H * s->W;
  mi_pw_desc d1 = conv1x1((void*)256, s->C, (const float*)256, false, s->C, nullptr, nullptr, (void*)256, 3 * s->C, s->B, N, s->dtype);
  mi_pw_desc d2 = conv1x1((void*)256, s->C, (const float*)256, false, s->C, nullptr, (void*)256, (void*)256, s->C, s->B, N, s->dtype);
  d2.x1_bs = 3 * (int64_t)s->C * N;
  d2.w_bs = (int64_t)s->C * s->C;
  if (with_ln && !mi_pw_gemm_ln_ok(&d1)) return 0;
  return mi_pw_gemm_f8_ok(&d1) && mi_pw_gemm_f8_ok(&d2);
}
extern "C" int mi_mdta_fwd_f8(const mi_mdta_shape* s, const mi_mdta_params* p, const mi_ln_head* ln, const mi_f8_scales* f8,
                              const void* x, const void* residual, void* out, void* ws, void* stream) {
  MI_TRY(f8_check(f8, "mdta_fwd_f8"));
  if (ln) MI_TRY(ln_head_check(ln, "mdta_fwd_f8"));
  MI_CHECK_ARG(mi_mdta_fwd_f8_ok(s, ln != nullptr), "mdta_fwd_f8: shape not covered (bf16, both projections on a wave-owned form)");
  return mdta_fwd_impl(s, p, x, residual, out, nullptr, ws, stream, ln, f8);
}

// ln == nullptr: x is the conv input (LayerNorm OUTPUT) and dx its gradient.  ln != nullptr: x is the LayerNorm INPUT; the
// qkv weight gradient, W_qkv^T dY, the LayerNorm backward and the residual add run as one launch (bwd_tail.hip) and dx is the
// gradient of the half-block's input.
static int mdta_bwd_impl(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* dout, void* dx,
                         const mi_mdta_grads* gr, const void* saved, void* ws, void* stream, const mi_ln_tail* ln) {
  MI_TRY(mdta_check(s));
  MI_CHECK_ARG(p && x && dout && dx && gr && saved && ws, "mdta_bwd: null pointer");
  MI_CHECK_ARG(gr->temperature && gr->qkv_w && gr->dw_w && gr->proj_w, "mdta_bwd: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, C = s->C, dt = s->dtype, acc = gr->accumulate;
  const int64_t N = (int64_t)s->H * s->W, bs = 3 * (int64_t)C * N;
  const size_t plane = (size_t)C * N * dtype_size(dt);
  MdtaWs w = mdta_ws_layout(s, ws);
  MdtaSaved sv = mdta_saved_layout(s, const_cast<void*>(saved));
  char* dq = (char*)w.dqkv;
  MI_TRY(attn_core_bwd(mdta_dims(s), mdta_view(s, sv.qkv), dout, dq, bs, dq + plane, bs, dq + 2 * plane, bs, sv.at,
                       p->temperature, p->proj_w, gr->temperature, gr->proj_w, gr->proj_b, acc, w.at, stream));
  // depthwise backward: d_qkv -> d_qkv0, weight/bias grads
  MI_TRY(mi_dwconv_bwd(w.dqkv, sv.qkv0, p->dw_w, w.dqkv0, gr->dw_w, gr->dw_b, B, 3 * C, s->H, s->W, s->ks, acc, dt, w.dw_ws,
                       stream));
  // qkv 1x1: weight grad (sum over batch), bias grad, input grad
  if (ln)
    return launch_bwd_tail(w.dqkv0, 3 * C, x, C, ln->dres, ln->mean, ln->rstd, p->qkv_w, ln->w, ln->b, dx, gr->qkv_w, ln->dw,
                           ln->db, B, N, acc, (char*)ws + align_up(w.bytes, 256), st);
  hipStream_t sd = co_fork(st);
  mi_gram_desc g2 = wgrad_gram(w.dqkv0, 3 * C, x, C, B, N, dt, gr->qkv_w, acc);
  MI_TRY(mi_gram(&g2, w.at.gram_ws, sd));
  if (gr->qkv_b) MI_TRY(launch_chan_sum(w.dqkv0, gr->qkv_b, B, 3 * C, N, dt, acc, w.at.cs_ws, sd));
  mi_pw_desc dxd = conv1x1(w.dqkv0, 3 * C, p->qkv_w, true, C, nullptr, nullptr, dx, C, B, N, dt);
  MI_TRY(mi_pw_gemm(&dxd, w.at.pw_ws, stream));
  return co_join(sd, st);
}
extern "C" int mi_mdta_bwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* dout, void* dx,
                           const mi_mdta_grads* gr, const void* saved, void* ws, void* stream) {
  return mdta_bwd_impl(s, p, x, dout, dx, gr, saved, ws, stream, nullptr);
}
static int ln_tail_check(const mi_ln_tail* ln, const char* who) {
  MI_CHECK_ARG(ln && ln->w && ln->b && ln->mean && ln->rstd && ln->dw && ln->db, "%s: LayerNorm tail needs w, b, mean, rstd, dw, db", who);
  return MI_OK;
}
extern "C" int mi_mdta_bwd_ln_ok(const mi_mdta_shape* s, int qkv_bias) {
  if (mdta_check(s) != MI_OK || qkv_bias) return 0;
  return bwd_tail_ok(3 * s->C, s->C, (int64_t)s->H * s->W, s->dtype) && bwd_tail_pays(3 * s->C, s->C) ? 1 : 0;
}
extern "C" size_t mi_mdta_bwd_ln_workspace(const mi_mdta_shape* s) {
  if (!mi_mdta_bwd_ln_ok(s, 0)) return 0;
  return align_up(mdta_ws_layout(s, nullptr).bytes, 256) + bwd_tail_workspace(3 * s->C, s->C);
}
extern "C" int mi_mdta_bwd_ln(const mi_mdta_shape* s, const mi_mdta_params* p, const mi_ln_tail* ln, const void* x,
                              const void* dout, void* dx, const mi_mdta_grads* gr, const void* saved, void* ws, void* stream) {
  MI_TRY(ln_tail_check(ln, "mdta_bwd_ln"));
  MI_CHECK_ARG(gr && !gr->qkv_b && mi_mdta_bwd_ln_ok(s, 0), "mdta_bwd_ln: shape not covered (bf16, C 48/96, H*W %% 64 == 0, no qkv bias)");
  return mdta_bwd_impl(s, p, x, dout, dx, gr, saved, ws, stream, ln);
}

extern "C" size_t mi_xmdta_saved_bytes(const mi_xmdta_shape* s) {
  if (xmdta_check(s) != MI_OK) return 0;
  return xmdta_saved_layout(s, nullptr).bytes;
}
extern "C" size_t mi_xmdta_workspace(const mi_xmdta_shape* s) {
  if (xmdta_check(s) != MI_OK) return 0;
  return xmdta_ws_layout(s, nullptr).bytes;
}

extern "C" int mi_xmdta_fwd(const mi_xmdta_shape* s, const mi_xmdta_params* p, const void* x, const void* y,
                            const void* residual, void* out, void* saved, void* ws, void* stream) {
  MI_TRY(xmdta_check(s));
  MI_CHECK_ARG(p && x && y && out && ws, "xmdta_fwd: null pointer");
  MI_CHECK_ARG(p->temperature && p->q_w && p->q_dw_w && p->kv_w && p->kv_dw_w && p->proj_w, "xmdta_fwd: null parameter");
  const int B = s->B, C = s->C, dt = s->dtype;
  const int64_t N = (int64_t)s->H * s->W;
  XmdtaWs w = xmdta_ws_layout(s, ws);
  XmdtaSaved sv = saved ? xmdta_saved_layout(s, saved) : w.inf;
  // q = q_dwconv(q(x)) ; kv = kv_dwconv(kv(y))                    moce_ir.py:348-349, AdaIR-main/net/model.py:197-198
  mi_pw_desc d1 = conv1x1(x, C, p->q_w, false, C, p->q_b, nullptr, sv.q0, C, B, N, dt);
  MI_TRY(mi_pw_gemm(&d1, w.at.pw_ws, stream));
  MI_TRY(mi_dwconv_fwd(sv.q0, p->q_dw_w, p->q_dw_b, sv.q, B, C, s->H, s->W, s->ks_q, dt, stream));
  mi_pw_desc d2 = conv1x1(y, C, p->kv_w, false, C, p->kv_b, nullptr, sv.kv0, 2 * C, B, N, dt);
  MI_TRY(mi_pw_gemm(&d2, w.at.pw_ws, stream));
  MI_TRY(mi_dwconv_fwd(sv.kv0, p->kv_dw_w, p->kv_dw_b, sv.kv, B, 2 * C, s->H, s->W, s->ks_kv, dt, stream));
  return attn_core_fwd(xmdta_dims(s), xmdta_view(s, sv.q, sv.kv), p->temperature, p->proj_w, p->proj_b, residual, out, sv.at,
                       w.at, stream);
}

extern "C" int mi_xmdta_bwd(const mi_xmdta_shape* s, const mi_xmdta_params* p, const void* x, const void* y,
                            const void* dout, void* dx, void* dy, const mi_xmdta_grads* gr, const void* saved, void* ws,
                            void* stream) {
  MI_TRY(xmdta_check(s));
  MI_CHECK_ARG(p && x && y && dout && dx && dy && gr && saved && ws, "xmdta_bwd: null pointer");
  MI_CHECK_ARG(gr->temperature && gr->q_w && gr->q_dw_w && gr->kv_w && gr->kv_dw_w && gr->proj_w,
               "xmdta_bwd: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, C = s->C, dt = s->dtype, acc = gr->accumulate;
  const int64_t N = (int64_t)s->H * s->W;
  const size_t plane = (size_t)C * N * dtype_size(dt);
  XmdtaWs w = xmdta_ws_layout(s, ws);
  XmdtaSaved sv = xmdta_saved_layout(s, const_cast<void*>(saved));
  MI_TRY(attn_core_bwd(xmdta_dims(s), xmdta_view(s, sv.q, sv.kv), dout, w.dq, (int64_t)C * N, w.dkv, 2 * (int64_t)C * N,
                       (char*)w.dkv + plane, 2 * (int64_t)C * N, sv.at, p->temperature, p->proj_w, gr->temperature,
                       gr->proj_w, gr->proj_b, acc, w.at, stream));
  // q branch
  MI_TRY(mi_dwconv_bwd(w.dq, sv.q0, p->q_dw_w, w.dq0, gr->q_dw_w, gr->q_dw_b, B, C, s->H, s->W, s->ks_q, acc, dt, w.dw_ws,
                       stream));
  mi_gram_desc g1 = wgrad_gram(w.dq0, C, x, C, B, N, dt, gr->q_w, acc);
  MI_TRY(mi_gram(&g1, w.at.gram_ws, stream));
  if (gr->q_b) MI_TRY(launch_chan_sum(w.dq0, gr->q_b, B, C, N, dt, acc, w.at.cs_ws, st));
  mi_pw_desc dxd = conv1x1(w.dq0, C, p->q_w, true, C, nullptr, nullptr, dx, C, B, N, dt);
  MI_TRY(mi_pw_gemm(&dxd, w.at.pw_ws, stream));
  // kv branch
  MI_TRY(mi_dwconv_bwd(w.dkv, sv.kv0, p->kv_dw_w, w.dkv0, gr->kv_dw_w, gr->kv_dw_b, B, 2 * C, s->H, s->W, s->ks_kv, acc, dt,
                       w.dw_ws, stream));
  mi_gram_desc g2 = wgrad_gram(w.dkv0, 2 * C, y, C, B, N, dt, gr->kv_w, acc);
  MI_TRY(mi_gram(&g2, w.at.gram_ws, stream));
  if (gr->kv_b) MI_TRY(launch_chan_sum(w.dkv0, gr->kv_b, B, 2 * C, N, dt, acc, w.at.cs_ws, st));
  mi_pw_desc dyd = conv1x1(w.dkv0, 2 * C, p->kv_w, true, C, nullptr, nullptr, dy, C, B, N, dt);
  return mi_pw_gemm(&dyd, w.at.pw_ws, stream);
}

extern "C" size_t mi_gdfn_saved_bytes(const mi_gdfn_shape* s) {
  if (gdfn_check(s) != MI_OK) return 0;
  return gdfn_saved_layout(s, nullptr).bytes;
}
extern "C" size_t mi_gdfn_workspace(const mi_gdfn_shape* s) {
  if (gdfn_check(s) != MI_OK) return 0;
  return gdfn_ws_layout(s, nullptr).bytes;
}

static int gdfn_fwd_impl(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* residual, void* out,
                         void* saved, void* ws, void* stream, const mi_ln_head* ln, const mi_f8_scales* f8 = nullptr) {
  MI_TRY(gdfn_check(s));
  MI_CHECK_ARG(p && x && out && ws, "gdfn_fwd: null pointer");
  MI_CHECK_ARG(p->in_w && p->dw_w && p->out_w, "gdfn_fwd: null parameter");
  const int B = s->B, C = s->C, h = s->hidden, dt = s->dtype;
  const int64_t N = (int64_t)s->H * s->W;
  GdfnWs w = gdfn_ws_layout(s, ws);
  GdfnSaved sv = saved ? gdfn_saved_layout(s, saved) : w.inf;
  mi_pw_desc d1 = conv1x1(x, C, p->in_w, false, C, p->in_b, nullptr, sv.h0, 2 * h, B, N, dt);       // Restormer.py:89
  ln_head_apply(&d1, ln);
  if (f8) { d1.f8 = 1; d1.f8_sx = f8->x1; d1.f8_sw = f8->w1; }
  MI_TRY(mi_pw_gemm(&d1, w.pw_ws, stream));
  MI_TRY(mi_dwconv_gate_fwd(sv.h0, p->dw_w, p->dw_b, (saved && !gdfn_recompute(s)) ? sv.h1 : nullptr, sv.g, B, 2 * h, s->H,
                            s->W, s->ks, dt, stream));                                               // :90-91
  mi_pw_desc d2 = conv1x1(sv.g, h, p->out_w, false, h, p->out_b, residual, out, C, B, N, dt);       // :92
  if (f8) { d2.f8 = 1; d2.f8_sx = f8->x2; d2.f8_sw = f8->w2; }
  MI_TRY(mi_pw_gemm(&d2, w.pw_ws, stream));
  return MI_OK;
}
extern "C" int mi_gdfn_fwd(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* residual, void* out,
                           void* saved, void* ws, void* stream) {
  return gdfn_fwd_impl(s, p, x, residual, out, saved, ws, stream, nullptr);
}
extern "C" int mi_gdfn_fwd_ln_ok(const mi_gdfn_shape* s) {
  if (gdfn_check(s) != MI_OK) return 0;
  mi_pw_desc d = conv1x1((void*)256, s->C, (const float*)256, false, s->C, nullptr, nullptr, (void*)256, 2 * s->hidden, s->B,
                         (int64_t)s->H * s->W, s->dtype);
  return mi_pw_gemm_ln_ok(&d);
}
extern "C" int mi_gdfn_fwd_ln(const mi_gdfn_shape* s, const mi_gdfn_params* p, const mi_ln_head* ln, const void* x,
                              const void* residual, void* out, void* saved, void* ws, void* stream) {
  MI_TRY(ln_head_check(ln, "gdfn_fwd_ln"));
  MI_CHECK_ARG(mi_gdfn_fwd_ln_ok(s), "gdfn_fwd_ln: shape not covered (bf16, C <= 128, H*W %% 64 == 0)");
  return gdfn_fwd_impl(s, p, x, residual, out, saved, ws, stream, ln);
}

extern "C" int mi_gdfn_fwd_f8_ok(const mi_gdfn_shape* s, int with_ln) {
  if (gdfn_check(s) != MI_OK || s->dtype != MI_BF16) return 0;
  const int64_t N = (int64_t)s->H * s->W;
  mi_pw_desc d1 = conv1x1((void*)256, s->C, (const float*)256, false, s->C, nullptr, nullptr, (void*)256, 2 * s->hidden, s->B, N, s->dtype);
  mi_pw_desc d2 = conv1x1((void*)256, s->hidden, (const float*)256, false, s->hidden, nullptr, (void*)256, (void*)256, s->C, s->B, N, s->dtype);
  if (with_ln && !mi_pw_gemm_ln_ok(&d1)) return 0;
  return mi_pw_gemm_f8_ok(&d1) && mi_pw_gemm_f8_ok(&d2);
}
extern "C" int mi_gdfn_fwd_f8(const mi_gdfn_shape* s, const mi_gdfn_params* p, const mi_ln_head* ln, const mi_f8_scales* f8,
                              const void* x, const void* residual, void* out, void* ws, void* stream) {
  MI_TRY(f8_check(f8, "gdfn_fwd_f8"));
  if (ln) MI_TRY(ln_head_check(ln, "gdfn_fwd_f8"));
  MI_CHECK_ARG(mi_gdfn_fwd_f8_ok(s, ln != nullptr), "gdfn_fwd_f8: shape not covered (bf16, both projections on a wave-owned form)");
  return gdfn_fwd_impl(s, p, x, residual, out, nullptr, ws, stream, ln, f8);
}

// One-launch forward of the LayerNorm + GDFN half-block on the TRAINING path (csrc/fused_gdfn.hip): `saved` is the blob
// mi_gdfn_saved_bytes sizes for this shape (flags 0: h0 and g; the conv output is recomputed in backward), so mi_gdfn_bwd /
// mi_gdfn_bwd_ln read it exactly as after mi_gdfn_fwd_ln.
extern "C" int mi_gdfn_fused_fwd_train_ok(const mi_gdfn_fused_shape* f) {
  if (!f || !mi_gdfn_fused_ok(f)) return 0;
  mi_gdfn_shape s = {f->B, f->C, f->hidden, f->H, f->W, MI_BF16, 3, 0};
  return gdfn_check(&s) == MI_OK && gdfn_recompute(&s) ? 1 : 0;
}
extern "C" int mi_gdfn_fused_fwd_train(const mi_gdfn_fused_shape* f, const void* pack, const void* y, void* out, float* mean,
                                       float* rstd, void* saved, void* stream) {
  MI_CHECK_ARG(mi_gdfn_fused_fwd_train_ok(f), "gdfn_fused_fwd_train: shape not covered (mi_gdfn_fused_fwd_train_ok)");
  MI_CHECK_ARG(saved, "gdfn_fused_fwd_train: null saved blob");
  mi_gdfn_shape s = {f->B, f->C, f->hidden, f->H, f->W, MI_BF16, 3, 0};
  GdfnSaved sv = gdfn_saved_layout(&s, saved);
  return fused_gdfn_fwd_save(f, pack, y, out, mean, rstd, sv.h0, sv.g, (hipStream_t)stream);
}

static int gdfn_bwd_impl(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* dout, void* dx,
                         const mi_gdfn_grads* gr, const void* saved, void* ws, void* stream, const mi_ln_tail* ln) {
  MI_TRY(gdfn_check(s));
  MI_CHECK_ARG(p && x && dout && dx && gr && saved && ws, "gdfn_bwd: null pointer");
  MI_CHECK_ARG(gr->in_w && gr->dw_w && gr->out_w, "gdfn_bwd: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, C = s->C, h = s->hidden, dt = s->dtype, acc = gr->accumulate;
  const int64_t N = (int64_t)s->H * s->W;
  GdfnWs w = gdfn_ws_layout(s, ws);
  GdfnSaved sv = gdfn_saved_layout(s, const_cast<void*>(saved));
  if (gr->out_b) MI_TRY(launch_chan_sum(dout, gr->out_b, B, C, N, dt, acc, w.cs_ws, st));
  mi_gram_desc g1 = wgrad_gram(dout, C, sv.g, h, B, N, dt, gr->out_w, acc);
  MI_TRY(mi_gram(&g1, w.gram_ws, stream));
  mi_pw_desc d1 = conv1x1(dout, C, p->out_w, true, h, nullptr, nullptr, w.dg, h, B, N, dt);
  MI_TRY(mi_pw_gemm(&d1, w.pw_ws, stream));
  if (gdfn_recompute(s))
    MI_TRY(mi_dwconv_gate_bwd_recompute(w.dg, sv.h0, p->dw_w, p->dw_b, w.dh0, gr->dw_w, gr->dw_b, B, 2 * h, s->H, s->W, s->ks,
                                        acc, dt, w.dw_ws, stream));
  else
    MI_TRY(mi_dwconv_gate_bwd(w.dg, sv.h1, sv.h0, p->dw_w, w.dh0, gr->dw_w, gr->dw_b, B, 2 * h, s->H, s->W, s->ks, acc, dt,
                              w.dw_ws, stream));
  if (ln)
    return launch_bwd_tail(w.dh0, 2 * h, x, C, ln->dres, ln->mean, ln->rstd, p->in_w, ln->w, ln->b, dx, gr->in_w, ln->dw, ln->db,
                           B, N, acc, (char*)ws + align_up(w.bytes, 256), st);
  hipStream_t sd = co_fork(st);
  mi_gram_desc g2 = wgrad_gram(w.dh0, 2 * h, x, C, B, N, dt, gr->in_w, acc);
  MI_TRY(mi_gram(&g2, w.gram_ws, sd));
  if (gr->in_b) MI_TRY(launch_chan_sum(w.dh0, gr->in_b, B, 2 * h, N, dt, acc, w.cs_ws, sd));
  mi_pw_desc d2 = conv1x1(w.dh0, 2 * h, p->in_w, true, C, nullptr, nullptr, dx, C, B, N, dt);
  MI_TRY(mi_pw_gemm(&d2, w.pw_ws, stream));
  return co_join(sd, st);
}
extern "C" int mi_gdfn_bwd(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* dout, void* dx,
                           const mi_gdfn_grads* gr, const void* saved, void* ws, void* stream) {
  return gdfn_bwd_impl(s, p, x, dout, dx, gr, saved, ws, stream, nullptr);
}
extern "C" int mi_gdfn_bwd_ln_ok(const mi_gdfn_shape* s, int in_bias) {
  if (gdfn_check(s) != MI_OK || in_bias) return 0;
  return bwd_tail_ok(2 * s->hidden, s->C, (int64_t)s->H * s->W, s->dtype) && bwd_tail_pays(2 * s->hidden, s->C) ? 1 : 0;
}
extern "C" size_t mi_gdfn_bwd_ln_workspace(const mi_gdfn_shape* s) {
  if (!mi_gdfn_bwd_ln_ok(s, 0)) return 0;
  return align_up(gdfn_ws_layout(s, nullptr).bytes, 256) + bwd_tail_workspace(2 * s->hidden, s->C);
}
extern "C" int mi_gdfn_bwd_ln(const mi_gdfn_shape* s, const mi_gdfn_params* p, const mi_ln_tail* ln, const void* x,
                              const void* dout, void* dx, const mi_gdfn_grads* gr, const void* saved, void* ws, void* stream) {
  MI_TRY(ln_tail_check(ln, "gdfn_bwd_ln"));
  MI_CHECK_ARG(gr && !gr->in_b && mi_gdfn_bwd_ln_ok(s, 0), "gdfn_bwd_ln: shape not covered (bf16, C 48/96, H*W %% 64 == 0, no project_in bias)");
  return gdfn_bwd_impl(s, p, x, dout, dx, gr, saved, ws, stream, ln);
}
