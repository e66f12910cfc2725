// Module-level entry points: MDTA (Attention.forward/backward, Restormer.py:99-132) and GDFN
// (FeedForward.forward/backward, Restormer.py:76-93) as fixed sequences of the three kernel
// archetypes (pointwise MFMA GEMM, pixel-axis Gram, depthwise stencil) plus the c x c glue.
// Host code only: carves the caller's saved/workspace blobs and enqueues kernels on the given stream.
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

// ------------------------------------------------------------------ MDTA
struct MdtaSaved {
  void* qkv0; void* qkv; float* A; float* P; float* nrm; float* M; size_t bytes;
};
static MdtaSaved mdta_saved_layout(const mi_mdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B, c = C / s->heads, Z = B * s->heads;
  Carver cv(base);
  MdtaSaved r;
  r.qkv0 = cv.take(tbytes(B * 3 * C * N, s->dtype));
  r.qkv = cv.take(tbytes(B * 3 * C * N, s->dtype));
  r.A = cv.take<float>(fbytes(Z * c * c));
  r.P = cv.take<float>(fbytes(Z * c * c));
  r.nrm = cv.take<float>(fbytes(Z * 2 * c));
  r.M = cv.take<float>(fbytes(B * C * C));
  r.bytes = cv.off;
  return r;
}

static mi_gram_desc mdta_qk_gram(const mi_mdta_shape* s, const void* qkv, float* graw, float* ss) {
  const int64_t N = (int64_t)s->H * s->W;
  const int C = s->C, c = C / s->heads;
  const size_t es = dtype_size(s->dtype);
  mi_gram_desc g;
  memset(&g, 0, sizeof(g));
  g.a = qkv; g.a_bs = 3 * (int64_t)C * N; g.a_gs = (int64_t)c * N; g.ma = c;
  g.b = (const char*)qkv + (size_t)C * N * es; g.b_bs = g.a_bs; g.b_gs = g.a_gs; g.mb = c;
  g.n = N; g.batch = s->B; g.groups = s->heads; g.dtype = s->dtype;
  g.sum_batch = 0; g.accumulate = 0; g.out = graw; g.out_ld = c; g.out_zs = (int64_t)c * c; g.sumsq = ss;
  return g;
}
static mi_gram_desc mdta_dm_gram(const mi_mdta_shape* s, const void* dout, const void* qkv, float* dM) {
  const int64_t N = (int64_t)s->H * s->W;
  const int C = s->C;
  const size_t es = dtype_size(s->dtype);
  mi_gram_desc g;
  memset(&g, 0, sizeof(g));
  g.a = dout; g.a_bs = (int64_t)C * N; g.ma = C;
  g.b = (const char*)qkv + (size_t)2 * C * N * es; g.b_bs = 3 * (int64_t)C * N; g.mb = C;
  g.n = N; g.batch = s->B; g.groups = 1; g.dtype = s->dtype;
  g.out = dM; g.out_ld = C; g.out_zs = (int64_t)C * C;
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

struct MdtaWs {
  // forward
  float* graw; float* ss; void* gram_ws; void* pw_ws; MdtaSaved inf;  // inf: saved-blob stand-in for inference
  // backward
  float* dM; float* dwo_part; float* dtemp_part; float* wdq; float* wdk; float* attn_scr; void* dqkv; void* dqkv0;
  void* dw_ws;
  void* cs_ws;
  size_t bytes;
};
static MdtaWs mdta_ws_layout(const mi_mdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B, c = C / s->heads, Z = B * s->heads;
  Carver cv(base);
  MdtaWs w;
  w.graw = cv.take<float>(fbytes(Z * c * c));
  w.ss = cv.take<float>(fbytes(Z * 2 * c));
  w.dM = cv.take<float>(fbytes(B * C * C));
  w.dwo_part = cv.take<float>(fbytes(B * C * C));
  w.dtemp_part = cv.take<float>(fbytes(Z));
  w.wdq = cv.take<float>(fbytes(Z * c * 2 * c));
  w.wdk = cv.take<float>(fbytes(Z * c * 2 * c));
  w.attn_scr = cv.take<float>(fbytes(attn_bwd_scratch_floats((int)B, (int)C, s->heads)));
  // gram scratch: max over the three contractions this module runs
  mi_gram_desc g1 = mdta_qk_gram(s, (void*)256, (float*)256, (float*)256);
  mi_gram_desc g2 = mdta_dm_gram(s, (void*)256, (void*)256, (float*)256);
  mi_gram_desc g3 = wgrad_gram((void*)256, 3 * (int)C, (void*)256, (int)C, (int)B, (int64_t)N, s->dtype, (float*)256, 0);
  size_t gw = max2(mi_gram_workspace(&g1), max2(mi_gram_workspace(&g2), mi_gram_workspace(&g3)));
  w.gram_ws = cv.take(gw);
  {  // weight-pack scratch of the largest 1x1 GEMM of this module (qkv: 3C x C; per-image c x 2c and C x C slices)
    mi_pw_desc a = conv1x1((void*)256, (int)C, (const float*)256, false, (int)C, nullptr, nullptr, (void*)256, 3 * (int)C,
                           (int)B, (int64_t)N, s->dtype);
    mi_pw_desc b = a;
    b.m = (int)C; b.w_bs = (int64_t)C * C;  // per-image C x C (M_b and its transpose)
    mi_pw_desc d = a;
    d.m = (int)c; d.k1 = (int)c; d.k2 = (int)c; d.x2 = (void*)256; d.groups = s->heads; d.w_bs = 1; d.w_gs = 1;
    mi_pw_desc e = conv1x1((void*)256, 3 * (int)C, (const float*)256, true, (int)C, nullptr, nullptr, (void*)256, (int)C,
                           (int)B, (int64_t)N, s->dtype);  // input gradient: W_qkv^T
    w.pw_ws = cv.take(max2(max2(mi_pw_gemm_workspace(&a), mi_pw_gemm_workspace(&e)),
                           max2(mi_pw_gemm_workspace(&b), mi_pw_gemm_workspace(&d))));
  }
  w.dw_ws = cv.take(mi_dwconv_bwd_workspace((int)B, 3 * (int)C, s->H, s->W, s->ks));
  w.cs_ws = cv.take(chan_sum_workspace(3 * (int)C, (int64_t)N));
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

// ------------------------------------------------------------------ GDFN
struct GdfnSaved { void* h0; void* h1; void* g; size_t bytes; };
static GdfnSaved gdfn_saved_layout(const mi_gdfn_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, B = s->B, h = s->hidden;
  Carver cv(base);
  GdfnSaved r;
  r.h0 = cv.take(tbytes(B * 2 * h * N, s->dtype));
  r.h1 = cv.take(tbytes(B * 2 * h * N, s->dtype));
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

extern "C" int mi_mdta_fwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* residual, void* out,
                           void* saved, void* ws, void* stream) {
  MI_TRY(mdta_check(s));
  MI_CHECK_ARG(p && x && out && ws, "mdta_fwd: null pointer");
  MI_CHECK_ARG(p->temperature && p->qkv_w && p->dw_w && p->proj_w, "mdta_fwd: null parameter");
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, C = s->C, dt = s->dtype;
  const int64_t N = (int64_t)s->H * s->W;
  const size_t es = dtype_size(dt);
  MdtaWs w = mdta_ws_layout(s, ws);
  MdtaSaved sv = saved ? mdta_saved_layout(s, saved) : w.inf;

  // qkv0 = qkv(x);  qkv = dw(qkv0)                               Restormer.py:114
  mi_pw_desc d1 = conv1x1(x, C, p->qkv_w, false, C, p->qkv_b, nullptr, sv.qkv0, 3 * C, B, N, dt);
  MI_TRY(mi_pw_gemm(&d1, w.pw_ws, stream));
  MI_TRY(mi_dwconv_fwd(sv.qkv0, p->dw_w, p->dw_b, sv.qkv, B, 3 * C, s->H, s->W, s->ks, dt, stream));
  // q k^T per head + row sums of squares                          Restormer.py:121-124
  mi_gram_desc g = mdta_qk_gram(s, sv.qkv, w.graw, w.ss);
  MI_TRY(mi_gram(&g, w.gram_ws, stream));
  // normalise, temperature, softmax, fold project_out             Restormer.py:124-125,131
  MI_TRY(launch_attn_fold(w.graw, w.ss, p->temperature, p->proj_w, sv.P, sv.A, sv.nrm, sv.M, B, C, s->heads, st));
  // out = M_b v (+bias) (+residual)                               Restormer.py:127-131
  mi_pw_desc d2 = conv1x1((const char*)sv.qkv + (size_t)2 * C * N * es, C, sv.M, false, C, p->proj_b, residual, out, C, B, N, dt);
  d2.x1_bs = 3 * (int64_t)C * N;
  d2.w_bs = (int64_t)C * C;
  MI_TRY(mi_pw_gemm(&d2, w.pw_ws, stream));
  return MI_OK;
}

extern "C" int mi_mdta_bwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* dout, void* dx,
                           const mi_mdta_grads* gr, const void* saved, void* ws, void* stream) {
  MI_TRY(mdta_check(s));
  MI_CHECK_ARG(p && x && dout && dx && gr && saved && ws, "mdta_bwd: null pointer");
  MI_CHECK_ARG(gr->temperature && gr->qkv_w && gr->dw_w && gr->proj_w, "mdta_bwd: null gradient buffer");
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, C = s->C, dt = s->dtype, hd = s->heads, c = C / hd, acc = gr->accumulate;
  const int64_t N = (int64_t)s->H * s->W;
  const size_t es = dtype_size(dt);
  MdtaWs w = mdta_ws_layout(s, ws);
  MdtaSaved sv = mdta_saved_layout(s, const_cast<void*>(saved));
  const char* q = (const char*)sv.qkv;
  const char* k = q + (size_t)C * N * es;
  char* dq = (char*)w.dqkv;
  char* dk = dq + (size_t)C * N * es;
  char* dv = dq + (size_t)2 * C * N * es;

  if (gr->proj_b) MI_TRY(launch_chan_sum(dout, gr->proj_b, B, C, N, dt, acc, w.cs_ws, st));
  // dM_b = dY V^T
  mi_gram_desc g1 = mdta_dm_gram(s, dout, sv.qkv, w.dM);
  MI_TRY(mi_gram(&g1, w.gram_ws, stream));
  MI_TRY(launch_attn_bwd_small(w.dM, sv.A, sv.P, sv.nrm, p->temperature, p->proj_w, w.dwo_part, w.dtemp_part, w.wdq, w.wdk,
                               w.attn_scr, B, C, hd, st));
  MI_TRY(launch_reduce_rows(w.dwo_part, gr->proj_w, B, (int64_t)C * C, (int64_t)C * C, acc, 1.0f, st));
  MI_TRY(launch_reduce_rows(w.dtemp_part, gr->temperature, B, hd, hd, acc, 1.0f, st));
  // dq = G1 k + D1 q ; dk = G1^T q + D2 k   (grouped over heads, per-image weights)
  mi_pw_desc dd;
  memset(&dd, 0, sizeof(dd));
  dd.x1 = k; dd.x1_bs = 3 * (int64_t)C * N; dd.x1_gs = (int64_t)c * N; dd.k1 = c;
  dd.x2 = q; dd.x2_bs = dd.x1_bs; dd.x2_gs = dd.x1_gs; dd.k2 = c;
  dd.w = w.wdq; dd.w_bs = (int64_t)hd * c * 2 * c; dd.w_gs = (int64_t)c * 2 * c; dd.w_sm = 2 * c; dd.w_sk = 1;
  dd.y = dq; dd.y_bs = 3 * (int64_t)C * N; dd.y_gs = (int64_t)c * N;
  dd.m = c; dd.n = N; dd.batch = B; dd.groups = hd; dd.dtype = dt;
  MI_TRY(mi_pw_gemm(&dd, w.pw_ws, stream));
  dd.x1 = q; dd.x2 = k; dd.w = w.wdk; dd.y = dk;
  MI_TRY(mi_pw_gemm(&dd, w.pw_ws, stream));
  // dv = M_b^T dY
  mi_pw_desc dvd = conv1x1(dout, C, sv.M, true, C, nullptr, nullptr, dv, C, B, N, dt);
  dvd.w_bs = (int64_t)C * C;
  dvd.y_bs = 3 * (int64_t)C * N;
  MI_TRY(mi_pw_gemm(&dvd, w.pw_ws, stream));
  // depthwise backward: d_qkv -> d_qkv0, weight/bias grads
  MI_TRY(mi_dwconv_bwd(w.dqkv, sv.qkv0, p->dw_w, w.dqkv0, gr->dw_w, gr->dw_b, B, 3 * C, s->H, s->W, s->ks, acc, dt, w.dw_ws,
                       stream));
  // qkv 1x1: weight grad (sum over batch), bias grad, input grad
  mi_gram_desc g2 = wgrad_gram(w.dqkv0, 3 * C, x, C, B, N, dt, gr->qkv_w, acc);
  MI_TRY(mi_gram(&g2, w.gram_ws, stream));
  if (gr->qkv_b) MI_TRY(launch_chan_sum(w.dqkv0, gr->qkv_b, B, 3 * C, N, dt, acc, w.cs_ws, st));
  mi_pw_desc dxd = conv1x1(w.dqkv0, 3 * C, p->qkv_w, true, C, nullptr, nullptr, dx, C, B, N, dt);
  MI_TRY(mi_pw_gemm(&dxd, w.pw_ws, stream));
  return MI_OK;
}

extern "C" size_t mi_gdfn_saved_bytes(const mi_gdfn_shape* s) {
  if (gdfn_check(s) != MI_OK) return 0;
  return gdfn_saved_layout(s, nullptr).bytes;
}
extern "C" size_t mi_gdfn_workspace(const mi_gdfn_shape* s) {
  if (gdfn_check(s) != MI_OK) return 0;
  return gdfn_ws_layout(s, nullptr).bytes;
}

extern "C" int mi_gdfn_fwd(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* residual, void* out,
                           void* saved, void* ws, void* stream) {
  MI_TRY(gdfn_check(s));
  MI_CHECK_ARG(p && x && out && ws, "gdfn_fwd: null pointer");
  MI_CHECK_ARG(p->in_w && p->dw_w && p->out_w, "gdfn_fwd: null parameter");
  const int B = s->B, C = s->C, h = s->hidden, dt = s->dtype;
  const int64_t N = (int64_t)s->H * s->W;
  GdfnWs w = gdfn_ws_layout(s, ws);
  GdfnSaved sv = saved ? gdfn_saved_layout(s, saved) : w.inf;
  mi_pw_desc d1 = conv1x1(x, C, p->in_w, false, C, p->in_b, nullptr, sv.h0, 2 * h, B, N, dt);       // Restormer.py:89
  MI_TRY(mi_pw_gemm(&d1, w.pw_ws, stream));
  MI_TRY(mi_dwconv_gate_fwd(sv.h0, p->dw_w, p->dw_b, saved ? sv.h1 : nullptr, sv.g, B, 2 * h, s->H, s->W, s->ks, dt,
                            stream));                                                                // :90-91
  mi_pw_desc d2 = conv1x1(sv.g, h, p->out_w, false, h, p->out_b, residual, out, C, B, N, dt);       // :92
  MI_TRY(mi_pw_gemm(&d2, w.pw_ws, stream));
  return MI_OK;
}

extern "C" int mi_gdfn_bwd(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* dout, void* dx,
                           const mi_gdfn_grads* gr, const void* saved, void* ws, void* stream) {
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
  MI_TRY(mi_dwconv_gate_bwd(w.dg, sv.h1, sv.h0, p->dw_w, w.dh0, gr->dw_w, gr->dw_b, B, 2 * h, s->H, s->W, s->ks, acc, dt,
                            w.dw_ws, stream));
  mi_gram_desc g2 = wgrad_gram(w.dh0, 2 * h, x, C, B, N, dt, gr->in_w, acc);
  MI_TRY(mi_gram(&g2, w.gram_ws, stream));
  if (gr->in_b) MI_TRY(launch_chan_sum(w.dh0, gr->in_b, B, 2 * h, N, dt, acc, w.cs_ws, st));
  mi_pw_desc d2 = conv1x1(w.dh0, 2 * h, p->in_w, true, C, nullptr, nullptr, dx, C, B, N, dt);
  MI_TRY(mi_pw_gemm(&d2, w.pw_ws, stream));
  return MI_OK;
}
