/*
 * mi_restore.h — C-ABI of the MI355X (gfx950) Restormer / MoCE-IR block library.
 *
 * Drop-in boundary for the hot path named in BASELINE.json: the reference has no
 * FFI of its own (it is pure PyTorch), so every entry point below replaces the
 * ATen/cuDNN/cuBLAS kernels that one reference nn.Module.forward()/autograd
 * backward launches.  The reference interface each entry point stands behind is
 * cited as file:line relative to the upstream repository root.
 *
 * Conventions:
 *  - plain pointers and sizes only; no torch types.  All pointers are DEVICE
 *    pointers unless said otherwise.  The caller owns every buffer (inputs,
 *    outputs, saved-for-backward blobs, workspaces); the library never
 *    allocates or frees device memory and keeps no pointer past return.
 *  - activations are NCHW contiguous, dtype MI_F32 or MI_BF16; parameters and
 *    parameter gradients are always fp32 in the reference's (PyTorch conv)
 *    layout: 1x1 conv [Cout,Cin], depthwise [C,k*k], LayerNorm [C].
 *  - every call takes the HIP stream to launch on (hipStream_t as void*).
 *  - return 0 on success, negative on failure; mi_last_error() gives the
 *    message (thread local).  No exceptions or aborts cross the ABI.
 *  - thread-safe: no global mutable state except the thread-local error string,
 *    the profiler and the opt-in packed-weight cache (mi_pw_cache_*), both mutex
 *    protected (backward is called from autograd worker threads).
 */
#ifndef MI_RESTORE_H
#define MI_RESTORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_RESTORE_VERSION 100

typedef enum { MI_F32 = 0, MI_BF16 = 1 } mi_dtype;

enum {
  MI_OK = 0,
  MI_ERR_ARG = -1,      /* bad shape / dtype / null pointer / unsupported size */
  MI_ERR_ALIGN = -2,    /* pointer not aligned as required */
  MI_ERR_HIP = -3       /* a HIP runtime call failed */
};

int mi_version(void);
const char* mi_last_error(void);
/* The MI_* A/B switches (environment variables consulted by the launch planners) are read once, at first use.  This re-reads
 * them: for tests and A/B tools that flip a switch inside one process.  Not meant to race with launches on other threads. */
int mi_env_reload(void);

/* Deferred parameter-gradient reductions.  Every weight gradient of the blocks ends in a fixed-order sum of partial rows into
 * the gradient buffer; as separate launches that was ~500 kernels of 5-15 us per training step.  Between mi_deferred_begin and
 * mi_deferred_end, backward entry points called with accumulate != 0 place those partials in the lent arena (device memory,
 * caller-owned, >= 2 MiB; a few GiB for a Restormer-base step - when it is full the entry points fall back to immediate sums)
 * and record the sum instead of launching it; mi_deferred_flush(stream) runs everything recorded so far in one table-driven
 * launch on `stream` (the stream the backward ran on), in a fixed order per gradient (bitwise reproducible), and
 * recycles the arena.  Contract: nothing reads such a gradient buffer between the backward call and the flush (the trainer
 * switches recording on in zero_grad() and flushes + switches it off before its all-reduce / optimizer step).  One context per
 * process. */
int mi_deferred_begin(void* arena, size_t bytes);
int mi_deferred_record(int on);         /* producers defer only while recording is on (the owner's backward window); begin() leaves it off */
int mi_deferred_pending(void);          /* reductions recorded and not yet flushed */
size_t mi_deferred_high_water(void);    /* most arena bytes in use at once since the library was loaded (sizing aid) */
int mi_deferred_flush(void* stream);
int mi_deferred_end(void);

/* ------------------------------------------------------------------------
 * Channel LayerNorm on NCHW  (Restormer.py:25-70; moce_ir.py:156-221;
 * AdaIR-main/net/model.py:25-71).  with_bias=1: (x-mu)/sqrt(var+1e-5)*w+b ;
 * with_bias=0 ("BiasFree"): x/sqrt(var+1e-5)*w, var about the mean.
 * mean/rstd: [B, H*W] fp32, written by fwd (may be NULL), read by bwd.
 * bwd: dx = LN'(dy) (+ dres if non-NULL);  dw/db are accumulated (+=) when
 * accumulate!=0, else overwritten.  ws: mi_ln_bwd_workspace() bytes.
 * ------------------------------------------------------------------------ */
int mi_ln_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd,
              int B, int C, int64_t N, int with_bias, int dtype, void* stream);
size_t mi_ln_bwd_workspace(int B, int C, int64_t N);
int mi_ln_bwd(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
              const void* dres, void* dx, float* dw, float* db,
              int B, int C, int64_t N, int with_bias, int accumulate, int dtype,
              void* ws, void* stream);

/* ------------------------------------------------------------------------
 * Depthwise k x k convolution, stride 1, zero pad k/2, groups = channels
 * (Restormer.py:84,106 3x3; moce_ir.py:341 7x7).  w: [C, k*k], bias [C] or NULL.
 *  fwd      : y = dw(x)
 *  fwd_gate : C = 2h channels; y (may be NULL) = dw(x) [B,2h,H,W];
 *             g = gelu_erf(y[:, :h]) * y[:, h:]  [B,h,H,W]   (Restormer.py:90-91)
 *  bwd      : dx = dw^T(dy);  dw/db accumulated or overwritten.
 *  bwd_gate : same, with dy of the 2h conv outputs formed on the fly from
 *             (dg [B,h,H,W], y [B,2h,H,W]) — the GELU-gate backward.
 * ------------------------------------------------------------------------ */
int mi_dwconv_fwd(const void* x, const float* w, const float* bias, void* y,
                  int B, int C, int H, int W, int ks, int dtype, void* stream);
int mi_dwconv_gate_fwd(const void* x, const float* w, const float* bias, void* y, void* g,
                       int B, int C2, int H, int W, int ks, int dtype, void* stream);
size_t mi_dwconv_bwd_workspace(int B, int C, int H, int W, int ks);
int mi_dwconv_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* db,
                  int B, int C, int H, int W, int ks, int accumulate, int dtype, void* ws, void* stream);
int mi_dwconv_gate_bwd(const void* dg, const void* y, const void* x, const float* w, void* dx, float* dw,
                       float* db, int B, int C2, int H, int W, int ks, int accumulate, int dtype,
                       void* ws, void* stream);
/* GDFN gate backward WITHOUT stored conv outputs: y1, y2 are recomputed from the conv input x inside the kernel (the
 * input rows are needed for the weight gradient anyway), so the forward need not write y at all (mi_dwconv_gate_fwd with
 * y = NULL) - a third less HBM traffic over the depthwise stage and 2h fewer saved planes.  Available for the shapes
 * mi_dwconv_gate_recompute_ok() reports (3x3, rows of 16..256 pixels, power of two); same workspace as mi_dwconv_gate_bwd. */
int mi_dwconv_gate_recompute_ok(int H, int W, int ks);
int mi_dwconv_gate_bwd_recompute(const void* dg, const void* x, const float* w, const float* bias, void* dx, float* dw,
                                 float* db, int B, int C2, int H, int W, int ks, int accumulate, int dtype, void* ws,
                                 void* stream);

/* ------------------------------------------------------------------------
 * Pointwise (1x1 conv) GEMM on N-contiguous planes (Restormer.py:82,86,105,107):
 *   Y[z][m][n] = sum_k Wz(m,k) * X[z][k][n]  (+ bias[m]) (+ R[z][m][n])
 * z = batch*groups + group.  X may be given as two K-panels (X1 rows 0..K1-1,
 * X2 rows K1..K1+K2-1): concat-free 1x1 over a channel concat.
 * Weight element (m,k) of slice z lives at W[b*w_bs + g*w_gs + m*w_sm + k*w_sk]
 * (w_sk==1: row-major [M,K]; w_sm==1: the transpose of a [K,M] matrix).
 * Strides are in elements.  ws: mi_pw_gemm_workspace() bytes (16-byte aligned); it receives the
 * weights re-packed into the kernel's LDS image (activation dtype), written and read by this call only.
 * ------------------------------------------------------------------------ */
typedef struct {
  const void* x1; int64_t x1_bs, x1_gs; int k1;
  const void* x2; int64_t x2_bs, x2_gs; int k2;   /* x2 may be NULL (k2 = 0) */
  const float* w; int64_t w_bs, w_gs, w_sm, w_sk;
  const float* bias; int64_t bias_gs;             /* may be NULL */
  const void* r; int64_t r_bs, r_gs;              /* residual, may be NULL */
  void* y; int64_t y_bs, y_gs;
  int m; int64_t n; int batch; int groups; int dtype;
  /* Optional LayerNorm over the K channels of X1, applied as the tile is loaded (Restormer.py:27-70 in front of :89 / :115):
   * ln_mode 0 = none, 1 = WithBias ((x - mu) rstd w + b), 2 = BiasFree (x rstd w); ln_w / ln_b [K]; ln_mean / ln_rstd
   * [batch, n] receive the statistics (both or neither).  Only where mi_pw_gemm_ln_ok() says so (the X-resident kernels:
   * bf16; 96 < M with K <= 96, or 256 <= M with K <= 128; one K panel, one group); zero-initialise the struct to leave it off. */
  const float* ln_w; const float* ln_b; float* ln_mean; float* ln_rstd; int ln_mode;
  /* Optional fp8 (OCP e4m3) MFMA operands - the "CDNA4 fp8 MFMA projections" of the tiled-inference configuration (BASELINE
   * configs[4]); inference only.  f8 = 1: X (after the optional LayerNorm) is divided by f8_sx and W (its packed bf16 image)
   * by f8_sw, both are rounded to e4m3 in registers (saturating at +-448) and multiplied with v_mfma_f32_16x16x32_fp8_fp8; the fp32 accumulator
   * is scaled back by f8_sx * f8_sw before bias / residual.  The hardware conversion uses the scales' exponents only: pass
   * powers of two, chosen so that |X| / f8_sx and |W| / f8_sw stay <= 448.  X, Y and the residual stay bf16 in HBM.  Only
   * where mi_pw_gemm_f8_ok() says so (the wave-owned bf16 kernels); zero-initialise the struct to leave it off. */
  int f8; float f8_sx, f8_sw;
  /* Optional second output: rows m >= y_split of the result are written to y2 (row m - y_split), the rows below to y - two
   * results of ONE pass over X (the q and k gradients of MDTA, which multiply the same q, k planes with two per-image
   * matrices).  No residual; only where mi_pw_gemm_split_ok() says so (the wave-owned bf16 kernels); zero = off. */
  int y_split; void* y2; int64_t y2_bs, y2_gs;
  /* Optional bf16 copy of PER-IMAGE weights (w_bs != 0): the same matrix values as `w`, already rounded to bf16, row-major
   * [m][k] with unit k stride and row stride w_b16_sm (elements), batch / group strides w_bs / w_gs as for `w` (all multiples of
   * 8, 16-byte aligned).  The wave-owned bf16 kernels then stage the weights straight from it and no pack launch runs (the
   * producers of such matrices - the c x c attention fold, the q / k gradient matrices - write both).  `w` must stay valid: the
   * other kernel forms pack from it.  NULL = off. */
  const void* w_b16; int64_t w_b16_sm;
} mi_pw_desc;
size_t mi_pw_gemm_workspace(const mi_pw_desc* d);
int mi_pw_gemm(const mi_pw_desc* d, void* ws, void* stream);
int mi_pw_gemm_ln_ok(const mi_pw_desc* d);
int mi_pw_gemm_f8_ok(const mi_pw_desc* d);
int mi_pw_gemm_split_ok(const mi_pw_desc* d);

/* Opt-in packed-weight cache.  By default every mi_pw_gemm (and every module entry point built on it) packs its weight
 * matrix into its own workspace, once per call, and the library keeps no state.  A caller that controls when the
 * weights change (a trainer: only at the optimizer step; an inference server: never) may lend a device buffer:
 *   mi_pw_cache_enable(buf, bytes, lo, hi)  buf stays owned by the caller and must outlive the cache; NULL disables it.
 *                                   Only weights whose address lies in [lo, hi) - the caller's parameter storage - are
 *                                   cached: a temporary weight matrix (a permuted copy, ...) may reuse an address with
 *                                   different values and always packs per call.
 *   mi_pw_cache_refresh(stream)     re-packs EVERY weight seen so far in one launch; from then on calls with those
 *                                   weights skip their pack launch.  Call it after each optimizer step (the weights
 *                                   registered since the last refresh are picked up; the first refresh after new
 *                                   registrations must run outside stream capture).
 *   mi_pw_cache_invalidate()        packed images are stale (weights were overwritten some other way): calls pack
 *                                   per call again until the next refresh.
 * Per-image weights (w_bs != 0) are never cached.  The cache is process-global and mutex-protected; this is the only
 * mutable state in the library besides the thread-local error string and the profiler.                              */
int mi_pw_cache_enable(void* buf, size_t bytes, const void* params_lo, const void* params_hi);
int mi_pw_cache_refresh(void* stream);
int mi_pw_cache_invalidate(void);
int mi_pw_cache_pending(void);   /* 1 while weights registered since the last refresh (or an invalidation) wait for one */

/* ------------------------------------------------------------------------
 * Row-Gram (reduction over the pixel axis), the contraction of MDTA's q k^T
 * (Restormer.py:124) and of every 1x1-conv weight gradient:
 *   G[z][i][j] = sum_n A[z][i][n] * Bm[z][j][n]
 * optionally summed over the batch index, optionally also the row sums of
 * squares of A and Bm (for F.normalize, Restormer.py:121-122).
 * Split over n across workgroups with a deterministic two-stage reduce.
 * out = (accumulate? out : 0) + G.   ws: mi_gram_workspace() bytes.
 * ------------------------------------------------------------------------ */
typedef struct {
  const void* a; int64_t a_bs, a_gs; int ma;
  const void* b; int64_t b_bs, b_gs; int mb;
  int64_t n; int batch; int groups; int dtype;
  int sum_batch;           /* 1: out is [groups, ma, mb] summed over batch */
  int accumulate;          /* 1: out += */
  float* out;              /* [batch*groups or groups][ma][mb]; row stride out_ld (>= mb) */
  int64_t out_ld, out_zs;  /* row stride / slice stride of out, elements */
  float* sumsq;            /* NULL or [batch*groups][ma+mb] */
} mi_gram_desc;
size_t mi_gram_workspace(const mi_gram_desc* d);
int mi_gram(const mi_gram_desc* d, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * MDTA — Attention.forward / backward  (Restormer.py:99-132; moce_ir.py:283-321;
 * AdaIR-main/net/model.py:99-130).  out = (residual?) + project_out(attn(qkv_dw(qkv(x)))).
 * saved: mi_mdta_saved_bytes() blob written by fwd when non-NULL (training),
 * read by bwd.  ws: mi_mdta_workspace() bytes scratch.
 * Gradients are accumulated (+=) when accumulate!=0, else overwritten; NULL
 * bias pointers mean the conv has no bias.
 * ------------------------------------------------------------------------ */
typedef struct { int B, C, heads, H, W, dtype, ks; } mi_mdta_shape;
typedef struct {
  const float* temperature;                 /* [heads] */
  const float* qkv_w;  const float* qkv_b;  /* [3C,C], [3C]|NULL */
  const float* dw_w;   const float* dw_b;   /* [3C,ks*ks], [3C]|NULL */
  const float* proj_w; const float* proj_b; /* [C,C], [C]|NULL */
} mi_mdta_params;
typedef struct {
  float* temperature; float* qkv_w; float* qkv_b; float* dw_w; float* dw_b; float* proj_w; float* proj_b;
  int accumulate;
} mi_mdta_grads;
size_t mi_mdta_saved_bytes(const mi_mdta_shape* s);
size_t mi_mdta_workspace(const mi_mdta_shape* s);
int mi_mdta_fwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* residual,
                void* out, void* saved, void* ws, void* stream);
int mi_mdta_bwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* x, const void* dout,
                void* dx, const mi_mdta_grads* g, const void* saved, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * Cross-MDTA: q from x (1x1 C->C + depthwise ks_q), k,v from y (1x1 C->2C + depthwise ks_kv), then the
 * MDTA core.  moce_ir.py:325-368 CrossAttention (ks_q 3, ks_kv 7); AdaIR-main/net/model.py:177-216
 * Chanel_Cross_Attention (3, 3).  bwd returns dx and dy.
 * ------------------------------------------------------------------------ */
typedef struct { int B, C, heads, H, W, dtype, ks_q, ks_kv; } mi_xmdta_shape;
typedef struct {
  const float* temperature;
  const float* q_w;  const float* q_b;  const float* q_dw_w;  const float* q_dw_b;     /* [C,C] [C] [C,ks_q^2] [C] */
  const float* kv_w; const float* kv_b; const float* kv_dw_w; const float* kv_dw_b;    /* [2C,C] [2C] [2C,ks_kv^2] [2C] */
  const float* proj_w; const float* proj_b;
} mi_xmdta_params;
typedef struct {
  float* temperature; float* q_w; float* q_b; float* q_dw_w; float* q_dw_b; float* kv_w; float* kv_b; float* kv_dw_w;
  float* kv_dw_b; float* proj_w; float* proj_b; int accumulate;
} mi_xmdta_grads;
size_t mi_xmdta_saved_bytes(const mi_xmdta_shape* s);
size_t mi_xmdta_workspace(const mi_xmdta_shape* s);
int mi_xmdta_fwd(const mi_xmdta_shape* s, const mi_xmdta_params* p, const void* x, const void* y,
                 const void* residual, void* out, void* saved, void* ws, void* stream);
int mi_xmdta_bwd(const mi_xmdta_shape* s, const mi_xmdta_params* p, const void* x, const void* y, const void* dout,
                 void* dx, void* dy, const mi_xmdta_grads* g, const void* saved, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * GDFN — FeedForward.forward / backward (Restormer.py:76-93; moce_ir.py:255-276;
 * AdaIR-main/net/model.py:76-94).  hidden = h (project_in has 2h outputs).
 * ------------------------------------------------------------------------ */
/* flags bit 0: the forward also stores the depthwise conv's output for backward (no recompute; chosen by the caller and
 * passed unchanged to mi_gdfn_saved_bytes / _fwd / _bwd of one forward-backward pair: it fixes the saved blob's layout) */
typedef struct { int B, C, hidden, H, W, dtype, ks, flags; } mi_gdfn_shape;
typedef struct {
  const float* in_w;  const float* in_b;   /* [2h,C], [2h]|NULL */
  const float* dw_w;  const float* dw_b;   /* [2h,ks*ks], [2h]|NULL */
  const float* out_w; const float* out_b;  /* [C,h], [C]|NULL */
} mi_gdfn_params;
typedef struct { float* in_w; float* in_b; float* dw_w; float* dw_b; float* out_w; float* out_b; int accumulate; } mi_gdfn_grads;
size_t mi_gdfn_saved_bytes(const mi_gdfn_shape* s);
size_t mi_gdfn_workspace(const mi_gdfn_shape* s);
int mi_gdfn_fwd(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* residual,
                void* out, void* saved, void* ws, void* stream);
int mi_gdfn_bwd(const mi_gdfn_shape* s, const mi_gdfn_params* p, const void* x, const void* dout,
                void* dx, const mi_gdfn_grads* g, const void* saved, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * Fused GDFN half-block (bf16 activations):  out = y + FeedForward(LayerNorm(y))
 * = the second half of TransformerBlock.forward (Restormer.py:148 `x = x + self.ffn(self.norm2(x))`;
 * moce_ir.py:834 EncoderBlock; AdaIR-main/net/model.py:170) as ONE kernel: y is read once (plus a
 * one-pixel halo), out is written once; LN output, project_in output, conv outputs and the gate never
 * reach HBM.  Covered shapes (mi_gdfn_fused_ok): 3x3 depthwise, W % 64 == 0, H % 8 == 0, C = 48 or 96;
 * any hidden size.  Other shapes use mi_ln_fwd + mi_gdfn_fwd.
 *   mi_gdfn_fused_pack : LayerNorm weight/bias + the six GDFN parameters -> the kernel's packed bf16/fp32
 *                        weight images (LN affine folded into project_in).  Re-run after every weight update.
 *   mi_gdfn_fused_fwd  : y, out [B,C,H,W] bf16; mean/rstd [B,H*W] fp32 (both or neither; LN statistics of y
 *                        for the backward pass).
 * ------------------------------------------------------------------------ */
typedef struct { int B, C, hidden, H, W, ln_with_bias; } mi_gdfn_fused_shape;
int mi_gdfn_fused_ok(const mi_gdfn_fused_shape* s);
size_t mi_gdfn_fused_pack_bytes(const mi_gdfn_fused_shape* s);
int mi_gdfn_fused_pack(const mi_gdfn_fused_shape* s, const float* ln_w, const float* ln_b, const mi_gdfn_params* p,
                       void* pack, void* stream);
int mi_gdfn_fused_fwd(const mi_gdfn_fused_shape* s, const void* pack, const void* y, void* out, float* mean,
                      float* rstd, void* stream);
/* Training form of the same launch: besides out and the statistics it writes what mi_gdfn_bwd / mi_gdfn_bwd_ln read - the
 * project_in output h0 [B,2h,H,W] and the gate output g [B,h,H,W] - into `saved`, a blob of mi_gdfn_saved_bytes() bytes for
 * the mi_gdfn_shape {B, C, hidden, H, W, MI_BF16, 3, flags 0}: the forward of `x + ffn(norm2(x))` (Restormer.py:148) is ONE
 * launch (10 C planes per pixel at hidden = 2.66 C) instead of GEMM -> depthwise gate -> GEMM (19).  _ok: covered shape. */
int mi_gdfn_fused_fwd_train_ok(const mi_gdfn_fused_shape* s);
int mi_gdfn_fused_fwd_train(const mi_gdfn_fused_shape* s, const void* pack, const void* y, void* out, float* mean,
                            float* rstd, void* saved, void* stream);

/* ------------------------------------------------------------------------
 * Backward tail of a half-block  out = x + F(LN(x)),  F starting in the 1x1 conv h = W LN(x)
 * (TransformerBlock.forward, Restormer.py:146-150; qkv :105,115; project_in :82,89; WithBias_LayerNorm :52-64).
 * One launch computes, from dY = dL/dh:  dW += / = dY LN(x)^T,  dxn = W^T dY,  dx = LNbackward(dxn) + dres,
 * dgamma, dbeta - dY, x and dres are read once, dx written once; LN(x) is rebuilt from x and the saved statistics.
 * bf16 activations, C in {48, 96}, M <= 256 (C = 48) / 512 (C = 96), H*W a multiple of 64.
 *   mi_bwd_tail        : the kernel by itself.  dy [B,M,N], x (LayerNorm INPUT) / dres / dx [B,C,N], mean/rstd [B,N],
 *                        w [M,C], gamma/beta [C]; dw [M,C], dgamma, dbeta [C] (accumulate: += instead of =).
 *   mi_mdta_bwd_ln, mi_gdfn_bwd_ln : mi_mdta_bwd / mi_gdfn_bwd with that tail: x is the LayerNorm INPUT, dx the gradient of
 *                        the half-block's input; workspace from mi_*_bwd_ln_workspace (0 = shape not covered, see *_ok).
 * ------------------------------------------------------------------------ */
typedef struct {
  const float* w; const float* b;        /* LayerNorm weight, bias [C] */
  const float* mean; const float* rstd;  /* [B, H*W], from mi_ln_fwd */
  const void* dres;                      /* [B,C,H,W] gradient arriving over the residual connection, or NULL */
  float* dw; float* db;                  /* LayerNorm parameter gradients [C] */
} mi_ln_tail;
/* LayerNorm in front of the half-block's first 1x1 conv, applied inside that GEMM (the normalised tensor never reaches HBM):
 * mi_mdta_fwd_ln / mi_gdfn_fwd_ln = mi_mdta_fwd / mi_gdfn_fwd with x the LayerNorm INPUT.  mean / rstd [B, H*W] receive the
 * statistics for the backward pass (both or neither).  Shapes: mi_*_fwd_ln_ok (bf16, C <= 128, H*W a multiple of 64). */
typedef struct { const float* w; const float* b; float* mean; float* rstd; int with_bias; } mi_ln_head;
int mi_mdta_fwd_ln_ok(const mi_mdta_shape* s);
int mi_mdta_fwd_ln(const mi_mdta_shape* s, const mi_mdta_params* p, const mi_ln_head* ln, const void* x, const void* residual,
                   void* out, void* saved, void* ws, void* stream);
int mi_gdfn_fwd_ln_ok(const mi_gdfn_shape* s);
int mi_gdfn_fwd_ln(const mi_gdfn_shape* s, const mi_gdfn_params* p, const mi_ln_head* ln, const void* x, const void* residual,
                   void* out, void* saved, void* ws, void* stream);
/* fp8 (e4m3) MFMA operands in both 1x1 projections of a half-block - inference only (nothing is saved), bf16 activations in
 * HBM; see mi_pw_desc.f8.  x1 / w1 scale the first projection's input and weight (qkv: Restormer.py:105,114; project_in:
 * :82,89), x2 / w2 the second's (project_out, :107,131 - for MDTA its input is v and its weight the per-image product
 * project_out . softmax(..), |entries| <= (C / heads) max|project_out|; :86,92 for GDFN, input gelu(x1) x2).  Powers of two
 * with |operand| / scale <= 448 (values past that saturate).  ln may be NULL (x is then the LayerNorm output).
 * Shapes: mi_*_fwd_f8_ok(shape, with_ln). */
typedef struct { float x1, w1, x2, w2; } mi_f8_scales;
int mi_mdta_fwd_f8_ok(const mi_mdta_shape* s, int with_ln);
int mi_mdta_fwd_f8(const mi_mdta_shape* s, const mi_mdta_params* p, const mi_ln_head* ln, const mi_f8_scales* f8, const void* x,
                   const void* residual, void* out, void* ws, void* stream);
int mi_gdfn_fwd_f8_ok(const mi_gdfn_shape* s, int with_ln);
int mi_gdfn_fwd_f8(const mi_gdfn_shape* s, const mi_gdfn_params* p, const mi_ln_head* ln, const mi_f8_scales* f8, const void* x,
                   const void* residual, void* out, void* ws, void* stream);
/* mi_gdfn_fused_fwd (the one-launch LayerNorm + GDFN half-block) on fp8 operands: x1 scales the NORMALISED input
 * ((y - mu) rstd, |.| <= sqrt(C)) and w1 the packed W_in . diag(gamma); x2 / w2 as above.  Default tile forms only. */
int mi_gdfn_fused_fwd_f8(const mi_gdfn_fused_shape* s, const void* pack, const mi_f8_scales* f8, const void* y, void* out,
                         void* stream);

/* ------------------------------------------------------------------------
 * Fused MDTA, pass A (csrc/fused_mdta.hip; Restormer.py:111-122): LayerNorm -> qkv 1x1 -> depthwise 3x3 -> q k^T partials
 * + row sums of squares + v in ONE launch - x is read once, only v is written; qkv0, q and k never reach HBM.  mi_mdta_fused_fwd
 * runs the whole half-block on it: pass A, the fixed-order sum of the per-workgroup partials, the c x c softmax / W_o fold and
 * the per-image-weight GEMM  out = (W_o . blockdiag(A)) v (+ proj bias) (+ residual).  bf16, 3x3, C = 48 (1 head) or
 * 96 (1 or 2 heads), W % 64 == 0, H % 8 == 0 (mi_mdta_fused_ok); nothing is saved for backward (the no_grad path).
 *  pack : mi_mdta_fused_pack_bytes() bytes, written by mi_mdta_fused_pack from the LayerNorm affine (folded into W_qkv), the
 *         qkv / depthwise weights and biases; re-pack after every weight update.
 *  ln_with_bias : 1 WithBias ((x - mu) rstd), 0 BiasFree (x rstd, not centred).   mean / rstd : optional [B][H*W] outputs.
 *  ws   : mi_mdta_fused_workspace() bytes (v, the partials, the c x c matrices).
 * ------------------------------------------------------------------------ */
int mi_mdta_fused_ok(const mi_mdta_shape* s);
int mi_mdta_fused_pays(const mi_mdta_shape* s);   /* covered and enough workgroups (B x splits >= 192) to beat the unfused chain */
size_t mi_mdta_fused_pack_bytes(const mi_mdta_shape* s);
int mi_mdta_fused_pack(const mi_mdta_shape* s, const float* ln_w, const float* ln_b, const mi_mdta_params* p, void* pack,
                       void* stream);
size_t mi_mdta_fused_workspace(const mi_mdta_shape* s);
int mi_mdta_fused_fwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* pack, int ln_with_bias, const void* x,
                      const void* residual, void* out, float* mean, float* rstd, void* ws, void* stream);
int mi_bwd_tail_ok(int M, int C, int64_t N, int dtype);
size_t mi_bwd_tail_workspace(int M, int C);
int mi_bwd_tail(const void* dy, int M, const void* x, int C, const void* dres, const float* mean, const float* rstd,
                const float* w, const float* gamma, const float* beta, void* dx, float* dw, float* dgamma, float* dbeta,
                int B, int64_t N, int accumulate, int dtype, void* ws, void* stream);
int mi_mdta_bwd_ln_ok(const mi_mdta_shape* s, int qkv_bias);
size_t mi_mdta_bwd_ln_workspace(const mi_mdta_shape* s);
int mi_mdta_bwd_ln(const mi_mdta_shape* s, const mi_mdta_params* p, const mi_ln_tail* ln, const void* x, const void* dout,
                   void* dx, const mi_mdta_grads* gr, const void* saved, void* ws, void* stream);
int mi_gdfn_bwd_ln_ok(const mi_gdfn_shape* s, int in_bias);
size_t mi_gdfn_bwd_ln_workspace(const mi_gdfn_shape* s);
int mi_gdfn_bwd_ln(const mi_gdfn_shape* s, const mi_gdfn_params* p, const mi_ln_tail* ln, const void* x, const void* dout,
                   void* dx, const mi_gdfn_grads* gr, const void* saved, void* ws, void* stream);

/* ------------------------------------------------------------------------
 * AdaIR frequency modules (AdaIR-main/net/model.py:230-372): the pieces of SpatialGate / ChannelGate / FreRefine / FreModule
 * that are not a block, a cross attention or a convolution (csrc/adair.hip).  Activations [B,C,H,W] in `dtype`, N = H*W.
 *   mi_box_down          F.interpolate(img, (H, W), 'bilinear') for the integer factors of the U-Net levels (model.py:321)
 *   mi_fre_rect          half sizes [B,2] of the low-frequency rectangle: int(h/n * sigmoid(rate_conv(avgpool))) (model.py:346-353)
 *   mi_fre_split_fwd/bwd FreModule.fft (model.py:343-372): high = |x - Px|, low = |Px|, P = the projection on the rectangle's
 *                        frequencies evaluated as a direct DFT (no FFT); half == NULL: empty rectangle for every sample.
 *                        coef: mi_fre_split_coef_bytes (kept for the backward); ws: mi_fre_split_workspace.
 *   mi_chan_maxmean_*    [max_c x, mean_c x] -> [B,2,H,W] + arg max [B,N] (model.py:240-242)
 *   mi_plane_max_fwd     global max pool + arg max per plane; mi_pool_pair_bwd: gradient of avg pool + max pool in one pass (model.py:250-264)
 *   mi_chan_gate_*       sigmoid(mlp(avg) + mlp(max)), mlp = W2 relu(W1 .) (model.py:253-268); hid [B,2,R] pre-activations
 *   mi_refine_mix_*      low * sigmoid(s0 + s1) + high * cw (model.py:284-288), s [B,2,H,W] = depthwise 7x7 of the max/mean planes
 *   mi_scale_add_*       a * p1[c] + y * p2[c] (model.py:331) and the parameter gradients
 * ------------------------------------------------------------------------ */
int mi_box_down(const void* img, void* out, int B, int C, int Hi, int Wi, int factor, int dtype, void* stream);
int mi_fre_rect(const float* pooled, const float* w0, const float* w2, int* half, int B, int C, int R, int H, int W, int n,
                void* stream);
size_t mi_fre_split_coef_bytes(int B, int C);
size_t mi_fre_split_workspace(int B, int C, int H, int W);
int mi_fre_split_max_hw(void);
int mi_fre_split_fwd(const void* feat, const int* half, void* high, void* low, float* coef, int B, int C, int H, int W, int dtype,
                     void* stream);
int mi_fre_split_bwd(const void* feat, const int* half, const float* coef, const void* dhigh, const void* dlow, void* dfeat, int B,
                     int C, int H, int W, int dtype, void* ws, void* stream);
int mi_chan_maxmean_fwd(const void* x, void* out, int* idx, int B, int C, int64_t N, int dtype, void* stream);
int mi_chan_maxmean_bwd(const void* dout, const int* idx, void* dx, int B, int C, int64_t N, int dtype, void* stream);
int mi_plane_max_fwd(const void* x, float* out, int* idx, int planes, int64_t N, int dtype, void* stream);
int mi_pool_pair_bwd(const float* davg, const float* dmax, const int* idx, void* dx, int planes, int64_t N, int dtype, void* stream);
int mi_chan_gate_fwd(const float* avg, const float* mx, const float* w1, const float* w2, float* cw, float* hid, int B, int C, int R,
                     void* stream);
int mi_chan_gate_bwd(const float* avg, const float* mx, const float* w1, const float* w2, const float* cw, const float* hid,
                     const float* dcw, float* davg, float* dmx, float* dw1, float* dw2, int B, int C, int R, int accumulate,
                     void* stream);
int mi_refine_mix_fwd(const void* low, const void* high, const void* s, const float* cw, void* out, int B, int C, int64_t N,
                      int dtype, void* stream);
int mi_refine_mix_bwd(const void* low, const void* high, const void* s, const float* cw, const void* dout, void* dlow, void* dhigh,
                      void* ds, float* dcw, int B, int C, int64_t N, int dtype, void* stream);
int mi_scale_add_fwd(const void* a, const void* y, const float* p1, const float* p2, void* out, int B, int C, int64_t N, int dtype,
                     void* stream);
int mi_scale_add_bwd(const void* a, const void* y, const float* p1, const float* p2, const void* dout, void* da, void* dy, float* dp1,
                     float* dp2, int B, int C, int64_t N, int accumulate, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Training-step tail on flat fp32 buffers (MoCE-IR-main/src/train.py:79-88:
 * AdamW(lr=2e-4), torch defaults betas (0.9,0.999), eps 1e-8, weight_decay 1e-2).
 * p, g, m, v: [n] fp32.  grad_scale multiplies g first (1/world for DDP mean).
 * dev_scalars (device, may be NULL): {lr, 1-beta1^step, sqrt(1-beta2^step)} read by the kernel
 * instead of the host lr/step, so that a captured HIP graph can be replayed with new values.
 * ------------------------------------------------------------------------ */
int mi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale,
                  const float* dev_scalars, void* stream);

/* ------------------------------------------------------------------------
 * MoCE SparseDispatcher data movement (moce_ir.py:71-143).  A "row" is one sample's feature map (C*H*W elements);
 * idx is the dispatcher's _batch_index (int64, device), scale its _nonzero_gates (fp32, device).
 *   rows_gather      : out[i] = x[idx[i]]                                    (dispatch, :103)
 *   rows_scatter_add : out[b] = sum_{i: idx[i]==b} scale[i] * src[i]          (combine: gate multiply + index_add into
 *                      zeros, fp32 accumulation, :116-124; scale NULL = 1; out is fp32 or the activation dtype)
 *   rows_gather_scaled: out[i] = scale[i] * x[idx[i]], fp32 x -> activation dtype (gradient of combine w.r.t. expert outputs)
 *   rows_dot         : out[i] = <g[idx[i]], src[i]>                           (gradient of the gate values)
 * The scatter is per destination row in list order: deterministic, no atomics.
 * ------------------------------------------------------------------------ */
int mi_rows_gather(const void* x, const int64_t* idx, void* out, int n_out, int64_t row, int dtype, void* stream);
int mi_rows_gather_scaled(const float* x, const int64_t* idx, const float* scale, void* out, int n_out, int64_t row,
                          int out_dtype, void* stream);
int mi_rows_scatter_add(const void* src, const int64_t* idx, const float* scale, void* out, int n_src, int n_rows,
                        int64_t row, int dtype, int out_f32, void* stream);
size_t mi_rows_dot_workspace(int n_src, int64_t row);
int mi_rows_dot(const float* g, const void* src, const int64_t* idx, float* out, int n_src, int64_t row, int dtype,
                void* ws, void* stream);

/* ------------------------------------------------------------------------
 * Grouped expert GEMM (csrc/grouped.hip): the 1x1 projections of all MoCE experts in ONE launch over a problem table whose
 * ragged part - rows per expert, segment starts - is read from DEVICE memory (the router's counts / offsets tables), so no
 * segment size has to reach the host for this stage.  Replaces the per-expert 1x1 calls of moce_ir.py:545-558 (ModExpert.process:
 * proj[0], proj[1], proj[2]) as iterated by :666-672, whose split sizes the reference reads back with .tolist() (:88).
 *   for every problem p, for i in [0, counts[p.expert]):
 *       Y_p[row_y] = W_p . X_p[row_x] (+ bias_p) (+ R_p[row_r]),   row_* = i (local) or offsets[p.expert] + i (stitched buffers)
 *   X rows are [K][N], Y / R rows [M][N] (N = H*W pixels, contiguous); W element (m, k) at w[m*w_sm + k*w_sk] (fp32; a
 *   transposed use - the input gradient W^T dY - is just swapped strides); *_rs row strides in elements (0 = dense).
 * Up to 16 problems per launch; the grid covers max_rows rows per problem, workgroups beyond an expert's count exit at once.
 * ------------------------------------------------------------------------ */
typedef struct {
  const void* x; int64_t x_rs;
  const float* w; int64_t w_sm, w_sk;
  const float* bias;
  const void* r; int64_t r_rs;
  void* y; int64_t y_rs;
  int m, k;
  int expert;                     /* index into dev_counts / dev_offsets */
  int x_local, y_local, r_local;  /* 1: the buffer is indexed by the row's position inside the expert's segment */
} mi_grouped_problem;
int mi_grouped_pw_gemm(const mi_grouped_problem* probs, int np, const int* dev_counts, const int* dev_offsets, int max_rows,
                       int64_t N, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * MoCE router in one launch each way (moce_ir.py:684-800 RoutingFunction; :82-91 SparseDispatcher index bookkeeping).
 *   fwd: logits = pooled . Wg^T + freq . Wf^T  (pooled = GAP of the adapter input, [B,C]; freq = frequency embedding [B,F]);
 *        noisy = logits + noise / E  (noise: the caller's N(0,1) draw, [B,E], applied in train AND eval as the reference does);
 *        scores = softmax(noisy); top-k -> topk_idx [B,k] (int64), topk_val [B,k]; gates [B,E] = scores scattered at the top-k;
 *        aux[0] = 0.5 CV^2(sum_b softmax(logits) * complexity) + 0.5 CV^2(mean_b (1 - Phi((thr_b - logits) * E)))  when
 *        training != 0 (complexity may be NULL = no complexity bias), else 0;
 *        dispatch tables: counts[E], offsets[E+1], and for the B*k dispatched rows grouped by expert (samples ascending within
 *        an expert): perm (sample index, int64), perm_gate (its gate value), perm_expert; row_of [B,k] = the dispatched row
 *        of sample b's j-th choice.
 *   bwd: given dgates [B,E] (may be NULL), drow [B*k] (gradient of perm_gate, may be NULL) and daux (device scalar, may be
 *        NULL): dpooled [B,C], dfreq [B,F], dwg [E,C], dwf [E,F] (overwritten).  E <= 8, B*E <= 8192.
 * ------------------------------------------------------------------------ */
int mi_moe_route_fwd(const float* pooled, const float* freq, const float* wg, const float* wf, const float* noise,
                     const float* complexity, float* logits, float* gates, int64_t* topk_idx, float* topk_val, float* aux,
                     int* counts, int* offsets, int64_t* perm, float* perm_gate, int* perm_expert, int* row_of, int B, int C, int F,
                     int E, int k, int training, void* stream);
int mi_moe_route_bwd(const float* pooled, const float* freq, const float* wg, const float* wf, const float* noise,
                     const float* complexity, const float* logits, const int64_t* topk_idx, const float* dgates,
                     const float* drow, const int* row_of, const float* daux, float* dpooled, float* dfreq, float* dwg, float* dwf,
                     int B, int C, int F, int E, int k, int training, void* stream);

/* ------------------------------------------------------------------------
 * FFTAttention's patch spectrum product (moce_ir.py:408-414): irfft2(rfft2(x) * rfft2(y)) over every patch x patch block of
 * each [H,W] plane = the 2-D circular convolution  out[u][v] = sum_{a,b} x[a][b] y[(u-a)%p][(v-b)%p], evaluated directly in
 * fp32 from/to NCHW planes (zero padded to the patch grid at the bottom/right edge, cropped back; patch in {4,8,16,32}).
 * flip != 0 uses y'[i][j] = y[-i][-j]: dx = circconv(dout, y, flip=1), dy = circconv(dout, x, flip=1).
 * x_bs / y_bs / out_bs: batch strides in elements (0 = C*H*W), so that channel slices of a wider tensor (k of kv, the
 * k half of d_kv) need no copy.
 * ------------------------------------------------------------------------ */
int mi_patch_circconv(const void* x, int64_t x_bs, const void* y, int64_t y_bs, void* out, int64_t out_bs, int B, int C, int H,
                      int W, int patch, int flip, int dtype, void* stream);

/* FrequencyEmbedding's GELU -> spatial mean (moce_ir.py:1062-1064,1071-1073): out[b,c] = mean_n gelu(x[b,c,n]) (fp32);
 * bwd: dx = dout[b,c]/N * gelu'(x). */
int mi_gelu_gap_fwd(const void* x, float* out, int B, int C, int64_t N, int dtype, void* stream);
int mi_gelu_gap_bwd(const void* x, const float* dout, void* dx, int B, int C, int64_t N, int dtype, void* stream);

/* Gating products of the expert path: op 0  out = a * b  (FFTAttention `out * v`, moce_ir.py:419);
 * op 1  out = a * silu(b)  (ModExpert `body(x) * silu(proj[1](shared))`, :555);  op 2  out = gelu_erf(a), b unused
 * (FrequencyEmbedding's MLP activation, :1071).  bwd writes da and db (op 2: da only). */
/* a, b (and da, db) are [rows][L] with row strides in elements (0 = L); out and dout are contiguous [rows][L]. */
int mi_ewise_fwd(const void* a, int64_t a_rs, const void* b, int64_t b_rs, void* out, int64_t rows, int64_t L, int op, int dtype,
                 void* stream);
int mi_ewise_bwd(const void* a, int64_t a_rs, const void* b, int64_t b_rs, const void* dout, void* da, int64_t da_rs, void* db,
                 int64_t db_rs, int64_t rows, int64_t L, int op, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * U-Net glue, thin dense 3x3 convolutions (Restormer.py:156-165 OverlapPatchEmbed 3->48; :243,281 output conv 2*dim->3
 * + input residual).  Layout kernels that turn them into the 1x1 GEMM / Gram above:
 *   im2col3x3: x[B,C,H,W] -> col[B,9C,H,W], col[c*9+ky*3+kx][y][x] = x[c][y+ky-1][x+kx-1] (zero padded);
 *              flip != 0 negates the shifts (the im2col of the transposed convolution).
 *   col2im3x3: z[B,9M,H,W] -> y[B,M,H,W],  y[m][y][x] = sum_taps z[m*9+ky*3+kx][y+ky-1][x+kx-1] (+ bias[m]) (+ residual);
 *              flip != 0 negates the shifts (scatter of the transposed convolution: input gradient of the im2col form).
 * conv(x;W) = W[Cout,9Cin] . im2col(x) when Cin is tiny; = col2im(Wz[9Cout,Cin] . x) when Cout is tiny.
 * Any H, W: rows of 16..256 pixels (power of two) on 16-byte aligned planes take the wave-streaming forms, everything
 * else an element-wise general form (mi_glue3x3_ok: H, W >= 1).
 * ------------------------------------------------------------------------ */
int mi_glue3x3_ok(int H, int W);
int mi_im2col3x3(const void* x, void* out, int B, int C, int H, int W, int flip, int dtype, void* stream);
int mi_col2im3x3(const void* z, const float* bias, const void* residual, void* y, int B, int M, int H, int W, int flip,
                 int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Dense 3x3 convolution (stride 1, zero padding 1) as an implicit GEMM (csrc/conv3x3.hip): the glue convs above and the
 * C -> C/2 / C -> 2C bodies of Downsample / Upsample (Restormer.py:171-189) without the 9-plane im2col expansion in HBM.
 * bf16 activations, fp32 accumulate, W % 8 == 0 (mi_conv3x3_ok); other planes / fp32 use the im2col forms above.
 *   mi_conv3x3_pack : w fp32 -> the kernel's fragment-major bf16 image (mi_conv3x3_pack_bytes(M, K) bytes; re-pack after every
 *                     weight update).  transpose_flip == 0: w is [M][K][3][3] and the op is y = conv(x; w).
 *                     transpose_flip != 0: w is the conv's own weight [K][M][3][3] and the op is its DATA GRADIENT
 *                     dx[M planes] = conv_transpose(dy[K planes]; w)  (taps flipped, channel roles swapped).
 *   mi_conv3x3_fwd  : y[B,M,H,W] = conv(x[B,K,H,W]) (+ bias[m]) (+ residual[B,M,H,W]); x_bs / r_bs / y_bs batch strides in
 *                     elements (0 = dense) so operands may be channel slices; channel planes dense H*W.
 *   mi_conv3x3_wgrad: dw[M][K][3][3] (+)= sum_{b,p} dy[b][m][p] . x[b][k][p + d(tap)]; ws = mi_conv3x3_wgrad_workspace bytes.
 * ------------------------------------------------------------------------ */
int mi_conv3x3_ok(int H, int W, int dtype);
size_t mi_conv3x3_pack_bytes(int M, int K);
int mi_conv3x3_pack(const float* w, int M, int K, int transpose_flip, void* pack, void* stream);
int mi_conv3x3_fwd(const void* pack, const void* x, int64_t x_bs, const float* bias, const void* residual, int64_t r_bs,
                   void* y, int64_t y_bs, int B, int M, int K, int H, int W, void* stream);
size_t mi_conv3x3_wgrad_workspace(int B, int M, int K, int H, int W);
int mi_conv3x3_wgrad(const void* dy, int64_t dy_bs, const void* x, int64_t x_bs, float* dw, int accumulate, int B, int M, int K,
                     int H, int W, void* ws, void* stream);

/* PixelShuffle(2) / PixelUnshuffle(2) of Upsample / Downsample (Restormer.py:171-189; moce_ir.py Downsample/Upsample):
 *   unshuffle == 0: in [B,4c,H,W] -> out [B,c,2H,2W], out[b][c][2y+i][2x+j] = in[b][4c+2i+j][y][x]
 *   unshuffle != 0: in [B,c,2H,2W] -> out [B,4c,H,W] (the inverse; also each other's backward).
 * in_bs / out_bs: batch strides in elements (0 = dense), so the high-resolution side may be a channel slice of the decoder's
 * concatenation buffer (Restormer.py:266 torch.cat).  mi_copy_rows moves [rows][L] with row strides (the skip half of it). */
int mi_pixel_shuffle2(const void* in, int64_t in_bs, void* out, int64_t out_bs, int B, int c, int H, int W, int unshuffle,
                      int dtype, void* stream);
int mi_copy_rows(const void* src, int64_t src_rs, void* dst, int64_t dst_rs, int64_t rows, int64_t L, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Router global average pool (moce_ir.py:703-707, RoutingFunction.gate[0]): out[b,c] = mean_n x[b,c,n] (fp32);
 * bwd: dx[b,c,:] = dout[b,c]/N.
 * ------------------------------------------------------------------------ */
int mi_gap_fwd(const void* x, float* out, int B, int C, int64_t N, int dtype, void* stream);
/* bias gradient of a conv (nn.Conv2d(bias=True)): out[c] (+)= sum over batch and pixels of x[b][c][n]; fixed summation order */
size_t mi_chan_sum_workspace(int C, int64_t N);
int mi_chan_sum(const void* x, float* out, int B, int C, int64_t N, int dtype, int accumulate, void* ws, void* stream);
int mi_gap_bwd(const float* dout, void* dx, int B, int C, int64_t N, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Training samples on the device (MoCE-IR-main/src/data/dataset_utils.py:156-165 AIOTrainDataset.__getitem__, denoise tasks;
 * src/data/degradation_utils.py:21-24; src/utils/image_utils.py data_augmentation): for sample b of the batch
 *   patch = image[sample[b]][top[b] : top[b]+P, left[b] : left[b]+P]      (decoded uint8 HWC images in one flat `pool`,
 *                                                                           image s at pool + src_off[s], src_h[s] x src_w[s])
 *   aug   = data_augmentation(patch, mode[b])                              (the 8 dihedral modes, numbered as the reference)
 *   clean[b]    = aug / 255                                                 (CHW, activation dtype)
 *   degraded[b] = uint8(clip(aug + sigma[b] * noise[b], 0, 255)) / 255      (noise: the caller's N(0,1) draw, [B,3,P,P])
 * clean or degraded may be NULL.  The caller keeps top/left inside the image.
 * ------------------------------------------------------------------------ */
int mi_patch_batch(const unsigned char* pool, const int64_t* src_off, const int* src_h, const int* src_w, const int* sample,
                   const int* top, const int* left, const int* mode, const float* sigma, const float* noise, void* clean,
                   void* degraded, int B, int P, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Evaluation metrics as the reference's test loops compute them through scikit-image (AdaIR-main/utils/val_utils.py:50-64;
 * MoCE-IR-main/src/test.py:82-123): inputs clipped to [0,1];  psnr[b] = 10 log10(1 / MSE);  ssim[b] = structural_similarity
 * with data_range 1, 7x7 uniform window, K1 0.01, K2 0.03, sample covariance, 3-pixel border excluded, channel mean.
 * ws: mi_psnr_ssim_workspace() bytes.  Deterministic (fixed-order sums).
 * ------------------------------------------------------------------------ */
size_t mi_psnr_ssim_workspace(int B, int C, int H, int W);
int mi_psnr_ssim(const void* restored, const void* clean, float* psnr, float* ssim, int B, int C, int H, int W, int dtype,
                 void* ws, void* stream);

/* dtype conversion / L1 loss helpers used by the harness */
int mi_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* loss[0] += mean|a-b| ; da = sign(a-b) * scale (da may be NULL); loss must hold 1+1024 floats
 * (entries 1.. are scratch for the block partials) */
int mi_l1_loss(const void* a, const void* b, void* da, float* loss, int64_t n, float scale, int dtype,
               void* stream);

/* ------------------------------------------------------------------------
 * Optional per-kernel profiler (measurement aid for bench.py's roofline object; nothing in the
 * reference corresponds to it).  When enabled every kernel launch is bracketed by two HIP events
 * recorded on the stream it is launched on, and its algorithmic HBM bytes / flops are booked.
 * mi_prof_collect synchronises, fills per-kernel-id totals (arrays of mi_prof_kernel_count()
 * entries: milliseconds, algorithmic bytes, flops, launches).  mi_prof_enable(1|0) clears records.
 * Not for use inside HIP-graph capture.
 * ------------------------------------------------------------------------ */
int mi_prof_enable(int on);
int mi_prof_kernel_count(void);
const char* mi_prof_kernel_name(int kid);
int mi_prof_collect(double* ms, double* bytes, double* flops, int64_t* launches, int n);

#ifdef __cplusplus
}
#endif
#endif /* MI_RESTORE_H */
