// Internal launchers shared between translation units (not part of the C-ABI).
#pragma once
#include "common.h"

namespace mi {
// kernel-side view of mi_pw_desc (strides in elements)
struct PwK {
  const void* x1; int64_t x1_bs, x1_gs; int k1;
  const void* x2; int64_t x2_bs, x2_gs; int k2;
  const float* w; int64_t w_bs, w_gs, w_sm, w_sk;
  const float* bias; int64_t bias_gs;
  const void* r; int64_t r_bs, r_gs;
  void* y; int64_t y_bs, y_gs;
  int m; int64_t n; int groups; int vec_ok;
  void* y2; int64_t y2_bs, y2_gs; int y_split;   // rows >= y_split go to y2 (wave-owned forms, no residual); 0 = one output
};
int launch_attn_fold(const float* graw, const float* ss, const float* temperature, const float* wo, float* P, float* A,
                     float* nrm, float* M, int B, int C, int heads, hipStream_t st, void* Mb = nullptr, void* Mtb = nullptr);
int launch_attn_bwd_small(const float* dM, const float* A, const float* P, const float* nrm, const float* temperature,
                          const float* wo, float* dwo_part, float* dtemp_part, float* wd, float* scratch,
                          int B, int C, int heads, hipStream_t st, void* wdb = nullptr);
size_t attn_bwd_scratch_floats(int B, int C, int heads);
size_t chan_sum_workspace(int C, int64_t N);
int launch_chan_sum(const void* x, float* out, int B, int C, int64_t N, int dtype, int accumulate, void* ws, hipStream_t st);

// ---- depthwise convolution (dwconv.hip: LDS-tiled k x k; dwstream.hip: register-streaming 3x3) ----
enum { IN_PLAIN = 0, IN_GATE_BWD = 1 };
struct DwArgs {
  const void* in;     // plain: x / dy  [B,Cc,H,W]
  const void* gy;     // gate-bwd: conv outputs y [B,2h,H,W] (in = dg [B,h,H,W])
  const float* w;     // [Cc, KS*KS]
  const float* bias;  // [Cc] or null
  void* out;          // [B,Cc,H,W] (may be null in gate fwd)
  void* gate;         // gate fwd: g [B,h,H,W]
  int Cc, H, W, hidden, tiles_x;
};
// Streaming 3x3 path: usable when the row is 16..256 pixels, a power of two, and every plane base is 16-byte aligned.
bool dws_eligible(int H, int W, int ks);
int dws_partial_rows(int B, int H, int W, int64_t planes);  // rows of weight-gradient partials the backward kernels write
int dws_fwd(const DwArgs& a, int B, bool gate, bool flip, int dtype, hipStream_t st);
int dws_bwd(const DwArgs& dya, const void* xin, float* part, int B, bool want_dx, int* rows_out, int dtype, hipStream_t st);
int dws_gate_bwd(const DwArgs& a, const void* xin, float* part, int B, bool want_dw, int* rows_out, int dtype,
                 hipStream_t st);
// same, with the conv outputs recomputed from the conv input: a.in = dg, a.gy = conv input x, a.bias = conv bias
int dws_gate_bwd_recompute(const DwArgs& a, float* part, int B, bool want_dw, int* rows_out, int dtype, hipStream_t st);
// ---- backward tail of a half-block (bwd_tail.hip): dW, W^T dY, LayerNorm backward and the residual add in one launch ----
bool bwd_tail_ok(int M, int C, int64_t N, int dtype);
bool bwd_tail_pays(int M, int C);   // covered AND faster than the unfused chain (the module entry points use the tail only then)
size_t bwd_tail_workspace(int M, int C);
int launch_bwd_tail(const void* dy, int M, const void* x, int C, const void* dres, const float* mean, const float* rstd,
                    const float* w, const float* gamma, const float* beta, void* dx, float* dw, float* dgamma, float* dbeta,
                    int B, int64_t N, int accumulate, void* ws, hipStream_t st);
// ---- LDS-tiled 1x1 GEMM for deep K (pw_lds.hip); mi_pw_gemm hands it the shapes pw_lds_ok accepts ----
bool pw_lds_ok(const mi_pw_desc* d);
size_t pw_lds_pack_bytes(const mi_pw_desc* d);
int pw_lds_launch(const mi_pw_desc* d, void* ws, hipStream_t st);
// ---- fused GDFN forward, training form (fused_gdfn.hip): also writes h0 [B][2h][H][W] and g [B][h][H][W] ----
int fused_gdfn_fwd_save(const mi_gdfn_fused_shape* s, const void* pack, const void* y, void* out, float* mean, float* rstd,
                        void* h0, void* g, hipStream_t st);
}  // namespace mi
