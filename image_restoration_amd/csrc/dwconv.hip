// Depthwise k x k convolution on NCHW planes (Restormer.py:84,106; moce_ir.py:341), its
// transposed (backward-data) form and the weight gradient, with the GDFN GELU gate
// (Restormer.py:90-91) fused as an output epilogue (forward) or an input prologue (backward).
// One workgroup = one (image, channel) plane tile staged in LDS with its halo; each thread
// produces 4 consecutive outputs of one row.  HBM-bound: reads the input once (+halo), writes once.
#include "common.h"

namespace mi {

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

template <typename T, int MODE>
__device__ __forceinline__ float dw_fetch(const DwArgs& a, int b, int cc, int y, int x) {
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t pix = (int64_t)y * a.W + x;
  if (MODE == IN_PLAIN) {
    return ld1((const T*)a.in + ((int64_t)b * a.Cc + cc) * HW + pix);
  } else {
    const int j = cc < a.hidden ? cc : cc - a.hidden;
    const float dg = ld1((const T*)a.in + ((int64_t)b * a.hidden + j) * HW + pix);
    const float y1 = ld1((const T*)a.gy + ((int64_t)b * a.Cc + j) * HW + pix);
    if (cc < a.hidden) {
      const float y2 = ld1((const T*)a.gy + ((int64_t)b * a.Cc + j + a.hidden) * HW + pix);
      return dg * y2 * gelu_erf_grad(y1);
    }
    return dg * gelu_erf(y1);
  }
}

template <int KS, int TW> struct DwGeom {
  static constexpr int P = KS / 2;
  static constexpr int TXN = TW / 4;         // threads along x
  static constexpr int TH = 256 / TXN;       // tile rows
  static constexpr int LW = TW + 2 * P + 1;  // LDS row stride (floats)
  static constexpr int LH = TH + 2 * P;
};

// 4 consecutive pixels (x..x+3) of plane cc at row y; out-of-image pixels read as 0.  vec_ok: W % 4 == 0 and
// 16-byte aligned bases, so a 4-group that starts inside the row lies entirely inside it.
template <typename T, int MODE>
__device__ __forceinline__ void dw_fetch4(const DwArgs& a, int b, int cc, int y, int x, bool vec_ok, float* o) {
  if (!(vec_ok && x >= 0 && x + 3 < a.W)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (x + j >= 0 && x + j < a.W) ? dw_fetch<T, MODE>(a, b, cc, y, x + j) : 0.f;
    return;
  }
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t pix = (int64_t)y * a.W + x;
  if (MODE == IN_PLAIN) {
    Vec<T, 4>::ld((const T*)a.in + ((int64_t)b * a.Cc + cc) * HW + pix, o);
  } else {
    const int j = cc < a.hidden ? cc : cc - a.hidden;
    float dg[4], y1[4];
    Vec<T, 4>::ld((const T*)a.in + ((int64_t)b * a.hidden + j) * HW + pix, dg);
    Vec<T, 4>::ld((const T*)a.gy + ((int64_t)b * a.Cc + j) * HW + pix, y1);
    if (cc < a.hidden) {
      float y2[4];
      Vec<T, 4>::ld((const T*)a.gy + ((int64_t)b * a.Cc + j + a.hidden) * HW + pix, y2);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = dg[e] * y2[e] * gelu_erf_grad(y1[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = dg[e] * gelu_erf(y1[e]);
    }
  }
}

// Stage the (TH+2P) x (TW+2P) input tile: the TW interior columns as 4-wide vector loads, the 2P halo columns scalar.
template <typename T, int KS, int TW, int MODE>
__device__ __forceinline__ void dw_stage(float* tile, const DwArgs& a, int b, int cc, int y0, int x0, bool vec_ok) {
  using G = DwGeom<KS, TW>;
  constexpr int VPR = TW / 4;  // vectors per row
  for (int e = threadIdx.x; e < G::LH * VPR; e += 256) {
    const int r = e / VPR, v = e - r * VPR;
    const int gy = y0 - G::P + r, gx = x0 + 4 * v;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (gy >= 0 && gy < a.H && gx < a.W) dw_fetch4<T, MODE>(a, b, cc, gy, gx, vec_ok, o);
    float* d = tile + r * G::LW + G::P + 4 * v;
    d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3];
  }
  for (int e = threadIdx.x; e < G::LH * 2 * G::P; e += 256) {
    const int r = e / (2 * G::P), hcol = e - r * (2 * G::P);
    const int c = hcol < G::P ? hcol : TW + hcol;  // left halo cols [0,P), right halo cols [TW+P, TW+2P)
    const int gy = y0 - G::P + r, gx = x0 - G::P + c;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = dw_fetch<T, MODE>(a, b, cc, gy, gx);
    tile[r * G::LW + c] = v;
  }
}

template <int KS, int TW, bool FLIP>
__device__ __forceinline__ void dw_compute(const float* tile, const float* wk, float bias, int ty, int tx, float* o) {
  using G = DwGeom<KS, TW>;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = bias;
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
    float row[4 + KS - 1];
#pragma unroll
    for (int i = 0; i < 4 + KS - 1; ++i) row[i] = tile[(ty + ky) * G::LW + 4 * tx + i];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const float wv = FLIP ? wk[KS * KS - 1 - (ky * KS + kx)] : wk[ky * KS + kx];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += wv * row[j + kx];
    }
  }
}

template <typename T>
__device__ __forceinline__ void dw_store4(T* plane, int H, int W, int y, int x, const float* o, bool vec_ok) {
  if (y >= H || x >= W) return;
  T* p = plane + (int64_t)y * W + x;
  if (vec_ok && x + 3 < W) {
    Vec<T, 4>::st(p, o);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x + j < W) st1(p + j, o[j]);
  }
}

// forward / backward-data.  GATE: blockIdx.y indexes the hidden channel j; planes j and j+hidden are convolved,
// y (optional) gets both, g = gelu(y1)*y2.
template <typename T, int KS, int TW, int MODE, bool GATE, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a, int vec_ok) {
  using G = DwGeom<KS, TW>;
  __shared__ float tile[(GATE ? 2 : 1) * G::LH * G::LW];
  __shared__ float wsm[(GATE ? 2 : 1) * KS * KS];
  const int tile_id = blockIdx.x;
  const int x0 = (tile_id % a.tiles_x) * TW, y0 = (tile_id / a.tiles_x) * G::TH;
  const int cc = blockIdx.y, b = blockIdx.z;
  const int tx = threadIdx.x % G::TXN, ty = threadIdx.x / G::TXN;
  if (threadIdx.x < KS * KS) {
    wsm[threadIdx.x] = a.w[(int64_t)cc * KS * KS + threadIdx.x];
    if (GATE) wsm[KS * KS + threadIdx.x] = a.w[(int64_t)(cc + a.hidden) * KS * KS + threadIdx.x];
  }
  dw_stage<T, KS, TW, MODE>(tile, a, b, cc, y0, x0, vec_ok);
  if (GATE) dw_stage<T, KS, TW, MODE>(tile + G::LH * G::LW, a, b, cc + a.hidden, y0, x0, vec_ok);
  __syncthreads();
  const int64_t HW = (int64_t)a.H * a.W;
  float o1[4];
  dw_compute<KS, TW, FLIP>(tile, wsm, a.bias ? a.bias[cc] : 0.f, ty, tx, o1);
  const int oy = y0 + ty, ox = x0 + 4 * tx;
  if (!GATE) {
    dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + cc) * HW, a.H, a.W, oy, ox, o1, vec_ok);
  } else {
    float o2[4], g[4];
    dw_compute<KS, TW, FLIP>(tile + G::LH * G::LW, wsm + KS * KS, a.bias ? a.bias[cc + a.hidden] : 0.f, ty, tx, o2);
    if (a.out) {
      dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + cc) * HW, a.H, a.W, oy, ox, o1, vec_ok);
      dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + cc + a.hidden) * HW, a.H, a.W, oy, ox, o2, vec_ok);
      // the gate is evaluated on the values as stored (what backward will re-read)
#pragma unroll
      for (int j = 0; j < 4; ++j) { o1[j] = to_f32(Cvt<T>::from(o1[j])); o2[j] = to_f32(Cvt<T>::from(o2[j])); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = gelu_erf(o1[j]) * o2[j];
    dw_store4<T>((T*)a.gate + ((int64_t)b * a.hidden + cc) * HW, a.H, a.W, oy, ox, g, vec_ok);
  }
}

// weight gradient: dw[c][ky][kx] = sum_{b,y,x} dy[b,c,y,x] * x[b,c,y+ky-P,x+kx-P];  db[c] = sum dy.
// grid (tiles, Cc); loops over the batch; block partial -> part[tile][Cc*KK | Cc].
template <typename T, int KS, int TW, int MODE>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(DwArgs dya, const T* __restrict__ xin, float* __restrict__ part,
                                                           int B, int vec_ok) {
  using G = DwGeom<KS, TW>;
  constexpr int KK = KS * KS;
  __shared__ float tile[G::LH * G::LW];
  __shared__ float red[4][KK + 1];
  const int tile_id = blockIdx.x;
  const int x0 = (tile_id % dya.tiles_x) * TW, y0 = (tile_id / dya.tiles_x) * G::TH;
  const int cc = blockIdx.y;
  const int tx = threadIdx.x % G::TXN, ty = threadIdx.x / G::TXN;
  const int oy = y0 + ty, ox = x0 + 4 * tx;
  DwArgs xa = dya;
  xa.in = xin;
  float acc[KK + 1];
#pragma unroll
  for (int i = 0; i <= KK; ++i) acc[i] = 0.f;
  for (int b = 0; b < B; ++b) {
    __syncthreads();
    dw_stage<T, KS, TW, IN_PLAIN>(tile, xa, b, cc, y0, x0, vec_ok);
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (oy < dya.H && ox < dya.W) dw_fetch4<T, MODE>(dya, b, cc, oy, ox, vec_ok, d);
    __syncthreads();
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      float row[4 + KS - 1];
#pragma unroll
      for (int i = 0; i < 4 + KS - 1; ++i) row[i] = tile[(ty + ky) * G::LW + 4 * tx + i];
#pragma unroll
      for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ky * KS + kx] += d[j] * row[j + kx];
    }
    acc[KK] += (d[0] + d[1]) + (d[2] + d[3]);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i <= KK; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x <= KK) {
    const int i = threadIdx.x;
    const float s = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    float* prow = part + (int64_t)tile_id * ((int64_t)dya.Cc * (KK + 1));
    if (i < KK) prow[(int64_t)cc * KK + i] = s;
    else prow[(int64_t)dya.Cc * KK + cc] = s;
  }
}

static int pick_tw(int W) { return W >= 48 ? 64 : (W >= 24 ? 32 : 16); }

template <typename T, int KS, int MODE, bool GATE, bool FLIP>
static int dw_launch(DwArgs a, int B, hipStream_t st) {
  const int tw = pick_tw(a.W);
  const int th = 256 / (tw / 4);
  a.tiles_x = cdiv(a.W, tw);
  const int tiles = a.tiles_x * cdiv(a.H, th);
  const int vec_ok = (a.W % 4 == 0) && aligned16(a.out) && aligned16(a.gate) && aligned16(a.in) && aligned16(a.gy);
  dim3 grid(tiles, GATE ? a.hidden : a.Cc, B), block(256);
  const double plane = (double)B * a.H * a.W * sizeof(T);
  const int kid = GATE ? K_DW_GATE_FWD : (FLIP ? (MODE == IN_GATE_BWD ? K_DW_GATE_BWD_DATA : K_DW_BWD_DATA) : K_DW_FWD);
  const double chans = GATE ? (a.Cc + (a.out ? a.Cc : 0) + a.hidden)
                            : (MODE == IN_GATE_BWD ? (a.hidden + 2.0 * a.Cc) : 2.0 * a.Cc);
  ProfScope ps(st, kid, chans * plane, 2.0 * KS * KS * a.Cc * (double)B * a.H * a.W);
  if (tw == 64) hipLaunchKernelGGL((dwconv_kernel<T, KS, 64, MODE, GATE, FLIP>), grid, block, 0, st, a, vec_ok);
  else if (tw == 32) hipLaunchKernelGGL((dwconv_kernel<T, KS, 32, MODE, GATE, FLIP>), grid, block, 0, st, a, vec_ok);
  else hipLaunchKernelGGL((dwconv_kernel<T, KS, 16, MODE, GATE, FLIP>), grid, block, 0, st, a, vec_ok);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <typename T, int KS, int MODE>
static int dw_wgrad_launch(DwArgs dya, const void* xin, float* part, int B, int* tiles_out, hipStream_t st) {
  const int tw = pick_tw(dya.W);
  const int th = 256 / (tw / 4);
  dya.tiles_x = cdiv(dya.W, tw);
  const int tiles = dya.tiles_x * cdiv(dya.H, th);
  *tiles_out = tiles;
  dim3 grid(tiles, dya.Cc), block(256);
  const int vec_ok = (dya.W % 4 == 0) && aligned16(dya.in) && aligned16(dya.gy) && aligned16(xin);
  const double plane = (double)B * dya.H * dya.W * sizeof(T);
  ProfScope ps(st, K_DW_WGRAD, (MODE == IN_GATE_BWD ? dya.hidden + 2.0 * dya.Cc : 2.0 * dya.Cc) * plane,
               2.0 * KS * KS * dya.Cc * (double)B * dya.H * dya.W);
  if (tw == 64) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, KS, 64, MODE>), grid, block, 0, st, dya, (const T*)xin, part, B, vec_ok);
  else if (tw == 32) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, KS, 32, MODE>), grid, block, 0, st, dya, (const T*)xin, part, B, vec_ok);
  else hipLaunchKernelGGL((dwconv_wgrad_kernel<T, KS, 16, MODE>), grid, block, 0, st, dya, (const T*)xin, part, B, vec_ok);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

static int dw_tiles(int H, int W) {
  const int tw = pick_tw(W);
  const int th = 256 / (tw / 4);
  return cdiv(W, tw) * cdiv(H, th);
}

static int check_common(const char* who, int B, int C, int H, int W, int ks, int dtype) {
  MI_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0, "%s: bad shape B=%d C=%d H=%d W=%d", who, B, C, H, W);
  MI_CHECK_ARG(ks == 3 || ks == 5 || ks == 7, "%s: kernel size %d unsupported (3,5,7)", who, ks);
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "%s: bad dtype %d", who, dtype);
  MI_CHECK_ARG(B <= 65535 && C <= 65535, "%s: B or C exceeds grid limits", who);
  return MI_OK;
}

#define DW_DISPATCH(T_, KS_, CALL)                 \
  do {                                             \
    if (dtype == MI_F32) {                         \
      using T_ = float;                            \
      if (ks == 3) { constexpr int KS_ = 3; CALL; } \
      else if (ks == 5) { constexpr int KS_ = 5; CALL; } \
      else { constexpr int KS_ = 7; CALL; }        \
    } else {                                       \
      using T_ = bf16;                             \
      if (ks == 3) { constexpr int KS_ = 3; CALL; } \
      else if (ks == 5) { constexpr int KS_ = 5; CALL; } \
      else { constexpr int KS_ = 7; CALL; }        \
    }                                              \
  } while (0)

}  // namespace mi

using namespace mi;

extern "C" int mi_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int B, int C, int H, int W, int ks,
                             int dtype, void* stream) {
  MI_CHECK_ARG(x && w && y, "dwconv_fwd: null pointer");
  MI_TRY(check_common("dwconv_fwd", B, C, H, W, ks, dtype));
  DwArgs a{x, nullptr, w, bias, y, nullptr, C, H, W, 0, 0};
  int rc = MI_OK;
  DW_DISPATCH(T, KS, (rc = dw_launch<T, KS, IN_PLAIN, false, false>(a, B, (hipStream_t)stream)));
  return rc;
}

extern "C" int mi_dwconv_gate_fwd(const void* x, const float* w, const float* bias, void* y, void* g, int B, int C2, int H,
                                  int W, int ks, int dtype, void* stream) {
  MI_CHECK_ARG(x && w && g, "dwconv_gate_fwd: null pointer");
  MI_CHECK_ARG(C2 % 2 == 0, "dwconv_gate_fwd: channel count %d must be even", C2);
  MI_TRY(check_common("dwconv_gate_fwd", B, C2, H, W, ks, dtype));
  DwArgs a{x, nullptr, w, bias, y, g, C2, H, W, C2 / 2, 0};
  int rc = MI_OK;
  DW_DISPATCH(T, KS, (rc = dw_launch<T, KS, IN_PLAIN, true, false>(a, B, (hipStream_t)stream)));
  return rc;
}

extern "C" size_t mi_dwconv_bwd_workspace(int B, int C, int H, int W, int ks) {
  (void)B;
  if (H <= 0 || W <= 0 || C <= 0) return 0;
  return align_up((size_t)dw_tiles(H, W) * C * (ks * ks + 1) * sizeof(float), 256);
}

static int dw_bwd_common(const void* dy_or_dg, const void* gy, const void* x, const float* w, void* dx, float* dwg,
                         float* dbg, int B, int Cc, int H, int W, int ks, int accumulate, int dtype, void* ws, int gate,
                         hipStream_t st) {
  DwArgs a{dy_or_dg, gy, w, nullptr, dx, nullptr, Cc, H, W, gate ? Cc / 2 : 0, 0};
  int rc = MI_OK;
  if (dx) {
    if (gate) DW_DISPATCH(T, KS, (rc = dw_launch<T, KS, IN_GATE_BWD, false, true>(a, B, st)));
    else DW_DISPATCH(T, KS, (rc = dw_launch<T, KS, IN_PLAIN, false, true>(a, B, st)));
    if (rc != MI_OK) return rc;
  }
  if (dwg) {
    MI_CHECK_ARG(ws && x, "dwconv_bwd: weight gradient needs x and a workspace");
    int tiles = 0;
    float* part = (float*)ws;
    if (gate) DW_DISPATCH(T, KS, (rc = dw_wgrad_launch<T, KS, IN_GATE_BWD>(a, x, part, B, &tiles, st)));
    else DW_DISPATCH(T, KS, (rc = dw_wgrad_launch<T, KS, IN_PLAIN>(a, x, part, B, &tiles, st)));
    if (rc != MI_OK) return rc;
    const int kk = ks * ks;
    const int64_t ld = (int64_t)Cc * (kk + 1);
    MI_TRY(launch_reduce_rows(part, dwg, tiles, (int64_t)Cc * kk, ld, accumulate, 1.0f, st));
    if (dbg) MI_TRY(launch_reduce_rows(part + (int64_t)Cc * kk, dbg, tiles, Cc, ld, accumulate, 1.0f, st));
  }
  return MI_OK;
}

extern "C" int mi_dwconv_bwd(const void* dy, const void* x, const float* w, void* dx, float* dwg, float* dbg, int B, int C,
                             int H, int W, int ks, int accumulate, int dtype, void* ws, void* stream) {
  MI_CHECK_ARG(dy && w, "dwconv_bwd: null pointer");
  MI_TRY(check_common("dwconv_bwd", B, C, H, W, ks, dtype));
  return dw_bwd_common(dy, nullptr, x, w, dx, dwg, dbg, B, C, H, W, ks, accumulate, dtype, ws, 0, (hipStream_t)stream);
}

extern "C" int mi_dwconv_gate_bwd(const void* dg, const void* y, const void* x, const float* w, void* dx, float* dwg,
                                  float* dbg, int B, int C2, int H, int W, int ks, int accumulate, int dtype, void* ws,
                                  void* stream) {
  MI_CHECK_ARG(dg && y && w, "dwconv_gate_bwd: null pointer");
  MI_CHECK_ARG(C2 % 2 == 0, "dwconv_gate_bwd: channel count %d must be even", C2);
  MI_TRY(check_common("dwconv_gate_bwd", B, C2, H, W, ks, dtype));
  return dw_bwd_common(dg, y, x, w, dx, dwg, dbg, B, C2, H, W, ks, accumulate, dtype, ws, 1, (hipStream_t)stream);
}
