// Depthwise k x k convolution on NCHW planes (Restormer.py:84,106; moce_ir.py:341), its
// transposed (backward-data) form and the weight gradient, with the GDFN GELU gate
// (Restormer.py:90-91) fused as an output epilogue (forward) or an input prologue (backward).
//
// One workgroup = one (image, channel) plane tile staged in LDS with its halo.  256 threads form a
// TXN x (256/TXN) grid; each thread produces RPT consecutive rows x 4 consecutive pixels with a sliding
// row window (RPT = 4 on large planes: 64 x 64 outputs per workgroup, ~8 KB of loads in flight per workgroup
// instead of 2 KB: these kernels are HBM/latency bound and MI355X wants >= 40 KB in flight per CU).
// Backward computes dx AND the weight/bias gradient partials in ONE pass over (dy, x): the dy tile is staged
// once and used for both.
#include "internal.h"
#include <algorithm>
#include <type_traits>

namespace mi {

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

template <int KS, int TW, int RPT> struct DwGeom {
  static constexpr int P = KS / 2;
  static constexpr int TXN = TW / 4;              // threads along x
  static constexpr int TYN = 256 / TXN;           // threads along y
  static constexpr int TH = TYN * RPT;            // tile rows
  static constexpr int LW = (TW + 2 * P + 3) / 4 * 4;  // LDS row stride (floats), multiple of 4: a thread's row window
                                                     // starts 16-byte aligned at column 4*tx -> ds_read_b128/b64
  static constexpr int LH = TH + 2 * P;
  static constexpr int WIN = RPT + KS - 1;        // input rows a thread touches
};

// Staging of the (TH+2P) x (TW+2P) input tile is split in two phases so that EVERY vector load of a workgroup's
// tile(s) is in flight before the first LDS write waits on one: dw_load_vecs (interior columns, 4-wide vectors, fully
// unrolled) ... dw_write_vecs; the 2P halo columns follow as scalars.
template <typename G, int TW> struct DwStage {
  static constexpr int VPR = TW / 4;                          // vectors per row
  static constexpr int NV = (G::LH * VPR + 255) / 256;        // vectors per thread
};

// threadIdx.x through an opaque move: inside the x-tile loops this keeps everything derived from the thread id (row /
// vector indices, LDS offsets) from being hoisted out of the loop and held in VGPRs for the whole kernel.
__device__ __forceinline__ int dw_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

template <typename T, typename G, int TW, int MODE>
__device__ __forceinline__ void dw_load_vecs(const DwArgs& a, int b, int cc, int y0, int x0, bool vec_ok,
                                             float (*o)[4]) {
  using S = DwStage<G, TW>;
  const int tid = dw_tid();
#pragma unroll
  for (int i = 0; i < S::NV; ++i) {
    const int e = tid + 256 * i;
    const int r = e / S::VPR, v = e - r * S::VPR;
    const int gy = y0 - G::P + r, gx = x0 + 4 * v;
    o[i][0] = o[i][1] = o[i][2] = o[i][3] = 0.f;
    if (e < G::LH * S::VPR && gy >= 0 && gy < a.H && gx < a.W) dw_fetch4<T, MODE>(a, b, cc, gy, gx, vec_ok, o[i]);
  }
}

template <typename G, int TW>
__device__ __forceinline__ void dw_write_vecs(float* tile, const float (*o)[4]) {
  using S = DwStage<G, TW>;
  const int tid = dw_tid();
#pragma unroll
  for (int i = 0; i < S::NV; ++i) {
    const int e = tid + 256 * i;
    if (e < G::LH * S::VPR) {
      const int r = e / S::VPR, v = e - r * S::VPR;
      float* d = tile + r * G::LW + G::P + 4 * v;
      d[0] = o[i][0]; d[1] = o[i][1]; d[2] = o[i][2]; d[3] = o[i][3];
    }
  }
}

template <typename T, typename G, int TW, int MODE>
__device__ __forceinline__ void dw_stage_halo(float* tile, const DwArgs& a, int b, int cc, int y0, int x0) {
  for (int e = dw_tid(); e < G::LH * 2 * G::P; e += 256) {
    const int r = e / (2 * G::P), hcol = e - r * (2 * G::P);
    const int c = hcol < G::P ? hcol : TW + hcol;  // left halo cols [0,P), right halo cols [TW+P, TW+2P)
    const int gy = y0 - G::P + r, gx = x0 - G::P + c;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = dw_fetch<T, MODE>(a, b, cc, gy, gx);
    tile[r * G::LW + c] = v;
  }
}

// N = KS+3 consecutive floats from a 16-byte aligned LDS address as wide reads (N is even: 6, 8 or 10)
template <int N>
__device__ __forceinline__ void dw_window(const float* p, float* row) {
#pragma unroll
  for (int i = 0; i + 4 <= N; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
    row[i] = v[0]; row[i + 1] = v[1]; row[i + 2] = v[2]; row[i + 3] = v[3];
  }
  if (N % 4) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(p + N / 4 * 4);
    row[N / 4 * 4] = v[0]; row[N / 4 * 4 + 1] = v[1];
  }
}

// o[r][j] = bias + sum_{ky,kx} w[ky][kx] * tile[RPT*ty + r + ky][4*tx + j + kx]; each input row is read once.
template <typename G, int KS, int RPT, bool FLIP>
__device__ __forceinline__ void dw_compute(const float* tile, const float* wk, float bias, int ty, int tx, float (*o)[4]) {
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) o[r][j] = bias;
#pragma unroll
  for (int wr = 0; wr < G::WIN; ++wr) {
    float row[4 + KS - 1];
    dw_window<4 + KS - 1>(&tile[(RPT * ty + wr) * G::LW + 4 * tx], row);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int ky = wr - r;
      if (ky < 0 || ky >= KS) continue;
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const float wv = FLIP ? wk[KS * KS - 1 - (ky * KS + kx)] : wk[ky * KS + kx];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[r][j] += wv * row[j + kx];
      }
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

#define DW_OPAQUE(v) asm volatile("" : "+s"(v))

// forward.  GATE: blockIdx.y indexes the hidden channel j; planes j and j+hidden are convolved,
// y (optional) gets both, g = gelu(y1)*y2.
template <typename T, int KS, int TW, int RPT, bool GATE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GATE && KS == 3 ? 4 : 1)))
void dwconv_kernel(DwArgs a, int vec_ok) {
  using G = DwGeom<KS, TW, RPT>;
  __shared__ __attribute__((aligned(16))) float tile[(GATE ? 2 : 1) * G::LH * G::LW];
  __shared__ float wsm[(GATE ? 2 : 1) * KS * KS];
  // One workgroup = one ROW BAND of a plane; it walks the band's x-tiles left to right, so the halo columns of a tile
  // are lines this CU has just read (or reads next) as interior: measured with FETCH_SIZE, independent 64-wide tiles
  // pulled 2.3x their algorithmic reads through the fabric (two extra 128-byte lines per row).
  const int y0_band = blockIdx.x * G::TH;
  const int cc = blockIdx.y, b = blockIdx.z;
  if (threadIdx.x < KS * KS) {
    wsm[threadIdx.x] = a.w[(int64_t)cc * KS * KS + threadIdx.x];
    if (GATE) wsm[KS * KS + threadIdx.x] = a.w[(int64_t)(cc + a.hidden) * KS * KS + threadIdx.x];
  }
 for (int txi = 0; txi < a.tiles_x; ++txi) {
  const int x0 = txi * TW;
  int y0 = y0_band;
  DW_OPAQUE(y0);             // keeps the per-thread row addresses out of loop-invariant hoisting (it doubled the VGPRs)
  const int tid_c = dw_tid(), tx = tid_c % G::TXN, ty = tid_c / G::TXN;
  if (txi) __syncthreads();  // everyone is done with the previous tile's LDS image
  {
    float va[DwStage<G, TW>::NV][4], vb[GATE ? DwStage<G, TW>::NV : 1][4];
    dw_load_vecs<T, G, TW, IN_PLAIN>(a, b, cc, y0, x0, vec_ok, va);
    if (GATE) dw_load_vecs<T, G, TW, IN_PLAIN>(a, b, cc + a.hidden, y0, x0, vec_ok, vb);
    dw_write_vecs<G, TW>(tile, va);
    if (GATE) dw_write_vecs<G, TW>(tile + G::LH * G::LW, vb);
    dw_stage_halo<T, G, TW, IN_PLAIN>(tile, a, b, cc, y0, x0);
    if (GATE) dw_stage_halo<T, G, TW, IN_PLAIN>(tile + G::LH * G::LW, a, b, cc + a.hidden, y0, x0);
  }
  __syncthreads();
  const int64_t HW = (int64_t)a.H * a.W;
  float o1[RPT][4];
  dw_compute<G, KS, RPT, false>(tile, wsm, a.bias ? a.bias[cc] : 0.f, ty, tx, o1);
  const int oy = y0 + RPT * ty, ox = x0 + 4 * tx;
  if (!GATE) {
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + cc) * HW, a.H, a.W, oy + r, ox, o1[r], vec_ok);
  } else {
    float o2[RPT][4];
    dw_compute<G, KS, RPT, false>(tile + G::LH * G::LW, wsm + KS * KS, a.bias ? a.bias[cc + a.hidden] : 0.f, ty, tx, o2);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      float g[4];
      if (a.out) {
        dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + cc) * HW, a.H, a.W, oy + r, ox, o1[r], vec_ok);
        dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + cc + a.hidden) * HW, a.H, a.W, oy + r, ox, o2[r], vec_ok);
        // the gate is evaluated on the values as stored (what backward will re-read)
#pragma unroll
        for (int j = 0; j < 4; ++j) { o1[r][j] = to_f32(Cvt<T>::from(o1[r][j])); o2[r][j] = to_f32(Cvt<T>::from(o2[r][j])); }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = gelu_erf(o1[r][j]) * o2[r][j];
      dw_store4<T>((T*)a.gate + ((int64_t)b * a.hidden + cc) * HW, a.H, a.W, oy + r, ox, g, vec_ok);
    }
  }
 }  // x-tiles
}

// Backward: dx = dw^T(dy) (flipped taps over the dy tile) and, from the same staged dy tile plus the x tile,
//   dW[c][ky][kx] partial = sum_{tile pixels} dy[y][x] * x[y+ky-P][x+kx-P],  db partial = sum dy.
// grid (bands, Cc, B); block partial -> part[b*bands + band][Cc*KK | Cc].  WANT_DX / WANT_DW select the halves.
template <typename T, int KS, int TW, int RPT, int MODE, bool WANT_DX, bool WANT_DW>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(DwArgs dya, const T* __restrict__ xin, float* __restrict__ part,
                                                         int vec_ok) {
  using G = DwGeom<KS, TW, RPT>;
  constexpr int KK = KS * KS;
  constexpr int P = G::P;
  __shared__ __attribute__((aligned(16))) float dyt[G::LH * G::LW];
  __shared__ __attribute__((aligned(16))) float xt[WANT_DW ? G::LH * G::LW : 1];
  __shared__ float wsm[KK];
  __shared__ float red[4][KK + 1];
  const int band = blockIdx.x, bands = gridDim.x;   // one row band per workgroup, x-tiles walked in order (see dwconv_kernel)
  const int y0_band = band * G::TH;
  const int cc = blockIdx.y, b = blockIdx.z;
  if (threadIdx.x < KK) wsm[threadIdx.x] = dya.w[(int64_t)cc * KK + threadIdx.x];
  float acc[KK + 1];
#pragma unroll
  for (int i = 0; i <= KK; ++i) acc[i] = 0.f;
 for (int txi = 0; txi < dya.tiles_x; ++txi) {
  int y0 = y0_band;
  DW_OPAQUE(y0);
  const int tid_c = dw_tid(), tx = tid_c % G::TXN, ty = tid_c / G::TXN;
  const int x0 = txi * TW, ox = x0 + 4 * tx, oy = y0 + RPT * ty;
  if (txi) __syncthreads();
  {
    DwArgs xa = dya;
    xa.in = xin;
    float va[DwStage<G, TW>::NV][4], vb[WANT_DW ? DwStage<G, TW>::NV : 1][4];
    dw_load_vecs<T, G, TW, MODE>(dya, b, cc, y0, x0, vec_ok, va);
    if (WANT_DW) dw_load_vecs<T, G, TW, IN_PLAIN>(xa, b, cc, y0, x0, vec_ok, vb);
    dw_write_vecs<G, TW>(dyt, va);
    if (WANT_DW) dw_write_vecs<G, TW>(xt, vb);
    dw_stage_halo<T, G, TW, MODE>(dyt, dya, b, cc, y0, x0);
    if (WANT_DW) dw_stage_halo<T, G, TW, IN_PLAIN>(xt, xa, b, cc, y0, x0);
  }
  __syncthreads();
  if (WANT_DX) {
    float o[RPT][4];
    dw_compute<G, KS, RPT, true>(dyt, wsm, 0.f, ty, tx, o);
    const int64_t HW = (int64_t)dya.H * dya.W;
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      dw_store4<T>((T*)dya.out + ((int64_t)b * dya.Cc + cc) * HW, dya.H, dya.W, oy + r, ox, o[r], vec_ok);
  }
  if (WANT_DW) {
    // the thread's dy values (tile centre; out-of-image entries of the staged tile are already zero)
    float d[RPT][4];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d[r][j] = dyt[(RPT * ty + r + P) * G::LW + 4 * tx + j + P];
        acc[KK] += d[r][j];
      }
#pragma unroll
    for (int wr = 0; wr < G::WIN; ++wr) {
      float row[4 + KS - 1];
      dw_window<4 + KS - 1>(&xt[(RPT * ty + wr) * G::LW + 4 * tx], row);
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int ky = wr - r;
        if (ky < 0 || ky >= KS) continue;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * KS + kx] += d[r][j] * row[j + kx];
      }
    }
  }
 }  // x-tiles
  if (WANT_DW) {
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
      float* prow = part + ((int64_t)b * bands + band) * ((int64_t)dya.Cc * (KK + 1));
      if (i < KK) prow[(int64_t)cc * KK + i] = s;
      else prow[(int64_t)dya.Cc * KK + cc] = s;
    }
  }
}

// GDFN gate backward, both halves of a channel pair in ONE workgroup (Restormer.py:90-91 backwards):
//   d1 = dg * y2 * gelu'(y1)  (gradient of conv output j),   d2 = dg * gelu(y1)  (gradient of conv output j+h)
// are formed once per pixel from one read of (dg, y1, y2), staged as two dy tiles, and pushed through the transposed
// depthwise conv of planes j and j+h; the weight/bias gradient partials of both planes come from the same tiles.
// grid (bands, hidden, B); LDS holds 4 tiles, so the tile is 2 rows per thread.
template <typename T, int KS, int TW, int RPT, bool WANT_DW>
__global__ __launch_bounds__(256) void dwconv_gate_bwd_kernel(DwArgs a, const T* __restrict__ xin, float* __restrict__ part,
                                                              int vec_ok) {
  using G = DwGeom<KS, TW, RPT>;
  using S = DwStage<G, TW>;
  constexpr int KK = KS * KS;
  constexpr int P = G::P;
  constexpr int TSZ = G::LH * G::LW;
  __shared__ __attribute__((aligned(16))) float d1t[TSZ];
  __shared__ __attribute__((aligned(16))) float d2t[TSZ];
  __shared__ __attribute__((aligned(16))) float x1t[WANT_DW ? TSZ : 1];
  __shared__ __attribute__((aligned(16))) float x2t[WANT_DW ? TSZ : 1];
  __shared__ float wsm[2 * KK];
  __shared__ float red[4][2 * (KK + 1)];
  const int band = blockIdx.x, bands = gridDim.x;   // one row band per workgroup, x-tiles walked in order (see dwconv_kernel)
  const int y0_band = band * G::TH;
  const int j = blockIdx.y, b = blockIdx.z, h = a.hidden;
  const int64_t HW = (int64_t)a.H * a.W;
  const T* dgp = (const T*)a.in + ((int64_t)b * h + j) * HW;
  const T* y1p = (const T*)a.gy + ((int64_t)b * a.Cc + j) * HW;
  const T* y2p = y1p + (int64_t)h * HW;
  const T* x1p = xin + ((int64_t)b * a.Cc + j) * HW;
  const T* x2p = x1p + (int64_t)h * HW;
  if (threadIdx.x < KK) {
    wsm[threadIdx.x] = a.w[(int64_t)j * KK + threadIdx.x];
    wsm[KK + threadIdx.x] = a.w[(int64_t)(j + h) * KK + threadIdx.x];
  }
  auto ld4 = [&](const T* plane, int gy, int gx, float* o) {
    if (vec_ok && gx + 3 < a.W) {
      Vec<T, 4>::ld(plane + (int64_t)gy * a.W + gx, o);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = gx + e < a.W ? ld1(plane + (int64_t)gy * a.W + gx + e) : 0.f;
    }
  };
  float acc[2][KK + 1];
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int i = 0; i <= KK; ++i) acc[pl][i] = 0.f;
 for (int txi = 0; txi < a.tiles_x; ++txi) {
  int y0 = y0_band;
  DW_OPAQUE(y0);
  const int tid_c = dw_tid(), tx = tid_c % G::TXN, ty = tid_c / G::TXN;
  const int x0 = txi * TW, ox = x0 + 4 * tx, oy = y0 + RPT * ty;
  if (txi) __syncthreads();
  {  // interior columns: all vector loads first, then the gate math and the LDS writes
    float vdg[S::NV][4], vy1[S::NV][4], vy2[S::NV][4], vx1[WANT_DW ? S::NV : 1][4], vx2[WANT_DW ? S::NV : 1][4];
    const int tid = dw_tid();
#pragma unroll
    for (int i = 0; i < S::NV; ++i) {
      const int e = tid + 256 * i;
      const int r = e / S::VPR, v = e - r * S::VPR;
      const int gy = y0 - P + r, gx = x0 + 4 * v;
      const bool in = e < G::LH * S::VPR && gy >= 0 && gy < a.H && gx < a.W;
#pragma unroll
      for (int q = 0; q < 4; ++q) { vdg[i][q] = 0.f; vy1[i][q] = 0.f; vy2[i][q] = 0.f; }
      if (WANT_DW) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { vx1[i][q] = 0.f; vx2[i][q] = 0.f; }
      }
      if (in) {
        ld4(dgp, gy, gx, vdg[i]); ld4(y1p, gy, gx, vy1[i]); ld4(y2p, gy, gx, vy2[i]);
        if (WANT_DW) { ld4(x1p, gy, gx, vx1[i]); ld4(x2p, gy, gx, vx2[i]); }
      }
    }
#pragma unroll
    for (int i = 0; i < S::NV; ++i) {
      const int e = tid + 256 * i;
      if (e < G::LH * S::VPR) {
        const int r = e / S::VPR, v = e - r * S::VPR;
        const int o = r * G::LW + P + 4 * v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float cdf, pdf;
          gelu_parts(vy1[i][q], cdf, pdf);
          d1t[o + q] = vdg[i][q] * vy2[i][q] * (cdf + vy1[i][q] * pdf);
          d2t[o + q] = vdg[i][q] * vy1[i][q] * cdf;
          if (WANT_DW) { x1t[o + q] = vx1[i][q]; x2t[o + q] = vx2[i][q]; }
        }
      }
    }
  }
  for (int e = dw_tid(); e < G::LH * 2 * P; e += 256) {  // halo columns
    const int r = e / (2 * P), hcol = e - r * (2 * P);
    const int c = hcol < P ? hcol : TW + hcol;
    const int gy = y0 - P + r, gx = x0 - P + c;
    float v1 = 0.f, v2 = 0.f, xa = 0.f, xb = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      const int64_t pix = (int64_t)gy * a.W + gx;
      const float dg = ld1(dgp + pix), y1 = ld1(y1p + pix), y2 = ld1(y2p + pix);
      float cdf, pdf;
      gelu_parts(y1, cdf, pdf);
      v1 = dg * y2 * (cdf + y1 * pdf);
      v2 = dg * y1 * cdf;
      if (WANT_DW) { xa = ld1(x1p + pix); xb = ld1(x2p + pix); }
    }
    d1t[r * G::LW + c] = v1;
    d2t[r * G::LW + c] = v2;
    if (WANT_DW) { x1t[r * G::LW + c] = xa; x2t[r * G::LW + c] = xb; }
  }
  __syncthreads();
  if (a.out) {
    float o[RPT][4];
    dw_compute<G, KS, RPT, true>(d1t, wsm, 0.f, ty, tx, o);
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + j) * HW, a.H, a.W, oy + r, ox, o[r], vec_ok);
    dw_compute<G, KS, RPT, true>(d2t, wsm + KK, 0.f, ty, tx, o);
#pragma unroll
    for (int r = 0; r < RPT; ++r)
      dw_store4<T>((T*)a.out + ((int64_t)b * a.Cc + j + h) * HW, a.H, a.W, oy + r, ox, o[r], vec_ok);
  }
  if (WANT_DW) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const float* dyt = pl ? d2t : d1t;
      const float* xt = pl ? x2t : x1t;
      float d[RPT][4];
#pragma unroll
      for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          d[r][q] = dyt[(RPT * ty + r + P) * G::LW + 4 * tx + q + P];
          acc[pl][KK] += d[r][q];
        }
#pragma unroll
      for (int wr = 0; wr < G::WIN; ++wr) {
        float row[4 + KS - 1];
        dw_window<4 + KS - 1>(&xt[(RPT * ty + wr) * G::LW + 4 * tx], row);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const int ky = wr - r;
          if (ky < 0 || ky >= KS) continue;
#pragma unroll
          for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[pl][ky * KS + kx] += d[r][q] * row[q + kx];
        }
      }
    }
  }
 }  // x-tiles
  if (WANT_DW) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i <= KK; ++i) {
        const float sres = wave_sum(acc[pl][i]);
        if (lane == 0) red[wv][pl * (KK + 1) + i] = sres;
      }
    __syncthreads();
    if (threadIdx.x < 2 * (KK + 1)) {
      const int pl = threadIdx.x / (KK + 1), i = threadIdx.x - pl * (KK + 1);
      const int cc = pl ? j + h : j;
      const int k = threadIdx.x;
      const float sres = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
      float* prow = part + ((int64_t)b * bands + band) * ((int64_t)a.Cc * (KK + 1));
      if (i < KK) prow[(int64_t)cc * KK + i] = sres;
      else prow[(int64_t)a.Cc * KK + cc] = sres;
    }
  }
}

struct DwTiling { int tw, rpt, th, tiles_x, bands, tiles; };
static DwTiling dw_tiling(int H, int W, int ks, int max_rpt = 4) {
  DwTiling t;
  t.tw = W >= 48 ? 64 : (W >= 24 ? 32 : 16);
  const int tyn = 256 / (t.tw / 4);
  t.rpt = (ks == 3 && H >= 2 * tyn) ? max_rpt : 1;  // tall tiles for 3x3 where the plane has the rows for them
  t.th = tyn * t.rpt;
  t.tiles_x = cdiv(W, t.tw);
  t.bands = cdiv(H, t.th);  // one workgroup per row band; it loops over the band's tiles_x tiles
  t.tiles = t.tiles_x * t.bands;
  return t;
}

#define DW_TILE_SWITCH(LAUNCH)                                                                         \
  do {                                                                                                 \
    constexpr int R4 = KS == 3 ? 4 : 1; /* RPT=4 instantiated for 3x3 only (dw_tiling never picks it otherwise) */ \
    if (tl.tw == 64) { if (tl.rpt == 4) { LAUNCH(64, R4); } else { LAUNCH(64, 1); } }                   \
    else if (tl.tw == 32) { if (tl.rpt == 4) { LAUNCH(32, R4); } else { LAUNCH(32, 1); } }              \
    else { if (tl.rpt == 4) { LAUNCH(16, R4); } else { LAUNCH(16, 1); } }                               \
  } while (0)

template <typename T, int KS, bool GATE>
static int dw_launch(DwArgs a, int B, hipStream_t st) {
  constexpr int RMAX = KS == 3 ? (GATE ? 2 : 4) : 1;  // the gate form holds two tiles and two accumulator sets: 2 rows/thread
  const DwTiling tl = dw_tiling(a.H, a.W, KS, RMAX);
  a.tiles_x = tl.tiles_x;
  const int vec_ok = (a.W % 4 == 0) && aligned16(a.out) && aligned16(a.gate) && aligned16(a.in);
  dim3 grid(tl.bands, GATE ? a.hidden : a.Cc, B), block(256);
  const double plane = (double)B * a.H * a.W * sizeof(T);
  const double chans = GATE ? (a.Cc + (a.out ? a.Cc : 0) + a.hidden) : 2.0 * a.Cc;
  ProfScope ps(st, GATE ? K_DW_GATE_FWD : K_DW_FWD, chans * plane, 2.0 * KS * KS * a.Cc * (double)B * a.H * a.W);
  if (KS == 3 && vec_ok && dws_eligible(a.H, a.W, KS))   // 3x3 on power-of-two rows: register-streaming kernels (dwstream.hip)
    return dws_fwd(a, B, GATE, false, std::is_same<T, float>::value ? MI_F32 : MI_BF16, st);
#define DW_FWD_LAUNCH(TWV, RPTV) \
  hipLaunchKernelGGL((dwconv_kernel<T, KS, TWV, RPTV, GATE>), grid, block, 0, st, a, vec_ok)
  if (tl.tw == 64) { if (tl.rpt > 1) DW_FWD_LAUNCH(64, RMAX); else DW_FWD_LAUNCH(64, 1); }
  else if (tl.tw == 32) { if (tl.rpt > 1) DW_FWD_LAUNCH(32, RMAX); else DW_FWD_LAUNCH(32, 1); }
  else { if (tl.rpt > 1) DW_FWD_LAUNCH(16, RMAX); else DW_FWD_LAUNCH(16, 1); }
#undef DW_FWD_LAUNCH
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <typename T, int KS, int MODE>
static int dw_bwd_launch(DwArgs dya, const void* xin, float* part, int B, bool want_dx, bool want_dw, int* rows_out,
                         hipStream_t st) {
  const DwTiling tl = dw_tiling(dya.H, dya.W, KS);
  dya.tiles_x = tl.tiles_x;
  *rows_out = tl.bands * B;
  const int vec_ok = (dya.W % 4 == 0) && aligned16(dya.in) && aligned16(dya.gy) && aligned16(xin) && aligned16(dya.out);
  dim3 grid(tl.bands, dya.Cc, B), block(256);
  const double plane = (double)B * dya.H * dya.W * sizeof(T);
  const double in_ch = MODE == IN_GATE_BWD ? dya.hidden + (double)dya.Cc : (double)dya.Cc;
  const int kid = want_dx ? (MODE == IN_GATE_BWD ? K_DW_GATE_BWD_DATA : K_DW_BWD_DATA) : K_DW_WGRAD;
  ProfScope ps(st, kid, (in_ch + (want_dw ? dya.Cc : 0) + (want_dx ? dya.Cc : 0)) * plane,
               ((want_dx ? 2.0 : 0.0) + (want_dw ? 2.0 : 0.0)) * KS * KS * dya.Cc * (double)B * dya.H * dya.W);
  if (KS == 3 && MODE == IN_PLAIN && vec_ok && dws_eligible(dya.H, dya.W, KS)) {
    const int dt = std::is_same<T, float>::value ? MI_F32 : MI_BF16;
    if (want_dw) return dws_bwd(dya, xin, part, B, want_dx, rows_out, dt, st);
    return dws_fwd(dya, B, false, true, dt, st);
  }
#define DW_BWD_LAUNCH(TWV, RPTV)                                                                                          \
  do {                                                                                                                    \
    if (want_dx && want_dw)                                                                                               \
      hipLaunchKernelGGL((dwconv_bwd_kernel<T, KS, TWV, RPTV, MODE, true, true>), grid, block, 0, st, dya, (const T*)xin, \
                         part, vec_ok);                                                                                   \
    else if (want_dx)                                                                                                     \
      hipLaunchKernelGGL((dwconv_bwd_kernel<T, KS, TWV, RPTV, MODE, true, false>), grid, block, 0, st, dya,               \
                         (const T*)xin, part, vec_ok);                                                                    \
    else                                                                                                                  \
      hipLaunchKernelGGL((dwconv_bwd_kernel<T, KS, TWV, RPTV, MODE, false, true>), grid, block, 0, st, dya,               \
                         (const T*)xin, part, vec_ok);                                                                    \
  } while (0)
  DW_TILE_SWITCH(DW_BWD_LAUNCH);
#undef DW_BWD_LAUNCH
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <typename T, int KS>
static int dw_gate_bwd_launch(DwArgs a, const void* xin, float* part, int B, bool want_dw, int* rows_out, hipStream_t st) {
  const DwTiling tl = dw_tiling(a.H, a.W, KS, 2);
  a.tiles_x = tl.tiles_x;
  *rows_out = tl.bands * B;
  const int vec_ok = (a.W % 4 == 0) && aligned16(a.in) && aligned16(a.gy) && aligned16(xin) && aligned16(a.out);
  dim3 grid(tl.bands, a.hidden, B), block(256);
  const double plane = (double)B * a.H * a.W * sizeof(T);
  ProfScope ps(st, a.out ? K_DW_GATE_BWD_DATA : K_DW_WGRAD,
               (a.hidden + (double)a.Cc + (want_dw ? a.Cc : 0) + (a.out ? a.Cc : 0)) * plane,
               ((a.out ? 2.0 : 0.0) + (want_dw ? 2.0 : 0.0)) * KS * KS * a.Cc * (double)B * a.H * a.W);
  if (KS == 3 && vec_ok && dws_eligible(a.H, a.W, KS))
    return dws_gate_bwd(a, xin, part, B, want_dw, rows_out, std::is_same<T, float>::value ? MI_F32 : MI_BF16, st);
#define DW_GB_LAUNCH(TWV, RPTV)                                                                                       \
  do {                                                                                                                \
    if (want_dw)                                                                                                      \
      hipLaunchKernelGGL((dwconv_gate_bwd_kernel<T, KS, TWV, RPTV, true>), grid, block, 0, st, a, (const T*)xin, part, \
                         vec_ok);                                                                                     \
    else                                                                                                              \
      hipLaunchKernelGGL((dwconv_gate_bwd_kernel<T, KS, TWV, RPTV, false>), grid, block, 0, st, a, (const T*)xin, part, \
                         vec_ok);                                                                                     \
  } while (0)
  constexpr int R2 = KS == 3 ? 2 : 1;
  if (tl.tw == 64) { if (tl.rpt == 2) DW_GB_LAUNCH(64, R2); else DW_GB_LAUNCH(64, 1); }
  else if (tl.tw == 32) { if (tl.rpt == 2) DW_GB_LAUNCH(32, R2); else DW_GB_LAUNCH(32, 1); }
  else { if (tl.rpt == 2) DW_GB_LAUNCH(16, R2); else DW_GB_LAUNCH(16, 1); }
#undef DW_GB_LAUNCH
  MI_LAUNCH_CHECK();
  return MI_OK;
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
  DW_DISPATCH(T, KS, (rc = dw_launch<T, KS, false>(a, B, (hipStream_t)stream)));
  return rc;
}

extern "C" int mi_dwconv_gate_fwd(const void* x, const float* w, const float* bias, void* y, void* g, int B, int C2, int H,
                                  int W, int ks, int dtype, void* stream) {
  MI_CHECK_ARG(x && w && g, "dwconv_gate_fwd: null pointer");
  MI_CHECK_ARG(C2 % 2 == 0, "dwconv_gate_fwd: channel count %d must be even", C2);
  MI_TRY(check_common("dwconv_gate_fwd", B, C2, H, W, ks, dtype));
  DwArgs a{x, nullptr, w, bias, y, g, C2, H, W, C2 / 2, 0};
  int rc = MI_OK;
  DW_DISPATCH(T, KS, (rc = dw_launch<T, KS, true>(a, B, (hipStream_t)stream)));
  return rc;
}

extern "C" size_t mi_dwconv_bwd_workspace(int B, int C, int H, int W, int ks) {
  if (H <= 0 || W <= 0 || C <= 0 || B <= 0) return 0;
  const DwTiling tl = dw_tiling(H, W, ks, 2);  // the gate backward uses the smaller (2 rows/thread) tiles: upper bound
  const size_t cols = (size_t)C * (ks * ks + 1);
  size_t rows = (size_t)tl.bands * B;
  if (dws_eligible(H, W, ks)) {  // the streaming kernels write one partial row per (image, band): cover both plans
    rows = std::max(rows, (size_t)dws_partial_rows(B, H, W, (int64_t)B * C));
    rows = std::max(rows, (size_t)dws_partial_rows(B, H, W, (int64_t)B * C / 2));
  }
  return align_up((rows + 2 * REDUCE_GROUPS) * cols * sizeof(float), 256);
}

static int dw_bwd_common(const void* dy_or_dg, const void* gy, const void* x, const float* w, void* dx, float* dwg,
                         float* dbg, int B, int Cc, int H, int W, int ks, int accumulate, int dtype, void* ws, int gate,
                         hipStream_t st) {
  DwArgs a{dy_or_dg, gy, w, nullptr, dx, nullptr, Cc, H, W, gate ? Cc / 2 : 0, 0};
  MI_CHECK_ARG(dx || dwg, "dwconv_bwd: nothing to compute");
  MI_CHECK_ARG(!dwg || (ws && x), "dwconv_bwd: weight gradient needs x and a workspace");
  int rc = MI_OK, rows = 0;
  float* part = (float*)ws;
  if (dwg && accumulate) {   // parameter gradients accumulated in place: partials may wait for mi_deferred_flush (common.h)
    float* arena = deferred_take(mi_dwconv_bwd_workspace(B, Cc, H, W, ks) / sizeof(float), st);
    if (arena) part = arena;
  }
  if (gate) DW_DISPATCH(T, KS, (rc = dw_gate_bwd_launch<T, KS>(a, x, part, B, dwg != nullptr, &rows, st)));
  else DW_DISPATCH(T, KS, (rc = dw_bwd_launch<T, KS, IN_PLAIN>(a, x, part, B, dx != nullptr, dwg != nullptr, &rows, st)));
  if (rc != MI_OK) return rc;
  if (dwg) {
    const int kk = ks * ks;
    const int64_t ld = (int64_t)Cc * (kk + 1);
    float* tmp = part + (int64_t)rows * ld;
    MI_TRY(launch_reduce_rows(part, dwg, rows, (int64_t)Cc * kk, ld, accumulate, 1.0f, st, tmp));
    if (dbg)
      MI_TRY(launch_reduce_rows(part + (int64_t)Cc * kk, dbg, rows, Cc, ld, accumulate, 1.0f, st,
                                tmp + (int64_t)REDUCE_GROUPS * Cc * kk));
  }
  return MI_OK;
}

extern "C" int mi_dwconv_gate_recompute_ok(int H, int W, int ks) { return dws_eligible(H, W, ks) ? 1 : 0; }

extern "C" int mi_dwconv_gate_bwd_recompute(const void* dg, const void* x, const float* w, const float* bias, void* dx,
                                            float* dwg, float* dbg, int B, int C2, int H, int W, int ks, int accumulate,
                                            int dtype, void* ws, void* stream) {
  MI_CHECK_ARG(dg && x && w, "dwconv_gate_bwd_recompute: null pointer");
  MI_CHECK_ARG(C2 % 2 == 0, "dwconv_gate_bwd_recompute: channel count %d must be even", C2);
  MI_TRY(check_common("dwconv_gate_bwd_recompute", B, C2, H, W, ks, dtype));
  MI_CHECK_ARG(dx || dwg, "dwconv_gate_bwd_recompute: nothing to compute");
  MI_CHECK_ARG(!dwg || ws, "dwconv_gate_bwd_recompute: weight gradient needs a workspace");
  MI_CHECK_ARG(dws_eligible(H, W, ks) && aligned16(dg) && aligned16(x) && aligned16(dx),
               "dwconv_gate_bwd_recompute: needs a 3x3 kernel, rows of 16..256 pixels (power of two) and 16-byte aligned "
               "planes (mi_dwconv_gate_recompute_ok); store y and use mi_dwconv_gate_bwd otherwise");
  hipStream_t st = (hipStream_t)stream;
  DwArgs a{dg, x, w, bias, dx, nullptr, C2, H, W, C2 / 2, 0};
  int rows = 0;
  float* part = (float*)ws;
  if (dwg && accumulate) {
    float* arena = deferred_take(mi_dwconv_bwd_workspace(B, C2, H, W, ks) / sizeof(float), st);
    if (arena) part = arena;
  }
  {
    const double plane = (double)B * H * W * dtype_size(dtype);
    ProfScope ps(st, dx ? K_DW_GATE_BWD_DATA : K_DW_WGRAD, (C2 / 2 + (double)C2 + (dx ? C2 : 0)) * plane,
                 ((dx ? 2.0 : 0.0) + (dwg ? 2.0 : 0.0) + 2.0) * 9 * C2 * (double)B * H * W);
    MI_TRY(dws_gate_bwd_recompute(a, part, B, dwg != nullptr, &rows, dtype, st));
  }
  if (dwg) {
    const int64_t ld = (int64_t)C2 * 10;
    float* tmp = part + (int64_t)rows * ld;
    MI_TRY(launch_reduce_rows(part, dwg, rows, (int64_t)C2 * 9, ld, accumulate, 1.0f, st, tmp));
    if (dbg) MI_TRY(launch_reduce_rows(part + (int64_t)C2 * 9, dbg, rows, C2, ld, accumulate, 1.0f, st,
                                       tmp + (int64_t)REDUCE_GROUPS * C2 * 9));
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
