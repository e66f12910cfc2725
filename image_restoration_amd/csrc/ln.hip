// Channel LayerNorm on NCHW planes (Restormer.py:25-70): the reduction runs over C with
// stride N = H*W, so lanes map to consecutive pixels (coalesced) and the 8 waves of a
// workgroup split the channels; per-pixel partial sums cross waves through LDS.
// HBM-bound: reads x once (held in registers for the two-pass variance), writes y once.
#include <stdlib.h>

#include "common.h"

namespace mi {

constexpr float LN_EPS = 1e-5f;

// A lane's VEC pixels of one channel exactly as loaded: bf16 x 2 is ONE register (two floats after conversion), so
// values that must survive between sweeps are kept raw and converted again where they are used.
template <typename T, int VEC> struct LnRaw;
template <> struct LnRaw<bf16, 2> {
  using V = unsigned int;
  static __device__ __forceinline__ V ld(const bf16* p) { return *reinterpret_cast<const unsigned int*>(p); }
  static __device__ __forceinline__ V zero() { return 0u; }
  static __device__ __forceinline__ void ex(V v, float* o) { o[0] = bf16_bits_to_f32(v & 0xffffu); o[1] = bf16_bits_to_f32(v >> 16); }
};
template <> struct LnRaw<float, 1> {
  using V = float;
  static __device__ __forceinline__ V ld(const float* p) { return *p; }
  static __device__ __forceinline__ V zero() { return 0.f; }
  static __device__ __forceinline__ void ex(V v, float* o) { o[0] = v; }
};
// the same conversion behind an opaque move, so that the compiler converts again at the point of use instead of keeping
// the floats of the first conversion alive (which would undo the point of holding the values raw)
template <typename RW> __device__ __forceinline__ void ln_ex_again(typename RW::V v, float* o) {
  asm volatile("" : "+v"(v));
  RW::ex(v, o);
}


template <typename T, int LN_WAVES, int CPT, int VEC, bool WITH_BIAS>
__global__ __launch_bounds__(64 * LN_WAVES) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ b, T* __restrict__ y,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                int C, int64_t N) {
  constexpr int TILE = 64 * VEC;
  __shared__ float red[LN_WAVES][TILE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * TILE + lane * VEC;
  const bool valid = n < N;  // VEC>1 is only used when N % VEC == 0
  const int64_t boff = (int64_t)blockIdx.y * C * N;
  const T* xb = x + boff;  // wave-uniform bases; per-channel addresses are 32-bit offsets (host checks C*N < 2^31)
  T* yb = y + boff;
  const unsigned N32 = (unsigned)N, n32 = (unsigned)n;

  float v[CPT][VEC];
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = wv + LN_WAVES * i;
    if (c < C && valid) {
      Vec<T, VEC>::ld(xb + ((unsigned)c * N32 + n32), v[i]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[i][j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] += v[i][j];
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[wv][lane * VEC + j] = s[j];
  __syncthreads();
  float mu[VEC], rs[VEC];
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LN_WAVES; ++k) t += red[k][lane * VEC + j];
    mu[j] = t * invC;
    s[j] = 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = wv + LN_WAVES * i;
    if (c < C) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { float d = v[i][j] - mu[j]; s[j] += d * d; }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[wv][lane * VEC + j] = s[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LN_WAVES; ++k) t += red[k][lane * VEC + j];
    rs[j] = 1.0f / sqrtf(t * invC + LN_EPS);
  }
  if (!valid) return;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = wv + LN_WAVES * i;
    if (c < C) {
      const float wc = w[c];
      const float bc = WITH_BIAS ? b[c] : 0.f;
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        o[j] = WITH_BIAS ? (v[i][j] - mu[j]) * rs[j] * wc + bc : v[i][j] * rs[j] * wc;
      Vec<T, VEC>::st(yb + ((unsigned)c * N32 + n32), o);
    }
  }
  if (wv == 0 && mean_out) {
    const int64_t so = (int64_t)blockIdx.y * N + n;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mean_out[so + j] = mu[j]; rstd_out[so + j] = rs[j]; }
  }
}

// Backward.  WithBias:  xh=(x-mu)r, g=dy*w, dx = r*(g - mean_c(g) - xh*mean_c(g*xh)), dw+=dy*xh, db+=dy.
// BiasFree (y = x*r*w): g=dy*w, dx = r*g - r^3*(x-mu)*mean_c(g*x), dw += dy*x*r.
// REREAD (wide C): the per-pixel channel sums are taken in a first sweep and dy,x are read again (from L2) for the
// second, so a thread holds only its 2*CPT partial sums instead of 2*CPT*VEC staged values as well.
template <typename T, int LN_WAVES, int CPT, int VEC, bool WITH_BIAS, bool REREAD>
__global__ __launch_bounds__(64 * LN_WAVES) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                const float* __restrict__ w, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const T* __restrict__ dres,
                                                                T* __restrict__ dx, float* __restrict__ part, int C, int64_t N,
                                                                int tiles_per_block, int tiles_per_image) {
  constexpr int TILE = 64 * VEC;
  constexpr int KEEP = REREAD ? 1 : CPT;
  __shared__ float red1[LN_WAVES][TILE];
  __shared__ float red2[LN_WAVES][TILE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t boff = (int64_t)blockIdx.y * C * N;
  const T* dyb = dy + boff;  // wave-uniform bases + 32-bit element offsets keep the address state out of the VGPRs
  const T* xb = x + boff;
  const T* rb = dres ? dres + boff : nullptr;
  T* dxb = dx + boff;
  const unsigned N32 = (unsigned)N;
  const float invC = 1.0f / (float)C;
  float aw[CPT], ab[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) { aw[i] = 0.f; ab[i] = 0.f; }
  (void)tiles_per_block; (void)tiles_per_image;
  {  // one pixel tile per workgroup: a tile loop makes LICM hoist every per-channel address out of it (spills)
    const int64_t n = (int64_t)blockIdx.x * TILE + lane * VEC;
    const bool valid = n < N;
    const unsigned n32 = (unsigned)n;
    float mu[VEC], rs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mu[j] = valid ? mean[(int64_t)blockIdx.y * N + n + j] : 0.f;
      rs[j] = valid ? rstd[(int64_t)blockIdx.y * N + n + j] : 0.f;
    }
    float g[KEEP][VEC], xv[KEEP][VEC];
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // sweep 1: channel sums (and, unless REREAD, the staged g = dy*w and xh / x-mu)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = wv + LN_WAVES * i;
      const float wc = c < C ? w[c] : 0.f;
      float gi[VEC], xi[VEC];
      if (c < C && valid) {
        Vec<T, VEC>::ld(dyb + ((unsigned)c * N32 + n32), gi);
        Vec<T, VEC>::ld(xb + ((unsigned)c * N32 + n32), xi);
      } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { gi[j] = 0.f; xi[j] = 0.f; }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float dyv = gi[j];
        const float gw = dyv * wc;
        if (WITH_BIAS) {
          const float xh = (xi[j] - mu[j]) * rs[j];
          s1[j] += gw; s2[j] += gw * xh;
          if (!REREAD) { aw[i] += dyv * xh; ab[i] += dyv; g[i][j] = gw; xv[i][j] = xh; }
        } else {
          s2[j] += gw * xi[j];
          if (!REREAD) { aw[i] += dyv * xi[j] * rs[j]; g[i][j] = gw; xv[i][j] = xi[j] - mu[j]; }
        }
      }
      // bound the live ranges: without this the scheduler hoists every load of the unrolled sweep to the top
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red1[wv][lane * VEC + j] = s1[j]; red2[wv][lane * VEC + j] = s2[j]; }
    __syncthreads();
    float m1[VEC], m2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = 0.f, bsum = 0.f;
#pragma unroll
      for (int k = 0; k < LN_WAVES; ++k) { a += red1[k][lane * VEC + j]; bsum += red2[k][lane * VEC + j]; }
      m1[j] = a * invC; m2[j] = bsum * invC;
    }
    __syncthreads();
    // sweep 2: dx (+dres) and, with REREAD, the weight/bias partial sums
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = wv + LN_WAVES * i;
      if (c < C && valid) {
        float gw[VEC], xq[VEC];
        if (REREAD) {
          const float wc = w[c];
          float gi[VEC], xi[VEC];
          Vec<T, VEC>::ld(dyb + ((unsigned)c * N32 + n32), gi);
          Vec<T, VEC>::ld(xb + ((unsigned)c * N32 + n32), xi);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            gw[j] = gi[j] * wc;
            if (WITH_BIAS) {
              xq[j] = (xi[j] - mu[j]) * rs[j];
              aw[i] += gi[j] * xq[j]; ab[i] += gi[j];
            } else {
              xq[j] = xi[j] - mu[j];
              aw[i] += gi[j] * xi[j] * rs[j];
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { gw[j] = g[REREAD ? 0 : i][j]; xq[j] = xv[REREAD ? 0 : i][j]; }
        }
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (WITH_BIAS) o[j] = rs[j] * (gw[j] - m1[j] - xq[j] * m2[j]);
          else o[j] = rs[j] * gw[j] - rs[j] * rs[j] * rs[j] * xq[j] * m2[j];
        }
        if (rb) {
          float r[VEC];
          Vec<T, VEC>::ld(rb + ((unsigned)c * N32 + n32), r);
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] += r[j];
        }
        Vec<T, VEC>::st(dxb + ((unsigned)c * N32 + n32), o);
      }
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // every channel is owned by exactly one wave: reduce over its 64 lanes, lane 0 writes the block partial
  float* prow = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * C);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = wv + LN_WAVES * i;
    const float sw = wave_sum(aw[i]);
    const float sb = wave_sum(ab[i]);
    if (lane == 0 && c < C) { prow[c] = sw; prow[C + c] = sb; }
  }
}

// ---- wave-owned form (C <= 384): no barrier for C <= 96, one or two for wider C --------------------------------------------
// A WAVE owns 64 x VEC pixels for up to CB = 96 channels: a lane keeps its pixels of every channel RAW in registers (one
// register per bf16 pair; 2 CB of them in backward), so the channel reduction is an in-thread loop.  All row loads of a
// tile - 256 bytes each, 12-49 KB per wave - are in flight together.  WS = 1: the four waves of a workgroup are unrelated
// tiles (no LDS, no barrier).  WS = 2 / 4 (C <= 192 / 384): WS waves share a tile, each owning a 96-channel slice, and
// exchange their per-pixel partial sums through LDS (one barrier per statistic).
template <typename T, int CB, int VEC, bool WITH_BIAS, int WS>
__global__ __launch_bounds__(256) void ln_fwd_wave_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, T* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          int C, int64_t N) {
  using RW = LnRaw<T, VEC>;
  constexpr int TILE = 64 * VEC;
  __shared__ float red[WS > 1 ? 4 : 1][WS > 1 ? TILE : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);   // wave-uniform, and the compiler may know it (scalar addressing)
  const int grp = wvu / WS, c0 = (wvu % WS) * CB;        // tile within the workgroup, first channel of this wave's slice
  const int64_t n = ((int64_t)blockIdx.x * (4 / WS) + grp) * TILE + lane * VEC;
  const bool valid = n < N;                               // whole lanes (N % VEC == 0)
  if (WS == 1 && !valid) return;
  const int64_t boff = (int64_t)blockIdx.y * C * N;
  const T* xb = x + boff;
  T* yb = y + boff;
  const unsigned N32 = (unsigned)N, n32 = (unsigned)n;
  typename RW::V raw[CB];
#pragma unroll
  for (int c = 0; c < CB; ++c) raw[c] = (c0 + c < C && valid) ? RW::ld(xb + ((unsigned)(c0 + c) * N32 + n32)) : RW::zero();
  float s[VEC], mu[VEC], rs[VEC];
  auto across = [&](float* v) {   // sum over the WS slices of the tile (no-op for WS = 1)
    if (WS > 1) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[wv][lane * VEC + j] = v[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < WS; ++k) t += red[grp * WS + k][lane * VEC + j];
        v[j] = t;
      }
    }
  };
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    float v[VEC];
    RW::ex(raw[c], v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] += v[j];   // channels >= C are zeros
  }
  across(s);
  const float invC = 1.0f / (float)C;
#pragma unroll
  for (int j = 0; j < VEC; ++j) { mu[j] = s[j] * invC; s[j] = 0.f; }
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    if (c0 + c < C) {
      float v[VEC];
      ln_ex_again<RW>(raw[c], v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float d = v[j] - mu[j]; s[j] += d * d; }
    }
  }
  across(s);
#pragma unroll
  for (int j = 0; j < VEC; ++j) rs[j] = 1.0f / sqrtf(s[j] * invC + LN_EPS);
  if (!valid) return;
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    if (c0 + c < C) {
      const float wc = w[c0 + c], bc = WITH_BIAS ? b[c0 + c] : 0.f;
      float v[VEC], o[VEC];
      ln_ex_again<RW>(raw[c], v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = WITH_BIAS ? (v[j] - mu[j]) * rs[j] * wc + bc : v[j] * rs[j] * wc;
      Vec<T, VEC>::st(yb + ((unsigned)(c0 + c) * N32 + n32), o);
    }
  }
  if (mean_out && c0 == 0) {
    const int64_t so = (int64_t)blockIdx.y * N + n;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mean_out[so + j] = mu[j]; rstd_out[so + j] = rs[j]; }
  }
}

// backward, same ownership; one partial row [2C] per tile: part[(b * tiles + tile)][c | C + c]
template <typename T, int CB, int VEC, bool WITH_BIAS, int WS, bool WGRAD, bool RPRE, int NW = 4>
__global__ __launch_bounds__(64 * NW) void ln_bwd_wave_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ w, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const T* __restrict__ dres,
                                                          T* __restrict__ dx, float* __restrict__ part, int C, int64_t N,
                                                          int tiles) {
  using RW = LnRaw<T, VEC>;
  constexpr int TILE = 64 * VEC;
  __shared__ float red[WS > 1 ? NW : 1][2][WS > 1 ? TILE : 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int grp = wvu / WS, c0 = (wvu % WS) * CB;
  const int tile = blockIdx.x * (NW / WS) + grp;
  const bool tile_ok = tile < tiles;                      // wave-uniform
  if (WS == 1 && !tile_ok) return;
  const int64_t n = (int64_t)tile * TILE + lane * VEC;
  const bool valid = tile_ok && n < N;
  const int64_t boff = (int64_t)blockIdx.y * C * N;
  const T* dyb = dy + boff;
  const T* xb = x + boff;
  const T* rb = dres ? dres + boff : nullptr;
  T* dxb = dx + boff;
  const unsigned N32 = (unsigned)N, n32 = (unsigned)n;
  typename RW::V graw[CB], xraw[CB], rraw[RPRE ? CB : 1];
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const bool in = c0 + c < C && valid;
    graw[c] = in ? RW::ld(dyb + ((unsigned)(c0 + c) * N32 + n32)) : RW::zero();
    xraw[c] = in ? RW::ld(xb + ((unsigned)(c0 + c) * N32 + n32)) : RW::zero();
    if (RPRE) rraw[c] = (in && rb) ? RW::ld(rb + ((unsigned)(c0 + c) * N32 + n32)) : RW::zero();   // residual gradient, in flight with the rest
  }
  float mu[VEC], rs[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    mu[j] = valid ? mean[(int64_t)blockIdx.y * N + n + j] : 0.f;
    rs[j] = valid ? rstd[(int64_t)blockIdx.y * N + n + j] : 0.f;
    s1[j] = 0.f; s2[j] = 0.f;
  }
  float* prow = part + ((int64_t)blockIdx.y * tiles + (tile_ok ? tile : 0)) * (2 * C);
  // sweep 1: per-pixel channel sums, and the per-channel weight / bias gradient partials of this tile
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    if (c0 + c < C) {
      const float wc = w[c0 + c];
      float gi[VEC], xi[VEC];
      RW::ex(graw[c], gi); RW::ex(xraw[c], xi);
      float aw = 0.f, ab = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gw = gi[j] * wc;
        if (WITH_BIAS) {
          const float xh = (xi[j] - mu[j]) * rs[j];
          s1[j] += gw; s2[j] += gw * xh;
          aw += gi[j] * xh; ab += gi[j];
        } else {
          s2[j] += gw * xi[j];
          aw += gi[j] * xi[j] * rs[j];
        }
      }
      if (WGRAD) {
        aw = wave_sum(aw);
        if (WITH_BIAS) ab = wave_sum(ab);
        if (lane == 0 && tile_ok) { prow[c0 + c] = aw; prow[C + c0 + c] = ab; }
      }
    }
  }
  if (WS > 1) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[wv][0][lane * VEC + j] = s1[j]; red[wv][1][lane * VEC + j] = s2[j]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < WS; ++k) { t1 += red[grp * WS + k][0][lane * VEC + j]; t2 += red[grp * WS + k][1][lane * VEC + j]; }
      s1[j] = t1; s2[j] = t2;
    }
  }
  if (!valid) return;
  const float invC = 1.0f / (float)C;
  float m1[VEC], m2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { m1[j] = s1[j] * invC; m2[j] = s2[j] * invC; }
  // sweep 2: dx (+ the residual branch's gradient)
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    if (c0 + c < C) {
      const float wc = w[c0 + c];
      float gi[VEC], xi[VEC], o[VEC];
      ln_ex_again<RW>(graw[c], gi); ln_ex_again<RW>(xraw[c], xi);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gw = gi[j] * wc;
        if (WITH_BIAS) o[j] = rs[j] * (gw - m1[j] - (xi[j] - mu[j]) * rs[j] * m2[j]);
        else o[j] = rs[j] * gw - rs[j] * rs[j] * rs[j] * (xi[j] - mu[j]) * m2[j];
      }
      if (RPRE) {
        float r[VEC];
        RW::ex(rraw[c], r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += r[j];     // zeros when there is no residual branch
      } else if (rb) {
        float r[VEC];
        Vec<T, VEC>::ld(rb + ((unsigned)(c0 + c) * N32 + n32), r);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += r[j];
      }
      Vec<T, VEC>::st(dxb + ((unsigned)(c0 + c) * N32 + n32), o);
    }
  }
}

// Shape -> (waves W, channels per thread CPT, pixels per lane VEC).  W*CPT >= C.  Measured on MI355X: wider per-lane
// vectors (8 or 16 bytes) LOSE here (ln_fwd C=48: 27 -> 40 us) because the extra registers cut the resident
// workgroups per CU, and these kernels live on bytes in flight; so bf16 keeps 2 pixels (one dword) per lane.
// which shapes take the wave-owned kernels: MI_LN_FORM=wave|block forces one form for A/B runs
// Measured with operands beyond the Infinity Cache (bs 32, profiles/r01_w_ln_forms_bs32.log): forward, wave-owned wins
// 1.2-1.75x up to C = 192 (4.7-5.1 TB/s against 2.7-2.9); backward with 24-channel slices wins 1.05-2.3x up to C = 192.
static bool ln_wave_form(int C, bool bwd) {
  if (C > 384) return false;
  if (const char* e = MI_ENV(MI_LN_FORM)) return e[0] == 'w';
  return bwd ? C <= 384 : C <= 192;   // (forward at C = 384: 17.1 us block vs 21.1 us wave at bs 32 x 32^2; backward 69 vs 31)
}
struct LnCfg { int waves, cpt, vec; };
static LnCfg ln_cfg(int C, bool bwd, bool f32) {
  const int v = f32 ? 1 : 2;
  LnCfg c;
  if (C <= 16) c = {8, 2, v};
  else if (C <= 48) c = {8, 6, v};
  else if (C <= 96) c = {8, 12, v};
  else if (C <= 192) c = {8, 24, v};
  else if (C <= 384) c = {16, 24, bwd ? 1 : v};
  else c = {16, 48, 1};
  return c;
}
static bool ln_aligned(int vec, size_t es, int64_t N, const void* a, const void* b, const void* c, const void* d) {
  const uintptr_t m = (uintptr_t)vec * es - 1;
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                         reinterpret_cast<uintptr_t>(d);
  return N % vec == 0 && (bits & m) == 0;
}

template <typename T, bool WB>
static int ln_fwd_dispatch(const T* x, const float* w, const float* b, T* y, float* mean, float* rstd, int B, int C,
                           int64_t N, hipStream_t st) {
  constexpr bool F32 = sizeof(T) == 4;
  LnCfg cf = ln_cfg(C, false, F32);
  if (C > 768) { set_error("ln_fwd: C=%d > 768 unsupported", C); return MI_ERR_ARG; }
  if (!ln_aligned(cf.vec, sizeof(T), N, x, y, nullptr, nullptr)) cf.vec = 1;
  dim3 grid(cdiv(N, 64 * cf.vec), B);
  ProfScope ps(st, K_LN_FWD, 2.0 * B * C * N * sizeof(T) + (mean ? 8.0 * B * N : 0.0), 8.0 * B * C * N);
  constexpr int WVEC = F32 ? 1 : 2;
  if (ln_wave_form(C, false) && cf.vec == WVEC) {   // wave-owned form
    const int ws = C <= 96 ? 1 : (C <= 192 ? 2 : 4);
    dim3 wgrid(cdiv(N, 64 * WVEC * (4 / ws)), B);
#define LN_FWDW_CASE(CB, WS_) \
    hipLaunchKernelGGL((ln_fwd_wave_kernel<T, CB, WVEC, WB, WS_>), wgrid, dim3(256), 0, st, x, w, b, y, mean, rstd, C, N)
    if (C <= 16) LN_FWDW_CASE(16, 1); else if (C <= 48) LN_FWDW_CASE(48, 1); else if (C <= 96) LN_FWDW_CASE(96, 1);
    else if (C <= 192) LN_FWDW_CASE(96, 2); else LN_FWDW_CASE(96, 4);
#undef LN_FWDW_CASE
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
#define LN_FWD_CASE(WV, CPT, VEC)                                                                                   \
  if (cf.waves == WV && cf.cpt == CPT && cf.vec == VEC)                                                                 \
    hipLaunchKernelGGL((ln_fwd_kernel<T, WV, CPT, VEC, WB>), grid, dim3(64 * WV), 0, st, x, w, b, y, mean, rstd, C, N)
  if constexpr (!F32) {
    LN_FWD_CASE(8, 2, 2); LN_FWD_CASE(8, 6, 2); LN_FWD_CASE(8, 12, 2); LN_FWD_CASE(8, 24, 2); LN_FWD_CASE(16, 24, 2);
  }
  LN_FWD_CASE(8, 2, 1); LN_FWD_CASE(8, 6, 1); LN_FWD_CASE(8, 12, 1); LN_FWD_CASE(8, 24, 1); LN_FWD_CASE(16, 24, 1);
  LN_FWD_CASE(16, 48, 1);
#undef LN_FWD_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <typename T, bool WB>
static int ln_bwd_dispatch(const T* dy, const T* x, const float* w, const float* mean, const float* rstd, const T* dres,
                           T* dx, float* part, int B, int C, int64_t N, int* rows_out, hipStream_t st) {
  constexpr bool F32 = sizeof(T) == 4;
  LnCfg cf = ln_cfg(C, true, F32);
  if (C > 768) { set_error("ln_bwd: C=%d > 768 unsupported", C); return MI_ERR_ARG; }
  if (!ln_aligned(cf.vec, sizeof(T), N, dy, x, dres, dx)) cf.vec = 1;
  const int gx = cdiv(N, 64 * cf.vec);
  *rows_out = gx * B;
  dim3 grid(gx, B);
  ProfScope ps(st, K_LN_BWD, (dres ? 4.0 : 3.0) * B * C * N * sizeof(T) + 8.0 * B * N, 16.0 * B * C * N);
  constexpr int WVEC = F32 ? 1 : 2;
  if (ln_wave_form(C, true) && ln_aligned(WVEC, sizeof(T), N, dy, x, dres, dx)) {
    // wave-owned form; same partial-row layout (one row per 64*VEC-pixel tile)
    const int wtiles = cdiv(N, 64 * WVEC);
    *rows_out = wtiles * B;
#define LN_BWDW_CASE(CB, WS_, NW_)                                                                                      \
    hipLaunchKernelGGL((ln_bwd_wave_kernel<T, CB, WVEC, WB, WS_, true, true, NW_>), dim3(cdiv(wtiles, NW_ / WS_), B),       \
                       dim3(64 * NW_), 0, st, dy, x, w, mean, rstd, dres, dx, part, C, N, wtiles)
    // 24-channel slices: C / 24 waves share a 128-pixel tile (measured: 96-channel slices 1.35x slower than the block
    // kernel at C = 96, 48-channel 1.28x faster, 24-channel 1.63x faster; profiles/r01_w_ln_forms_bs32.log)
    if (C <= 16) LN_BWDW_CASE(16, 1, 4); else if (C <= 24) LN_BWDW_CASE(24, 1, 4); else if (C <= 48) LN_BWDW_CASE(24, 2, 4);
    else if (C <= 96) LN_BWDW_CASE(24, 4, 4); else if (C <= 192) LN_BWDW_CASE(24, 8, 8); else LN_BWDW_CASE(24, 16, 16);
#undef LN_BWDW_CASE
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
#define LN_BWD_CASE(WV, CPT, VEC)                                                                                     \
  if (cf.waves == WV && cf.cpt == CPT && cf.vec == VEC)                                                                   \
    hipLaunchKernelGGL((ln_bwd_kernel<T, WV, CPT, VEC, WB, (CPT >= 24)>), grid, dim3(64 * WV), 0, st, dy, x, w, mean, rstd, \
                       dres, dx, part, C, N, 1, gx)
  if constexpr (!F32) {
    LN_BWD_CASE(8, 2, 2); LN_BWD_CASE(8, 6, 2); LN_BWD_CASE(8, 12, 2); LN_BWD_CASE(8, 24, 2);
  }
  LN_BWD_CASE(8, 2, 1); LN_BWD_CASE(8, 6, 1); LN_BWD_CASE(8, 12, 1); LN_BWD_CASE(8, 24, 1); LN_BWD_CASE(16, 24, 1);
  LN_BWD_CASE(16, 48, 1);
#undef LN_BWD_CASE
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_ln_fwd(const void* x, const float* w, const float* b, void* y, float* mean, float* rstd, int B, int C,
                         int64_t N, int with_bias, int dtype, void* stream) {
  MI_CHECK_ARG(x && w && y, "ln_fwd: null pointer");
  MI_CHECK_ARG(B > 0 && C > 0 && N > 0, "ln_fwd: bad shape B=%d C=%d N=%lld", B, C, (long long)N);
  MI_CHECK_ARG((int64_t)C * N < (1ll << 31), "ln_fwd: C*H*W must be below 2^31");
  MI_CHECK_ARG(!with_bias || b, "ln_fwd: with_bias needs b");
  MI_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "ln_fwd: mean/rstd must both be given or both NULL");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "ln_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI_F32)
    return with_bias ? ln_fwd_dispatch<float, true>((const float*)x, w, b, (float*)y, mean, rstd, B, C, N, st)
                     : ln_fwd_dispatch<float, false>((const float*)x, w, b, (float*)y, mean, rstd, B, C, N, st);
  return with_bias ? ln_fwd_dispatch<bf16, true>((const bf16*)x, w, b, (bf16*)y, mean, rstd, B, C, N, st)
                   : ln_fwd_dispatch<bf16, false>((const bf16*)x, w, b, (bf16*)y, mean, rstd, B, C, N, st);
}

extern "C" size_t mi_ln_bwd_workspace(int B, int C, int64_t N) {
  // one partial row of 2C floats per 64-pixel tile in the worst case (one pixel per lane), plus the two-stage scratch
  return align_up(((size_t)cdiv(N, 64) * B + 2 * REDUCE_GROUPS) * 2 * C * sizeof(float), 256);
}

extern "C" int mi_ln_bwd(const void* dy, const void* x, const float* w, const float* mean, const float* rstd,
                         const void* dres, void* dx, float* dw, float* db, int B, int C, int64_t N, int with_bias,
                         int accumulate, int dtype, void* ws, void* stream) {
  MI_CHECK_ARG(dy && x && w && mean && rstd && dx && dw && ws, "ln_bwd: null pointer");
  MI_CHECK_ARG(B > 0 && C > 0 && N > 0, "ln_bwd: bad shape");
  MI_CHECK_ARG((int64_t)C * N < (1ll << 31), "ln_bwd: C*H*W must be below 2^31");
  MI_CHECK_ARG(!with_bias || db, "ln_bwd: with_bias needs db");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "ln_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  if (accumulate) {          // parameter gradients accumulated in place: partials may wait for mi_deferred_flush (common.h)
    float* arena = deferred_take(mi_ln_bwd_workspace(B, C, N) / sizeof(float), st);
    if (arena) part = arena;
  }
  int rows = 0, rc;
  if (dtype == MI_F32) {
    rc = with_bias ? ln_bwd_dispatch<float, true>((const float*)dy, (const float*)x, w, mean, rstd, (const float*)dres,
                                                  (float*)dx, part, B, C, N, &rows, st)
                   : ln_bwd_dispatch<float, false>((const float*)dy, (const float*)x, w, mean, rstd, (const float*)dres,
                                                   (float*)dx, part, B, C, N, &rows, st);
  } else {
    rc = with_bias ? ln_bwd_dispatch<bf16, true>((const bf16*)dy, (const bf16*)x, w, mean, rstd, (const bf16*)dres,
                                                 (bf16*)dx, part, B, C, N, &rows, st)
                   : ln_bwd_dispatch<bf16, false>((const bf16*)dy, (const bf16*)x, w, mean, rstd, (const bf16*)dres,
                                                  (bf16*)dx, part, B, C, N, &rows, st);
  }
  if (rc != MI_OK) return rc;
  float* tmp = part + (int64_t)rows * 2 * C;  // two-stage scratch: [REDUCE_GROUPS][2C]
  // d gamma and d beta are adjacent column blocks of the partial rows: one reduction writes both tensors
  if (with_bias) MI_TRY(launch_reduce_rows(part, dw, rows, 2 * C, 2 * C, accumulate, 1.0f, st, tmp, db, C));
  else MI_TRY(launch_reduce_rows(part, dw, rows, C, 2 * C, accumulate, 1.0f, st, tmp));
  return MI_OK;
}
