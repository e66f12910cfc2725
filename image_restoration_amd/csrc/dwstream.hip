// Register-streaming depthwise 3x3 (Restormer.py:84,106 and their backward), the form every Restormer / MoCE-IR plane
// at 256^2 training takes (rows of 16..256 pixels).
//
// The LDS-tiled kernels in dwconv.hip run load -> barrier -> compute -> store phases, and with 3-5 workgroups per CU
// the memory pipe idles while a workgroup computes (measured: 70 % of wave time in waits at 2.3-3.3 TB/s, and removing
// the halo-column reads entirely changed nothing).  Here a WAVE owns a band of rows of one plane and streams down it:
//   * a lane owns 4 consecutive pixels of the row; 64 lanes cover a 256-pixel row (narrower rows: 64/LPR bands side by
//     side in one wave), so every load/store instruction moves whole 128-byte lines and nothing is fetched twice
//     except the two halo rows of a band;
//   * the left/right neighbours come from the adjacent lanes with one DPP wave shift each (v_mov_b32_dpp wave_shr:1 /
//     wave_shl:1) - no LDS, no barrier, waves never wait for each other;
//   * the rows of the next PF steps are already in flight (raw, unconverted) while the current ones are computed, so
//     each wave keeps PF rows x planes of loads outstanding and 8 waves/SIMD keep the HBM queues full.
// Backward forms dx with the flipped taps from the same dy window and accumulates the weight gradient as
//   dW[ky][kx] += x[y][x'] * dy[y-ky+1][x'-kx+1]
// (the dy window is needed anyway; x is then only needed at the centre row, without neighbours).
#include "internal.h"

namespace mi {
namespace {

template <typename T> struct Raw;
template <> struct Raw<bf16> {
  using V = u32x2;
  static __device__ __forceinline__ V zero() { V z = {0u, 0u}; return z; }
  static __device__ __forceinline__ void expand(const V& t, float* o) {
    o[0] = bf16_bits_to_f32(t[0] & 0xffffu); o[1] = bf16_bits_to_f32(t[0] >> 16);
    o[2] = bf16_bits_to_f32(t[1] & 0xffffu); o[3] = bf16_bits_to_f32(t[1] >> 16);
  }
};
template <> struct Raw<float> {
  using V = f32x4;
  static __device__ __forceinline__ V zero() { V z = {0.f, 0.f, 0.f, 0.f}; return z; }
  static __device__ __forceinline__ void expand(const V& t, float* o) { o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
};

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// r[0] = pixel left of the lane's 4, r[1..4] = own, r[5] = right; zero outside the row.
__device__ __forceinline__ void window_row(const float* v, bool first, bool last, float* r) {
  const float l = dpp_mov<0x138>(v[3]);  // wave_shr:1  lane i <- lane i-1
  const float g = dpp_mov<0x130>(v[0]);  // wave_shl:1  lane i <- lane i+1
  r[0] = first ? 0.f : l;
  r[1] = v[0]; r[2] = v[1]; r[3] = v[2]; r[4] = v[3];
  r[5] = last ? 0.f : g;
}
// out[j] = b + sum_{ky,kx} w[ky*3+kx] * r_ky[j+kx]
__device__ __forceinline__ void stencil(const float* w, float b, const float* r0, const float* r1, const float* r2, float* o) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = b;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += w[kx] * r0[j + kx];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += w[3 + kx] * r1[j + kx];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += w[6 + kx] * r2[j + kx];
    o[j] = s;
  }
}
// acc[ky*3+kx] += sum_j x[j] * d_(y-ky+1)[j-kx+1]   (window index j-kx+2; rows: ky=0 -> below, 1 -> centre, 2 -> above)
__device__ __forceinline__ void wgrad_row(const float* x, const float* up, const float* mid, const float* dn, float* acc) {
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[kx] += x[j] * dn[j - kx + 2];
      acc[3 + kx] += x[j] * mid[j - kx + 2];
      acc[6 + kx] += x[j] * up[j - kx + 2];
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[9] += mid[j + 1];
}
// Cache policy by row length: rows of >= 128 pixels (LPR >= 32) stream with the non-temporal loads / stores of common.h; the short
// rows of the deep levels keep the default policy - their whole tensors are 150-330 MB at bs 32, the producer has just left them in
// the Infinity Cache and the consumer follows at once (per-shape A/B inside the step, profiles/r04_d_nontemporal_ab.txt: e.g.
// dwconv_fwd C = 576 @64^2 46.7 us with nt against 32.0 without, while C = 144 @256^2 is 434 against 487).
#define DWS_NT (LPR >= 32)
template <bool NT, typename V> __device__ __forceinline__ V dws_ld(const V* p) {
  if constexpr (NT) return MI_STREAM_LD(p);
  else return *p;
}
template <bool NT, typename T> __device__ __forceinline__ void store4(T* p, const float* v) {
  if constexpr (NT) {
    Vec<T, 4>::st(p, v);
  } else if constexpr (std::is_same<T, float>::value) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
  } else {
    *reinterpret_cast<u32x2*>(p) = (u32x2){cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3])};
  }
}
__device__ __forceinline__ void copy6(float* d, const float* s) {
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = s[i];
}

// Geometry shared by the three kernels: which band of which plane this lane group works on.
template <int LPR, bool UNI> struct Unit {
  int lx, plane, band, y0;
  bool active;
  __device__ __forceinline__ Unit(int planes, int nb, int band_rows) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lx = lane % LPR;
    const int64_t u = ((int64_t)blockIdx.x * 4 + wave) * G + lane / LPR;
    plane = (int)(u / nb);
    band = (int)(u - (int64_t)plane * nb);
    // UNI: every lane group of the wave works on the same plane (nb is a multiple of the groups per wave): say so,
    // and the tap weights, bias and plane bases live in SGPRs instead of 20-40 VGPRs
    if (UNI) plane = __builtin_amdgcn_readfirstlane(plane);
    if (LPR == 64) band = __builtin_amdgcn_readfirstlane(band);
    active = plane < planes;
    y0 = band * band_rows;
  }
};
// sum over the LPR lanes of a group (xor butterfly); every lane of the group ends with the total
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int PF = 4;  // rows in flight per plane (forward / plain backward)

// ---- forward (and backward-data alone, with flip = 1) -----------------------------------------------------------------
template <typename T, bool GATE, int LPR, bool UNI>
__global__ __launch_bounds__(256) void dws_fwd_kernel(DwArgs a, int planes, int nb, int band_rows, int flip) {
  using R = Raw<T>;
  using RV = typename R::V;
  const Unit<LPR, UNI> u(planes, nb, band_rows);
  const int CH = GATE ? a.hidden : a.Cc;
  const int b = u.active ? u.plane / CH : 0, cc = u.active ? u.plane - b * CH : 0;
  const int64_t HW = (int64_t)a.H * a.W;
  const int x = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const T* in1 = (const T*)a.in + ((int64_t)b * a.Cc + cc) * HW + x;
  const T* in2 = in1 + (int64_t)a.hidden * HW;
  float w1[9], w2[GATE ? 9 : 1], b1 = 0.f, b2 = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    w1[i] = a.w[(int64_t)cc * 9 + (flip ? 8 - i : i)];
    if (GATE) w2[i] = a.w[(int64_t)(cc + a.hidden) * 9 + i];
  }
  if (a.bias) { b1 = a.bias[cc]; if (GATE) b2 = a.bias[cc + a.hidden]; }
  const int yend = min(u.y0 + band_rows, a.H);      // rows [y0, yend) are this unit's outputs
  auto ld = [&](const T* base, int y) -> RV {
    return (u.active && y >= 0 && y <= yend && y < a.H) ? dws_ld<DWS_NT>(reinterpret_cast<const RV*>(base + (int64_t)y * a.W)) : R::zero();
  };
  float p0[6], p1[6], p2[6], q0[GATE ? 6 : 1], q1[GATE ? 6 : 1], q2[GATE ? 6 : 1], v[4];
  RV c1[PF], c2[GATE ? PF : 1];
  {
    const RV ra = ld(in1, u.y0 - 1), rb = ld(in1, u.y0);
    RV sa = R::zero(), sb = R::zero();
    if (GATE) { sa = ld(in2, u.y0 - 1); sb = ld(in2, u.y0); }
#pragma unroll
    for (int i = 0; i < PF; ++i) { c1[i] = ld(in1, u.y0 + 1 + i); if (GATE) c2[i] = ld(in2, u.y0 + 1 + i); }
    R::expand(ra, v); window_row(v, first, last, p0);
    R::expand(rb, v); window_row(v, first, last, p1);
    if (GATE) {
      R::expand(sa, v); window_row(v, first, last, q0);
      R::expand(sb, v); window_row(v, first, last, q1);
    }
  }
  T* out1 = a.out ? (T*)a.out + ((int64_t)b * a.Cc + cc) * HW + x : nullptr;
  T* out2 = out1 ? out1 + (int64_t)a.hidden * HW : nullptr;
  T* outg = GATE ? (T*)a.gate + ((int64_t)b * a.hidden + cc) * HW + x : nullptr;
  for (int yy = 0; yy < band_rows; yy += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int y = u.y0 + yy + i;
      const bool st = u.active && y < yend;
      float o1[4];
      R::expand(c1[i], v);
      c1[i] = ld(in1, y + PF + 1);          // the slot is free again: keep PF rows of every plane in flight
      window_row(v, first, last, p2);
      stencil(w1, b1, p0, p1, p2, o1);
      if (!GATE) {
        if (st) store4<DWS_NT>(out1 + (int64_t)y * a.W, o1);
      } else {
        float o2[4], g[4];
        R::expand(c2[i], v);
        c2[i] = ld(in2, y + PF + 1);
        window_row(v, first, last, q2);
        stencil(w2, b2, q0, q1, q2, o2);
        if (out1) {
          if (st) { store4<DWS_NT>(out1 + (int64_t)y * a.W, o1); store4<DWS_NT>(out2 + (int64_t)y * a.W, o2); }
          // the gate is evaluated on the values as stored (what backward re-reads)
#pragma unroll
          for (int j = 0; j < 4; ++j) { o1[j] = to_f32(Cvt<T>::from(o1[j])); o2[j] = to_f32(Cvt<T>::from(o2[j])); }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = gelu_fwd<T>(o1[j]) * o2[j];
        if (st) store4<DWS_NT>(outg + (int64_t)y * a.W, g);
        copy6(q0, q1); copy6(q1, q2);
      }
      copy6(p0, p1); copy6(p1, p2);
    }
  }
}

// ---- plain backward: dx (optional) + dW/db partials ---------------------------------------------------------------------
template <typename T, int LPR, bool UNI, bool WANT_DX>
__global__ __launch_bounds__(256) void dws_bwd_kernel(DwArgs a, const T* __restrict__ xin, float* __restrict__ part,
                                                      int planes, int nb, int band_rows) {
  using R = Raw<T>;
  using RV = typename R::V;
  const Unit<LPR, UNI> u(planes, nb, band_rows);
  const int b = u.active ? u.plane / a.Cc : 0, cc = u.active ? u.plane - b * a.Cc : 0;
  const int64_t HW = (int64_t)a.H * a.W;
  const int x = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const T* dyp = (const T*)a.in + ((int64_t)b * a.Cc + cc) * HW + x;
  const T* xp = xin + ((int64_t)b * a.Cc + cc) * HW + x;
  float wf[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) wf[i] = a.w[(int64_t)cc * 9 + 8 - i];
  const int yend = min(u.y0 + band_rows, a.H);
  auto ld = [&](const T* base, int y, int ymax) -> RV {
    return (u.active && y >= 0 && y <= ymax && y < a.H) ? dws_ld<DWS_NT>(reinterpret_cast<const RV*>(base + (int64_t)y * a.W)) : R::zero();
  };
  float p0[6], p1[6], p2[6], v[4], acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.f;
  RV cd[PF], cx[PF];
  {
    const RV ra = ld(dyp, u.y0 - 1, yend), rb = ld(dyp, u.y0, yend);
#pragma unroll
    for (int i = 0; i < PF; ++i) { cd[i] = ld(dyp, u.y0 + 1 + i, yend); cx[i] = ld(xp, u.y0 + i, yend - 1); }
    R::expand(ra, v); window_row(v, first, last, p0);
    R::expand(rb, v); window_row(v, first, last, p1);
  }
  T* outp = WANT_DX ? (T*)a.out + ((int64_t)b * a.Cc + cc) * HW + x : nullptr;
  for (int yy = 0; yy < band_rows; yy += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int y = u.y0 + yy + i;
      const bool st = u.active && y < yend;
      R::expand(cd[i], v);
      cd[i] = ld(dyp, y + PF + 1, yend);
      window_row(v, first, last, p2);
      if (WANT_DX) {
        float o[4];
        stencil(wf, 0.f, p0, p1, p2, o);
        if (st) store4<DWS_NT>(outp + (int64_t)y * a.W, o);
      }
      float xr[4];
      R::expand(cx[i], xr);                 // zero beyond the band: those rows belong to the next unit
      cx[i] = ld(xp, y + PF, yend - 1);
      if (y < yend) wgrad_row(xr, p0, p1, p2, acc);
      copy6(p0, p1); copy6(p1, p2);
    }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = group_sum<LPR>(acc[i]);
  if (u.active && u.lx == 0) {
    float* prow = part + ((int64_t)b * nb + u.band) * ((int64_t)a.Cc * 10);
#pragma unroll
    for (int i = 0; i < 9; ++i) prow[(int64_t)cc * 9 + i] = acc[i];
    prow[(int64_t)a.Cc * 9 + cc] = acc[9];
  }
}

// ---- GDFN gate backward (Restormer.py:90-91 backwards), channel pair (j, j+h) per unit -----------------------------------
//   d1 = dg * y2 * gelu'(y1),  d2 = dg * gelu(y1);  dx_j = conv^T(d1, w_j), dx_{j+h} = conv^T(d2, w_{j+h});  dW/db of both.
constexpr int PFG = 2;
template <typename T, int LPR, bool UNI, bool WANT_DW>
__global__ __launch_bounds__(256) void dws_gate_bwd_kernel(DwArgs a, const T* __restrict__ xin, float* __restrict__ part,
                                                           int planes, int nb, int band_rows) {
  using R = Raw<T>;
  using RV = typename R::V;
  const Unit<LPR, UNI> u(planes, nb, band_rows);
  const int h = a.hidden;
  const int b = u.active ? u.plane / h : 0, j = u.active ? u.plane - b * h : 0;
  const int64_t HW = (int64_t)a.H * a.W;
  const int x = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const T* dgp = (const T*)a.in + ((int64_t)b * h + j) * HW + x;
  const T* y1p = (const T*)a.gy + ((int64_t)b * a.Cc + j) * HW + x;
  const T* y2p = y1p + (int64_t)h * HW;
  const T* x1p = xin + ((int64_t)b * a.Cc + j) * HW + x;
  const T* x2p = x1p + (int64_t)h * HW;
  float w1[9], w2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { w1[i] = a.w[(int64_t)j * 9 + 8 - i]; w2[i] = a.w[(int64_t)(j + h) * 9 + 8 - i]; }
  const int yend = min(u.y0 + band_rows, a.H);
  auto ld = [&](const T* base, int y, int ymax) -> RV {
    return (u.active && y >= 0 && y <= ymax && y < a.H) ? dws_ld<DWS_NT>(reinterpret_cast<const RV*>(base + (int64_t)y * a.W)) : R::zero();
  };
  // one row of (dg, y1, y2) -> window rows of d1 and d2
  auto gate_row = [&](const RV& rdg, const RV& ry1, const RV& ry2, float* r1, float* r2) {
    float dg[4], y1[4], y2[4], d1[4], d2[4];
    R::expand(rdg, dg); R::expand(ry1, y1); R::expand(ry2, y2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float cdf, pdf;
      gelu_parts(y1[q], cdf, pdf);
      d1[q] = dg[q] * y2[q] * (cdf + y1[q] * pdf);
      d2[q] = dg[q] * y1[q] * cdf;
    }
    window_row(d1, first, last, r1);
    window_row(d2, first, last, r2);
  };
  float p0[6], p1[6], p2[6], q0[6], q1[6], q2[6];
  float acc1[WANT_DW ? 10 : 1], acc2[WANT_DW ? 10 : 1];
  if (WANT_DW) {
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc1[i] = 0.f; acc2[i] = 0.f; }
  }
  RV cg[PFG], ca[PFG], cb[PFG], cx1[WANT_DW ? PFG : 1], cx2[WANT_DW ? PFG : 1];
  {
    const RV g0 = ld(dgp, u.y0 - 1, yend), a0 = ld(y1p, u.y0 - 1, yend), b0 = ld(y2p, u.y0 - 1, yend);
    const RV g1 = ld(dgp, u.y0, yend), a1 = ld(y1p, u.y0, yend), b1 = ld(y2p, u.y0, yend);
#pragma unroll
    for (int i = 0; i < PFG; ++i) {
      cg[i] = ld(dgp, u.y0 + 1 + i, yend); ca[i] = ld(y1p, u.y0 + 1 + i, yend); cb[i] = ld(y2p, u.y0 + 1 + i, yend);
      if (WANT_DW) { cx1[i] = ld(x1p, u.y0 + i, yend - 1); cx2[i] = ld(x2p, u.y0 + i, yend - 1); }
    }
    gate_row(g0, a0, b0, p0, q0);
    gate_row(g1, a1, b1, p1, q1);
  }
  T* o1p = a.out ? (T*)a.out + ((int64_t)b * a.Cc + j) * HW + x : nullptr;
  T* o2p = o1p ? o1p + (int64_t)h * HW : nullptr;
  for (int yy = 0; yy < band_rows; yy += PFG) {
#pragma unroll
    for (int i = 0; i < PFG; ++i) {
      const int y = u.y0 + yy + i;
      const bool st = u.active && y < yend;
      gate_row(cg[i], ca[i], cb[i], p2, q2);
      cg[i] = ld(dgp, y + PFG + 1, yend); ca[i] = ld(y1p, y + PFG + 1, yend); cb[i] = ld(y2p, y + PFG + 1, yend);
      if (o1p) {
        float o[4];
        stencil(w1, 0.f, p0, p1, p2, o);
        if (st) store4<DWS_NT>(o1p + (int64_t)y * a.W, o);
        stencil(w2, 0.f, q0, q1, q2, o);
        if (st) store4<DWS_NT>(o2p + (int64_t)y * a.W, o);
      }
      if (WANT_DW) {
        float xr[4];
        R::expand(cx1[i], xr);
        cx1[i] = ld(x1p, y + PFG, yend - 1);
        if (y < yend) wgrad_row(xr, p0, p1, p2, acc1);
        R::expand(cx2[i], xr);
        cx2[i] = ld(x2p, y + PFG, yend - 1);
        if (y < yend) wgrad_row(xr, q0, q1, q2, acc2);
      }
      copy6(p0, p1); copy6(p1, p2); copy6(q0, q1); copy6(q1, q2);
    }
  }
  if (WANT_DW) {
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc1[i] = group_sum<LPR>(acc1[i]); acc2[i] = group_sum<LPR>(acc2[i]); }
    if (u.active && u.lx == 0) {
      float* prow = part + ((int64_t)b * nb + u.band) * ((int64_t)a.Cc * 10);
#pragma unroll
      for (int i = 0; i < 9; ++i) { prow[(int64_t)j * 9 + i] = acc1[i]; prow[(int64_t)(j + h) * 9 + i] = acc2[i]; }
      prow[(int64_t)a.Cc * 9 + j] = acc1[9];
      prow[(int64_t)a.Cc * 9 + j + h] = acc2[9];
    }
  }
}

// ---- GDFN gate backward WITHOUT the stored conv outputs: y1, y2 are recomputed from the conv input -----------------------
// The fused form above reads (dg, y1, y2, x1, x2) = 5 planes per channel pair and the forward had to write y1, y2 for
// it: 4 of the 12 plane-passes of a GDFN's depthwise stage exist only to carry y from forward to backward.  The input
// rows x1, x2 are needed here anyway (weight gradient), so this kernel slides a 3-row window over THEM too, re-does the
// two 3x3 stencils (72 FMAs per lane and row - the VALU is idle in these kernels) and forms d1, d2 from the result:
// 3 planes read instead of 5, and the forward writes g only.  Rows: d(r) needs x rows r-1..r+1, dx(y) needs d rows
// y-1..y+1, so a band [y0, yend) reads x rows y0-2 .. yend+1 and dg rows y0-1 .. yend.
// The two planes of a pair go through identical arithmetic, so they ride in the two halves of packed-fp32 registers
// (f32x2 = (plane j, plane j+h); v_pk_fma_f32 does both FMAs in one issue slot): the recomputation doubles the stencil
// work of this kernel and the VALU, not HBM, is what bounds it.
__device__ __forceinline__ void window_row2(const f32x2* v, bool first, bool last, f32x2* r) {
  f32x2 l, g;
  l[0] = dpp_mov<0x138>(v[3][0]); l[1] = dpp_mov<0x138>(v[3][1]);   // wave_shr:1
  g[0] = dpp_mov<0x130>(v[0][0]); g[1] = dpp_mov<0x130>(v[0][1]);   // wave_shl:1
  const f32x2 zero = {0.f, 0.f};
  r[0] = first ? zero : l;
  r[1] = v[0]; r[2] = v[1]; r[3] = v[2]; r[4] = v[3];
  r[5] = last ? zero : g;
}
template <bool FLIP>
__device__ __forceinline__ void stencil2(const f32x2* w, f32x2 b, const f32x2* r0, const f32x2* r1, const f32x2* r2, f32x2* o) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x2 s = b;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += w[FLIP ? 8 - kx : kx] * r0[j + kx];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += w[FLIP ? 5 - kx : 3 + kx] * r1[j + kx];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) s += w[FLIP ? 2 - kx : 6 + kx] * r2[j + kx];
    o[j] = s;
  }
}
__device__ __forceinline__ void wgrad_row2(const f32x2* x, const f32x2* up, const f32x2* mid, const f32x2* dn, f32x2* acc) {
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[kx] += x[j] * dn[j - kx + 2];
      acc[3 + kx] += x[j] * mid[j - kx + 2];
      acc[6 + kx] += x[j] * up[j - kx + 2];
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[9] += mid[j + 1];
}
__device__ __forceinline__ void copy6x2(f32x2* d, const f32x2* s) {
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = s[i];
}

template <typename T, int LPR, bool UNI, bool WANT_DW>
__global__ __launch_bounds__(256) void dws_gate_bwd_rc_kernel(DwArgs a, float* __restrict__ part, int planes, int nb,
                                                              int band_rows) {
  using R = Raw<T>;
  using RV = typename R::V;
  const Unit<LPR, UNI> u(planes, nb, band_rows);
  const int h = a.hidden;
  const int b = u.active ? u.plane / h : 0, j = u.active ? u.plane - b * h : 0;
  const int64_t HW = (int64_t)a.H * a.W;
  const int x = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const T* dgp = (const T*)a.in + ((int64_t)b * h + j) * HW + x;          // dg [B, h, H, W]
  const T* x1p = (const T*)a.gy + ((int64_t)b * a.Cc + j) * HW + x;       // conv input [B, 2h, H, W] (passed in a.gy)
  const T* x2p = x1p + (int64_t)h * HW;
  f32x2 w[9], bias2 = {0.f, 0.f};                                          // (taps of plane j, taps of plane j+h)
#pragma unroll
  for (int i = 0; i < 9; ++i) { w[i][0] = a.w[(int64_t)j * 9 + i]; w[i][1] = a.w[(int64_t)(j + h) * 9 + i]; }
  if (a.bias) { bias2[0] = a.bias[j]; bias2[1] = a.bias[j + h]; }
  const int yend = min(u.y0 + band_rows, a.H);
  auto ld = [&](const T* base, int y, int ymax) -> RV {
    return (u.active && y >= 0 && y <= ymax && y < a.H) ? dws_ld<DWS_NT>(reinterpret_cast<const RV*>(base + (int64_t)y * a.W)) : R::zero();
  };
  f32x2 hA[6], hB[6], hC[6];            // x rows rho-1, rho, rho+1
  f32x2 p0[6], p1[6], p2[6];            // (d1, d2) rows y-1, y, y+1
  f32x2 acc[WANT_DW ? 10 : 1];
  const f32x2 zero2 = {0.f, 0.f};
  if (WANT_DW) {
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = zero2;
  }
  auto pair_row = [&](const RV& r1, const RV& r2, f32x2* out4) {
    float v1[4], v2[4];
    R::expand(r1, v1); R::expand(r2, v2);
#pragma unroll
    for (int q = 0; q < 4; ++q) { out4[q][0] = v1[q]; out4[q][1] = v2[q]; }
  };
  // d row rho from dg row rho and x rows rho-1 (hA), rho (hB), rho+1 (raw, becomes hC); rows outside the image give d = 0
  auto d_row = [&](int rho, const RV& rdg, const RV& rx1, const RV& rx2, f32x2* pd) {
    f32x2 xv[4], y[4], d[4];
    float dg[4];
    pair_row(rx1, rx2, xv);
    window_row2(xv, first, last, hC);
    stencil2<false>(w, bias2, hA, hB, hC, y);       // (y1, y2) in fp32, exactly what the forward gated
    R::expand(rdg, dg);                             // zero for rows outside the image / band halo limits
    const bool inside = rho >= 0 && rho < a.H;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float cdf, pdf;
      gelu_parts(y[q][0], cdf, pdf);
      d[q][0] = inside ? dg[q] * y[q][1] * (cdf + y[q][0] * pdf) : 0.f;
      d[q][1] = inside ? dg[q] * y[q][0] * cdf : 0.f;
    }
    window_row2(d, first, last, pd);
  };
  auto h_shift = [&]() { copy6x2(hA, hB); copy6x2(hB, hC); };

  RV cg[PFG], c1[PFG], c2[PFG];
  {
    f32x2 v[4];
    const RV ra1 = ld(x1p, u.y0 - 2, yend + 1), ra2 = ld(x2p, u.y0 - 2, yend + 1);
    const RV rb1 = ld(x1p, u.y0 - 1, yend + 1), rb2 = ld(x2p, u.y0 - 1, yend + 1);
    const RV rc1 = ld(x1p, u.y0, yend + 1), rc2 = ld(x2p, u.y0, yend + 1);
    const RV rd1 = ld(x1p, u.y0 + 1, yend + 1), rd2 = ld(x2p, u.y0 + 1, yend + 1);
    const RV g0 = ld(dgp, u.y0 - 1, yend), g1 = ld(dgp, u.y0, yend);
#pragma unroll
    for (int i = 0; i < PFG; ++i) {
      cg[i] = ld(dgp, u.y0 + 1 + i, yend);
      c1[i] = ld(x1p, u.y0 + 2 + i, yend + 1);
      c2[i] = ld(x2p, u.y0 + 2 + i, yend + 1);
    }
    pair_row(ra1, ra2, v); window_row2(v, first, last, hA);
    pair_row(rb1, rb2, v); window_row2(v, first, last, hB);
    d_row(u.y0 - 1, g0, rc1, rc2, p0); h_shift();
    d_row(u.y0, g1, rd1, rd2, p1); h_shift();
  }
  T* o1p = a.out ? (T*)a.out + ((int64_t)b * a.Cc + j) * HW + x : nullptr;
  T* o2p = o1p ? o1p + (int64_t)h * HW : nullptr;
  for (int yy = 0; yy < band_rows; yy += PFG) {
#pragma unroll
    for (int i = 0; i < PFG; ++i) {
      const int y = u.y0 + yy + i;
      const bool st = u.active && y < yend;
      // now hA = x row y, hB = x row y+1; the raw slots hold dg row y+1 and x row y+2
      d_row(y + 1, cg[i], c1[i], c2[i], p2);
      cg[i] = ld(dgp, y + PFG + 1, yend); c1[i] = ld(x1p, y + PFG + 2, yend + 1); c2[i] = ld(x2p, y + PFG + 2, yend + 1);
      if (o1p) {
        f32x2 o[4];
        stencil2<true>(w, zero2, p0, p1, p2, o);
        if (st) {
          float oa[4] = {o[0][0], o[1][0], o[2][0], o[3][0]}, ob[4] = {o[0][1], o[1][1], o[2][1], o[3][1]};
          store4<DWS_NT>(o1p + (int64_t)y * a.W, oa);
          store4<DWS_NT>(o2p + (int64_t)y * a.W, ob);
        }
      }
      if (WANT_DW && y < yend) wgrad_row2(&hA[1], p0, p1, p2, acc);    // x row y, own 4 pixels
      h_shift();
      copy6x2(p0, p1); copy6x2(p1, p2);
    }
  }
  if (WANT_DW) {
    float a1[10], a2[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { a1[i] = group_sum<LPR>(acc[i][0]); a2[i] = group_sum<LPR>(acc[i][1]); }
    if (u.active && u.lx == 0) {
      float* prow = part + ((int64_t)b * nb + u.band) * ((int64_t)a.Cc * 10);
#pragma unroll
      for (int i = 0; i < 9; ++i) { prow[(int64_t)j * 9 + i] = a1[i]; prow[(int64_t)(j + h) * 9 + i] = a2[i]; }
      prow[(int64_t)a.Cc * 9 + j] = a1[9];
      prow[(int64_t)a.Cc * 9 + j + h] = a2[9];
    }
  }
}

// ---- GDFN gate forward with the channel pair in packed-fp32 registers (same arithmetic as dws_fwd_kernel<GATE>) ----------
// Without the y store the gate forward moves 3 planes per pair and the scalar-FMA version was issue-bound at 4.25 TB/s.
template <typename T, int LPR, bool UNI>
__global__ __launch_bounds__(256) void dws_gate_fwd2_kernel(DwArgs a, int planes, int nb, int band_rows) {
  using R = Raw<T>;
  using RV = typename R::V;
  const Unit<LPR, UNI> u(planes, nb, band_rows);
  const int h = a.hidden;
  const int b = u.active ? u.plane / h : 0, cc = u.active ? u.plane - b * h : 0;
  const int64_t HW = (int64_t)a.H * a.W;
  const int x = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const T* in1 = (const T*)a.in + ((int64_t)b * a.Cc + cc) * HW + x;
  const T* in2 = in1 + (int64_t)h * HW;
  f32x2 w[9], bias2 = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) { w[i][0] = a.w[(int64_t)cc * 9 + i]; w[i][1] = a.w[(int64_t)(cc + h) * 9 + i]; }
  if (a.bias) { bias2[0] = a.bias[cc]; bias2[1] = a.bias[cc + h]; }
  const int yend = min(u.y0 + band_rows, a.H);
  auto ld = [&](const T* base, int y) -> RV {
    return (u.active && y >= 0 && y <= yend && y < a.H) ? dws_ld<DWS_NT>(reinterpret_cast<const RV*>(base + (int64_t)y * a.W)) : R::zero();
  };
  auto pair_row = [&](const RV& r1, const RV& r2, f32x2* out6) {
    float v1[4], v2[4];
    f32x2 v[4];
    R::expand(r1, v1); R::expand(r2, v2);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q][0] = v1[q]; v[q][1] = v2[q]; }
    window_row2(v, first, last, out6);
  };
  f32x2 p0[6], p1[6], p2[6];
  RV c1[PF], c2[PF];
  {
    const RV ra = ld(in1, u.y0 - 1), sa = ld(in2, u.y0 - 1), rb = ld(in1, u.y0), sb = ld(in2, u.y0);
#pragma unroll
    for (int i = 0; i < PF; ++i) { c1[i] = ld(in1, u.y0 + 1 + i); c2[i] = ld(in2, u.y0 + 1 + i); }
    pair_row(ra, sa, p0);
    pair_row(rb, sb, p1);
  }
  T* out1 = a.out ? (T*)a.out + ((int64_t)b * a.Cc + cc) * HW + x : nullptr;
  T* out2 = out1 ? out1 + (int64_t)h * HW : nullptr;
  T* outg = (T*)a.gate + ((int64_t)b * h + cc) * HW + x;
  for (int yy = 0; yy < band_rows; yy += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int y = u.y0 + yy + i;
      const bool st = u.active && y < yend;
      pair_row(c1[i], c2[i], p2);
      c1[i] = ld(in1, y + PF + 1); c2[i] = ld(in2, y + PF + 1);
      f32x2 o[4];
      stencil2<false>(w, bias2, p0, p1, p2, o);
      float o1[4] = {o[0][0], o[1][0], o[2][0], o[3][0]}, o2[4] = {o[0][1], o[1][1], o[2][1], o[3][1]}, g[4];
      if (out1) {
        if (st) { store4<DWS_NT>(out1 + (int64_t)y * a.W, o1); store4<DWS_NT>(out2 + (int64_t)y * a.W, o2); }
        // the gate is evaluated on the values as stored (what a backward that reads y sees)
#pragma unroll
        for (int q = 0; q < 4; ++q) { o1[q] = to_f32(Cvt<T>::from(o1[q])); o2[q] = to_f32(Cvt<T>::from(o2[q])); }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = gelu_fwd<T>(o1[q]) * o2[q];
      if (st) store4<DWS_NT>(outg + (int64_t)y * a.W, g);
      copy6x2(p0, p1); copy6x2(p1, p2);
    }
  }
}

// rows per band: tall bands amortise the two halo rows; short ones keep >= ~8 waves per SIMD on small launches
static int pick_band(int H, int64_t planes, int lpr) {
  const int G = 64 / lpr;
  for (int band : {64, 32, 16}) {
    const int64_t waves = planes * cdiv(H, band) / G;
    if (waves >= 8192) return band;
  }
  return 8;
}
struct Plan { int band, nb; unsigned blocks; bool uni; };
static Plan make_plan(int H, int W, int64_t planes) {
  const int lpr = W / 4, G = 64 / lpr;
  Plan p;
  p.band = pick_band(H, planes, lpr);
  p.nb = cdiv(H, p.band);
  const int64_t units = planes * p.nb;
  p.blocks = (unsigned)((units + 4 * G - 1) / (4 * G));
  p.uni = p.nb % G == 0;
  return p;
}

#define DWS_UNI_SWITCH(LPR_, ...)                                                    \
  if (p.uni) { constexpr int LPR = LPR_; constexpr bool UNI = true; __VA_ARGS__; }  \
  else { constexpr int LPR = LPR_; constexpr bool UNI = false; __VA_ARGS__; }
#define DWS_LPR_SWITCH(W_, ...)                                                              \
  switch ((W_) / 4) {                                                                        \
    case 64: { constexpr int LPR = 64; constexpr bool UNI = true; __VA_ARGS__; } break;      \
    case 32: { DWS_UNI_SWITCH(32, __VA_ARGS__) } break;                                      \
    case 16: { DWS_UNI_SWITCH(16, __VA_ARGS__) } break;                                      \
    case 8: { DWS_UNI_SWITCH(8, __VA_ARGS__) } break;                                        \
    default: { DWS_UNI_SWITCH(4, __VA_ARGS__) } break;                                       \
  }

}  // namespace

bool dws_eligible(int H, int W, int ks) {
  if (ks != 3 || H < 1) return false;
  if (MI_ENV(MI_DW_LDS)) return false;  // A/B switch: force the LDS-tiled kernels
  return W == 16 || W == 32 || W == 64 || W == 128 || W == 256;
}

int dws_partial_rows(int B, int H, int W, int64_t planes) { return make_plan(H, W, planes).nb * B; }

int dws_fwd(const DwArgs& a, int B, bool gate, bool flip, int dtype, hipStream_t st) {
  const int64_t planes = (int64_t)B * (gate ? a.hidden : a.Cc);
  const Plan p = make_plan(a.H, a.W, planes);
  dim3 grid(p.blocks), block(256);
#define DWS_FWD(T_)                                                                                                      \
  DWS_LPR_SWITCH(a.W, {                                                                                                  \
    if (gate) hipLaunchKernelGGL((dws_gate_fwd2_kernel<T_, LPR, UNI>), grid, block, 0, st, a, (int)planes, p.nb, p.band);     \
    else hipLaunchKernelGGL((dws_fwd_kernel<T_, false, LPR, UNI>), grid, block, 0, st, a, (int)planes, p.nb, p.band, flip ? 1 : 0); \
  })
  if (dtype == MI_F32) { DWS_FWD(float); } else { DWS_FWD(bf16); }
#undef DWS_FWD
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int dws_bwd(const DwArgs& a, const void* xin, float* part, int B, bool want_dx, int* rows_out, int dtype, hipStream_t st) {
  const int64_t planes = (int64_t)B * a.Cc;
  const Plan p = make_plan(a.H, a.W, planes);
  *rows_out = p.nb * B;
  dim3 grid(p.blocks), block(256);
#define DWS_BWD(T_)                                                                                              \
  DWS_LPR_SWITCH(a.W, {                                                                                          \
    if (want_dx) hipLaunchKernelGGL((dws_bwd_kernel<T_, LPR, UNI, true>), grid, block, 0, st, a, (const T_*)xin, part, \
                                    (int)planes, p.nb, p.band);                                                  \
    else hipLaunchKernelGGL((dws_bwd_kernel<T_, LPR, UNI, false>), grid, block, 0, st, a, (const T_*)xin, part,        \
                            (int)planes, p.nb, p.band);                                                          \
  })
  if (dtype == MI_F32) { DWS_BWD(float); } else { DWS_BWD(bf16); }
#undef DWS_BWD
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int dws_gate_bwd(const DwArgs& a, const void* xin, float* part, int B, bool want_dw, int* rows_out, int dtype,
                 hipStream_t st) {
  const int64_t planes = (int64_t)B * a.hidden;
  const Plan p = make_plan(a.H, a.W, planes);
  *rows_out = p.nb * B;
  dim3 grid(p.blocks), block(256);
#define DWS_GB(T_)                                                                                                    \
  DWS_LPR_SWITCH(a.W, {                                                                                               \
    if (want_dw) hipLaunchKernelGGL((dws_gate_bwd_kernel<T_, LPR, UNI, true>), grid, block, 0, st, a, (const T_*)xin, part, \
                                    (int)planes, p.nb, p.band);                                                       \
    else hipLaunchKernelGGL((dws_gate_bwd_kernel<T_, LPR, UNI, false>), grid, block, 0, st, a, (const T_*)xin, part,        \
                            (int)planes, p.nb, p.band);                                                               \
  })
  if (dtype == MI_F32) { DWS_GB(float); } else { DWS_GB(bf16); }
#undef DWS_GB
  MI_LAUNCH_CHECK();
  return MI_OK;
}


// gate backward with y recomputed from the conv input: a.in = dg, a.gy = conv input x [B, 2h, H, W], a.bias = conv bias
int dws_gate_bwd_recompute(const DwArgs& a, float* part, int B, bool want_dw, int* rows_out, int dtype, hipStream_t st) {
  const int64_t planes = (int64_t)B * a.hidden;
  const Plan p = make_plan(a.H, a.W, planes);
  *rows_out = p.nb * B;
  dim3 grid(p.blocks), block(256);
#define DWS_GBR(T_)                                                                                                       \
  DWS_LPR_SWITCH(a.W, {                                                                                                   \
    if (want_dw) hipLaunchKernelGGL((dws_gate_bwd_rc_kernel<T_, LPR, UNI, true>), grid, block, 0, st, a, part, (int)planes, \
                                    p.nb, p.band);                                                                        \
    else hipLaunchKernelGGL((dws_gate_bwd_rc_kernel<T_, LPR, UNI, false>), grid, block, 0, st, a, part, (int)planes, p.nb,  \
                            p.band);                                                                                      \
  })
  if (dtype == MI_F32) { DWS_GBR(float); } else { DWS_GBR(bf16); }
#undef DWS_GBR
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi
