// AdaIR's frequency modules (AdaIR-main/net/model.py:230-372): the plane-sized pieces that are not a Restormer block, a cross
// attention, a 1x1 / 3x3 convolution or a depthwise convolution (those run on the kernels of the other files).
//
//   box_down        bilinear resize of the input image to a feature level (model.py:321).  The U-Net levels are integer
//                   factors 1/2/4/8 below the image, where bilinear with half-pixel centres is the mean of the 2 x 2 pixels
//                   around (f i + f/2 - 1/2): a box, no general resampler needed.
//   fre_rect        the per-sample low-frequency rectangle (model.py:346-353): sigmoid(rate_conv(avgpool)) -> integer half
//                   sizes, on the device (the reference reads them back to slice a mask).
//   fre_split       FreModule.fft (model.py:343-372) WITHOUT an FFT: the mask keeps at most (2 h/128) x (2 w/128) centred
//                   frequencies (none at all on feature maps under 128 pixels, i.e. in every training step), so the low band
//                   L = P x is a direct DFT at those few frequencies (analysis: one reduction per plane; synthesis: per pixel)
//                   and  high = |x - L|,  low = |L|.  Backward:  dx = Re(g_h) - Re(P (g_h - g_l)),  g_h = dhigh (x-L)/|x-L|,
//                   g_l = dlow L/|L|  (P is a Hermitian projection).
//   chan_maxmean    SpatialGate's [max_c, mean_c] planes (model.py:240-242) with the arg-max kept for the backward.
//   plane_max       ChannelGate's global max pool (model.py:251, 264) with its arg-max.
//   chan_gate       ChannelGate's two-layer MLP on both pooled vectors + sigmoid (model.py:253-268), one workgroup.
//   refine_mix      FreRefine's  low * sigmoid(s0 + s1) + high * cw  (model.py:284-288; s = the 7x7 depthwise conv of the two
//                   SpatialGate planes, whose channel sum is the reference's dense 2->1 conv).
//   scale_add       out * para1 + y * para2 (model.py:331) with the two per-channel parameter gradients.
// Simple one-pass kernels: these planes are the latent / decoder levels (a few MB), nothing here is on the step's critical path.
#include <math.h>

#include "internal.h"

namespace mi {
namespace {

constexpr float TWO_PI = 6.283185307179586f;

template <typename T>
__global__ __launch_bounds__(256) void box_down_kernel(const T* __restrict__ in, T* __restrict__ out, int Hi, int Wi, int f,
                                                       int64_t total) {
  const int H = Hi / f, W = Wi / f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int j = (int)(e % W), i = (int)((e / W) % H);
    const int64_t plane = e / ((int64_t)H * W);
    const T* p = in + plane * Hi * Wi;
    float v;
    if (f == 1) {
      v = to_f32(p[(int64_t)i * Wi + j]);
    } else {
      const int y0 = f * i + f / 2 - 1, x0 = f * j + f / 2 - 1;
      v = 0.25f * ((to_f32(p[(int64_t)y0 * Wi + x0]) + to_f32(p[(int64_t)y0 * Wi + x0 + 1])) +
                   (to_f32(p[(int64_t)(y0 + 1) * Wi + x0]) + to_f32(p[(int64_t)(y0 + 1) * Wi + x0 + 1])));
    }
    out[e] = Cvt<T>::from(v);
  }
}

// one workgroup per sample: t = sigmoid(W2 gelu(W0 pooled)); half = (int(h/n * t0), int(w/n * t1))
__global__ __launch_bounds__(64) void fre_rect_kernel(const float* __restrict__ pooled, const float* __restrict__ w0,
                                                      const float* __restrict__ w2, int* __restrict__ half, int C, int R, int hq,
                                                      int wq) {
  __shared__ float hid[64];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < R) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w0[t * C + c] * pooled[b * C + c];
    hid[t] = 0.5f * a * (1.0f + erff(a * 0.70710678118654752f));
  }
  __syncthreads();
  if (t < 2) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += w2[t * R + r] * hid[r];
    const float s = 1.0f / (1.0f + expf(-a));
    half[2 * b + t] = (int)((float)(t == 0 ? hq : wq) * s);
  }
}

// analysis: coef[plane][f] = (1 / HW) sum_p field(p) exp(-2 pi i (k y / H + l x / W)), f = (k + hh) * 2 ww + (l + ww),
// k in [-hh, hh), l in [-ww, ww).  field = re (+ i im when im != null).  NF = compile-time bound of the frequency count.
template <typename T, int MAXH, int MAXW>
__global__ __launch_bounds__(256) void lowfreq_coef_kernel(const T* __restrict__ re, const float* __restrict__ cre,
                                                           const float* __restrict__ cim, const int* __restrict__ half,
                                                           float* __restrict__ coef, int C, int H, int W) {
  constexpr int NF = 4 * MAXH * MAXW;
  __shared__ float red[4][2 * NF];
  const int plane = blockIdx.x, b = plane / C;
  const int hh = min(half[2 * b], MAXH), ww = min(half[2 * b + 1], MAXW);
  float* cp = coef + (int64_t)plane * 2 * NF;
  if (hh <= 0 || ww <= 0) {
    for (int i = threadIdx.x; i < 2 * NF; i += 256) cp[i] = 0.f;
    return;
  }
  float ar[NF], ai[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { ar[f] = 0.f; ai[f] = 0.f; }
  const int64_t N = (int64_t)H * W;
  for (int64_t p = threadIdx.x; p < N; p += 256) {
    const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
    float vr, vi = 0.f;
    if (re) vr = to_f32(re[(int64_t)plane * N + p]);
    else { vr = cre[(int64_t)plane * N + p]; vi = cim[(int64_t)plane * N + p]; }
    float cy[2 * MAXH], sy[2 * MAXH], cx[2 * MAXW], sx[2 * MAXW];
#pragma unroll
    for (int k = 0; k < 2 * MAXH; ++k) sincosf(-TWO_PI * (float)((k - hh) * y % H) / (float)H, &sy[k], &cy[k]);
#pragma unroll
    for (int l = 0; l < 2 * MAXW; ++l) sincosf(-TWO_PI * (float)((l - ww) * x % W) / (float)W, &sx[l], &cx[l]);
#pragma unroll
    for (int k = 0; k < 2 * MAXH; ++k)
#pragma unroll
      for (int l = 0; l < 2 * MAXW; ++l) {
        if (k < 2 * hh && l < 2 * ww) {
          const float tr = cy[k] * cx[l] - sy[k] * sx[l], ti = cy[k] * sx[l] + sy[k] * cx[l];   // e^{-i(ay + ax)}
          ar[k * 2 * MAXW + l] += vr * tr - vi * ti;
          ai[k * 2 * MAXW + l] += vr * ti + vi * tr;
        }
      }
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float r = wave_sum(ar[f]), i = wave_sum(ai[f]);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][2 * f] = r; red[threadIdx.x >> 6][2 * f + 1] = i; }
  }
  __syncthreads();
  const float inv = 1.0f / (float)N;
  for (int i = threadIdx.x; i < 2 * NF; i += 256) cp[i] = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) * inv;
}

// synthesis at one pixel: L = sum_f coef[f] exp(+2 pi i (k y / H + l x / W))
template <int MAXH, int MAXW>
__device__ __forceinline__ void lowfreq_at(const float* __restrict__ cp, int hh, int ww, int y, int x, int H, int W, float& lr,
                                           float& li) {
  lr = 0.f; li = 0.f;
  if (hh <= 0 || ww <= 0) return;
  float cy[2 * MAXH], sy[2 * MAXH], cx[2 * MAXW], sx[2 * MAXW];
#pragma unroll
  for (int k = 0; k < 2 * MAXH; ++k) sincosf(TWO_PI * (float)((k - hh) * y % H) / (float)H, &sy[k], &cy[k]);
#pragma unroll
  for (int l = 0; l < 2 * MAXW; ++l) sincosf(TWO_PI * (float)((l - ww) * x % W) / (float)W, &sx[l], &cx[l]);
#pragma unroll
  for (int k = 0; k < 2 * MAXH; ++k)
#pragma unroll
    for (int l = 0; l < 2 * MAXW; ++l) {
      if (k < 2 * hh && l < 2 * ww) {
        const float tr = cy[k] * cx[l] - sy[k] * sx[l], ti = cy[k] * sx[l] + sy[k] * cx[l];
        const float cr = cp[2 * (k * 2 * MAXW + l)], ci = cp[2 * (k * 2 * MAXW + l) + 1];
        lr += cr * tr - ci * ti;
        li += cr * ti + ci * tr;
      }
    }
}

template <typename T, int MAXH, int MAXW>
__global__ __launch_bounds__(256) void fre_split_fwd_kernel(const T* __restrict__ feat, const int* __restrict__ half,
                                                            const float* __restrict__ coef, T* __restrict__ high,
                                                            T* __restrict__ low, int C, int H, int W) {
  constexpr int NF = 4 * MAXH * MAXW;
  const int plane = blockIdx.y, b = plane / C;
  const int hh = half ? min(half[2 * b], MAXH) : 0, ww = half ? min(half[2 * b + 1], MAXW) : 0;
  const int64_t N = (int64_t)H * W, p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  float lr, li;
  lowfreq_at<MAXH, MAXW>(coef ? coef + (int64_t)plane * 2 * NF : nullptr, coef ? hh : 0, ww, y, x, H, W, lr, li);
  const float v = to_f32(feat[(int64_t)plane * N + p]);
  high[(int64_t)plane * N + p] = Cvt<T>::from(sqrtf((v - lr) * (v - lr) + li * li));
  low[(int64_t)plane * N + p] = Cvt<T>::from(sqrtf(lr * lr + li * li));
}

// backward, first pass: g_h = dhigh (x - L)/|x - L|, g_l = dlow L/|L|; dfeat = Re(g_h); the complex field g_h - g_l goes to
// (fre, fim) for the projection (only when some sample has a non-empty rectangle: fre != null)
template <typename T, int MAXH, int MAXW>
__global__ __launch_bounds__(256) void fre_split_bwd1_kernel(const T* __restrict__ feat, const int* __restrict__ half,
                                                             const float* __restrict__ coef, const T* __restrict__ dhigh,
                                                             const T* __restrict__ dlow, float* __restrict__ dre,
                                                             float* __restrict__ fre, float* __restrict__ fim, int C, int H,
                                                             int W) {
  constexpr int NF = 4 * MAXH * MAXW;
  const int plane = blockIdx.y, b = plane / C;
  const int hh = half ? min(half[2 * b], MAXH) : 0, ww = half ? min(half[2 * b + 1], MAXW) : 0;
  const int64_t N = (int64_t)H * W, p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  float lr, li;
  lowfreq_at<MAXH, MAXW>(coef ? coef + (int64_t)plane * 2 * NF : nullptr, coef ? hh : 0, ww, y, x, H, W, lr, li);
  const int64_t o = (int64_t)plane * N + p;
  const float v = to_f32(feat[o]), zr = v - lr, zi = -li;
  const float za = sqrtf(zr * zr + zi * zi), la = sqrtf(lr * lr + li * li);
  const float dh = to_f32(dhigh[o]), dl = to_f32(dlow[o]);
  const float ghr = za > 0.f ? dh * zr / za : 0.f, ghi = za > 0.f ? dh * zi / za : 0.f;
  const float glr = la > 0.f ? dl * lr / la : 0.f, gli = la > 0.f ? dl * li / la : 0.f;
  dre[o] = ghr;
  if (fre) { fre[o] = ghr - glr; fim[o] = ghi - gli; }
}
// second pass: dfeat = dre - Re(P field)
template <typename T, int MAXH, int MAXW>
__global__ __launch_bounds__(256) void fre_split_bwd2_kernel(const float* __restrict__ dre, const int* __restrict__ half,
                                                             const float* __restrict__ coef_f, T* __restrict__ dfeat, int C,
                                                             int H, int W) {
  constexpr int NF = 4 * MAXH * MAXW;
  const int plane = blockIdx.y, b = plane / C;
  const int hh = half ? min(half[2 * b], MAXH) : 0, ww = half ? min(half[2 * b + 1], MAXW) : 0;
  const int64_t N = (int64_t)H * W, p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
  float lr, li;
  lowfreq_at<MAXH, MAXW>(coef_f ? coef_f + (int64_t)plane * 2 * NF : nullptr, coef_f ? hh : 0, ww, y, x, H, W, lr, li);
  const int64_t o = (int64_t)plane * N + p;
  dfeat[o] = Cvt<T>::from(dre[o] - lr);
}

// ---- SpatialGate planes: out[b][0][p] = max_c x, out[b][1][p] = mean_c x; idx[b][p] = arg max (first)
template <typename T>
__global__ __launch_bounds__(256) void chan_maxmean_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                               int* __restrict__ idx, int C, int64_t N) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const T* xb = x + (int64_t)b * C * N + p;
  float m = to_f32(xb[0]), s = m;
  int am = 0;
  for (int c = 1; c < C; ++c) {
    const float v = to_f32(xb[(int64_t)c * N]);
    s += v;
    if (v > m) { m = v; am = c; }
  }
  out[((int64_t)b * 2) * N + p] = Cvt<T>::from(m);
  out[((int64_t)b * 2 + 1) * N + p] = Cvt<T>::from(s / (float)C);
  idx[(int64_t)b * N + p] = am;
}
template <typename T>
__global__ __launch_bounds__(256) void chan_maxmean_bwd_kernel(const T* __restrict__ dout, const int* __restrict__ idx,
                                                               T* __restrict__ dx, int C, int64_t N) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const float dm = to_f32(dout[((int64_t)b * 2) * N + p]), da = to_f32(dout[((int64_t)b * 2 + 1) * N + p]) / (float)C;
  const int am = idx[(int64_t)b * N + p];
  T* db = dx + (int64_t)b * C * N + p;
  for (int c = 0; c < C; ++c) db[(int64_t)c * N] = Cvt<T>::from(da + (c == am ? dm : 0.f));
}

// ---- global max pool of every plane with its position
template <typename T>
__global__ __launch_bounds__(256) void plane_max_fwd_kernel(const T* __restrict__ x, float* __restrict__ out,
                                                            int* __restrict__ idx, int64_t N) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const T* row = x + (int64_t)blockIdx.x * N;
  float m = -INFINITY;
  int am = 0;
  for (int64_t n = threadIdx.x; n < N; n += 256) {
    const float v = to_f32(row[n]);
    if (v > m) { m = v; am = (int)n; }
  }
  sv[threadIdx.x] = m; si[threadIdx.x] = am;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float o = sv[threadIdx.x + s];
      const int oi = si[threadIdx.x + s];
      if (o > sv[threadIdx.x] || (o == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = o; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x] = sv[0]; idx[blockIdx.x] = si[0]; }
}
// dx[plane][:] = davg[plane] / N  (+ dmax[plane] at the arg max): the two pooled branches of ChannelGate in one pass
template <typename T>
__global__ __launch_bounds__(256) void pool_pair_bwd_kernel(const float* __restrict__ davg, const float* __restrict__ dmax,
                                                            const int* __restrict__ idx, T* __restrict__ dx, int64_t N) {
  const float a = davg[blockIdx.x] / (float)N, m = dmax[blockIdx.x];
  const int am = idx[blockIdx.x];
  T* row = dx + (int64_t)blockIdx.x * N;
  for (int64_t n = threadIdx.x; n < N; n += 256) row[n] = Cvt<T>::from(a + (n == am ? m : 0.f));
}

// ---- ChannelGate MLP: cw = sigmoid(W2 relu(W1 avg) + W2 relu(W1 mx)); hid [B][2][R] keeps the pre-activations
__global__ __launch_bounds__(256) void chan_gate_fwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                            const float* __restrict__ w1, const float* __restrict__ w2,
                                                            float* __restrict__ cw, float* __restrict__ hid, int C, int R) {
  extern __shared__ float sh[];   // [2][R]
  const int b = blockIdx.x;
  for (int e = threadIdx.x; e < 2 * R; e += 256) {
    const int which = e / R, r = e - which * R;
    const float* v = (which ? mx : avg) + (int64_t)b * C;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w1[r * C + c] * v[c];
    sh[e] = a;
    hid[(int64_t)b * 2 * R + e] = a;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += w2[c * R + r] * (fmaxf(sh[r], 0.f) + fmaxf(sh[R + r], 0.f));
    cw[(int64_t)b * C + c] = 1.0f / (1.0f + expf(-a));
  }
}
// one workgroup: ds = dcw cw (1 - cw); dW2 += ds (x) (relu a + relu m); dh = W2^T ds masked; dW1 += dh (x) v; dv = W1^T dh
__global__ __launch_bounds__(256) void chan_gate_bwd_kernel(const float* __restrict__ avg, const float* __restrict__ mx,
                                                            const float* __restrict__ w1, const float* __restrict__ w2,
                                                            const float* __restrict__ cw, const float* __restrict__ hid,
                                                            const float* __restrict__ dcw, float* __restrict__ davg,
                                                            float* __restrict__ dmx, float* __restrict__ dw1,
                                                            float* __restrict__ dw2, int B, int C, int R, int accumulate) {
  extern __shared__ float sh[];   // ds [B][C] | dh [B][2][R]
  float* ds = sh;
  float* dh = sh + (int64_t)B * C;
  for (int e = threadIdx.x; e < B * C; e += 256) { const float s = cw[e]; ds[e] = dcw[e] * s * (1.0f - s); }
  __syncthreads();
  for (int e = threadIdx.x; e < B * 2 * R; e += 256) {
    const int b = e / (2 * R), r = e % R;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w2[c * R + r] * ds[b * C + c];
    dh[e] = hid[e] > 0.f ? a : 0.f;
  }
  for (int e = threadIdx.x; e < C * R; e += 256) {
    const int c = e / R, r = e - c * R;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += ds[b * C + c] * (fmaxf(hid[b * 2 * R + r], 0.f) + fmaxf(hid[b * 2 * R + R + r], 0.f));
    dw2[e] = (accumulate ? dw2[e] : 0.f) + a;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * C; e += 256) {
    const int r = e / C, c = e - r * C;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dh[b * 2 * R + r] * avg[b * C + c] + dh[b * 2 * R + R + r] * mx[b * C + c];
    dw1[e] = (accumulate ? dw1[e] : 0.f) + a;
  }
  for (int e = threadIdx.x; e < B * C; e += 256) {
    const int b = e / C, c = e - b * C;
    float a = 0.f, m = 0.f;
    for (int r = 0; r < R; ++r) { a += w1[r * C + c] * dh[b * 2 * R + r]; m += w1[r * C + c] * dh[b * 2 * R + R + r]; }
    davg[e] = a; dmx[e] = m;
  }
}

// ---- FreRefine mix: out = low * sigmoid(s0 + s1) + high * cw[b][c]
template <typename T>
__global__ __launch_bounds__(256) void refine_mix_fwd_kernel(const T* __restrict__ low, const T* __restrict__ high,
                                                             const T* __restrict__ s, const float* __restrict__ cw,
                                                             T* __restrict__ out, int C, int64_t N) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const float sw = 1.0f / (1.0f + expf(-(to_f32(s[((int64_t)b * 2) * N + p]) + to_f32(s[((int64_t)b * 2 + 1) * N + p]))));
  for (int c = 0; c < C; ++c) {
    const int64_t o = ((int64_t)b * C + c) * N + p;
    out[o] = Cvt<T>::from(to_f32(low[o]) * sw + to_f32(high[o]) * cw[b * C + c]);
  }
}
// per pixel: dlow, dhigh, ds (both planes of s get the same value)
template <typename T>
__global__ __launch_bounds__(256) void refine_mix_bwd_kernel(const T* __restrict__ low, const T* __restrict__ s,
                                                             const float* __restrict__ cw, const T* __restrict__ dout,
                                                             T* __restrict__ dlow, T* __restrict__ dhigh, T* __restrict__ ds,
                                                             int C, int64_t N) {
  const int b = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const float sw = 1.0f / (1.0f + expf(-(to_f32(s[((int64_t)b * 2) * N + p]) + to_f32(s[((int64_t)b * 2 + 1) * N + p]))));
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const int64_t o = ((int64_t)b * C + c) * N + p;
    const float d = to_f32(dout[o]);
    dlow[o] = Cvt<T>::from(d * sw);
    dhigh[o] = Cvt<T>::from(d * cw[b * C + c]);
    acc += d * to_f32(low[o]);
  }
  const float g = acc * sw * (1.0f - sw);
  ds[((int64_t)b * 2) * N + p] = Cvt<T>::from(g);
  ds[((int64_t)b * 2 + 1) * N + p] = Cvt<T>::from(g);
}
// out[plane] = sum_p a[plane][p] b[plane][p]   (dcw = sum_p dout * high)
template <typename T>
__global__ __launch_bounds__(256) void plane_dot_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ out,
                                                        int64_t N) {
  __shared__ float sm[4];
  const T* ra = a + (int64_t)blockIdx.x * N;
  const T* rb = b + (int64_t)blockIdx.x * N;
  float acc = 0.f;
  for (int64_t n = threadIdx.x; n < N; n += 256) acc += to_f32(ra[n]) * to_f32(rb[n]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// ---- out = a * p1[c] + y * p2[c]
template <typename T>
__global__ __launch_bounds__(256) void scale_add_fwd_kernel(const T* __restrict__ a, const T* __restrict__ y,
                                                            const float* __restrict__ p1, const float* __restrict__ p2,
                                                            T* __restrict__ out, int C, int64_t N) {
  const int plane = blockIdx.y, c = plane % C;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int64_t o = (int64_t)plane * N + p;
  out[o] = Cvt<T>::from(to_f32(a[o]) * p1[c] + to_f32(y[o]) * p2[c]);
}
template <typename T>
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ p1,
                                                            const float* __restrict__ p2, T* __restrict__ da, T* __restrict__ dy,
                                                            int C, int64_t N) {
  const int plane = blockIdx.y, c = plane % C;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int64_t o = (int64_t)plane * N + p;
  const float d = to_f32(dout[o]);
  da[o] = Cvt<T>::from(d * p1[c]);
  dy[o] = Cvt<T>::from(d * p2[c]);
}
// dp1[c] (+)= sum_{b,p} a dout ; dp2[c] (+)= sum_{b,p} y dout : one workgroup per channel, fixed order
template <typename T>
__global__ __launch_bounds__(256) void scale_add_wgrad_kernel(const T* __restrict__ a, const T* __restrict__ y,
                                                              const T* __restrict__ dout, float* __restrict__ dp1,
                                                              float* __restrict__ dp2, int B, int C, int64_t N, int accumulate) {
  __shared__ float sm[2][4];
  const int c = blockIdx.x;
  float s1 = 0.f, s2 = 0.f;
  for (int b = 0; b < B; ++b) {
    const int64_t base = ((int64_t)b * C + c) * N;
    for (int64_t n = threadIdx.x; n < N; n += 256) {
      const float d = to_f32(dout[base + n]);
      s1 += to_f32(a[base + n]) * d;
      s2 += to_f32(y[base + n]) * d;
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s1; sm[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dp1[c] = (accumulate ? dp1[c] : 0.f) + ((sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]));
    dp2[c] = (accumulate ? dp2[c] : 0.f) + ((sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]));
  }
}

constexpr int FS_MAXH = 4, FS_MAXW = 4;   // rectangles up to 8 x 8 frequencies: feature maps up to 639 pixels (h / 128 <= 4)

#define AD_DISPATCH(dtype, CALL_F32, CALL_BF16)           \
  do {                                                    \
    if ((dtype) == MI_F32) { CALL_F32; }                  \
    else if ((dtype) == MI_BF16) { CALL_BF16; }           \
    else { set_error("adair: bad dtype %d", (int)(dtype)); return MI_ERR_ARG; } \
  } while (0)

}  // namespace
}  // namespace mi

using namespace mi;

extern "C" int mi_box_down(const void* img, void* out, int B, int C, int Hi, int Wi, int factor, int dtype, void* stream) {
  MI_CHECK_ARG(img && out && B > 0 && C > 0, "box_down: bad arguments");
  MI_CHECK_ARG(factor == 1 || (factor > 0 && factor % 2 == 0 && Hi % factor == 0 && Wi % factor == 0),
               "box_down: the level must be an even integer factor below the image (got %d for %d x %d)", factor, Hi, Wi);
  const int64_t total = (int64_t)B * C * (Hi / factor) * (Wi / factor);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  AD_DISPATCH(dtype, hipLaunchKernelGGL((box_down_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)img, (float*)out, Hi, Wi, factor, total),
              hipLaunchKernelGGL((box_down_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)img, (bf16*)out, Hi, Wi, factor, total));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fre_rect(const float* pooled, const float* w0, const float* w2, int* half, int B, int C, int R, int H, int W,
                           int n, void* stream) {
  MI_CHECK_ARG(pooled && w0 && w2 && half && B > 0 && C > 0 && R > 0 && R <= 64 && n > 0, "fre_rect: bad arguments (R <= 64)");
  hipLaunchKernelGGL(fre_rect_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, pooled, w0, w2, half, C, R, H / n, W / n);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

// coefficient buffer of one field: [B*C][2 * 64] floats
extern "C" size_t mi_fre_split_coef_bytes(int B, int C) { return (size_t)B * C * 2 * 4 * FS_MAXH * FS_MAXW * sizeof(float); }
extern "C" size_t mi_fre_split_workspace(int B, int C, int H, int W) {
  return align_up((size_t)B * C * H * W * sizeof(float), 256) * 3 + align_up(mi_fre_split_coef_bytes(B, C), 256);
}
extern "C" int mi_fre_split_max_hw(void) { return 128 * (FS_MAXH + 1) - 1; }

// half == NULL: the rectangle is empty for every sample (feature maps under n pixels): high = |feat|, low = 0, coef unused
extern "C" int mi_fre_split_fwd(const void* feat, const int* half, void* high, void* low, float* coef, int B, int C, int H, int W,
                                int dtype, void* stream) {
  MI_CHECK_ARG(feat && high && low && B > 0 && C > 0 && H > 0 && W > 0, "fre_split_fwd: bad arguments");
  MI_CHECK_ARG(!half || coef, "fre_split_fwd: a rectangle needs the coefficient buffer");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  const int64_t N = (int64_t)H * W;
  dim3 grid(cdiv(N, 256), B * C);
  MI_CHECK_ARG(grid.y <= 65535, "fre_split_fwd: too many planes");
  if (half) {
    AD_DISPATCH(dtype, hipLaunchKernelGGL((lowfreq_coef_kernel<float, FS_MAXH, FS_MAXW>), dim3(B * C), dim3(256), 0, st, (const float*)feat, nullptr, nullptr, half, coef, C, H, W),
                hipLaunchKernelGGL((lowfreq_coef_kernel<bf16, FS_MAXH, FS_MAXW>), dim3(B * C), dim3(256), 0, st, (const bf16*)feat, nullptr, nullptr, half, coef, C, H, W));
    MI_LAUNCH_CHECK();
  }
  AD_DISPATCH(dtype, hipLaunchKernelGGL((fre_split_fwd_kernel<float, FS_MAXH, FS_MAXW>), grid, dim3(256), 0, st, (const float*)feat, half, half ? coef : nullptr, (float*)high, (float*)low, C, H, W),
              hipLaunchKernelGGL((fre_split_fwd_kernel<bf16, FS_MAXH, FS_MAXW>), grid, dim3(256), 0, st, (const bf16*)feat, half, half ? coef : nullptr, (bf16*)high, (bf16*)low, C, H, W));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_fre_split_bwd(const void* feat, const int* half, const float* coef, const void* dhigh, const void* dlow,
                                void* dfeat, int B, int C, int H, int W, int dtype, void* ws, void* stream) {
  MI_CHECK_ARG(feat && dhigh && dlow && dfeat && ws && B > 0 && C > 0, "fre_split_bwd: bad arguments");
  MI_CHECK_ARG(!half || coef, "fre_split_bwd: a rectangle needs the forward's coefficients");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  const int64_t N = (int64_t)H * W;
  dim3 grid(cdiv(N, 256), B * C);
  Carver cv(ws);
  float* dre = cv.take<float>((size_t)B * C * N * sizeof(float));
  float* fre = cv.take<float>((size_t)B * C * N * sizeof(float));
  float* fim = cv.take<float>((size_t)B * C * N * sizeof(float));
  float* coef_f = cv.take<float>(mi_fre_split_coef_bytes(B, C));
  AD_DISPATCH(dtype, hipLaunchKernelGGL((fre_split_bwd1_kernel<float, FS_MAXH, FS_MAXW>), grid, dim3(256), 0, st, (const float*)feat, half, half ? coef : nullptr, (const float*)dhigh, (const float*)dlow, dre, half ? fre : nullptr, fim, C, H, W),
              hipLaunchKernelGGL((fre_split_bwd1_kernel<bf16, FS_MAXH, FS_MAXW>), grid, dim3(256), 0, st, (const bf16*)feat, half, half ? coef : nullptr, (const bf16*)dhigh, (const bf16*)dlow, dre, half ? fre : nullptr, fim, C, H, W));
  MI_LAUNCH_CHECK();
  if (half) {
    hipLaunchKernelGGL((lowfreq_coef_kernel<float, FS_MAXH, FS_MAXW>), dim3(B * C), dim3(256), 0, st, (const float*)nullptr, fre, fim, half, coef_f, C, H, W);
    MI_LAUNCH_CHECK();
  }
  AD_DISPATCH(dtype, hipLaunchKernelGGL((fre_split_bwd2_kernel<float, FS_MAXH, FS_MAXW>), grid, dim3(256), 0, st, dre, half, half ? coef_f : nullptr, (float*)dfeat, C, H, W),
              hipLaunchKernelGGL((fre_split_bwd2_kernel<bf16, FS_MAXH, FS_MAXW>), grid, dim3(256), 0, st, dre, half, half ? coef_f : nullptr, (bf16*)dfeat, C, H, W));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_chan_maxmean_fwd(const void* x, void* out, int* idx, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(x && out && idx && B > 0 && C > 0 && N > 0, "chan_maxmean_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  dim3 grid(cdiv(N, 256), B);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((chan_maxmean_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (float*)out, idx, C, N),
              hipLaunchKernelGGL((chan_maxmean_fwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, (bf16*)out, idx, C, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_chan_maxmean_bwd(const void* dout, const int* idx, void* dx, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(dout && idx && dx && B > 0 && C > 0 && N > 0, "chan_maxmean_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  dim3 grid(cdiv(N, 256), B);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((chan_maxmean_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dout, idx, (float*)dx, C, N),
              hipLaunchKernelGGL((chan_maxmean_bwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)dout, idx, (bf16*)dx, C, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_plane_max_fwd(const void* x, float* out, int* idx, int planes, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(x && out && idx && planes > 0 && N > 0, "plane_max_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((plane_max_fwd_kernel<float>), dim3(planes), dim3(256), 0, st, (const float*)x, out, idx, N),
              hipLaunchKernelGGL((plane_max_fwd_kernel<bf16>), dim3(planes), dim3(256), 0, st, (const bf16*)x, out, idx, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_pool_pair_bwd(const float* davg, const float* dmax, const int* idx, void* dx, int planes, int64_t N, int dtype,
                                void* stream) {
  MI_CHECK_ARG(davg && dmax && idx && dx && planes > 0 && N > 0, "pool_pair_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((pool_pair_bwd_kernel<float>), dim3(planes), dim3(256), 0, st, davg, dmax, idx, (float*)dx, N),
              hipLaunchKernelGGL((pool_pair_bwd_kernel<bf16>), dim3(planes), dim3(256), 0, st, davg, dmax, idx, (bf16*)dx, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_chan_gate_fwd(const float* avg, const float* mx, const float* w1, const float* w2, float* cw, float* hid, int B,
                                int C, int R, void* stream) {
  MI_CHECK_ARG(avg && mx && w1 && w2 && cw && hid && B > 0 && C > 0 && R > 0, "chan_gate_fwd: bad arguments");
  hipLaunchKernelGGL(chan_gate_fwd_kernel, dim3(B), dim3(256), 2 * R * sizeof(float), (hipStream_t)stream, avg, mx, w1, w2, cw, hid, C, R);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_chan_gate_bwd(const float* avg, const float* mx, const float* w1, const float* w2, const float* cw,
                                const float* hid, const float* dcw, float* davg, float* dmx, float* dw1, float* dw2, int B, int C,
                                int R, int accumulate, void* stream) {
  MI_CHECK_ARG(avg && mx && w1 && w2 && cw && hid && dcw && davg && dmx && dw1 && dw2 && B > 0 && C > 0 && R > 0,
               "chan_gate_bwd: bad arguments");
  const size_t lds = ((size_t)B * C + (size_t)B * 2 * R) * sizeof(float);
  MI_CHECK_ARG(lds <= 64 * 1024, "chan_gate_bwd: batch x channels too large for one workgroup (%zu bytes of LDS)", lds);
  hipLaunchKernelGGL(chan_gate_bwd_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, avg, mx, w1, w2, cw, hid, dcw, davg, dmx, dw1, dw2, B, C, R, accumulate);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_refine_mix_fwd(const void* low, const void* high, const void* s, const float* cw, void* out, int B, int C,
                                 int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(low && high && s && cw && out && B > 0 && C > 0 && N > 0, "refine_mix_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  dim3 grid(cdiv(N, 256), B);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((refine_mix_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)low, (const float*)high, (const float*)s, cw, (float*)out, C, N),
              hipLaunchKernelGGL((refine_mix_fwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)low, (const bf16*)high, (const bf16*)s, cw, (bf16*)out, C, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_refine_mix_bwd(const void* low, const void* high, const void* s, const float* cw, const void* dout, void* dlow,
                                 void* dhigh, void* ds, float* dcw, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(low && high && s && cw && dout && dlow && dhigh && ds && dcw && B > 0 && C > 0 && N > 0, "refine_mix_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  dim3 grid(cdiv(N, 256), B);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((refine_mix_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)low, (const float*)s, cw, (const float*)dout, (float*)dlow, (float*)dhigh, (float*)ds, C, N),
              hipLaunchKernelGGL((refine_mix_bwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)low, (const bf16*)s, cw, (const bf16*)dout, (bf16*)dlow, (bf16*)dhigh, (bf16*)ds, C, N));
  MI_LAUNCH_CHECK();
  AD_DISPATCH(dtype, hipLaunchKernelGGL((plane_dot_kernel<float>), dim3(B * C), dim3(256), 0, st, (const float*)dout, (const float*)high, dcw, N),
              hipLaunchKernelGGL((plane_dot_kernel<bf16>), dim3(B * C), dim3(256), 0, st, (const bf16*)dout, (const bf16*)high, dcw, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_scale_add_fwd(const void* a, const void* y, const float* p1, const float* p2, void* out, int B, int C, int64_t N,
                                int dtype, void* stream) {
  MI_CHECK_ARG(a && y && p1 && p2 && out && B > 0 && C > 0 && N > 0 && B * C <= 65535, "scale_add_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  dim3 grid(cdiv(N, 256), B * C);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((scale_add_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)a, (const float*)y, p1, p2, (float*)out, C, N),
              hipLaunchKernelGGL((scale_add_fwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)a, (const bf16*)y, p1, p2, (bf16*)out, C, N));
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_scale_add_bwd(const void* a, const void* y, const float* p1, const float* p2, const void* dout, void* da, void* dy,
                                float* dp1, float* dp2, int B, int C, int64_t N, int accumulate, int dtype, void* stream) {
  MI_CHECK_ARG(a && y && p1 && p2 && dout && da && dy && dp1 && dp2 && B > 0 && C > 0 && N > 0 && B * C <= 65535,
               "scale_add_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_ADAIR, 0.0, 0.0);
  dim3 grid(cdiv(N, 256), B * C);
  AD_DISPATCH(dtype, hipLaunchKernelGGL((scale_add_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)dout, p1, p2, (float*)da, (float*)dy, C, N),
              hipLaunchKernelGGL((scale_add_bwd_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)dout, p1, p2, (bf16*)da, (bf16*)dy, C, N));
  MI_LAUNCH_CHECK();
  AD_DISPATCH(dtype, hipLaunchKernelGGL((scale_add_wgrad_kernel<float>), dim3(C), dim3(256), 0, st, (const float*)a, (const float*)y, (const float*)dout, dp1, dp2, B, C, N, accumulate),
              hipLaunchKernelGGL((scale_add_wgrad_kernel<bf16>), dim3(C), dim3(256), 0, st, (const bf16*)a, (const bf16*)y, (const bf16*)dout, dp1, dp2, B, C, N, accumulate));
  MI_LAUNCH_CHECK();
  return MI_OK;
}
