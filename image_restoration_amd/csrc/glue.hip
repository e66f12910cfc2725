// U-Net glue, thin 3x3 convolutions: OverlapPatchEmbed 3 -> 48 (Restormer.py:156-165) and the output
// conv 96 -> 3 (+ input residual, Restormer.py:243,281).  At 256^2 these move 50-100 planes for 3, i.e. they are pure
// HBM streams (ideal 45-80 us at bs 32), and MIOpen's implicit-GEMM kernels take 0.9-1.0 ms each.  They are built here from
// the 1x1 GEMM / Gram kernels plus two streaming layout kernels:
//   im2col3x3 : x[B,C,H,W] (C tiny) -> col[B,9C,H,W],  col[c*9+ky*3+kx][y][x] = x[c][y+ky-1][x+kx-1]   (flip: shifts negated)
//   col2im3x3 : z[B,9M,H,W]         -> y[B,M,H,W],     y[m][y][x] = sum_taps z[m*9+ky*3+kx][y+ky-1][x+kx-1] (+bias +residual)
//               (flip: shifts negated - the scatter of a transposed convolution)
// so that   conv(x; W) = W[Cout, 9Cin] . im2col(x)                 when Cin is tiny  (weight gradient: Gram(dy, im2col(x)))
//           conv(x; W) = col2im(Wz[9Cout, Cin] . x)                when Cout is tiny (d z = im2col_flipped(dy))
// Same wave-streaming scheme as dwstream.hip: a wave owns a band of rows of one plane, a lane 4 pixels, neighbours by DPP.
#include "common.h"

namespace mi {
namespace {

template <typename T> struct GRaw;
template <> struct GRaw<bf16> {
  using V = u32x2;
  static __device__ __forceinline__ V zero() { V z = {0u, 0u}; return z; }
  static __device__ __forceinline__ void expand(const V& t, float* o) {
    o[0] = bf16_bits_to_f32(t[0] & 0xffffu); o[1] = bf16_bits_to_f32(t[0] >> 16);
    o[2] = bf16_bits_to_f32(t[1] & 0xffffu); o[3] = bf16_bits_to_f32(t[1] >> 16);
  }
};
template <> struct GRaw<float> {
  using V = f32x4;
  static __device__ __forceinline__ V zero() { V z = {0.f, 0.f, 0.f, 0.f}; return z; }
  static __device__ __forceinline__ void expand(const V& t, float* o) { o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
};
template <int CTRL> __device__ __forceinline__ float g_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// r[0] = pixel left of the lane's 4, r[1..4] = own, r[5] = right; zero outside the row
__device__ __forceinline__ void g_window(const float* v, bool first, bool last, float* r) {
  const float l = g_dpp<0x138>(v[3]);  // wave_shr:1
  const float g = g_dpp<0x130>(v[0]);  // wave_shl:1
  r[0] = first ? 0.f : l;
  r[1] = v[0]; r[2] = v[1]; r[3] = v[2]; r[4] = v[3];
  r[5] = last ? 0.f : g;
}

// which band of which plane this lane group works on (LPR lanes per row, 64/LPR bands side by side in a wave)
template <int LPR> struct GUnit {
  int lx, plane, y0;
  bool active;
  __device__ __forceinline__ GUnit(int planes, int nb, int band_rows) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lx = lane % LPR;
    const int64_t u = ((int64_t)blockIdx.x * 4 + wave) * G + lane / LPR;
    plane = (int)(u / nb);
    active = plane < planes;
    y0 = (int)(u - (int64_t)plane * nb) * band_rows;
  }
};

// (FLIP is a template parameter: as a run-time argument it made the window arrays dynamically indexed, the compiler put them in
//  scratch memory - 96 bytes per lane - and the kernel ran at 1.95 TB/s)
template <typename T, int LPR, bool FLIP>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ x, T* __restrict__ out, int planes, int H, int W,
                                                        int nb, int band_rows) {
  using R = GRaw<T>;
  using RV = typename R::V;
  const GUnit<LPR> u(planes, nb, band_rows);
  const int64_t HW = (int64_t)H * W;
  const int px = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const T* xp = x + (int64_t)(u.active ? u.plane : 0) * HW + px;
  T* op = out + (int64_t)(u.active ? u.plane : 0) * 9 * HW + px;
  const int yend = min(u.y0 + band_rows, H);
  auto ld = [&](int y) -> RV {
    return (u.active && y >= 0 && y <= yend && y < H) ? *reinterpret_cast<const RV*>(xp + (int64_t)y * W) : R::zero();
  };
  float w0[6], w1[6], w2[6], v[4];
  R::expand(ld(u.y0 - 1), v); g_window(v, first, last, w0);
  R::expand(ld(u.y0), v); g_window(v, first, last, w1);
  RV nxt = ld(u.y0 + 1);
  for (int yy = 0; yy < band_rows; ++yy) {
    const int y = u.y0 + yy;
    R::expand(nxt, v);
    nxt = ld(y + 2);
    g_window(v, first, last, w2);
    if (u.active && y < yend) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int src = FLIP ? 8 - tap : tap;
        const int ky = src / 3, kx = src - 3 * ky;
        const float* r = ky == 0 ? w0 : (ky == 1 ? w1 : w2);
        float o[4] = {r[kx], r[kx + 1], r[kx + 2], r[kx + 3]};
        Vec<T, 4>::st(op + (int64_t)tap * HW + (int64_t)y * W, o);
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) { w0[i] = w1[i]; w1[i] = w2[i]; }
  }
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const T* __restrict__ z, const float* __restrict__ bias,
                                                        const T* __restrict__ residual, T* __restrict__ yout, int planes,
                                                        int M, int H, int W, int nb, int band_rows, int flip) {
  using R = GRaw<T>;
  using RV = typename R::V;
  const GUnit<LPR> u(planes, nb, band_rows);
  const int64_t HW = (int64_t)H * W;
  const int px = 4 * u.lx;
  const bool first = u.lx == 0, last = u.lx == LPR - 1;
  const int plane = u.active ? u.plane : 0;
  const T* zp = z + (int64_t)plane * 9 * HW + px;
  const T* rp = residual ? residual + (int64_t)plane * HW + px : nullptr;
  T* yp = yout + (int64_t)plane * HW + px;
  const float bv = bias ? bias[plane % M] : 0.f;
  const int yend = min(u.y0 + band_rows, H);
  auto ld = [&](const T* base, int y) -> RV {
    return (u.active && y >= 0 && y < yend + 1 && y < H) ? *reinterpret_cast<const RV*>(base + (int64_t)y * W) : R::zero();
  };
  for (int yy = 0; yy < band_rows; ++yy) {
    const int y = u.y0 + yy;
    RV raw[9], rres = R::zero();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) raw[tap] = ld(zp + (int64_t)(flip ? 8 - tap : tap) * HW, y + tap / 3 - 1);   // flip: plane of the opposite tap
    if (rp && u.active && y < yend) rres = *reinterpret_cast<const RV*>(rp + (int64_t)y * W);
    float acc[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kx = tap % 3;
      float v[4], r[6];
      R::expand(raw[tap], v);
      if (kx == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      } else {
        g_window(v, first, last, r);       // uniform control flow: every lane takes part in the DPP exchange
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += r[j + kx];
      }
    }
    float rr[4];
    R::expand(rres, rr);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += rr[j];
    if (u.active && y < yend) Vec<T, 4>::st(yp + (int64_t)y * W, acc);
  }
}

// General forms (any H, W: rows that are not a power of two in 16..256 - a 512-pixel tile, an untiled 1024^2 image, the 8 x 8
// latent plane of a 64^2 input): one thread per output element, grid-stride.  Off the training path; they exist so that no plane
// the three networks can produce leaves the native kernels (round-2 verdict: silent MIOpen fallbacks).
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void im2col3x3_any_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t planes, int H, int W) {
  const int64_t HW = (int64_t)H * W, total = planes * 9 * HW;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int xx = (int)(t % W);
    const int yy = (int)((t / W) % H);
    const int64_t pt = t / HW;
    const int tap = (int)(pt % 9);
    const int64_t plane = pt / 9;
    const int src = FLIP ? 8 - tap : tap;
    const int sy = yy + src / 3 - 1, sx = xx + src % 3 - 1;
    T v = Cvt<T>::from(0.f);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = x[plane * HW + (int64_t)sy * W + sx];
    out[t] = v;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void col2im3x3_any_kernel(const T* __restrict__ z, const float* __restrict__ bias,
                                                            const T* __restrict__ residual, T* __restrict__ yout, int64_t planes,
                                                            int M, int H, int W, int flip) {
  const int64_t HW = (int64_t)H * W, total = planes * HW;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int xx = (int)(t % W);
    const int yy = (int)((t / W) % H);
    const int64_t plane = t / HW;
    float acc = bias ? bias[plane % M] : 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
      if (sy >= 0 && sy < H && sx >= 0 && sx < W)
        acc += to_f32(z[(plane * 9 + (flip ? 8 - tap : tap)) * HW + (int64_t)sy * W + sx]);
    }
    if (residual) acc += to_f32(residual[t]);
    yout[t] = Cvt<T>::from(acc);
  }
}

// PixelShuffle(2) / PixelUnshuffle(2) (Restormer.py:175-176,186-187) as one streaming pass with batch strides on both sides,
// so that the shuffled map can land directly in (or be read back from) one half of the decoder's concatenation buffer.
//   shuffle  : out[b][c][2y+i][2x+j] = in[b][4c+2i+j][y][x]      (in: [B,4c,H,W], out: [B,c,2H,2W])
//   unshuffle: out[b][4c+2i+j][y][x] = in[b][c][2y+i][2x+j]      (in: [B,c,2H,2W], out: [B,4c,H,W])
// A thread moves V pixels of one low-resolution row for both column phases j: two V-wide accesses on the planar side,
// one 2V-wide interleaved access on the high-resolution side.
template <typename T, int V, bool UNSHUFFLE>
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const T* __restrict__ in, T* __restrict__ out, int c, int H, int W,
                                                            int64_t in_bs, int64_t out_bs, int64_t total) {
  const int wv = W / V;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    int64_t r = t;
    const int xv = (int)(r % wv); r /= wv;
    const int y = (int)(r % H); r /= H;
    const int i = (int)(r % 2); r /= 2;
    const int ch = (int)(r % c);
    const int64_t b = r / c;
    const int64_t lo0 = ((int64_t)(4 * ch + 2 * i) * H + y) * W + xv * V;        // planar side, phase j = 0 (j = 1: + H*W)
    const int64_t hi = ((int64_t)ch * 2 * H + 2 * y + i) * 2 * W + 2 * xv * V;    // high-resolution side
    float a0[V], a1[V], z[2 * V];
    if (UNSHUFFLE) {
      Vec<T, V>::ld(in + b * in_bs + hi, z);
      Vec<T, V>::ld(in + b * in_bs + hi + V, z + V);
#pragma unroll
      for (int k = 0; k < V; ++k) { a0[k] = z[2 * k]; a1[k] = z[2 * k + 1]; }
      Vec<T, V>::st(out + b * out_bs + lo0, a0);
      Vec<T, V>::st(out + b * out_bs + lo0 + (int64_t)H * W, a1);
    } else {
      Vec<T, V>::ld(in + b * in_bs + lo0, a0);
      Vec<T, V>::ld(in + b * in_bs + lo0 + (int64_t)H * W, a1);
#pragma unroll
      for (int k = 0; k < V; ++k) { z[2 * k] = a0[k]; z[2 * k + 1] = a1[k]; }
      Vec<T, V>::st(out + b * out_bs + hi, z);
      Vec<T, V>::st(out + b * out_bs + hi + V, z + V);
    }
  }
}

// dst[r][0..L) = src[r][0..L) with row strides (a channel slice of a wider NCHW tensor <-> a dense tensor)
template <typename T, int V>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, int64_t src_rs, T* __restrict__ dst,
                                                        int64_t dst_rs, int64_t L, int64_t total) {
  const int64_t lv = L / V;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / lv, c = (t - r * lv) * V;
    float v[V];
    Vec<T, V>::ld(src + r * src_rs + c, v);
    Vec<T, V>::st(dst + r * dst_rs + c, v);
  }
}

struct GPlan { int band, nb; unsigned blocks; };
static GPlan g_plan(int H, int W, int64_t planes) {
  const int lpr = W / 4, G = 64 / lpr;
  GPlan p;
  p.band = 8;
  for (int band : {32, 16})
    if (planes * cdiv(H, band) / G >= 4096) { p.band = band; break; }
  p.nb = cdiv(H, p.band);
  p.blocks = (unsigned)((planes * p.nb + 4 * G - 1) / (4 * G));
  return p;
}
static bool g_fast(int H, int W) { return H >= 1 && (W == 16 || W == 32 || W == 64 || W == 128 || W == 256); }   // wave-streaming forms
static bool g_ok(int H, int W) { return H >= 1 && W >= 1; }

#define G_LPR_SWITCH(W_, ...)                                      \
  switch ((W_) / 4) {                                              \
    case 64: { constexpr int LPR = 64; __VA_ARGS__; } break;       \
    case 32: { constexpr int LPR = 32; __VA_ARGS__; } break;       \
    case 16: { constexpr int LPR = 16; __VA_ARGS__; } break;       \
    case 8: { constexpr int LPR = 8; __VA_ARGS__; } break;         \
    default: { constexpr int LPR = 4; __VA_ARGS__; } break;        \
  }

}  // namespace
}  // namespace mi

using namespace mi;

extern "C" int mi_glue3x3_ok(int H, int W) { return g_ok(H, W) ? 1 : 0; }

extern "C" int mi_im2col3x3(const void* x, void* out, int B, int C, int H, int W, int flip, int dtype, void* stream) {
  MI_CHECK_ARG(x && out && B > 0 && C > 0 && g_ok(H, W), "im2col3x3: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "im2col3x3: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int64_t planes = (int64_t)B * C;
  ProfScope ps(st, K_IM2COL, 10.0 * planes * H * W * dtype_size(dtype), 0.0);
  if (!(g_fast(H, W) && aligned16(x) && aligned16(out))) {             // general form
    const int64_t total = planes * 9 * H * W;
    const unsigned blocks = (unsigned)(total / 256 + 1 > 65536 ? 65536 : total / 256 + 1);
    if (dtype == MI_F32) {
      if (flip) hipLaunchKernelGGL((im2col3x3_any_kernel<float, true>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, planes, H, W);
      else hipLaunchKernelGGL((im2col3x3_any_kernel<float, false>), dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)out, planes, H, W);
    } else {
      if (flip) hipLaunchKernelGGL((im2col3x3_any_kernel<bf16, true>), dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)out, planes, H, W);
      else hipLaunchKernelGGL((im2col3x3_any_kernel<bf16, false>), dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)out, planes, H, W);
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
  const GPlan p = g_plan(H, W, planes);
  if (dtype == MI_F32) {
    if (flip) { G_LPR_SWITCH(W, hipLaunchKernelGGL((im2col3x3_kernel<float, LPR, true>), dim3(p.blocks), dim3(256), 0, st, (const float*)x,
                                                   (float*)out, (int)planes, H, W, p.nb, p.band)); }
    else { G_LPR_SWITCH(W, hipLaunchKernelGGL((im2col3x3_kernel<float, LPR, false>), dim3(p.blocks), dim3(256), 0, st, (const float*)x,
                                              (float*)out, (int)planes, H, W, p.nb, p.band)); }
  } else {
    if (flip) { G_LPR_SWITCH(W, hipLaunchKernelGGL((im2col3x3_kernel<bf16, LPR, true>), dim3(p.blocks), dim3(256), 0, st, (const bf16*)x,
                                                   (bf16*)out, (int)planes, H, W, p.nb, p.band)); }
    else { G_LPR_SWITCH(W, hipLaunchKernelGGL((im2col3x3_kernel<bf16, LPR, false>), dim3(p.blocks), dim3(256), 0, st, (const bf16*)x,
                                              (bf16*)out, (int)planes, H, W, p.nb, p.band)); }
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_col2im3x3(const void* z, const float* bias, const void* residual, void* y, int B, int M, int H, int W,
                            int flip, int dtype, void* stream) {
  MI_CHECK_ARG(z && y && B > 0 && M > 0 && g_ok(H, W), "col2im3x3: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "col2im3x3: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int64_t planes = (int64_t)B * M;
  ProfScope ps(st, K_COL2IM, (10.0 + (residual ? 1.0 : 0.0)) * planes * H * W * dtype_size(dtype), 9.0 * planes * H * W);
  if (!(g_fast(H, W) && aligned16(z) && aligned16(y) && aligned16(residual))) {   // general form
    const int64_t total = planes * H * W;
    const unsigned blocks = (unsigned)(total / 256 + 1 > 65536 ? 65536 : total / 256 + 1);
    if (dtype == MI_F32)
      hipLaunchKernelGGL((col2im3x3_any_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)z, bias, (const float*)residual,
                         (float*)y, planes, M, H, W, flip);
    else
      hipLaunchKernelGGL((col2im3x3_any_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)z, bias, (const bf16*)residual,
                         (bf16*)y, planes, M, H, W, flip);
    MI_LAUNCH_CHECK();
    return MI_OK;
  }
  const GPlan p = g_plan(H, W, planes);
  if (dtype == MI_F32) {
    G_LPR_SWITCH(W, hipLaunchKernelGGL((col2im3x3_kernel<float, LPR>), dim3(p.blocks), dim3(256), 0, st, (const float*)z, bias,
                                       (const float*)residual, (float*)y, (int)planes, M, H, W, p.nb, p.band, flip));
  } else {
    G_LPR_SWITCH(W, hipLaunchKernelGGL((col2im3x3_kernel<bf16, LPR>), dim3(p.blocks), dim3(256), 0, st, (const bf16*)z, bias,
                                       (const bf16*)residual, (bf16*)y, (int)planes, M, H, W, p.nb, p.band, flip));
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_pixel_shuffle2(const void* in, int64_t in_bs, void* out, int64_t out_bs, int B, int c, int H, int W,
                                 int unshuffle, int dtype, void* stream) {
  MI_CHECK_ARG(in && out && B > 0 && c > 0 && H > 0 && W > 0, "pixel_shuffle2: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "pixel_shuffle2: bad dtype %d", dtype);
  const int64_t dense_lo = (int64_t)4 * c * H * W;  // both sides hold 4*c*H*W elements per image
  const int64_t ibs = in_bs ? in_bs : dense_lo, obs = out_bs ? out_bs : dense_lo;
  int V = dtype == MI_BF16 ? 8 : 4;                 // 16-byte accesses on the planar side ...
  if (W % V != 0 || !aligned16(in) || !aligned16(out) || ibs % V != 0 || obs % V != 0) V = 1;   // ... or the element-wise general form
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * c * 2 * H * (W / V);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  ProfScope ps(st, K_COL2IM, 2.0 * B * dense_lo * dtype_size(dtype), 0.0);
#define PS_LAUNCH(T, VV, UN) hipLaunchKernelGGL((pixel_shuffle_kernel<T, VV, UN>), dim3(blocks), dim3(256), 0, st, (const T*)in, (T*)out, c, H, W, ibs, obs, total)
  if (dtype == MI_BF16) {
    if (V == 8) { if (unshuffle) PS_LAUNCH(bf16, 8, true); else PS_LAUNCH(bf16, 8, false); }
    else { if (unshuffle) PS_LAUNCH(bf16, 1, true); else PS_LAUNCH(bf16, 1, false); }
  } else {
    if (V == 4) { if (unshuffle) PS_LAUNCH(float, 4, true); else PS_LAUNCH(float, 4, false); }
    else { if (unshuffle) PS_LAUNCH(float, 1, true); else PS_LAUNCH(float, 1, false); }
  }
#undef PS_LAUNCH
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_copy_rows(const void* src, int64_t src_rs, void* dst, int64_t dst_rs, int64_t rows, int64_t L, int dtype,
                            void* stream) {
  MI_CHECK_ARG(src && dst && rows > 0 && L > 0, "copy_rows: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "copy_rows: bad dtype %d", dtype);
  int V = dtype == MI_BF16 ? 8 : 4;
  if (!src_rs) src_rs = L;
  if (!dst_rs) dst_rs = L;
  if (L % V != 0 || src_rs % V != 0 || dst_rs % V != 0 || !aligned16(src) || !aligned16(dst)) V = 1;   // element-wise general form
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = rows * (L / V);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  ProfScope ps(st, K_CAST, 2.0 * rows * L * dtype_size(dtype), 0.0);
  if (dtype == MI_BF16) {
    if (V == 8) hipLaunchKernelGGL((copy_rows_kernel<bf16, 8>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, src_rs, (bf16*)dst, dst_rs, L, total);
    else hipLaunchKernelGGL((copy_rows_kernel<bf16, 1>), dim3(blocks), dim3(256), 0, st, (const bf16*)src, src_rs, (bf16*)dst, dst_rs, L, total);
  } else {
    if (V == 4) hipLaunchKernelGGL((copy_rows_kernel<float, 4>), dim3(blocks), dim3(256), 0, st, (const float*)src, src_rs, (float*)dst, dst_rs, L, total);
    else hipLaunchKernelGGL((copy_rows_kernel<float, 1>), dim3(blocks), dim3(256), 0, st, (const float*)src, src_rs, (float*)dst, dst_rs, L, total);
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}
