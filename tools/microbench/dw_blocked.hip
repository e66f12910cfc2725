// Prototype (NOT part of the library): the 3x3 depthwise convolution on the channel-blocked layout of pw_blocked.hip
//     X, Y: [z][C/32][H*W][32] bf16, weights [C][9] fp32, zero padding.
// Lane map: a wave-wide 16-byte load covers 16 consecutive pixels x 32 channels (lane = 4*pixel... li = pixel, g = 8-channel group);
// a wave owns a 16-pixel-wide strip of one 32-channel block and walks down a band of rows with a 3-row window in registers.  The
// left/right neighbours come from two more (L1-resident) loads per row; per-channel weights live in VGPRs (72 per lane) because in
// this layout a lane's channels differ from its neighbours' - the price of the layout for the stencil kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -o dw_blocked dw_blocked.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef unsigned short bf16_t;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
static inline bf16_t f2bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t b) { unsigned u = (unsigned)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
__device__ inline unsigned pack2(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fff + ((ua >> 16) & 1); ub += 0x7fff + ((ub >> 16) & 1);
  return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ inline float lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ inline float hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

struct Row3 { u32x4 l, c, r; };     // the three horizontally shifted copies of one input row (8 channels per lane)

// left / right neighbours: the adjacent lane of the same 16-lane row (DPP row_shr:1 / row_shl:1); only the strip's two edge lanes need
// memory, and one sparse load serves both (lane li == 0 fetches x - 1, lane li == 15 fetches x + 1)
__device__ inline unsigned dpp_shr1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }
__device__ inline unsigned dpp_shl1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }
struct RowRaw { u32x4 c, h; };
__device__ inline RowRaw load_raw(const bf16_t* plane, int y, int x, int H, int W, int g, int li) {
  RowRaw o; o.c = o.h = (u32x4){0u, 0u, 0u, 0u};
  if (y >= 0 && y < H) {
    const bf16_t* p = plane + ((long)y * W + x) * 32 + 8 * g;
    o.c = *reinterpret_cast<const u32x4*>(p);
    if (li == 0 && x > 0) o.h = *reinterpret_cast<const u32x4*>(p - 32);
    if (li == 15 && x + 1 < W) o.h = *reinterpret_cast<const u32x4*>(p + 32);
  }
  return o;
}
__device__ inline Row3 finish_row(const RowRaw& q, int li) {
  Row3 o; o.c = q.c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned l = dpp_shr1(q.c[j]), r = dpp_shl1(q.c[j]);
    o.l[j] = li == 0 ? q.h[j] : l;
    o.r[j] = li == 15 ? q.h[j] : r;
  }
  return o;
}
__device__ inline Row3 load_row(const bf16_t* plane, int y, int x, int H, int W, int g, int li) {
  return finish_row(load_raw(plane, y, x, H, W, g, li), li);
}

__device__ inline void fma_row(float (&acc)[8], const Row3& r, const float (&w)[9][8], int ky) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    acc[2 * j] += lo(r.l[j]) * w[3 * ky][2 * j] + lo(r.c[j]) * w[3 * ky + 1][2 * j] + lo(r.r[j]) * w[3 * ky + 2][2 * j];
    acc[2 * j + 1] += hi(r.l[j]) * w[3 * ky][2 * j + 1] + hi(r.c[j]) * w[3 * ky + 1][2 * j + 1] + hi(r.r[j]) * w[3 * ky + 2][2 * j + 1];
  }
}

// grid: x = strips (W/16) * bands, y = channel blocks, z = images; block = MW waves, wave w -> strip index blockIdx.x * MW + w
template <int MW, int MODE>
__global__ __launch_bounds__(64 * MW) void dw_blocked_kernel(const bf16_t* __restrict__ X, const float* __restrict__ Wt, bf16_t* __restrict__ Y,
                                                             int C, int H, int W, int band) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int strips = W / 16;
  const int unit = blockIdx.x * MW + wv;
  const int strip = unit % strips, y0 = (unit / strips) * band;
  if (y0 >= H) return;
  const int cb = blockIdx.y, z = blockIdx.z;
  const long plane_off = ((long)z * (C / 32) + cb) * (long)H * W * 32;
  const bf16_t* xp = X + plane_off;
  bf16_t* yp = Y + plane_off;
  const int x = strip * 16 + li;
  float w[9][8];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k][c] = Wt[(long)(cb * 32 + 8 * g + c) * 9 + k];
  Row3 r0 = load_row(xp, y0 - 1, x, H, W, g, li), r1 = load_row(xp, y0, x, H, W, g, li);
  const int y1 = min(y0 + band, H);
  RowRaw q2 = load_raw(xp, y0 + 1, x, H, W, g, li), q3 = load_raw(xp, y0 + 2, x, H, W, g, li);
  for (int y = y0; y < y1; ++y) {
    RowRaw q4 = load_raw(xp, y + 3, x, H, W, g, li);   // two rows of raw loads in flight ahead of the window
    const Row3 r2 = finish_row(q2, li);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32x4 o;
    if (MODE == 0) {
      fma_row(acc, r0, w, 0); fma_row(acc, r1, w, 1); fma_row(acc, r2, w, 2);
      o = (u32x4){pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7])};
    } else {                                        // memory-pattern ceiling: same loads and stores, no arithmetic
      o = r0.c ^ r1.l ^ r2.r ^ r1.c;
    }
    *reinterpret_cast<u32x4*>(yp + ((long)y * W + x) * 32 + 8 * g) = o;
    r0 = r1; r1 = r2; q2 = q3; q3 = q4;
  }
}

// ---- v2: 4 channels per lane (8-byte loads, lane = 8 pixels x 8 channel groups = 512 contiguous bytes per wave load), scatter form:
// an arriving input row is unpacked once and added into the three output rows it touches (three fp32 accumulator rows instead of a
// three-row input window), packed fp32 FMAs, 36 weight registers per lane.  SW strips of 8 pixels per wave side by side.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
struct Raw3 { u32x2 l, c, r; };
// always-executed loads from clamped addresses (no divergent branch around a load, so the compiler can count them in s_waitcnt);
// out-of-image taps are zeroed when the row is consumed
__device__ inline Raw3 load_raw3(const bf16_t* plane, int y, int x, int H, int W, int cg) {
  Raw3 o;
  const int yc = min(max(y, 0), H - 1);
  const bf16_t* p = plane + ((long)yc * W + x) * 32 + 4 * cg;
  o.c = *reinterpret_cast<const u32x2*>(p);
  o.l = *reinterpret_cast<const u32x2*>(x > 0 ? p - 32 : p);
  o.r = *reinterpret_cast<const u32x2*>(x + 1 < W ? p + 32 : p);
  return o;
}
__device__ inline Raw3 mask_raw3(Raw3 o, int y, int x, int H, int W) {
  const bool row = y >= 0 && y < H;
  const unsigned mc = row ? ~0u : 0u, ml = (row && x > 0) ? ~0u : 0u, mr = (row && x + 1 < W) ? ~0u : 0u;
  o.c[0] &= mc; o.c[1] &= mc; o.l[0] &= ml; o.l[1] &= ml; o.r[0] &= mr; o.r[1] &= mr;
  return o;
}
__device__ inline f32x2 up(unsigned u) { return (f32x2){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }

template <int MW, int SW>
__global__ __launch_bounds__(64 * MW) void dw_blocked4_kernel(const bf16_t* __restrict__ X, const float* __restrict__ Wt, bf16_t* __restrict__ Y,
                                                              int C, int H, int W, int band) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int px = lane >> 3, cg = lane & 7;
  const int strips = W / (8 * SW);
  const int unit = blockIdx.x * MW + wv;
  const int strip = unit % strips, y0 = (unit / strips) * band;
  if (y0 >= H) return;
  const int cb = blockIdx.y, z = blockIdx.z;
  const long plane_off = ((long)z * (C / 32) + cb) * (long)H * W * 32;
  const bf16_t* xp = X + plane_off;
  bf16_t* yp = Y + plane_off;
  const int x0 = strip * 8 * SW + px;
  f32x2 w[9][2];                      // taps x channel pairs
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      w[k][j] = (f32x2){Wt[(long)(cb * 32 + 4 * cg + 2 * j) * 9 + k], Wt[(long)(cb * 32 + 4 * cg + 2 * j + 1) * 9 + k]};
  const int y1 = min(y0 + band, H);
  // accumulators of output rows r-1 (a0), r (a1), r+1 (a2) while input row r is being scattered
  f32x2 a0[SW][2], a1[SW][2], a2[SW][2];
#pragma unroll
  for (int s = 0; s < SW; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) a0[s][j] = a1[s][j] = a2[s][j] = (f32x2){0.f, 0.f};
  constexpr int D = 3;                 // input rows in flight ahead of the one being scattered
  Raw3 q[D + 1][SW];
#pragma unroll
  for (int d = 0; d <= D; ++d)
#pragma unroll
    for (int s = 0; s < SW; ++s) q[d][s] = load_raw3(xp, y0 - 1 + d, x0 + 8 * s, H, W, cg);
  // the ring is indexed statically (the row loop is unrolled D+1 times): a register that a load is still filling is never copied
  for (int rb = y0 - 1; rb <= y1; rb += D + 1) {
#pragma unroll
    for (int u = 0; u <= D; ++u) {
      const int r = rb + u;
      if (r > y1) break;
#pragma unroll
      for (int s = 0; s < SW; ++s) {
        const Raw3 cur = mask_raw3(q[u][s], r, x0 + 8 * s, H, W);
        q[u][s] = load_raw3(xp, r + D + 1, x0 + 8 * s, H, W, cg);    // (rows past the band are loaded clamped and never used)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x2 l = up(cur.l[j]), c = up(cur.c[j]), rr = up(cur.r[j]);
          // input row r is tap row ky = 2 of output r-1, ky = 1 of output r, ky = 0 of output r+1
          a0[s][j] += l * w[6][j] + c * w[7][j] + rr * w[8][j];
          a1[s][j] += l * w[3][j] + c * w[4][j] + rr * w[5][j];
          a2[s][j] += l * w[0][j] + c * w[1][j] + rr * w[2][j];
        }
        if (r - 1 >= y0) {
          u32x2 o = {pack2(a0[s][0][0], a0[s][0][1]), pack2(a0[s][1][0], a0[s][1][1])};
          *reinterpret_cast<u32x2*>(yp + ((long)(r - 1) * W + x0 + 8 * s) * 32 + 4 * cg) = o;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) { a0[s][j] = a1[s][j]; a1[s][j] = a2[s][j]; a2[s][j] = (f32x2){0.f, 0.f}; }
      }
    }
  }
}

int main() {
  const int H = 256, Wd = 256, Z = 32;
  const int band = getenv("BAND") ? atoi(getenv("BAND")) : 32;
  for (int C : {288, 96, 512}) {
    const size_t e = (size_t)Z * C * H * Wd;
    bf16_t *X, *Y; float* Wt;
    CK(hipMalloc(&X, e * 2)); CK(hipMalloc(&Y, e * 2)); CK(hipMalloc(&Wt, (size_t)C * 9 * 4));
    std::vector<bf16_t> hx((size_t)(C / 32) * H * Wd * 32);
    std::vector<float> hw((size_t)C * 9);
    srand(2);
    for (auto& v : hw) v = (rand() % 200 - 100) / 100.f;
    for (auto& v : hx) v = f2bf((rand() % 200 - 100) / 100.f);
    CK(hipMemset(X, 0x3c, e * 2)); CK(hipDeviceSynchronize());
    for (size_t off = 0; off < hx.size(); off += (1u << 22))                  // image 0 carries the checked data
      CK(hipMemcpy(X + off, hx.data() + off, std::min<size_t>(1u << 22, hx.size() - off) * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    constexpr int MW = 4;
    const int units = (Wd / 16) * ((H + band - 1) / band);
    dim3 grid((units + MW - 1) / MW, C / 32, Z);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool copy = getenv("COPY") != nullptr;
    const int v2 = getenv("V2") ? atoi(getenv("V2")) : 0;     // strips of 8 pixels per wave
    dim3 grid2(((Wd / (8 * std::max(v2, 1))) * ((H + band - 1) / band) + MW - 1) / MW, C / 32, Z);
    auto launch = [&]() {
      if (v2 == 1) hipLaunchKernelGGL((dw_blocked4_kernel<MW, 1>), grid2, dim3(64 * MW), 0, 0, X, Wt, Y, C, H, Wd, band);
      else if (v2 == 2) hipLaunchKernelGGL((dw_blocked4_kernel<MW, 2>), grid2, dim3(64 * MW), 0, 0, X, Wt, Y, C, H, Wd, band);
      else if (v2 == 4) hipLaunchKernelGGL((dw_blocked4_kernel<MW, 4>), grid2, dim3(64 * MW), 0, 0, X, Wt, Y, C, H, Wd, band);
      else if (copy) hipLaunchKernelGGL((dw_blocked_kernel<MW, 1>), grid, dim3(64 * MW), 0, 0, X, Wt, Y, C, H, Wd, band);
      else hipLaunchKernelGGL((dw_blocked_kernel<MW, 0>), grid, dim3(64 * MW), 0, 0, X, Wt, Y, C, H, Wd, band);
    };
    for (int i = 0; i < 2; ++i) launch();
    CK(hipGetLastError());
    CK(hipEventRecord(e0)); const int it = 5;
    for (int i = 0; i < it; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / it * 1e3, gb = 2.0 * e * 2 / 1e9;
    std::vector<bf16_t> hy(hx.size());
    for (size_t off = 0; off < hy.size(); off += (1u << 22))
      CK(hipMemcpy(hy.data() + off, Y + off, std::min<size_t>(1u << 22, hy.size() - off) * 2, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int c = 0; c < C; c += 5) for (int y = 0; y < H; y += 37) for (int x = 0; x < Wd; x += 15) {
      double ref = 0;
      for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= Wd) continue;
        ref += (double)hw[(size_t)c * 9 + ky * 3 + kx] * bf2f(hx[((size_t)(c / 32) * H * Wd + (size_t)yy * Wd + xx) * 32 + (c & 31)]);
      }
      const double got = bf2f(hy[((size_t)(c / 32) * H * Wd + (size_t)y * Wd + x) * 32 + (c & 31)]);
      maxerr = fmax(maxerr, fabs(got - ref) / fmax(1.0, fabs(ref)));
    }
    printf("blocked dw3x3 v2=%d C=%4d %dx%d x%d band %d: %8.1f us  %6.0f GB/s   max rel err %.3g\n", v2, C, H, Wd, Z, band, us, gb / us * 1e6, maxerr);
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(Wt));
  }
  return 0;
}
