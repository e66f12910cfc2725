// Dense 3x3 convolution (stride 1, zero padding 1) as an IMPLICIT GEMM on the MFMA units - the U-Net glue of Restormer /
// MoCE-IR / AdaIR: OverlapPatchEmbed (Restormer.py:156-165), Downsample / Upsample bodies (:171-189), the output conv (:243,281).
//   y[b][m][p] = sum_{k, tap} W[m][k][tap] . x[b][k][p + d(tap)]  (+ bias[m]) (+ residual[b][m][p])        d(tap) = (ky-1, kx-1)
// Until round 3 this ran as im2col3x3 -> 1x1 GEMM (or GEMM -> col2im3x3): the 9-plane expansion was written to and read from
// HBM.  Here a workgroup owns 16.MT output channels x 256 pixels (TR rows x TWP columns of one image) and walks the input
// channels in chunks of 16: the chunk's rows (TR + 2, with the halo) are staged in LDS THREE times - as they are, shifted one
// pixel left and one pixel right (built in registers from the 16-byte global loads, neighbours by DPP) - so that every tap is an
// ALIGNED window of one of the copies and the B operands come from ds_read_b64_tr_b16 like in the 1x1 kernels.  One
// v_mfma_f32_16x16x32_bf16 contracts 16 channels of TWO taps (k = 0..15 tap 2i, k = 16..31 tap 2i+1; the ninth tap rides with a
// zero half), the packed weights are stored fragment-major (one ds_read_b128 per lane per fragment, no address math).
// The same kernel is the data gradient: dx = conv(dy, W^T flipped) - only the weight pack differs (c3_pack_kernel, transpose_flip).
// The weight gradient (c3_wgrad_kernel) contracts over the PIXELS instead: dW[m][k][tap] = sum_p dy[m][p] . x[k][p + d(tap)],
// A = dy rows, B = the same three shifted copies read along the pixel axis (plain 8-byte reads), per-workgroup partial sums.
#include "common.h"
#include "fused_common.h"
#include "internal.h"

namespace mi {
using namespace fz;

constexpr int C3_XS = 400;          // LDS plane stride (elements) of a staged channel: >= (TR + 2) * TWP and = 16 mod 128 (conflict-free tr reads)
constexpr int C3_KS = 5;            // k-steps per 16-channel chunk: tap pairs (0,1) (2,3) (4,5) (6,7) (8,-)

struct C3Args {
  const bf16* x; const bf16* wp; const float* bias; const bf16* res; bf16* y;
  int64_t xbs, ybs, rbs;            // batch strides (elements); channel planes are dense H*W
  int B, M, K, H, W, nchunk, tiles_x, tiles_y;
};

// Packed weights: [co tile][chunk][k-step][m-tile][lane][8] bf16.  Lane (li = lane & 15, g = lane >> 4) of fragment (ks, mt)
// holds A[m = 16 mt + li][k]: elements 0..3 = channels c0 + 4g .. 4g+3 of tap 2 ks, elements 4..7 = the same channels of tap
// 2 ks + 1 (zero past tap 8 / past K / past M).  Source element (m, k, tap) = w[m * s_m + k * s_k + (flip ? 8 - tap : tap)].
struct C3PackArgs { const float* w; bf16* wp; int M, K, MT, ncot, nchunk, flip; int64_t s_m, s_k; };
__global__ __launch_bounds__(256) void c3_pack_kernel(C3PackArgs a) {
  const int64_t total = (int64_t)a.ncot * a.nchunk * C3_KS * a.MT * 512;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t r = e;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int mt = (int)(r % a.MT); r /= a.MT;
    const int ks = (int)(r % C3_KS); r /= C3_KS;
    const int ch = (int)(r % a.nchunk);
    const int cot = (int)(r / a.nchunk);
    const int li = lane & 15, g = lane >> 4;
    const int m = (cot * a.MT + mt) * 16 + li, k = ch * 16 + 4 * g + (j & 3), tap = 2 * ks + (j >> 2);
    float v = 0.f;
    if (m < a.M && k < a.K && tap < 9) v = a.w[(int64_t)m * a.s_m + (int64_t)k * a.s_k + (a.flip ? 8 - tap : tap)];
    a.wp[e] = (bf16)v;
  }
}

template <int MT, int TWP>
__global__ __launch_bounds__(256) void c3_kernel(C3Args a) {
  constexpr int TR = 256 / TWP, SR = TR + 2, VPR = TWP / 8;
  constexpr int NV = 16 * SR * VPR;                     // 16-byte vectors of one staged chunk
  constexpr int NVT = (NV + 255) / 256;
  constexpr int WV = C3_KS * MT * 64;                   // 16-byte vectors of one weight chunk
  constexpr int WVT = (WV + 255) / 256;
  constexpr int X_BYTES = 3 * 16 * C3_XS * 2;
  constexpr int W_BYTES = WV * 16;
  constexpr int ES = 260;                               // epilogue row stride (floats)
  constexpr int EH = MT > 2 ? 2 : MT;                   // m-tiles per epilogue pass
  static_assert(SR * TWP <= C3_XS && C3_XS % 128 == 16, "plane stride");
  static_assert(EH * 16 * ES * 4 <= X_BYTES + W_BYTES, "epilogue tile lives in the operand region");
  __shared__ __attribute__((aligned(16))) unsigned char lds[X_BYTES + W_BYTES];
  bf16* const X = reinterpret_cast<bf16*>(lds);         // [3 copies][16 channels][C3_XS]
  bf16* const Wl = reinterpret_cast<bf16*>(lds + X_BYTES);   // [KS][MT][64 lanes][8]
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
  const int cot = blockIdx.y, b = blockIdx.z;
  const int x0 = tx * TWP, y0 = ty * TR;
  const int64_t HW = (int64_t)a.H * a.W;
  const bf16* const xb = a.x + (int64_t)b * a.xbs;
  const bf16* const wb = a.wp + (int64_t)cot * a.nchunk * (WV * 8);

  u32x4 xr[NVT], wr[WVT];
  unsigned int xe[NVT];                                 // low half: the element left of the vector, high half: the one right of it
  auto issue = [&](int ch) {
#pragma unroll
    for (int n = 0; n < NVT; ++n) {
      const int v = t + 256 * n;
      const int c = v / (SR * VPR), rem = v - c * (SR * VPR), r = rem / VPR, u = rem - r * VPR;
      const int Y = y0 - 1 + r, Xc = x0 + 8 * u, k = ch * 16 + c;
      xr[n] = (u32x4){0u, 0u, 0u, 0u};
      xe[n] = 0u;
      if (v < NV && k < a.K && Y >= 0 && Y < a.H && Xc < a.W) {
        const bf16* src = xb + (int64_t)k * HW + (int64_t)Y * a.W + Xc;
        xr[n] = *reinterpret_cast<const u32x4*>(src);
        // tile-edge vectors fetch their outer neighbour themselves (inner neighbours arrive by DPP at store time)
        if (u == 0 && Xc > 0) xe[n] = reinterpret_cast<const u16*>(src)[-1];
        if (u == VPR - 1 && Xc + 8 < a.W) xe[n] |= (unsigned int)reinterpret_cast<const u16*>(src)[8] << 16;
      }
    }
    const u32x4* ws = reinterpret_cast<const u32x4*>(wb + (int64_t)ch * (WV * 8));
#pragma unroll
    for (int n = 0; n < WVT; ++n) {
      const int v = t + 256 * n;
      if (v < WV) wr[n] = ws[v];
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int n = 0; n < NVT; ++n) {
      const int v = t + 256 * n;
      const int c = v / (SR * VPR), rem = v - c * (SR * VPR), r = rem / VPR, u = rem - r * VPR;
      const u32x4 w = xr[n];
      // neighbours inside the row come from the adjacent lanes (consecutive lanes hold consecutive vectors of a row)
      // (the halves are isolated BEFORE the cross-lane move and its result is opaque to the optimiser: with `dpp(w[3]) >> 16` the
      //  DPP combine folded the move into the shift and the left neighbours came back wrong)
      unsigned int pl = __builtin_bit_cast(unsigned int, from_prev_lane(__builtin_bit_cast(float, w[3] >> 16)));
      unsigned int nx = __builtin_bit_cast(unsigned int, from_next_lane(__builtin_bit_cast(float, w[0] & 0xffffu)));
      asm volatile("" : "+v"(pl), "+v"(nx));
      const unsigned int le = (u == 0) ? (xe[n] & 0xffffu) : pl;
      const unsigned int re = (u == VPR - 1) ? (xe[n] >> 16) : nx;
      u32x4 l, rr;                                      // l[j] = x[j - 1] (tap kx = 0), rr[j] = x[j + 1] (tap kx = 2)
      l[0] = (w[0] << 16) | le;           l[1] = (w[1] << 16) | (w[0] >> 16);
      l[2] = (w[2] << 16) | (w[1] >> 16); l[3] = (w[3] << 16) | (w[2] >> 16);
      rr[0] = (w[0] >> 16) | (w[1] << 16); rr[1] = (w[1] >> 16) | (w[2] << 16);
      rr[2] = (w[2] >> 16) | (w[3] << 16); rr[3] = (w[3] >> 16) | (re << 16);
      if (v < NV) {
        bf16* d = X + c * C3_XS + r * TWP + 8 * u;
        *reinterpret_cast<u32x4*>(d) = l;
        *reinterpret_cast<u32x4*>(d + 16 * C3_XS) = w;
        *reinterpret_cast<u32x4*>(d + 32 * C3_XS) = rr;
      }
    }
#pragma unroll
    for (int n = 0; n < WVT; ++n) {
      const int v = t + 256 * n;
      if (v < WV) reinterpret_cast<u32x4*>(Wl)[v] = wr[n];
    }
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[mt][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0);
  for (int ch = 0; ch < a.nchunk; ++ch) {
    __syncthreads();                                    // the previous chunk's operands are no longer read
    stash();
    __syncthreads();
    if (ch + 1 < a.nchunk) issue(ch + 1);
    // the wave's four 16-pixel n-tiles: pixels 64 wv + 16 n + li of the tile
    int boff[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int p0 = 64 * wv + 16 * n;
      boff[n] = (4 * g + qq) * C3_XS + (p0 / TWP) * TWP + (p0 % TWP) + 4 * pp;
    }
#pragma unroll
    for (int ks = 0; ks < C3_KS; ++ks) {
      const int t0 = 2 * ks, t1 = 2 * ks + 1;
      const int o0 = (t0 % 3) * 16 * C3_XS + (t0 / 3) * TWP;          // copy kx, row offset ky
      const int o1 = (t1 % 3) * 16 * C3_XS + (t1 / 3) * TWP;
      s16x8 af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const s16x8*>(&Wl[((ks * MT + mt) * 64 + lane) * 8]);
      s16x4 lo[4], hi[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        lo[n] = tr_b16(&X[boff[n] + o0]);
        hi[n] = t1 < 9 ? tr_b16(&X[boff[n] + o1]) : (s16x4){0, 0, 0, 0};
      }
      if (t1 < 9) lds_wait(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
      else lds_wait(lo[0], lo[1], lo[2], lo[3]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const s16x8 bf = cat8(lo[n], hi[n]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][n] = mfma32(af[mt], bf, acc[mt][n]);
      }
    }
  }

  // epilogue: accumulators -> fp32 tile in LDS -> (+ bias, + residual) -> 16-byte stores along the rows
  float* const E = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int h = 0; h < MT / EH; ++h) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EH; ++e) {
      const int mt = h * EH + e;
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) E[(e * 16 + 4 * g + r) * ES + 64 * wv + 16 * n + li] = acc[mt][n][r];
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < (EH * 16 * 32 + 255) / 256; ++n) {
      const int v = t + 256 * n;                        // vector: row (channel) v / 32, pixels 8 (v % 32) ..
      const int row = v / 32, pv = v % 32;
      const int m = (cot * MT + h * EH) * 16 + row;
      const int p0 = 8 * pv, Y = y0 + p0 / TWP, Xc = x0 + p0 % TWP;
      if (row < EH * 16 && m < a.M && Y < a.H && Xc < a.W) {
        float o[8];
        Vec<float, 4>::ld(&E[row * ES + p0], o);
        Vec<float, 4>::ld(&E[row * ES + p0 + 4], o + 4);
        const float bv = a.bias ? a.bias[m] : 0.f;
        const int64_t off = (int64_t)m * HW + (int64_t)Y * a.W + Xc;
        if (a.res) {
          const u32x4 rv = *reinterpret_cast<const u32x4*>(a.res + (int64_t)b * a.rbs + off);
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[2 * k] += bf_lo(rv[k]); o[2 * k + 1] += bf_hi(rv[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] += bv;
        Vec<bf16, 8>::st(a.y + (int64_t)b * a.ybs + off, o);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[m][k][tap] = sum_{b,p} dy[b][m][p] . x[b][k][p + d(tap)]: the contraction runs over the PIXELS.  A workgroup owns 64 rows m
// x one 16-channel chunk of x x 9 taps and walks its share of the 256-pixel tiles; per tile it stages dy [64][256] and the
// chunk's three shifted copies (as the forward kernel), then per 32 pixels: A = 8 pixels of a dy row (ds_read_b128), B = 8
// pixels of a channel of the tap's copy (ds_read_b128, aligned because the shift is baked into the copy).  Wave w holds the
// accumulators of m-tiles {2 (w & 1), +1} x taps {0..4} (w < 2) or {5..8} (w >= 2); partial sums go to part[split][M][K][9].
constexpr int C3_XSW = 392;         // plane stride of the staged channels here: = 8 mod 128 (conflict-free 16-byte reads along the row)
constexpr int C3_DS = 264;          // dy row stride (elements)
// Either operand can play the row role: rows = dy channels, shifted columns = x channels (out[m][k][tap]), or - when that wastes
// less of the 64 x 16 tile - rows = x channels, shifted columns = dy channels with the taps negated
// (sum_p x[k][p] dy[m][p - d] is the same number): M = rows, K = shifted columns here, the output index is
// row * s_r + col * s_c + (flip ? 8 - tap : tap) inside a partial of `cols` floats.
struct C3WArgs {
  const bf16* dy; const bf16* x; float* part;
  int64_t dybs, xbs, s_r, s_c, cols;
  int B, M, K, H, W, tiles_x, tiles_y, S, flip;
};

template <int TWP>
__global__ __launch_bounds__(256) void c3_wgrad_kernel(C3WArgs a) {
  constexpr int TR = 256 / TWP, SR = TR + 2, VPR = TWP / 8;
  constexpr int NV = 16 * SR * VPR, NVT = (NV + 255) / 256;
  constexpr int DV = 64 * 32, DVT = DV / 256;           // dy tile: 64 rows x 32 vectors
  constexpr int X_BYTES = 3 * 16 * C3_XSW * 2;
  static_assert(SR * TWP <= C3_XSW && C3_XSW % 128 == 8, "plane stride");
  extern __shared__ __attribute__((aligned(16))) unsigned char c3w_lds[];
  bf16* const X = reinterpret_cast<bf16*>(c3w_lds);                  // [3][16][C3_XSW]
  bf16* const DY = reinterpret_cast<bf16*>(c3w_lds + X_BYTES);       // [64][C3_DS]
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int kc = blockIdx.x, mb = blockIdx.y, sp = blockIdx.z;
  const int mp = wv & 1, ts = wv >> 1;
  const int64_t HW = (int64_t)a.H * a.W;
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int total = a.B * tiles_img;

  u32x4 xr[NVT], dr[DVT];
  unsigned int xe[NVT];
  auto issue = [&](int tile) {
    const int b = tile / tiles_img, ti = tile - b * tiles_img;
    const int x0 = (ti % a.tiles_x) * TWP, y0 = (ti / a.tiles_x) * TR;
    const bf16* const xb = a.x + (int64_t)b * a.xbs;
    const bf16* const db = a.dy + (int64_t)b * a.dybs;
#pragma unroll
    for (int n = 0; n < NVT; ++n) {
      const int v = t + 256 * n;
      const int c = v / (SR * VPR), rem = v - c * (SR * VPR), r = rem / VPR, u = rem - r * VPR;
      const int Y = y0 - 1 + r, Xc = x0 + 8 * u, k = kc * 16 + c;
      xr[n] = (u32x4){0u, 0u, 0u, 0u};
      xe[n] = 0u;
      if (v < NV && k < a.K && Y >= 0 && Y < a.H && Xc < a.W) {
        const bf16* src = xb + (int64_t)k * HW + (int64_t)Y * a.W + Xc;
        xr[n] = *reinterpret_cast<const u32x4*>(src);
        if (u == 0 && Xc > 0) xe[n] = reinterpret_cast<const u16*>(src)[-1];
        if (u == VPR - 1 && Xc + 8 < a.W) xe[n] |= (unsigned int)reinterpret_cast<const u16*>(src)[8] << 16;
      }
    }
#pragma unroll
    for (int n = 0; n < DVT; ++n) {
      const int v = t + 256 * n;
      const int row = v >> 5, pv = v & 31, p0 = 8 * pv;
      const int m = mb * 64 + row, Y = y0 + p0 / TWP, Xc = x0 + p0 % TWP;
      dr[n] = (u32x4){0u, 0u, 0u, 0u};
      if (m < a.M && Y < a.H && Xc < a.W) dr[n] = *reinterpret_cast<const u32x4*>(db + (int64_t)m * HW + (int64_t)Y * a.W + Xc);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int n = 0; n < NVT; ++n) {
      const int v = t + 256 * n;
      const int c = v / (SR * VPR), rem = v - c * (SR * VPR), r = rem / VPR, u = rem - r * VPR;
      const u32x4 w = xr[n];
      unsigned int pl = __builtin_bit_cast(unsigned int, from_prev_lane(__builtin_bit_cast(float, w[3] >> 16)));
      unsigned int nx = __builtin_bit_cast(unsigned int, from_next_lane(__builtin_bit_cast(float, w[0] & 0xffffu)));
      asm volatile("" : "+v"(pl), "+v"(nx));             // (see c3_kernel)
      const unsigned int le = (u == 0) ? (xe[n] & 0xffffu) : pl;
      const unsigned int re = (u == VPR - 1) ? (xe[n] >> 16) : nx;
      u32x4 l, rr;
      l[0] = (w[0] << 16) | le;           l[1] = (w[1] << 16) | (w[0] >> 16);
      l[2] = (w[2] << 16) | (w[1] >> 16); l[3] = (w[3] << 16) | (w[2] >> 16);
      rr[0] = (w[0] >> 16) | (w[1] << 16); rr[1] = (w[1] >> 16) | (w[2] << 16);
      rr[2] = (w[2] >> 16) | (w[3] << 16); rr[3] = (w[3] >> 16) | (re << 16);
      if (v < NV) {
        bf16* d = X + c * C3_XSW + r * TWP + 8 * u;
        *reinterpret_cast<u32x4*>(d) = l;
        *reinterpret_cast<u32x4*>(d + 16 * C3_XSW) = w;
        *reinterpret_cast<u32x4*>(d + 32 * C3_XSW) = rr;
      }
    }
#pragma unroll
    for (int n = 0; n < DVT; ++n) {
      const int v = t + 256 * n;
      *reinterpret_cast<u32x4*>(&DY[(v >> 5) * C3_DS + 8 * (v & 31)]) = dr[n];
    }
  };

  f32x4 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int tile = sp;
  if (tile < total) issue(tile);
  for (; tile < total; tile += a.S) {
    __syncthreads();
    stash();
    __syncthreads();
    if (tile + a.S < total) issue(tile + a.S);
    const bf16* const ar = &DY[(32 * mp + li) * C3_DS + 8 * g];
    const bf16* const br = &X[li * C3_XSW + 8 * g];
#pragma unroll 2
    for (int ks = 0; ks < 8; ++ks) {
      const s16x8 a0 = *reinterpret_cast<const s16x8*>(ar + 32 * ks);
      const s16x8 a1 = *reinterpret_cast<const s16x8*>(ar + 16 * C3_DS + 32 * ks);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int tap = 5 * ts + j;
        if (tap < 9) {
          const s16x8 bv = *reinterpret_cast<const s16x8*>(br + (tap % 3) * 16 * C3_XSW + (tap / 3) * TWP + 32 * ks);
          acc[0][j] = mfma32(a0, bv, acc[0][j]);
          acc[1][j] = mfma32(a1, bv, acc[1][j]);
        }
      }
    }
  }
  // partial sums of this workgroup: part[sp][m][k][tap]
  float* const po = a.part + (int64_t)sp * a.cols;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tap = 5 * ts + j, k = kc * 16 + li;
      if (tap < 9 && k < a.K) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mb * 64 + 32 * mp + 16 * i + 4 * g + r;
          if (m < a.M) po[(int64_t)m * a.s_r + (int64_t)k * a.s_c + (a.flip ? 8 - tap : tap)] = acc[i][j][r];
        }
      }
    }
}

static int c3w_splits(int B, int M, int K, int H, int W, int TWP) {
  const int64_t tiles = (int64_t)B * cdiv(W, TWP) * cdiv(H, 256 / TWP);
  const int64_t groups = (int64_t)cdiv(K, 16) * cdiv(M, 64);
  int64_t S = cdiv(1024, groups);                       // ~4 workgroups per CU in flight over the whole grid
  if (S > tiles / 2) S = tiles / 2;
  if (S < 1) S = 1;
  if (S > 4096) S = 4096;
  return (int)S;
}

static int c3_mt(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : 4); }
static int c3_twp(int W) { return W > 32 ? 64 : (W > 16 ? 32 : 16); }

}  // namespace mi

using namespace mi;

extern "C" int mi_conv3x3_ok(int H, int W, int dtype) {
  return (dtype == MI_BF16 && H >= 1 && W >= 8 && W % 8 == 0) ? 1 : 0;
}

extern "C" size_t mi_conv3x3_pack_bytes(int M, int K) {
  if (M < 1 || K < 1) return 0;
  const int MT = c3_mt(M);
  return (size_t)cdiv(M, 16 * MT) * cdiv(K, 16) * C3_KS * MT * 512 * 2;
}

extern "C" int mi_conv3x3_pack(const float* w, int M, int K, int transpose_flip, void* pack, void* stream) {
  MI_CHECK_ARG(w && pack && M >= 1 && K >= 1, "conv3x3_pack: null pointer / bad shape");
  MI_CHECK_ARG(aligned16(pack), "conv3x3_pack: pack buffer must be 16-byte aligned");
  C3PackArgs a;
  a.w = w; a.wp = (bf16*)pack; a.M = M; a.K = K; a.MT = c3_mt(M); a.ncot = cdiv(M, 16 * a.MT); a.nchunk = cdiv(K, 16);
  a.flip = transpose_flip ? 1 : 0;
  // forward: w is [M][K][9]; data gradient: the conv's own weight is [K][M][9] and this op's output channels are its inputs
  a.s_m = transpose_flip ? 9 : (int64_t)K * 9;
  a.s_k = transpose_flip ? (int64_t)M * 9 : 9;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)a.ncot * a.nchunk * C3_KS * a.MT * 512;
  ProfScope ps(st, K_FUSED_PACK, (double)total * 2 + (double)M * K * 36, 0.0);
  hipLaunchKernelGGL(c3_pack_kernel, dim3((unsigned)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048)), dim3(256), 0, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_conv3x3_fwd(const void* pack, const void* x, int64_t x_bs, const float* bias, const void* residual, int64_t r_bs,
                              void* y, int64_t y_bs, int B, int M, int K, int H, int W, void* stream) {
  MI_CHECK_ARG(pack && x && y && B >= 1 && M >= 1 && K >= 1, "conv3x3_fwd: null pointer / bad shape");
  MI_CHECK_ARG(mi_conv3x3_ok(H, W, MI_BF16), "conv3x3_fwd: plane not covered (bf16, W %% 8 == 0)");
  MI_CHECK_ARG(aligned16(pack) && aligned16(x) && aligned16(y) && (!residual || aligned16(residual)),
               "conv3x3_fwd: pointers must be 16-byte aligned");
  const int64_t HW = (int64_t)H * W;
  C3Args a;
  a.x = (const bf16*)x; a.wp = (const bf16*)pack; a.bias = bias; a.res = (const bf16*)residual; a.y = (bf16*)y;
  a.xbs = x_bs ? x_bs : (int64_t)K * HW; a.ybs = y_bs ? y_bs : (int64_t)M * HW; a.rbs = r_bs ? r_bs : (int64_t)M * HW;
  MI_CHECK_ARG(a.xbs % 8 == 0 && a.ybs % 8 == 0 && a.rbs % 8 == 0, "conv3x3_fwd: batch strides must keep 16-byte alignment");
  a.B = B; a.M = M; a.K = K; a.H = H; a.W = W; a.nchunk = cdiv(K, 16);
  const int MT = c3_mt(M), TWP = c3_twp(W), TR = 256 / TWP;
  a.tiles_x = cdiv(W, TWP); a.tiles_y = cdiv(H, TR);
  const int64_t tiles = (int64_t)a.tiles_x * a.tiles_y;
  MI_CHECK_ARG(tiles < (1ll << 31) && B <= 65535 && cdiv(M, 16 * MT) <= 65535, "conv3x3_fwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  const double N = (double)HW * B;
  ProfScope ps(st, K_CONV3X3, (double)(K + M + (residual ? M : 0)) * N * 2.0, 2.0 * 9.0 * M * K * N);
  const dim3 grid((unsigned)tiles, (unsigned)cdiv(M, 16 * MT), (unsigned)B);
#define C3_CASE(MT_, TWP_) \
  if (MT == MT_ && TWP == TWP_) { hipLaunchKernelGGL((c3_kernel<MT_, TWP_>), grid, dim3(256), 0, st, a); MI_LAUNCH_CHECK(); return MI_OK; }
  C3_CASE(1, 64) C3_CASE(2, 64) C3_CASE(4, 64)
  C3_CASE(1, 32) C3_CASE(2, 32) C3_CASE(4, 32)
  C3_CASE(1, 16) C3_CASE(2, 16) C3_CASE(4, 16)
#undef C3_CASE
  set_error("conv3x3_fwd: no kernel for MT=%d TWP=%d", MT, TWP);
  return MI_ERR_ARG;
}

// which operand takes the row role: the one that wastes less of the 64-row x 16-column tile
static bool c3w_swap(int M, int K) {
  const int64_t normal = (int64_t)cdiv(M, 64) * 64 * cdiv(K, 16) * 16, swapped = (int64_t)cdiv(K, 64) * 64 * cdiv(M, 16) * 16;
  return swapped < normal;
}

extern "C" size_t mi_conv3x3_wgrad_workspace(int B, int M, int K, int H, int W) {
  if (B < 1 || M < 1 || K < 1 || !mi_conv3x3_ok(H, W, MI_BF16)) return 0;
  const bool sw = c3w_swap(M, K);
  return (size_t)c3w_splits(B, sw ? K : M, sw ? M : K, H, W, c3_twp(W)) * M * K * 9 * sizeof(float);
}

extern "C" int mi_conv3x3_wgrad(const void* dy, int64_t dy_bs, const void* x, int64_t x_bs, float* dw, int accumulate, int B, int M,
                                int K, int H, int W, void* ws, void* stream) {
  MI_CHECK_ARG(dy && x && dw && B >= 1 && M >= 1 && K >= 1, "conv3x3_wgrad: null pointer / bad shape");
  MI_CHECK_ARG(mi_conv3x3_ok(H, W, MI_BF16), "conv3x3_wgrad: plane not covered (bf16, W %% 8 == 0)");
  MI_CHECK_ARG(aligned16(dy) && aligned16(x), "conv3x3_wgrad: pointers must be 16-byte aligned");
  const int64_t HW = (int64_t)H * W;
  const int TWP = c3_twp(W), TR = 256 / TWP;
  if (!dy_bs) dy_bs = (int64_t)M * HW;
  if (!x_bs) x_bs = (int64_t)K * HW;
  MI_CHECK_ARG(dy_bs % 8 == 0 && x_bs % 8 == 0, "conv3x3_wgrad: batch strides must keep 16-byte alignment");
  const bool sw = c3w_swap(M, K);
  C3WArgs a;
  if (!sw) { a.dy = (const bf16*)dy; a.dybs = dy_bs; a.M = M; a.x = (const bf16*)x; a.xbs = x_bs; a.K = K; a.s_r = (int64_t)K * 9; a.s_c = 9; }
  else     { a.dy = (const bf16*)x; a.dybs = x_bs; a.M = K; a.x = (const bf16*)dy; a.xbs = dy_bs; a.K = M; a.s_r = 9; a.s_c = (int64_t)K * 9; }
  a.flip = sw ? 1 : 0;
  a.B = B; a.H = H; a.W = W; a.tiles_x = cdiv(W, TWP); a.tiles_y = cdiv(H, TR);
  a.S = c3w_splits(B, a.M, a.K, H, W, TWP);
  const size_t cols = (size_t)M * K * 9;
  a.cols = (int64_t)cols;
  // partials: the deferred arena when the sum accumulates into a trainer's gradient buffer (one table-driven launch at the end of
  // backward runs all such sums), else the caller's workspace
  float* part = accumulate ? deferred_take((size_t)a.S * cols, (hipStream_t)stream) : nullptr;
  if (!part) {
    MI_CHECK_ARG(ws, "conv3x3_wgrad: null workspace");
    part = (float*)ws;
  }
  a.part = part;
  hipStream_t st = (hipStream_t)stream;
  constexpr int LDSB = 3 * 16 * C3_XSW * 2 + 64 * C3_DS * 2;
  const dim3 grid((unsigned)cdiv(a.K, 16), (unsigned)cdiv(a.M, 64), (unsigned)a.S);
  MI_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "conv3x3_wgrad: grid too large");
  {
    const double N = (double)HW * B;
    ProfScope ps(st, K_CONV3X3, (double)(K + M) * N * 2.0 + (double)a.S * cols * 4.0, 2.0 * 9.0 * M * K * N);
#define C3W_CASE(TWP_) \
    if (TWP == TWP_) { \
      MI_CHECK_HIP(hipFuncSetAttribute((const void*)c3_wgrad_kernel<TWP_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
      hipLaunchKernelGGL((c3_wgrad_kernel<TWP_>), grid, dim3(256), LDSB, st, a); \
    }
    C3W_CASE(64) C3W_CASE(32) C3W_CASE(16)
#undef C3W_CASE
    MI_LAUNCH_CHECK();
  }
  return launch_reduce_rows(part, dw, a.S, (int64_t)cols, (int64_t)cols, accumulate, 1.0f, st, nullptr, nullptr, 0);
}
