// Fused GDFN half-block, forward:  out = y + project_out( gelu(dw(project_in(LN(y)))[0:h]) * dw(...)[h:2h] )
// (Restormer.py:76-93 FeedForward + :148 `x + ffn(norm2(x))`; moce_ir.py:255-276,834; AdaIR-main/net/model.py:76-94,170)
// in ONE launch: y is read once (plus a one-pixel halo), out is written once, nothing else touches HBM.
//
// Work decomposition (bf16 activations, fp32 accumulate):
//   * a workgroup (8 waves) owns a TH x 64 pixel tile of one image and keeps, for the whole kernel,
//       - LN(y) on the tile + halo as MFMA A-operand fragments in registers (wave w owns 16-pixel m-tiles w, w+8, ...),
//       - the fp32 accumulators of the output tile (wave w owns tile rows w, w+8: 96 registers per lane).
//   * the hidden dimension is walked in chunks of PC gate pairs (channels j and j+h ride together):
//       GEMM1   h0[2PC][tile+halo] = W_in'[chunk] . LN(y)          MFMA 16x16x32/16x16x16, result -> LDS (bf16, zero outside
//                                                                    the image = the depthwise conv's zero padding)
//       conv    g[PC][tile] = gelu(dw3x3(h0[j])) * dw3x3(h0[j+h])  VALU, packed fp32 (the pair in the two halves), 8 pixels
//                                                                    of a row per lane, neighbours by DPP, result -> LDS
//       GEMM2   acc[tile][C] += g^T . W_out[chunk]^T               MFMA, A operand = transposed LDS reads of g
//     with two workgroup barriers per chunk; the next chunk's weights are staged into LDS behind the conv phase.
//   * LayerNorm's affine is folded into the packed weights (W_in' = W_in diag(gamma), b' = W_in beta + b_in), so the
//     kernel only centres and scales; mean / rstd of the tile's own pixels are written for the backward pass.
// Pixel index space of a tile ("linear halo pixel"): [0, HR*64) = rows y0-1 .. y0+TH of the 64 tile columns,
// then HR left-halo pixels (column x0-1), then HR right-halo pixels (column x0+64), padded to a multiple of 16.
#include <stdlib.h>

#include <atomic>

#include "fused_common.h"
#include "internal.h"

namespace mi {
using namespace fz;
#ifndef FG_SAVE_PREFETCH
#define FG_SAVE_PREFETCH 1
#endif

template <int C_, int TH_, int TW_, int PC_, int NW_> struct FgCfg {
  static constexpr int C = C_, TH = TH_, TW = TW_, PC = PC_, NW = NW_;
  static constexpr int NT = 64 * NW;                   // threads per workgroup
  static constexpr int HR = TH + 2;
  static constexpr int BODY = HR * TW;
  static constexpr int HPX = BODY + 2 * HR;
  static constexpr int HPXP = (HPX + 15) / 16 * 16;
  static constexpr int MT = HPXP / 16;                 // 16-pixel m-tiles of GEMM1
  static constexpr int MTW = (MT + NW - 1) / NW;       // per wave
  static constexpr int PLANE = (HPXP % 16 == 8) ? HPXP : HPXP + 8;   // plane stride (elements): 16-byte aligned planes
  static constexpr int KS32 = C / 32, KT16 = (C % 32) / 16;
  static constexpr int NV = 8 * KS32 + 4 * KT16;       // channel values per lane per pixel (C / 4)
  static constexpr int W1S = C + 8;                    // LDS row strides (elements): conflict-free 8-byte operand reads
  static constexpr int W2S = PC + 8;
  static constexpr int GS = TH * TW + 16;              // g row stride: conflict-free transposed reads
  static constexpr int NT1 = 2 * PC / 16;              // 16-channel n-tiles of GEMM1
  static constexpr int CT = C / 16;                    // 16-channel tiles of the output
  static constexpr int VPR = TW / 8;                   // 16-byte vectors per tile row
  static constexpr int QT = TW / 16;                   // 16-pixel tiles per tile row
  static constexpr int RPW = TH / NW;                  // tile rows owned by a wave (GEMM2 / epilogue)
  // conv phase: a lane owns 8 pixels of one row; a wave covers ROWS rows of NPAIR gate pairs per pass
  static constexpr int CG = TW / 8;
  static constexpr int RPP = 64 / CG;
  static constexpr int ROWS = TH < RPP ? TH : RPP;
  static constexpr int NPAIR = RPP / ROWS;
  static constexpr int PASSES = TH / ROWS;
  static constexpr int PPW = PC / NW;                  // gate pairs per wave per chunk
  static constexpr int H0_BYTES = 2 * PC * PLANE * 2;
  static constexpr int G_BYTES = PC * GS * 2;
  static constexpr int W1_BYTES = 2 * PC * W1S * 2;
  static constexpr int W2_BYTES = C * W2S * 2;
  static constexpr int WD_BYTES = PC * 20 * 4;         // depthwise taps + bias of a chunk, fp32 (two buffers)
  static constexpr int S_BYTES = C * PLANE * 2;        // prologue: raw y staged plane-major (aliases everything)
  static constexpr int MAIN_BYTES = H0_BYTES + G_BYTES + W1_BYTES + W2_BYTES + 2 * WD_BYTES;
  static constexpr int LDS_BYTES = MAIN_BYTES > S_BYTES ? MAIN_BYTES : S_BYTES;
  static constexpr int SLAB_OS = TW + 4;               // epilogue: wave-private fp32 slab [16 ch][TW px + pad]
  static constexpr int LPR = TW / 8;                   // epilogue: lanes per output row
  static constexpr int RPI = 64 / LPR;                 // rows per store instruction
  static constexpr int ITS = 16 / RPI;
  static_assert(C % 16 == 0 && (TW == 32 || TW == 64) && (PC == 16 || PC == 32) && (NW == 4 || NW == 8), "unsupported tile");
  static_assert(TH % NW == 0 && TH % ROWS == 0 && PPW % NPAIR == 0 && PC % NW == 0, "tile rows / pairs must split over the waves");
  static_assert(GS % 128 == 16, "g row stride");
  static_assert(NW * 16 * SLAB_OS * 4 <= H0_BYTES, "epilogue slabs live in the h0 region");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct FgArgs {
  const bf16* y; bf16* out; float* mean; float* rstd;
  const bf16* w1p; const bf16* w2p; const float* wdp; const float* b2;
  int B, H, W, nch, with_bias, tiles_x, tiles_y, dbg, xcd_pairs;
  float f8_x1, f8_w1, f8_x2, f8_w2;    // fp8 operand form: scales of the normalised input, W_in', the gated hidden tensor, W_out
  bf16* h0s; bf16* gs; int hidden;     // training form (SAVE): project_in output [B][2h][H][W] and gate output [B][h][H][W] for the backward
};

// packed-weight blob layout (bytes from its base), shared by the pack kernel and the launcher
struct FgPackLayout { size_t w1p, w2p, wdp, b2, bytes; int nch; };
static FgPackLayout fg_pack_layout(int C, int hidden, int PC) {
  FgPackLayout l;
  l.nch = cdiv(hidden, PC);
  size_t off = 0;
  l.w1p = off; off = align_up(off + (size_t)l.nch * 2 * PC * (C + 8) * 2, 256);
  l.w2p = off; off = align_up(off + (size_t)l.nch * C * (PC + 8) * 2, 256);
  l.wdp = off; off = align_up(off + (size_t)l.nch * PC * 20 * 4, 256);
  l.b2 = off; off = align_up(off + (size_t)C * 4, 256);
  l.bytes = off;
  return l;
}

struct FgPackArgs {
  const float *ln_w, *ln_b, *in_w, *in_b, *dw_w, *dw_b, *out_w, *out_b;
  bf16* w1p; bf16* w2p; float* wdp; float* b2;
  int C, h, PC, nch;
};

__global__ __launch_bounds__(256) void fg_pack_kernel(FgPackArgs a) {
  const int C = a.C, h = a.h, PC = a.PC, W1S = C + 8, W2S = PC + 8;
  const int64_t n_w1 = (int64_t)a.nch * 2 * PC * W1S, n_w2 = (int64_t)a.nch * C * W2S,
                n_wd = (int64_t)a.nch * PC * 20, n_b2 = C;
  const int64_t total = n_w1 + n_w2 + n_wd + n_b2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t r = e;
    if (r < n_w1) {                                   // W_in'[chunk][row][k] = W_in[hidden row][k] * gamma[k]
      const int k = (int)(r % W1S); r /= W1S;
      const int rr = (int)(r % (2 * PC)), c = (int)(r / (2 * PC));
      const int jl = c * PC + rr % PC, hid = (rr / PC) * h + jl;
      if (k < C) {
        a.w1p[e] = (bf16)(jl < h ? a.in_w[(int64_t)hid * C + k] * a.ln_w[k] : 0.f);
      } else if (k == C) {                            // the row's fp32 bias b' = b_in + W_in . beta rides in the row padding
        float sb = 0.f;
        if (jl < h) {
          sb = a.in_b ? a.in_b[hid] : 0.f;
          if (a.ln_b)
            for (int kk = 0; kk < C; ++kk) sb += a.in_w[(int64_t)hid * C + kk] * a.ln_b[kk];
        }
        *reinterpret_cast<float*>(&a.w1p[e]) = sb;
      } else if (k >= C + 2) {
        a.w1p[e] = (bf16)0.f;
      }
      continue;
    }
    r -= n_w1;
    if (r < n_w2) {                                   // W_out^T chunk: [chunk][out channel][pair]
      const int kk = (int)(r % W2S); r /= W2S;
      const int n = (int)(r % C), c = (int)(r / C);
      const int jl = c * PC + kk;
      a.w2p[r * W2S + kk] = (bf16)((kk < PC && jl < h) ? a.out_w[(int64_t)n * h + jl] : 0.f);
      continue;
    }
    r -= n_w2;
    if (r < n_wd) {                                   // [chunk][pair][tap 0..8, bias][half]
      const int half = (int)(r % 2); int64_t q = r / 2;
      const int tp = (int)(q % 10); q /= 10;
      const int p = (int)(q % PC), c = (int)(q / PC);
      const int jl = c * PC + p, hid = half * h + jl;
      float v = 0.f;
      if (jl < h) v = tp < 9 ? a.dw_w[(int64_t)hid * 9 + tp] : (a.dw_b ? a.dw_b[hid] : 0.f);
      a.wdp[r] = v;
      continue;
    }
    r -= n_wd;
    a.b2[r] = a.out_b ? a.out_b[r] : 0.f;
  }
}

template <int C, int TH, int TW, int PC, int NW, bool F8, bool SAVE = false>
__global__ __launch_bounds__(64 * NW, 2) void fg_fwd_kernel(FgArgs a) {
  using K = FgCfg<C, TH, TW, PC, NW>;
  constexpr int NT = K::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char fg_lds[];
  bf16* const H0 = reinterpret_cast<bf16*>(fg_lds);
  bf16* const G = reinterpret_cast<bf16*>(fg_lds + K::H0_BYTES);
  bf16* const W1 = reinterpret_cast<bf16*>(fg_lds + K::H0_BYTES + K::G_BYTES);
  bf16* const W2 = reinterpret_cast<bf16*>(fg_lds + K::H0_BYTES + K::G_BYTES + K::W1_BYTES);
  float* const WD = reinterpret_cast<float*>(fg_lds + K::H0_BYTES + K::G_BYTES + K::W1_BYTES + K::W2_BYTES);   // [2][PC][20]
  bf16* const S = reinterpret_cast<bf16*>(fg_lds);
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  int bid = blockIdx.x;
  if (a.xcd_pairs) {                                   // x-adjacent 32-pixel tiles (the two halves of 128-byte lines) on one XCD
    const int q16 = bid >> 4, r16 = bid & 15;
    bid = 2 * (8 * q16 + (r16 & 7)) + (r16 >> 3);
  }
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int64_t HW = (int64_t)a.H * a.W;
  const bf16* const yb = a.y + (int64_t)b * C * HW;

  // ---------------------------------------------------------------- prologue: stage raw y (tile + halo), plane-major
  {
    constexpr int NB = C * K::HR * K::VPR;             // 16-byte vectors of the tile body
    constexpr int NBV = (NB + NT - 1) / NT;
    constexpr int NE = K::HPXP - K::BODY;              // halo-column pixels + padding per plane
    constexpr int NEV = (C * NE + NT - 1) / NT;
    u32x4 raw[NBV];
    u16 rawe[NEV];
#pragma unroll
    for (int n = 0; n < NBV; ++n) {
      const int idx = t + NT * n;
      const int c = idx / (K::HR * K::VPR), rem = idx - c * (K::HR * K::VPR), r = rem / K::VPR, u = rem % K::VPR;
      const int Y = y0 - 1 + r;
      raw[n] = (u32x4){0u, 0u, 0u, 0u};
      if (idx < NB && Y >= 0 && Y < a.H)
        raw[n] = *reinterpret_cast<const u32x4*>(yb + (int64_t)c * HW + (int64_t)Y * a.W + x0 + 8 * u);
    }
#pragma unroll
    for (int n = 0; n < NEV; ++n) {
      const int idx = t + NT * n;
      const int c = idx / NE, k = idx - c * NE;
      rawe[n] = 0;
      if (idx < C * NE && k < 2 * K::HR) {
        const int side = k >= K::HR ? 1 : 0, r = k - side * K::HR;
        const int Y = y0 - 1 + r, X = side ? x0 + TW : x0 - 1;
        if (Y >= 0 && Y < a.H && X >= 0 && X < a.W)
          rawe[n] = reinterpret_cast<const u16*>(yb)[(int64_t)c * HW + (int64_t)Y * a.W + X];
      }
    }
#pragma unroll
    for (int n = 0; n < NBV; ++n) {
      const int idx = t + NT * n;
      const int c = idx / (K::HR * K::VPR), rem = idx - c * (K::HR * K::VPR), r = rem / K::VPR, u = rem % K::VPR;
      if (idx < NB) *reinterpret_cast<u32x4*>(&S[c * K::PLANE + r * TW + 8 * u]) = raw[n];
    }
#pragma unroll
    for (int n = 0; n < NEV; ++n) {
      const int idx = t + NT * n;
      const int c = idx / NE, k = idx - c * NE;
      if (idx < C * NE) reinterpret_cast<u16*>(S)[c * K::PLANE + K::BODY + k] = rawe[n];
    }
  }
  __syncthreads();

  // ---------------------------------------------------------------- LN(y) -> A-operand fragments (registers)
  // element order of a 32-k fragment (same for A and B): j < 4 is k = 4g + j, j >= 4 is k = 16 + 4g + (j - 4)
  s16x8 xa[K::MTW][K::KS32 > 0 ? K::KS32 : 1];
  s16x4 xt[K::MTW];
  unsigned long long vmask = 0;                        // bit 4i + r: pixel r of this lane's row group in m-tile i is inside the image
  static_assert(4 * K::MTW <= 64, "validity mask");
#pragma unroll
  for (int i = 0; i < K::MTW; ++i) {
    const int mt = wv + NW * i;
    if (mt < K::MT) {
      const bf16* sp = &S[(4 * g + qq) * K::PLANE + mt * 16 + 4 * pp];
      s16x4 lo[K::KS32 > 0 ? K::KS32 : 1], hi[K::KS32 > 0 ? K::KS32 : 1], tl = {0, 0, 0, 0}, dm = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < K::KS32; ++ks) {
        lo[ks] = tr_b16(sp + (ks * 32) * K::PLANE);
        hi[ks] = tr_b16(sp + (ks * 32 + 16) * K::PLANE);
      }
      if (K::KT16) tl = tr_b16(sp + (K::KS32 * 32) * K::PLANE);
      if constexpr (K::KS32 == 3) lds_wait(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], tl, dm);
      else if constexpr (K::KS32 == 2) lds_wait(lo[0], hi[0], lo[1], hi[1], tl, dm);
      else if constexpr (K::KS32 == 1) lds_wait(lo[0], hi[0], tl, dm);
      else lds_wait(tl);
      float v[K::NV];
#pragma unroll
      for (int ks = 0; ks < K::KS32; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[8 * ks + j] = bf_s(lo[ks][j]); v[8 * ks + 4 + j] = bf_s(hi[ks][j]); }
      if (K::KT16)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[8 * K::KS32 + j] = bf_s(tl[j]);
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < K::NV; ++j) s += v[j];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mu = s * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < K::NV; ++j) { const float d = v[j] - mu; q += d * d; }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q * (1.0f / C) + 1e-5f);
      const float sub = a.with_bias ? mu : 0.f;        // BiasFree: x / sqrt(var + eps), x not centred (Restormer.py:37-39)
#pragma unroll
      for (int j = 0; j < K::NV; ++j) v[j] = (v[j] - sub) * rstd;
#pragma unroll
      for (int ks = 0; ks < K::KS32; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[i][ks][j] = bf_bits(v[8 * ks + j]);
      if (K::KT16)
#pragma unroll
        for (int j = 0; j < 4; ++j) xt[i][j] = bf_bits(v[8 * K::KS32 + j]);
      // statistics of the tile's own pixels (each image pixel is the centre pixel of exactly one tile)
      if (a.mean && g == 0) {
        const int ipx = mt * 16 + li;
        if (ipx < K::BODY) {
          const int rr = ipx / TW, col = ipx % TW;
          if (rr >= 1 && rr <= TH) {
            const int64_t o = (int64_t)b * HW + (int64_t)(y0 - 1 + rr) * a.W + x0 + col;
            a.mean[o] = mu; a.rstd[o] = rstd;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ipx = mt * 16 + 4 * g + r;
        bool ok;
        if (ipx < K::BODY) { const int Y = y0 - 1 + ipx / TW; ok = Y >= 0 && Y < a.H; }
        else if (ipx < K::HPX) {
          const int k = ipx - K::BODY, side = k >= K::HR ? 1 : 0, rr = k - side * K::HR;
          const int Y = y0 - 1 + rr, X = side ? x0 + TW : x0 - 1;
          ok = Y >= 0 && Y < a.H && X >= 0 && X < a.W;
        } else ok = false;
        vmask |= (ok ? 1ull : 0ull) << (4 * i + r);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < K::KS32; ++ks) xa[i][ks] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
      xt[i] = (s16x4){0, 0, 0, 0};
    }
  }
  if ((a.dbg & 15) == 1) {                             // debug: LN output of tile 0 as [pixel][channel] bf16
    if (blockIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < K::MTW; ++i) {
        const int mt = wv + NW * i;
        if (mt < K::MT) {
          bf16* o = a.out + (int64_t)(mt * 16 + li) * C;
#pragma unroll
          for (int ks = 0; ks < K::KS32; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j)
              reinterpret_cast<short*>(o)[ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4)] = xa[i][ks][j];
          if (K::KT16)
#pragma unroll
            for (int j = 0; j < 4; ++j) reinterpret_cast<short*>(o)[K::KS32 * 32 + 4 * g + j] = xt[i][j];
        }
      }
    }
    return;
  }
  // MFMA A operands of GEMM1 in their final form (bf16 as they are; fp8: converted once, here)
  using Op = MfmaOp<F8>;
  Op::enter();
  typename Op::Frag xaf[K::MTW][K::KS32 > 0 ? K::KS32 : 1], xtf[K::MTW];
#pragma unroll
  for (int i = 0; i < K::MTW; ++i) {
#pragma unroll
    for (int ks = 0; ks < K::KS32; ++ks) xaf[i][ks] = Op::cvt(xa[i][ks], a.f8_x1);
    xtf[i] = Op::cvt(cat8(xt[i], (s16x4){0, 0, 0, 0}), a.f8_x1);
  }
  __syncthreads();                                     // the staged y is dead: the region becomes h0 / g / weights

  constexpr int W1V = K::W1_BYTES / 16, W2V = K::W2_BYTES / 16, WDV = K::WD_BYTES / 16;
  constexpr int W1N = (W1V + NT - 1) / NT, W2N = (W2V + NT - 1) / NT;
  static_assert(WDV <= NT, "one vector per thread");
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.w1p);
    for (int v = t; v < W1V; v += NT) reinterpret_cast<u32x4*>(W1)[v] = src[v];
    if (t < WDV) reinterpret_cast<u32x4*>(WD)[t] = reinterpret_cast<const u32x4*>(a.wdp)[t];
  }
  __syncthreads();

  f32x4 acc[K::RPW][K::QT][K::CT];
#pragma unroll
  for (int j = 0; j < K::RPW; ++j)
#pragma unroll
    for (int q = 0; q < K::QT; ++q)
#pragma unroll
      for (int ct = 0; ct < K::CT; ++ct) acc[j][q][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int lane_outer = lane;
  for (int c = 0; c < a.nch; ++c) {
    // Lane coordinates of the chunk loop come from an opaque copy of the lane id (they shadow the outer ones): the per-lane LDS
    // addresses of the three phases are loop-invariant, and hoisted out of this loop they are what the kernel spills.
    int lane_o = lane_outer;
    asm volatile("" : "+v"(lane_o));
    const int lane = lane_o, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3, t = wv * 64 + lane;
    // ------------------------------------------------------------ GEMM1: h0 chunk = W_in'[chunk] . LN(y), to LDS
    if (!(a.dbg & 32)) {
#pragma unroll
      for (int nt = 0; nt < K::NT1; ++nt) {
        const int row = nt * 16 + li;
        const bf16* wr = &W1[row * K::W1S + 4 * g];
        s16x8 bw[K::KS32 > 0 ? K::KS32 : 1];
        s16x4 bt = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks)
          bw[ks] = cat8(*reinterpret_cast<const s16x4*>(wr + ks * 32), *reinterpret_cast<const s16x4*>(wr + ks * 32 + 16));
        if (K::KT16) bt = *reinterpret_cast<const s16x4*>(wr + K::KS32 * 32);
        const float bias = *reinterpret_cast<const float*>(&W1[row * K::W1S + C]);
        typename Op::Frag bwf[K::KS32 > 0 ? K::KS32 : 1], btf;
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks) bwf[ks] = Op::cvt(bw[ks], a.f8_w1);
        btf = Op::cvt(cat8(bt, (s16x4){0, 0, 0, 0}), a.f8_w1);
        const float os1 = F8 ? a.f8_x1 * a.f8_w1 : 1.f;
        bf16* hrow = &H0[row * K::PLANE + 4 * g];
#pragma unroll
        for (int i = 0; i < K::MTW; ++i) {
          const int mt = wv + NW * i;
          if (mt < K::MT) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < K::KS32; ++ks) d = Op::mma(xaf[i][ks], bwf[ks], d);
            // a 16-deep tail rides in a zero-padded 32-deep MFMA: a dependent chain that mixes the 16x16x32 and
            // 16x16x16 shapes returned garbage accumulators on gfx950 (ROCm 7.2) - one opcode per accumulator chain
            if (K::KT16) d = Op::mma(xtf[i], btf, d);
            if (F8) d *= os1;
            const unsigned m = (unsigned)(vmask >> (4 * i));
            u32x2 o;
            o[0] = pack_bf2((m & 1u) ? d[0] + bias : 0.f, (m & 2u) ? d[1] + bias : 0.f);
            o[1] = pack_bf2((m & 4u) ? d[2] + bias : 0.f, (m & 8u) ? d[3] + bias : 0.f);
            *reinterpret_cast<u32x2*>(hrow + mt * 16) = o;
          }
        }
      }
    }
    __syncthreads();
    if ((a.dbg & 15) == 2) {                           // debug: h0 chunk 0 of tile 0 as [2PC][HPXP]
      if (blockIdx.x == 0)
        for (int e = t; e < 2 * PC * K::HPXP; e += NT) a.out[e] = H0[(e / K::HPXP) * K::PLANE + e % K::HPXP];
      return;
    }

    if constexpr (SAVE) {
      // training form: the finished h0 chunk (tile rows only, no halo) goes to HBM for the backward's recompute of the conv
      // (before the weight prefetch below claims its registers)
      constexpr int SV = 2 * PC * TH * K::VPR;         // 16-byte vectors
      static_assert(SV % NT == 0, "h0 rows split evenly over the threads");
      bf16* const hb = a.h0s + (((int64_t)b * 2 * a.hidden + c * PC) * a.H + y0) * a.W + x0;      // wave-uniform base
      const int hstep = a.hidden * a.H * a.W;
#pragma unroll
      for (int n = 0; n < SV / NT; ++n) {
        const int v = t + NT * n;
        const int row = v / (TH * K::VPR), rem = v - row * (TH * K::VPR), r = rem / K::VPR, u = rem - r * K::VPR;
        const int half = row / PC, p = row - half * PC;
        if (c * PC + p < a.hidden) {
          const u32x4 hv = *reinterpret_cast<const u32x4*>(&H0[row * K::PLANE + (r + 1) * TW + 8 * u]);
          *reinterpret_cast<u32x4*>(hb + (int64_t)half * hstep + (p * a.H + r) * a.W + 8 * u) = hv;
        }
      }
    }
    // ------------------------------------------------------------ depthwise 3x3 + GELU gate, VALU; weights for the next GEMMs in flight
    {
      u32x4 wr1[W1N], wr2[W2N], wrd = {0u, 0u, 0u, 0u};
      const bool more = c + 1 < a.nch;
      const float* const wdc = WD + (c & 1) * (PC * 20);
      const u32x4* const s1 = reinterpret_cast<const u32x4*>(a.w1p + (int64_t)(c + 1) * 2 * PC * K::W1S);
      const u32x4* const s2 = reinterpret_cast<const u32x4*>(a.w2p + (int64_t)c * C * K::W2S);
      if constexpr (!SAVE || FG_SAVE_PREFETCH) {
        if (more && t < WDV) wrd = reinterpret_cast<const u32x4*>(a.wdp + (int64_t)(c + 1) * PC * 20)[t];
#pragma unroll
        for (int n = 0; n < W1N; ++n) { const int v = t + NT * n; if (more && v < W1V) wr1[n] = s1[v]; }
#pragma unroll
        for (int n = 0; n < W2N; ++n) { const int v = t + NT * n; if (v < W2V) wr2[n] = s2[v]; }
      }
      const int cg = lane % K::CG, rl = (lane / K::CG) % K::ROWS, psel = lane / (K::CG * K::ROWS);
      bf16* const gbase = SAVE ? a.gs + (((int64_t)b * a.hidden + c * PC) * a.H + y0) * a.W + x0 : nullptr;   // wave-uniform
#pragma unroll 1
      for (int s = 0; s < ((a.dbg & 64) ? 0 : K::PPW / K::NPAIR); ++s) {
        const int p = wv * K::PPW + s * K::NPAIR + psel;
        const float* wp = wdc + p * 20;
        f32x2 w[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          if constexpr (K::NPAIR == 1) {               // one pair per wave: the taps live in scalar registers
            w[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wp[2 * i])));
            w[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wp[2 * i + 1])));
          } else {
            w[i] = *reinterpret_cast<const f32x2*>(wp + 2 * i);
          }
        }
        const bf16* h1 = &H0[p * K::PLANE];
        const bf16* h2 = &H0[(PC + p) * K::PLANE];
        const int eoff = K::BODY + (cg == K::CG - 1 ? K::HR : 0);
#pragma unroll
        for (int rp = 0; rp < K::PASSES; ++rp) {
          const int row = rp * K::ROWS + rl;
          f32x2 o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = w[9];
#pragma unroll
          for (int dr = 0; dr < 3; ++dr) {
            const int rin = row + dr;
            const u32x4 r1 = *reinterpret_cast<const u32x4*>(h1 + rin * TW + 8 * cg);
            const u32x4 r2 = *reinterpret_cast<const u32x4*>(h2 + rin * TW + 8 * cg);
            const u16 e1 = reinterpret_cast<const u16*>(h1)[eoff + rin];
            const u16 e2 = reinterpret_cast<const u16*>(h2)[eoff + rin];
            f32x2 v[10];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[1 + 2 * k][0] = bf_lo(r1[k]); v[2 + 2 * k][0] = bf_hi(r1[k]);
              v[1 + 2 * k][1] = bf_lo(r2[k]); v[2 + 2 * k][1] = bf_hi(r2[k]);
            }
            f32x2 edge, lft, rgt;
            edge[0] = bf_lo(e1); edge[1] = bf_lo(e2);
            lft[0] = from_prev_lane(v[8][0]); lft[1] = from_prev_lane(v[8][1]);
            rgt[0] = from_next_lane(v[1][0]); rgt[1] = from_next_lane(v[1][1]);
            v[0] = cg == 0 ? edge : lft;
            v[9] = cg == K::CG - 1 ? edge : rgt;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) o[j] += w[dr * 3 + kx] * v[j + kx];
          }
          float gg[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) gg[j] = gelu_fwd<bf16>(o[j][0]) * o[j][1];
          Vec<bf16, 8>::st(&G[p * K::GS + row * TW + 8 * cg], gg);
          if constexpr (SAVE) {
            if (c * PC + p < a.hidden)
              Vec<bf16, 8>::st(gbase + (p * a.H + row) * a.W + 8 * cg, gg);
          }
        }
      }
      if constexpr (SAVE && !FG_SAVE_PREFETCH) {
        if (more && t < WDV) wrd = reinterpret_cast<const u32x4*>(a.wdp + (int64_t)(c + 1) * PC * 20)[t];
#pragma unroll
        for (int n = 0; n < W1N; ++n) { const int v = t + NT * n; if (more && v < W1V) wr1[n] = s1[v]; }
#pragma unroll
        for (int n = 0; n < W2N; ++n) { const int v = t + NT * n; if (v < W2V) wr2[n] = s2[v]; }
      }
#pragma unroll
      for (int n = 0; n < W1N; ++n) { const int v = t + NT * n; if (more && v < W1V) reinterpret_cast<u32x4*>(W1)[v] = wr1[n]; }
#pragma unroll
      for (int n = 0; n < W2N; ++n) { const int v = t + NT * n; if (v < W2V) reinterpret_cast<u32x4*>(W2)[v] = wr2[n]; }
      if (more && t < WDV) reinterpret_cast<u32x4*>(WD + ((c + 1) & 1) * (PC * 20))[t] = wrd;
    }
    __syncthreads();
    if ((a.dbg & 15) == 3) {                           // debug: gate output chunk 0 of tile 0 as [PC][TH*TW]
      if (blockIdx.x == 0)
        for (int e = t; e < PC * TH * TW; e += NT) a.out[e] = G[(e / (TH * TW)) * K::GS + e % (TH * TW)];
      return;
    }

    // ------------------------------------------------------------ GEMM2: acc[tile rows of this wave][C] += g^T . W_out[chunk]^T
    if (!(a.dbg & 128)) {
      s16x8 bo[K::CT];
#pragma unroll
      for (int ct = 0; ct < K::CT; ++ct) {
        const bf16* wr = &W2[(ct * 16 + li) * K::W2S + 4 * g];
        if (PC == 32) bo[ct] = cat8(*reinterpret_cast<const s16x4*>(wr), *reinterpret_cast<const s16x4*>(wr + 16));
        else bo[ct] = cat8(*reinterpret_cast<const s16x4*>(wr), (s16x4){0, 0, 0, 0});   // 16 pairs: zero-padded k
      }
      typename Op::Frag bof[K::CT];
#pragma unroll
      for (int ct = 0; ct < K::CT; ++ct) bof[ct] = Op::cvt(bo[ct], a.f8_w2);
#pragma unroll
      for (int j = 0; j < K::RPW; ++j) {
        const bf16* gp = &G[(4 * g + qq) * K::GS + (wv + NW * j) * TW + 4 * pp];
        s16x4 lo[K::QT], hi[K::QT];
#pragma unroll
        for (int q = 0; q < K::QT; ++q) {
          lo[q] = tr_b16(gp + 16 * q);
          hi[q] = (s16x4){0, 0, 0, 0};
          if (PC == 32) hi[q] = tr_b16(gp + 16 * K::GS + 16 * q);
        }
        if constexpr (K::QT == 4) lds_wait(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
        else lds_wait(lo[0], lo[1], hi[0], hi[1]);
#pragma unroll
        for (int q = 0; q < K::QT; ++q) {
          const typename Op::Frag gf = Op::cvt(cat8(lo[q], hi[q]), a.f8_x2);
#pragma unroll
          for (int ct = 0; ct < K::CT; ++ct) acc[j][q][ct] = Op::mma(gf, bof[ct], acc[j][q][ct]);
        }
      }
    }
  }

  // ---------------------------------------------------------------- epilogue: + bias + residual y, whole row segments
  // (wave-private fp32 slab inside the h0 region: nobody reads h0 after the last barrier)
  {
    float* slab = reinterpret_cast<float*>(fg_lds) + wv * 16 * K::SLAB_OS;
    const int e_row = lane / K::LPR, e_col = (lane % K::LPR) * 8;
    u32x4 rr[K::RPW][K::CT][K::ITS];
    float bv[K::CT][K::ITS];
#pragma unroll
    for (int ct = 0; ct < K::CT; ++ct)
#pragma unroll
      for (int it = 0; it < K::ITS; ++it) bv[ct][it] = a.b2[ct * 16 + it * K::RPI + e_row];
#pragma unroll
    for (int j = 0; j < K::RPW; ++j) {
      const int64_t pix = (int64_t)(y0 + wv + NW * j) * a.W + x0 + e_col;
#pragma unroll
      for (int ct = 0; ct < K::CT; ++ct)
#pragma unroll
        for (int it = 0; it < K::ITS; ++it)
          rr[j][ct][it] = *reinterpret_cast<const u32x4*>(yb + (int64_t)(ct * 16 + it * K::RPI + e_row) * HW + pix);
    }
#pragma unroll
    for (int j = 0; j < K::RPW; ++j) {
      const int64_t pix = (int64_t)(y0 + wv + NW * j) * a.W + x0 + e_col;
#pragma unroll
      for (int ct = 0; ct < K::CT; ++ct) {
#pragma unroll
        for (int q = 0; q < K::QT; ++q) {
          const float os2 = F8 ? a.f8_x2 * a.f8_w2 : 1.f;
          float v[4] = {acc[j][q][ct][0] * os2, acc[j][q][ct][1] * os2, acc[j][q][ct][2] * os2, acc[j][q][ct][3] * os2};
          Vec<float, 4>::st(&slab[li * K::SLAB_OS + 16 * q + 4 * g], v);
        }
        wave_sync();
#pragma unroll
        for (int it = 0; it < K::ITS; ++it) {
          const int m = ct * 16 + it * K::RPI + e_row;
          float v[8];
          Vec<float, 4>::ld(&slab[(it * K::RPI + e_row) * K::SLAB_OS + e_col], v);
          Vec<float, 4>::ld(&slab[(it * K::RPI + e_row) * K::SLAB_OS + e_col + 4], v + 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[2 * k] += bv[ct][it] + bf_lo(rr[j][ct][it][k]);
            v[2 * k + 1] += bv[ct][it] + bf_hi(rr[j][ct][it][k]);
          }
          Vec<bf16, 8>::st(a.out + ((int64_t)b * C + m) * HW + pix, v);
        }
        wave_sync();
      }
    }
  }
}

// ================================================================================================ round 4: fourth form (fg4)
// The half-block on the machinery of fused_mdta.hip's fourth form: the VALU was the bound of the kernel above (both depthwise
// convs, the bias adds, masks and packs: ~5000 vector instructions per wave and tile), so here
//   * the depthwise 3x3 of BOTH gate halves runs on the matrix cores (im2col MFMA against block-diagonal tap matrices over a
//     PIXEL-major h0 in LDS; bias = a tenth tap against a record of ones), GEMM1 is flipped, its bias and the conv's zero padding
//     ride in the GEMM ("ones" channel) - see the comment block above Fm3Cfg in fused_mdta.hip;
//   * a persistent 8-wave workgroup walks 8 x 32 tiles; the hidden dimension is walked in groups of GP = 2 pair chunks (32 gate
//     pairs = 64 hidden channels); phase p runs GEMM1 of group p + 1 (all waves, split by pixels) into the other h0 buffer and,
//     WAVE-LOCAL, the rest of group p for tile row w: conv of the pair's two halves -> GELU gate in registers -> a wave-private
//     staging patch (from which the SAVE form stores g, and the A operand of GEMM2 is read back transposed) -> one 32-deep
//     k-step of GEMM2 into the wave's output accumulators.  ONE barrier per group.
//   * the group's weights (W_in fragments, tap words, W_out fragments) stream from L2 into double-buffered LDS by LDS-DMA
//     (global_load_lds, no registers), requested a whole phase ahead; the next tile's x is prefetched behind the epilogue.
//   SAVE (training): h0 rows leave through transposing LDS reads, g from the staging patch - the blob mi_gdfn_bwd_ln reads.
template <int C_, int GP_> struct Fg4Cfg {
  static constexpr int C = C_, TH = 8, TW = 32, NW = 8, GP = GP_;
  static constexpr int NT = 64 * NW;
  static constexpr int HR = TH + 2;
  static constexpr int BODY = HR * TW;
  static constexpr int HPX = BODY + 2 * HR;
  static constexpr int HPXP = (HPX + 15) / 16 * 16;
  static constexpr int MT = HPXP / 16;
  static constexpr int MTW = (MT + NW - 1) / NW;
  static constexpr int PLANE = (HPXP % 16 == 8) ? HPXP : HPXP + 8;
  static constexpr int KS32 = C / 32, KT16 = (C % 32) / 16;
  static constexpr int NKS = KS32 + 1;                    // GEMM1 weight fragments per chunk (the last: 16-deep tail, if any, + bias slot)
  static constexpr int NV = 8 * KS32 + 4 * KT16;
  static constexpr int CT = C / 16;
  static constexpr int VPR = TW / 8;
  static constexpr int RW = TW + 2, REC = 32;
  static constexpr int H0S_BYTES = HR * RW * REC + 64;    // one chunk: records + dummy record + ones record
  static constexpr int NCG = 2 * GP;                      // channel chunks per group (GP chunks of the first half, then GP of the second)
  static constexpr int H0G_BYTES = NCG * H0S_BYTES;
  static constexpr int FRAG = 1024;
  static constexpr int W1G_BYTES = NCG * NKS * FRAG;
  static constexpr int TBG_BYTES = (NCG * 5 * 128 + FRAG - 1) / FRAG * FRAG;   // tap words, padded to whole DMA pieces
  static constexpr int W2G_BYTES = (GP / 2) * CT * FRAG;  // GEMM2: one 32-pair k-step per two pair chunks
  static constexpr int TW2G_BYTES = TBG_BYTES + W2G_BYTES;
  static constexpr int PS = 72;                           // staging patch: bytes per pair row (64 + 8)
  static constexpr int PATCH_BYTES = 16 * GP * PS > 16 * (TW + 4) * 4 ? 16 * GP * PS : 16 * (TW + 4) * 4;   // also the fp32 epilogue slab
  static constexpr int SLAB_OS = TW + 4;
  static constexpr int S_BYTES = C * PLANE * 2;
  static constexpr int A_BYTES = ((2 * H0G_BYTES > S_BYTES ? 2 * H0G_BYTES : S_BYTES) + 15) / 16 * 16;
  static constexpr int LDS_BYTES = A_BYTES + 2 * W1G_BYTES + 2 * TW2G_BYTES + NW * PATCH_BYTES;
  static constexpr int NBV = C * HR * VPR;
  static constexpr int NBN = (NBV + NT - 1) / NT;
  static constexpr int NE = HPXP - BODY;
  static constexpr int NEN = (C * NE + NT - 1) / NT;
  static_assert(C % 16 == 0 && GP == 2, "unsupported configuration");
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

// packed blob of the fourth form (appended to the blob of the forms above): per GROUP of GP pair chunks
//   [W1 fragments NCG x NKS KiB | tap words (padded) | W_out fragments (GP/2) x CT KiB], then the output bias [C] fp32
struct Fg4PackLayout { size_t grp, b2, bytes; int ngr; size_t gstride; };
template <typename K> static Fg4PackLayout fg4_pack_layout(int hidden, size_t base) {
  Fg4PackLayout l;
  const int npc = (hidden + 15) / 16;
  l.ngr = (npc + K::GP - 1) / K::GP;
  if (l.ngr & 1) l.ngr += 1;                               // an even number of groups: buffer parities carry over from tile to tile
  l.gstride = (size_t)K::W1G_BYTES + K::TW2G_BYTES;
  size_t off = align_up(base, 1024);
  l.grp = off; off += (size_t)l.ngr * l.gstride;
  l.b2 = off; off += align_up((size_t)K::C * 4, 256);
  l.bytes = align_up(off, 256);
  return l;
}

struct Fg4PackArgs {
  const float *ln_w, *ln_b, *in_w, *in_b, *dw_w, *dw_b, *out_w, *out_b;
  unsigned char* grp; float* b2;
  int C, h, ngr;
};
template <typename K>
__global__ __launch_bounds__(256) void fg4_pack_kernel(Fg4PackArgs a) {
  constexpr int C = K::C, ks32 = K::KS32;
  const int h = a.h;
  constexpr int64_t n_w1 = (int64_t)K::W1G_BYTES / 2, n_tb = (int64_t)K::TBG_BYTES / 2, n_w2 = (int64_t)K::W2G_BYTES / 2;
  constexpr int64_t n_g = n_w1 + n_tb + n_w2;
  const int64_t total = (int64_t)a.ngr * n_g + C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    if (e >= (int64_t)a.ngr * n_g) { const int c = (int)(e - (int64_t)a.ngr * n_g); a.b2[c] = a.out_b ? a.out_b[c] : 0.f; continue; }
    const int gi = (int)(e / n_g);
    int64_t r = e - (int64_t)gi * n_g;
    bf16* const gb = reinterpret_cast<bf16*>(a.grp + (size_t)gi * ((size_t)K::W1G_BYTES + K::TW2G_BYTES));
    if (r < n_w1) {                                        // [chunk cc][ks][lane][8]; chunk cc = half * GP + pc_local
      const int j = (int)(r & 7), l = (int)((r >> 3) & 63);
      const int64_t q = r >> 9;
      const int ks = (int)(q % K::NKS), cc = (int)(q / K::NKS);
      const int li = l & 15, g = l >> 4;
      const int half = cc / K::GP, jl = (gi * K::GP + cc % K::GP) * 16 + li, hid = half * h + jl;
      int k;
      if (ks < ks32) k = ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
      else k = (j < 4 && ks32 * 32 + 4 * g + j < C) ? ks32 * 32 + 4 * g + j : -1;
      float v = 0.f;
      if (jl < h) {
        if (k >= 0) v = a.in_w[(int64_t)hid * C + k] * a.ln_w[k];
        if (ks == ks32 && g == 0 && j == 4) {             // the "ones" slot: b' = b_in + W_in . beta
          v = a.in_b ? a.in_b[hid] : 0.f;
          if (a.ln_b)
            for (int kk = 0; kk < C; ++kk) v += a.in_w[(int64_t)hid * C + kk] * a.ln_b[kk];
        }
      }
      gb[r] = (bf16)v;
      continue;
    }
    r -= n_w1;
    if (r < n_tb) {                                        // [chunk cc][k-step][lane] u16, then padding
      u16 v = 0;
      if (r < (int64_t)K::NCG * 5 * 64) {
        const int l = (int)(r & 63);
        const int64_t q = r >> 6;
        const int s = (int)(q % 5), cc = (int)(q / 5);
        const int li = l & 15, g = l >> 4, t = 2 * s + (g >> 1);
        const int half = cc / K::GP, jl = (gi * K::GP + cc % K::GP) * 16 + li, hid = half * h + jl;
        if (jl < h && (g & 1) == (li >> 3)) {
          const bf16 w = (bf16)(t <= 8 ? a.dw_w[(int64_t)hid * 9 + t] : (a.dw_b ? a.dw_b[hid] : 0.f));
          v = __builtin_bit_cast(u16, w);
        }
      }
      reinterpret_cast<u16*>(gb)[n_w1 + r] = v;
      continue;
    }
    r -= n_tb;
    {                                                      // W_out fragments [k-step][ct][lane][8]: lane (output channel li of tile ct, k group g)
      const int j = (int)(r & 7), l = (int)((r >> 3) & 63);
      const int64_t q = r >> 9;
      const int ct = (int)(q % K::CT), ks = (int)(q / K::CT);
      const int li = l & 15, g = l >> 4;
      const int pl = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);       // pair within the 32-pair k-step (two pair chunks of 16)
      const int jl = (gi * K::GP + 2 * ks) * 16 + pl;
      gb[n_w1 + n_tb + r] = (bf16)(jl < h ? a.out_w[(int64_t)(ct * 16 + li) * h + jl] : 0.f);
    }
  }
}

struct Fg4Args {
  const bf16* y; bf16* out; float* mean; float* rstd;
  const unsigned char* grp; const float* b2;
  bf16* h0s; bf16* gs;
  int B, H, W, hidden, with_bias, tiles_x, tiles_y, S, ngr, dbg;
};

template <int C, int GP, bool SAVE, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void fg4_fwd_kernel(Fg4Args a) {
  using K = Fg4Cfg<C, GP>;
  constexpr int NT = K::NT, TW = K::TW, TH = K::TH, NW = K::NW, CT = K::CT, NCG = K::NCG;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) unsigned char fg4_lds[];
  unsigned char* const H0 = fg4_lds;                                       // [2][NCG chunk buffers]
  unsigned char* const W1B = fg4_lds + K::A_BYTES;                         // [2][W1G_BYTES]
  unsigned char* const TW2B = W1B + 2 * K::W1G_BYTES;                      // [2][tap words | W_out fragments]
  unsigned char* const PATCH = TW2B + 2 * K::TW2G_BYTES;                   // [NW][PATCH_BYTES]
  bf16* const S = reinterpret_cast<bf16*>(fg4_lds);
  const int t = threadIdx.x, lane_outer = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x / a.S, sp = blockIdx.x - b * a.S;
  const int tiles = a.tiles_x * a.tiles_y;
  const int t0 = (int)((int64_t)sp * tiles / a.S), t1 = (int)((int64_t)(sp + 1) * tiles / a.S);
  const int64_t HW = (int64_t)a.H * a.W;
  const bf16* const yb = a.y + (int64_t)b * C * HW;
  const int ngr = a.ngr;
  constexpr size_t GSTRIDE = (size_t)K::W1G_BYTES + K::TW2G_BYTES;

  // LDS-DMA of one group's weight sections (1-KiB pieces, dealt round-robin to the waves)
  auto dma_w1 = [&](int gi, int buf) {
    const unsigned char* src = a.grp + (size_t)gi * GSTRIDE + lane_outer * 16;
    for (int pc = wv; pc < K::W1G_BYTES / 1024; pc += NW)
      __builtin_amdgcn_global_load_lds((const void*)(src + pc * 1024), (lds_ptr)(W1B + buf * K::W1G_BYTES + pc * 1024), 16, 0, 0);
  };
  auto dma_tw2 = [&](int gi, int buf) {
    const unsigned char* src = a.grp + (size_t)gi * GSTRIDE + K::W1G_BYTES + lane_outer * 16;
    for (int pc = wv; pc < K::TW2G_BYTES / 1024; pc += NW)
      __builtin_amdgcn_global_load_lds((const void*)(src + pc * 1024), (lds_ptr)(TW2B + buf * K::TW2G_BYTES + pc * 1024), 16, 0, 0);
  };

  unsigned long long tk0 = 0, ta = 0, tb_ = 0, cyc[6] = {0, 0, 0, 0, 0, 0};
  if (STAMP) tk0 = fm_clock();
#define FG4_STAMP(i) do { if (STAMP) { tb_ = fm_clock(); cyc[i] += tb_ - ta; ta = tb_; } } while (0)
  FmStage<K> stg;
  if (t0 < t1) {
    fm_stage_load<K>(stg, yb, t, (t0 % a.tiles_x) * TW, (t0 / a.tiles_x) * TH, a.H, a.W, HW);
    dma_w1(0, 0); dma_w1(1 % ngr, 1); dma_tw2(0, 0);
  }

  for (int tile = t0; tile < t1; ++tile) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    if (STAMP) ta = fm_clock();
    __syncthreads();                                      // every wave is done with the previous tile's h0
    fm_stage_store<K>(stg, S, t);
    __syncthreads();
    FG4_STAMP(0);

    int recoff[K::MTW], prow[K::MTW], pcol[K::MTW];
    {
      int lane_g = lane_outer;
      asm volatile("" : "+v"(lane_g));
      const int li = lane_g & 15, g = lane_g >> 4;
#pragma unroll
      for (int i = 0; i < K::MTW; ++i) {
        const int ipx = (wv + NW * i) * 16 + li;
        int rr, cc;
        if (ipx < K::BODY) { rr = ipx / TW; cc = 1 + ipx % TW; }
        else if (ipx < K::HPX) { const int k = ipx - K::BODY, side = k >= K::HR ? 1 : 0; rr = k - side * K::HR; cc = side ? K::RW - 1 : 0; }
        else { rr = K::HR; cc = 0; }
        recoff[i] = fm4_rec(rr * K::RW + cc, g);
        prow[i] = rr < K::HR ? rr - 1 : -(1 << 20);
        pcol[i] = cc - 1;
      }
    }
    // ---------------------------------------------------------------- LN(y) -> operand fragments (as fm4_fwd_kernel)
    s16x8 xa[K::MTW][K::KS32 > 0 ? K::KS32 : 1];
    s16x8 xt[K::MTW];
    {
      int lane_o = lane_outer;
      asm volatile("" : "+v"(lane_o));
      const int lane = lane_o, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
#pragma unroll
      for (int i = 0; i < K::MTW; ++i) {
        const int mt = wv + NW * i < K::MT ? wv + NW * i : K::MT - 1;
        const bf16* sp_ = &S[(4 * g + qq) * K::PLANE + mt * 16 + 4 * pp];
        s16x4 lo[K::KS32 > 0 ? K::KS32 : 1], hi[K::KS32 > 0 ? K::KS32 : 1], tl = {0, 0, 0, 0}, dm = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks) {
          lo[ks] = tr_b16(sp_ + (ks * 32) * K::PLANE);
          hi[ks] = tr_b16(sp_ + (ks * 32 + 16) * K::PLANE);
        }
        if (K::KT16) tl = tr_b16(sp_ + (K::KS32 * 32) * K::PLANE);
        if constexpr (K::KS32 == 3) lds_wait(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], tl, dm);
        else if constexpr (K::KS32 == 2) lds_wait(lo[0], hi[0], lo[1], hi[1], tl, dm);
        else if constexpr (K::KS32 == 1) lds_wait(lo[0], hi[0], tl, dm);
        else lds_wait(tl);
        float v[K::NV];
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[8 * ks + j] = bf_s(lo[ks][j]); v[8 * ks + 4 + j] = bf_s(hi[ks][j]); }
        if (K::KT16)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[8 * K::KS32 + j] = bf_s(tl[j]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < K::NV; ++j) s += v[j];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float mu = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < K::NV; ++j) { const float d = v[j] - mu; q += d * d; }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        const float rstd = 1.0f / sqrtf(q * (1.0f / C) + 1e-5f);
        const float sub = a.with_bias ? mu : 0.f;         // BiasFree: x / sqrt(var + eps), x not centred (Restormer.py:37-39)
#pragma unroll
        for (int j = 0; j < K::NV; ++j) v[j] = (v[j] - sub) * rstd;
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks) {
          const u32x4 f = {pk_bf2(v[8 * ks], v[8 * ks + 1]), pk_bf2(v[8 * ks + 2], v[8 * ks + 3]), pk_bf2(v[8 * ks + 4], v[8 * ks + 5]),
                           pk_bf2(v[8 * ks + 6], v[8 * ks + 7])};
          xa[i][ks] = __builtin_bit_cast(s16x8, f);
        }
        const int Y = y0 + prow[i], X = x0 + pcol[i];
        const bool ok = Y >= 0 && Y < a.H && X >= 0 && X < a.W && wv + NW * i < K::MT;
        u32x4 ft = {0u, 0u, (g == 0 && ok) ? 0x3F80u : 0u, 0u};
        if (K::KT16) { ft[0] = pk_bf2(v[8 * K::KS32], v[8 * K::KS32 + 1]); ft[1] = pk_bf2(v[8 * K::KS32 + 2], v[8 * K::KS32 + 3]); }
        xt[i] = __builtin_bit_cast(s16x8, ft);
        if (!STAMP && a.mean && g == 0 && (unsigned)prow[i] < (unsigned)TH && (unsigned)pcol[i] < (unsigned)TW && wv + NW * i < K::MT) {
          const int64_t o = (int64_t)b * HW + (int64_t)Y * a.W + X;
          a.mean[o] = mu; a.rstd[o] = rstd;
        }
      }
    }
    FG4_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's weight pieces (requested a phase ago / at the start) have landed
    __syncthreads();                                      // the staged y is dead: the region becomes the two h0 group buffers
    FG4_STAMP(2);
    if (t < 4 * NCG)                                      // the ones record of every chunk buffer (the staging overwrote it)
      *reinterpret_cast<u32x4*>(H0 + (t >> 1) * K::H0S_BYTES + (K::HR * K::RW + 1) * K::REC + (t & 1) * 16) =
          (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    // GEMM1 of group gi: h0[chunk][pixel][16 ch] = W' LN(y) + b' inside the image, 0 outside, into h0 buffer gi & 1
    auto gemm1 = [&](int gi) {
      int lane_c = lane_outer;
      asm volatile("" : "+v"(lane_c));
      const unsigned char* const wf = W1B + (gi & 1) * K::W1G_BYTES + lane_c * 16;
      unsigned char* const hb = H0 + (gi & 1) * K::H0G_BYTES;
      s16x8 bw[K::NKS], bwn[K::NKS];
#pragma unroll
      for (int ks = 0; ks < K::NKS; ++ks) bw[ks] = *reinterpret_cast<const s16x8*>(wf + ks * K::FRAG);
#pragma unroll
      for (int cc = 0; cc < NCG; ++cc) {
        if (cc + 1 < NCG) {
#pragma unroll
          for (int ks = 0; ks < K::NKS; ++ks) bwn[ks] = *reinterpret_cast<const s16x8*>(wf + ((cc + 1) * K::NKS + ks) * K::FRAG);
        }
        unsigned char* const h = hb + cc * K::H0S_BYTES;
#pragma unroll
        for (int i = 0; i < K::MTW; ++i) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < K::KS32; ++ks) d = mfma32(bw[ks], xa[i][ks], d);
          d = mfma32(bw[K::KS32], xt[i], d);
          *reinterpret_cast<u32x2*>(h + recoff[i]) = (u32x2){pk_bf2(d[0], d[1]), pk_bf2(d[2], d[3])};
        }
        if (cc + 1 < NCG) {
#pragma unroll
          for (int ks = 0; ks < K::NKS; ++ks) bw[ks] = bwn[ks];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    gemm1(0);
    FG4_STAMP(3);
    __syncthreads();
    FG4_STAMP(4);

    f32x4 acc2[2][CT];                                    // this wave's output tile row: [pixel half][16-channel tile], lane (channel li, pixels 4g..4g+3)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc2[hf][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int gi = 0; gi < ngr; ++gi) {
      // weights requested a whole phase ahead: W_in of group gi + 2 into the buffer GEMM1(gi) just left, tap words / W_out of group
      // gi + 1 into the buffer the wave-local phase of gi - 1 just left (groups wrap around into the next tile)
      { const int g2 = gi + 2 >= ngr ? gi + 2 - ngr : gi + 2, g1 = gi + 1 >= ngr ? gi + 1 - ngr : gi + 1;
        dma_w1(g2, gi & 1); dma_tw2(g1, (gi + 1) & 1); }
      if (gi + 1 < ngr) gemm1(gi + 1);
      // ------------------------------------------------------------ wave-local: conv of both halves, gate, staging, GEMM2 k-step
      {
        int lane_c = lane_outer;
        asm volatile("" : "+v"(lane_c));
        const int lane = lane_c, li = lane & 15, g = lane >> 4;
        const unsigned char* const hb = H0 + (gi & 1) * K::H0G_BYTES;
        unsigned aA[5][2];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const int tp = 2 * s + (g >> 1);
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int r = tp < 9 ? (wv + tp / 3) * K::RW + li + 16 * hf + tp % 3 : K::HR * K::RW + 1;
            aA[s][hf] = (unsigned)(uintptr_t)hb + r * K::REC + (((g & 1) ^ ((r >> 2) & 1)) << 4);
          }
        }
        const unsigned wsel = 0xffffu << (16 * (li & 1));
        const int sel = (li & 7) >> 1;
        const unsigned m0 = sel == 0 ? wsel : 0u, m1 = sel == 1 ? wsel : 0u, m2 = sel == 2 ? wsel : 0u, m3 = sel == 3 ? wsel : 0u;
        unsigned char* const patch = PATCH + wv * K::PATCH_BYTES;
        const unsigned char* const tw2 = TW2B + (gi & 1) * K::TW2G_BYTES;
        const unsigned tbbase = (unsigned)(uintptr_t)(tw2 + lane * 2);
        const int s_ch = lane >> 2, s_cg = lane & 3;
        const int64_t row_off = (int64_t)(y0 + wv) * a.W + x0;
        u32x4 opA[2][10];
        unsigned opT[2][5];
#define FG4_RD(BUF, CC)                                                                                               \
        _Pragma("unroll")                                                                                             \
        for (int s_ = 0; s_ < 5; ++s_) {                                                                              \
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(opA[BUF][2 * s_]) : "v"(aA[s_][0]), "n"((CC) * K::H0S_BYTES));      \
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(opA[BUF][2 * s_ + 1]) : "v"(aA[s_][1]), "n"((CC) * K::H0S_BYTES));  \
          asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(opT[BUF][s_]) : "v"(tbbase), "n"(((CC) * 5 + s_) * 128));            \
        }
#define FG4_LANDED(BUF, N)                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                      \
                     : "+v"(opA[BUF][0]), "+v"(opA[BUF][1]), "+v"(opA[BUF][2]), "+v"(opA[BUF][3]), "+v"(opA[BUF][4]),       \
                       "+v"(opA[BUF][5]), "+v"(opA[BUF][6]), "+v"(opA[BUF][7]), "+v"(opA[BUF][8]), "+v"(opA[BUF][9]),       \
                       "+v"(opT[BUF][0]), "+v"(opT[BUF][1]), "+v"(opT[BUF][2]), "+v"(opT[BUF][3]), "+v"(opT[BUF][4]))
        // conv order inside a group: (pair chunk 0: first half, second half), (pair chunk 1: first half, second half); channel chunk
        // buffers are laid out [half][pair chunk]: chunk index cc = half * GP + pc
        FG4_RD(0, 0)
#pragma unroll
        for (int pc = 0; pc < GP; ++pc) {
          f32x4 y1a, y1b;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int step = 2 * pc + half, cur = step & 1;
            if (step + 1 < 2 * GP) {
              const int nh = (step + 1) & 1, npc = (step + 1) >> 1;
              if (cur == 0) { FG4_RD(1, nh * GP + npc) FG4_LANDED(0, 15); }
              else { FG4_RD(0, nh * GP + npc) FG4_LANDED(1, 15); }
            } else {
              if (cur == 0) FG4_LANDED(0, 0); else FG4_LANDED(1, 0);
            }
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
              const unsigned w = opT[cur][s] * 0x10001u;
              const u32x4 f = {w & m0, w & m1, w & m2, w & m3};
              const s16x8 tf = __builtin_bit_cast(s16x8, f);
              d0 = mfma32(__builtin_bit_cast(s16x8, opA[cur][2 * s]), tf, d0);
              d1 = mfma32(__builtin_bit_cast(s16x8, opA[cur][2 * s + 1]), tf, d1);
            }
            if (half == 0) { y1a = d0; y1b = d1; }
            else {                                        // lane (pair li, group g): pixels 4g..4g+3 and 16+4g..16+4g+3 of row wv
              u32x4 o;
              o[0] = pk_bf2(gelu_fwd<bf16>(y1a[0]) * d0[0], gelu_fwd<bf16>(y1a[1]) * d0[1]);
              o[1] = pk_bf2(gelu_fwd<bf16>(y1a[2]) * d0[2], gelu_fwd<bf16>(y1a[3]) * d0[3]);
              o[2] = pk_bf2(gelu_fwd<bf16>(y1b[0]) * d1[0], gelu_fwd<bf16>(y1b[1]) * d1[1]);
              o[3] = pk_bf2(gelu_fwd<bf16>(y1b[2]) * d1[2], gelu_fwd<bf16>(y1b[3]) * d1[3]);
              *reinterpret_cast<u32x2*>(patch + (16 * pc + li) * K::PS + g * 8) = (u32x2){o[0], o[1]};
              *reinterpret_cast<u32x2*>(patch + (16 * pc + li) * K::PS + 32 + g * 8) = (u32x2){o[2], o[3]};
            }
            if constexpr (SAVE) {                         // h0 of row wv (this chunk): records -> channel rows, 64-byte segments
              const int qq = li >> 2, pp = li & 3;
              const unsigned char* hr = hb + (half * GP + pc) * K::H0S_BYTES;
              const int rs = (wv + 1) * K::RW + 1 + 8 * g + qq;
              s16x4 u0 = tr_b16(hr + fm4_rec(rs, pp));
              s16x4 u1 = tr_b16(hr + fm4_rec(rs + 4, pp));
              lds_wait(u0, u1);
              const int jl = (gi * GP + pc) * 16 + li;
              if (jl < a.hidden)
                *reinterpret_cast<s16x8*>(a.h0s + ((int64_t)b * 2 * a.hidden + half * a.hidden + jl) * HW + row_off + 8 * g) = cat8(u0, u1);
            }
          }
        }
#undef FG4_RD
#undef FG4_LANDED
        wave_sync();
        if constexpr (SAVE) {                             // g of row wv: 16-byte row pieces out of the staging patch
#pragma unroll
          for (int pc = 0; pc < GP; ++pc) {
            const u32x2 p0 = *reinterpret_cast<const u32x2*>(patch + (16 * pc + s_ch) * K::PS + s_cg * 16);
            const u32x2 p1 = *reinterpret_cast<const u32x2*>(patch + (16 * pc + s_ch) * K::PS + s_cg * 16 + 8);
            const int jl = (gi * GP + pc) * 16 + s_ch;
            if (jl < a.hidden)
              *reinterpret_cast<u32x4*>(a.gs + ((int64_t)b * a.hidden + jl) * HW + row_off + 8 * s_cg) = (u32x4){p0[0], p0[1], p1[0], p1[1]};
          }
        }
        // GEMM2 k-step: acc2[px][c_out] += g^T [px][32 pairs] . W_out^T [32 pairs][c_out]; the A operand comes back transposed from the patch
        {
          const int qq = li >> 2, pp = li & 3;
          s16x4 lo[2], hi[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            lo[hf] = tr_b16(patch + (4 * g + qq) * K::PS + (16 * hf + 4 * pp) * 2);
            hi[hf] = tr_b16(patch + (16 + 4 * g + qq) * K::PS + (16 * hf + 4 * pp) * 2);
          }
          lds_wait(lo[0], hi[0], lo[1], hi[1]);
          const s16x8 ga0 = cat8(lo[0], hi[0]), ga1 = cat8(lo[1], hi[1]);
          const unsigned char* w2 = tw2 + K::TBG_BYTES + lane * 16;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const s16x8 wb = *reinterpret_cast<const s16x8*>(w2 + ct * K::FRAG);
            acc2[0][ct] = mfma32(ga0, wb, acc2[0][ct]);
            acc2[1][ct] = mfma32(ga1, wb, acc2[1][ct]);
          }
        }
        wave_sync();                                      // the patch is free for the next group
      }
      // this wave's weight pieces for the next phase have landed; the SAVE form's 3 GP stores of this phase, the youngest vector-memory
      // operations of the wave, stay in flight (vmcnt counts loads and stores together, in issue order)
      // (only when every pair chunk of the group has a live channel: a store whose lanes are all off may not be issued at all)
      if (SAVE && (gi * GP + GP - 1) * 16 < a.hidden) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FG4_STAMP(5);
      __syncthreads();
      FG4_STAMP(4);
    }

    // ---------------------------------------------------------------- epilogue: + bias + residual y, 64-byte row segments
    if (tile + 1 < t1) {                                  // the next tile's y leaves HBM behind the epilogue
      const int nt_ = tile + 1;
      fm_stage_load<K>(stg, yb, wv * 64 + lane_outer, (nt_ % a.tiles_x) * TW, (nt_ / a.tiles_x) * TH, a.H, a.W, HW);
    }
    {
      int lane_c = lane_outer;
      asm volatile("" : "+v"(lane_c));
      const int lane = lane_c, li = lane & 15, g = lane >> 4;
      float* const slab = reinterpret_cast<float*>(PATCH + wv * K::PATCH_BYTES);
      const int e_row = lane >> 2, e_col = (lane & 3) * 8;
      const int64_t pix = (int64_t)(y0 + wv) * a.W + x0 + e_col;
      u32x4 rr[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) rr[ct] = *reinterpret_cast<const u32x4*>(yb + (int64_t)(ct * 16 + e_row) * HW + pix);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float bv = a.b2[ct * 16 + e_row];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float v4[4] = {acc2[hf][ct][0], acc2[hf][ct][1], acc2[hf][ct][2], acc2[hf][ct][3]};
          Vec<float, 4>::st(&slab[li * K::SLAB_OS + 16 * hf + 4 * g], v4);
        }
        wave_sync();
        float v[8];
        Vec<float, 4>::ld(&slab[e_row * K::SLAB_OS + e_col], v);
        Vec<float, 4>::ld(&slab[e_row * K::SLAB_OS + e_col + 4], v + 4);
        u32x4 ov;
#pragma unroll
        for (int k = 0; k < 4; ++k) ov[k] = pk_bf2(v[2 * k] + bv + bf_lo(rr[ct][k]), v[2 * k + 1] + bv + bf_hi(rr[ct][k]));
        *reinterpret_cast<u32x4*>(a.out + ((int64_t)b * C + ct * 16 + e_row) * HW + pix) = ov;
        wave_sync();
      }
    }
  }
  if (STAMP && a.mean && lane_outer == 0) {
    float* o = a.mean + ((int64_t)blockIdx.x * NW + wv) * 8;
    for (int i = 0; i < 6; ++i) o[i] = (float)cyc[i];
    o[6] = (float)(fm_clock() - tk0); o[7] = (float)(t1 - t0);
  }
#undef FG4_STAMP
}

// ------------------------------------------------------------------------------------------------ host side
// Tile configurations.  Default: 4-wave workgroups on 32-pixel-wide tiles, two workgroups per CU - the two run out of
// phase, so one's HBM prologue / epilogue and MFMA phases overlap the other's VALU-bound conv phase (a single 8-wave
// workgroup per CU on 64-wide tiles serialises them: 1.5 ms instead of ... at C = 96, 256^2, bs 32).
// A/B switch (read per call): MI_FG_CFG=w64 selects the 8-wave 64-wide tiles; th8 / pc16 / pc32 override rows / pairs per chunk.
struct FgSel { int th, tw, pc, nw; };
static FgSel fg_select(int C, int H) {
  const char* e = MI_ENV(MI_FG_CFG);
  FgSel f;
  if (e && strstr(e, "w64")) {
    f.tw = 64; f.nw = 8;
    f.th = (C == 48 && !strstr(e, "th8") && H % 16 == 0) ? 16 : 8;
    f.pc = (C == 48 && f.th == 16) ? 16 : 32;
    if (strstr(e, "pc16")) f.pc = 16;
    return f;
  }
  f.tw = 32; f.nw = 4;
  f.th = (C == 48 && !(e && strstr(e, "th8")) && H % 16 == 0) ? 16 : 8;
  f.pc = 16;
  if (e && strstr(e, "pc32") && f.th == 8) f.pc = 32;
  return f;
}
enum FgKind { FG_NONE = 0, FG_C48, FG_C96 };
static FgKind fg_kind(const mi_gdfn_fused_shape* s) {
  if (!s || s->B <= 0 || s->hidden <= 0) return FG_NONE;
  if (s->W % 64 != 0 || s->H % 8 != 0) return FG_NONE;
  if (s->C == 48) return FG_C48;
  if (s->C == 96) return FG_C96;
  return FG_NONE;
}

static bool fg_use_v2() { const char* e = MI_ENV(MI_FG_CFG); return e && strstr(e, "v2"); }
static size_t fg_v2_pack_bytes(int C, int hidden) {
  const size_t a = fg_pack_layout(C, hidden, 16).bytes, b = fg_pack_layout(C, hidden, 32).bytes;
  return a > b ? a : b;
}
static int fg4_splits(int B, int tiles) {
  int S = 256 / B;                                        // one persistent 8-wave workgroup per CU
  if (S < 1) S = 1;
  if (S > tiles / 2) S = tiles / 2 > 0 ? tiles / 2 : 1;
  return S;
}
template <int C, bool SAVE>
static int fg4_launch(const mi_gdfn_fused_shape* s, const void* pack, const void* y, void* out, float* mean, float* rstd, void* h0s,
                      void* gs, hipStream_t st) {
  using K = Fg4Cfg<C, 2>;
  const Fg4PackLayout l = fg4_pack_layout<K>(s->hidden, fg_v2_pack_bytes(s->C, s->hidden));
  Fg4Args a;
  const unsigned char* pk = (const unsigned char*)pack;
  a.y = (const bf16*)y; a.out = (bf16*)out; a.mean = mean; a.rstd = rstd;
  a.grp = pk + l.grp; a.b2 = (const float*)(pk + l.b2);
  a.h0s = (bf16*)h0s; a.gs = (bf16*)gs;
  a.B = s->B; a.H = s->H; a.W = s->W; a.hidden = s->hidden; a.with_bias = s->ln_with_bias;
  a.tiles_x = s->W / 32; a.tiles_y = s->H / 8; a.ngr = l.ngr;
  a.S = fg4_splits(s->B, a.tiles_x * a.tiles_y);
  { const char* e = MI_ENV(MI_FG_DEBUG); a.dbg = e ? atoi(e) : 0; }
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  MI_CHECK_HIP(hipGetDevice(&dev));
  const unsigned bit = 1u << (dev & 31);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)fg4_fwd_kernel<C, 2, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES));
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)fg4_fwd_kernel<C, 2, SAVE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)K::LDS_BYTES));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const double N = (double)s->H * s->W * s->B, h = s->hidden;
  ProfScope ps(st, K_GDFN_FUSED_FWD, (2.0 * C + (SAVE ? 3.0 * h : 0.0)) * N * 2.0, 2.0 * N * (3.0 * C * h) + 2.0 * N * 9.0 * 2.0 * h);
  if (a.dbg & 0x1000)
    hipLaunchKernelGGL((fg4_fwd_kernel<C, 2, SAVE, true>), dim3((unsigned)(s->B * a.S)), dim3(64 * K::NW), K::LDS_BYTES, st, a);
  else
    hipLaunchKernelGGL((fg4_fwd_kernel<C, 2, SAVE>), dim3((unsigned)(s->B * a.S)), dim3(64 * K::NW), K::LDS_BYTES, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
template <int C>
static int fg4_pack(const mi_gdfn_fused_shape* s, const float* ln_w, const float* ln_b, const mi_gdfn_params* p, void* pack, hipStream_t st) {
  using K = Fg4Cfg<C, 2>;
  const Fg4PackLayout l = fg4_pack_layout<K>(s->hidden, fg_v2_pack_bytes(s->C, s->hidden));
  unsigned char* pk = (unsigned char*)pack;
  Fg4PackArgs a;
  a.ln_w = ln_w; a.ln_b = ln_b; a.in_w = p->in_w; a.in_b = p->in_b; a.dw_w = p->dw_w; a.dw_b = p->dw_b; a.out_w = p->out_w; a.out_b = p->out_b;
  a.grp = pk + l.grp; a.b2 = (float*)(pk + l.b2);
  a.C = s->C; a.h = s->hidden; a.ngr = l.ngr;
  hipLaunchKernelGGL((fg4_pack_kernel<K>), dim3(128), dim3(256), 0, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

template <int C, int TH, int TW, int PC, int NW, bool F8 = false, bool SAVE = false>
static int fg_launch(const mi_gdfn_fused_shape* s, const FgPackLayout& l, const void* pack, const void* y, void* out,
                     float* mean, float* rstd, hipStream_t st, const mi_f8_scales* f8 = nullptr, void* h0s = nullptr,
                     void* gs = nullptr) {
  using K = FgCfg<C, TH, TW, PC, NW>;
  FgArgs a;
  const unsigned char* pk = (const unsigned char*)pack;
  a.y = (const bf16*)y; a.out = (bf16*)out; a.mean = mean; a.rstd = rstd;
  a.w1p = (const bf16*)(pk + l.w1p); a.w2p = (const bf16*)(pk + l.w2p);
  a.wdp = (const float*)(pk + l.wdp); a.b2 = (const float*)(pk + l.b2);
  a.B = s->B; a.H = s->H; a.W = s->W; a.nch = l.nch; a.with_bias = s->ln_with_bias;
  a.tiles_x = s->W / TW; a.tiles_y = s->H / TH;
  a.f8_x1 = f8 ? f8->x1 : 1.f; a.f8_w1 = f8 ? f8->w1 : 1.f; a.f8_x2 = f8 ? f8->x2 : 1.f; a.f8_w2 = f8 ? f8->w2 : 1.f;
  a.h0s = (bf16*)h0s; a.gs = (bf16*)gs; a.hidden = s->hidden;
  { const char* e = MI_ENV(MI_FG_DEBUG); a.dbg = e ? atoi(e) : 0; }
  const int64_t tiles = (int64_t)s->B * a.tiles_x * a.tiles_y;
  MI_CHECK_ARG(tiles < (1ll << 31), "gdfn_fused: grid too large");
  a.xcd_pairs = (TW == 32 && tiles % 16 == 0 && !MI_ENV(MI_FG_NOXCD)) ? 1 : 0;
  MI_CHECK_HIP(hipFuncSetAttribute((const void*)fg_fwd_kernel<C, TH, TW, PC, NW, F8, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)K::LDS_BYTES));
  const double N = (double)s->H * s->W * s->B, h = s->hidden;
  ProfScope ps(st, K_GDFN_FUSED_FWD, (2.0 * C + (SAVE ? 3.0 * h : 0.0)) * N * 2.0, 2.0 * N * (3.0 * C * h) + 2.0 * N * 9.0 * 2.0 * h);
  hipLaunchKernelGGL((fg_fwd_kernel<C, TH, TW, PC, NW, F8, SAVE>), dim3((unsigned)tiles), dim3(64 * NW), K::LDS_BYTES, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_gdfn_fused_ok(const mi_gdfn_fused_shape* s) { return fg_kind(s) != FG_NONE ? 1 : 0; }

extern "C" size_t mi_gdfn_fused_pack_bytes(const mi_gdfn_fused_shape* s) {
  const FgKind k = fg_kind(s);
  if (k == FG_NONE) return 0;
  const size_t v2 = fg_v2_pack_bytes(s->C, s->hidden);
  return s->C == 48 ? fg4_pack_layout<Fg4Cfg<48, 2>>(s->hidden, v2).bytes : fg4_pack_layout<Fg4Cfg<96, 2>>(s->hidden, v2).bytes;
}

extern "C" int mi_gdfn_fused_pack(const mi_gdfn_fused_shape* s, const float* ln_w, const float* ln_b,
                                  const mi_gdfn_params* p, void* pack, void* stream) {
  const FgKind k = fg_kind(s);
  MI_CHECK_ARG(k != FG_NONE, "gdfn_fused_pack: shape not covered by the fused kernels (mi_gdfn_fused_ok)");
  MI_CHECK_ARG(ln_w && p && p->in_w && p->dw_w && p->out_w && pack, "gdfn_fused_pack: null pointer");
  MI_CHECK_ARG((s->ln_with_bias != 0) == (ln_b != nullptr), "gdfn_fused_pack: ln_with_bias does not match ln_b");
  MI_CHECK_ARG(aligned16(pack), "gdfn_fused_pack: pack buffer must be 16-byte aligned");
  const int PC = fg_select(s->C, s->H).pc;
  const FgPackLayout l = fg_pack_layout(s->C, s->hidden, PC);
  unsigned char* pk = (unsigned char*)pack;
  FgPackArgs a;
  a.ln_w = ln_w; a.ln_b = ln_b; a.in_w = p->in_w; a.in_b = p->in_b; a.dw_w = p->dw_w; a.dw_b = p->dw_b;
  a.out_w = p->out_w; a.out_b = p->out_b;
  a.w1p = (bf16*)(pk + l.w1p); a.w2p = (bf16*)(pk + l.w2p); a.wdp = (float*)(pk + l.wdp);
  a.b2 = (float*)(pk + l.b2);
  a.C = s->C; a.h = s->hidden; a.PC = PC; a.nch = l.nch;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_FUSED_PACK, (double)l.bytes, 0.0);
  hipLaunchKernelGGL(fg_pack_kernel, dim3(128), dim3(256), 0, st, a);
  MI_LAUNCH_CHECK();
  return s->C == 48 ? fg4_pack<48>(s, ln_w, ln_b, p, pack, st) : fg4_pack<96>(s, ln_w, ln_b, p, pack, st);   // the fourth form's sections
}

extern "C" int mi_gdfn_fused_fwd(const mi_gdfn_fused_shape* s, const void* pack, const void* y, void* out, float* mean,
                                 float* rstd, void* stream) {
  const FgKind k = fg_kind(s);
  MI_CHECK_ARG(k != FG_NONE, "gdfn_fused_fwd: shape not covered by the fused kernels (mi_gdfn_fused_ok)");
  MI_CHECK_ARG(pack && y && out, "gdfn_fused_fwd: null pointer");
  MI_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "gdfn_fused_fwd: mean and rstd go together");
  MI_CHECK_ARG(aligned16(pack) && aligned16(y) && aligned16(out), "gdfn_fused_fwd: pointers must be 16-byte aligned");
  const FgSel f = fg_select(s->C, s->H);
  const FgPackLayout l = fg_pack_layout(s->C, s->hidden, f.pc);
  hipStream_t st = (hipStream_t)stream;
  // fourth form (depthwise conv on the matrix cores): default at C = 48 (607 vs 666 us at 256^2, bs 32); at C = 96 its 8-wave
  // form spills (xa + output accumulators + operand pipeline > 256 registers) and loses to the form above: MI_FG_CFG=v4 forces it,
  // MI_FG_CFG=v2... keeps the forms above everywhere
  { const char* e = MI_ENV(MI_FG_CFG);
    const bool v4 = e && strstr(e, "v4");
    if (!fg_use_v2() && (s->C == 48 || v4))
      return s->C == 48 ? fg4_launch<48, false>(s, pack, y, out, mean, rstd, nullptr, nullptr, st)
                        : fg4_launch<96, false>(s, pack, y, out, mean, rstd, nullptr, nullptr, st); }
#define FG_CASE(CC, TH, TW, PC, NW) \
  if (s->C == CC && f.th == TH && f.tw == TW && f.pc == PC && f.nw == NW) \
    return fg_launch<CC, TH, TW, PC, NW>(s, l, pack, y, out, mean, rstd, st)
  FG_CASE(48, 16, 32, 16, 4);
  FG_CASE(48, 8, 32, 16, 4);
  FG_CASE(48, 8, 32, 32, 4);
  FG_CASE(96, 8, 32, 16, 4);
  FG_CASE(96, 8, 32, 32, 4);
  FG_CASE(48, 16, 64, 16, 8);
  FG_CASE(48, 8, 64, 32, 8);
  FG_CASE(48, 8, 64, 16, 8);
  FG_CASE(96, 8, 64, 32, 8);
  FG_CASE(96, 8, 64, 16, 8);
#undef FG_CASE
  set_error("gdfn_fused_fwd: no kernel for C=%d th=%d tw=%d pc=%d", s->C, f.th, f.tw, f.pc);
  return MI_ERR_ARG;
}

// Training form: the same launch also writes what the backward reads - the project_in output h0 [B][2h][H][W] (the depthwise
// conv is recomputed from it) and the gate output g [B][h][H][W] (operand of project_out's weight gradient) - so the forward of
// the half-block is one launch instead of GEMM -> depthwise gate -> GEMM: it reads y once and writes out, h0, g once
// (10 C planes per pixel with h = 2.66 C against the chain's 19).  Default tile forms only.
namespace mi {
int fused_gdfn_fwd_save(const mi_gdfn_fused_shape* s, const void* pack, const void* y, void* out, float* mean, float* rstd,
                        void* h0, void* g, hipStream_t st) {
  const FgKind k = fg_kind(s);
  MI_CHECK_ARG(k != FG_NONE, "gdfn_fused_fwd_train: shape not covered by the fused kernels (mi_gdfn_fused_ok)");
  MI_CHECK_ARG(pack && y && out && mean && rstd && h0 && g, "gdfn_fused_fwd_train: null pointer");
  MI_CHECK_ARG(aligned16(pack) && aligned16(y) && aligned16(out) && aligned16(h0) && aligned16(g),
               "gdfn_fused_fwd_train: pointers must be 16-byte aligned");
  // the fourth form's SAVE kernel is opt-in (MI_FG_CFG=v4): its stores ADD their time (918 vs 607 us at C = 48, 256^2) - every later
  // wait on a load (weight DMA, residual, prefetch) sits behind them in the wave's in-order vmcnt - so it does not beat the chain
  { const char* e = MI_ENV(MI_FG_CFG);
    if (e && strstr(e, "v4"))
      return s->C == 48 ? fg4_launch<48, true>(s, pack, y, out, mean, rstd, h0, g, st) : fg4_launch<96, true>(s, pack, y, out, mean, rstd, h0, g, st); }
  const FgSel f = fg_select(s->C, s->H);
  const FgPackLayout l = fg_pack_layout(s->C, s->hidden, f.pc);
#define FGS_CASE(CC, TH, TW, PC, NW) \
  if (s->C == CC && f.th == TH && f.tw == TW && f.pc == PC && f.nw == NW) \
    return fg_launch<CC, TH, TW, PC, NW, false, true>(s, l, pack, y, out, mean, rstd, st, nullptr, h0, g)
  FGS_CASE(48, 16, 32, 16, 4);
  FGS_CASE(48, 8, 32, 16, 4);
  FGS_CASE(96, 8, 32, 16, 4);
  FGS_CASE(48, 16, 64, 16, 8);
  FGS_CASE(96, 8, 64, 32, 8);
#undef FGS_CASE
  set_error("gdfn_fused_fwd_train: no saving kernel for C=%d th=%d tw=%d pc=%d", s->C, f.th, f.tw, f.pc);
  return MI_ERR_ARG;
}
}  // namespace mi

// The same launch with fp8 (e4m3) MFMA operands in both projections (inference; default tile forms only).  f8->x1 scales the
// NORMALISED input ((y - mu) rstd, |.| <= sqrt(C)) and f8->w1 the packed W_in . diag(gamma); x2 / w2 as in mi_gdfn_fwd_f8.
extern "C" int mi_gdfn_fused_fwd_f8(const mi_gdfn_fused_shape* s, const void* pack, const mi_f8_scales* f8, const void* y, void* out,
                                    void* stream) {
  const FgKind k = fg_kind(s);
  MI_CHECK_ARG(k != FG_NONE, "gdfn_fused_fwd_f8: shape not covered by the fused kernels (mi_gdfn_fused_ok)");
  MI_CHECK_ARG(pack && y && out && f8, "gdfn_fused_fwd_f8: null pointer");
  MI_CHECK_ARG(f8->x1 > 0.f && f8->w1 > 0.f && f8->x2 > 0.f && f8->w2 > 0.f, "gdfn_fused_fwd_f8: fp8 scales must be positive");
  MI_CHECK_ARG(aligned16(pack) && aligned16(y) && aligned16(out), "gdfn_fused_fwd_f8: pointers must be 16-byte aligned");
  const FgSel f = fg_select(s->C, s->H);
  const FgPackLayout l = fg_pack_layout(s->C, s->hidden, f.pc);
  hipStream_t st = (hipStream_t)stream;
#define FG8_CASE(CC, TH, TW, PC, NW) \
  if (s->C == CC && f.th == TH && f.tw == TW && f.pc == PC && f.nw == NW) \
    return fg_launch<CC, TH, TW, PC, NW, true>(s, l, pack, y, out, nullptr, nullptr, st, f8)
  FG8_CASE(48, 16, 32, 16, 4);
  FG8_CASE(48, 8, 32, 16, 4);
  FG8_CASE(96, 8, 32, 16, 4);
#undef FG8_CASE
  set_error("gdfn_fused_fwd_f8: fp8 operands are built for the default tile forms only (C=%d th=%d tw=%d pc=%d)", s->C, f.th, f.tw, f.pc);
  return MI_ERR_ARG;
}
