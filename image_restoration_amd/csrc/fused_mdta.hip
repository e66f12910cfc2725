// Fused MDTA, pass A:   x -> LayerNorm -> qkv 1x1 -> depthwise 3x3 -> { q k^T partials, row sums of squares of q and k, v }
// (Restormer.py:111-122 Attention.forward up to the normalised q k^T; moce_ir.py:300-312; AdaIR-main/net/model.py:113-122)
// in ONE launch: x is read once (plus a one-pixel halo) and ONLY v is written.  qkv0 (the 1x1 output), q and k never reach HBM:
// the unfused chain moves 12 C planes per pixel for these steps (C + 3C for the GEMM, 3C + 3C for the conv, 2C for the Gram),
// this kernel 2 C.  Pass B stays what it was: out = (W_o . blockdiag(softmax)) . v + x, one per-image-weight 1x1 GEMM.
//
// Work decomposition (bf16 activations, fp32 accumulate; same skeleton as fused_gdfn.hip):
//   * a 4-wave workgroup walks TH x 32 pixel tiles of ONE image (a contiguous range of the image's tiles: grid = B x S) and
//     keeps, across all of its tiles, the fp32 accumulators of its share of the C x c Gram blocks (a 16 x 16 block (rb, cb) of
//     head h belongs to wave cb mod 4) and of the diagonal blocks of q q^T / k k^T, whose diagonals are the row sums of squares
//     F.normalize needs.  One partial per workgroup is written at the end; a small kernel sums the S partials of an image in
//     a fixed order (deterministic).
//   * per tile: LN(x) on tile + halo as MFMA A-operand fragments in registers (LayerNorm's affine is folded into the packed
//     W_qkv' = W_qkv diag(gamma), b' = W_qkv beta + b), then the 3C output channels in chunks of 16, k first, then q, then v:
//       GEMM1   h0[16][tile+halo] = W'[chunk] . LN(x)            MFMA, result -> LDS (bf16, zero outside the image: the conv's padding)
//       conv    dw3x3 on the VALU, packed fp32 (channels r and r+8 of the chunk in the two halves), 8 pixels of a row per lane
//                 k chunk -> LDS K image (resident for the tile) ; q chunk -> LDS Q image ; v chunk -> HBM
//       Gram    (q chunks) acc[rb][cb] += Q_rb . K_cb^T over the tile's pixels       MFMA 16x16x32, k index = pixel
//   * the K / Q images are FRAGMENT-MAJOR: one MFMA operand fragment (16 channels x 32 pixels = one tile row) is a contiguous
//     KiB, lane l's 16 bytes (8 pixels of channel l&15, pixel octet l>>4) at slot (l>>4)*16 + ((l&15) ^ swz): a fragment read is
//     one conflict-free ds_read_b128, and the conv lanes (8 pixels of a row each) store exactly such 16-byte pieces.
#include <stdlib.h>

#include <atomic>

#include "fused_common.h"
#include "internal.h"

namespace mi {
using namespace fz;

template <int C_, int HEADS_, int TH_, int NW_> struct FmCfg {
  static constexpr int C = C_, HEADS = HEADS_, TH = TH_, TW = 32, NW = NW_;
  static constexpr int c = C / HEADS;
  static constexpr int NT = 64 * NW;
  static constexpr int HR = TH + 2;
  static constexpr int BODY = HR * TW;
  static constexpr int HPX = BODY + 2 * HR;
  static constexpr int HPXP = (HPX + 15) / 16 * 16;
  static constexpr int MT = HPXP / 16;
  static constexpr int MTW = (MT + NW - 1) / NW;
  static constexpr int PLANE = (HPXP % 16 == 8) ? HPXP : HPXP + 8;
  static constexpr int KS32 = C / 32, KT16 = (C % 32) / 16;
  static constexpr int NV = 8 * KS32 + 4 * KT16;
  static constexpr int W1S = C + 8;
  static constexpr int CH = 16;                          // channels per chunk
  static constexpr int PC = 8;                           // "pairs" per chunk: channels r and r + 8 ride in one packed lane
  static constexpr int NB = C / 16;                      // 16-row blocks of q (= of k, = of v)
  static constexpr int NCHUNK = 3 * NB;                  // chunk order: k blocks, q blocks, v blocks
  static constexpr int CB = c / 16;                      // column (k) blocks per head
  static constexpr int NCBW = (CB + NW - 1) / NW;        // column blocks of a head owned by one wave
  static constexpr int NQD = (NB + NW - 1) / NW;         // q diagonal blocks per wave
  static constexpr int VPR = TW / 8;
  static constexpr int CG = TW / 8;                      // conv: lanes per tile row
  static constexpr int RPP = 64 / CG;
  static constexpr int ROWS = TH < RPP ? TH : RPP;
  static constexpr int NPAIR = RPP / ROWS;
  static constexpr int PASSES = TH / ROWS;
  static constexpr int PPW = PC / NW;
  static constexpr int FRAG = 1024;                      // bytes of one operand fragment (16 channels x 32 pixels)
  static constexpr int H0_BYTES = CH * PLANE * 2;
  static constexpr int KT_BYTES = NB * TH * FRAG;
  static constexpr int QC_BYTES = TH * FRAG;
  static constexpr int W1_BYTES = CH * W1S * 2;
  static constexpr int WD_BYTES = PC * 20 * 4;           // taps + bias of a chunk (two buffers)
  static constexpr int S_BYTES = C * PLANE * 2;          // prologue: raw x staged plane-major (aliases everything)
  static constexpr int MAIN_BYTES = H0_BYTES + KT_BYTES + QC_BYTES + W1_BYTES + 2 * WD_BYTES;
  static constexpr int LDS_BYTES = MAIN_BYTES > S_BYTES ? MAIN_BYTES : S_BYTES;
  static_assert(C % 16 == 0 && c % 16 == 0 && NW == 4 && TH % ROWS == 0 && PPW % NPAIR == 0 && PC % NW == 0, "unsupported tile");
  static_assert(4 * MTW <= 64, "validity mask");
  static_assert(H0_BYTES % 16 == 0 && W1_BYTES % 16 == 0, "LDS carve alignment");
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

struct FmArgs {
  const bf16* x; bf16* v; float* part;                  // part: [B][S][HEADS*c*c + 2C] fp32
  float* mean; float* rstd;                              // optional LayerNorm statistics [B][H*W]
  const bf16* w1p; const float* wdp;
  int B, H, W, with_bias, tiles_x, tiles_y, S, dbg;
};

// packed-weight blob: W1p [NCHUNK][16][C + 8] bf16 (fp32 bias in the row padding), WDp [NCHUNK][8][10][2] fp32
struct FmPackLayout { size_t w1p, wdp, bytes; int nchunk; };
static FmPackLayout fm_pack_layout(int C) {
  FmPackLayout l;
  l.nchunk = 3 * C / 16;
  size_t off = 0;
  l.w1p = off; off = align_up(off + (size_t)l.nchunk * 16 * (C + 8) * 2, 256);
  l.wdp = off; off = align_up(off + (size_t)l.nchunk * 8 * 20 * 4, 256);
  l.bytes = off;
  return l;
}
// channel of the qkv output that row r of chunk ci holds: chunks walk k (C..2C), then q (0..C), then v (2C..3C)
__host__ __device__ static inline int fm_channel(int C, int ci, int r) {
  const int nb = C / 16;
  if (ci < nb) return C + 16 * ci + r;
  if (ci < 2 * nb) return 16 * (ci - nb) + r;
  return 2 * C + 16 * (ci - 2 * nb) + r;
}

struct FmPackArgs {
  const float *ln_w, *ln_b, *qkv_w, *qkv_b, *dw_w, *dw_b;
  bf16* w1p; float* wdp;
  int C, nchunk;
};
__global__ __launch_bounds__(256) void fm_pack_kernel(FmPackArgs a) {
  const int C = a.C, W1S = C + 8;
  const int64_t n_w1 = (int64_t)a.nchunk * 16 * W1S, n_wd = (int64_t)a.nchunk * 8 * 20;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_w1 + n_wd; e += (int64_t)gridDim.x * 256) {
    if (e < n_w1) {
      const int k = (int)(e % W1S);
      const int64_t q = e / W1S;
      const int r = (int)(q % 16), ci = (int)(q / 16);
      const int ch = fm_channel(C, ci, r);
      if (k < C) {
        a.w1p[e] = (bf16)(a.qkv_w[(int64_t)ch * C + k] * a.ln_w[k]);
      } else if (k == C) {                             // the row's fp32 bias b' = b + W . beta rides in the row padding
        float sb = a.qkv_b ? a.qkv_b[ch] : 0.f;
        if (a.ln_b)
          for (int kk = 0; kk < C; ++kk) sb += a.qkv_w[(int64_t)ch * C + kk] * a.ln_b[kk];
        *reinterpret_cast<float*>(&a.w1p[e]) = sb;
      } else if (k >= C + 2) {
        a.w1p[e] = (bf16)0.f;
      }
      continue;
    }
    const int64_t r0 = e - n_w1;                       // [chunk][pair][tap 0..8, bias][half]
    const int half = (int)(r0 % 2);
    int64_t q = r0 / 2;
    const int tp = (int)(q % 10); q /= 10;
    const int p = (int)(q % 8), ci = (int)(q / 8);
    const int ch = fm_channel(C, ci, p + 8 * half);
    a.wdp[r0] = tp < 9 ? a.dw_w[(int64_t)ch * 9 + tp] : (a.dw_b ? a.dw_b[ch] : 0.f);
  }
}

// slot of channel row `chrow` (0..15) inside the 256-byte octet block of a fragment: XOR on the two upper row bits, keyed by
// the tile row and the octet so that (a) a fragment read (fixed tile row; the 16-lane read groups mix octets o and o^1) stays a
// permutation of the 16 slots and (b) the four octets of one conv store group land on four different slots
__device__ __forceinline__ int fm_slot(int chrow, int trow, int octet) {
  const int g = (0x9C >> (2 * octet)) & 3;             // g = {0, 3, 1, 2}: g(o) ^ g(o ^ 1) = 3
  return chrow ^ (4 * ((trow & 3) ^ g));
}
__device__ __forceinline__ s16x8 fm_ldfrag(const unsigned char* base, int trow, int lane) {
  const int row = lane & 15, oct = lane >> 4;
  return *reinterpret_cast<const s16x8*>(base + trow * 1024 + (oct * 16 + fm_slot(row, trow, oct)) * 16);
}

template <int C, int HEADS, int TH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void fm_fwd_kernel(FmArgs a) {
  using K = FmCfg<C, HEADS, TH, NW>;
  constexpr int NT = K::NT, TW = K::TW;
  extern __shared__ __attribute__((aligned(16))) unsigned char fm_lds[];
  bf16* const H0 = reinterpret_cast<bf16*>(fm_lds);
  unsigned char* const KT = fm_lds + K::H0_BYTES;
  unsigned char* const QC = KT + K::KT_BYTES;
  bf16* const W1 = reinterpret_cast<bf16*>(QC + K::QC_BYTES);
  float* const WD = reinterpret_cast<float*>(QC + K::QC_BYTES + K::W1_BYTES);     // [2][PC][20]
  bf16* const S = reinterpret_cast<bf16*>(fm_lds);
  const int t = threadIdx.x, lane_outer = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x / a.S, sp = blockIdx.x - b * a.S;
  const int tiles = a.tiles_x * a.tiles_y;
  const int t0 = (int)((int64_t)sp * tiles / a.S), t1 = (int)((int64_t)(sp + 1) * tiles / a.S);
  const int64_t HW = (int64_t)a.H * a.W;
  const bf16* const xb = a.x + (int64_t)b * C * HW;

  f32x4 acc[K::NB][K::NCBW], kd[HEADS][K::NCBW], qd[K::NQD];
#pragma unroll
  for (int i = 0; i < K::NB; ++i)
#pragma unroll
    for (int j = 0; j < K::NCBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < HEADS; ++i)
#pragma unroll
    for (int j = 0; j < K::NCBW; ++j) kd[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < K::NQD; ++i) qd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = t0; tile < t1; ++tile) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    int lane_o = lane_outer;
    asm volatile("" : "+v"(lane_o));                    // per-tile lane coordinates: nothing derived from them is hoisted
    const int lane = lane_o, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
    const int tt = wv * 64 + lane;
    // ---------------------------------------------------------------- stage raw x (tile + halo), plane-major
    {
      constexpr int NBV = C * K::HR * K::VPR;           // 16-byte vectors of the tile body
      constexpr int NBN = (NBV + NT - 1) / NT;
      constexpr int NE = K::HPXP - K::BODY;             // halo-column pixels + padding per plane
      constexpr int NEN = (C * NE + NT - 1) / NT;
      u32x4 raw[NBN];
      u16 rawe[NEN];
#pragma unroll
      for (int n = 0; n < NBN; ++n) {
        const int idx = tt + NT * n;
        const int cc = idx / (K::HR * K::VPR), rem = idx - cc * (K::HR * K::VPR), r = rem / K::VPR, u = rem % K::VPR;
        const int Y = y0 - 1 + r;
        raw[n] = (u32x4){0u, 0u, 0u, 0u};
        if (idx < NBV && Y >= 0 && Y < a.H)
          raw[n] = *reinterpret_cast<const u32x4*>(xb + (int64_t)cc * HW + (int64_t)Y * a.W + x0 + 8 * u);
      }
#pragma unroll
      for (int n = 0; n < NEN; ++n) {
        const int idx = tt + NT * n;
        const int cc = idx / NE, k = idx - cc * NE;
        rawe[n] = 0;
        if (idx < C * NE && k < 2 * K::HR) {
          const int side = k >= K::HR ? 1 : 0, r = k - side * K::HR;
          const int Y = y0 - 1 + r, X = side ? x0 + TW : x0 - 1;
          if (Y >= 0 && Y < a.H && X >= 0 && X < a.W)
            rawe[n] = reinterpret_cast<const u16*>(xb)[(int64_t)cc * HW + (int64_t)Y * a.W + X];
        }
      }
#pragma unroll
      for (int n = 0; n < NBN; ++n) {
        const int idx = tt + NT * n;
        const int cc = idx / (K::HR * K::VPR), rem = idx - cc * (K::HR * K::VPR), r = rem / K::VPR, u = rem % K::VPR;
        if (idx < NBV) *reinterpret_cast<u32x4*>(&S[cc * K::PLANE + r * TW + 8 * u]) = raw[n];
      }
#pragma unroll
      for (int n = 0; n < NEN; ++n) {
        const int idx = tt + NT * n;
        const int cc = idx / NE, k = idx - cc * NE;
        if (idx < C * NE) reinterpret_cast<u16*>(S)[cc * K::PLANE + K::BODY + k] = rawe[n];
      }
    }
    __syncthreads();

    // ---------------------------------------------------------------- LN(x) -> A-operand fragments (registers)
    // element order of a 32-k fragment (same for A and B): j < 4 is k = 4g + j, j >= 4 is k = 16 + 4g + (j - 4)
    s16x8 xa[K::MTW][K::KS32 > 0 ? K::KS32 : 1];
    s16x8 xt[K::MTW];
    unsigned long long vmask = 0;
#pragma unroll
    for (int i = 0; i < K::MTW; ++i) {
      const int mt = wv + NW * i;
      if (mt < K::MT) {
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
        const float sub = a.with_bias ? mu : 0.f;       // BiasFree: x / sqrt(var + eps), x not centred (Restormer.py:37-39)
#pragma unroll
        for (int j = 0; j < K::NV; ++j) v[j] = (v[j] - sub) * rstd;
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) xa[i][ks][j] = bf_bits(v[8 * ks + j]);
        xt[i] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (K::KT16)
#pragma unroll
          for (int j = 0; j < 4; ++j) xt[i][j] = bf_bits(v[8 * K::KS32 + j]);
        if (a.mean && g == 0) {                         // statistics of the tile's own pixels
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
        xt[i] = (s16x8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    __syncthreads();                                    // the staged x is dead: the region becomes h0 / K / Q / weights

    constexpr int W1V = K::W1_BYTES / 16, WDV = K::WD_BYTES / 16;
    constexpr int W1N = (W1V + NT - 1) / NT;
    static_assert(WDV <= NT, "one vector per thread");
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(a.w1p);
      for (int vv = tt; vv < W1V; vv += NT) reinterpret_cast<u32x4*>(W1)[vv] = src[vv];
      if (tt < WDV) reinterpret_cast<u32x4*>(WD)[tt] = reinterpret_cast<const u32x4*>(a.wdp)[tt];
    }
    __syncthreads();

#pragma unroll 1
    for (int ci = 0; ci < K::NCHUNK; ++ci) {
      int lane_c = lane_outer;
      asm volatile("" : "+v"(lane_c));
      const int lane = lane_c, li = lane & 15, g = lane >> 4, tt = wv * 64 + lane;
      // ------------------------------------------------------------ GEMM1: h0 chunk = W'[chunk] . LN(x), to LDS
      if (!(a.dbg & 32)) {
        const bf16* wr = &W1[li * K::W1S + 4 * g];
        s16x8 bw[K::KS32 > 0 ? K::KS32 : 1];
        s16x8 bt = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks)
          bw[ks] = cat8(*reinterpret_cast<const s16x4*>(wr + ks * 32), *reinterpret_cast<const s16x4*>(wr + ks * 32 + 16));
        if (K::KT16) bt = cat8(*reinterpret_cast<const s16x4*>(wr + K::KS32 * 32), (s16x4){0, 0, 0, 0});
        const float bias = *reinterpret_cast<const float*>(&W1[li * K::W1S + C]);
        bf16* hrow = &H0[li * K::PLANE + 4 * g];
#pragma unroll
        for (int i = 0; i < K::MTW; ++i) {
          const int mt = wv + NW * i;
          if (mt < K::MT) {
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < K::KS32; ++ks) d = mfma32(xa[i][ks], bw[ks], d);
            if (K::KT16) d = mfma32(xt[i], bt, d);      // 16-deep tail in a zero-padded 32-deep MFMA (one opcode per chain)
            const unsigned m = (unsigned)(vmask >> (4 * i));
            u32x2 o;
            o[0] = pack_bf2((m & 1u) ? d[0] + bias : 0.f, (m & 2u) ? d[1] + bias : 0.f);
            o[1] = pack_bf2((m & 4u) ? d[2] + bias : 0.f, (m & 8u) ? d[3] + bias : 0.f);
            *reinterpret_cast<u32x2*>(hrow + mt * 16) = o;
          }
        }
      }
      __syncthreads();

      // ------------------------------------------------------------ depthwise 3x3 (VALU); next chunk's weights in flight
      {
        u32x4 wr1[W1N], wrd = {0u, 0u, 0u, 0u};
        const bool more = ci + 1 < K::NCHUNK;
        const float* const wdc = WD + (ci & 1) * (K::PC * 20);
        {
          if (more && tt < WDV) wrd = reinterpret_cast<const u32x4*>(a.wdp + (int64_t)(ci + 1) * K::PC * 20)[tt];
          const u32x4* s1 = reinterpret_cast<const u32x4*>(a.w1p + (int64_t)(ci + 1) * K::CH * K::W1S);
#pragma unroll
          for (int n = 0; n < W1N; ++n) { const int vv = tt + NT * n; if (more && vv < W1V) wr1[n] = s1[vv]; }
        }
        const int kind = ci < K::NB ? 0 : (ci < 2 * K::NB ? 1 : 2);            // 0: k, 1: q, 2: v
        const int blk = ci - kind * K::NB;                                       // 16-row block of its kind
        unsigned char* const img = kind == 0 ? KT + blk * (TH * K::FRAG) : QC;
        const int cg = lane % K::CG, rl = (lane / K::CG) % K::ROWS, psel = lane / (K::CG * K::ROWS);
#pragma unroll 1
        for (int s = 0; s < ((a.dbg & 64) ? 0 : K::PPW / K::NPAIR); ++s) {
          const int p = wv * K::PPW + s * K::NPAIR + psel;
          const float* wp = wdc + p * 20;
          f32x2 w[10];
#pragma unroll
          for (int i = 0; i < 10; ++i) {
            if constexpr (K::NPAIR == 1) {
              w[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wp[2 * i])));
              w[i][1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wp[2 * i + 1])));
            } else {
              w[i] = *reinterpret_cast<const f32x2*>(wp + 2 * i);
            }
          }
          const bf16* h1 = &H0[p * K::PLANE];
          const bf16* h2 = &H0[(K::PC + p) * K::PLANE];
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
            u32x4 oa, ob;                               // channel p (half 0) and channel p + 8 (half 1), 8 pixels each
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              oa[k] = pack_bf2(o[2 * k][0], o[2 * k + 1][0]);
              ob[k] = pack_bf2(o[2 * k][1], o[2 * k + 1][1]);
            }
            if (kind == 2) {
              bf16* vp = a.v + ((int64_t)b * C + 16 * blk + p) * HW + (int64_t)(y0 + row) * a.W + x0 + 8 * cg;
              *reinterpret_cast<u32x4*>(vp) = oa;
              *reinterpret_cast<u32x4*>(vp + 8 * HW) = ob;
            } else {
              unsigned char* fr = img + row * K::FRAG + cg * 256;
              *reinterpret_cast<u32x4*>(fr + fm_slot(p, row, cg) * 16) = oa;
              *reinterpret_cast<u32x4*>(fr + fm_slot(p + 8, row, cg) * 16) = ob;
            }
          }
        }
#pragma unroll
        for (int n = 0; n < W1N; ++n) { const int vv = tt + NT * n; if (more && vv < W1V) reinterpret_cast<u32x4*>(W1)[vv] = wr1[n]; }
        if (more && tt < WDV) reinterpret_cast<u32x4*>(WD + ((ci + 1) & 1) * (K::PC * 20))[tt] = wrd;
      }
      __syncthreads();
      if ((a.dbg & 15) == 2 && ci == (a.dbg >> 8)) {      // debug: image of chunk (dbg >> 8) as [16][TH*32] bf16 into v of tile 0
        if (blockIdx.x == 0 && tile == t0) {
          const unsigned char* img = ci < K::NB ? KT + ci * (TH * K::FRAG) : QC;
          for (int e = tt; e < 16 * TH * 32; e += NT) {
            const int r = e / (TH * 32), px = e % (TH * 32), trow = px / 32, oct = (px % 32) / 8, w8 = px % 8;
            a.v[e] = reinterpret_cast<const bf16*>(img + trow * K::FRAG + (oct * 16 + fm_slot(r, trow, oct)) * 16)[w8];
          }
        }
        return;
      }

      // ------------------------------------------------------------ Gram: acc[rb][.] += Q_rb . K_cb^T over the tile's pixels
      if (ci >= K::NB && ci < 2 * K::NB && !(a.dbg & 128)) {
        const int rb = ci - K::NB;
#define FM_GRAM(RB)                                                                                              \
        case RB: {                                                                                               \
          if constexpr (RB < K::NB) {                                                                            \
            constexpr int head = RB / K::CB;                                                                     \
            constexpr bool first = (RB % K::CB) == 0;                                                            \
            const bool qdiag = wv == (NW - 1 - RB % NW);                                                         \
            _Pragma("unroll")                                                                                    \
            for (int ks = 0; ks < TH; ++ks) {                                                                    \
              const s16x8 af = fm_ldfrag(QC, ks, lane);                                                          \
              _Pragma("unroll")                                                                                  \
              for (int sl = 0; sl < K::NCBW; ++sl) {                                                             \
                const int cb = wv + NW * sl;                                                                     \
                if (cb < K::CB) {                                                                                \
                  const s16x8 bf_ = fm_ldfrag(KT + (head * K::CB + cb) * (TH * K::FRAG), ks, lane);             \
                  acc[RB][sl] = mfma32(af, bf_, acc[RB][sl]);                                                    \
                  if (first) kd[head][sl] = mfma32(bf_, bf_, kd[head][sl]);                                      \
                }                                                                                                \
              }                                                                                                  \
              if (qdiag) qd[RB / NW] = mfma32(af, af, qd[RB / NW]);                                              \
            }                                                                                                    \
          }                                                                                                      \
        } break;
        switch (rb) {
          FM_GRAM(0) FM_GRAM(1) FM_GRAM(2) FM_GRAM(3) FM_GRAM(4) FM_GRAM(5)
          default: break;
        }
#undef FM_GRAM
      }
    }
  }

  // ------------------------------------------------------------------ this workgroup's partial: G blocks, then sums of squares
  {
    const int lane = lane_outer, li = lane & 15, g = lane >> 4;
    float* pz = a.part + (int64_t)blockIdx.x * (HEADS * K::c * K::c + 2 * C);
#pragma unroll
    for (int rb = 0; rb < K::NB; ++rb) {
      const int head = rb / K::CB, ib = rb % K::CB;
#pragma unroll
      for (int sl = 0; sl < K::NCBW; ++sl) {
        const int cb = wv + NW * sl;
        if (cb < K::CB) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pz[head * K::c * K::c + (ib * 16 + 4 * g + r) * K::c + cb * 16 + li] = acc[rb][sl][r];
        }
      }
      if (wv == (NW - 1 - rb % NW) && (li >> 2) == g) {
        const f32x4 d = qd[rb / NW];
        pz[HEADS * K::c * K::c + rb * 16 + li] = (li & 3) == 0 ? d[0] : ((li & 3) == 1 ? d[1] : ((li & 3) == 2 ? d[2] : d[3]));
      }
    }
#pragma unroll
    for (int head = 0; head < HEADS; ++head)
#pragma unroll
      for (int sl = 0; sl < K::NCBW; ++sl) {
        const int cb = wv + NW * sl;
        if (cb < K::CB && (li >> 2) == g) {
          const f32x4 d = kd[head][sl];
          pz[HEADS * K::c * K::c + C + head * K::c + cb * 16 + li] =
              (li & 3) == 0 ? d[0] : ((li & 3) == 1 ? d[1] : ((li & 3) == 2 ? d[2] : d[3]));
        }
      }
  }
}

// graw[z][c*c] and ss[z][2c] (q sums of squares, then k) from the S partials of every image, summed in a fixed order
__global__ __launch_bounds__(256) void fm_reduce_kernel(const float* __restrict__ part, float* __restrict__ graw, float* __restrict__ ss,
                                                        int S, int C, int heads) {
  const int c = C / heads, z = blockIdx.x, b = z / heads, head = z - b * heads;
  const int64_t pstride = (int64_t)heads * c * c + 2 * C;
  const float* p0 = part + (int64_t)b * S * pstride;
  for (int e = threadIdx.x; e < c * c + 2 * c; e += 256) {
    int64_t off;
    if (e < c * c) off = (int64_t)head * c * c + e;
    else if (e < c * c + c) off = (int64_t)heads * c * c + head * c + (e - c * c);
    else off = (int64_t)heads * c * c + C + head * c + (e - c * c - c);
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += p0[k * pstride + off];
    if (e < c * c) graw[(int64_t)z * c * c + e] = s;
    else ss[(int64_t)z * 2 * c + (e - c * c)] = s;
  }
}

// ------------------------------------------------------------------------------------------------ host side
enum FmKind { FM_NONE = 0, FM_48_1, FM_96_2, FM_96_1 };
static FmKind fm_kind(const mi_mdta_shape* s) {
  if (!s || s->B <= 0 || s->dtype != MI_BF16 || s->ks != 3) return FM_NONE;
  if (s->W % 64 != 0 || s->H % 8 != 0) return FM_NONE;
  if (s->C == 48 && s->heads == 1) return FM_48_1;
  if (s->C == 96 && s->heads == 2) return FM_96_2;
  if (s->C == 96 && s->heads == 1) return FM_96_1;
  return FM_NONE;
}
static int fm_splits(const mi_mdta_shape* s, int TH) {
  const int tiles = (s->H / TH) * (s->W / 32);
  int S = 512 / s->B;                                   // two 4-wave workgroups per CU
  if (S < 1) S = 1;
  if (S > 32) S = 32;
  if (S > tiles) S = tiles;
  return S;
}

struct FmWs { void* v; float* part; float* graw; float* ss; float* P; float* A; float* nrm; float* M; void* pw_ws; size_t bytes; };
static FmWs fm_ws_layout(const mi_mdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B, c = C / s->heads, Z = B * s->heads;
  Carver cv(base);
  FmWs w;
  const int S = fm_splits(s, 8);
  w.part = cv.take<float>(B * S * (s->heads * c * c + 2 * C) * sizeof(float));
  w.graw = cv.take<float>(Z * c * c * sizeof(float));
  w.ss = cv.take<float>(Z * 2 * c * sizeof(float));
  w.P = cv.take<float>(Z * c * c * sizeof(float));
  w.A = cv.take<float>(Z * c * c * sizeof(float));
  w.nrm = cv.take<float>(Z * 2 * c * sizeof(float));
  w.M = cv.take<float>(B * C * C * sizeof(float));
  {
    mi_pw_desc d;
    memset(&d, 0, sizeof(d));
    d.x1 = (void*)256; d.x1_bs = (int64_t)C * N; d.k1 = (int)C;
    d.w = (const float*)256; d.w_sm = (int64_t)C; d.w_sk = 1; d.w_bs = (int64_t)C * C;
    d.r = (void*)256; d.r_bs = (int64_t)C * N;
    d.y = (void*)256; d.y_bs = (int64_t)C * N;
    d.m = (int)C; d.n = (int64_t)N; d.batch = (int)B; d.groups = 1; d.dtype = s->dtype;
    w.pw_ws = cv.take(mi_pw_gemm_workspace(&d));
  }
  w.v = cv.take(align_up(B * C * N * 2, 256));
  w.bytes = cv.off;
  return w;
}

template <int C, int HEADS, int TH, int NW>
static int fm_launch(const mi_mdta_shape* s, const FmPackLayout& l, const void* pack, const void* x, void* v, float* part, float* mean,
                     float* rstd, int with_bias, int S, hipStream_t st) {
  using K = FmCfg<C, HEADS, TH, NW>;
  FmArgs a;
  const unsigned char* pk = (const unsigned char*)pack;
  a.x = (const bf16*)x; a.v = (bf16*)v; a.part = part; a.mean = mean; a.rstd = rstd;
  a.w1p = (const bf16*)(pk + l.w1p); a.wdp = (const float*)(pk + l.wdp);
  a.B = s->B; a.H = s->H; a.W = s->W; a.with_bias = with_bias;
  a.tiles_x = s->W / 32; a.tiles_y = s->H / TH; a.S = S;
  { const char* e = MI_ENV(MI_FM_DEBUG); a.dbg = e ? atoi(e) : 0; }
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  MI_CHECK_HIP(hipGetDevice(&dev));
  const unsigned bit = 1u << (dev & 31);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)fm_fwd_kernel<C, HEADS, TH, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)K::LDS_BYTES));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const double N = (double)s->H * s->W * s->B;
  ProfScope ps(st, K_MDTA_FUSED_A, 2.0 * C * N * 2.0, 2.0 * N * (3.0 * C * C + (double)C * K::c) + 2.0 * N * 9.0 * 3.0 * C);
  hipLaunchKernelGGL((fm_fwd_kernel<C, HEADS, TH, NW>), dim3((unsigned)(s->B * S)), dim3(64 * NW), K::LDS_BYTES, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_mdta_fused_ok(const mi_mdta_shape* s) { return (fm_kind(s) != FM_NONE && !MI_ENV(MI_NO_FUSED_MDTA)) ? 1 : 0; }

extern "C" size_t mi_mdta_fused_pack_bytes(const mi_mdta_shape* s) {
  if (fm_kind(s) == FM_NONE) return 0;
  return fm_pack_layout(s->C).bytes;
}

extern "C" int mi_mdta_fused_pack(const mi_mdta_shape* s, const float* ln_w, const float* ln_b, const mi_mdta_params* p, void* pack,
                                  void* stream) {
  MI_CHECK_ARG(fm_kind(s) != FM_NONE, "mdta_fused_pack: shape not covered by the fused kernel (mi_mdta_fused_ok)");
  MI_CHECK_ARG(ln_w && p && p->qkv_w && p->dw_w && pack && aligned16(pack), "mdta_fused_pack: null / unaligned pointer");
  const FmPackLayout l = fm_pack_layout(s->C);
  unsigned char* pk = (unsigned char*)pack;
  FmPackArgs a;
  a.ln_w = ln_w; a.ln_b = ln_b; a.qkv_w = p->qkv_w; a.qkv_b = p->qkv_b; a.dw_w = p->dw_w; a.dw_b = p->dw_b;
  a.w1p = (bf16*)(pk + l.w1p); a.wdp = (float*)(pk + l.wdp);
  a.C = s->C; a.nchunk = l.nchunk;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_FUSED_PACK, (double)l.bytes, 0.0);
  hipLaunchKernelGGL(fm_pack_kernel, dim3(64), dim3(256), 0, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" size_t mi_mdta_fused_workspace(const mi_mdta_shape* s) {
  if (fm_kind(s) == FM_NONE) return 0;
  return fm_ws_layout(s, nullptr).bytes;
}

// out = residual + project_out(softmax(temperature * q^ k^T) v) with q, k, v = dw3x3(qkv(LN(x))): pass A (this file), the
// partial sum, the c x c fold (attn_small.hip) and the per-image-weight GEMM (pw_gemm.hip).  ln_with_bias: WithBias / BiasFree
// LayerNorm (its affine lives in `pack`).  mean / rstd: optional [B][H*W] statistics outputs.
extern "C" int mi_mdta_fused_fwd(const mi_mdta_shape* s, const mi_mdta_params* p, const void* pack, int ln_with_bias, const void* x,
                                 const void* residual, void* out, float* mean, float* rstd, void* ws, void* stream) {
  const FmKind k = fm_kind(s);
  MI_CHECK_ARG(k != FM_NONE, "mdta_fused_fwd: shape not covered by the fused kernel (mi_mdta_fused_ok)");
  MI_CHECK_ARG(p && p->temperature && p->proj_w && pack && x && out && ws, "mdta_fused_fwd: null pointer");
  MI_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "mdta_fused_fwd: mean and rstd go together");
  MI_CHECK_ARG(aligned16(pack) && aligned16(x) && aligned16(out) && aligned16(ws), "mdta_fused_fwd: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const FmPackLayout l = fm_pack_layout(s->C);
  FmWs w = fm_ws_layout(s, ws);
  const int S = fm_splits(s, 8);
  if (k == FM_48_1) MI_TRY((fm_launch<48, 1, 8, 4>(s, l, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, st)));
  else if (k == FM_96_2) MI_TRY((fm_launch<96, 2, 8, 4>(s, l, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, st)));
  else MI_TRY((fm_launch<96, 1, 8, 4>(s, l, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, st)));
  {
    ProfScope ps(st, K_GRAM_REDUCE, 4.0 * s->B * (S + 1) * ((double)s->C * (s->C / s->heads) + 2.0 * s->C), 0.0);
    hipLaunchKernelGGL(fm_reduce_kernel, dim3(s->B * s->heads), dim3(256), 0, st, w.part, w.graw, w.ss, S, s->C, s->heads);
    MI_LAUNCH_CHECK();
  }
  MI_TRY(launch_attn_fold(w.graw, w.ss, p->temperature, p->proj_w, w.P, w.A, w.nrm, w.M, s->B, s->C, s->heads, st));
  const int64_t N = (int64_t)s->H * s->W;
  mi_pw_desc d;
  memset(&d, 0, sizeof(d));
  d.x1 = w.v; d.x1_bs = (int64_t)s->C * N; d.k1 = s->C;
  d.w = w.M; d.w_sm = s->C; d.w_sk = 1; d.w_bs = (int64_t)s->C * s->C;
  d.bias = p->proj_b;
  d.r = residual; d.r_bs = (int64_t)s->C * N;
  d.y = out; d.y_bs = (int64_t)s->C * N;
  d.m = s->C; d.n = N; d.batch = s->B; d.groups = 1; d.dtype = s->dtype;
  return mi_pw_gemm(&d, w.pw_ws, stream);
}
