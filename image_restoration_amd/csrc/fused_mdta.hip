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
  static constexpr int NB = C / 16;                      // 16-row blocks of q (= of k, = of v)
  static constexpr int NCHUNK = 3 * NB;                  // chunk order: k blocks, q blocks, v blocks
  static constexpr int CB = c / 16;                      // column (k) blocks per head: block cb belongs to wave cb
  static constexpr int NQD = (NB + 1) / 2;               // q diagonal blocks per diagonal wave (waves NW-2, NW-1)
  static constexpr int VPR = TW / 8;
  static constexpr int FRAG = 1024;                      // bytes of one operand fragment (16 channels x 32 pixels)
  static constexpr int H0_BYTES = CH * PLANE * 2;        // one h0 chunk buffer (two of them)
  static constexpr int KT_BYTES = NB * TH * FRAG;
  static constexpr int QC_BYTES = TH * FRAG;             // one q chunk buffer (two of them)
  static constexpr int W1_BYTES = NCHUNK * CH * W1S * 2; // ALL chunks' weights, resident for the workgroup's life
  static constexpr int WD_BYTES = NCHUNK * CH * 10 * 4;  // depthwise taps + bias [chunk][16][10] fp32
  static constexpr int S_BYTES = C * PLANE * 2;          // raw x of a tile, staged plane-major: aliases the h0 buffers + K image
  static constexpr int HK_BYTES = 2 * H0_BYTES + KT_BYTES;
  static constexpr int A_BYTES = HK_BYTES > S_BYTES ? HK_BYTES : S_BYTES;
  static constexpr int LDS_BYTES = A_BYTES + 2 * QC_BYTES + W1_BYTES + WD_BYTES;
  static constexpr int NBV = C * HR * VPR;               // 16-byte vectors of a tile body (rows y0-1 .. y0+TH)
  static constexpr int NBN = (NBV + NT - 1) / NT;
  static constexpr int NE = HPXP - BODY;                 // halo-column pixels + padding per plane
  static constexpr int NEN = (C * NE + NT - 1) / NT;
  static_assert(C % 16 == 0 && c % 16 == 0 && NW == 8 && TH == 8, "unsupported tile");
  static_assert(CB <= NW - 2, "waves NW-2 / NW-1 take the q diagonal blocks");
  static_assert(4 * MTW <= 64, "validity mask");
  static_assert(A_BYTES % 16 == 0 && W1_BYTES % 16 == 0 && WD_BYTES % 16 == 0, "LDS carve alignment");
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

struct FmArgs {
  const bf16* x; bf16* v; float* part;                  // part: [B][S][HEADS*c*c + 2C] fp32
  float* mean; float* rstd;                              // optional LayerNorm statistics [B][H*W]
  const bf16* w1p; const float* wdp;
  int B, H, W, with_bias, tiles_x, tiles_y, S, dbg;
};

// packed-weight blob: W1p [NCHUNK][16][C + 8] bf16 (fp32 bias in the row padding), WDp [NCHUNK][16][10] fp32 (taps, bias)
struct FmPackLayout { size_t w1p, wdp, bytes; int nchunk; };
static FmPackLayout fm_pack_layout(int C) {
  FmPackLayout l;
  l.nchunk = 3 * C / 16;
  size_t off = 0;
  l.w1p = off; off = align_up(off + (size_t)l.nchunk * 16 * (C + 8) * 2, 256);
  l.wdp = off; off = align_up(off + (size_t)l.nchunk * 16 * 10 * 4, 256);
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
  const int64_t n_w1 = (int64_t)a.nchunk * 16 * W1S, n_wd = (int64_t)a.nchunk * 16 * 10;
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
    const int64_t r0 = e - n_w1;                       // [chunk][row 0..15][tap 0..8, bias]
    const int tp = (int)(r0 % 10);
    const int64_t q = r0 / 10;
    const int r = (int)(q % 16), ci = (int)(q / 16);
    const int ch = fm_channel(C, ci, r);
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

// One 8-wave workgroup per CU walks a contiguous range of one image's TH x 32 tiles.  Resident for its whole life: the packed
// weights of all 3C output channels (LDS), its share of the Gram accumulators (registers).  Per tile, per 16-channel chunk:
//   GEMM1(ci) -> H0[ci & 1]   |barrier|   Gram(ci - 1, if that was a q chunk)   conv(ci) -> K image / Q[ci & 1] / HBM
// ONE barrier per chunk: the h0 and q buffers are double-buffered, so a wave may run one phase ahead of the slowest.
template <int C, int HEADS, int TH, int NW>
__global__ __launch_bounds__(64 * NW, 2) void fm_fwd_kernel(FmArgs a) {
  using K = FmCfg<C, HEADS, TH, NW>;
  constexpr int NT = K::NT, TW = K::TW;
  extern __shared__ __attribute__((aligned(16))) unsigned char fm_lds[];
  bf16* const H0 = reinterpret_cast<bf16*>(fm_lds);                       // [2][16][PLANE]
  unsigned char* const KT = fm_lds + 2 * K::H0_BYTES;                     // [NB][TH][FRAG]
  unsigned char* const QC = fm_lds + K::A_BYTES;                          // [2][TH][FRAG]
  bf16* const W1 = reinterpret_cast<bf16*>(QC + 2 * K::QC_BYTES);         // [NCHUNK][16][W1S]
  float* const WD = reinterpret_cast<float*>(QC + 2 * K::QC_BYTES + K::W1_BYTES);   // [NCHUNK][16][10]
  bf16* const S = reinterpret_cast<bf16*>(fm_lds);
  const int t = threadIdx.x, lane_outer = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x / a.S, sp = blockIdx.x - b * a.S;
  const int tiles = a.tiles_x * a.tiles_y;
  const int t0 = (int)((int64_t)sp * tiles / a.S), t1 = (int)((int64_t)(sp + 1) * tiles / a.S);
  const int64_t HW = (int64_t)a.H * a.W;
  const bf16* const xb = a.x + (int64_t)b * C * HW;

  f32x4 acc[K::NB], kd[HEADS], qd[K::NQD];
#pragma unroll
  for (int i = 0; i < K::NB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < HEADS; ++i) kd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < K::NQD; ++i) qd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool stamp = (a.dbg & 0x1000) != 0;
  unsigned long long tk0 = 0, ta = 0, tb = 0, c_stage = 0, c_ln = 0, c_gemm = 0, c_bar = 0, c_gram = 0, c_conv = 0;
  if (stamp) tk0 = fm_clock();
  FmStage<K> stg;
  if (t0 < t1) fm_stage_load<K>(stg, xb, t, (t0 % a.tiles_x) * TW, (t0 / a.tiles_x) * TH, a.H, a.W, HW);
  {                                                       // the weights of every chunk: global -> LDS, once
    const u32x4* s1 = reinterpret_cast<const u32x4*>(a.w1p);
    for (int vv = t; vv < K::W1_BYTES / 16; vv += NT) reinterpret_cast<u32x4*>(W1)[vv] = s1[vv];
    const u32x4* s2 = reinterpret_cast<const u32x4*>(a.wdp);
    for (int vv = t; vv < K::WD_BYTES / 16; vv += NT) reinterpret_cast<u32x4*>(WD)[vv] = s2[vv];
  }

  for (int tile = t0; tile < t1; ++tile) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    if (stamp) ta = fm_clock();
    __syncthreads();                                      // every wave is done with the previous tile: the aliased region is free
    fm_stage_store<K>(stg, S, t);
    __syncthreads();
    if (stamp) { tb = fm_clock(); c_stage += tb - ta; ta = tb; }

    // ---------------------------------------------------------------- LN(x) -> A-operand fragments (registers)
    // element order of a 32-k fragment (same for A and B): j < 4 is k = 4g + j, j >= 4 is k = 16 + 4g + (j - 4)
    s16x8 xa[K::MTW][K::KS32 > 0 ? K::KS32 : 1];
    s16x8 xt[K::MTW];
    unsigned long long vmask = 0;
    {
      int lane_o = lane_outer;
      asm volatile("" : "+v"(lane_o));
      const int lane = lane_o, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
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
          const float sub = a.with_bias ? mu : 0.f;     // BiasFree: x / sqrt(var + eps), x not centred (Restormer.py:37-39)
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
          if (a.mean && g == 0 && !(a.dbg & 0x1000)) {   // statistics of the tile's own pixels
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
    }
    if (stamp) { tb = fm_clock(); c_ln += tb - ta; ta = tb; }
    __syncthreads();                                      // the staged x is dead: the region becomes the h0 buffers / the K image
    if (stamp) { tb = fm_clock(); c_bar += tb - ta; }

#pragma unroll 1
    for (int ci = 0; ci < K::NCHUNK; ++ci) {
      if (stamp) ta = fm_clock();
      int lane_c = lane_outer;
      asm volatile("" : "+v"(lane_c));
      const int lane = lane_c, li = lane & 15, g = lane >> 4;
      // the next tile's x leaves HBM while this tile's v chunks are computed
      if (ci == 2 * K::NB && tile + 1 < t1) {
        const int nt_ = tile + 1;
        fm_stage_load<K>(stg, xb, wv * 64 + lane, (nt_ % a.tiles_x) * TW, (nt_ / a.tiles_x) * TH, a.H, a.W, HW);
      }
      bf16* const H0c = H0 + (ci & 1) * (K::CH * K::PLANE);
      // ------------------------------------------------------------ GEMM1: h0 chunk = W'[chunk] . LN(x), to LDS
      if (!(a.dbg & 32)) {
        const bf16* wrow = &W1[(ci * K::CH + li) * K::W1S];
        const bf16* wr = wrow + 4 * g;
        s16x8 bw[K::KS32 > 0 ? K::KS32 : 1];
        s16x8 bt = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < K::KS32; ++ks)
          bw[ks] = cat8(*reinterpret_cast<const s16x4*>(wr + ks * 32), *reinterpret_cast<const s16x4*>(wr + ks * 32 + 16));
        if (K::KT16) bt = cat8(*reinterpret_cast<const s16x4*>(wr + K::KS32 * 32), (s16x4){0, 0, 0, 0});
        const float bias = *reinterpret_cast<const float*>(wrow + C);
        bf16* hrow = &H0c[li * K::PLANE + 4 * g];
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
      if (stamp) { tb = fm_clock(); c_gemm += tb - ta; ta = tb; }
      __syncthreads();
      if (stamp) { tb = fm_clock(); c_bar += tb - ta; ta = tb; }

      // ------------------------------------------------------------ Gram of the PREVIOUS chunk if it was a q chunk
      if (ci > K::NB && ci <= 2 * K::NB && !(a.dbg & 128)) {
        const int rb = ci - 1 - K::NB;
        const unsigned char* const Qp = QC + ((ci - 1) & 1) * K::QC_BYTES;
#define FM_GRAM(RB)                                                                                              \
        case RB: {                                                                                               \
          if constexpr (RB < K::NB) {                                                                            \
            constexpr int head = RB / K::CB;                                                                     \
            constexpr bool first = (RB % K::CB) == 0;                                                            \
            if (wv < K::CB) {                                                                                    \
              const unsigned char* const Kp = KT + (head * K::CB + wv) * (TH * K::FRAG);                         \
              _Pragma("unroll")                                                                                  \
              for (int ks = 0; ks < TH; ++ks) {                                                                  \
                const s16x8 af = fm_ldfrag(Qp, ks, lane);                                                        \
                const s16x8 bf_ = fm_ldfrag(Kp, ks, lane);                                                       \
                acc[RB] = mfma32(af, bf_, acc[RB]);                                                              \
                if (first) kd[head] = mfma32(bf_, bf_, kd[head]);                                                \
              }                                                                                                  \
            } else if (wv == NW - 2 + (RB & 1)) {                                                                \
              _Pragma("unroll")                                                                                  \
              for (int ks = 0; ks < TH; ++ks) {                                                                  \
                const s16x8 af = fm_ldfrag(Qp, ks, lane);                                                        \
                qd[RB / 2] = mfma32(af, af, qd[RB / 2]);                                                         \
              }                                                                                                  \
            }                                                                                                    \
          }                                                                                                      \
        } break;
        switch (rb) {
          FM_GRAM(0) FM_GRAM(1) FM_GRAM(2) FM_GRAM(3) FM_GRAM(4) FM_GRAM(5)
          default: break;
        }
#undef FM_GRAM
      }

      if (stamp) { tb = fm_clock(); c_gram += tb - ta; ta = tb; }
      // ------------------------------------------------------------ depthwise 3x3 (VALU): one (channel, tile row, octet) per lane
      if (!(a.dbg & 64)) {
        const int tt = wv * 64 + lane;
        const int chrow = tt >> 5, row = (tt >> 2) & 7, cg = tt & 3;
        const int kind = ci < K::NB ? 0 : (ci < 2 * K::NB ? 1 : 2);              // 0: k, 1: q, 2: v
        const int blk = ci - kind * K::NB;
        const float* wp = WD + (ci * K::CH + chrow) * 10;
        float w[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) w[i] = wp[i];
        const bf16* h = &H0c[chrow * K::PLANE];
        const int eoff = K::BODY + (cg == 3 ? K::HR : 0);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = w[9];
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {
          const int rin = row + dr;
          const u32x4 r1 = *reinterpret_cast<const u32x4*>(h + rin * TW + 8 * cg);
          const u16 e1 = reinterpret_cast<const u16*>(h)[eoff + rin];
          float v[10];
#pragma unroll
          for (int k = 0; k < 4; ++k) { v[1 + 2 * k] = bf_lo(r1[k]); v[2 + 2 * k] = bf_hi(r1[k]); }
          const float edge = bf_lo(e1), lft = from_prev_lane(v[8]), rgt = from_next_lane(v[1]);
          v[0] = cg == 0 ? edge : lft;
          v[9] = cg == 3 ? edge : rgt;
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) o[j] += w[dr * 3 + kx] * v[j + kx];
        }
        u32x4 oa;
#pragma unroll
        for (int k = 0; k < 4; ++k) oa[k] = pack_bf2(o[2 * k], o[2 * k + 1]);
        if (kind == 2) {
          *reinterpret_cast<u32x4*>(a.v + ((int64_t)b * C + 16 * blk + chrow) * HW + (int64_t)(y0 + row) * a.W + x0 + 8 * cg) = oa;
        } else {
          unsigned char* img = kind == 0 ? KT + blk * (TH * K::FRAG) : QC + (ci & 1) * K::QC_BYTES;
          *reinterpret_cast<u32x4*>(img + row * K::FRAG + cg * 256 + fm_slot(chrow, row, cg) * 16) = oa;
        }
      }
      if (stamp) { tb = fm_clock(); c_conv += tb - ta; }
      if ((a.dbg & 15) == 2 && ci == ((a.dbg >> 8) & 15)) { // debug: image of chunk (dbg >> 8) as [16][TH*32] bf16 into v of tile 0
        __syncthreads();
        if (blockIdx.x == 0 && tile == t0) {
          const unsigned char* img = ci < K::NB ? KT + ci * (TH * K::FRAG) : QC + (ci & 1) * K::QC_BYTES;
          for (int e = t; e < 16 * TH * 32; e += NT) {
            const int r = e / (TH * 32), px = e % (TH * 32), trow = px / 32, oct = (px % 32) / 8, w8 = px % 8;
            a.v[e] = reinterpret_cast<const bf16*>(img + trow * K::FRAG + (oct * 16 + fm_slot(r, trow, oct)) * 16)[w8];
          }
        }
        return;
      }
    }
  }

  if (stamp && a.mean && lane_outer == 0) {
    float* o = a.mean + ((int64_t)blockIdx.x * NW + wv) * 8;
    o[0] = (float)c_stage; o[1] = (float)c_ln; o[2] = (float)c_gemm; o[3] = (float)c_bar; o[4] = (float)c_gram; o[5] = (float)c_conv;
    o[6] = (float)(fm_clock() - tk0); o[7] = (float)(t1 - t0);
  }
  // ------------------------------------------------------------------ this workgroup's partial: G blocks, then sums of squares
  {
    const int lane = lane_outer, li = lane & 15, g = lane >> 4;
    float* pz = a.part + (int64_t)blockIdx.x * (HEADS * K::c * K::c + 2 * C);
#pragma unroll
    for (int rb = 0; rb < K::NB; ++rb) {
      const int head = rb / K::CB, ib = rb % K::CB;
      if (wv < K::CB) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pz[head * K::c * K::c + (ib * 16 + 4 * g + r) * K::c + wv * 16 + li] = acc[rb][r];
      }
      if (wv == NW - 2 + (rb & 1) && (li >> 2) == g) {
        const f32x4 d = qd[rb / 2];
        pz[HEADS * K::c * K::c + rb * 16 + li] = (li & 3) == 0 ? d[0] : ((li & 3) == 1 ? d[1] : ((li & 3) == 2 ? d[2] : d[3]));
      }
    }
#pragma unroll
    for (int head = 0; head < HEADS; ++head)
      if (wv < K::CB && (li >> 2) == g) {
        const f32x4 d = kd[head];
        pz[HEADS * K::c * K::c + C + head * K::c + wv * 16 + li] =
            (li & 3) == 0 ? d[0] : ((li & 3) == 1 ? d[1] : ((li & 3) == 2 ? d[2] : d[3]));
      }
  }
}

// graw[z][c*c] and ss[z][2c] (q sums of squares, then k) from the S partials of every image, summed in a fixed order
__global__ __launch_bounds__(256) void fm_reduce_kernel(const float* __restrict__ part, float* __restrict__ graw, float* __restrict__ ss,
                                                        int S, int C, int heads) {
  const int c = C / heads, z = blockIdx.y, b = z / heads, head = z - b * heads;
  const int64_t pstride = (int64_t)heads * c * c + 2 * C;
  const float* p0 = part + (int64_t)b * S * pstride;
  const int e = blockIdx.x * 256 + threadIdx.x;         // one output element per thread: the S partials in a fixed order
  if (e >= c * c + 2 * c) return;
  int64_t off;
  if (e < c * c) off = (int64_t)head * c * c + e;
  else if (e < c * c + c) off = (int64_t)heads * c * c + head * c + (e - c * c);
  else off = (int64_t)heads * c * c + C + head * c + (e - c * c - c);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < S; k += 4) {
    s0 += p0[(int64_t)k * pstride + off]; s1 += p0[(int64_t)(k + 1) * pstride + off];
    s2 += p0[(int64_t)(k + 2) * pstride + off]; s3 += p0[(int64_t)(k + 3) * pstride + off];
  }
  for (; k < S; ++k) s0 += p0[(int64_t)k * pstride + off];
  const float s = (s0 + s1) + (s2 + s3);
  if (e < c * c) graw[(int64_t)z * c * c + e] = s;
  else ss[(int64_t)z * 2 * c + (e - c * c)] = s;
}


// ================================================================================================ round 4: the depthwise conv on the matrix cores
// Same contract as fm_fwd_kernel (x in, v + Gram partials out), rebuilt so that the VALU does as little as possible (the round-3
// kernel was bound by its ~5000 vector instructions per wave and tile: 4 cycles each on a pipe two waves share; the SQ counters
// of rocprofv3 on this pool report a third of the true instruction counts, which hid that for a round):
//   * the depthwise 3x3 runs on the MATRIX cores.  The 1x1 output h0 of a 16-channel chunk is kept PIXEL-major in LDS (one 32-byte
//     record of 16 channels per halo pixel, rows of 34 records), and the conv is an im2col GEMM against a block-diagonal tap
//     matrix: D[px][ch] = sum_{t, ch'} h0[px + t][ch'] * (tap[ch][t] if ch' == ch).  One MFMA 16x16x32 contracts 2 taps x 16
//     channels, five of them give 16 pixels x 16 channels (the tenth "tap" is the depthwise bias against a record of ones); the A
//     operand of every tap is ONE aligned ds_read_b128 of a neighbouring record (no halo special cases, no DPP, no unpacking), the B
//     operand of a chunk (one bf16 per lane and k-step, masked into place with four ands) is shared by all of its pixel tiles.
//     The taps are rounded to bf16 - what a bf16 autocast of the reference's depthwise conv does (MoCE-IR-main/src/train.py:258
//     precision="16-mixed"); h0 is bf16 in the unfused chain too.
//   * GEMM1 is flipped (D[channel][pixel]: a lane holds 4 channels of one pixel = 8 bytes of its record), its weights are
//     fragment-major in LDS (one ds_read_b128 per fragment), and its bias AND the conv's zero padding ride in the GEMM: a "ones"
//     channel (1 inside the image, 0 outside; LayerNorm of the zero-filled out-of-image pixels is 0) against the bias column.
//   * the chunks are processed in GROUPS of G (all 9 at C = 48): GEMM1 of a whole group -> ONE barrier -> everything else of the
//     group is WAVE-LOCAL: wave w owns tile row w, runs the conv of its 32 pixels for every chunk of the group, keeps its k
//     fragments in registers, feeds each q fragment straight into the Gram MFMAs against them (the conv's output registers ARE
//     the Gram operand: no K / Q image in LDS), sums the squares for F.normalize on the way, and sends v (SAVE: q, k, qkv0 too)
//     through a wave-private staging patch to 64-byte row segments.  Per tile: 3 + NG barriers instead of 12 / 21.  Every wave
//     carries its own partial of the c x c Gram (its rows): the partial-sum kernel adds 8 S partials per image instead of S.
template <int C_, int HEADS_> struct Fm3Cfg {
  static constexpr int C = C_, HEADS = HEADS_, TH = 8, TW = 32, NW = 8;
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
  static constexpr int NKS = KS32 + 1;                    // weight fragments per chunk: the last holds the 16-deep tail (if any) + the bias slot
  static constexpr int NV = 8 * KS32 + 4 * KT16;
  static constexpr int NB = C / 16;
  static constexpr int NCHUNK = 3 * NB;                   // chunk order: k blocks, q blocks, v blocks (fm_channel)
  static constexpr int CB = c / 16;
  static constexpr int VPR = TW / 8;
  static constexpr int RW = TW + 2;                       // records per h0 row (halo columns inline)
  static constexpr int REC = 32;                          // bytes of a record: 16 channels bf16
  static constexpr int H0_BYTES = HR * RW * REC;
  static constexpr int FRAG = 1024;
  static constexpr int W1_BYTES = NCHUNK * NKS * FRAG;
  static constexpr int TB_BYTES = NCHUNK * 5 * 64 * 2;
  static constexpr int WT_BYTES = W1_BYTES + TB_BYTES;
  static constexpr int S_BYTES = C * PLANE * 2;
  static constexpr int NBV = C * HR * VPR;
  static constexpr int NBN = (NBV + NT - 1) / NT;
  static constexpr int NE = HPXP - BODY;
  static constexpr int NEN = (C * NE + NT - 1) / NT;
  static_assert(C % 16 == 0 && c % 16 == 0, "unsupported channel count");
  static_assert(W1_BYTES % 16 == 0 && TB_BYTES % 16 == 0, "LDS carve alignment");
};

struct Fm3PackLayout { size_t w1f, tb, bytes; };
static Fm3PackLayout fm3_pack_layout(int C, size_t base) {
  const int nchunk = 3 * C / 16, nks = C / 32 + 1;
  Fm3PackLayout l;
  size_t off = align_up(base, 256);
  l.w1f = off; off += (size_t)nchunk * nks * 1024;
  l.tb = off; off += (size_t)nchunk * 5 * 64 * 2;
  l.bytes = align_up(off, 256);
  return l;
}

struct Fm3PackArgs {
  const float *ln_w, *ln_b, *qkv_w, *qkv_b, *dw_w, *dw_b;
  bf16* w1f; u16* tb;
  int C, nchunk, nks;
};
__global__ __launch_bounds__(256) void fm3_pack_kernel(Fm3PackArgs a) {
  const int C = a.C, ks32 = C / 32;
  const int64_t n_w1 = (int64_t)a.nchunk * a.nks * 512, n_tb = (int64_t)a.nchunk * 5 * 64;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_w1 + n_tb; e += (int64_t)gridDim.x * 256) {
    if (e < n_w1) {                                        // [chunk][ks][lane][8]: element j of lane (li, g) is k = ks*32 + {4g+j | 16+4g+j-4}
      const int j = (int)(e & 7), l = (int)((e >> 3) & 63);
      const int64_t q = e >> 9;
      const int ks = (int)(q % a.nks), ci = (int)(q / a.nks);
      const int li = l & 15, g = l >> 4;
      const int ch = fm_channel(C, ci, li);
      int k;
      if (ks < ks32) k = ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
      else k = (j < 4 && ks32 * 32 + 4 * g + j < C) ? ks32 * 32 + 4 * g + j : -1;   // tail / bias fragment: zero-padded 32-deep
      float v = (k >= 0 && k < C) ? a.qkv_w[(int64_t)ch * C + k] * a.ln_w[k] : 0.f;
      if (ks == ks32 && g == 0 && j == 4) {               // the "ones" slot: b' = b_qkv + W_qkv . beta rides in the GEMM
        v = a.qkv_b ? a.qkv_b[ch] : 0.f;
        if (a.ln_b)
          for (int kk = 0; kk < C; ++kk) v += a.qkv_w[(int64_t)ch * C + kk] * a.ln_b[kk];
      }
      a.w1f[e] = (bf16)v;
      continue;
    }
    const int64_t r = e - n_w1;                            // [chunk][k-step][lane]: the lane's only non-zero of the block-diagonal tap fragment
    const int l = (int)(r & 63);
    const int64_t q = r >> 6;
    const int s = (int)(q % 5), ci = (int)(q / 5);
    const int li = l & 15, g = l >> 4, t = 2 * s + (g >> 1);
    const int ch = fm_channel(C, ci, li);
    u16 v = 0;
    if ((g & 1) == (li >> 3)) {                           // tap 9 is the depthwise bias: its A operand is a record of ones
      const bf16 w = (bf16)(t <= 8 ? a.dw_w[(int64_t)ch * 9 + t] : (a.dw_b ? a.dw_b[ch] : 0.f));
      v = __builtin_bit_cast(u16, w);
    }
    a.tb[r] = v;
  }
}

struct Fm3Args {
  const bf16* x; bf16* v; float* part; float* mean; float* rstd;
  const unsigned char* wt;                               // packed [W1 fragments | tap words]
  bf16* qkv0; bf16* qk;                                  // SAVE: 1x1 output [B][3C][H][W] and post-conv q, k [B][2C][H][W] (channels 0..2C of qkv)
  int B, H, W, with_bias, tiles_x, tiles_y, S, dbg;
};

template <int C_, int HEADS_, int G_> struct Fm4Cfg : Fm3Cfg<C_, HEADS_> {
  using B3 = Fm3Cfg<C_, HEADS_>;
  static constexpr int G = G_;
  static constexpr int NG = B3::NCHUNK / G;
  static constexpr int H0S_BYTES = B3::H0_BYTES + 64;       // one chunk's records + a dummy record (padding pixels land there) + a ones record
  static constexpr int H0G_BYTES = G * H0S_BYTES;
  static constexpr int VCS = 72;                           // bytes per channel row of the staging patch (64 + 8: conflict-free b64 writes)
  static constexpr int VST_BYTES = B3::NW * 16 * VCS;
  static constexpr int NBUF = G < B3::NCHUNK ? 2 : 1;       // several groups: two h0 buffers, GEMM1 of group g + 1 shares a barrier interval with the wave-local phase of group g
  static constexpr int A4_BYTES = ((NBUF * H0G_BYTES > B3::S_BYTES ? NBUF * H0G_BYTES : B3::S_BYTES) + 15) / 16 * 16;
  static constexpr int LDS4_BYTES = A4_BYTES + VST_BYTES + B3::WT_BYTES;
  static constexpr int NACC = B3::NB * B3::CB;             // Gram blocks (q block rb x the column blocks of its head), all in every wave
  static_assert(B3::NCHUNK % G == 0, "groups must tile the chunks");
  static_assert(G == B3::NCHUNK || G == B3::NB, "a group is everything or one of the k / q / v thirds");
  static_assert(LDS4_BYTES <= 160 * 1024, "one workgroup per CU");
};

template <int C, int HEADS, int G, bool SAVE, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void fm4_fwd_kernel(Fm3Args a) {
  using K = Fm4Cfg<C, HEADS, G>;
  constexpr int NT = K::NT, TW = K::TW, TH = K::TH, NW = K::NW, NB = K::NB, CB = K::CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char fm4_lds[];
  unsigned char* const H0 = fm4_lds;                                       // [G][HR * RW records + dummy + ones]
  unsigned char* const VST = fm4_lds + K::A4_BYTES;                        // [NW][16][VCS]
  unsigned char* const W1 = VST + K::VST_BYTES;                            // [NCHUNK][NKS][FRAG]
  const u16* const TB = reinterpret_cast<const u16*>(W1 + K::W1_BYTES);    // [NCHUNK][5][64]
  bf16* const S = reinterpret_cast<bf16*>(fm4_lds);
  const int t = threadIdx.x, lane_outer = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int b = blockIdx.x / a.S, sp = blockIdx.x - b * a.S;
  const int tiles = a.tiles_x * a.tiles_y;
  const int t0 = (int)((int64_t)sp * tiles / a.S), t1 = (int)((int64_t)(sp + 1) * tiles / a.S);
  const int64_t HW = (int64_t)a.H * a.W;
  const bf16* const xb = a.x + (int64_t)b * C * HW;

  f32x4 acc[K::NACC];
  float ssq[NB], ssk[NB];
#pragma unroll
  for (int i = 0; i < K::NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NB; ++i) { ssq[i] = 0.f; ssk[i] = 0.f; }

  // STAMP build (MI_FM_DEBUG = 0x1000): shader-clock stamps at the phase boundaries, summed per wave and written to `mean`
  // ([workgroup][wave][8] floats: top barrier + stage, LayerNorm, barrier, GEMM1, barrier, wave-local phase, lifetime, tiles)
  unsigned long long tk0 = 0, ta = 0, tb_ = 0, cyc[6] = {0, 0, 0, 0, 0, 0};
  if (STAMP) tk0 = fm_clock();
#define FM4_STAMP(i) do { if (STAMP) { tb_ = fm_clock(); cyc[i] += tb_ - ta; ta = tb_; } } while (0)
  FmStage<K> stg;
  if (t0 < t1) fm_stage_load<K>(stg, xb, t, (t0 % a.tiles_x) * TW, (t0 / a.tiles_x) * TH, a.H, a.W, HW);
  {
    const u32x4* s1 = reinterpret_cast<const u32x4*>(a.wt);
    for (int vv = t; vv < K::WT_BYTES / 16; vv += NT) reinterpret_cast<u32x4*>(W1)[vv] = s1[vv];
  }

  for (int tile = t0; tile < t1; ++tile) {
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    if (STAMP) ta = fm_clock();
    __syncthreads();                                      // every wave is done with the previous tile's h0
    fm_stage_store<K>(stg, S, t);
    __syncthreads();
    FM4_STAMP(0);

    // Lane geometry of the three m-tiles a wave owns in the LayerNorm / GEMM1 phases (m-tile = 16 consecutive pixels of the
    // tile's linear halo-pixel order: HR rows of TW, then the left and the right halo column): where the lane's pixel sits in
    // the tile (for the validity test) and where its 8 bytes go in an h0 chunk buffer.  (Tile-invariant, but re-derived per
    // tile from an opaque lane id: nine registers the wave-local phase needs more.)
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
        else { rr = K::HR; cc = 0; }                       // padding pixels: the dummy record, never inside the image
        recoff[i] = fm4_rec(rr * K::RW + cc, g);
        prow[i] = rr < K::HR ? rr - 1 : -(1 << 20);        // tile-relative row / column of the pixel (halo: -1, TH / -1, TW)
        pcol[i] = cc - 1;
      }
    }
    // ---------------------------------------------------------------- LN(x) -> operand fragments (registers)
    // element order of a 32-k fragment: j < 4 is k = 4g + j, j >= 4 is k = 16 + 4g + (j - 4).  xt: the 16-deep tail (C = 48) in
    // slots 0..3 and, in slot 4 of the g = 0 lanes, the constant 1 of pixels inside the image - the "ones" channel whose
    // weight is the 1x1 bias, so that GEMM1 delivers W' LN(x) + b' inside the image and exactly 0 outside it (LN of the
    // zero-filled out-of-image pixels is 0): no bias add, no padding mask on the VALU.
    s16x8 xa[K::MTW][K::KS32 > 0 ? K::KS32 : 1];
    s16x8 xt[K::MTW];
    {
      int lane_o = lane_outer;
      asm volatile("" : "+v"(lane_o));
      const int lane = lane_o, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
#pragma unroll
      for (int i = 0; i < K::MTW; ++i) {
        const int mt = wv + NW * i < K::MT ? wv + NW * i : K::MT - 1;     // (waves 6, 7 repeat the last m-tile into the dummy record)
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
          const int64_t o = (int64_t)b * HW + (int64_t)Y * a.W + X;   // statistics of the tile's own pixels
          a.mean[o] = mu; a.rstd[o] = rstd;
        }
      }
    }
    FM4_STAMP(1);
    __syncthreads();                                      // the staged x is dead: the region becomes the h0 group
    FM4_STAMP(2);
    if (t < 2 * G * K::NBUF)                              // the ones record of every chunk buffer (the staging overwrote it)
      *reinterpret_cast<u32x4*>(H0 + (t >> 1) * K::H0S_BYTES + (K::HR * K::RW + 1) * K::REC + (t & 1) * 16) =
          (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    s16x8 kf[NB];                                         // this wave's k fragments (tile row wv: 16 channels x 32 pixels each)
    // GEMM1 of group grp: h0[chunk][pixel][16 ch] = W' LN(x) + b' (0 outside the image), into h0 buffer grp % NBUF
    auto gemm1 = [&](const int grp) {
      int lane_c = lane_outer;
      asm volatile("" : "+v"(lane_c));
      const int lane = lane_c;
      const unsigned char* const wf = W1 + lane * 16;
      s16x8 bw[K::NKS], bwn[K::NKS];
#pragma unroll
      for (int ks = 0; ks < K::NKS; ++ks) bw[ks] = *reinterpret_cast<const s16x8*>(wf + ((grp * G) * K::NKS + ks) * K::FRAG);
#pragma unroll
      for (int cg = 0; cg < G; ++cg) {
        const int ci = grp * G + cg;
        if (cg + 1 < G) {                               // the next chunk's weights are requested before this chunk's stores
#pragma unroll
          for (int ks = 0; ks < K::NKS; ++ks) bwn[ks] = *reinterpret_cast<const s16x8*>(wf + ((ci + 1) * K::NKS + ks) * K::FRAG);
        }
        unsigned char* const h = H0 + (grp % K::NBUF) * K::H0G_BYTES + cg * K::H0S_BYTES;
#pragma unroll
        for (int i = 0; i < K::MTW; ++i) {              // (m-tiles past the last one multiply the last again and land in the dummy record)
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < K::KS32; ++ks) d = mfma32(bw[ks], xa[i][ks], d);
          d = mfma32(bw[K::KS32], xt[i], d);
          *reinterpret_cast<u32x2*>(h + recoff[i]) = (u32x2){pk_bf2(d[0], d[1]), pk_bf2(d[2], d[3])};
        }
        if (cg + 1 < G) {
#pragma unroll
          for (int ks = 0; ks < K::NKS; ++ks) bw[ks] = bwn[ks];
        }
        __builtin_amdgcn_sched_barrier(0);              // one chunk's accumulators at a time
      }
    };

    gemm1(0);
#pragma unroll
    for (int grp = 0; grp < K::NG; ++grp) {
      FM4_STAMP(3);
      if (grp == K::NG - 1 && tile + 1 < t1) {            // the next tile's x leaves HBM behind the last group's wave-local work
        const int nt_ = tile + 1;
        fm_stage_load<K>(stg, xb, wv * 64 + lane_outer, (nt_ % a.tiles_x) * TW, (nt_ / a.tiles_x) * TH, a.H, a.W, HW);
      }
      __syncthreads();
      FM4_STAMP(4);
      if (K::NBUF == 2 && grp + 1 < K::NG) gemm1(grp + 1);   // into the other h0 buffer, inside this barrier interval
      // ------------------------------------------------------------ wave-local: conv, Gram, stores of tile row wv
      {
        int lane_c = lane_outer;
        asm volatile("" : "+v"(lane_c));
        const int lane = lane_c, li = lane & 15, g = lane >> 4;
        // byte addresses of the A-operand reads of chunk 0: (k-step, pixel half); chunk cg adds cg * H0S_BYTES as an immediate.
        // k-step s contracts taps 2s (lanes g < 2) and 2s + 1 (g >= 2); the tenth "tap" is the depthwise bias against the ones record
        unsigned aA[5][2];
        const unsigned h0base = (unsigned)(uintptr_t)(H0 + (grp % K::NBUF) * K::H0G_BYTES);
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          const int tp = 2 * s + (g >> 1);
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int r = tp < 9 ? (wv + tp / 3) * K::RW + li + 16 * hf + tp % 3 : K::HR * K::RW + 1;
            aA[s][hf] = h0base + r * K::REC + (((g & 1) ^ ((r >> 2) & 1)) << 4);
          }
        }
        // the four dwords of a tap fragment: the lane's one non-zero word masked into dword (li & 7) >> 1
        const unsigned wsel = 0xffffu << (16 * (li & 1));
        const int sel = (li & 7) >> 1;
        const unsigned m0 = sel == 0 ? wsel : 0u, m1 = sel == 1 ? wsel : 0u, m2 = sel == 2 ? wsel : 0u, m3 = sel == 3 ? wsel : 0u;
        unsigned char* const vst = VST + wv * (16 * K::VCS);
        const int s_ch = lane >> 2, s_cg = lane & 3;      // store role: (channel row, 8-pixel octet)
        const int64_t row_off = (int64_t)(y0 + wv) * a.W + x0 + 8 * s_cg;
        // Operand pipeline, pinned with inline asm (left to itself the compiler issues every read one MFMA ahead of its use): the
        // 10 fragments + 5 tap words of chunk cg + 1 are requested BEFORE the MFMAs of chunk cg; LDS operations complete in order,
        // so lgkmcnt(15) = "everything older than those 15 has landed".
        u32x4 opA[2][10];
        unsigned opT[2][5];
        const unsigned tbbase = (unsigned)(uintptr_t)(TB + grp * G * 5 * 64 + lane);
        constexpr int CGB = 65535 / K::H0S_BYTES;         // chunks an immediate offset reaches; for later ones the bases move up
#define FM4_RD(BUF, CG, OFFCG)                                                                                        \
        _Pragma("unroll")                                                                                             \
        for (int s_ = 0; s_ < 5; ++s_) {                                                                              \
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(opA[BUF][2 * s_]) : "v"(aA[s_][0]), "n"((OFFCG) * K::H0S_BYTES));      \
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(opA[BUF][2 * s_ + 1]) : "v"(aA[s_][1]), "n"((OFFCG) * K::H0S_BYTES));  \
          asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(opT[BUF][s_]) : "v"(tbbase), "n"(((CG) * 5 + s_) * 128));               \
        }
#define FM4_LANDED(BUF, N)                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                      \
                     : "+v"(opA[BUF][0]), "+v"(opA[BUF][1]), "+v"(opA[BUF][2]), "+v"(opA[BUF][3]), "+v"(opA[BUF][4]),       \
                       "+v"(opA[BUF][5]), "+v"(opA[BUF][6]), "+v"(opA[BUF][7]), "+v"(opA[BUF][8]), "+v"(opA[BUF][9]),       \
                       "+v"(opT[BUF][0]), "+v"(opT[BUF][1]), "+v"(opT[BUF][2]), "+v"(opT[BUF][3]), "+v"(opT[BUF][4]))
        FM4_RD(0, 0, 0)
#pragma unroll
        for (int cg = 0; cg < G; ++cg) {
          const int ci = grp * G + cg;
          const int kind = ci < NB ? 0 : (ci < 2 * NB ? 1 : 2);            // 0: k, 1: q, 2: v
          const int blk = ci - kind * NB;
          const int cur = cg & 1;
          if (cg + 1 < G) {
            if (cg + 1 == CGB + 1) {                      // (aA is rebuilt for every tile and group)
#pragma unroll
              for (int s = 0; s < 5; ++s) { aA[s][0] += (CGB + 1) * K::H0S_BYTES; aA[s][1] += (CGB + 1) * K::H0S_BYTES; }
            }
            if (cg + 1 <= CGB) {
              if (cur == 0) { FM4_RD(1, cg + 1, cg + 1) FM4_LANDED(0, 15); }
              else { FM4_RD(0, cg + 1, cg + 1) FM4_LANDED(1, 15); }
            } else {
              if (cur == 0) { FM4_RD(1, cg + 1, cg - CGB) FM4_LANDED(0, 15); }
              else { FM4_RD(0, cg + 1, cg - CGB) FM4_LANDED(1, 15); }
            }
          } else {
            if (cur == 0) FM4_LANDED(0, 0); else FM4_LANDED(1, 0);
          }
          f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 5; ++s) {
            const unsigned w = opT[cur][s] * 0x10001u;    // both halves; the masks keep the lane's half of its one dword
            const u32x4 f = {w & m0, w & m1, w & m2, w & m3};
            const s16x8 tf = __builtin_bit_cast(s16x8, f);
            d0 = mfma32(__builtin_bit_cast(s16x8, opA[cur][2 * s]), tf, d0);
            d1 = mfma32(__builtin_bit_cast(s16x8, opA[cur][2 * s + 1]), tf, d1);
          }
          // lane (channel li, group g): pixels 4g..4g+3 and 16+4g..16+4g+3 of row wv (bias included: the tenth tap)
          const u32x4 o = {pk_bf2(d0[0], d0[1]), pk_bf2(d0[2], d0[3]), pk_bf2(d1[0], d1[1]), pk_bf2(d1[2], d1[3])};
          if (kind < 2) {
            // row sums of squares for F.normalize, from the fp32 values (the bf16 rounding of the operands averages out over the
            // pixels: a relative 2^-9 / sqrt(N) on the norm)
            const float sq = (d0[0] * d0[0] + d0[1] * d0[1]) + (d0[2] * d0[2] + d0[3] * d0[3]) + (d1[0] * d1[0] + d1[1] * d1[1]) +
                             (d1[2] * d1[2] + d1[3] * d1[3]);
            const s16x8 fr = __builtin_bit_cast(s16x8, o);
            if (kind == 0) { kf[blk] = fr; ssk[blk] += sq; }
            else {
              ssq[blk] += sq;
              const int head = blk / CB;
#pragma unroll
              for (int cb = 0; cb < CB; ++cb) acc[blk * CB + cb] = mfma32(fr, kf[head * CB + cb], acc[blk * CB + cb]);
            }
          }
          if (kind == 2 || SAVE) {                        // rows of 32 pixels leave through the wave's staging patch
            wave_sync();
            *reinterpret_cast<u32x2*>(vst + li * K::VCS + g * 8) = (u32x2){o[0], o[1]};
            *reinterpret_cast<u32x2*>(vst + li * K::VCS + 32 + g * 8) = (u32x2){o[2], o[3]};
            wave_sync();
            const u32x2 p0 = *reinterpret_cast<const u32x2*>(vst + s_ch * K::VCS + s_cg * 16);
            const u32x2 p1 = *reinterpret_cast<const u32x2*>(vst + s_ch * K::VCS + s_cg * 16 + 8);
            const u32x4 val = {p0[0], p0[1], p1[0], p1[1]};
            if (kind == 2) *reinterpret_cast<u32x4*>(a.v + ((int64_t)b * C + 16 * blk + s_ch) * HW + row_off) = val;
            else *reinterpret_cast<u32x4*>(a.qk + ((int64_t)b * 2 * C + (kind == 1 ? 0 : C) + 16 * blk + s_ch) * HW + row_off) = val;
          }
          if constexpr (SAVE) {                           // qkv0 of row wv: records -> channel rows (transposing reads), 64-byte segments
            const int qq = li >> 2, pp = li & 3;
            const unsigned char* hr = H0 + (grp % K::NBUF) * K::H0G_BYTES + cg * K::H0S_BYTES;
            const int rs = (wv + 1) * K::RW + 1 + 8 * g + qq;
            // per 16-lane group g: pixels 8g..8g+7; lane 4q+p supplies record (pixel) q of a 4-pixel block, channels 4p..4p+3
            s16x4 u0 = tr_b16(hr + fm4_rec(rs, pp));
            s16x4 u1 = tr_b16(hr + fm4_rec(rs + 4, pp));
            lds_wait(u0, u1);
            const s16x8 both = cat8(u0, u1);              // lane (channel li): pixels 8g..8g+7
            const int ch = kind == 0 ? C + 16 * blk + li : (kind == 1 ? 16 * blk + li : 2 * C + 16 * blk + li);
            *reinterpret_cast<s16x8*>(a.qkv0 + ((int64_t)b * 3 * C + ch) * HW + (int64_t)(y0 + wv) * a.W + x0 + 8 * g) = both;
          }
        }
#undef FM4_RD
#undef FM4_LANDED
      }
      FM4_STAMP(5);
    }
  }

  if (STAMP && a.mean && lane_outer == 0) {
    float* o = a.mean + ((int64_t)blockIdx.x * NW + wv) * 8;
    for (int i = 0; i < 6; ++i) o[i] = (float)cyc[i];
    o[6] = (float)(fm_clock() - tk0); o[7] = (float)(t1 - t0);
  }
#undef FM4_STAMP
  // ------------------------------------------------------------------ this WAVE's partial: G blocks, then sums of squares
  {
    const int lane = lane_outer, li = lane & 15, g = lane >> 4;
    float* pz = a.part + ((int64_t)blockIdx.x * NW + wv) * (HEADS * K::c * K::c + 2 * C);
#pragma unroll
    for (int rb = 0; rb < NB; ++rb) {
      const int head = rb / CB, ib = rb % CB;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pz[head * K::c * K::c + (ib * 16 + 4 * g + r) * K::c + cb * 16 + li] = acc[rb * CB + cb][r];
      float sq = ssq[rb], sk = ssk[rb];
      sq += __shfl_xor(sq, 16); sq += __shfl_xor(sq, 32);
      sk += __shfl_xor(sk, 16); sk += __shfl_xor(sk, 32);
      if (g == 0) {
        pz[HEADS * K::c * K::c + rb * 16 + li] = sq;
        pz[HEADS * K::c * K::c + C + rb * 16 + li] = sk;
      }
    }
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
  int S = 256 / s->B;                                   // one 8-wave workgroup per CU
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  if (S > tiles / 4) S = tiles / 4 > 0 ? tiles / 4 : 1;   // at least 4 tiles per workgroup: the resident weights and the partial amortise
  return S;
}

struct FmWs { void* v; float* part; float* graw; float* ss; float* P; float* A; float* nrm; float* M; void* Mb; void* pw_ws; size_t bytes; };
static FmWs fm_ws_layout(const mi_mdta_shape* s, void* base) {
  const size_t N = (size_t)s->H * s->W, C = s->C, B = s->B, c = C / s->heads, Z = B * s->heads;
  Carver cv(base);
  FmWs w;
  const int S = fm_splits(s, 8);
  w.part = cv.take<float>(B * S * 8 * (s->heads * c * c + 2 * C) * sizeof(float));   // (one partial per WAVE in the fourth form)
  w.graw = cv.take<float>(Z * c * c * sizeof(float));
  w.ss = cv.take<float>(Z * 2 * c * sizeof(float));
  w.P = cv.take<float>(Z * c * c * sizeof(float));
  w.A = cv.take<float>(Z * c * c * sizeof(float));
  w.nrm = cv.take<float>(Z * 2 * c * sizeof(float));
  w.M = cv.take<float>(B * C * C * sizeof(float));
  w.Mb = cv.take(B * C * C * 2);                          // bf16 M_b for the per-image-weight GEMM (mi_pw_desc.w_b16)
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

template <int C, int HEADS, int G, bool SAVE>
static int fm4_launch(const mi_mdta_shape* s, const void* pack, const void* x, void* v, float* part, float* mean, float* rstd,
                      int with_bias, int S, void* qkv0, void* qk, hipStream_t st) {
  using K = Fm4Cfg<C, HEADS, G>;
  const Fm3PackLayout l3 = fm3_pack_layout(C, fm_pack_layout(C).bytes);
  Fm3Args a;
  a.x = (const bf16*)x; a.v = (bf16*)v; a.part = part; a.mean = mean; a.rstd = rstd;
  a.wt = (const unsigned char*)pack + l3.w1f;
  a.qkv0 = (bf16*)qkv0; a.qk = (bf16*)qk;
  a.B = s->B; a.H = s->H; a.W = s->W; a.with_bias = with_bias;
  a.tiles_x = s->W / 32; a.tiles_y = s->H / 8; a.S = S;
  { const char* e = MI_ENV(MI_FM_DEBUG); a.dbg = e ? atoi(e) : 0; }
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  MI_CHECK_HIP(hipGetDevice(&dev));
  const unsigned bit = 1u << (dev & 31);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)fm4_fwd_kernel<C, HEADS, G, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)K::LDS4_BYTES));
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)fm4_fwd_kernel<C, HEADS, G, SAVE, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)K::LDS4_BYTES));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  const double N = (double)s->H * s->W * s->B;
  ProfScope ps(st, K_MDTA_FUSED_A, (2.0 + (SAVE ? 5.0 : 0.0)) * C * N * 2.0,
               2.0 * N * (3.0 * C * C + (double)C * K::c) + 2.0 * N * 9.0 * 3.0 * C);
  if (a.dbg & 0x1000)
    hipLaunchKernelGGL((fm4_fwd_kernel<C, HEADS, G, SAVE, true>), dim3((unsigned)(s->B * S)), dim3(64 * K::NW), K::LDS4_BYTES, st, a);
  else
    hipLaunchKernelGGL((fm4_fwd_kernel<C, HEADS, G, SAVE>), dim3((unsigned)(s->B * S)), dim3(64 * K::NW), K::LDS4_BYTES, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
static bool fm_use_v2() { const char* e = MI_ENV(MI_FM_CFG); return e && strstr(e, "v2"); }

}  // namespace mi

using namespace mi;

extern "C" int mi_mdta_fused_ok(const mi_mdta_shape* s) { return (fm_kind(s) != FM_NONE && !MI_ENV(MI_NO_FUSED_MDTA)) ? 1 : 0; }
// Covered AND expected to beat the unfused chain: pass A runs one persistent 8-wave workgroup per CU, so it needs (close to) a
// workgroup for every CU - B * splits >= 192 of the 256.  Measured (profiles/r03_c_*): bs 8 at 96 x 128^2 gives 128 workgroups and
// 0.75x of the chain; bs 8 at 256^2 and bs 25 / 32 everywhere fill the chip (1.06 - 1.18x).
extern "C" int mi_mdta_fused_pays(const mi_mdta_shape* s) {
  if (!mi_mdta_fused_ok(s)) return 0;
  return (int64_t)s->B * fm_splits(s, 8) >= 192 ? 1 : 0;
}

extern "C" size_t mi_mdta_fused_pack_bytes(const mi_mdta_shape* s) {
  if (fm_kind(s) == FM_NONE) return 0;
  return fm3_pack_layout(s->C, fm_pack_layout(s->C).bytes).bytes;
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
  {                                                       // the fourth form's sections of the same blob
    const Fm3PackLayout l3 = fm3_pack_layout(s->C, l.bytes);
    Fm3PackArgs a3;
    a3.ln_w = ln_w; a3.ln_b = ln_b; a3.qkv_w = p->qkv_w; a3.qkv_b = p->qkv_b; a3.dw_w = p->dw_w; a3.dw_b = p->dw_b;
    a3.w1f = (bf16*)(pk + l3.w1f); a3.tb = (u16*)(pk + l3.tb);
    a3.C = s->C; a3.nchunk = l.nchunk; a3.nks = s->C / 32 + 1;
    hipLaunchKernelGGL(fm3_pack_kernel, dim3(64), dim3(256), 0, st, a3);
    MI_LAUNCH_CHECK();
  }
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
  int part_mult = 1;
  if (fm_use_v2() || k != FM_48_1) {                      // round-3 form (depthwise conv on the VALU): MI_FM_CFG=v2, and the shapes the fourth form does not cover yet
    if (k == FM_48_1) MI_TRY((fm_launch<48, 1, 8, 8>(s, l, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, st)));
    else if (k == FM_96_2) MI_TRY((fm_launch<96, 2, 8, 8>(s, l, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, st)));
    else MI_TRY((fm_launch<96, 1, 8, 8>(s, l, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, st)));
  } else {                                                // fourth form: the conv on the matrix cores, grouped phases, wave-local Gram
    // one group of all 9 chunks: 308 us at 256^2, bs 32; three double-buffered groups of 3 (GEMM1 of group g + 1 inside the
    // wave-local interval of group g) were measured at 418 us - two more barriers per tile cost more than the overlap buys
    MI_TRY((fm4_launch<48, 1, 9, false>(s, pack, x, w.v, w.part, mean, rstd, ln_with_bias, S, nullptr, nullptr, st)));
    part_mult = 8;
  }
  {
    ProfScope ps(st, K_GRAM_REDUCE, 4.0 * s->B * (S + 1) * ((double)s->C * (s->C / s->heads) + 2.0 * s->C), 0.0);
    const int cc = s->C / s->heads;
    hipLaunchKernelGGL(fm_reduce_kernel, dim3(cdiv(cc * cc + 2 * cc, 256), s->B * s->heads), dim3(256), 0, st, w.part, w.graw, w.ss,
                       S * part_mult, s->C, s->heads);
    MI_LAUNCH_CHECK();
  }
  MI_TRY(launch_attn_fold(w.graw, w.ss, p->temperature, p->proj_w, w.P, w.A, w.nrm, w.M, s->B, s->C, s->heads, st, w.Mb, nullptr));
  const int64_t N = (int64_t)s->H * s->W;
  mi_pw_desc d;
  memset(&d, 0, sizeof(d));
  d.x1 = w.v; d.x1_bs = (int64_t)s->C * N; d.k1 = s->C;
  d.w = w.M; d.w_sm = s->C; d.w_sk = 1; d.w_bs = (int64_t)s->C * s->C;
  d.w_b16 = w.Mb; d.w_b16_sm = s->C;
  d.bias = p->proj_b;
  d.r = residual; d.r_bs = (int64_t)s->C * N;
  d.y = out; d.y_bs = (int64_t)s->C * N;
  d.m = s->C; d.n = N; d.batch = s->B; d.groups = 1; d.dtype = s->dtype;
  return mi_pw_gemm(&d, w.pw_ws, stream);
}
