// Backward tail of a half-block in one launch.
//
// A Restormer half-block is  out = x + F(LN(x))  with F starting in a 1x1 conv  h = W LN(x)  (qkv, Restormer.py:105,115;
// project_in, Restormer.py:82,89; LN = WithBias_LayerNorm, Restormer.py:52-64).  Given dY = dL/dh the backward still owes
//     dW  = dY LN(x)^T                       (weight gradient,  [M, C], summed over batch and pixels)
//     dxn = W^T dY                           (input gradient of the conv)
//     dx  = LNbackward(dxn; x, mean, rstd) + dres      (dres = gradient arriving over the residual connection)
//     dgamma, dbeta                          (LayerNorm parameter gradients)
// The unfused chain runs a Gram kernel, a GEMM and the LayerNorm backward: dY is streamed twice, dxn written and read, LN(x)
// stored by the forward and read here - (2M + 6C) planes of traffic per pixel.  This kernel streams dY, x and dres once and
// writes dx: (M + 3C) planes, and LN(x) no longer has to be saved by the forward.
//
// With xh = (x - mean) rstd (no affine):  LN(x) = gamma xh + beta, so with  G = dY xh^T  and  S = rowsum(dY):
//     dW[m,c] = gamma_c G[m,c] + beta_c S[m],    dgamma_c = sum_m W[m,c] G[m,c],    dbeta_c = sum_m W[m,c] S[m]
// (dgamma_c = sum_p dxn[c,p] xh[c,p] with dxn = W^T dY).  The streaming kernel therefore only accumulates G and S; a tiny
// finishing kernel turns them into the three parameter gradients.
//
// Work split: a workgroup owns 64-pixel tiles (persistent loop), wave w owns dY rows [16 MPW w, 16 MPW (w+1)):
//   * G: the wave's rows x all C columns stay in its accumulators for the whole kernel (MFMA, contraction over pixels);
//   * dxn: each wave contracts over ITS rows only (W^T fragments in registers, dY transposed by ds_read_b64_tr_b16 from the
//     wave-private patch) and adds its partial tile into a shared fp32 LDS accumulator with ds_add_f32;
//   * after a barrier all lanes run the LayerNorm backward on the accumulator (8 or 4 lanes per pixel), transpose the result
//     through LDS and store 128-byte rows of dx.
// The next tile's first dY half, x, dres and statistics are in flight while a tile is computed, and the LayerNorm phase and
// store of tile i-1 run inside tile i behind the request for its second dY half (software pipeline across tiles).
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "internal.h"
#include "fused_common.h"

namespace mi {
namespace {
using namespace fz;

#ifndef BT_PS_ELEMS
#define BT_PS_ELEMS 72
#endif
constexpr int BT_PS = BT_PS_ELEMS;   // bf16 row stride of every 64-pixel LDS tile (64 + pad: conflict-free 8-byte and transposed reads)

struct BtArgs {
  const bf16* dy;      // [B][M][N]
  const bf16* x;       // [B][C][N]   LayerNorm input
  const bf16* dres;    // [B][C][N]   or null
  const float* mean;   // [B][N]
  const float* rstd;   // [B][N]
  const float* w;      // [M][C]
  const float* gamma;  // [C]
  bf16* dx;            // [B][C][N]
  float* gpart;        // [grid][M][C + 1]: G and, in column C, the row sums S
  int M, mpad;
  int64_t N;
  int tiles_per_image;
  int ntiles;
  int dbg;             // MI_BT_DEBUG ablation mask (timing only: results are wrong when set)
};

template <int C_, int NW_, int MPW_>
struct BtCfg {
  static constexpr int C = C_, NW = NW_, MPW = MPW_;
  static constexpr int CT = C / 16;
  static constexpr int ROWS = 16 * MPW;          // dY rows of one wave
  static constexpr int KS = (ROWS + 31) / 32;    // 32-deep steps of the dxn contraction over those rows
  static constexpr int PROWS = 32 * KS;          // patch rows (rows past ROWS stay zero)
  static constexpr int NT = 64 * NW;
  static constexpr int XS = 68;                  // fp32 stride of the dxn accumulator [channel][64 pixels + pad]

  static constexpr int LPP = NT / 64;            // lanes per pixel in the LayerNorm-backward phase
  static constexpr int CPL = C / LPP;            // channels per lane there
  static constexpr int ITEMS = C * 16 / NT;      // (channel row, 4-pixel quad) items per lane when dx is stored
  static constexpr int XP = C / 8;               // 1 KiB LDS-DMA pieces (8 rows x 64 px) of the x tile; as many of dres
  static constexpr int PPW = 2 * XP / NW;        // pieces per wave
  static constexpr int PATCH_E = PROWS * BT_PS;
  static constexpr int TILE_B = C * 128;         // bytes of one unpadded, chunk-swizzled [C][64] bf16 tile
  static constexpr int OFF_XH = NW * PATCH_E * 2;
  static constexpr int OFF_DR = OFF_XH + 2 * TILE_B;
  static constexpr int OFF_ACC = OFF_DR + 2 * TILE_B;
  static constexpr int OFF_GAM = OFF_ACC + C * XS * 4;
  static constexpr int OFF_ST = OFF_GAM + C * 4;
  static constexpr int BYTES = OFF_ST + NW * 512;
  static_assert(C % 16 == 0 && (LPP == 4 || LPP == 8) && C % LPP == 0 && (C * 16) % NT == 0 && (2 * XP) % NW == 0,
                "unsupported shape");
  static_assert(C / 16 <= NW, "one wave per 16-channel output tile of the dxn product");
  static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// element index of (channel c, pixel px) in an unpadded [C][64] bf16 tile whose 16-byte chunks are XOR-swizzled by the row:
// 128-byte rows would put every row on the same banks; with chunk position (px/8) ^ ((c/2) & 7) the 8-byte operand reads of 16
// consecutive rows, the 2-byte reads of the LayerNorm phase and the row-wise store all spread over the banks, and the tile can
// be filled by LDS-DMA (which writes a wave's 1 KiB linearly) by permuting the SOURCE chunks instead.
__device__ __forceinline__ int bt_swz(int c, int px) { return c * 64 + ((((px >> 3) ^ (c >> 1)) & 7) << 3) + (px & 7); }

template <int C, int NW, int MPW>
__global__ __launch_bounds__(64 * NW, (BtCfg<C, NW, MPW>::BYTES <= 80 * 1024 ? 2 : 1)) void bt_kernel(BtArgs a) {
  using K = BtCfg<C, NW, MPW>;
  constexpr int CT = K::CT, KS = K::KS, PS = BT_PS, XS = K::XS, NT = K::NT, LPP = K::LPP, CPL = K::CPL, ITEMS = K::ITEMS;
  constexpr int XP = K::XP, PPW = K::PPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);               // wave index in a scalar register: everything derived from
                                                                        // it (row range, patch, DMA pieces, channel tile) is scalar too
  bf16* const patch = reinterpret_cast<bf16*>(lds) + wv * K::PATCH_E;
  float* const acc = reinterpret_cast<float*>(lds + K::OFF_ACC);
  float* const st = reinterpret_cast<float*>(lds + K::OFF_ST + wv * 512);   // this wave's copy of mean[64], rstd[64]
  const int m0 = wv * K::ROWS;
  const bool active = m0 < a.M;                                         // wave-uniform
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const unsigned un = (unsigned)a.N;

  for (int i = lane; i < K::PATCH_E / 8; i += 64) reinterpret_cast<u32x4*>(patch)[i] = zero4;

  // (W diag(gamma))^T fragments (B operand of the dxn product) of THIS wave's channel tile, ct = wv (waves past CT take no part
  // in the product): column c = 16 wv + li, k slots = the rows of owner wave o in its patch's slot order.
  const bool dxn_wave = wv < CT;                                        // wave-uniform
  s16x8 wt[NW][KS];
#pragma unroll
  for (int o = 0; o < NW; ++o)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ml = 32 * ks + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4), m = o * K::ROWS + ml;
        const bool ok = dxn_wave && ml < K::ROWS && m < a.M;
        const int cc = dxn_wave ? 16 * wv + li : 0;
        const float v = a.w[(ok ? m : 0) * C + cc] * a.gamma[cc];       // gamma folded in: the product is g = gamma dxn
        wt[o][ks][j] = ok ? bf_bits(v) : (short)0;
      }
  f32x4 G[MPW][CT];
  float S[MPW];
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    S[i] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) G[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // dY rows travel through registers in two 32-pixel halves: while one half is computed the next is in flight.
  // x, dres and the statistics of the NEXT tile go straight to LDS (global_load_lds), no registers involved.
  u32x4 R[MPW];
  const int rl = lane >> 2;                                            // row inside a 16-row fragment, 64 bytes (4 lanes) per row
  auto issue_dy = [&](int tile, int h) {
    const int b = tile / a.tiles_per_image, p0 = (tile - b * a.tiles_per_image) * 64 + 32 * h;
    int ln = lane;                                                      // (opaque copy: see wgrad_half)
    asm volatile("" : "+v"(ln));
    const int rl = ln >> 2;
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      const int lim = a.M - 1 - (m0 + 16 * j);                          // last valid row of this fragment (wave-uniform)
      if (lim >= 0) {
        const bf16* base = a.dy + ((int64_t)b * a.M + m0 + 16 * j) * a.N + p0;
        const unsigned off = (unsigned)(rl < lim ? rl : lim) * un + 8u * (ln & 3);
        // (default cache policy: the two 32-pixel halves of a tile share 128-byte lines - non-temporal loads fetched them twice,
        //  22.4 -> 27.0 ms per step; the LDS-DMA loads of x / dres measured neutral: profiles/r04_d_nontemporal_ab.txt)
        R[j] = *reinterpret_cast<const u32x4*>(base + off);
      }
    }
  };
  auto stage_dy = [&](int h) {
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      const int lim = a.M - 1 - (m0 + 16 * j);
      *reinterpret_cast<u32x4*>(&patch[(16 * j + rl) * PS + 32 * h + 8 * (lane & 3)]) = rl <= lim ? R[j] : zero4;
    }
  };
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue_x = [&](int tile, int nb) {
    const int b = tile / a.tiles_per_image, p0 = (tile - b * a.tiles_per_image) * 64;
    const int64_t img = (int64_t)b * C * a.N + p0;
    int ln = lane;                                                      // (opaque copy: hoisted, these per-lane address terms
    asm volatile("" : "+v"(ln));                                        //  were spilled and reloaded behind vmcnt(0) waits)
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int pid = wv + NW * k;                                      // wave-uniform
      const bool isx = pid < XP;
      const int u = isx ? pid : pid - XP, c = 8 * u + (ln >> 3), kk = ((ln & 7) ^ (c >> 1)) & 7;
      if (isx || a.dres) {
        const bf16* src = (isx ? a.x : a.dres) + img + (unsigned)c * un + 8u * kk;
        unsigned char* dst = lds + (isx ? K::OFF_XH : K::OFF_DR) + nb * K::TILE_B + u * 1024;
        __builtin_amdgcn_global_load_lds((const void*)src, (lds_ptr)dst, 16, 0, 0);
      }
    }
    const int64_t so = (int64_t)b * a.N + p0 + ln;
    __builtin_amdgcn_global_load_lds((const void*)(a.mean + so), (lds_ptr)(lds + K::OFF_ST + wv * 512), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const void*)(a.rstd + so), (lds_ptr)(lds + K::OFF_ST + wv * 512 + 256), 4, 0, 0);
  };
  // One 32-pixel half of the weight-gradient work: G += dY xh^T (and the row sums S).  Lane coordinates are re-derived from an
  // opaque copy of the lane id: every LDS address below is loop-invariant, and hoisting the few dozen of them out of the tile
  // loop is what pushes the kernel over its register budget.
  auto wgrad_half = [&](int h, const bf16* xh) {
    if (a.dbg & 1) return;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int li_ = ln & 15, g_ = ln >> 4;
    constexpr int GRP = MPW;                                             // row fragments held at once
    const bf16* ar0 = &patch[li_ * PS + 32 * h + 4 * g_];
    // swizzled chunk of pixel 32h + 16e + 4g in row 16ct + li: the row term (li/2) does not depend on ct
    const int f = (li_ >> 1) & 7, k0 = 4 * h + (g_ >> 1);
    const bf16* b_lo = xh + li_ * 64 + ((k0 ^ f) << 3) + 4 * (g_ & 1);
    const bf16* b_hi = xh + li_ * 64 + (((k0 + 2) ^ f) << 3) + 4 * (g_ & 1);
#pragma unroll
    for (int i0 = 0; i0 < MPW; i0 += GRP) {
      s16x8 A[GRP];
#pragma unroll
      for (int i = 0; i < GRP; ++i) {
        const bf16* ar = ar0 + 16 * (i0 + i) * PS;
        A[i] = cat8(*reinterpret_cast<const s16x4*>(ar), *reinterpret_cast<const s16x4*>(ar + 16));
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sm += bf_s(A[i][j]);
        S[i0 + i] += sm;
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const s16x8 Bx = cat8(*reinterpret_cast<const s16x4*>(b_lo + ct * 16 * 64), *reinterpret_cast<const s16x4*>(b_hi + ct * 16 * 64));
#pragma unroll
        for (int i = 0; i < GRP; ++i) G[i0 + i][ct] = mfma32(A[i], Bx, G[i0 + i][ct]);
      }
    }
  };
  // dxn = W^T dY needs the sum over ALL rows, i.e. over every wave's dY.  Two schemes lost: ds_add_f32 of per-wave partials into
  // a shared tile (the LDS float atomic measured ~190 cycles per wave-instruction on gfx950 - lane-serialised - 8x slower than
  // the rest of the kernel together) and a rotating read-add-write of slices (NW barriers per tile, 30 % of the kernel).  Now
  // wave w < CT owns output channels 16w..16w+15 for all 64 pixels: once every wave has staged its rows (one barrier) it walks
  // ALL patches - transposed reads of the others' rows, its own W^T column tile in registers - and stores the finished tile.
  auto dxn_tile = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int li_ = ln & 15, g_ = ln >> 4, qq_ = li_ >> 2, pp_ = li_ & 3;
    f32x4 d[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) d[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16* tr0 = reinterpret_cast<const bf16*>(lds) + (4 * g_ + qq_) * PS + 4 * pp_;
#pragma unroll
    for (int o = 0; o < NW; ++o) {
      if (o * K::ROWS < a.M) {                                           // (uniform) owners past M hold no rows
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s16x4 lo[4], hi[4];
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            lo[nt] = tr_b16(tr0 + o * K::PATCH_E + (32 * ks) * PS + 16 * nt);
            hi[nt] = tr_b16(tr0 + o * K::PATCH_E + (32 * ks + 16) * PS + 16 * nt);
          }
          lds_wait(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2], lo[3], hi[3]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) d[nt] = mfma32(cat8(lo[nt], hi[nt]), wt[o][ks], d[nt]);
        }
      }
    }
    float* const ap0 = &acc[(16 * wv + li_) * XS + 4 * g_];              // 4 consecutive pixels of channel 16 wv + li
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(ap0 + 16 * nt) = d[nt];
  };

  // LayerNorm backward on a finished dxn tile, result into the dres tile in place.  A lane takes TWO adjacent pixels of its
  // channels (c = sub + L2 j over the L2 = 2 LPP lanes of the pair): every LDS access is then a bf16 pair (4 bytes) or an fp32
  // pair (8 bytes) - half the LDS instructions of the one-pixel form, which bounded this phase (48 two-byte accesses per lane).
  auto ln_phase = [&](const bf16* xh, bf16* dr, f32x2 rstd_p) {
    if (a.dbg & 8) return;
    int tt = t;
    asm volatile("" : "+v"(tt));
    constexpr int L2 = 2 * LPP, CPL2 = C / L2;
    static_assert(C % L2 == 0 && (L2 == 8 || L2 == 16), "pixel-pair LayerNorm phase");
    const int px0 = 2 * (tt / L2), sub = tt % L2;
    // swizzled slot of (c = sub + L2 j, px0): c >> 1 = (sub >> 1) + (L2 / 2) j; with L2 = 16 the j term vanishes under & 7, with
    // L2 = 8 (sub >> 1 < 4) it toggles bit 2 for odd j
    int base[2];
    base[0] = sub * 64 + ((((px0 >> 3) ^ (sub >> 1)) & 7) << 3) + (px0 & 7);
    base[1] = sub * 64 + ((((px0 >> 3) ^ (sub >> 1) ^ (L2 == 8 ? 4 : 0)) & 7) << 3) + (px0 & 7);
    const float* const ap = &acc[sub * XS + px0];
    f32x2 gv[CPL2], xv[CPL2], s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < CPL2; ++j) {
      gv[j] = *reinterpret_cast<const f32x2*>(ap + L2 * j * XS);          // g = gamma dxn (gamma rides in the W^T fragments)
      const unsigned xb = *reinterpret_cast<const unsigned*>(xh + base[j & 1] + L2 * j * 64);
      xv[j] = (f32x2){bf_lo(xb), bf_hi(xb)};
      s1 += gv[j];
      s2 += gv[j] * xv[j];
    }
    // sum over the L2 consecutive lanes of this pixel pair, every lane gets the totals: DPP only (quad swaps, the half-row
    // mirror adds the other quad's total, the row mirror the other half row's) - no ds_bpermute round trips
    auto dpp = [](float v, auto ctrl) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float u1 = s1[e], u2 = s2[e];
      u1 += dpp(u1, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
      u2 += dpp(u2, std::integral_constant<int, 0xB1>{});
      u1 += dpp(u1, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
      u2 += dpp(u2, std::integral_constant<int, 0x4E>{});
      u1 += dpp(u1, std::integral_constant<int, 0x141>{});  // row_half_mirror
      u2 += dpp(u2, std::integral_constant<int, 0x141>{});
      if (L2 == 16) {
        u1 += dpp(u1, std::integral_constant<int, 0x140>{});  // row_mirror
        u2 += dpp(u2, std::integral_constant<int, 0x140>{});
      }
      s1[e] = u1 * (1.0f / C); s2[e] = u2 * (1.0f / C);
    }
#pragma unroll
    for (int j = 0; j < CPL2; ++j) {
      unsigned* const slot = reinterpret_cast<unsigned*>(dr + base[j & 1] + L2 * j * 64);
      const unsigned rb = *slot;
      const f32x2 o = rstd_p * (gv[j] - s1 - xv[j] * s2);
      *slot = pack_bf2(o[0] + bf_lo(rb), o[1] + bf_hi(rb));
    }
  };
  auto store_phase = [&](const bf16* dr, int b, int p0) {
    if (a.dbg & 16) return;
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int c0 = tt >> 4, q = tt & 15;                                // rows c0 + (NT/16) i: their swizzle term equals that of c0
    static_assert((NT / 32) % 8 == 0, "row step of the store loop must keep the swizzle term");
    const bf16* src = dr + c0 * 64 + ((((q >> 1) ^ (c0 >> 1)) & 7) << 3) + 4 * (q & 1);
    bf16* const ob = a.dx + (int64_t)b * C * a.N + p0 + (unsigned)c0 * un + 4u * q;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
      MI_STREAM_ST(reinterpret_cast<u32x2*>(ob + (unsigned)(NT / 16 * i) * un), *reinterpret_cast<const u32x2*>(src + NT / 16 * i * 64));
  };

  // Tile loop, software-pipelined across tiles: the LayerNorm phase and the dx store of tile i-1 run inside tile i, between the
  // request for the second half of tile i's dY rows and its first use - the only place where a load had no lead.
#ifdef BT_STAMP   // variant build (tools/bt_stamps.py): shader-clock time per phase, summed per wave, written in place of [G | S]
  unsigned long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = fm_clock();
#define BT_TICK(i) { const unsigned long long tn_ = fm_clock(); tk[i] += tn_ - t_prev; t_prev = tn_; }
#else
#define BT_TICK(i)
#endif
  int tile = blockIdx.x;
  if (tile < a.ntiles) {
    if (active) issue_dy(tile, 0);
    issue_x(tile, 0);
  }
  __syncthreads();
  int buf = 0, pb = 0, pp0 = 0;
  bool have_prev = false;
  f32x2 rstd_prev = {0.f, 0.f};
  for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
    const int b = tile / a.tiles_per_image, p0 = (tile - b * a.tiles_per_image) * 64;
    const int nxt = tile + gridDim.x;
    bf16* const xh = reinterpret_cast<bf16*>(lds + K::OFF_XH + buf * K::TILE_B);
    bf16* const dr = reinterpret_cast<bf16*>(lds + K::OFF_DR + buf * K::TILE_B);
    bf16* const xh_prev = reinterpret_cast<bf16*>(lds + K::OFF_XH + (buf ^ 1) * K::TILE_B);
    bf16* const dr_prev = reinterpret_cast<bf16*>(lds + K::OFF_DR + (buf ^ 1) * K::TILE_B);
    // ---- this wave's DMA pieces and dY half have landed: dY -> patch, x -> xh = (x - mean) rstd in place
    BT_TICK(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BT_TICK(1);
    if (active) stage_dy(0);
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int pid = wv + NW * k;
      if (pid < XP) {
        const int c = 8 * pid + (lane >> 3), kk = ((lane & 7) ^ (c >> 1)) & 7;
        u32x4* slot = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(xh) + pid * 1024 + lane * 16);
        const u32x4 v = *slot;
        const f32x4 m_lo = *reinterpret_cast<const f32x4*>(&st[8 * kk]), m_hi = *reinterpret_cast<const f32x4*>(&st[8 * kk + 4]);
        const f32x4 r_lo = *reinterpret_cast<const f32x4*>(&st[64 + 8 * kk]), r_hi = *reinterpret_cast<const f32x4*>(&st[64 + 8 * kk + 4]);
        u32x4 o;
        o[0] = pack_bf2((bf_lo(v[0]) - m_lo[0]) * r_lo[0], (bf_hi(v[0]) - m_lo[1]) * r_lo[1]);
        o[1] = pack_bf2((bf_lo(v[1]) - m_lo[2]) * r_lo[2], (bf_hi(v[1]) - m_lo[3]) * r_lo[3]);
        o[2] = pack_bf2((bf_lo(v[2]) - m_hi[0]) * r_hi[0], (bf_hi(v[2]) - m_hi[1]) * r_hi[1]);
        o[3] = pack_bf2((bf_lo(v[3]) - m_hi[2]) * r_hi[2], (bf_hi(v[3]) - m_hi[3]) * r_hi[3]);
        *slot = o;
      } else if (!a.dres) {
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(dr) + (pid - XP) * 1024 + lane * 16) = zero4;
      }
    }
    const f32x2 rstd_p = *reinterpret_cast<const f32x2*>(&st[64 + 2 * (t / (2 * LPP))]);   // (this lane's pixel pair) read before the next tile's statistics are requested
    BT_TICK(2);
    __syncthreads();
    BT_TICK(3);
    if (active) issue_dy(tile, 1);
    // ---- tile i-1: LayerNorm backward, transpose through its dres tile, store dx
    if (have_prev) ln_phase(xh_prev, dr_prev, rstd_prev);
    BT_TICK(4);
    __syncthreads();
    BT_TICK(5);
    if (have_prev) store_phase(dr_prev, pb, pp0);
    BT_TICK(6);
    if (active && !(a.dbg & 32)) {
      wgrad_half(0, xh);
      BT_TICK(7);
      stage_dy(1);
      wave_sync();
      BT_TICK(8);
      if (nxt < a.ntiles) issue_dy(nxt, 0);
      wgrad_half(1, xh);
    }
    BT_TICK(9);
    __syncthreads();                                                     // every wave's dY rows of this tile are in LDS
    BT_TICK(10);
    // every wave is past the store of tile i-1 now too: its x / dres buffers can take tile i+1
    if (nxt < a.ntiles) issue_x(nxt, buf ^ 1);
    if (dxn_wave && !(a.dbg & 34)) dxn_tile();
    BT_TICK(11);
    __syncthreads();                                                     // dxn tile complete; the patches may be overwritten
    have_prev = true; pb = b; pp0 = p0; rstd_prev = rstd_p;
  }
  if (have_prev) {                                               // the last tile's LayerNorm phase and store
    const int lb = buf ^ 1;
    bf16* const xh_l = reinterpret_cast<bf16*>(lds + K::OFF_XH + lb * K::TILE_B);
    bf16* const dr_l = reinterpret_cast<bf16*>(lds + K::OFF_DR + lb * K::TILE_B);
    ln_phase(xh_l, dr_l, rstd_prev);
    __syncthreads();
    store_phase(dr_l, pb, pp0);
  }
#ifdef BT_STAMP
  if (lane == 0) {
    float* const gp = a.gpart + (int64_t)blockIdx.x * a.M * (C + 1) + wv * 16;
#pragma unroll
    for (int i = 0; i < 12; ++i) gp[i] = (float)tk[i];
  }
  return;
#endif
  // ---- this workgroup's partial [G | S]: one [M][C + 1] matrix, the row sums in column C
  if (active) {
    float* const gp = a.gpart + (int64_t)blockIdx.x * a.M * (C + 1);
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + 16 * i + 4 * g + r;
          if (m < a.M) gp[(int64_t)m * (C + 1) + 16 * ct + li] = G[i][ct][r];
        }
      float sv = S[i];
      sv += __shfl_xor(sv, 16);
      sv += __shfl_xor(sv, 32);
      const int m = m0 + 16 * i + li;
      if (g == 0 && m < a.M) gp[(int64_t)m * (C + 1) + C] = sv;
    }
  }
}

// The per-workgroup partials [nparts][M][C + 1] -> dW = gamma G + beta S (into dw, accumulating or not) and, per row m, the
// contributions W[m,c] G[m,c] | W[m,c] S[m] to dgamma | dbeta (gb[M][2C]; a row sum - deferred when the caller's step records
// deferred sums - finishes them).  One workgroup per row m: 128 column slots x 8 phases over the partial copies, eight loads in
// flight per thread.  Replaces the two-stage row sum (49 k workgroups of one load per thread at M = 510) + bt_finish_kernel:
// 48 -> 16 us behind the largest tail (profiles/r04_g_bwd_tail_ablation.txt).
__global__ __launch_bounds__(1024) void bt_finish2_kernel(const float* __restrict__ gpart, int nparts, const float* __restrict__ w,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ dw, float* __restrict__ gb, int M, int C, int accumulate) {
  __shared__ float sm[8][128];
  const int m = blockIdx.x, cx = threadIdx.x & 127, ph = threadIdx.x >> 7;
  const int64_t mc = (int64_t)M * (C + 1);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cx <= C) {
    const float* p = gpart + (int64_t)m * (C + 1) + cx;
    int r = ph;
    for (; r + 56 < nparts; r += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += p[(int64_t)(r + 8 * u) * mc];
    }
    for (; r < nparts; r += 8) a[0] += p[(int64_t)r * mc];
  }
  sm[ph][cx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  float t = 0.f;
  if (ph == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[k][cx];
  }
  __syncthreads();
  if (ph == 0) sm[0][cx] = t;
  __syncthreads();
  if (ph == 0 && cx < C) {
    const float gg = sm[0][cx], ss = sm[0][C];
    float* o = dw + (int64_t)m * C + cx;
    *o = (accumulate ? *o : 0.f) + gamma[cx] * gg + (beta ? beta[cx] * ss : 0.f);
    const float ww = w[(int64_t)m * C + cx];
    gb[(int64_t)m * 2 * C + cx] = ww * gg;
    gb[(int64_t)m * 2 * C + C + cx] = ww * ss;
  }
}

struct BtPlan { int C, NW, MPW, grid, mpad; size_t lds; };
static bool bt_plan(int M, int C, BtPlan* p) {
  int nw, mpw;
  const int mt = (M + 15) / 16;
  if (C == 96) { nw = 8; mpw = mt <= 24 ? 3 : 4; if (mt > 32) return false; }
  else if (C == 48) { nw = 4; mpw = mt <= 12 ? 3 : 4; if (mt > 16) return false; }
  else return false;
  p->C = C; p->NW = nw; p->MPW = mpw; p->mpad = 16 * nw * mpw;
  p->lds = C == 96 ? (mpw == 3 ? BtCfg<96, 8, 3>::BYTES : BtCfg<96, 8, 4>::BYTES)
                   : (mpw == 3 ? BtCfg<48, 4, 3>::BYTES : BtCfg<48, 4, 4>::BYTES);
  p->grid = C == 96 ? 256 : 512;
  return true;
}
}  // namespace

bool bwd_tail_ok(int M, int C, int64_t N, int dtype) {
  BtPlan p;
  return dtype == MI_BF16 && N > 0 && N % 64 == 0 && M >= 16 && bt_plan(M, C, &p);
}
// Where the tail beats the three kernels it replaces (tools/bench_tail.py, profiles/r02_d_bwd_tail_bs32.txt): every covered
// shape, 1.06-1.30x.  (The 4-fragments-per-wave form, C = 96 and M > 384, runs at the register limit: with its reduction steps
// unrolled it spilled 13 registers and lost, 0.92x; rolled it wins 1.10x.  MI_BT_WIDE=0 switches it off.)
bool bwd_tail_pays(int M, int C) {
  BtPlan p;
  if (!bt_plan(M, C, &p)) return false;
  const char* e = MI_ENV(MI_BT_WIDE);                                 // A/B switch, read per call: 0 keeps the 4-fragment form off
  return !(C == 96 && p.MPW == 4) || !(e && atoi(e) == 0);
}
// partial [G | S] per workgroup, its sum, and the two-stage row reduction's scratch
size_t bwd_tail_workspace(int M, int C) {
  BtPlan p;
  if (!bt_plan(M, C, &p)) return 0;
  const size_t mc = (size_t)M * (C + 1);
  return align_up((size_t)p.grid * mc * 4, 256) + align_up(mc * 4, 256) + align_up((size_t)REDUCE_GROUPS * mc * 4, 256);
}

template <int C, int NW, int MPW>
static int bt_launch(const BtArgs& a, int grid, hipStream_t st) {
  using K = BtCfg<C, NW, MPW>;
  // raising the dynamic-LDS limit is a per-function, per-DEVICE setting: remembered per device (one process may drive several
  // GPUs), not per process
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  MI_CHECK_HIP(hipGetDevice(&dev));
  const unsigned bit = 1u << (dev & 31);
  if (!(attr_set.load(std::memory_order_relaxed) & bit)) {
    MI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bt_kernel<C, NW, MPW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, K::BYTES));
    attr_set.fetch_or(bit, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((bt_kernel<C, NW, MPW>), dim3(grid), dim3(64 * NW), K::BYTES, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int launch_bwd_tail(const void* dy, int M, const void* x, int C, const void* dres, const float* mean, const float* rstd,
                    const float* w, const float* gamma, const float* beta, void* dx, float* dw, float* dgamma, float* dbeta,
                    int B, int64_t N, int accumulate, void* ws, hipStream_t st) {
  BtPlan p;
  MI_CHECK_ARG(bt_plan(M, C, &p) && N % 64 == 0, "bwd_tail: unsupported shape M=%d C=%d N=%lld", M, C, (long long)N);
  MI_CHECK_ARG(dy && x && mean && rstd && w && gamma && dx && dw && dgamma && ws, "bwd_tail: null pointer");
  MI_CHECK_ARG(aligned16(dy) && aligned16(x) && aligned16(dx) && (!dres || aligned16(dres)) && aligned16(mean) && aligned16(rstd),
               "bwd_tail: operands must be 16-byte aligned");
  const size_t mc = (size_t)M * (C + 1);
  Carver cv(ws);
  float* gpart = cv.take<float>((size_t)p.grid * mc * 4);
  float* gsum = cv.take<float>(mc * 4);
  float* tmp = cv.take<float>((size_t)REDUCE_GROUPS * mc * 4);
  BtArgs a;
  a.dy = (const bf16*)dy; a.x = (const bf16*)x; a.dres = (const bf16*)dres; a.mean = mean; a.rstd = rstd; a.w = w;
  a.gamma = gamma; a.dx = (bf16*)dx; a.gpart = gpart; a.M = M; a.mpad = p.mpad; a.N = N;
  a.tiles_per_image = (int)(N / 64); a.ntiles = B * a.tiles_per_image;
  { const char* e = MI_ENV(MI_BT_DEBUG); a.dbg = e ? atoi(e) : 0; }
  const int grid = a.ntiles < p.grid ? a.ntiles : p.grid;
  {
    const double px = (double)B * N;
    ProfScope ps(st, K_BWD_TAIL, ((double)M + 3.0 * C) * px * 2 + 8.0 * px, 4.0 * M * C * px);
    if (C == 96 && p.MPW == 4) MI_TRY((bt_launch<96, 8, 4>(a, grid, st)));
    else if (C == 96) MI_TRY((bt_launch<96, 8, 3>(a, grid, st)));
    else if (p.MPW == 4) MI_TRY((bt_launch<48, 4, 4>(a, grid, st)));
    else MI_TRY((bt_launch<48, 4, 3>(a, grid, st)));
  }
  // partials -> dW and the per-row dgamma | dbeta contributions; their row sum is a parameter-gradient sum like any other
  // (deferred into the step's flush when the caller records, launched here otherwise)
  (void)gsum;
  float* gb = accumulate ? deferred_take((size_t)M * 2 * C, st) : nullptr;
  if (!gb) gb = tmp;
  {
    ProfScope ps(st, K_BWD_TAIL_FIN, 4.0 * (grid + 2.0) * mc, (double)grid * mc);
    hipLaunchKernelGGL(bt_finish2_kernel, dim3(M), dim3(1024), 0, st, gpart, grid, w, gamma, beta, dw, gb, M, C, accumulate);
    MI_LAUNCH_CHECK();
  }
  if (dbeta) MI_TRY(launch_reduce_rows(gb, dgamma, M, 2 * C, 2 * C, accumulate, 1.0f, st, nullptr, dbeta, C));
  else MI_TRY(launch_reduce_rows(gb, dgamma, M, C, 2 * C, accumulate, 1.0f, st, nullptr, nullptr, 0));
  return MI_OK;
}

}  // namespace mi

// C-ABI: the tail by itself (tests, and callers that build their own half-blocks)
extern "C" int mi_bwd_tail_ok(int M, int C, int64_t N, int dtype) { return mi::bwd_tail_ok(M, C, N, dtype) ? 1 : 0; }
extern "C" size_t mi_bwd_tail_workspace(int M, int C) { return mi::bwd_tail_workspace(M, C); }
extern "C" int mi_bwd_tail(const void* dy, int M, const void* x, int C, const void* dres, const float* mean, const float* rstd,
                           const float* w, const float* gamma, const float* beta, void* dx, float* dw, float* dgamma,
                           float* dbeta, int B, int64_t N, int accumulate, int dtype, void* ws, void* stream) {
  MI_CHECK_ARG(dtype == MI_BF16, "bwd_tail: bf16 activations only");
  return mi::launch_bwd_tail(dy, M, x, C, dres, mean, rstd, w, gamma, beta, dx, dw, dgamma, dbeta, B, N, accumulate, ws,
                             (hipStream_t)stream);
}
