// 1x1 convolution as a classic LDS-tiled GEMM for the DEEP levels of the U-Net (K > 128 input channels: Restormer levels 3 / 4,
// 192 and 384 channels, and the wide GDFN projections there: 192 -> 1020, 384 -> 2042, 1021 -> 384; Restormer.py:82-92,105-125).
//   Y[b][m][p] = sum_k W[m][k] . X[b][k][p]  (+ bias[m]) (+ R[b][m][p])
// The wave-owned kernels of pw_gemm.hip keep a tile's weights in registers and stream X once per 48..96-row tile: right while
// the whole K fits (K <= 192), but at K >= 384 they ran at 1.5 TB/s / 420-475 TFLOP/s (profiles/r02_z_shape_table_bs32.txt) -
// neither roof.  Here a workgroup owns 128.NWM rows x 256 pixels; per 32-channel chunk it stages X [32][256] (as loaded, read
// back with ds_read_b64_tr_b16) and the weight tile as fragment-major bf16 (built from the fp32 weights on the fly, any stride:
// the transposed products of the backward pass included); every wave holds 8 m-tiles x 4 n-tiles of accumulators (128 VGPRs),
// so one k-step costs 12 KB of LDS reads per 32 MFMAs.  Same skeleton as csrc/conv3x3.hip (which reaches 0.9 PFLOP/s).
#include "common.h"
#include "fused_common.h"
#include "internal.h"

namespace mi {
using namespace fz;

constexpr int PL_XS = 272;          // LDS row stride of the staged X chunk (elements): 256 pixels + 16 (= 16 mod 128: conflict-free tr reads)

struct PlArgs {
  const bf16* x; const bf16* wp; const float* bias; const bf16* r; bf16* y;
  int64_t x_bs, r_bs, y_bs, N;
  int M, K, nchunk, tiles_n;
};

// Packed weights: [row tile (TM)][chunk (64 channels)][k-step (2)][m-tile (TM / 16)][lane][8] bf16; lane (li, g) of a fragment
// holds A[m = 16 mt + li][k0 + 4g .. 4g+3, k0 + 16 + 4g .. +3] (the k order of the transposed LDS reads of X).
struct PlPackArgs { const float* w; bf16* wp; int64_t w_sm, w_sk; int M, K, TM, nchunk, ntile; };
__global__ __launch_bounds__(256) void pl_pack_kernel(PlPackArgs a) {
  const int64_t per_tile = (int64_t)a.nchunk * 2 * (a.TM / 16) * 512;
  const int64_t total = per_tile * a.ntile;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    int64_t r = e;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int mt = (int)(r % (a.TM / 16)); r /= (a.TM / 16);
    const int ks = (int)(r % 2); r /= 2;
    const int ch = (int)(r % a.nchunk);
    const int tile = (int)(r / a.nchunk);
    const int li = lane & 15, g = lane >> 4;
    const int m = tile * a.TM + 16 * mt + li;
    const int k = ch * 64 + ks * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
    a.wp[e] = (bf16)((m < a.M && k < a.K) ? a.w[(int64_t)m * a.w_sm + (int64_t)k * a.w_sk] : 0.f);
  }
}

template <int NWM>                  // wave rows: the tile is 128 NWM rows x 256 pixels, 4 NWM waves
__global__ __launch_bounds__(256 * NWM) void pl_kernel(PlArgs a) {
  constexpr int NT = 256 * NWM, TM = 128 * NWM;
  constexpr int XV = 64 * 32;                           // 16-byte vectors of an X chunk: 64 channels x 256 pixels
  constexpr int XVT = XV / NT;
  constexpr int WV = TM * 64 / 8;                       // 16-byte vectors of a weight chunk
  constexpr int WVT = WV / NT;                          // 8
  constexpr int X_BYTES = 64 * PL_XS * 2;
  constexpr int SLAB = 16 * 68 * 4;                     // epilogue: wave-private fp32 slab [16 rows][64 px + 4]
  static_assert(4 * NWM * SLAB <= X_BYTES, "epilogue slabs live in the X region");
  extern __shared__ __attribute__((aligned(16))) unsigned char pl_lds[];
  bf16* const X = reinterpret_cast<bf16*>(pl_lds);      // [64][PL_XS]
  bf16* const Wl = reinterpret_cast<bf16*>(pl_lds + X_BYTES);   // [2 k-steps][TM / 16 m-tiles][64 lanes][8]
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wv >> 2, wn = wv & 3;                  // wave row (128 rows each) and pixel quarter (64 pixels each)
  const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  const int mt0 = blockIdx.x, b = blockIdx.z;
  const int64_t n0 = (int64_t)blockIdx.y * 256;
  const bf16* const xb = a.x + (int64_t)b * a.x_bs;
  const bf16* const wb = a.wp + (int64_t)mt0 * a.nchunk * (WV * 8);
  const int m_base = mt0 * TM;

  u32x4 xr[XVT], wr[WVT];
  auto issue = [&](int ch) {
    const int k0 = ch * 64;
#pragma unroll
    for (int n = 0; n < XVT; ++n) {
      const int v = t + NT * n, kr = v >> 5, pv = v & 31;
      const int64_t p = n0 + 8 * pv;
      xr[n] = (u32x4){0u, 0u, 0u, 0u};
      if (k0 + kr < a.K && p < a.N) xr[n] = *reinterpret_cast<const u32x4*>(xb + (int64_t)(k0 + kr) * a.N + p);
    }
    const u32x4* ws = reinterpret_cast<const u32x4*>(wb + (int64_t)ch * (WV * 8));
#pragma unroll
    for (int n = 0; n < WVT; ++n) wr[n] = ws[t + NT * n];
  };
  auto stash = [&]() {
#pragma unroll
    for (int n = 0; n < XVT; ++n) {
      const int v = t + NT * n, kr = v >> 5, pv = v & 31;
      *reinterpret_cast<u32x4*>(&X[kr * PL_XS + 8 * pv]) = xr[n];
    }
#pragma unroll
    for (int n = 0; n < WVT; ++n) reinterpret_cast<u32x4*>(Wl)[t + NT * n] = wr[n];
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[i][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0);
  for (int ch = 0; ch < a.nchunk; ++ch) {
    __syncthreads();
    stash();
    __syncthreads();
    if (ch + 1 < a.nchunk) issue(ch + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      s16x4 lo[4], hi[4];
      const bf16* const bp = &X[(32 * ks + 4 * g + qq) * PL_XS + 64 * wn + 4 * pp];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        lo[n] = tr_b16(bp + 16 * n);
        hi[n] = tr_b16(bp + 16 * PL_XS + 16 * n);
      }
      s16x8 af[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const s16x8*>(&Wl[((ks * (TM / 16) + 8 * wm + i) * 64 + lane) * 8]);
      lds_wait(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const s16x8 bf = cat8(lo[n], hi[n]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][n] = mfma32(af[i], bf, acc[i][n]);
      }
    }
  }

  // epilogue: per m-tile through a wave-private fp32 slab -> (+ bias, + residual) -> 16-byte stores along the rows
  __syncthreads();
  float* const slab = reinterpret_cast<float*>(pl_lds) + wv * (SLAB / 4);
  const int e_row = lane >> 3, e_col = (lane & 7) * 8;  // 8 lanes per row, 8 rows per pass
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float v[4] = {acc[i][n][0], acc[i][n][1], acc[i][n][2], acc[i][n][3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * 68 + 16 * n + li] = v[r];
    }
    wave_sync();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = 8 * ps + e_row;
      const int m = m_base + 128 * wm + 16 * i + row;
      const int64_t p = n0 + 64 * wn + e_col;
      if (m < a.M && p < a.N) {
        float o[8];
        Vec<float, 4>::ld(&slab[row * 68 + e_col], o);
        Vec<float, 4>::ld(&slab[row * 68 + e_col + 4], o + 4);
        const float bv = a.bias ? a.bias[m] : 0.f;
        const int64_t off = (int64_t)m * a.N + p;
        if (a.r) {
          const u32x4 rv = *reinterpret_cast<const u32x4*>(a.r + (int64_t)b * a.r_bs + off);
#pragma unroll
          for (int k = 0; k < 4; ++k) { o[2 * k] += bf_lo(rv[k]); o[2 * k + 1] += bf_hi(rv[k]); }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] += bv;
        Vec<bf16, 8>::st(a.y + (int64_t)b * a.y_bs + off, o);
      }
    }
    wave_sync();
  }
}

// Shapes this kernel takes from the planner of pw_gemm.hip: bf16, one X operand, static weights, no LayerNorm head / fp8 /
// split output, deep K, enough rows to fill the 128-row tiles, 16-byte aligned pixel rows.
bool pw_lds_ok(const mi_pw_desc* d) {
  if (MI_ENV(MI_NO_PW_LDS)) return false;
  if (d->dtype != MI_BF16 || d->k2 != 0 || d->groups != 1 || d->w_bs != 0 || d->ln_mode != 0 || d->f8 || d->y2) return false;
  if (d->k1 <= 128 || d->m < 128 || d->n % 8 != 0 || d->n < 256) return false;
  // Where it wins (profiles/r03_s_pw_lds_deep_gemm.txt, bs 32, incl. its pack launch): very wide outputs of a deep K (384 ->
  // 2042 / 1021: 1.42x) and deep K into few rows (1020 / 576 -> 192: 1.3-1.6x).  Elsewhere the wave-owned kernels hold
  // 430-530 TFLOP/s and stay (MI_PW_LDS=all takes every covered shape: tests, A/B).
  const char* e = MI_ENV(MI_PW_LDS);
  if (!(e && e[0] == 'a') && !((d->k1 >= 320 && d->m >= 1000) || (d->k1 >= 560 && d->m <= 256))) return false;
  if (!aligned16(d->x1) || !aligned16(d->y) || (d->r && !aligned16(d->r))) return false;
  if (d->x1_bs % 8 != 0 || d->y_bs % 8 != 0 || (d->r && d->r_bs % 8 != 0)) return false;
  return true;
}

static int pl_tm(int M) {                               // 256-row tiles unless 128-row tiles waste fewer padded rows
  return (int64_t)cdiv(M, 256) * 256 <= (int64_t)cdiv(M, 128) * 128 ? 256 : 128;
}
size_t pw_lds_pack_bytes(const mi_pw_desc* d) {
  const int TM = pl_tm(d->m);
  return (size_t)cdiv(d->m, TM) * cdiv(d->k1, 64) * TM * 64 * 2;
}

// ws: pw_lds_pack_bytes(d) bytes for the packed weight image (packed per call; TODO the trainer's pack cache)
int pw_lds_launch(const mi_pw_desc* d, void* ws, hipStream_t st) {
  const int TM = pl_tm(d->m);
  PlArgs a;
  a.x = (const bf16*)d->x1; a.wp = (const bf16*)ws; a.bias = d->bias; a.r = (const bf16*)d->r; a.y = (bf16*)d->y;
  a.x_bs = d->x1_bs; a.r_bs = d->r_bs; a.y_bs = d->y_bs; a.N = d->n;
  a.M = d->m; a.K = d->k1; a.nchunk = cdiv(d->k1, 64); a.tiles_n = cdiv(d->n, 256);
  MI_CHECK_ARG(a.tiles_n <= 65535 && d->batch <= 65535, "pw_gemm: grid too large");
  {
    PlPackArgs pa;
    pa.w = d->w; pa.wp = (bf16*)ws; pa.w_sm = d->w_sm; pa.w_sk = d->w_sk; pa.M = d->m; pa.K = d->k1; pa.TM = TM;
    pa.nchunk = a.nchunk; pa.ntile = cdiv(d->m, TM);
    const int64_t total = (int64_t)pa.ntile * pa.nchunk * TM * 64;
    ProfScope ps(st, K_PW_PACK, (double)total * 2 + 4.0 * d->m * d->k1, 0.0);
    hipLaunchKernelGGL(pl_pack_kernel, dim3((unsigned)(cdiv(total, 1024) < 1024 ? cdiv(total, 1024) : 1024)), dim3(256), 0, st, pa);
    MI_LAUNCH_CHECK();
  }
  const dim3 grid((unsigned)cdiv(d->m, TM), (unsigned)a.tiles_n, (unsigned)d->batch);
  const double N = (double)d->n * d->batch;
  ProfScope ps(st, K_PW_GEMM, (double)(d->k1 + d->m + (d->r ? d->m : 0)) * N * 2.0, 2.0 * d->m * d->k1 * N);
  if (TM == 256) {
    constexpr int LB = 64 * PL_XS * 2 + 256 * 64 * 2;
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)pl_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LB));
    hipLaunchKernelGGL((pl_kernel<2>), grid, dim3(512), LB, st, a);
  } else {
    constexpr int LB = 64 * PL_XS * 2 + 128 * 64 * 2;
    hipLaunchKernelGGL((pl_kernel<1>), grid, dim3(256), LB, st, a);
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi
