// Grouped expert GEMM (BASELINE configs[3]): the 1x1 projections of ALL MoCE experts in ONE launch.
//   moce_ir.py:545-558 ModExpert.process  - proj[0] C -> r_e, proj[1] C -> r_e, proj[2] r_e -> C (+ shortcut), per expert
//   moce_ir.py:666-672 AdapterLayer       - a Python list comprehension over the experts, each on its slice of the batch,
//   moce_ir.py:88      SparseDispatcher   - whose sizes the reference reads back to the host (.tolist()).
// Here the launch is described by a problem table (one entry per expert and projection: pointers, M = rank, K, strides) and the
// ragged part - how many rows of the dispatched batch each expert got, and where its segment starts - is read by the kernel from
// DEVICE memory: the counts / offsets tables the router launch wrote (csrc/moce.hip).  The grid is sized for the worst case
// (every row to one expert); a workgroup whose row lies beyond its expert's count exits at once.  Ranks differ per expert
// (rank_type "spread": C/8 .. C), so every problem carries its own M and K; tiles are 64 output channels x 64 pixels.
//   Y_p[row][M][N] = W_p[M x K] . X_p[row][K][N] (+ bias_p) (+ R_p[row][M][N])        row = local index i or offsets[e] + i
#include "common.h"
#include "fused_common.h"

namespace mi {
using namespace fz;

constexpr int GP_MAX = 16;   // problems per launch (E experts x up to 2 projections sharing an input)

struct GpK {
  const void* x; const float* w; const float* bias; const void* r; void* y;
  int64_t x_rs, y_rs, r_rs, w_sm, w_sk;
  int m, k, expert, x_local, y_local, r_local, tile0;   // tile0: first m-tile of this problem on grid.y
};
struct GpArgs {
  GpK p[GP_MAX];
  int np, tiles_m_total;
  const int* counts; const int* offsets;
  int64_t N;
};

// bf16: v_mfma_f32_16x16x32_bf16, X tile [32 k][64 px] staged in LDS and read transposed; fp32: the exact v_mfma_f32_16x16x4_f32.
template <typename T>
__global__ __launch_bounds__(256) void grouped_pw_kernel(GpArgs a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int KC = BF ? 32 : 16;                      // k rows staged per step
  __shared__ __attribute__((aligned(16))) unsigned char lds[KC * 72 * 4];
  T* const X = reinterpret_cast<T*>(lds);              // [KC][XS]
  constexpr int XS = BF ? 72 : 68;                     // row stride (elements): 16-byte aligned rows, conflict-free column reads
  // which problem
  int pi = 0;
#pragma unroll
  for (int i = 1; i < GP_MAX; ++i)
    if (i < a.np && (int)blockIdx.y >= a.p[i].tile0) pi = i;
  const GpK& p = a.p[pi];
  const int row = blockIdx.z;
  if (row >= a.counts[p.expert]) return;                // ragged: this expert got fewer rows
  const int grow = a.offsets[p.expert] + row;
  const int mt = blockIdx.y - p.tile0;
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const T* xr = reinterpret_cast<const T*>(p.x) + (int64_t)(p.x_local ? row : grow) * p.x_rs;
  T* yr = reinterpret_cast<T*>(p.y) + (int64_t)(p.y_local ? row : grow) * p.y_rs;
  const T* rr = p.r ? reinterpret_cast<const T*>(p.r) + (int64_t)(p.r_local ? row : grow) * p.r_rs : nullptr;
  const int m_row = mt * 64 + wv * 16 + li;            // the A-operand row (output channel) this lane feeds
  const bool m_ok = m_row < p.m;
  const int64_t N = a.N;
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.k; k0 += KC) {
    // stage X[k0 .. k0+KC)[n0 .. n0+64): coalesced rows, zero beyond K / N
    __syncthreads();
    if constexpr (BF) {
      const int kr = t >> 3, c8 = (t & 7) * 8;         // 32 rows x 8 vectors of 8 px
      u32x4 v = {0u, 0u, 0u, 0u};
      const int64_t n = n0 + c8;
      if (k0 + kr < p.k) {
        const T* src = xr + (int64_t)(k0 + kr) * N + n;
        if (n + 8 <= N && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) v = *reinterpret_cast<const u32x4*>(src);
        else {
          u16 tmp[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) tmp[j] = (n + j < N) ? reinterpret_cast<const u16*>(src)[j] : (u16)0;
          v[0] = tmp[0] | ((unsigned)tmp[1] << 16); v[1] = tmp[2] | ((unsigned)tmp[3] << 16);
          v[2] = tmp[4] | ((unsigned)tmp[5] << 16); v[3] = tmp[6] | ((unsigned)tmp[7] << 16);
        }
      }
      *reinterpret_cast<u32x4*>(&X[kr * XS + c8]) = v;
    } else {
      const int kr = t >> 4, c4 = (t & 15) * 4;        // 16 rows x 16 vectors of 4 px
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int64_t n = n0 + c4;
      if (k0 + kr < p.k) {
        const float* src = reinterpret_cast<const float*>(xr) + (int64_t)(k0 + kr) * N + n;
        if (n + 4 <= N && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) v = *reinterpret_cast<const f32x4*>(src);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (n + j < N) ? src[j] : 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(&reinterpret_cast<float*>(X)[kr * XS + c4]) = v;
    }
    __syncthreads();
    if constexpr (BF) {
      // A fragment: W[m_row][k0 + (4g + j | 16 + 4g + j - 4)], fp32 -> bf16 (same k order as the B fragment below)
      s16x8 af;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = k0 + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
        const float wv_ = (m_ok && kk < p.k) ? p.w[(int64_t)m_row * p.w_sm + (int64_t)kk * p.w_sk] : 0.f;
        af[j] = bf_bits(wv_);
      }
      const int qq = li >> 2, pp = li & 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const T* sp = &X[(4 * g + qq) * XS + q * 16 + 4 * pp];
        s16x4 lo = tr_b16(sp), hi = tr_b16(sp + 16 * XS);
        lds_wait(lo, hi);
        acc[q] = mfma32(af, cat8(lo, hi), acc[q]);
      }
    } else {
      const float* Xf = reinterpret_cast<const float*>(X);
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int kk = k0 + 4 * ks + g;
        const float av = (m_ok && kk < p.k) ? p.w[(int64_t)m_row * p.w_sm + (int64_t)kk * p.w_sk] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xf[(4 * ks + g) * XS + q * 16 + li], acc[q], 0, 0, 0);
      }
    }
  }
  // epilogue: D[row = 4g + r (channel)][col = li (pixel)]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t n = n0 + q * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mt * 64 + wv * 16 + 4 * g + r;
      if (m < p.m && n < N) {
        float v = acc[q][r];
        if (p.bias) v += p.bias[m];
        if (rr) v += to_f32(rr[(int64_t)m * N + n]);
        yr[(int64_t)m * N + n] = Cvt<T>::from(v);
      }
    }
  }
}

}  // namespace mi

using namespace mi;

extern "C" int mi_grouped_pw_gemm(const mi_grouped_problem* probs, int np, const int* dev_counts, const int* dev_offsets,
                                  int max_rows, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(probs && np >= 1 && np <= GP_MAX, "grouped_pw_gemm: 1..%d problems per launch", GP_MAX);
  MI_CHECK_ARG(dev_counts && dev_offsets && max_rows >= 1 && max_rows <= 65535 && N >= 1, "grouped_pw_gemm: bad row table / shape");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "grouped_pw_gemm: bad dtype %d", dtype);
  GpArgs a;
  memset(&a, 0, sizeof(a));
  int tiles = 0;
  double bytes = 0, flops = 0;
  for (int i = 0; i < np; ++i) {
    const mi_grouped_problem& q = probs[i];
    MI_CHECK_ARG(q.x && q.w && q.y && q.m >= 1 && q.k >= 1 && q.expert >= 0, "grouped_pw_gemm: problem %d: null pointer / bad shape", i);
    GpK& k = a.p[i];
    k.x = q.x; k.w = q.w; k.bias = q.bias; k.r = q.r; k.y = q.y;
    k.x_rs = q.x_rs ? q.x_rs : (int64_t)q.k * N;
    k.y_rs = q.y_rs ? q.y_rs : (int64_t)q.m * N;
    k.r_rs = q.r_rs ? q.r_rs : (int64_t)q.m * N;
    k.w_sm = q.w_sm; k.w_sk = q.w_sk;
    k.m = q.m; k.k = q.k; k.expert = q.expert;
    k.x_local = q.x_local; k.y_local = q.y_local; k.r_local = q.r_local;
    k.tile0 = tiles;
    tiles += cdiv(q.m, 64);
    bytes += (double)(q.m + q.k + (q.r ? q.m : 0)) * N * dtype_size(dtype);
    flops += 2.0 * q.m * q.k * N;
  }
  a.np = np; a.tiles_m_total = tiles;
  a.counts = dev_counts; a.offsets = dev_offsets; a.N = N;
  hipStream_t st = (hipStream_t)stream;
  // (algorithmic bytes / flops are booked for ONE row per problem: the row counts live on the device)
  ProfScope ps(st, K_PW_GEMM, bytes, flops);
  const dim3 grid((unsigned)cdiv(N, 64), (unsigned)tiles, (unsigned)max_rows);
  if (dtype == MI_BF16) hipLaunchKernelGGL((grouped_pw_kernel<bf16>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((grouped_pw_kernel<float>), grid, dim3(256), 0, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
