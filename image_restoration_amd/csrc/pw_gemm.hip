// Pointwise (1x1-conv) GEMM on N-contiguous NCHW planes, MFMA tiles for gfx950.
//   Y[z][m][n] = sum_k W_z(m,k) X[z][k][n] (+bias[m]) (+R[z][m][n])
// Orientation: the MFMA "A" operand is X^T (rows = pixels), "B" is W^T (cols = output channels), so the
// accumulator holds 4 consecutive pixels of one output channel per lane -> packed 8/16-byte stores
// along the contiguous pixel axis.  X is staged [k][n] exactly as it lies in HBM (coalesced 16-byte
// loads) and transposed for free on the LDS read (ds_read_b64_tr_b16); weights are staged either
// row-major [m][k] or, for transposed-weight calls (backward-data), [k][m] and read the same way.
// bf16 activations use v_mfma_f32_16x16x32_bf16, fp32 activations the exact v_mfma_f32_16x16x4_f32.
#include <type_traits>

#include "internal.h"

namespace mi {

constexpr int PW_TN = 64;  // pixels per workgroup tile
constexpr int PW_KC = 32;  // k per staged chunk
constexpr int PW_XS = 80;  // LDS row stride (elements) of the X chunk: conflict-free tr-reads / b32 reads


template <typename T, int TM, bool WT> struct PwLds {
  static constexpr int WS_ROW = std::is_same<T, float>::value ? 34 : 40;  // [m][k] row stride
  static constexpr int WS_T = TM + 16;                                   // [k][m] row stride
  static constexpr int W_ELEMS = WT ? PW_KC * WS_T : TM * WS_ROW;
  static constexpr int X_ELEMS = PW_KC * PW_XS;
};

__device__ __forceinline__ s16x4 lds_tr_b16(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(uintptr_t)p;  // LDS aperture: low 32 bits are the LDS byte address
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <typename T, int MF, bool WT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwK p) {
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr int TM = 64 * MF;
  using L = PwLds<T, TM, WT>;
  constexpr int STAGE_BYTES = (L::X_ELEMS + L::W_ELEMS) * (int)sizeof(T);
  constexpr int SLAB_BYTES = 4 * 16 * (PW_TN + 4) * (int)sizeof(float);  // epilogue: 4 waves x 16 rows x fp32
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE_BYTES > SLAB_BYTES ? STAGE_BYTES : SLAB_BYTES];
  T* const Xs = reinterpret_cast<T*>(lds_raw);
  T* const Ws = Xs + L::X_ELEMS;

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z, zb = z / p.groups, zg = z - zb * p.groups;
  const int64_t n0 = (int64_t)blockIdx.x * PW_TN;
  const int m0 = blockIdx.y * TM;
  const int ktot = p.k1 + p.k2;
  const int nchunks = (ktot + PW_KC - 1) / PW_KC;

  const T* x1 = (const T*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  const T* x2 = p.x2 ? (const T*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : nullptr;
  const float* wz = p.w + zb * p.w_bs + zg * p.w_gs;

  // ---- staging registers ----
  constexpr int XV = F32 ? 2 : 1;      // 16-byte vectors per thread for its 8 X elements
  constexpr int WP = TM / 16;          // weight pairs per thread per chunk
  u32x4 xreg[XV];
  float wreg[WP][2];
  const int xr_row = t >> 3, xr_col = (t & 7) * 8;

  auto load_stage = [&](int chunk) {
    const int k = chunk * PW_KC + xr_row;
    const T* row = nullptr;
    if (k < p.k1) row = x1 + (int64_t)k * p.n;
    else if (k < ktot) row = x2 + (int64_t)(k - p.k1) * p.n;
    const int64_t n = n0 + xr_col;
    constexpr int EPV = 8 / XV;  // elements per 16-byte vector
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int64_t nn = n + v * EPV;
      if (row && p.vec_ok && nn < p.n) {  // vec_ok: n % EPV == 0 for every row, so the vector is all-in or all-out
        xreg[v] = *reinterpret_cast<const u32x4*>(row + nn);
      } else {
        __attribute__((aligned(16))) T tmp[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) tmp[j] = (row && nn + j < p.n) ? row[nn + j] : Cvt<T>::from(0.f);
        xreg[v] = *reinterpret_cast<u32x4*>(tmp);
      }
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int idx = t + 256 * i;
      int m, k2;
      if (WT) { k2 = idx / (TM / 2); m = (idx - k2 * (TM / 2)) * 2; }
      else { m = idx >> 4; k2 = (idx & 15) * 2; }
      const int kk = chunk * PW_KC + k2;
      const int mm = m0 + m;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int me = WT ? mm + e : mm, ke = WT ? kk : kk + e;
        wreg[i][e] = (me < p.m && ke < ktot) ? wz[(int64_t)me * p.w_sm + (int64_t)ke * p.w_sk] : 0.f;
      }
    }
  };
  auto write_stage = [&]() {
#pragma unroll
    for (int v = 0; v < XV; ++v)
      *reinterpret_cast<u32x4*>(&Xs[xr_row * PW_XS + xr_col + v * (8 / XV)]) = xreg[v];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int idx = t + 256 * i;
      int m, k2;
      if (WT) { k2 = idx / (TM / 2); m = (idx - k2 * (TM / 2)) * 2; }
      else { m = idx >> 4; k2 = (idx & 15) * 2; }
      T* dst = WT ? &Ws[k2 * L::WS_T + m] : &Ws[m * L::WS_ROW + k2];
      float pr[2] = {wreg[i][0], wreg[i][1]};
      Vec<T, 2>::st(dst, pr);
    }
  };

  f32x4 acc[4][MF];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int mw = wv * MF * 16;  // this wave's first m inside the tile

  load_stage(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    write_stage();
    __syncthreads();
    if (c + 1 < nchunks) load_stage(c + 1);
    if constexpr (F32) {
#pragma unroll
      for (int ks = 0; ks < PW_KC / 4; ++ks) {
        const int kk = 4 * ks + g;
        float a[4], b[MF];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) a[nf] = Xs[kk * PW_XS + 16 * nf + li];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          b[mf] = WT ? Ws[kk * L::WS_T + mw + 16 * mf + li] : Ws[(mw + 16 * mf + li) * L::WS_ROW + kk];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
            acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nf], b[mf], acc[nf][mf], 0, 0, 0);
      }
    } else {
      // k slots: element j<4 of lane group g is k = 4g+j, element j>=4 is k = 16+4g+(j-4), for A and B alike.
      // The tr-reads are inline asm, so their results are tied through the s_waitcnt statement below:
      // every consumer is data-dependent on the wait and cannot be scheduled ahead of it.
      const int q = li >> 2, pp = li & 3;
      s16x4 alo[4], ahi[4], blo[MF], bhi[MF];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        alo[nf] = lds_tr_b16(&Xs[(4 * g + q) * PW_XS + 16 * nf + 4 * pp]);
        ahi[nf] = lds_tr_b16(&Xs[(16 + 4 * g + q) * PW_XS + 16 * nf + 4 * pp]);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if (WT) {
          blo[mf] = lds_tr_b16(&Ws[(4 * g + q) * L::WS_T + mw + 16 * mf + 4 * pp]);
          bhi[mf] = lds_tr_b16(&Ws[(16 + 4 * g + q) * L::WS_T + mw + 16 * mf + 4 * pp]);
        } else {
          const T* wr = &Ws[(mw + 16 * mf + li) * L::WS_ROW + 4 * g];
          blo[mf] = *reinterpret_cast<const s16x4*>(wr);
          bhi[mf] = *reinterpret_cast<const s16x4*>(wr + 16);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]),
                     "+v"(ahi[3])
                   :
                   : "memory");
      if (WT) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) asm volatile("" : "+v"(blo[mf]), "+v"(bhi[mf]));
      }
      s16x8 a[4], b[MF];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) a[nf] = __builtin_shufflevector(alo[nf], ahi[nf], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) b[mf] = __builtin_shufflevector(blo[mf], bhi[mf], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nf], b[mf], acc[nf][mf], 0, 0, 0);
    }
  }

  // ---- epilogue: lane holds pixels n0+16nf+4g+{0..3} of output channel m0+mw+16mf+li.  Stored from the
  // accumulator that is 8 bytes per lane scattered over 16 rows; the rows are therefore staged through LDS (one
  // 16-row slab per wave at a time) and written as whole 128-byte row segments, 16 bytes per lane.
  T* yz = (T*)p.y + zb * p.y_bs + zg * p.y_gs;
  const T* rz = p.r ? (const T*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  const float* bz = p.bias ? p.bias + zg * p.bias_gs : nullptr;
  constexpr int EPV = F32 ? 4 : 8;        // pixels per 16-byte vector of the output type
  constexpr int OS = PW_TN + 4;           // fp32 slab row stride (floats): 272 bytes
  constexpr int LPR = PW_TN / EPV;        // lanes per row on the way out
  constexpr int RPI = 64 / LPR;           // rows per store instruction
  float* slab = reinterpret_cast<float*>(lds_raw) + wv * 16 * OS;  // kept in fp32: the only rounding is the final store
  __syncthreads();                        // every wave is done reading the last chunk
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      float o[4] = {acc[nf][mf][0], acc[nf][mf][1], acc[nf][mf][2], acc[nf][mf][3]};
      Vec<float, 4>::st(&slab[li * OS + 16 * nf + 4 * g], o);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16 / RPI; ++it) {
      const int row = it * RPI + lane / LPR, col = (lane % LPR) * EPV;
      const int m = m0 + mw + 16 * mf + row;
      const int64_t n = n0 + col;
      if (m < p.m && n < p.n) {
        float o[EPV];
#pragma unroll
        for (int v = 0; v < EPV / 4; ++v) Vec<float, 4>::ld(&slab[row * OS + col + 4 * v], o + 4 * v);
        const float bv = bz ? bz[m] : 0.f;
#pragma unroll
        for (int j = 0; j < EPV; ++j) o[j] += bv;
        const int64_t off = (int64_t)m * p.n + n;
        if (p.vec_ok) {
          if (rz) {
            float rr[EPV];
            Vec<T, EPV>::ld(rz + off, rr);
#pragma unroll
            for (int j = 0; j < EPV; ++j) o[j] += rr[j];
          }
          Vec<T, EPV>::st(yz + off, o);
        } else {
#pragma unroll
          for (int j = 0; j < EPV; ++j)
            if (n + j < p.n) st1(yz + off + j, o[j] + (rz ? ld1(rz + off + j) : 0.f));
        }
      }
    }
    __syncthreads();
  }
}

template <typename T, bool WT>
static int pw_launch(const PwK& k, int batch, hipStream_t st) {
  // largest m-tile whose padding stays within 25% of the 64-granular minimum
  const int mmin = cdiv(k.m, 64) * 64;
  int tm = 64;
  if (cdiv(k.m, 256) * 256 * 4 <= mmin * 5) tm = 256;
  else if (cdiv(k.m, 128) * 128 * 4 <= mmin * 5) tm = 128;
  dim3 grid(cdiv(k.n, PW_TN), cdiv(k.m, tm), batch * k.groups), block(256);
  if (grid.y > 65535 || grid.z > 65535) { set_error("pw_gemm: grid too large"); return MI_ERR_ARG; }
  const double Z = (double)batch * k.groups, kt = k.k1 + k.k2;
  ProfScope ps(st, K_PW_GEMM, (kt + k.m + (k.r ? k.m : 0)) * (double)k.n * Z * sizeof(T) + 4.0 * k.m * kt,
               2.0 * k.m * kt * (double)k.n * Z);
  if (tm == 256) hipLaunchKernelGGL((pw_gemm_kernel<T, 4, WT>), grid, block, 0, st, k);
  else if (tm == 128) hipLaunchKernelGGL((pw_gemm_kernel<T, 2, WT>), grid, block, 0, st, k);
  else hipLaunchKernelGGL((pw_gemm_kernel<T, 1, WT>), grid, block, 0, st, k);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_pw_gemm(const mi_pw_desc* d, void* stream) {
  MI_CHECK_ARG(d && d->x1 && d->w && d->y, "pw_gemm: null pointer");
  MI_CHECK_ARG(d->m > 0 && d->n > 0 && d->k1 > 0 && d->k2 >= 0 && d->batch > 0 && d->groups > 0, "pw_gemm: bad shape");
  MI_CHECK_ARG((d->k2 == 0) == (d->x2 == nullptr), "pw_gemm: x2/k2 mismatch");
  MI_CHECK_ARG(d->dtype == MI_F32 || d->dtype == MI_BF16, "pw_gemm: bad dtype %d", d->dtype);
  MI_CHECK_ARG(d->w_sm == 1 || d->w_sk == 1, "pw_gemm: weight must be contiguous along m or k");
  const int64_t vec = d->dtype == MI_BF16 ? 8 : 4;
  PwK k;
  k.x1 = d->x1; k.x1_bs = d->x1_bs; k.x1_gs = d->x1_gs; k.k1 = d->k1;
  k.x2 = d->x2; k.x2_bs = d->x2_bs; k.x2_gs = d->x2_gs; k.k2 = d->k2;
  k.w = d->w; k.w_bs = d->w_bs; k.w_gs = d->w_gs; k.w_sm = d->w_sm; k.w_sk = d->w_sk;
  k.bias = d->bias; k.bias_gs = d->bias_gs;
  k.r = d->r; k.r_bs = d->r_bs; k.r_gs = d->r_gs;
  k.y = d->y; k.y_bs = d->y_bs; k.y_gs = d->y_gs;
  k.m = d->m; k.n = d->n; k.groups = d->groups;
  bool ok = (d->n % vec == 0) && aligned16(d->x1) && aligned16(d->x2) && aligned16(d->y) && aligned16(d->r);
  ok = ok && d->x1_bs % vec == 0 && d->x1_gs % vec == 0 && d->x2_bs % vec == 0 && d->x2_gs % vec == 0;
  ok = ok && d->y_bs % vec == 0 && d->y_gs % vec == 0 && d->r_bs % vec == 0 && d->r_gs % vec == 0;
  k.vec_ok = ok ? 1 : 0;
  const bool wt = (d->w_sk != 1);  // m-contiguous weights: stage [k][m]
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == MI_BF16) {  // HBM-bound shapes: weights resident in LDS, tiles streamed (pw_stream.hip)
    int launched = 0;
    MI_TRY(pw_stream_try(k, d->batch, st, &launched));
    if (launched) return MI_OK;
  }
  if (d->dtype == MI_F32) return wt ? pw_launch<float, true>(k, d->batch, st) : pw_launch<float, false>(k, d->batch, st);
  return wt ? pw_launch<bf16, true>(k, d->batch, st) : pw_launch<bf16, false>(k, d->batch, st);
}
