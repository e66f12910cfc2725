// Streaming form of the pointwise GEMM for the HBM-bound levels (bf16 activations, K <= 512, weights that fit LDS):
//   * the whole weight matrix of the slice is converted to bf16 ONCE per workgroup and stays in LDS;
//   * the workgroup then walks a run of consecutive 64-pixel tiles; the next tile's X rows are already in flight
//     (global -> registers) while the current tile is multiplied, so HBM loads never drain;
//   * each of the 4 waves owns one 16-pixel column group: its X^T fragments for ALL of K are read from LDS once per
//     tile (ds_read_b64_tr_b16) and kept in registers; it then sweeps the output channels, reading only W fragments.
// X is read from HBM exactly once, Y written once, W once per workgroup (L2-resident).  Same math and fragment
// conventions as pw_gemm.hip (k slots: element j<4 of lane group g is k=4g+j, j>=4 is k=16+4g+(j-4)).
#include <stdlib.h>

#include "internal.h"

namespace mi {

constexpr int PS_TN = 64;
constexpr int PS_XS = 80;       // X tile LDS row stride (elements)
constexpr int PS_XROWS = 256;   // X rows staged per pass
constexpr int PS_MF = 4;        // output-channel fragments per sweep step

struct PwS {
  PwK k;
  int tiles_per_block, n_tiles, ksteps, kpad, mpad, w_stride, x_rows, passes;
  int nt;  // 64-pixel sub-tiles loaded together (one "super-tile"): keeps nt * K * 128 bytes of loads in flight
};

__device__ __forceinline__ s16x4 ps_tr(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <int KS_MAX, bool WT>
__global__ __launch_bounds__(256) void pw_stream_kernel(PwS s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const PwK& p = s.k;
  bf16* Ws = reinterpret_cast<bf16*>(smem);
  const int w_elems = WT ? s.kpad * s.w_stride : s.mpad * s.w_stride;
  bf16* Xs = Ws + ((w_elems + 7) & ~7);

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4, q = li >> 2, pp = li & 3;
  const int z = blockIdx.y, zb = z / p.groups, zg = z - zb * p.groups;
  const int ktot = p.k1 + p.k2;
  const bf16* x1 = (const bf16*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  const bf16* x2 = p.x2 ? (const bf16*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : nullptr;
  const float* wz = p.w + zb * p.w_bs + zg * p.w_gs;
  bf16* yz = (bf16*)p.y + zb * p.y_bs + zg * p.y_gs;
  const bf16* rz = p.r ? (const bf16*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  const float* bz = p.bias ? p.bias + zg * p.bias_gs : nullptr;

  // ---- weights: fp32 global -> bf16 LDS, pairs along the contiguous axis ----
  if (WT) {
    const int half = s.mpad / 2, total = s.kpad * half;
    for (int idx = t; idx < total; idx += 256) {
      const int k = idx / half, m = (idx - k * half) * 2;
      float pr[2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
        pr[e] = (m + e < p.m && k < ktot) ? wz[(int64_t)(m + e) * p.w_sm + (int64_t)k * p.w_sk] : 0.f;
      Vec<bf16, 2>::st(&Ws[k * s.w_stride + m], pr);
    }
  } else {
    const int half = s.kpad / 2, total = s.mpad * half;
    for (int idx = t; idx < total; idx += 256) {
      const int m = idx / half, k = (idx - m * half) * 2;
      float pr[2];
#pragma unroll
      for (int e = 0; e < 2; ++e)
        pr[e] = (m < p.m && k + e < ktot) ? wz[(int64_t)m * p.w_sm + (int64_t)(k + e) * p.w_sk] : 0.f;
      Vec<bf16, 2>::st(&Ws[m * s.w_stride + k], pr);
    }
  }

  // ---- X staging: rows (pass*256 + r) of the tile, 16-byte vectors, thread -> (row = vid>>3, col = (vid&7)*8) ----
  u32x4 xreg[8];
  const int xv = s.x_rows * s.nt / 32;  // vectors per thread per pass (<= 8)
  const int rows8 = s.x_rows * 8;       // vectors per sub-tile
  auto load_x = [&](int64_t n0, int pass) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < xv) {
        const int vid = t + 256 * i;
        const int sub = vid / rows8, v2 = vid - sub * rows8;
        const int k = pass * PS_XROWS + (v2 >> 3);
        const int64_t n = n0 + sub * PS_TN + (v2 & 7) * 8;
        const bf16* row = nullptr;
        if (k < p.k1) row = x1 + (int64_t)k * p.n;
        else if (k < ktot) row = x2 + (int64_t)(k - p.k1) * p.n;
        if (row && p.vec_ok && n < p.n) {
          xreg[i] = *reinterpret_cast<const u32x4*>(row + n);
        } else {
          __attribute__((aligned(16))) bf16 tmp[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) tmp[j] = (row && n + j < p.n) ? row[n + j] : (bf16)0.f;
          xreg[i] = *reinterpret_cast<u32x4*>(tmp);
        }
      }
    }
  };
  auto write_x = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < xv) {
        const int vid = t + 256 * i;  // sub-tile-major image: [nt][x_rows][PS_XS]
        *reinterpret_cast<u32x4*>(&Xs[(vid >> 3) * PS_XS + (vid & 7) * 8]) = xreg[i];
      }
    }
  };

  s16x8 afr[KS_MAX];
  auto read_a = [&](int pass, const bf16* Xt) {  // this wave's X^T fragments (pixel group wv), k-steps of one pass
    s16x4 lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ks = pass * 8 + j;
      if (j < KS_MAX && ks < s.ksteps) {
        lo[j] = ps_tr(&Xt[(32 * j + 4 * g + q) * PS_XS + 16 * wv + 4 * pp]);
        hi[j] = ps_tr(&Xt[(32 * j + 16 + 4 * g + q) * PS_XS + 16 * wv + 4 * pp]);
      } else {
        lo[j] = (s16x4){0, 0, 0, 0};
        hi[j] = (s16x4){0, 0, 0, 0};
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                   "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7])
                 :
                 : "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ks = pass * 8 + j;
      if (ks < KS_MAX) afr[ks] = __builtin_shufflevector(lo[j], hi[j], 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  // tiles are counted in super-tiles of nt*64 pixels
  const int tile0 = blockIdx.x * s.tiles_per_block;
  int tile_end = tile0 + s.tiles_per_block;
  if (tile_end > s.n_tiles) tile_end = s.n_tiles;
  const int64_t super_px = (int64_t)s.nt * PS_TN;
  if (tile0 < tile_end) load_x((int64_t)tile0 * super_px, 0);
  const int mfrags = s.mpad / 16;

  for (int tile = tile0; tile < tile_end; ++tile) {
    const int64_t ns = (int64_t)tile * super_px;
    __syncthreads();  // W ready (first tile) / everyone done with the previous X image
    write_x();
    __syncthreads();
    if (s.passes == 1 && tile + 1 < tile_end) load_x(ns + super_px, 0);  // next super-tile flies during the multiply
   for (int sub = 0; sub < s.nt; ++sub) {
    const int64_t n0 = ns + (int64_t)sub * PS_TN;
    if (n0 >= p.n) break;  // uniform: ragged last super-tile
    read_a(0, Xs + sub * s.x_rows * PS_XS);
    if (KS_MAX > 8 && s.passes > 1) {  // K > 256: second half of the rows through the same LDS image (nt == 1)
      __syncthreads();
      load_x(n0, 1);
      write_x();
      __syncthreads();
      read_a(1, Xs);
      if (tile + 1 < tile_end) load_x(ns + super_px, 0);
    }

    const int64_t n = n0 + 16 * wv + 4 * g;  // this lane's 4 consecutive pixels
    for (int mb = 0; mb < mfrags; mb += PS_MF) {
      f32x4 acc[PS_MF];
#pragma unroll
      for (int mf = 0; mf < PS_MF; ++mf) acc[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS_MAX; ++ks) {
        if (ks < s.ksteps) {
          s16x4 blo[PS_MF], bhi[PS_MF];
#pragma unroll
          for (int mf = 0; mf < PS_MF; ++mf) {
            const int mfr = mb + mf < mfrags ? mb + mf : mfrags - 1;  // clamp: a valid address, result discarded
            if (WT) {
              blo[mf] = ps_tr(&Ws[(32 * ks + 4 * g + q) * s.w_stride + 16 * mfr + 4 * pp]);
              bhi[mf] = ps_tr(&Ws[(32 * ks + 16 + 4 * g + q) * s.w_stride + 16 * mfr + 4 * pp]);
            } else {
              const bf16* wr = &Ws[(16 * mfr + li) * s.w_stride + 32 * ks + 4 * g];
              blo[mf] = *reinterpret_cast<const s16x4*>(wr);
              bhi[mf] = *reinterpret_cast<const s16x4*>(wr + 16);
            }
          }
          if (WT) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(blo[0]), "+v"(blo[1]), "+v"(blo[2]), "+v"(blo[3]), "+v"(bhi[0]), "+v"(bhi[1]), "+v"(bhi[2]),
                           "+v"(bhi[3])
                         :
                         : "memory");
          }
#pragma unroll
          for (int mf = 0; mf < PS_MF; ++mf) {
            const s16x8 b = __builtin_shufflevector(blo[mf], bhi[mf], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], b, acc[mf], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int mf = 0; mf < PS_MF; ++mf) {
        const int m = 16 * (mb + mf) + li;
        if (m >= p.m || n >= p.n) continue;
        const float bv = bz ? bz[m] : 0.f;
        float o[4] = {acc[mf][0] + bv, acc[mf][1] + bv, acc[mf][2] + bv, acc[mf][3] + bv};
        const int64_t off = (int64_t)m * p.n + n;
        if (p.vec_ok) {
          if (rz) {
            float rr[4];
            Vec<bf16, 4>::ld(rz + off, rr);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += rr[j];
          }
          Vec<bf16, 4>::st(yz + off, o);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < p.n) st1(yz + off + j, o[j] + (rz ? ld1(rz + off + j) : 0.f));
        }
      }
    }
   }  // sub-tiles
  }
}

// Returns 1 and launches when the problem suits the streaming form, 0 when the caller should use the chunked kernel.
int pw_stream_try(const PwK& k, int batch, hipStream_t st, int* launched) {
  *launched = 0;
  const int ktot = k.k1 + k.k2;
  if (ktot > 512) return MI_OK;
  const bool wt = (k.w_sk != 1);
  PwS s;
  s.k = k;
  s.kpad = (ktot + 31) / 32 * 32;
  s.ksteps = s.kpad / 32;
  s.passes = s.ksteps > 8 ? 2 : 1;
  s.x_rows = s.kpad < PS_XROWS ? s.kpad : PS_XROWS;
  s.mpad = wt ? (k.m + 31) / 32 * 32 : (k.m + 15) / 16 * 16;
  s.w_stride = wt ? s.mpad + 16 : s.kpad + 8;
  const size_t w_elems = ((size_t)(wt ? s.kpad : s.mpad) * s.w_stride + 7) & ~(size_t)7;
  // (nt is chosen below; the X image never exceeds 256 rows in total)
  const size_t lds = (w_elems + (size_t)256 * PS_XS) * sizeof(bf16);
  if (lds > 150 * 1024) return MI_OK;
  s.nt = 1;
  if (s.passes == 1) {  // as many sub-tiles as the 8 staging vectors per thread allow (kpad * nt <= 256)
    s.nt = 256 / s.kpad;
    if (s.nt > 4) s.nt = 4;
    if (s.nt < 1) s.nt = 1;
    while (s.nt > 1 && (int64_t)(s.nt - 1) * PS_TN >= k.n) --s.nt;
  }
  s.n_tiles = cdiv(k.n, PS_TN * s.nt);
  const int Z = batch * k.groups;
  const int64_t total_tiles = (int64_t)s.n_tiles * Z;
  // weights are re-staged per workgroup: only worth it when each workgroup gets a few tiles
  const int target_blocks = lds <= 72 * 1024 ? 768 : 256;
  int tpb = cdiv(total_tiles, target_blocks);
  if (tpb < 1) tpb = 1;
  if (tpb > s.n_tiles) tpb = s.n_tiles;
  const bool force = getenv("MI_PW_FORCE_STREAM") != nullptr;  // tests: exercise this kernel on small shapes
  // Measured on MI355X (profiles/r01_c_kernel_microbench.log): the chunked kernel is faster on every Restormer shape,
  // so this form is opt-in (MI_PW_STREAM=1) until its per-tile phases overlap; kept because it is the base for the
  // LDS-DMA ring version and is covered by exact-integer tests.
  const bool on = getenv("MI_PW_STREAM") != nullptr;
  const int tpb_env = getenv("MI_PW_TPB") ? atoi(getenv("MI_PW_TPB")) : 0;
  if (!on && !force) return MI_OK;
  if (tpb_env > 0) tpb = tpb_env < s.n_tiles ? tpb_env : s.n_tiles;
  if (!force && tpb < 2 && (size_t)k.m * ktot > 4096) return MI_OK;  // tiny images with large weights: chunked kernel
  s.tiles_per_block = tpb;
  dim3 grid(cdiv(s.n_tiles, tpb), Z), block(256);
  if (grid.y > 65535) { set_error("pw_gemm: too many slices"); return MI_ERR_ARG; }
  ProfScope ps(st, K_PW_GEMM, ((double)ktot + k.m + (k.r ? k.m : 0)) * (double)k.n * Z * 2.0 + 4.0 * k.m * ktot,
               2.0 * k.m * ktot * (double)k.n * Z);
#define PS_LAUNCH(KS, WTV)                                                                                          \
  do {                                                                                                              \
    if (lds > 64 * 1024)                                                                                            \
      MI_CHECK_HIP(hipFuncSetAttribute((const void*)pw_stream_kernel<KS, WTV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds));                                                                  \
    hipLaunchKernelGGL((pw_stream_kernel<KS, WTV>), grid, block, lds, st, s);                                       \
  } while (0)
  if (wt) {
    if (s.ksteps <= 2) PS_LAUNCH(2, true); else if (s.ksteps <= 4) PS_LAUNCH(4, true);
    else if (s.ksteps <= 8) PS_LAUNCH(8, true); else PS_LAUNCH(16, true);
  } else {
    if (s.ksteps <= 2) PS_LAUNCH(2, false); else if (s.ksteps <= 4) PS_LAUNCH(4, false);
    else if (s.ksteps <= 8) PS_LAUNCH(8, false); else PS_LAUNCH(16, false);
  }
#undef PS_LAUNCH
  MI_LAUNCH_CHECK();
  *launched = 1;
  return MI_OK;
}

}  // namespace mi
