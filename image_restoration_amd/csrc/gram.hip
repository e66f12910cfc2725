// Row-Gram contraction over the pixel axis:  G[z][i][j] = sum_n A[z][i][n] * Bm[z][j][n].
// This one kernel is MDTA's q k^T (Restormer.py:124, with the F.normalize row sums of squares of
// :121-122 taken from the same operand fragments), the per-image dY V^T of its backward, and every
// 1x1-conv weight gradient (sum over batch).  Both operands are pixel-contiguous, so both MFMA
// fragments are plain 16-byte row reads from LDS.  The pixel axis is split across workgroups; partial
// tiles go to a workspace and are summed in a fixed order by a second kernel (bitwise reproducible).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace mi {

struct GramK {
  const void* a; int64_t a_bs, a_gs; int ma;
  const void* b; int64_t b_bs, b_gs; int mb;
  int64_t n; int groups; int Z;
  float* part;     // [splits][Z][ma][mb]
  float* ss_part;  // [splits][Z][ma+mb] or null
  int chunks_per_split, nchunks, tiles_b, vec_ok;
  int fold;        // > 0: the batch is folded into the pixel axis (weight gradients): chunk / step c lies in image c / fold
};

template <typename T, int FA, int FB, bool SS>   // tile = (32 FA) rows of A x (32 FB) rows of B; the 2 x 2 waves own 16 FA x 16 FB fragments each
__global__ __launch_bounds__(256) void gram_kernel(GramK p) {
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr int TA = 32 * FA, TB = 32 * FB;
  constexpr int KC = F32 ? 32 : 64;        // pixels per staged chunk (128 bytes per row)
  constexpr int AS = F32 ? 34 : 72;        // LDS row stride, elements
  constexpr int EPV = F32 ? 4 : 8;         // elements per 16-byte vector
  __shared__ __attribute__((aligned(16))) T As[TA * AS];
  __shared__ __attribute__((aligned(16))) T Bs[TB * AS];

  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar wave index)
  const int li = lane & 15, g = lane >> 4;
  const int wr = wv >> 1, wc = wv & 1;
  const int z = blockIdx.z, zb = p.fold ? 0 : z / p.groups, zg = p.fold ? z : z - zb * p.groups;
  const int ta = blockIdx.y / p.tiles_b, tb = blockIdx.y - ta * p.tiles_b;
  const int i0 = ta * TA, j0 = tb * TB;
  const T* A = (const T*)p.a + zb * p.a_bs + zg * p.a_gs;
  const T* B = (const T*)p.b + zb * p.b_bs + zg * p.b_gs;
  const int c_begin = blockIdx.x * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  if (c_end > p.nchunks) c_end = p.nchunks;

  u32x4 areg[FA], breg[FB];
  // (thread / lane ids below are opaque copies: the per-lane address terms are loop-invariant, and kept across the chunk loop
  //  they are what lifts this kernel over the 170 registers that allow a third workgroup per CU)
  auto load_rows = [&](auto cnt, const T* base, int64_t img_stride, int row0, int rows, int chunk, u32x4* regs) {
    constexpr int F = decltype(cnt)::value;
    int tt = t;
    asm volatile("" : "+v"(tt));
    if (p.fold) {                                                    // (uniform) image of this chunk, chunk inside the image
      const int img = chunk / p.fold;
      base += img * img_stride;
      chunk -= img * p.fold;
    }
#pragma unroll
    for (int i = 0; i < F; ++i) {
      const int v = tt + 256 * i;
      const int row = v >> 3, seg = v & 7;
      const int64_t n = (int64_t)chunk * KC + seg * EPV;
      const int r = row0 + row;
      if (r < rows && p.vec_ok && n < p.n) {
        regs[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)r * p.n + n);
      } else {
        __attribute__((aligned(16))) T tmp[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) tmp[e] = (r < rows && n + e < p.n) ? base[(int64_t)r * p.n + n + e] : Cvt<T>::from(0.f);
        regs[i] = *reinterpret_cast<u32x4*>(tmp);
      }
    }
  };
  auto write_rows = [&](auto cnt, T* dst, const u32x4* regs) {
    constexpr int F = decltype(cnt)::value;
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int i = 0; i < F; ++i) {
      const int v = tt + 256 * i;
      const int row = v >> 3, seg = v & 7;
      T* d = dst + row * AS + seg * EPV;
      if constexpr (F32) {  // 136-byte rows: 8-byte aligned only
        *reinterpret_cast<u32x2*>(d) = (u32x2){regs[i][0], regs[i][1]};
        *reinterpret_cast<u32x2*>(d + 2) = (u32x2){regs[i][2], regs[i][3]};
      } else {
        *reinterpret_cast<u32x4*>(d) = regs[i];
      }
    }
  };

  constexpr std::integral_constant<int, FA> NA{};
  constexpr std::integral_constant<int, FB> NB{};
  f32x4 acc[FA][FB];
  float ssa[FA], ssb[FB];
#pragma unroll
  for (int a = 0; a < FA; ++a) {
    ssa[a] = 0.f;
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int b = 0; b < FB; ++b) ssb[b] = 0.f;
  const int ra = wr * 16 * FA, cb = wc * 16 * FB;  // wave's first row / col inside the tile

  if (c_begin < c_end) {
    load_rows(NA, A, p.a_bs, i0, p.ma, c_begin, areg);
    load_rows(NB, B, p.b_bs, j0, p.mb, c_begin, breg);
  }
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();
    write_rows(NA, As, areg);
    write_rows(NB, Bs, breg);
    __syncthreads();
    if (c + 1 < c_end) {
      load_rows(NA, A, p.a_bs, i0, p.ma, c + 1, areg);
      load_rows(NB, B, p.b_bs, j0, p.mb, c + 1, breg);
    }
    if constexpr (F32) {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int kk = 4 * ks + g;
        float av[FA], bv[FB];
#pragma unroll
        for (int f = 0; f < FA; ++f) {
          av[f] = As[(ra + 16 * f + li) * AS + kk];
          if (SS) ssa[f] += av[f] * av[f];
        }
#pragma unroll
        for (int f = 0; f < FB; ++f) {
          bv[f] = Bs[(cb + 16 * f + li) * AS + kk];
          if (SS) ssb[f] += bv[f] * bv[f];
        }
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb)
            acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fa], bv[fb], acc[fa][fb], 0, 0, 0);
      }
    } else {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const T* ap = &As[(ra + (ln & 15)) * AS + 8 * (ln >> 4)];
      const T* bp = &Bs[(cb + (ln & 15)) * AS + 8 * (ln >> 4)];
#pragma unroll
      for (int ks = 0; ks < KC / 32; ++ks) {
        s16x8 av[FA], bv[FB];
#pragma unroll
        for (int f = 0; f < FA; ++f) {
          av[f] = *reinterpret_cast<const s16x8*>(ap + 16 * f * AS + 32 * ks);
          if (SS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float x = bf16_bits_to_f32((unsigned int)(unsigned short)av[f][e]);
              ssa[f] += x * x;
            }
          }
        }
#pragma unroll
        for (int f = 0; f < FB; ++f) {
          bv[f] = *reinterpret_cast<const s16x8*>(bp + 16 * f * AS + 32 * ks);
          if (SS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float y = bf16_bits_to_f32((unsigned int)(unsigned short)bv[f][e]);
              ssb[f] += y * y;
            }
          }
        }
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
          for (int fb = 0; fb < FB; ++fb)
            acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[fa], bv[fb], acc[fa][fb], 0, 0, 0);
      }
    }
  }

  // lane holds G[i0+ra+16fa+4g+r][j0+cb+16fb+li]
  float* pz = p.part + ((int64_t)blockIdx.x * p.Z + z) * ((int64_t)p.ma * p.mb);
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const int j = j0 + cb + 16 * fb + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + ra + 16 * fa + 4 * g + r;
        if (i < p.ma && j < p.mb) pz[(int64_t)i * p.mb + j] = acc[fa][fb][r];
      }
    }
  if (SS && p.ss_part) {
    float* sz = p.ss_part + ((int64_t)blockIdx.x * p.Z + z) * (p.ma + p.mb);
#pragma unroll
    for (int f = 0; f < FA; ++f) {
      float sa = ssa[f];
      sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
      const int i = i0 + ra + 16 * f + li;
      if (g == 0 && wc == 0 && tb == 0 && i < p.ma) sz[i] = sa;
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
      float sb = ssb[f];
      sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
      const int j = j0 + cb + 16 * f + li;
      if (g == 0 && wr == 0 && ta == 0 && j < p.mb) sz[p.ma + j] = sb;
    }
  }
}

// ---- bf16 streaming form ------------------------------------------------------------------------------------------------
// Both operands are pixel-contiguous, and the MFMA k index may be ANY permutation of the pixels as long as A and B use
// the same one - so a lane's fragment (row li, 8 consecutive pixels) is one plain 16-byte global load and nothing needs
// LDS.  A wave owns the whole (16 FA) x (16 FB) tile over its own pixels: per 64-pixel step it loads the two halves of
// one 128-byte line of each row (k-steps 0 and 1) and issues 2 FA FB MFMAs, with the next step's (FA + FB) x 2 loads
// already in flight; the four waves of a workgroup take interleaved steps of the workgroup's pixel range and are summed
// in a fixed order through LDS at the end.  No barrier in the streaming loop (the LDS-staged kernel above: two per chunk).
template <int FA, int FB, bool SS>
__global__ __launch_bounds__(256) void gram_stream_kernel(GramK p, int steps_per_block, int nsteps) {
  using T = bf16;
  __shared__ __attribute__((aligned(16))) f32x4 red[2][FA * FB][64];
  __shared__ float ssm[SS ? 4 : 1][SS ? (FA + FB) * 16 : 1];
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar wave index)
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z, zb = p.fold ? 0 : z / p.groups, zg = p.fold ? z : z - zb * p.groups;
  const int ta = blockIdx.y / p.tiles_b, tb = blockIdx.y - ta * p.tiles_b;
  const int i0 = ta * 16 * FA, j0 = tb * 16 * FB;
  const T* A0 = (const T*)p.a + zb * p.a_bs + zg * p.a_gs;
  const T* B0 = (const T*)p.b + zb * p.b_bs + zg * p.b_gs;
  const int s_begin = blockIdx.x * steps_per_block;
  const int s_end = min(s_begin + steps_per_block, nsteps);

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // Row addresses are rebuilt for every step from an opaque copy of the lane id: kept across the loop (FA + FB 64-bit pointers
  // and their validity flags) they cost ~30 registers of a kernel that needs every one for its two steps of loads in flight.
  auto load_step = [&](int s, u32x4 (*av)[2], u32x4 (*bv)[2]) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int li_ = ln & 15, g_ = ln >> 4;
    // 32-bit element offsets from the (uniform) operand bases: one slice is at most rows * n < 2^31 elements (gram_stream_ok)
    const T* A = A0;
    const T* B = B0;
    int sl = s;                                          // step inside its image
    if (p.fold) {
      const int img = s / p.fold;
      sl = s - img * p.fold;
      A += img * p.a_bs; B += img * p.b_bs;
    }
    const unsigned px = (unsigned)sl * 64u + 8u * g_;    // this lane's first pixel of k-step 0; k-step 1 is 32 further
    const unsigned un = (unsigned)p.n;
    const bool in0 = s < s_end && px + 8 <= un, in1 = s < s_end && px + 40 <= un;
#pragma unroll
    for (int f = 0; f < FA; ++f) {
      const int r = i0 + 16 * f + li_;
      const bool ok = r < p.ma;
      const u32x4* q = reinterpret_cast<const u32x4*>(A + ((unsigned)(ok ? r : 0) * un + px));
      av[f][0] = (ok && in0) ? q[0] : zero4;
      av[f][1] = (ok && in1) ? q[4] : zero4;
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
      const int r = j0 + 16 * f + li_;
      const bool ok = r < p.mb;
      const u32x4* q = reinterpret_cast<const u32x4*>(B + ((unsigned)(ok ? r : 0) * un + px));
      bv[f][0] = (ok && in0) ? q[0] : zero4;
      bv[f][1] = (ok && in1) ? q[4] : zero4;
    }
  };

  f32x4 acc[FA][FB];
  float ssa[FA], ssb[FB];
#pragma unroll
  for (int a = 0; a < FA; ++a) {
    ssa[a] = 0.f;
#pragma unroll
    for (int b = 0; b < FB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int b = 0; b < FB; ++b) ssb[b] = 0.f;

  u32x4 ca[FA][2], cb[FB][2];
  load_step(s_begin + wv, ca, cb);
  for (int s = s_begin + wv; s < s_end; s += 4) {
    u32x4 na[FA][2], nb[FB][2];
    load_step(s + 4, na, nb);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      s16x8 av[FA], bv[FB];
#pragma unroll
      for (int f = 0; f < FA; ++f) av[f] = __builtin_bit_cast(s16x8, ca[f][h]);
#pragma unroll
      for (int f = 0; f < FB; ++f) bv[f] = __builtin_bit_cast(s16x8, cb[f][h]);
      if (SS) {
#pragma unroll
        for (int f = 0; f < FA; ++f)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = bf16_bits_to_f32((unsigned int)(unsigned short)av[f][e]);
            ssa[f] += x * x;
          }
#pragma unroll
        for (int f = 0; f < FB; ++f)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float y = bf16_bits_to_f32((unsigned int)(unsigned short)bv[f][e]);
            ssb[f] += y * y;
          }
      }
#pragma unroll
      for (int fa = 0; fa < FA; ++fa)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
          acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[fa], bv[fb], acc[fa][fb], 0, 0, 0);
    }
#pragma unroll
    for (int f = 0; f < FA; ++f) { ca[f][0] = na[f][0]; ca[f][1] = na[f][1]; }
#pragma unroll
    for (int f = 0; f < FB; ++f) { cb[f][0] = nb[f][0]; cb[f][1] = nb[f][1]; }
  }

  // fixed-order sum of the four waves: (w0 + w2) + (w1 + w3)
  if (wv >= 2) {
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) red[wv - 2][fa * FB + fb][lane] = acc[fa][fb];
  }
  if (SS) {
#pragma unroll
    for (int f = 0; f < FA; ++f) {
      float v = ssa[f];
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (g == 0) ssm[wv][16 * f + li] = v;
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
      float v = ssb[f];
      v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (g == 0) ssm[wv][16 * (FA + f) + li] = v;
    }
  }
  __syncthreads();
  if (wv < 2) {
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const f32x4 o = red[wv][fa * FB + fb][lane];
        acc[fa][fb] += o;
      }
  }
  __syncthreads();
  if (wv == 1) {
#pragma unroll
    for (int fa = 0; fa < FA; ++fa)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) red[0][fa * FB + fb][lane] = acc[fa][fb];
  }
  __syncthreads();
  if (wv != 0) return;
  float* pz = p.part + ((int64_t)blockIdx.x * p.Z + z) * ((int64_t)p.ma * p.mb);
#pragma unroll
  for (int fa = 0; fa < FA; ++fa)
#pragma unroll
    for (int fb = 0; fb < FB; ++fb) {
      const f32x4 o = acc[fa][fb] + red[0][fa * FB + fb][lane];
      const int j = j0 + 16 * fb + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 16 * fa + 4 * g + r;
        if (i < p.ma && j < p.mb) pz[(int64_t)i * p.mb + j] = o[r];
      }
    }
  if (SS && p.ss_part) {
    float* sz = p.ss_part + ((int64_t)blockIdx.x * p.Z + z) * (p.ma + p.mb);
    for (int e = lane; e < (FA + FB) * 16; e += 64) {
      const float v = (ssm[0][e] + ssm[2][e]) + (ssm[1][e] + ssm[3][e]);
      if (e < FA * 16) { const int i = i0 + e; if (tb == 0 && i < p.ma) sz[i] = v; }
      else { const int j = j0 + e - FA * 16; if (ta == 0 && j < p.mb) sz[p.ma + j] = v; }
    }
  }
}

// out[zo][i*ld + j] (+)= sum_{split} sum_{b if sum_batch} part[split][b*groups+g][i][j]
// 256 threads = 16 consecutive elements x 16 slice-phases; each thread walks its slices 4 at a time (loads in flight),
// phases are combined through LDS in a fixed order (reproducible).
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                          int splits, int batch, int groups, int ma, int mb, int64_t out_ld,
                                                          int64_t out_zs, int sum_batch, int accumulate) {
  __shared__ float sm[16][17];
  const int ex = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int64_t e = (int64_t)blockIdx.x * 16 + ex;
  const int64_t per = (int64_t)ma * mb;
  const int zo = blockIdx.y;
  const int Z = batch * groups;
  const int nb = sum_batch ? batch : 1;
  const int total = splits * nb;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < per) {
    auto addr = [&](int q) -> int64_t {
      const int sp = q / nb, b = q - sp * nb;
      const int64_t zz = sum_batch ? ((int64_t)b * groups + zo) : zo;
      return ((int64_t)sp * Z + zz) * per + e;
    };
    int q = ph;
    for (; q + 48 < total; q += 64) {
      s0 += part[addr(q)]; s1 += part[addr(q + 16)]; s2 += part[addr(q + 32)]; s3 += part[addr(q + 48)];
    }
    for (; q < total; q += 16) s0 += part[addr(q)];
  }
  sm[ph][ex] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && e < per) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k][ex];
    const int i = (int)(e / mb), j = (int)(e - (int64_t)i * mb);
    float* o = out + (int64_t)zo * out_zs + (int64_t)i * out_ld + j;
    *o = (accumulate ? *o : 0.f) + s;
  }
}

// Few slices, or enough output elements to fill the chip: one thread per output element, the slices walked in order NS at a time
// (the 16-phase kernel above keeps 1/16 of its lanes busy there: 33 us for 43 MB at C = 384, profiles/r04_f_attn_small.txt).
template <int NS>
__global__ __launch_bounds__(256) void gram_reduce_few_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int total, int batch, int groups, int ma, int mb, int64_t out_ld,
                                                              int64_t out_zs, int sum_batch, int accumulate) {
  const int64_t per = (int64_t)ma * mb;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int zo = blockIdx.y;
  if (e >= per) return;
  const int Z = batch * groups;
  const int nb = sum_batch ? batch : 1;
  float s = 0.f;
  for (int q0 = 0; q0 < total; q0 += NS) {       // NS loads in flight, summed in slice order
    float v[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int q = q0 + u;
      const int sp = q / nb, b = q - sp * nb;
      const int64_t zz = sum_batch ? ((int64_t)b * groups + zo) : zo;
      v[u] = q < total ? part[((int64_t)sp * Z + zz) * per + e] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) s += v[u];
  }
  const int i = (int)(e / mb), j = (int)(e - (int64_t)i * mb);
  float* o = out + (int64_t)zo * out_zs + (int64_t)i * out_ld + j;
  *o = (accumulate ? *o : 0.f) + s;
}
static void launch_gram_reduce(const float* part, float* out, int splits, int batch, int groups, int ma, int mb, int64_t out_ld,
                               int64_t out_zs, int sum_batch, int accumulate, int zo, hipStream_t st) {
  const int64_t per = (int64_t)ma * mb;
  const int total = splits * (sum_batch ? batch : 1);
  if (total <= 4)
    hipLaunchKernelGGL(gram_reduce_few_kernel<4>, dim3(cdiv(per, 256), zo), dim3(256), 0, st, part, out, total, batch, groups, ma, mb,
                       out_ld, out_zs, sum_batch, accumulate);
  else if (total <= 16 || per * zo >= 16384)      // enough elements to fill the chip with one thread each: slices walked in the thread
    hipLaunchKernelGGL(gram_reduce_few_kernel<16>, dim3(cdiv(per, 256), zo), dim3(256), 0, st, part, out, total, batch, groups, ma, mb,
                       out_ld, out_zs, sum_batch, accumulate);
  else
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(cdiv(per, 16), zo), dim3(256), 0, st, part, out, splits, batch, groups, ma, mb, out_ld,
                       out_zs, sum_batch, accumulate);
}

// A weight-gradient Gram that accumulates into a dense [ma][mb] gradient may leave its split partials in the deferred arena
// and have them summed by mi_deferred_flush (common.h): rows = splits x images of ma*mb floats, a plain fixed-order row sum.
static bool gram_deferrable(const mi_gram_desc* d) {
  return d->accumulate && d->sum_batch && d->groups == 1 && d->out_ld == d->mb && !d->sumsq;
}

struct GramPlan { int FA, FB, kc, nchunks, tiles_a, tiles_b, splits, cps, Z, fold; size_t part_bytes, ss_bytes; };

// Weight gradients sum over the batch: instead of one partial tile per image (and a reduction over batch x splits partials),
// the images are chained along the contraction axis - a workgroup's pixel range may cross image boundaries - whenever an image
// is a whole number of chunks / steps.  Returns the chunks per image, or 0.  Measured at bs 32 (profiles/r02_k_gram_fold_ab.txt):
// it pays where an image is short (32 x 32 planes: 16 chunks per workgroup against a 64 KB partial tile each - 1152 x 384
// 102 -> 76 us, 2042 x 384 144 -> 127 us) and where images x tiles leaves the chip half empty (576 x 192 at 64 x 64: 320
// workgroups, 163 -> 125 us); on long planes the per-chunk image lookup costs 5-10 % and buys nothing, so those stay per image.
static int gram_fold(const mi_gram_desc* d, int unit, int64_t tiles_per_image) {
  const char* e = MI_ENV(MI_GRAM_FOLD);                            // A/B switch: 0 never, 2 wherever possible
  if (e && e[0] == '0') return 0;
  if (!d->sum_batch || d->batch <= 1 || d->sumsq || d->n % unit != 0 || d->n / unit > (1 << 20)) return 0;
  // (second case: the per-image plan would run ONE split - 512 / workgroups rounds to 1 - on a chip it does not fill)
  const int64_t wgs = tiles_per_image * d->batch;
  const bool pays = d->n / unit <= 16 || (wgs > 256 && wgs < 448);
  if (!pays && !(e && e[0] == '2')) return 0;
  return (int)(d->n / unit);
}
static int gram_want(int dflt) {
  const char* e = MI_ENV(MI_GRAM_WANT);                            // A/B switch: workgroups the pixel split aims for
  return e ? atoi(e) : dflt;
}

static GramPlan gram_plan(const mi_gram_desc* d) {
  GramPlan g;
  g.FA = g.FB = (d->ma > 64 || d->mb > 64) ? 4 : 2;
  // rectangular tiles (bf16): a tile that holds ALL of the shorter operand's rows reads the other operand once instead of once
  // per 128-row tile of it - 96 x 255 (project_out's weight gradient) as ONE 96 x 256 tile, M x 192 (the C = 192 level) and the
  // wide M x 384 as 128 x 192 tiles.  bs 32 (profiles/r02_o_gram_rect_tiles.txt): 96 x 255 at 256^2 343 -> 304 us, 1020 x 192
  // 158 -> 116, 576 x 192 124 -> 85, 2042 x 384 128 -> 105 (384 x 384 loses: 35 -> 39, stays square); step 153.6 -> 152.1 ms.
  // MI_GRAM_RECT=0: square tiles only (A/B switch).
  {
    const char* e = MI_ENV(MI_GRAM_RECT);
    if (d->dtype == MI_BF16 && !(e && e[0] == '0') && d->ma > 64 && d->ma <= 96 && d->mb > 64 && d->mb <= 96) {
      g.FA = g.FB = 3;                                               // 96 x 96 (q k^T and dM at c = 96): no padded fragments
    } else if (d->dtype == MI_BF16 && !d->sumsq && !(e && e[0] == '0')) {
      if (d->ma <= 96 && d->mb > 128 && d->mb <= 256) { g.FA = 3; g.FB = 8; }
      else if (d->ma > 128 && d->mb > 128 && (d->mb <= 192 || (d->mb == 384 && d->ma > 384))) { g.FA = 4; g.FB = 6; }
      else if ((d->ma > 128 && d->ma <= 192 && d->mb > 192) || (d->ma == 384 && d->mb > 384)) { g.FA = 6; g.FB = 4; }   // 192 x 510 (86 -> 65 us), 384 x 1021 (71 -> 62)
    }
  }
  g.kc = d->dtype == MI_BF16 ? 64 : 32;
  g.nchunks = cdiv(d->n, g.kc);
  g.tiles_a = cdiv(d->ma, 32 * g.FA);
  g.tiles_b = cdiv(d->mb, 32 * g.FB);
  g.fold = gram_fold(d, g.kc, (int64_t)g.tiles_a * g.tiles_b * d->groups);
  if (g.fold) g.nchunks = d->batch * g.fold;
  g.Z = (g.fold ? 1 : d->batch) * d->groups;
  const int64_t tiles = (int64_t)g.tiles_a * g.tiles_b * g.Z;
  int64_t want = gram_want(512) / (tiles > 0 ? tiles : 1);
  if (want < 1) want = 1;
  if (want > g.nchunks) want = g.nchunks;
  g.cps = cdiv(g.nchunks, want);
  g.splits = cdiv(g.nchunks, g.cps);
  g.part_bytes = align_up((size_t)g.splits * g.Z * d->ma * d->mb * sizeof(float), 256);
  g.ss_bytes = d->sumsq ? align_up((size_t)g.splits * g.Z * (d->ma + d->mb) * sizeof(float), 256) : 0;
  return g;
}

// streaming plan (bf16, 16-byte aligned rows): fragment counts per tile and the pixel split
struct GramSPlan { int fa, fb, tiles_a, tiles_b, nsteps, spb, splits, Z, fold; size_t part_bytes, ss_bytes; };
static int gram_pick_frags(int m) {
  int best = 3, best_pad = 1 << 30;
  for (int f : {3, 4, 6}) {
    const int pad = cdiv(m, 16 * f) * 16 * f;
    if (pad < best_pad || (pad == best_pad && f > best)) { best = f; best_pad = pad; }
  }
  return best;
}
static GramSPlan gram_splan(const mi_gram_desc* d) {
  GramSPlan g;
  g.fa = gram_pick_frags(d->ma);
  g.fb = gram_pick_frags(d->mb);
  if (g.fa * g.fb > 24) { if (g.fa >= g.fb) g.fa = 3; else g.fb = 3; }  // 6x6 -> 3x6: accumulators + two stages of loads <= 256 VGPRs
  g.tiles_a = cdiv(d->ma, 16 * g.fa);
  g.tiles_b = cdiv(d->mb, 16 * g.fb);
  g.fold = gram_fold(d, 64, 0);                                     // (streaming form: short planes only)
  g.Z = (g.fold ? 1 : d->batch) * d->groups;
  g.nsteps = g.fold ? d->batch * g.fold : (int)cdiv(d->n, 64);
  const int64_t tiles = (int64_t)g.tiles_a * g.tiles_b * g.Z;
  int64_t want = gram_want(768) / (tiles > 0 ? tiles : 1);
  const int64_t cap = g.nsteps / 32 > 0 ? g.nsteps / 32 : 1;   // >= 8 steps per wave: the prefetch pipeline needs a run
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  // several tiles re-read the same pixels: a split count that is a multiple of 8 puts them on one XCD (shared L2)
  if (g.tiles_a * g.tiles_b > 1 && want >= 8) want = want / 8 * 8;
  g.spb = (int)((cdiv(g.nsteps, want) + 3) / 4 * 4);
  g.splits = (int)cdiv(g.nsteps, g.spb);
  g.part_bytes = align_up((size_t)g.splits * g.Z * d->ma * d->mb * sizeof(float), 256);
  g.ss_bytes = align_up((size_t)g.splits * g.Z * (d->ma + d->mb) * sizeof(float), 256);
  return g;
}
// One split, one output slice per z, plain overwrite into a dense [Z][ma][mb] output: the kernel's "partial" IS the result,
// so it writes straight to the output and the reduce launch disappears (dM = dy v^T at C = 384: a 19 MB copy at 330 GB/s).
static bool gram_direct(const mi_gram_desc* d, int splits) {
  return splits == 1 && !d->sum_batch && !d->accumulate && d->out_ld == d->mb && d->out_zs == (int64_t)d->ma * d->mb;
}
static bool gram_stream_ok(const mi_gram_desc* d) {
  if (d->dtype != MI_BF16 || MI_ENV(MI_GRAM_LDS)) return false;
  bool ok = (d->n % 8 == 0) && aligned16(d->a) && aligned16(d->b);
  ok = ok && d->a_bs % 8 == 0 && d->a_gs % 8 == 0 && d->b_bs % 8 == 0 && d->b_gs % 8 == 0;
  if (!ok) return false;
  // Measured (profiles/r01_s_gram_microbench.log): the streaming form wins while one or a few SMALL tiles cover the
  // output (48x48 q k^T, the 144/254/127 x 48 weight gradients: 1.1-1.7x); with 96-wide or many tiles the operand
  // re-reads and 1-wave occupancy lose to the LDS-staged 128x128 tiles (0.6-0.85x).
  const GramSPlan g = gram_splan(d);
  if (MI_ENV(MI_GRAM_STREAM_ALL)) return true;   // A/B switch
  if ((g.fold ? (int64_t)d->batch * d->n : d->n) < 4096) return false;
  if (g.fa <= 4 && g.fb <= 4 && g.tiles_a * g.tiles_b <= 4) return true;
  // ... and where the 128 x 128 LDS tiles would be mostly padding (288 x 96 fills 56% of 3 x 1 tiles: streaming 1.27x
  // faster at bs 32; 96 x 255 and 510 x 96 fill 75% and stay on the LDS kernel; profiles/r01_y_gram_bs32.log)
  const GramPlan lp = gram_plan(d);                                  // (the tile the LDS kernel would actually use)
  const double lds_fill = (double)d->ma * d->mb / ((double)lp.tiles_a * 32 * lp.FA * lp.tiles_b * 32 * lp.FB);
  return (d->ma > 64 || d->mb > 64) && lds_fill < 0.6;
}

static int gram_check(const mi_gram_desc* d) {
  MI_CHECK_ARG(d && d->a && d->b && d->out, "gram: null pointer");
  MI_CHECK_ARG(d->ma > 0 && d->mb > 0 && d->n > 0 && d->batch > 0 && d->groups > 0, "gram: bad shape");
  MI_CHECK_ARG(d->dtype == MI_F32 || d->dtype == MI_BF16, "gram: bad dtype %d", d->dtype);
  MI_CHECK_ARG(d->out_ld >= d->mb, "gram: out_ld < mb");
  MI_CHECK_ARG(!(d->sumsq && d->sum_batch), "gram: sumsq with sum_batch is not supported");
  return MI_OK;
}

}  // namespace mi

using namespace mi;

static int gram_stream_launch(const mi_gram_desc* d, void* ws, hipStream_t st) {
  const GramSPlan g = gram_splan(d);
  GramK k;
  k.a = d->a; k.a_bs = d->a_bs; k.a_gs = d->a_gs; k.ma = d->ma;
  k.b = d->b; k.b_bs = d->b_bs; k.b_gs = d->b_gs; k.mb = d->mb;
  k.n = d->n; k.groups = d->groups; k.Z = g.Z; k.fold = g.fold;
  const bool direct = gram_direct(d, g.splits);
  k.part = direct ? d->out : (float*)ws;
  bool deferred = false;
  if (!direct && gram_deferrable(d)) {
    float* arena = deferred_take(g.part_bytes / sizeof(float), st);
    if (arena) { k.part = arena; deferred = true; }
  }
  k.ss_part = d->sumsq ? (float*)((char*)ws + g.part_bytes) : nullptr;
  k.chunks_per_split = 0; k.nchunks = 0; k.tiles_b = g.tiles_b; k.vec_ok = 1;
  dim3 grid(g.splits, g.tiles_a * g.tiles_b, g.Z), block(256);
  MI_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gram: grid too large");
  const bool ss = d->sumsq != nullptr;
  {
    const double ZZ = (double)d->batch * d->groups;
    ProfScope ps(st, d->sumsq ? K_GRAM_QK : K_GRAM, (double)(d->ma + d->mb) * d->n * ZZ * 2.0 + 4.0 * g.splits * g.Z * d->ma * d->mb,
                 2.0 * d->ma * d->mb * (double)d->n * ZZ);
#define GS_CASE(FA_, FB_)                                                                                           \
  if (g.fa == FA_ && g.fb == FB_) {                                                                                 \
    if (ss) hipLaunchKernelGGL((gram_stream_kernel<FA_, FB_, true>), grid, block, 0, st, k, g.spb, g.nsteps);       \
    else hipLaunchKernelGGL((gram_stream_kernel<FA_, FB_, false>), grid, block, 0, st, k, g.spb, g.nsteps);         \
  }
    GS_CASE(3, 3) else GS_CASE(3, 4) else GS_CASE(3, 6) else GS_CASE(4, 3) else GS_CASE(4, 4) else GS_CASE(4, 6)
    else GS_CASE(6, 3) else GS_CASE(6, 4) else { MI_CHECK_ARG(false, "gram: no streaming tile %dx%d", g.fa, g.fb); }
#undef GS_CASE
  }
  MI_LAUNCH_CHECK();
  if (deferred) {
    const int64_t per = (int64_t)d->ma * d->mb;
    MI_TRY(launch_reduce_rows(k.part, d->out, (int64_t)g.splits * (g.fold ? 1 : d->batch), per, per, d->accumulate, 1.0f, st));
  } else if (!direct) {
    ProfScope ps2(st, K_GRAM_REDUCE, 4.0 * (g.splits + 1) * g.Z * d->ma * d->mb, (double)g.splits * g.Z * d->ma * d->mb);
    const int zo = d->sum_batch ? d->groups : g.Z;
    const int64_t per = (int64_t)d->ma * d->mb;
    launch_gram_reduce(k.part, d->out, g.splits, g.fold ? 1 : d->batch, d->groups, d->ma, d->mb, d->out_ld, d->out_zs, d->sum_batch,
                       d->accumulate, zo, st);
    MI_LAUNCH_CHECK();
  }
  if (ss) {
    const int64_t cols = (int64_t)g.Z * (d->ma + d->mb);
    MI_TRY(launch_reduce_rows(k.ss_part, d->sumsq, g.splits, cols, cols, 0, 1.0f, st));
  }
  return MI_OK;
}

extern "C" size_t mi_gram_workspace(const mi_gram_desc* d) {
  if (!d || d->ma <= 0 || d->mb <= 0 || d->n <= 0 || d->batch <= 0 || d->groups <= 0) return 0;
  GramPlan g = gram_plan(d);
  // sumsq presence may differ between the sizing call and the real call: always reserve it
  const size_t ss = align_up((size_t)g.splits * g.Z * (d->ma + d->mb) * sizeof(float), 256);
  const GramSPlan sp = gram_splan(d);  // either kernel may run (alignment decides at call time): cover both
  return std::max(g.part_bytes + ss, sp.part_bytes + sp.ss_bytes);
}

extern "C" int mi_gram(const mi_gram_desc* d, void* ws, void* stream) {
  MI_TRY(gram_check(d));
  MI_CHECK_ARG(ws, "gram: null workspace");
  if (gram_stream_ok(d)) return gram_stream_launch(d, ws, (hipStream_t)stream);
  GramPlan g = gram_plan(d);
  hipStream_t st = (hipStream_t)stream;
  GramK k;
  k.a = d->a; k.a_bs = d->a_bs; k.a_gs = d->a_gs; k.ma = d->ma;
  k.b = d->b; k.b_bs = d->b_bs; k.b_gs = d->b_gs; k.mb = d->mb;
  k.n = d->n; k.groups = d->groups; k.Z = g.Z; k.fold = g.fold;
  const bool direct = gram_direct(d, g.splits);
  k.part = direct ? d->out : (float*)ws;
  bool deferred = false;
  if (!direct && gram_deferrable(d)) {
    float* arena = deferred_take(g.part_bytes / sizeof(float), st);
    if (arena) { k.part = arena; deferred = true; }
  }
  k.ss_part = d->sumsq ? (float*)((char*)ws + g.part_bytes) : nullptr;
  k.chunks_per_split = g.cps; k.nchunks = g.nchunks; k.tiles_b = g.tiles_b;
  const int64_t vec = d->dtype == MI_BF16 ? 8 : 4;
  bool ok = (d->n % vec == 0) && aligned16(d->a) && aligned16(d->b);
  ok = ok && d->a_bs % vec == 0 && d->a_gs % vec == 0 && d->b_bs % vec == 0 && d->b_gs % vec == 0;
  k.vec_ok = ok ? 1 : 0;
  dim3 grid(g.splits, g.tiles_a * g.tiles_b, g.Z), block(256);
  MI_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "gram: grid too large");
  const bool ss = d->sumsq != nullptr;
  const double es = d->dtype == MI_BF16 ? 2.0 : 4.0;
  {
  const double ZZ = (double)d->batch * d->groups;
  ProfScope ps(st, d->sumsq ? K_GRAM_QK : K_GRAM, (double)(d->ma + d->mb) * d->n * ZZ * es + 4.0 * g.splits * g.Z * d->ma * d->mb,
               2.0 * d->ma * d->mb * (double)d->n * ZZ);
#define GRAM_CASE(T, FA_, FB_)                                                                      \
  do {                                                                                              \
    if (ss) hipLaunchKernelGGL((gram_kernel<T, FA_, FB_, true>), grid, block, 0, st, k);            \
    else hipLaunchKernelGGL((gram_kernel<T, FA_, FB_, false>), grid, block, 0, st, k);              \
  } while (0)
  if (d->dtype == MI_F32) { if (g.FA == 4) GRAM_CASE(float, 4, 4); else GRAM_CASE(float, 2, 2); }
  else if (g.FA == 3 && g.FB == 3) GRAM_CASE(bf16, 3, 3);
  else if (g.FA == 3 && g.FB == 8) GRAM_CASE(bf16, 3, 8);
  else if (g.FA == 4 && g.FB == 6) GRAM_CASE(bf16, 4, 6);
  else if (g.FA == 6 && g.FB == 4) GRAM_CASE(bf16, 6, 4);
  else { if (g.FA == 4) GRAM_CASE(bf16, 4, 4); else GRAM_CASE(bf16, 2, 2); }
#undef GRAM_CASE
  }
  MI_LAUNCH_CHECK();
  if (deferred) {
    const int64_t per = (int64_t)d->ma * d->mb;
    MI_TRY(launch_reduce_rows(k.part, d->out, (int64_t)g.splits * (g.fold ? 1 : d->batch), per, per, d->accumulate, 1.0f, st));
  } else if (!direct) {
    ProfScope ps2(st, K_GRAM_REDUCE, 4.0 * (g.splits + 1) * g.Z * d->ma * d->mb, (double)g.splits * g.Z * d->ma * d->mb);
    const int zo = d->sum_batch ? d->groups : g.Z;
    const int64_t per = (int64_t)d->ma * d->mb;
    launch_gram_reduce(k.part, d->out, g.splits, g.fold ? 1 : d->batch, d->groups, d->ma, d->mb, d->out_ld, d->out_zs, d->sum_batch,
                       d->accumulate, zo, st);
    MI_LAUNCH_CHECK();
  }
  if (ss) {
    const int64_t cols = (int64_t)g.Z * (d->ma + d->mb);
    MI_TRY(launch_reduce_rows(k.ss_part, d->sumsq, g.splits, cols, cols, 0, 1.0f, st));
  }
  return MI_OK;
}
