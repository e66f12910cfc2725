// The c x c side of MDTA (c = C/heads <= 120): everything that happens between the long pixel-axis
// contractions (Restormer.py:121-127).  One workgroup per (image, head); fp32 throughout.
//
// forward  : cosine P = (q k^T)/(|q||k|), S = temperature*P, A = softmax_row(S), and the fold
//            M_b[:, head cols] = W_o[:, head cols] * A   so that  project_out(attn @ v) == M_b @ v
//            (one 1x1 GEMM per image instead of attn@v followed by project_out).
// backward : from dM_b = dY V^T:  dW_o, dA, dS (softmax bwd), d temperature, and the small matrices
//            that turn the gradients of the L2-normalised q,k into two more per-image 1x1 GEMMs:
//              dq = G1 k + diag(D1) q,  dk = G1^T q + diag(D2) k
//            with G1 = temperature*dS/(|q_i||k_j|), D1_i = -sum_j dS_ij S_ij/|q_i|^2, D2_j likewise.
#include "internal.h"

namespace mi {

constexpr float NORM_EPS = 1e-12f;  // F.normalize eps (Restormer.py:121-122)
constexpr int ATTN_MAX_C = 128;     // 16x16 threads x (up to) 8x8 register tile
constexpr int ATT_RC = 16;          // rows of W_o / dM handled by one workgroup

static inline int attn_ld(int c) { return c + 1; }
static inline int attn_ct(int c) { return (c + 15) / 16; }
static inline int attn_rchunks(int C) { return (C + ATT_RC - 1) / ATT_RC; }

// All-reduce over the 16 lanes of a DPP row (xor-1, xor-2 inside the quad, half-row mirror, row mirror): every lane ends up
// with the row's result, no LDS-pipe permutes and no readlane.
#define MI_ROW16_STEP(OP, ctrl)                                                                                    \
  v = OP(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                  ctrl, 0xf, 0xf, true)))
__device__ __forceinline__ float row16_sum(float v) {
#define MI_ADDF(a, b) ((a) + (b))
  MI_ROW16_STEP(MI_ADDF, 0xB1); MI_ROW16_STEP(MI_ADDF, 0x4E); MI_ROW16_STEP(MI_ADDF, 0x141); MI_ROW16_STEP(MI_ADDF, 0x140);
#undef MI_ADDF
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  MI_ROW16_STEP(fmaxf, 0xB1); MI_ROW16_STEP(fmaxf, 0x4E); MI_ROW16_STEP(fmaxf, 0x141); MI_ROW16_STEP(fmaxf, 0x140);
  return v;
}
#undef MI_ROW16_STEP

// LDS floats of attn_fold_kernel<CT> with rpw row chunks of W_o per workgroup
static inline size_t attn_fold_lds_floats(int ct, int rpw) {
  const size_t cp = 16 * (size_t)ct;
  return cp * (cp + 1) + 2 * cp + (size_t)rpw * ATT_RC * (cp + 1);
}

// forward: grid (Z, ceil(C/16 / rpw)).  Every workgroup redoes the c x c cosine / temperature / row softmax of its (image, head)
// and folds rpw chunks of 16 rows of W_o:  M[b][r][h*c+j] = sum_i W_o[r][h*c+i] A[i][j].  The workgroups of row group 0 also
// store P, A and the norms.
//  * softmax: a 16-lane DPP row owns a matrix row (lane l16 holds columns l16 + 16 q), 16 matrix rows per pass, CT passes; all
//    global loads are issued up front, max / sum are 4-step DPP all-reduces, the 2c inverse norms come from LDS.  (The first
//    form - one wave per row, integer divisions to find (i, j), three sqrt / div per element - took 29 us at c = 96 and 57 us
//    at C = 384 where 24 row chunks each redid it: profiles/r04_f_attn_small.txt.)
//  * fold: fp32 MFMA (16x16x4): a wave owns 16 x 16 output tiles, both operands read from LDS one float per lane and step.
template <int CT>
__global__ __launch_bounds__(256) void attn_fold_kernel(const float* __restrict__ graw, const float* __restrict__ ss,
                                                        const float* __restrict__ temperature, const float* __restrict__ wo,
                                                        float* __restrict__ P, float* __restrict__ A, float* __restrict__ nrm,
                                                        float* __restrict__ M, bf16* __restrict__ Mb, bf16* __restrict__ Mtb, int C,
                                                        int heads, int rpw) {
  constexpr int CP = 16 * CT, LD = CP + 1;
  extern __shared__ float sm[];
  float* As = sm;                       // [CP][LD]; rows and columns >= c are zero
  float* inr = As + CP * LD;            // [2 CP] 1 / max(|q_i|, eps), 1 / max(|k_j|, eps)
  float* Wt = inr + 2 * CP;             // [rpw * 16][LD]; columns >= c and rows >= C are zero
  const int c = C / heads;
  const int z = blockIdx.x, b = z / heads, h = z - b * heads;
  const int rch = (C + ATT_RC - 1) / ATT_RC;
  const int rc0 = blockIdx.y * rpw;
  const int nrc = rch - rc0 < rpw ? rch - rc0 : rpw;     // row chunks of this workgroup
  const int t = threadIdx.x, l16 = t & 15, g16 = t >> 4;
  const bool first = blockIdx.y == 0;
  const float* gz = graw + (int64_t)z * c * c;
  const float* sz = ss + (int64_t)z * 2 * c;

  float gv[CT][CT];
#pragma unroll
  for (int it = 0; it < CT; ++it) {
    const int i = g16 + 16 * it;
#pragma unroll
    for (int q = 0; q < CT; ++q) {
      const int j = l16 + 16 * q;
      gv[it][q] = (i < c && j < c) ? gz[i * c + j] : 0.f;
    }
  }
  for (int e = t; e < nrc * ATT_RC * CP; e += 256) {
    const int rr = e / CP, col = e - rr * CP, r = rc0 * ATT_RC + rr;
    Wt[rr * LD + col] = (r < C && col < c) ? wo[(int64_t)r * C + h * c + col] : 0.f;
  }
  for (int e = t; e < 2 * CP; e += 256) {
    const int k = e < CP ? e : e - CP;
    float v = 0.f;
    if (k < c) {
      const float n = fmaxf(sqrtf(sz[(e < CP ? 0 : c) + k]), NORM_EPS);
      if (first) nrm[(int64_t)z * 2 * c + (e < CP ? 0 : c) + k] = n;
      v = 1.0f / n;
    }
    inr[e] = v;
  }
  __syncthreads();
  const float temp = temperature[h];
  float* Pz = P + (int64_t)z * c * c;
  float* Az = A + (int64_t)z * c * c;
#pragma unroll
  for (int it = 0; it < CT; ++it) {
    const int i = g16 + 16 * it;
    const bool rok = i < c;
    const float iq = inr[i];
    float sv[CT], mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < CT; ++q) {
      const int j = l16 + 16 * q;
      const float pv = gv[it][q] * iq * inr[CP + j];
      if (first && rok && j < c) Pz[i * c + j] = pv;
      sv[q] = (j < c || !rok) ? pv * temp : -INFINITY;     // (padding rows: all zeros, kept finite)
      mx = fmaxf(mx, sv[q]);
    }
    mx = row16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < CT; ++q) { sv[q] = expf(sv[q] - mx); sum += sv[q]; }
    sum = row16_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int q = 0; q < CT; ++q) {
      const int j = l16 + 16 * q;
      const bool in = rok && j < c;
      const float av = in ? sv[q] * inv : 0.f;
      As[i * LD + j] = av;
      if (first && in) Az[i * c + j] = av;
    }
  }
  __syncthreads();

  const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, li = lane & 15, kq = lane >> 4;
  const bool tb_vec = (C & 3) == 0;
  for (int tile = wv; tile < nrc * CT; tile += 4) {
    const int rr = tile / CT, nt = tile - rr * CT;
    const float* wp = Wt + (rr * ATT_RC + li) * LD + kq;
    const float* ap = As + kq * LD + 16 * nt + li;
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < CP / 4; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[4 * ks], ap[4 * ks * LD], d, 0, 0, 0);
    // lane holds M[r .. r + 3][h c + j],  r = 16 (rc0 + rr) + 4 kq,  j = 16 nt + li
    const int r = (rc0 + rr) * ATT_RC + 4 * kq, j = 16 * nt + li;
    if (j < c) {
      const int64_t mb = (int64_t)b * C * C;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (r + i < C) {
          M[mb + (int64_t)(r + i) * C + h * c + j] = d[i];
          // bf16 copies for the per-image-weight GEMMs (mi_pw_desc.w_b16): M_b for out = M_b v, its transpose for dv = M_b^T dy
          if (Mb) Mb[mb + (int64_t)(r + i) * C + h * c + j] = (bf16)d[i];
        }
      if (Mtb) {
        bf16* tp = Mtb + mb + (int64_t)(h * c + j) * C + r;
        if (tb_vec && r + 3 < C) {
          *reinterpret_cast<u32x2*>(tp) = (u32x2){cvt_pk_bf16(d[0], d[1]), cvt_pk_bf16(d[2], d[3])};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (r + i < C) tp[i] = (bf16)d[i];
        }
      }
    }
  }
}

// backward of the c x c side, one launch: grid (Z, 1 + ceil(C / 64)).
//  y == 0: dA = W_o[:, head]^T dM[:, head] over all C rows (fp32 MFMA, both operands staged through LDS in chunks of 32 rows), then in
//          registers: softmax backward dS = A (dA - rowdot), d temperature, and the per-image weights of the gradient GEMM(s)
//          dq = G1 k + D1 q,  dk = G1^T q + D2 k  with G1 = temperature dS/(|q_i||k_j|), D1_i = -sum_j dS_ij S_ij/|q_i|^2,
//          D2_j = -sum_i dS_ij S_ij/|k_j|^2, as ONE [2c][2c] matrix per (image, head) acting on the stacked operand [k; q]:
//            rows 0..c-1  (dq): [ G1    | D1 ]        rows c..2c-1 (dk): [ D2 | G1^T ]
//          so that both gradients come out of one pass over q and k (mi_pw_desc.y_split), or of two GEMMs over row halves.
//          Wave w owns the row blocks mt = w, w + 4 of dA (all their column tiles): row sums are 16-lane DPP all-reduces.
//  y >= 1: 64 rows of dW_o:  dWo_part[b][r][h*c+i] = sum_j dM[r][h*c+j] A[i][j]   (fp32 MFMA, 16 rows per wave).
// (The first form - a partial kernel per 16-row chunk writing c x c partials of dA, CT x CT outer products on the VALU, and a
//  finishing kernel summing them - took 65 us at c = 96 and 67 us at C = 384: profiles/r04_f_attn_small.txt.)
constexpr int ATT_KR = 32;    // rows of W_o / dM per staged chunk (y == 0)
constexpr int ATT_RW = 64;    // rows of dW_o per workgroup (y >= 1)
static inline size_t attn_bwd_lds_floats(int ct) {
  const size_t cp = 16 * (size_t)ct, ld = cp + 1;
  const size_t r0 = 2 * ATT_KR * ld + 16 * cp + 2 * cp + 8, r1 = cp * ld + ATT_RW * ld;
  return r0 > r1 ? r0 : r1;
}
template <int CT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ dM, const float* __restrict__ A,
                                                       const float* __restrict__ P, const float* __restrict__ nrm,
                                                       const float* __restrict__ temperature, const float* __restrict__ wo,
                                                       float* __restrict__ dwo_part, float* __restrict__ dtemp_part,
                                                       float* __restrict__ wd, bf16* __restrict__ wdb, int C, int heads) {
  constexpr int CP = 16 * CT, LD = CP + 1, MB = (CT + 3) / 4;
  extern __shared__ float sm[];
  const int c = C / heads;
  const int z = blockIdx.x, b = z / heads, h = z - b * heads;
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, kq = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const float* dMb = dM + (int64_t)b * C * C + h * c;      // column block of this head
  const float* Az = A + (int64_t)z * c * c;

  if (blockIdx.y > 0) {
    // ---- rows of dW_o
    float* As = sm;                 // [CP][LD]  A, zero padded
    float* Ds = As + CP * LD;       // [RW][LD]  dM rows, zero padded
    const int r0 = (blockIdx.y - 1) * ATT_RW;
    for (int e = t; e < CP * CP; e += 256) {
      const int i = e / CP, j = e - i * CP;
      As[i * LD + j] = (i < c && j < c) ? Az[i * c + j] : 0.f;
    }
    for (int e = t; e < ATT_RW * CP; e += 256) {
      const int rr = e / CP, j = e - rr * CP;
      Ds[rr * LD + j] = (r0 + rr < C && j < c) ? dMb[(int64_t)(r0 + rr) * C + j] : 0.f;
    }
    __syncthreads();
    if (r0 + 16 * wv >= C) return;
    const float* dp = Ds + (16 * wv + li) * LD + kq;
    f32x4 d[CT];
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) d[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int ks = 0; ks < CP / 4; ++ks) {
      const float av = dp[4 * ks];
#pragma unroll
      for (int nt = 0; nt < CT; ++nt)
        d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, As[(16 * nt + li) * LD + 4 * ks + kq], d[nt], 0, 0, 0);
    }
    float* op = dwo_part + (int64_t)b * C * C + h * c;
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) {
      const int i = 16 * nt + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * wv + 4 * kq + r;
        if (row < C && i < c) op[(int64_t)row * C + i] = d[nt][r];
      }
    }
    return;
  }

  // ---- dA and everything behind it
  float* Ws = sm;                      // [KR][LD]
  float* Ds = Ws + ATT_KR * LD;        // [KR][LD]
  float* colp = Ds + ATT_KR * LD;      // [16][CP]
  float* rks = colp + 16 * CP;         // [CP]
  float* red = rks + 2 * CP;           // [4]
  const float* wob = wo + h * c;
  constexpr int NLD = (ATT_KR * CP + 255) / 256;     // staged elements per thread and operand
  float wr[NLD], dr[NLD];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int e = t + 256 * u, rr = e / CP, j = e - rr * CP;
      const bool in = e < ATT_KR * CP && k0 + rr < C && j < c;
      wr[u] = in ? wob[(int64_t)(k0 + rr) * C + j] : 0.f;
      dr[u] = in ? dMb[(int64_t)(k0 + rr) * C + j] : 0.f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int e = t + 256 * u, rr = e / CP, j = e - rr * CP;
      if (e < ATT_KR * CP) { Ws[rr * LD + j] = wr[u]; Ds[rr * LD + j] = dr[u]; }
    }
  };
  fetch(0);
  // A and P in the accumulator layout: rows 16 mt + 4 kq + r, column 16 nt + li
  float av[MB][CT][4], pv[MB][CT][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nt = 0; nt < CT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * (wv + 4 * mb) + 4 * kq + r, j = 16 * nt + li;
        const bool in = i < c && j < c;
        av[mb][nt][r] = in ? Az[i * c + j] : 0.f;
        pv[mb][nt][r] = in ? P[(int64_t)z * c * c + i * c + j] : 0.f;
      }
  f32x4 acc[MB][CT];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) acc[mb][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < C; k0 += ATT_KR) {
    __syncthreads();
    stage();
    __syncthreads();
    if (k0 + ATT_KR < C) fetch(k0 + ATT_KR);
#pragma unroll
    for (int ks = 0; ks < ATT_KR / 4; ++ks) {
      float bv[CT];
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) bv[nt] = Ds[(4 * ks + kq) * LD + 16 * nt + li];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int mt = wv + 4 * mb;
        if (mt < CT) {
          const float a = Ws[(4 * ks + kq) * LD + 16 * mt + li];
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) acc[mb][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[nt], acc[mb][nt], 0, 0, 0);
        }
      }
    }
  }
  const float temp = temperature[h];
  const float* nz = nrm + (int64_t)z * 2 * c;
  float tsum = 0.f, colsum[CT], rq[MB][4];
#pragma unroll
  for (int nt = 0; nt < CT; ++nt) colsum[nt] = 0.f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float dot = 0.f;
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) dot += acc[mb][nt][r] * av[mb][nt][r];
      dot = row16_sum(dot);
      float rqa = 0.f;
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) {
        const float ds = av[mb][nt][r] * (acc[mb][nt][r] - dot);      // (zero outside the c x c block: av is)
        const float sp = ds * pv[mb][nt][r];
        tsum += sp;
        rqa += sp * temp;
        colsum[nt] += sp * temp;
        acc[mb][nt][r] = ds;
      }
      rq[mb][r] = row16_sum(rqa);
    }
#pragma unroll
  for (int nt = 0; nt < CT; ++nt) colp[(4 * wv + kq) * CP + 16 * nt + li] = colsum[nt];
  tsum = wave_sum(tsum);
  if (lane == 0) red[wv] = tsum;
  __syncthreads();
  if (t == 0) dtemp_part[z] = (red[0] + red[1]) + (red[2] + red[3]);
  for (int j = t; j < CP; j += 256) {
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s2 += colp[k * CP + j];
    rks[j] = s2;
  }
  __syncthreads();
  float* wq = wd + (int64_t)z * 2 * c * 2 * c;     // rows of dq
  float* wk = wq + (int64_t)c * 2 * c;            // rows of dk
  bf16* bq = wdb ? wdb + (int64_t)z * 2 * c * 2 * c : nullptr;
  bf16* bk = wdb ? bq + (int64_t)c * 2 * c : nullptr;
  const bool vec = (c & 3) == 0;                  // 4 consecutive rows of a lane: one vector store into the transposed blocks
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int mt = wv + 4 * mb;
    if (mt >= CT) continue;
    const int i0 = 16 * mt + 4 * kq;
    float nq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) nq[r] = i0 + r < c ? nz[i0 + r] : 1.f;
#pragma unroll
    for (int nt = 0; nt < CT; ++nt) {
      const int j = 16 * nt + li;
      if (j >= c) continue;
      const float nk = nz[c + j];
      float g1[4], dq_[4], dk_[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + r;
        g1[r] = temp * acc[mb][nt][r] / (nq[r] * nk);
        // diagonal blocks: projection terms of d(x/|x|); zero when the norm was clamped
        const bool dg = i == j;
        dq_[r] = (dg && nq[r] > NORM_EPS) ? -rq[mb][r] / (nq[r] * nq[r]) : 0.f;
        dk_[r] = (dg && nk > NORM_EPS) ? -rks[j] / (nk * nk) : 0.f;
        if (i < c) {
          wq[i * 2 * c + j] = g1[r];            // dq_i += g1 * k_j
          wq[i * 2 * c + c + j] = dq_[r];
          wk[i * 2 * c + j] = dk_[r];
          if (bq) { bq[i * 2 * c + j] = (bf16)g1[r]; bq[i * 2 * c + c + j] = (bf16)dq_[r]; bk[i * 2 * c + j] = (bf16)dk_[r]; }
        }
      }
      // dk_j += g1 * q_i: the transposed block, 4 consecutive i per lane
      if (vec && i0 + 3 < c) {
        *reinterpret_cast<f32x4*>(wk + j * 2 * c + c + i0) = (f32x4){g1[0], g1[1], g1[2], g1[3]};
        if (bk) *reinterpret_cast<u32x2*>(bk + j * 2 * c + c + i0) = (u32x2){cvt_pk_bf16(g1[0], g1[1]), cvt_pk_bf16(g1[2], g1[3])};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (i0 + r < c) {
            wk[j * 2 * c + c + i0 + r] = g1[r];
            if (bk) bk[j * 2 * c + c + i0 + r] = (bf16)g1[r];
          }
      }
    }
  }
}

// per-channel sum over batch and pixels:  part[split][c] = sum x[b][c][n-range]
// The (image, vector) items of a workgroup form one index space walked with four loads in flight per thread: as a loop over the
// images with one dependent load each this kernel took an HBM round trip per image (19 us per call at bs 8 on MoCE-IR's planes).
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const T* __restrict__ x, float* __restrict__ part, int B, int C,
                                                       int64_t N, int64_t per_split) {
  __shared__ float red[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int64_t nb = (int64_t)sp * per_split;
  int64_t ne = nb + per_split;
  if (ne > N) ne = N;
  float acc = 0.f;
  constexpr int V = 16 / (int)sizeof(T);                // elements per 16-byte load
  // (vec: whole 16-byte vectors - N, the split bounds and the base pointer allow it; a 2-byte load per lane made this kernel
  //  instruction-bound: 34 us per call on MoCE-IR's planes, 7 % of its step)
  const bool vec = (N % V == 0) && (nb % V == 0) && (ne % V == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec && ne > nb) {
    const int64_t nvec = (ne - nb) / V, items = (int64_t)B * nvec;
    const T* base = x + (int64_t)c * N + nb;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i0 = threadIdx.x; i0 < items; i0 += 4 * 256) {
      float v[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + 256 * u;
        const int64_t b = i / nvec, k = i - b * nvec;
        if (i < items) Vec<T, V>::ld(base + b * C * N + k * V, v[u]);
        else {
#pragma unroll
          for (int j = 0; j < V; ++j) v[u][j] = 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) a4[u] += v[u][j];
    }
    acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  } else {
    for (int b = 0; b < B; ++b) {
      const T* row = x + ((int64_t)b * C + c) * N;
      for (int64_t n = nb + threadIdx.x; n < ne; n += 256) acc += ld1(row + n);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)sp * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

int chan_sum_splits(int C, int64_t N) {
  int s = 512 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  const int maxs = cdiv(N, 1024);
  if (s > maxs) s = maxs;
  return s < 1 ? 1 : s;
}
size_t chan_sum_workspace(int C, int64_t N) { return align_up((size_t)chan_sum_splits(C, N) * C * sizeof(float), 256); }

int launch_chan_sum(const void* x, float* out, int B, int C, int64_t N, int dtype, int accumulate, void* ws, hipStream_t st) {
  const int splits = chan_sum_splits(C, N);
  int64_t per = (N + splits - 1) / splits;
  per = (per + 7) / 8 * 8;                              // split bounds on 16-byte vectors (the last split takes the remainder)
  float* part = (float*)ws;
  if (accumulate) {          // a parameter gradient accumulated in place: the split sum may wait for mi_deferred_flush (common.h)
    float* arena = deferred_take((size_t)splits * C, st);
    if (arena) part = arena;
  }
  dim3 grid(C, splits), block(256);
  ProfScope ps(st, K_CHAN_SUM, (double)B * C * N * dtype_size(dtype), (double)B * C * N);
  if (dtype == MI_F32) hipLaunchKernelGGL((chan_sum_kernel<float>), grid, block, 0, st, (const float*)x, part, B, C, N, per);
  else hipLaunchKernelGGL((chan_sum_kernel<bf16>), grid, block, 0, st, (const bf16*)x, part, B, C, N, per);
  MI_LAUNCH_CHECK();
  return launch_reduce_rows(part, out, splits, C, C, accumulate, 1.0f, st);
}

#define ATTN_CT_SWITCH(ct, CALL)                                       \
  do {                                                                 \
    switch (ct) {                                                      \
      case 1: { constexpr int CT = 1; CALL; } break;                   \
      case 2: { constexpr int CT = 2; CALL; } break;                   \
      case 3: { constexpr int CT = 3; CALL; } break;                   \
      case 4: { constexpr int CT = 4; CALL; } break;                   \
      case 5: case 6: { constexpr int CT = 6; CALL; } break;           \
      default: { constexpr int CT = 8; CALL; } break;                  \
    }                                                                  \
  } while (0)

size_t attn_bwd_scratch_floats(int B, int C, int heads) {
  (void)B; (void)C; (void)heads;
  return 64;            // (the one-launch backward keeps dA in registers: nothing to stage; the carve stays for the ABI)
}

int launch_attn_fold(const float* graw, const float* ss, const float* temperature, const float* wo, float* P, float* A,
                     float* nrm, float* M, int B, int C, int heads, hipStream_t st, void* Mb, void* Mtb) {
  const int c = C / heads;
  MI_CHECK_ARG(c >= 1 && c <= ATTN_MAX_C && c * heads == C, "mdta: channels per head %d unsupported (1..%d)", c, ATTN_MAX_C);
  MI_CHECK_ARG(graw && ss && temperature && wo && P && A && nrm && M, "mdta: null pointer in the attention fold");
  // row chunks per workgroup: every workgroup redoes the softmax, so deep levels (many heads, many chunks) fold several chunks
  // each; about 768 workgroups (3 per CU) are kept
  const int Z = B * heads, rch = attn_rchunks(C);
  int rpw = (int)(((int64_t)Z * rch) / 768);
  rpw = rpw < 1 ? 1 : (rpw > rch ? rch : (rpw > 8 ? 8 : rpw));
  dim3 grid(Z, cdiv(rch, rpw));
  ProfScope ps(st, K_ATTN_FOLD, 4.0 * B * (3.0 * C * c + 2.0 * C * C), 2.0 * B * C * (double)c * C);
  ATTN_CT_SWITCH(attn_ct(c), {
    const size_t lds = attn_fold_lds_floats(CT, rpw) * sizeof(float);
    if (lds > 64 * 1024)
      MI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fold_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((attn_fold_kernel<CT>), grid, dim3(256), lds, st, graw, ss, temperature, wo, P, A, nrm, M, (bf16*)Mb, (bf16*)Mtb, C,
                       heads, rpw);
  });
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int launch_attn_bwd_small(const float* dM, const float* A, const float* P, const float* nrm, const float* temperature,
                          const float* wo, float* dwo_part, float* dtemp_part, float* wd, float* scratch,
                          int B, int C, int heads, hipStream_t st, void* wdb) {
  (void)scratch;
  const int c = C / heads;
  MI_CHECK_ARG(c >= 1 && c <= ATTN_MAX_C && c * heads == C, "mdta: channels per head %d unsupported (1..%d)", c, ATTN_MAX_C);
  MI_CHECK_ARG(dM && A && P && nrm && temperature && wo && dwo_part && dtemp_part && wd, "mdta: null pointer in the attention backward");
  ProfScope ps(st, K_ATTN_BWD_SMALL, 4.0 * B * (6.0 * C * c + 3.0 * C * C), 4.0 * B * C * (double)c * C);
  ATTN_CT_SWITCH(attn_ct(c), {
    const size_t lds = attn_bwd_lds_floats(CT) * sizeof(float);
    if (lds > 64 * 1024)
      MI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((attn_bwd_kernel<CT>), dim3(B * heads, 1 + cdiv(C, ATT_RW)), dim3(256), lds, st, dM, A, P, nrm, temperature, wo,
                       dwo_part, dtemp_part, wd, (bf16*)wdb, C, heads);
  });
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi

// Per-channel sum over batch and pixels - the bias gradient of any conv (nn.Conv2d(bias=True): Restormer.py:82-86 with bias,
// moce_ir.py expert / decoder projections): out[c] (+)= sum_{b,n} x[b][c][n].  ws: mi_chan_sum_workspace(C, N) bytes.
extern "C" size_t mi_chan_sum_workspace(int C, int64_t N) { return (C > 0 && N > 0) ? mi::chan_sum_workspace(C, N) : 0; }
extern "C" int mi_chan_sum(const void* x, float* out, int B, int C, int64_t N, int dtype, int accumulate, void* ws, void* stream) {
  MI_CHECK_ARG(x && out && ws && B >= 1 && C >= 1 && N >= 1, "chan_sum: null pointer / bad shape");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "chan_sum: bad dtype %d", dtype);
  return mi::launch_chan_sum(x, out, B, C, N, dtype, accumulate, ws, (hipStream_t)stream);
}
