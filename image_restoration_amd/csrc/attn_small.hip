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

constexpr int ATTN_MAX_C = 128;  // 16x16 threads x 8x8 register tile in the backward glue
static inline int attn_ld(int c) { return c + 1; }

__global__ __launch_bounds__(256) void attn_fold_kernel(const float* __restrict__ graw, const float* __restrict__ ss,
                                                        const float* __restrict__ temperature, const float* __restrict__ wo,
                                                        float* __restrict__ P, float* __restrict__ A, float* __restrict__ nrm,
                                                        float* __restrict__ M, int C, int heads, int ld) {
  extern __shared__ float sm[];  // [c][ld]
  const int c = C / heads;
  const int z = blockIdx.x, b = z / heads, h = z - b * heads;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const float temp = temperature[h];
  const float* gz = graw + (int64_t)z * c * c;
  const float* sz = ss + (int64_t)z * 2 * c;
  for (int e = t; e < 2 * c; e += 256) nrm[(int64_t)z * 2 * c + e] = fmaxf(sqrtf(sz[e]), NORM_EPS);
  for (int e = t; e < c * c; e += 256) {
    const int i = e / c, j = e - i * c;
    const float nq = fmaxf(sqrtf(sz[i]), NORM_EPS), nk = fmaxf(sqrtf(sz[c + j]), NORM_EPS);
    const float pv = gz[e] / (nq * nk);
    P[(int64_t)z * c * c + e] = pv;
    sm[i * ld + j] = pv * temp;
  }
  __syncthreads();
  for (int i = wv; i < c; i += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < c; j += 64) mx = fmaxf(mx, sm[i * ld + j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < c; j += 64) {
      const float ev = expf(sm[i * ld + j] - mx);
      sm[i * ld + j] = ev;
      sum += ev;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < c; j += 64) {
      const float a = sm[i * ld + j] * inv;
      sm[i * ld + j] = a;
      A[(int64_t)z * c * c + i * c + j] = a;
    }
  }
  __syncthreads();
  // M[b][r][h*c+j] = sum_i wo[r][h*c+i] * A[i][j]
  float* Mb = M + (int64_t)b * C * C;
  for (int e = t; e < C * c; e += 256) {
    const int r = e / c, j = e - r * c;
    const float* wrow = wo + (int64_t)r * C + h * c;
    float acc = 0.f;
    for (int i = 0; i < c; ++i) acc += wrow[i] * sm[i * ld + j];
    Mb[(int64_t)r * C + h * c + j] = acc;
  }
}

// Backward glue.  256 threads form a 16x16 grid; thread (ti,tj) owns dA/dS entries (ti+16a, tj+16b), a,b < 8
// (c <= 128), kept in registers.  W_o and dM column blocks stream through LDS in RC-row chunks (coalesced rows
// of c floats), A stays in LDS for the whole kernel.
constexpr int ATT_RC = 16;
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ dM, const float* __restrict__ A,
                                                             const float* __restrict__ P, const float* __restrict__ nrm,
                                                             const float* __restrict__ temperature,
                                                             const float* __restrict__ wo, float* __restrict__ dwo_part,
                                                             float* __restrict__ dtemp_part, float* __restrict__ wdq,
                                                             float* __restrict__ wdk, int C, int heads, int ld) {
  extern __shared__ float sm[];
  const int c = C / heads;
  float* As = sm;                          // [c][ld]
  float* Wt = As + c * ld;                 // [RC][c]   W_o[r][h*c + i]
  float* Dt = Wt + ATT_RC * c;             // [RC][c]   dM[r][h*c + j]
  float* colp = Dt + ATT_RC * c;           // [16][c]   per-ti partial column sums
  float* rqs = colp + 16 * c;              // [c]
  float* rks = rqs + c;                    // [c]
  float* red = rks + c;                    // [4]
  const int z = blockIdx.x, b = z / heads, h = z - b * heads;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ti = t >> 4, tj = t & 15;
  const float temp = temperature[h];
  const float* dMb = dM + (int64_t)b * C * C;
  const float* Az = A + (int64_t)z * c * c;
  const float* Pz = P + (int64_t)z * c * c;
  const float* nz = nrm + (int64_t)z * 2 * c;
  float* dwo = dwo_part + (int64_t)b * C * C;

  for (int e = t; e < c * c; e += 256) { const int i = e / c; As[i * ld + (e - i * c)] = Az[e]; }

  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) acc[a][bb] = 0.f;

  for (int r0 = 0; r0 < C; r0 += ATT_RC) {
    __syncthreads();  // previous chunk fully consumed (and As complete on the first pass)
    for (int e = t; e < ATT_RC * c; e += 256) {
      const int rr = e / c, col = e - rr * c;
      const int r = r0 + rr;
      Wt[e] = r < C ? wo[(int64_t)r * C + h * c + col] : 0.f;
      Dt[e] = r < C ? dMb[(int64_t)r * C + h * c + col] : 0.f;
    }
    __syncthreads();
    // dA[i][j] += sum_rr Wt[rr][i] * Dt[rr][j]
#pragma unroll 4
    for (int rr = 0; rr < ATT_RC; ++rr) {
      float wa[8], db[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) wa[a] = (ti + 16 * a < c) ? Wt[rr * c + ti + 16 * a] : 0.f;
#pragma unroll
      for (int bb = 0; bb < 8; ++bb) db[bb] = (tj + 16 * bb < c) ? Dt[rr * c + tj + 16 * bb] : 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) acc[a][bb] += wa[a] * db[bb];
    }
    // dWo_part[b][r][h*c+i] = sum_j dM[r][h*c+j] * A[i][j]   for the RC rows of this chunk
    for (int e = t; e < ATT_RC * c; e += 256) {
      const int rr = e / c, i = e - rr * c;
      const int r = r0 + rr;
      if (r < C) {
        float s2 = 0.f;
        for (int j = 0; j < c; ++j) s2 += Dt[rr * c + j] * As[i * ld + j];
        dwo[(int64_t)r * C + h * c + i] = s2;
      }
    }
  }

  // softmax backward: dS = A * (dA - rowdot),  rowdot_i = sum_j dA_ij A_ij  (16 lanes share one ti)
  float tsum = 0.f;
  float colsum[8];
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) colsum[bb] = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int i = ti + 16 * a;
    float dot = 0.f;
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int j = tj + 16 * bb;
      if (i < c && j < c) dot += acc[a][bb] * As[i * ld + j];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    float rqa = 0.f;
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int j = tj + 16 * bb;
      float ds = 0.f;
      if (i < c && j < c) {
        ds = As[i * ld + j] * (acc[a][bb] - dot);
        const float pv = Pz[i * c + j];
        tsum += ds * pv;
        rqa += ds * pv * temp;
        colsum[bb] += ds * pv * temp;
      }
      acc[a][bb] = ds;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) rqa += __shfl_xor(rqa, o, 64);
    if (tj == 0 && i < c) rqs[i] = rqa;
  }
#pragma unroll
  for (int bb = 0; bb < 8; ++bb) {
    const int j = tj + 16 * bb;
    if (j < c) colp[ti * c + j] = colsum[bb];
  }
  tsum = wave_sum(tsum);
  if (lane == 0) red[wv] = tsum;
  __syncthreads();
  if (t == 0) dtemp_part[z] = (red[0] + red[1]) + (red[2] + red[3]);
  for (int j = t; j < c; j += 256) {
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s2 += colp[k * c + j];
    rks[j] = s2;
  }
  __syncthreads();
  float* wq = wdq + (int64_t)z * c * 2 * c;
  float* wk = wdk + (int64_t)z * c * 2 * c;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int i = ti + 16 * a;
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int j = tj + 16 * bb;
      if (i >= c || j >= c) continue;
      const float nq = nz[i], nk = nz[c + j];
      const float g1 = temp * acc[a][bb] / (nq * nk);
      wq[i * 2 * c + j] = g1;            // dq_i += g1 * k_j
      wk[j * 2 * c + i] = g1;            // dk_j += g1 * q_i
      // diagonal blocks: projection terms of d(x/|x|); zero when the norm was clamped
      const float nki = nz[c + i];
      wq[i * 2 * c + c + j] = (i == j) ? ((nq > NORM_EPS) ? -rqs[i] / (nq * nq) : 0.f) : 0.f;
      wk[i * 2 * c + c + j] = (i == j) ? ((nki > NORM_EPS) ? -rks[i] / (nki * nki) : 0.f) : 0.f;
    }
  }
}

// per-channel sum over batch and pixels:  part[split][c] = sum x[b][c][n-range]
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const T* __restrict__ x, float* __restrict__ part, int B, int C,
                                                       int64_t N, int64_t per_split) {
  __shared__ float red[4];
  const int c = blockIdx.x, sp = blockIdx.y;
  const int64_t nb = (int64_t)sp * per_split;
  int64_t ne = nb + per_split;
  if (ne > N) ne = N;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const T* row = x + ((int64_t)b * C + c) * N;
    for (int64_t n = nb + threadIdx.x; n < ne; n += 256) acc += ld1(row + n);
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
  const int64_t per = (N + splits - 1) / splits;
  float* part = (float*)ws;
  dim3 grid(C, splits), block(256);
  ProfScope ps(st, K_CHAN_SUM, (double)B * C * N * dtype_size(dtype), (double)B * C * N);
  if (dtype == MI_F32) hipLaunchKernelGGL((chan_sum_kernel<float>), grid, block, 0, st, (const float*)x, part, B, C, N, per);
  else hipLaunchKernelGGL((chan_sum_kernel<bf16>), grid, block, 0, st, (const bf16*)x, part, B, C, N, per);
  MI_LAUNCH_CHECK();
  return launch_reduce_rows(part, out, splits, C, C, accumulate, 1.0f, st);
}

int launch_attn_fold(const float* graw, const float* ss, const float* temperature, const float* wo, float* P, float* A,
                     float* nrm, float* M, int B, int C, int heads, hipStream_t st) {
  const int c = C / heads;
  MI_CHECK_ARG(c >= 1 && c <= ATTN_MAX_C && c * heads == C, "mdta: channels per head %d unsupported (1..%d)", c, ATTN_MAX_C);
  const int ld = attn_ld(c);
  const size_t lds = (size_t)c * ld * sizeof(float);
  if (lds > 64 * 1024)
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_fold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps(st, K_ATTN_FOLD, 4.0 * B * (3.0 * C * c + 2.0 * C * C), 2.0 * B * C * (double)c * C);
  hipLaunchKernelGGL(attn_fold_kernel, dim3(B * heads), dim3(256), lds, st, graw, ss, temperature, wo, P, A, nrm, M, C, heads,
                     ld);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

int launch_attn_bwd_small(const float* dM, const float* A, const float* P, const float* nrm, const float* temperature,
                          const float* wo, float* dwo_part, float* dtemp_part, float* wdq, float* wdk, int B, int C,
                          int heads, hipStream_t st) {
  const int c = C / heads;
  MI_CHECK_ARG(c >= 1 && c <= ATTN_MAX_C && c * heads == C, "mdta: channels per head %d unsupported (1..%d)", c, ATTN_MAX_C);
  const int ld = attn_ld(c);
  const size_t lds = ((size_t)c * ld + 2 * ATT_RC * c + 16 * c + 2 * c + 4) * sizeof(float);
  if (lds > 64 * 1024)
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps(st, K_ATTN_BWD_SMALL, 4.0 * B * (6.0 * C * c + 3.0 * C * C), 4.0 * B * C * (double)c * C);
  hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(B * heads), dim3(256), lds, st, dM, A, P, nrm, temperature, wo, dwo_part,
                     dtemp_part, wdq, wdk, C, heads, ld);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi
