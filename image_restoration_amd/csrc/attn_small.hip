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

constexpr int ATTN_MAX_C = 120;  // c*(c+1)*4 bytes of dynamic LDS + the small static arrays must stay under 64 KiB
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

__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ dM, const float* __restrict__ A,
                                                             const float* __restrict__ P, const float* __restrict__ nrm,
                                                             const float* __restrict__ temperature,
                                                             const float* __restrict__ wo, float* __restrict__ dwo_part,
                                                             float* __restrict__ dtemp_part, float* __restrict__ wdq,
                                                             float* __restrict__ wdk, int C, int heads, int ld) {
  extern __shared__ float sm[];  // dA -> dS, [c][ld]
  __shared__ float rq[128], rk[128], red[4];
  const int c = C / heads;
  const int z = blockIdx.x, b = z / heads, h = z - b * heads;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const float temp = temperature[h];
  const float* dMb = dM + (int64_t)b * C * C;
  const float* Az = A + (int64_t)z * c * c;
  const float* Pz = P + (int64_t)z * c * c;
  const float* nz = nrm + (int64_t)z * 2 * c;

  // dWo_part[b][r][h*c+i] = sum_j dM[r][h*c+j] * A[i][j]
  float* dwo = dwo_part + (int64_t)b * C * C;
  for (int e = t; e < C * c; e += 256) {
    const int r = e / c, i = e - r * c;
    const float* drow = dMb + (int64_t)r * C + h * c;
    float acc = 0.f;
    for (int j = 0; j < c; ++j) acc += drow[j] * Az[i * c + j];
    dwo[(int64_t)r * C + h * c + i] = acc;
  }
  // dA[i][j] = sum_r wo[r][h*c+i] * dM[r][h*c+j]
  for (int e = t; e < c * c; e += 256) {
    const int i = e / c, j = e - i * c;
    float acc = 0.f;
    for (int r = 0; r < C; ++r) acc += wo[(int64_t)r * C + h * c + i] * dMb[(int64_t)r * C + h * c + j];
    sm[i * ld + j] = acc;
  }
  __syncthreads();
  // softmax backward per row: dS = A * (dA - sum_j dA*A);  rq_i = sum_j dS*S
  float tsum = 0.f;
  for (int i = wv; i < c; i += 4) {
    float dot = 0.f;
    for (int j = lane; j < c; j += 64) dot += sm[i * ld + j] * Az[i * c + j];
    dot = wave_sum(dot);
    float rqa = 0.f;
    for (int j = lane; j < c; j += 64) {
      const float ds = Az[i * c + j] * (sm[i * ld + j] - dot);
      sm[i * ld + j] = ds;
      const float pv = Pz[i * c + j];
      tsum += ds * pv;
      rqa += ds * pv * temp;
    }
    rqa = wave_sum(rqa);
    if (lane == 0) rq[i] = rqa;
  }
  tsum = wave_sum(tsum);
  if (lane == 0) red[wv] = tsum;
  __syncthreads();
  if (t == 0) dtemp_part[z] = (red[0] + red[1]) + (red[2] + red[3]);
  for (int j = t; j < c; j += 256) {
    float acc = 0.f;
    for (int i = 0; i < c; ++i) acc += sm[i * ld + j] * Pz[i * c + j] * temp;
    rk[j] = acc;
  }
  __syncthreads();
  float* wq = wdq + (int64_t)z * c * 2 * c;
  float* wk = wdk + (int64_t)z * c * 2 * c;
  for (int e = t; e < c * c; e += 256) {
    const int i = e / c, j = e - i * c;
    const float nq = nz[i], nk = nz[c + j];
    const float g1 = temp * sm[i * ld + j] / (nq * nk);
    wq[i * 2 * c + j] = g1;            // dq_i += g1 * k_j
    wk[j * 2 * c + i] = g1;            // dk_j += g1 * q_i
    // diagonal blocks: projection terms of d(x/|x|); zero when the norm was clamped
    const float d1 = (i == j) ? ((nq > NORM_EPS) ? -rq[i] / (nq * nq) : 0.f) : 0.f;
    const float nkj = nz[c + i];       // for the (i,i') slot of wk we need row index = i here
    const float d2 = (i == j) ? ((nkj > NORM_EPS) ? -rk[i] / (nkj * nkj) : 0.f) : 0.f;
    wq[i * 2 * c + c + j] = d1;
    wk[i * 2 * c + c + j] = d2;
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
  const size_t lds = (size_t)c * ld * sizeof(float);
  ProfScope ps(st, K_ATTN_BWD_SMALL, 4.0 * B * (6.0 * C * c + 3.0 * C * C), 4.0 * B * C * (double)c * C);
  hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(B * heads), dim3(256), lds, st, dM, A, P, nrm, temperature, wo, dwo_part,
                     dtemp_part, wdq, wdk, C, heads, ld);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

}  // namespace mi
