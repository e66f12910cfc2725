// MoCE-IR specific kernels (moce_ir.py of the reference):
//   * mi_moe_route_fwd / _bwd   RoutingFunction (:684-800) as ONE launch each way: logits of the two linear gates, injected
//                               noise, softmax, top-k, gate scatter, both CV^2 auxiliary losses, and the SparseDispatcher's
//                               index bookkeeping (:82-91) as device-side tables (counts, offsets, sample permutation, gates
//                               in dispatch order) - no nonzero / sort / tolist on the way.
//   * mi_patch_circconv         FFTAttention's  irfft2(rfft2(q) * rfft2(k))  (:408-414) per p x p patch, computed as what it
//                               is - a 2-D circular convolution - straight from / to NCHW planes (zero padded to the patch
//                               grid, cropped back), fp32 arithmetic like the reference's upcast.  flip = 1 convolves with the
//                               index-reversed second operand, which makes the two input gradients the same kernel.
//   * mi_gelu_gap_fwd / _bwd    FrequencyEmbedding's  GELU -> mean over the plane  (:1062-1064,1071-1073).
//   * mi_ewise_fwd / _bwd       the two gating products of the expert path: a * b (:419) and a * silu(b) (:555).
#include <math.h>

#include "internal.h"

namespace mi {

// ------------------------------------------------------------------------------------------------ router
constexpr int ROUTE_MAX_E = 8;
constexpr int ROUTE_MAX_BE = 8192;

struct RouteArgs {
  const float* pooled; const float* freq; const float* wg; const float* wf; const float* noise; const float* complexity;
  float* logits; float* gates; int64_t* topk_idx; float* topk_val; float* aux;
  int* counts; int* offsets; int64_t* perm; float* perm_gate; int* perm_expert; int* row_of;
  int B, C, F, E, k, training;
};

__device__ __forceinline__ float cv_squared(const float* v, int E) {   // (std_unbiased / (mean + 1e-8))^2   (:761-763,797-799)
  float m = 0.f;
  for (int e = 0; e < E; ++e) m += v[e];
  m /= (float)E;
  float var = 0.f;
  for (int e = 0; e < E; ++e) var += (v[e] - m) * (v[e] - m);
  var /= (float)(E - 1);
  return var / ((m + 1e-8f) * (m + 1e-8f));
}

__global__ __launch_bounds__(256) void route_fwd_kernel(RouteArgs a) {
  extern __shared__ float rsm[];
  float* L = rsm;                      // clean logits [B][E]
  float* Nz = rsm + a.B * a.E;         // noisy logits [B][E]
  float* Pc = rsm + 2 * a.B * a.E;     // softmax(clean) [B][E], later 1 - Phi(z)
  __shared__ int s_cnt[ROUTE_MAX_E + 1];
  __shared__ float s_vec[2][ROUTE_MAX_E];
  const int t = threadIdx.x, B = a.B, E = a.E, k = a.k;
  const float sigma = 1.0f / (float)E;                                       // noise_std (:733)
  for (int i = t; i < B * E; i += 256) {
    const int b = i / E, e = i - b * E;
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s += a.pooled[(int64_t)b * a.C + c] * a.wg[(int64_t)e * a.C + c];
    for (int f = 0; f < a.F; ++f) s += a.freq[(int64_t)b * a.F + f] * a.wf[(int64_t)e * a.F + f];
    L[i] = s;
    Nz[i] = s + a.noise[i] * sigma;
    a.logits[i] = s;
  }
  __syncthreads();
  for (int b = t; b < B; b += 256) {
    float n[ROUTE_MAX_E], l[ROUTE_MAX_E];
    float mx = -INFINITY, mc = -INFINITY;
    for (int e = 0; e < E; ++e) { n[e] = Nz[b * E + e]; l[e] = L[b * E + e]; mx = fmaxf(mx, n[e]); mc = fmaxf(mc, l[e]); }
    float den = 0.f, dc = 0.f;
    for (int e = 0; e < E; ++e) { n[e] = expf(n[e] - mx); den += n[e]; l[e] = expf(l[e] - mc); dc += l[e]; }
    for (int e = 0; e < E; ++e) { n[e] /= den; Pc[b * E + e] = l[e] / dc; a.gates[b * E + e] = 0.f; }
    unsigned taken = 0;
    for (int j = 0; j < k; ++j) {                                            // top-k of the gating scores (:743-744)
      int best = -1;
      for (int e = 0; e < E; ++e)
        if (!(taken >> e & 1u) && (best < 0 || n[e] > n[best])) best = e;
      taken |= 1u << best;
      a.topk_idx[(int64_t)b * k + j] = best;
      a.topk_val[(int64_t)b * k + j] = n[best];
      a.gates[b * E + best] = n[best];                                       // zeros.scatter_(1, idx, vals) (:753)
    }
  }
  __syncthreads();
  // dispatch tables: expert e's rows are the samples that picked it, in increasing sample order
  if (t < E) {
    int c = 0;
    for (int b = 0; b < B; ++b)
      for (int j = 0; j < k; ++j) c += a.topk_idx[(int64_t)b * k + j] == t ? 1 : 0;
    s_cnt[t] = c;
    a.counts[t] = c;
  }
  __syncthreads();
  if (t == 0) {
    int o = 0;
    for (int e = 0; e < E; ++e) { a.offsets[e] = o; o += s_cnt[e]; }
    a.offsets[E] = o;
  }
  __syncthreads();
  if (t < E) {
    int pos = a.offsets[t];
    for (int b = 0; b < B; ++b)
      for (int j = 0; j < k; ++j)
        if (a.topk_idx[(int64_t)b * k + j] == t) {
          a.perm[pos] = b; a.perm_gate[pos] = a.topk_val[(int64_t)b * k + j]; a.perm_expert[pos] = t;
          a.row_of[b * k + j] = pos;
          ++pos;
        }
  }
  // auxiliary loss (training only): 0.5 * CV^2(importance) + 0.5 * CV^2(load)   (:738-749,759-800)
  if (!a.training) { if (t == 0) a.aux[0] = 0.f; return; }
  for (int b = t; b < B; b += 256) {
    const int kth = (int)a.topk_idx[(int64_t)b * k + (k - 1)];
    const float thr = Nz[b * E + kth];
    for (int e = 0; e < E; ++e) {
      const float z = (thr - L[b * E + e]) / sigma;
      Nz[b * E + e] = 0.5f * erfcf(z * 0.70710678118654752440f);            // 1 - Phi(z)   (Nz is dead from here on)
    }
  }
  __syncthreads();
  if (t < E) {
    float imp = 0.f, ld = 0.f;
    for (int b = 0; b < B; ++b) { imp += Pc[b * E + t]; ld += Nz[b * E + t]; }
    s_vec[0][t] = a.complexity ? imp * a.complexity[t] : imp;               // importance * complexity * tau, tau = 1
    s_vec[1][t] = ld / (float)B;
  }
  __syncthreads();
  if (t == 0) a.aux[0] = 0.5f * cv_squared(s_vec[0], E) + 0.5f * cv_squared(s_vec[1], E);
}

struct RouteBwdArgs {
  const float* pooled; const float* freq; const float* wg; const float* wf; const float* noise; const float* complexity;
  const float* logits; const int64_t* topk_idx; const float* dgates; const float* drow; const int* row_of; const float* daux;
  float* dpooled; float* dfreq; float* dwg; float* dwf;
  int B, C, F, E, k, training;
};

// d cv^2 / d v[e]
__device__ __forceinline__ void cv_squared_grad(const float* v, int E, float* g) {
  float m = 0.f;
  for (int e = 0; e < E; ++e) m += v[e];
  m /= (float)E;
  float var = 0.f;
  for (int e = 0; e < E; ++e) var += (v[e] - m) * (v[e] - m);
  var /= (float)(E - 1);
  const float me = m + 1e-8f;
  for (int e = 0; e < E; ++e) g[e] = 2.f * (v[e] - m) / ((float)(E - 1) * me * me) - 2.f * var / (me * me * me * (float)E);
}

__global__ __launch_bounds__(256) void route_bwd_kernel(RouteBwdArgs a) {
  extern __shared__ float rsm[];
  float* DL = rsm;                     // d loss / d clean logits [B][E]
  float* Pc = rsm + a.B * a.E;         // softmax(clean)
  float* Pl = rsm + 2 * a.B * a.E;     // 1 - Phi(z)
  __shared__ float s_vec[2][ROUTE_MAX_E], s_g[2][ROUTE_MAX_E];
  const int t = threadIdx.x, B = a.B, E = a.E, k = a.k;
  const float sigma = 1.0f / (float)E;
  const float daux = (a.training && a.daux) ? a.daux[0] : 0.f;
  for (int b = t; b < B; b += 256) {
    // per-sample work vectors in LDS, not in registers: they are indexed by the top-k indices, and as local arrays they lived
    // in scratch memory - the one scratch user of the MoCE step, and what took HIP-graph capture of the step down at instantiation
    float* const n = rsm + (3 * B + b) * E;
    float* const l = rsm + (4 * B + b) * E;
    float* const ds = rsm + (5 * B + b) * E;
    float mx = -INFINITY, mc = -INFINITY;
    for (int e = 0; e < E; ++e) {
      l[e] = a.logits[b * E + e]; n[e] = l[e] + a.noise[b * E + e] * sigma;
      mx = fmaxf(mx, n[e]); mc = fmaxf(mc, l[e]); ds[e] = 0.f;
    }
    const int kth = (int)a.topk_idx[(int64_t)b * k + (k - 1)];
    const float thr = n[kth];
    float den = 0.f, dc = 0.f;
    for (int e = 0; e < E; ++e) {
      Pl[b * E + e] = 0.5f * erfcf((thr - l[e]) / sigma * 0.70710678118654752440f);
      n[e] = expf(n[e] - mx); den += n[e]; l[e] = expf(l[e] - mc); dc += l[e];
    }
    for (int j = 0; j < k; ++j) {
      const int e = (int)a.topk_idx[(int64_t)b * k + j];
      ds[e] = (a.dgates ? a.dgates[b * E + e] : 0.f) + (a.drow ? a.drow[a.row_of[b * k + j]] : 0.f);
    }
    float dot = 0.f;
    for (int e = 0; e < E; ++e) { n[e] /= den; dot += ds[e] * n[e]; Pc[b * E + e] = l[e] / dc; }
    for (int e = 0; e < E; ++e) DL[b * E + e] = n[e] * (ds[e] - dot);        // through softmax(noisy) of the selected gates
  }
  __syncthreads();
  if (daux != 0.f) {
    if (t < E) {
      float imp = 0.f, ld = 0.f;
      for (int b = 0; b < B; ++b) { imp += Pc[b * E + t]; ld += Pl[b * E + t]; }
      s_vec[0][t] = a.complexity ? imp * a.complexity[t] : imp;
      s_vec[1][t] = ld / (float)B;
    }
    __syncthreads();
    if (t == 0) { cv_squared_grad(s_vec[0], E, s_g[0]); cv_squared_grad(s_vec[1], E, s_g[1]); }
    __syncthreads();
    for (int b = t; b < B; b += 256) {
      const int kth = (int)a.topk_idx[(int64_t)b * k + (k - 1)];
      float* const dcv = rsm + (3 * B + b) * E;          // (the n vector is dead by now)
      float dot = 0.f, dthr = 0.f;
      for (int e = 0; e < E; ++e) {
        dcv[e] = s_g[0][e] * (a.complexity ? a.complexity[e] : 1.f);          // d L_imp / d softmax(clean)[b][e]
        dot += dcv[e] * Pc[b * E + e];
      }
      const float thr = a.logits[b * E + kth] + a.noise[b * E + kth] * sigma;
      for (int e = 0; e < E; ++e) {
        float g = 0.5f * daux * Pc[b * E + e] * (dcv[e] - dot);
        const float z = (thr - a.logits[b * E + e]) / sigma;
        const float dz = -0.39894228040143267794f * expf(-0.5f * z * z) * (s_g[1][e] / (float)B);   // d L_load / d z
        g += 0.5f * daux * dz * (-1.0f / sigma);
        dthr += dz / sigma;
        DL[b * E + e] += g;
      }
      DL[b * E + kth] += 0.5f * daux * dthr;
    }
    __syncthreads();
  }
  for (int i = t; i < B * a.C; i += 256) {
    const int b = i / a.C, c = i - b * a.C;
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += DL[b * E + e] * a.wg[(int64_t)e * a.C + c];
    a.dpooled[i] = s;
  }
  for (int i = t; i < B * a.F; i += 256) {
    const int b = i / a.F, f = i - b * a.F;
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += DL[b * E + e] * a.wf[(int64_t)e * a.F + f];
    a.dfreq[i] = s;
  }
  for (int i = t; i < E * a.C; i += 256) {
    const int e = i / a.C, c = i - e * a.C;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += DL[b * E + e] * a.pooled[(int64_t)b * a.C + c];
    a.dwg[i] = s;
  }
  for (int i = t; i < E * a.F; i += 256) {
    const int e = i / a.F, f = i - e * a.F;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += DL[b * E + e] * a.freq[(int64_t)b * a.F + f];
    a.dwf[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ patch circular convolution
// out[u][v] = sum_{a,b} x[a][b] * y[(u-a) mod P][(v-b) mod P]  per P x P patch of every (image, channel) plane.
// One workgroup = one band of P rows x 256 columns: of one plane, or - when the plane is narrower (W < 256, 256 % W == 0,
// W % P == 0: every training plane of MoCE-IR) - of 256 / W consecutive planes side by side, so that no thread idles on a 128-
// or 64-pixel row (the first form kept half / three quarters of the workgroup idle there: 60 us per launch on average at
// bs 8 x 128^2).  Both operand bands are staged in LDS as fp32 (rows 16-byte aligned: the P-float row pieces are read as
// vectors - with an odd stride the scalar reads of threads 8 floats apart were 8-way bank conflicts); a thread owns one output
// row of one patch (P accumulators); for every row a of x it holds the matching rows of x and y in registers and walks b with
// statically rotated indices.
template <typename T, int P>
__global__ __launch_bounds__(256) void circconv_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out,
                                                       int C, int H, int W, int flip, int64_t x_bs, int64_t y_bs, int64_t o_bs,
                                                       int64_t planes, int ppw) {
  constexpr int PPB = 256 / P;                // patches per workgroup
  constexpr int LS = 256 + 4;                 // LDS row stride (floats)
  extern __shared__ __attribute__((aligned(16))) float csm[];
  float* xs = csm;                            // [P][LS]
  float* ys = csm + P * LS;
  const int t = threadIdx.x;
  const int r0 = blockIdx.y * P;
  const int c0 = ppw > 1 ? 0 : blockIdx.x * 256;
  const int64_t plane0 = (int64_t)blockIdx.z * ppw;
  auto plane_of = [&](int c, int* X) -> int64_t {           // virtual column -> (plane, column)
    if (ppw > 1) { const int q = c / W; *X = c - q * W; return plane0 + q; }
    *X = c0 + c;
    return plane0;
  };
  {                                           // thread t stages virtual column t of every row of the band
    int X;
    const int64_t plane = plane_of(t, &X);
    const int64_t pb = plane / C, pc_ = plane - pb * C;
    const bool cin = X < W && plane < planes;
    const T* xq = x + pb * x_bs + pc_ * (int64_t)H * W + X;
    const T* yq = y + pb * y_bs + pc_ * (int64_t)H * W + X;
    // flip: y'[i][j] = y[(-i) mod P][(-j) mod P] within each patch (correlation form used by the input gradients)
    const int cc = flip ? (t / P) * P + (P - t % P) % P : t;
    float xv[P], yv[P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
      const bool in = cin && r0 + r < H;
      xv[r] = in ? to_f32(xq[(int64_t)(r0 + r) * W]) : 0.f;
      yv[r] = in ? to_f32(yq[(int64_t)(r0 + r) * W]) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < P; ++r) {
      xs[r * LS + t] = xv[r];
      ys[(flip ? (P - r) % P : r) * LS + cc] = yv[r];
    }
  }
  __syncthreads();
  const int u = t / PPB, pw = t % PPB;        // consecutive threads -> consecutive patches of one output row: coalesced stores
  const int pc = pw * P;
  float acc[P];
#pragma unroll
  for (int v = 0; v < P; ++v) acc[v] = 0.f;
  for (int a = 0; a < P; ++a) {
    const int yr = (u - a + P) % P;
    float kr[P], qr[P];
#pragma unroll
    for (int v = 0; v < P; v += 4) {
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(&ys[yr * LS + pc + v]);
      const f32x4 q4 = *reinterpret_cast<const f32x4*>(&xs[a * LS + pc + v]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kr[v + e] = k4[e]; qr[v + e] = q4[e]; }
    }
#pragma unroll
    for (int b = 0; b < P; ++b) {
#pragma unroll
      for (int v = 0; v < P; ++v) acc[v] += qr[b] * kr[(v - b + P) % P];
    }
  }
  const int Y = r0 + u;
  int X0;
  const int64_t plane = plane_of(pc, &X0);
  if (Y < H && plane < planes) {
    const int64_t pb = plane / C, pc_ = plane - pb * C;
    T* op = out + pb * o_bs + pc_ * (int64_t)H * W + (int64_t)Y * W;
#pragma unroll
    for (int v = 0; v < P; ++v)
      if (X0 + v < W) op[X0 + v] = Cvt<T>::from(acc[v]);
  }
}

template <typename T>
static int circconv_launch(const void* x, const void* y, void* out, int B, int C, int H, int W, int p, int flip, int64_t x_bs,
                           int64_t y_bs, int64_t o_bs, hipStream_t st) {
  const int64_t planes = (int64_t)B * C;
  const int ppw = (W < 256 && 256 % W == 0 && W % p == 0) ? 256 / W : 1;    // planes side by side in one workgroup
  dim3 grid(ppw > 1 ? 1u : (unsigned)cdiv(W, 256), (unsigned)cdiv(H, p), (unsigned)cdiv(planes, ppw));
  MI_CHECK_ARG(planes < 65536 * 32 && grid.y < 65536 && grid.z < 65536 * 32, "patch_circconv: grid too large");
  const size_t lds = 2 * (size_t)p * 260 * sizeof(float);
#define CC(PP)                                                                                                             \
  case PP:                                                                                                                 \
    if (lds > 64 * 1024) MI_CHECK_HIP(hipFuncSetAttribute((const void*)circconv_kernel<T, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((circconv_kernel<T, PP>), grid, dim3(256), lds, st, (const T*)x, (const T*)y, (T*)out, C, H, W, flip, x_bs, y_bs, o_bs, \
                       planes, ppw);                                                                                       \
    break
  switch (p) { CC(4); CC(8); CC(16); CC(32); default: set_error("patch_circconv: patch size %d not in {4,8,16,32}", p); return MI_ERR_ARG; }
#undef CC
  MI_LAUNCH_CHECK();
  return MI_OK;
}

// ------------------------------------------------------------------------------------------------ GELU + mean over the plane
template <typename T>
__global__ __launch_bounds__(256) void gelu_gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t N) {
  __shared__ float sm[4];
  const T* row = x + (int64_t)blockIdx.x * N;
  float acc = 0.f;
  for (int64_t n = threadIdx.x; n < N; n += 256) acc += gelu_erf(to_f32(row[n]));
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)N;
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_gap_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dout,
                                                           T* __restrict__ dx, int64_t N) {
  const float g = dout[blockIdx.x] / (float)N;
  const T* row = x + (int64_t)blockIdx.x * N;
  T* drow = dx + (int64_t)blockIdx.x * N;
  for (int64_t n = threadIdx.x; n < N; n += 256) drow[n] = Cvt<T>::from(g * gelu_erf_grad(to_f32(row[n])));
}

// ------------------------------------------------------------------------------------------------ gating products
// op 0: out = a * b ; op 1: out = a * silu(b) ; op 2: out = gelu_erf(a) (b unused; FrequencyEmbedding's MLP activation, :1071)
template <typename T>
__global__ __launch_bounds__(256) void ewise_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                        int64_t n, int op, int64_t L, int64_t a_rs, int64_t b_rs) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / L, c = i - r * L;
    const float av = to_f32(a[r * a_rs + c]), bv = to_f32(b[r * b_rs + c]);
    if (op == 2) { out[i] = Cvt<T>::from(gelu_erf(av)); continue; }
    const float f = op == 0 ? bv : bv / (1.f + __expf(-bv));
    out[i] = Cvt<T>::from(av * f);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void ewise_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        const T* __restrict__ dout, T* __restrict__ da, T* __restrict__ db,
                                                        int64_t n, int op, int64_t L, int64_t a_rs, int64_t b_rs, int64_t da_rs,
                                                        int64_t db_rs) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int64_t r = i / L, c = i - r * L;
    const float av = to_f32(a[r * a_rs + c]), bv = to_f32(b[r * b_rs + c]), g = to_f32(dout[i]);
    float f, df;
    if (op == 2) { da[r * da_rs + c] = Cvt<T>::from(g * gelu_erf_grad(av)); continue; }
    if (op == 0) { f = bv; df = 1.f; }
    else { const float s = 1.f / (1.f + __expf(-bv)); f = bv * s; df = s * (1.f + bv * (1.f - s)); }
    da[r * da_rs + c] = Cvt<T>::from(g * f);
    db[r * db_rs + c] = Cvt<T>::from(g * av * df);
  }
}

}  // namespace mi

using namespace mi;

extern "C" int mi_moe_route_fwd(const float* pooled, const float* freq, const float* wg, const float* wf, const float* noise,
                                const float* complexity, float* logits, float* gates, int64_t* topk_idx, float* topk_val,
                                float* aux, int* counts, int* offsets, int64_t* perm, float* perm_gate, int* perm_expert,
                                int* row_of, int B, int C, int F, int E, int k, int training, void* stream) {
  MI_CHECK_ARG(pooled && freq && wg && wf && noise && logits && gates && topk_idx && topk_val && aux && counts && offsets &&
               perm && perm_gate && perm_expert && row_of, "moe_route_fwd: null pointer");
  MI_CHECK_ARG(B > 0 && C > 0 && F > 0 && E >= 2 && E <= ROUTE_MAX_E && k >= 1 && k <= E && B * E <= ROUTE_MAX_BE,
               "moe_route_fwd: bad shape (B=%d C=%d F=%d E=%d k=%d)", B, C, F, E, k);
  // the backward keeps six [B][E] vectors in LDS (the forward three): a training forward refuses what its backward could not run
  MI_CHECK_ARG(!training || B * E <= ROUTE_MAX_BE * 3 / 4,
               "moe_route_fwd: training batch too large for the router backward (B * E <= %d)", ROUTE_MAX_BE * 3 / 4);
  RouteArgs a{pooled, freq, wg, wf, noise, complexity, logits, gates, topk_idx, topk_val, aux, counts, offsets, perm, perm_gate,
              perm_expert, row_of, B, C, F, E, k, training};
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 3 * (size_t)B * E * sizeof(float);
  if (lds > 48 * 1024)
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)route_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps(st, K_MOE_ROUTE, 4.0 * ((double)B * (C + F) + (double)E * (C + F)), 2.0 * B * E * (C + F));
  hipLaunchKernelGGL(route_fwd_kernel, dim3(1), dim3(256), lds, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_moe_route_bwd(const float* pooled, const float* freq, const float* wg, const float* wf, const float* noise,
                                const float* complexity, const float* logits, const int64_t* topk_idx, const float* dgates,
                                const float* drow, const int* row_of, const float* daux, float* dpooled, float* dfreq, float* dwg,
                                float* dwf, int B, int C, int F, int E, int k, int training, void* stream) {
  MI_CHECK_ARG(!drow || row_of, "moe_route_bwd: drow needs row_of");
  MI_CHECK_ARG(pooled && freq && wg && wf && noise && logits && topk_idx && dpooled && dfreq && dwg && dwf,
               "moe_route_bwd: null pointer");
  MI_CHECK_ARG(B > 0 && C > 0 && F > 0 && E >= 2 && E <= ROUTE_MAX_E && k >= 1 && k <= E && B * E <= ROUTE_MAX_BE * 3 / 4,
               "moe_route_bwd: bad shape (six [B][E] fp32 vectors must fit the 160 KB of LDS: B * E <= %d)", ROUTE_MAX_BE * 3 / 4);
  RouteBwdArgs a{pooled, freq, wg, wf, noise, complexity, logits, topk_idx, dgates, drow, row_of, daux, dpooled, dfreq, dwg, dwf,
                 B, C, F, E, k, training};
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 6 * (size_t)B * E * sizeof(float);   // DL, Pc, Pl + the three per-sample work vectors
  if (lds > 48 * 1024)
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)route_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ProfScope ps(st, K_MOE_ROUTE, 4.0 * ((double)B * (C + F) * 2 + (double)E * (C + F) * 2), 4.0 * B * E * (C + F));
  hipLaunchKernelGGL(route_bwd_kernel, dim3(1), dim3(256), lds, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_patch_circconv(const void* x, int64_t x_bs, const void* y, int64_t y_bs, void* out, int64_t out_bs, int B, int C,
                                 int H, int W, int patch, int flip, int dtype, void* stream) {
  MI_CHECK_ARG(x && y && out && B > 0 && C > 0 && H > 0 && W > 0, "patch_circconv: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "patch_circconv: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const double px = (double)B * C * H * W;
  ProfScope ps(st, K_CIRCCONV, 3.0 * px * dtype_size(dtype), 2.0 * px * patch * patch);
  if (!x_bs) x_bs = (int64_t)C * H * W;
  if (!y_bs) y_bs = (int64_t)C * H * W;
  if (!out_bs) out_bs = (int64_t)C * H * W;
  if (dtype == MI_F32) return circconv_launch<float>(x, y, out, B, C, H, W, patch, flip, x_bs, y_bs, out_bs, st);
  return circconv_launch<bf16>(x, y, out, B, C, H, W, patch, flip, x_bs, y_bs, out_bs, st);
}

extern "C" int mi_gelu_gap_fwd(const void* x, float* out, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(x && out && B > 0 && C > 0 && N > 0, "gelu_gap_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_GAP, (double)B * C * N * dtype_size(dtype), 10.0 * B * C * N);
  if (dtype == MI_F32) hipLaunchKernelGGL((gelu_gap_fwd_kernel<float>), dim3(B * C), dim3(256), 0, st, (const float*)x, out, N);
  else if (dtype == MI_BF16) hipLaunchKernelGGL((gelu_gap_fwd_kernel<bf16>), dim3(B * C), dim3(256), 0, st, (const bf16*)x, out, N);
  else { set_error("gelu_gap_fwd: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_gelu_gap_bwd(const void* x, const float* dout, void* dx, int B, int C, int64_t N, int dtype, void* stream) {
  MI_CHECK_ARG(x && dout && dx && B > 0 && C > 0 && N > 0, "gelu_gap_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(st, K_GAP, 2.0 * B * C * N * dtype_size(dtype), 10.0 * B * C * N);
  if (dtype == MI_F32) hipLaunchKernelGGL((gelu_gap_bwd_kernel<float>), dim3(B * C), dim3(256), 0, st, (const float*)x, dout, (float*)dx, N);
  else if (dtype == MI_BF16) hipLaunchKernelGGL((gelu_gap_bwd_kernel<bf16>), dim3(B * C), dim3(256), 0, st, (const bf16*)x, dout, (bf16*)dx, N);
  else { set_error("gelu_gap_bwd: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_ewise_fwd(const void* a, int64_t a_rs, const void* b, int64_t b_rs, void* out, int64_t rows, int64_t L, int op,
                            int dtype, void* stream) {
  const int64_t n = rows * L;
  MI_CHECK_ARG(a && b && out && rows > 0 && L > 0 && op >= 0 && op <= 2, "ewise_fwd: bad arguments");
  if (!a_rs) a_rs = L;
  if (!b_rs) b_rs = L;
  hipStream_t st = (hipStream_t)stream;
  int blocks = cdiv(n, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  ProfScope ps(st, K_EWISE, 3.0 * n * dtype_size(dtype), 4.0 * n);
  if (dtype == MI_F32) hipLaunchKernelGGL((ewise_fwd_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n, op, L, a_rs, b_rs);
  else if (dtype == MI_BF16) hipLaunchKernelGGL((ewise_fwd_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, n, op, L, a_rs, b_rs);
  else { set_error("ewise_fwd: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}
extern "C" int mi_ewise_bwd(const void* a, int64_t a_rs, const void* b, int64_t b_rs, const void* dout, void* da, int64_t da_rs,
                            void* db, int64_t db_rs, int64_t rows, int64_t L, int op, int dtype, void* stream) {
  const int64_t n = rows * L;
  MI_CHECK_ARG(a && b && dout && da && db && rows > 0 && L > 0 && op >= 0 && op <= 2, "ewise_bwd: bad arguments");
  if (!a_rs) a_rs = L;
  if (!b_rs) b_rs = L;
  if (!da_rs) da_rs = L;
  if (!db_rs) db_rs = L;
  hipStream_t st = (hipStream_t)stream;
  int blocks = cdiv(n, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  ProfScope ps(st, K_EWISE, 5.0 * n * dtype_size(dtype), 8.0 * n);
  if (dtype == MI_F32) hipLaunchKernelGGL((ewise_bwd_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)dout, (float*)da, (float*)db, n, op, L, a_rs, b_rs, da_rs, db_rs);
  else if (dtype == MI_BF16) hipLaunchKernelGGL((ewise_bwd_kernel<bf16>), dim3(blocks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (const bf16*)dout, (bf16*)da, (bf16*)db, n, op, L, a_rs, b_rs, da_rs, db_rs);
  else { set_error("ewise_bwd: bad dtype"); return MI_ERR_ARG; }
  MI_LAUNCH_CHECK();
  return MI_OK;
}
