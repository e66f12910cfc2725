// MoCE SparseDispatcher data movement (moce_ir.py:71-143): dispatch gathers whole [C,H,W] feature maps of the samples
// routed to each expert, combine scatters the gate-weighted expert outputs back with fp32 accumulation.  A "row" here is
// one sample's feature map (C*H*W elements).  The routing lists are tiny (<= batch * top_k entries), so the scatter is
// done per DESTINATION row - each workgroup scans the list for the sources that map to its row and adds them in list
// order: deterministic, no atomics.
#include "common.h"

namespace mi {
namespace {

constexpr int DV = 8;  // elements per thread step

// out[i] = x[idx[i]]
template <typename T>
__global__ __launch_bounds__(256) void rows_gather_kernel(const T* __restrict__ x, const int64_t* __restrict__ idx,
                                                          T* __restrict__ out, int64_t row, int vec_ok) {
  const int i = blockIdx.y;
  const T* src = x + idx[i] * row;
  T* dst = out + (int64_t)i * row;
  constexpr int V = 16 / (int)sizeof(T);
  if (vec_ok) {
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; e < row; e += (int64_t)gridDim.x * 256 * V)
      *reinterpret_cast<u32x4*>(dst + e) = *reinterpret_cast<const u32x4*>(src + e);
  } else {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < row; e += (int64_t)gridDim.x * 256) dst[e] = src[e];
  }
}

// out[i] = scale[i] * x[idx[i]] with fp32 x and activation-dtype out (backward of combine w.r.t. the expert outputs)
template <typename T>
__global__ __launch_bounds__(256) void rows_gather_scaled_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                                 const float* __restrict__ scale, T* __restrict__ out,
                                                                 int64_t row) {
  const int i = blockIdx.y;
  const float* src = x + idx[i] * row;
  const float sc = scale ? scale[i] : 1.0f;
  T* dst = out + (int64_t)i * row;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < row; e += (int64_t)gridDim.x * 256)
    dst[e] = Cvt<T>::from(sc * src[e]);
}

// out[b] = sum over i with idx[i] == b of scale[i] * src[i]   (fp32 accumulate; OUT = float or T)
template <typename T, typename OUT>
__global__ __launch_bounds__(256) void rows_scatter_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx,
                                                           const float* __restrict__ scale, OUT* __restrict__ out, int n_src,
                                                           int64_t row) {
  const int b = blockIdx.y;
  OUT* dst = out + (int64_t)b * row;
  for (int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e0 < row; e0 += (int64_t)gridDim.x * 256 * 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nv = row - e0 >= 4 ? 4 : (int)(row - e0);
    for (int i = 0; i < n_src; ++i) {
      if (idx[i] != b) continue;                      // uniform over the workgroup
      const float s = scale ? scale[i] : 1.0f;
      const T* p = src + (int64_t)i * row + e0;
      for (int j = 0; j < nv; ++j) acc[j] += s * to_f32(p[j]);
    }
    for (int j = 0; j < nv; ++j) {
      if constexpr (sizeof(OUT) == 4) dst[e0 + j] = acc[j];
      else dst[e0 + j] = Cvt<T>::from(acc[j]);
    }
  }
}

// part[i][blk] = sum over this workgroup's elements of g[idx[i]][e] * src[i][e]   (gradient of the gate values)
template <typename T>
__global__ __launch_bounds__(256) void rows_dot_kernel(const float* __restrict__ g, const T* __restrict__ src,
                                                       const int64_t* __restrict__ idx, float* __restrict__ part, int64_t row) {
  __shared__ float red[4];
  const int i = blockIdx.y;
  const float* gp = g + idx[i] * row;
  const T* sp = src + (int64_t)i * row;
  float acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < row; e += (int64_t)gridDim.x * 256) acc += gp[e] * to_f32(sp[e]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(int64_t)blockIdx.x * gridDim.y + i] = (red[0] + red[1]) + (red[2] + red[3]);
}

static int row_blocks(int64_t row) {
  int64_t b = cdiv(row, 256 * 8);
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace
}  // namespace mi

using namespace mi;

extern "C" int mi_rows_gather(const void* x, const int64_t* idx, void* out, int n_out, int64_t row, int dtype, void* stream) {
  MI_CHECK_ARG(x && idx && out && n_out >= 0 && row > 0, "rows_gather: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "rows_gather: bad dtype %d", dtype);
  if (n_out == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t vec = 16 / (int64_t)dtype_size(dtype);
  const int ok = (row % vec == 0) && aligned16(x) && aligned16(out);
  dim3 grid(row_blocks(row), n_out);
  if (dtype == MI_F32) hipLaunchKernelGGL((rows_gather_kernel<float>), grid, dim3(256), 0, st, (const float*)x, idx, (float*)out, row, ok);
  else hipLaunchKernelGGL((rows_gather_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, idx, (bf16*)out, row, ok);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_rows_gather_scaled(const float* x, const int64_t* idx, const float* scale, void* out, int n_out, int64_t row,
                                     int out_dtype, void* stream) {
  MI_CHECK_ARG(x && idx && out && n_out >= 0 && row > 0, "rows_gather_scaled: bad arguments");
  MI_CHECK_ARG(out_dtype == MI_F32 || out_dtype == MI_BF16, "rows_gather_scaled: bad dtype %d", out_dtype);
  if (n_out == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(row_blocks(row), n_out);
  if (out_dtype == MI_F32) hipLaunchKernelGGL((rows_gather_scaled_kernel<float>), grid, dim3(256), 0, st, x, idx, scale, (float*)out, row);
  else hipLaunchKernelGGL((rows_gather_scaled_kernel<bf16>), grid, dim3(256), 0, st, x, idx, scale, (bf16*)out, row);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" int mi_rows_scatter_add(const void* src, const int64_t* idx, const float* scale, void* out, int n_src, int n_rows,
                                   int64_t row, int dtype, int out_f32, void* stream) {
  MI_CHECK_ARG(src && idx && out && n_src >= 0 && n_rows > 0 && row > 0, "rows_scatter_add: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "rows_scatter_add: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(row_blocks(row), n_rows);
  if (dtype == MI_F32)
    hipLaunchKernelGGL((rows_scatter_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, idx, scale, (float*)out, n_src, row);
  else if (out_f32)
    hipLaunchKernelGGL((rows_scatter_kernel<bf16, float>), grid, dim3(256), 0, st, (const bf16*)src, idx, scale, (float*)out, n_src, row);
  else
    hipLaunchKernelGGL((rows_scatter_kernel<bf16, bf16>), grid, dim3(256), 0, st, (const bf16*)src, idx, scale, (bf16*)out, n_src, row);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" size_t mi_rows_dot_workspace(int n_src, int64_t row) {
  return align_up((size_t)row_blocks(row) * (size_t)(n_src > 0 ? n_src : 1) * sizeof(float), 256);
}

extern "C" int mi_rows_dot(const float* g, const void* src, const int64_t* idx, float* out, int n_src, int64_t row, int dtype,
                           void* ws, void* stream) {
  MI_CHECK_ARG(g && src && idx && out && ws && n_src >= 0 && row > 0, "rows_dot: bad arguments");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "rows_dot: bad dtype %d", dtype);
  if (n_src == 0) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nb = row_blocks(row);
  dim3 grid(nb, n_src);
  float* part = (float*)ws;
  if (dtype == MI_F32) hipLaunchKernelGGL((rows_dot_kernel<float>), grid, dim3(256), 0, st, g, (const float*)src, idx, part, row);
  else hipLaunchKernelGGL((rows_dot_kernel<bf16>), grid, dim3(256), 0, st, g, (const bf16*)src, idx, part, row);
  MI_LAUNCH_CHECK();
  return launch_reduce_rows(part, out, nb, n_src, n_src, 0, 1.0f, st);
}
