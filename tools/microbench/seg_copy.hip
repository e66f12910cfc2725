// Microbenchmark: NCHW channel-strided tile copy.  One workgroup copies a [C rows] x [S bytes] tile (row stride = N*2 B)
// from in to out; sweep S to see what per-row segment length the MI355X HBM path needs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SEG16>  // 16-byte vectors per row segment
__global__ __launch_bounds__(256) void seg_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, int C, long rowv /*vectors per row*/,
                                                long chan_stride_v) {
  constexpr int RP = 256 / SEG16;          // rows in parallel
  const int tv = threadIdx.x % SEG16, tr = threadIdx.x / SEG16;
  const long base = (long)blockIdx.y * C * chan_stride_v + (long)blockIdx.x * SEG16 + tv;
  for (int c0 = 0; c0 < C; c0 += RP * 4) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int c = c0 + u * RP + tr; if (c < C) v[u] = in[base + c * chan_stride_v]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { int c = c0 + u * RP + tr; if (c < C) out[base + c * chan_stride_v] = v[u]; }
  }
}
// flat streaming copy for reference
__global__ __launch_bounds__(256) void flat_copy(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n) {
  long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = in[i];
}

template <int SEG16> float run(const u32x4* in, u32x4* out, int B, int C, long N) {
  long rowv = N * 2 / 16;
  dim3 grid(rowv / SEG16, B);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(seg_copy<SEG16>, grid, dim3(256), 0, 0, in, out, C, rowv, rowv);
  CK(hipEventRecord(e0)); const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(seg_copy<SEG16>, grid, dim3(256), 0, 0, in, out, C, rowv, rowv);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / it;
}
int main() {
  const int B = getenv("B") ? atoi(getenv("B")) : 8; const long N = 65536;
  for (int C : {48, 144, 254}) {
    size_t bytes = (size_t)B * C * N * 2;
    u32x4 *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes));
    double gb = 2.0 * bytes / 1e9;
    printf("C=%d  (%.0f MB each way)\n", C, bytes / 1e6);
    printf("  seg  128B: %7.0f GB/s\n", gb / (run<8>(in, out, B, C, N) * 1e-3));
    printf("  seg  256B: %7.0f GB/s\n", gb / (run<16>(in, out, B, C, N) * 1e-3));
    printf("  seg  512B: %7.0f GB/s\n", gb / (run<32>(in, out, B, C, N) * 1e-3));
    printf("  seg 1024B: %7.0f GB/s\n", gb / (run<64>(in, out, B, C, N) * 1e-3));
    printf("  seg 2048B: %7.0f GB/s\n", gb / (run<128>(in, out, B, C, N) * 1e-3));
    printf("  seg 4096B: %7.0f GB/s\n", gb / (run<256>(in, out, B, C, N) * 1e-3));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long nv = bytes / 16;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(flat_copy, dim3(2048), dim3(256), 0, 0, in, out, nv);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(flat_copy, dim3(2048), dim3(256), 0, 0, in, out, nv);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  flat copy: %7.0f GB/s\n", gb / (ms / 10 * 1e-3));
    CK(hipFree(in)); CK(hipFree(out));
  }
  return 0;
}
