// Microbenchmark: HBM read-only / write-only / mixed streaming ceilings on MI355X (16 B per lane, grid-stride).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256;
  u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = v;
}
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ in, unsigned* __restrict__ sink, long n) {
  const long stride = (long)gridDim.x * 256;
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { u32x4 v = in[i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x12345678u) sink[0] = acc;
}
// reads n_in vectors, writes R times as many (R output rows per input row), like a 1x1 conv with M = R*K
template <int R>
__global__ __launch_bounds__(256) void k_mixed(const u32x4* __restrict__ in, u32x4* __restrict__ out, long n_in) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_in; i += stride) {
    u32x4 v = in[i];
#pragma unroll
    for (int r = 0; r < R; ++r) { v[0] += r; out[(long)r * n_in + i] = v; }
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
  CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10;
}
int main() {
  const size_t bytes = (size_t)1 << 30;  // 1 GiB buffers: far beyond the 256 MiB Infinity Cache
  u32x4 *a, *b; unsigned* sink; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, 6 * bytes / 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, bytes));
  const long n = bytes / 16;
  for (int blocks : {2048, 8192}) {
    printf("grid %d blocks:\n", blocks);
    float ms = timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, a, n); });
    printf("  write-only : %7.0f GB/s\n", bytes / 1e9 / (ms * 1e-3));
    ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, sink, n); });
    printf("  read-only  : %7.0f GB/s\n", bytes / 1e9 / (ms * 1e-3));
    const long nin = n / 4;  // 256 MiB in
    ms = timeit([&] { hipLaunchKernelGGL(k_mixed<1>, dim3(blocks), dim3(256), 0, 0, a, b, nin); });
    printf("  1 read : 1 write : %7.0f GB/s total\n", 2.0 * nin * 16 / 1e9 / (ms * 1e-3));
    ms = timeit([&] { hipLaunchKernelGGL(k_mixed<3>, dim3(blocks), dim3(256), 0, 0, a, b, nin); });
    printf("  1 read : 3 write : %7.0f GB/s total\n", 4.0 * nin * 16 / 1e9 / (ms * 1e-3));
    ms = timeit([&] { hipLaunchKernelGGL(k_mixed<5>, dim3(blocks), dim3(256), 0, 0, a, b, nin); });
    printf("  1 read : 5 write : %7.0f GB/s total\n", 6.0 * nin * 16 / 1e9 / (ms * 1e-3));
  }
  return 0;
}
