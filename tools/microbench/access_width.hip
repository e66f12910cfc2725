// Microbenchmark: streaming copy (1 read : 1 write, 1 GiB) with 4-, 8- and 16-byte per-lane accesses, at two occupancies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <typename V, int UNROLL>
__global__ __launch_bounds__(256) void copy_k(const V* __restrict__ in, V* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    V v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) out[i + u * stride] = v[u];
  }
  for (; i < n; i += stride) out[i] = in[i];
}
template <typename V, int UNROLL> void run(const void* a, void* b, size_t bytes, int blocks, const char* name) {
  const long n = bytes / sizeof(V);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((copy_k<V, UNROLL>), dim3(blocks), dim3(256), 0, 0, (const V*)a, (V*)b, n);
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((copy_k<V, UNROLL>), dim3(blocks), dim3(256), 0, 0, (const V*)a, (V*)b, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  %-28s blocks %5d: %6.0f GB/s\n", name, blocks, 2.0 * bytes / 1e9 / (ms / 5 * 1e-3));
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  for (int blocks : {1024, 2048, 8192}) {
    run<unsigned int, 1>(a, b, bytes, blocks, "4 B/lane, 1 in flight");
    run<unsigned int, 4>(a, b, bytes, blocks, "4 B/lane, 4 in flight");
    run<u32x2, 1>(a, b, bytes, blocks, "8 B/lane, 1 in flight");
    run<u32x2, 4>(a, b, bytes, blocks, "8 B/lane, 4 in flight");
    run<u32x4, 1>(a, b, bytes, blocks, "16 B/lane, 1 in flight");
    run<u32x4, 4>(a, b, bytes, blocks, "16 B/lane, 4 in flight");
  }
  return 0;
}
