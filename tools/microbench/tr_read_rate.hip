// Throughput of ds_read_b64_tr_b16 on the backward tail's dY patch layout (rows of PS bf16, 16-row x 32-byte blocks per wave
// instruction) against the row stride, with 1..8 waves per workgroup hammering LDS at once, and of plain ds_read_b64 for scale.
//   hipcc --offload-arch=gfx950 -O3 -o tr_read_rate tools/microbench/tr_read_rate.hip && ./tr_read_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 tr_b16(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(size_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ s16x4 rd_b64(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(size_t)p;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int PS, bool TR>
__global__ __launch_bounds__(512) void k(long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  for (int i = threadIdx.x; i < 64 * PS * 8 / 2; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  // the tail's transposed read: 16-lane group g reads rows 4g + qq, 8 bytes at column chunk pp; plain: row li, 8 bytes at 4 g
  const short* base = reinterpret_cast<const short*>(lds) + (TR ? (4 * g + qq) * PS + 4 * pp : li * PS + 4 * g);
  s16x4 acc = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    s16x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const short* p = base + ((it + wv) & 7) * 64 * PS / 8 * 0 + (j & 3) * 16 + (j >> 2) * 16 * PS + (wv & 3) * 16 * PS * 0;
      v[j] = TR ? tr_b16(p) : rd_b64(p);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 8 + wv] = t1 - t0;
  if (acc[0] == 12345 && acc[1] == 777) out[0] = 0;
}
template <int PS, bool TR> void run(int waves, long long* d) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<PS, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * PS * 8 * 2 + 1024);
  hipLaunchKernelGGL((k<PS, TR>), dim3(256), dim3(64 * waves), 64 * PS * 8 * 2 + 1024, 0, d, iters);
  hipDeviceSynchronize();
  long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("  %s PS=%3d waves=%d: %6.1f clock64 ticks per wave-read (wave 0), %6.1f per read CU-wide\n", TR ? "tr_b16" : "b64   ", PS, waves,
         (double)h[0] / (iters * 8), (double)h[0] / (iters * 8) / waves);
}
int main() {
  long long* d;
  hipMalloc(&d, 256 * 8 * sizeof(long long));
  for (int waves : {1, 2, 4, 6, 8}) {
    run<72, true>(waves, d); run<80, true>(waves, d); run<88, true>(waves, d); run<68, true>(waves, d); run<136, true>(waves, d);
    run<72, false>(waves, d);
  }
  return 0;
}
