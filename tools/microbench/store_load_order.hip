// What does a load cost a wave when stores are ahead of it in the wave's vector-memory queue?  (gfx950: one in-order vmcnt for
// loads and stores.)  Every wave of a persistent 512-thread workgroup per CU loops: [optionally store 1 KiB per wave to a
// streaming buffer] -> load 16 B per lane from a small L2-resident table -> s_waitcnt vmcnt(0) on the load's data -> ~W cycles of
// ALU work.  Prints cycles per iteration for the variants.   hipcc --offload-arch=gfx950 -O3 store_load_order.hip -o store_load_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(512) void k(u32x4* out, const u32x4* tab, unsigned long long* cyc, int iters, int nstore, int work, int dep) {
  const int t = threadIdx.x, w = t >> 6;
  u32x4* o = out + ((size_t)blockIdx.x * 8 + w) * (size_t)iters * 64 * 8 + (t & 63);
  const u32x4* tb = tab + ((blockIdx.x * 8 + w) & 63) * 64 + (t & 63);
  u32x4 acc = {1u, 2u, 3u, 4u};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    for (int s = 0; s < nstore; ++s) o[((size_t)i * 8 + s) * 64] = acc;
    u32x4 v = __builtin_nontemporal_load(tb + (i & 7) * 4096);
    if (dep) { acc[0] += v[0]; acc[1] ^= v[1]; }            // consume the load right away: waits for it (and everything older)
    for (int j = 0; j < work; ++j) { acc[2] = acc[2] * 1664525u + 1013904223u; acc[3] += acc[2] >> 3; }
    if (!dep) { acc[0] += v[0]; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((t & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
  if (acc[0] == 0x12345678u) out[0] = acc;
}
int main() {
  const int iters = 2000, NB = 256;
  u32x4 *out, *tab; unsigned long long* cyc;
  hipMalloc(&out, (size_t)NB * 8 * iters * 64 * 8 * 16);
  hipMalloc(&tab, 64 * 4096 * 16 * 2);
  hipMemset(tab, 1, 64 * 4096 * 16 * 2);
  hipMalloc(&cyc, NB * 8 * 8);
  unsigned long long* h = (unsigned long long*)malloc(NB * 8 * 8);
  for (int work : {64, 512, 2048})
    for (int nstore : {0, 1, 4, 8})
      for (int dep : {1, 0}) {
        hipLaunchKernelGGL(k, dim3(NB), dim3(512), 0, 0, out, tab, cyc, iters, nstore, work, dep);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(NB), dim3(512), 0, 0, out, tab, cyc, iters, nstore, work, dep); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, NB * 8 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < NB * 8; ++i) s += h[i];
        printf("work %4d  stores/iter %d (%4.0f GB/s chip-wide)  load consumed %s: %8.0f cycles/iter  (%.1f us total)\n", work, nstore,
               (double)NB * 8 * iters * nstore * 1024 / ms / 1e6, dep ? "at once " : "after ALU", s / (NB * 8) / iters * 1.0, ms * 1e3);
      }
  return 0;
}
