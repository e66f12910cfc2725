// Prototype (NOT part of the library): the pixel-axis contraction ("Gram": 1x1-conv weight gradients, q k^T) on the channel-blocked
// layout of pw_blocked.hip:   G[ma][mb] = sum_px A[px][ma] * B[px][mb],   A: [z][MA/32][N][32], B: [z][MB/32][N][32]  (bf16).
// The MFMA wants the contraction index (pixels) contiguous per lane, the layout has channels contiguous, so every operand fragment
// is transposed on the way: 16-byte global loads -> the wave's private LDS patch ([32 px][32 ch]) -> ds_read_b64_tr_b16.  No workgroup
// barrier: the patch belongs to one wave.  One wave owns a (16 FA) x (16 FB) tile of G and a slice of the pixels; slices are summed
// with fp32 atomics into a zeroed G (prototype shortcut - the library would use its two-stage reduce).
// Build: hipcc --offload-arch=gfx950 -O3 -o gram_blocked gram_blocked.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
static inline bf16_t f2bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t b) { unsigned u = (unsigned)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }

__device__ __forceinline__ s16x4 lds_tr_b16(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int PROW = 40;   // LDS patch row stride (elements): 32 channels + pad

// BA, BB: 32-channel blocks of A / B per wave tile (FA = 2 BA, FB = 2 BB fragments)
template <int BA, int BB, int MW>
__global__ __launch_bounds__(64 * MW) void gram_blocked_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ G,
                                                               int MA, int MB, long N, int px_per_wave) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[MW][BA + BB][32 * PROW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  const int z = blockIdx.z;
  const int tiles_b = (MB / 32 + BB - 1) / BB;
  const int ta = blockIdx.y / tiles_b, tb = blockIdx.y % tiles_b;
  const long n_begin = ((long)blockIdx.x * MW + wv) * px_per_wave;
  if (n_begin >= N) return;
  const long n_end = min(N, n_begin + px_per_wave);
  const bf16_t* Az = A + (long)z * (MA / 32) * N * 32;
  const bf16_t* Bz = B + (long)z * (MB / 32) * N * 32;
  f32x4 acc[2 * BA][2 * BB];
#pragma unroll
  for (int a = 0; a < 2 * BA; ++a)
#pragma unroll
    for (int b = 0; b < 2 * BB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // raw[blk][half]: pixels n + 16 half + li, channels 8 g .. 8 g + 7 of block blk
  u32x4 raw[BA + BB][2], nxt[BA + BB][2];
  auto load = [&](u32x4 (&r)[BA + BB][2], long n) {
#pragma unroll
    for (int k = 0; k < BA + BB; ++k) {
      // (a tile hanging over the last 32-channel block re-reads that block; its rows are dropped at the end)
      const bf16_t* base = k < BA ? Az + (long)min(ta * BA + k, MA / 32 - 1) * N * 32 : Bz + (long)min(tb * BB + k - BA, MB / 32 - 1) * N * 32;
#pragma unroll
      for (int h = 0; h < 2; ++h) r[k][h] = *reinterpret_cast<const u32x4*>(base + (n + 16 * h + li) * 32 + 8 * g);
    }
  };
  load(raw, n_begin);
  for (long n = n_begin; n < n_end; n += 32) {
    if (n + 32 < n_end) load(nxt, n + 32);
#pragma unroll
    for (int k = 0; k < BA + BB; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(&patch[wv][k][(16 * h + li) * PROW + 8 * g]) = raw[k][h];
    wave_lds_sync();
    // operand fragment f (16 channels) of block k: lane (li = channel, g) gets pixels 4g..4g+3 (lo) and 16+4g..+3 (hi)
    s16x4 lo[2 * (BA + BB)], hi[2 * (BA + BB)];
#pragma unroll
    for (int k = 0; k < BA + BB; ++k)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        lo[2 * k + f] = lds_tr_b16(&patch[wv][k][(4 * g + qq) * PROW + 16 * f + 4 * pp]);
        hi[2 * k + f] = lds_tr_b16(&patch[wv][k][(16 + 4 * g + qq) * PROW + 16 * f + 4 * pp]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 2 * (BA + BB); ++q) asm volatile("" : "+v"(lo[q]), "+v"(hi[q]));
    s16x8 frag[2 * (BA + BB)];
#pragma unroll
    for (int q = 0; q < 2 * (BA + BB); ++q) frag[q] = __builtin_shufflevector(lo[q], hi[q], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int a = 0; a < 2 * BA; ++a)
#pragma unroll
      for (int b = 0; b < 2 * BB; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag[a], frag[2 * BA + b], acc[a][b], 0, 0, 0);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < BA + BB; ++k) { raw[k][0] = nxt[k][0]; raw[k][1] = nxt[k][1]; }
  }
  // acc[a][b]: rows (A channels) 4g..4g+3, column (B channel) li
  float* Gz = G + (long)z * MA * MB;
#pragma unroll
  for (int a = 0; a < 2 * BA; ++a)
#pragma unroll
    for (int b = 0; b < 2 * BB; ++b) {
      const int col = (tb * BB) * 32 + 16 * b + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (ta * BA) * 32 + 16 * a + 4 * g + r;
        if (row < MA && col < MB) atomicAdd(&Gz[(long)row * MB + col], acc[a][b][r]);
      }
    }
}

template <int BA, int BB> float run(const bf16_t* A, const bf16_t* B, float* G, int MA, int MB, long N, int Z, int ppw) {
  constexpr int MW = 4;
  const int ta = (MA / 32 + BA - 1) / BA, tb = (MB / 32 + BB - 1) / BB;
  dim3 grid((unsigned)((N + (long)MW * ppw - 1) / ((long)MW * ppw)), ta * tb, Z);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemset(G, 0, (size_t)Z * MA * MB * 4));
  hipLaunchKernelGGL((gram_blocked_kernel<BA, BB, MW>), grid, dim3(64 * MW), 0, 0, A, B, G, MA, MB, N, ppw);
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); const int it = 5;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gram_blocked_kernel<BA, BB, MW>), grid, dim3(64 * MW), 0, 0, A, B, G, MA, MB, N, ppw);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / it * 1e3f;
}

int main() {
  const long N = 256 * 256; const int Z = 32;
  const int ppw = getenv("PPW") ? atoi(getenv("PPW")) : 1024;
  struct Shape { int MA, MB; } shapes[] = {{512, 96}, {288, 96}, {96, 96}, {256, 96}};
  for (auto sh : shapes) {
    const int MA = sh.MA, MB = sh.MB;
    const size_t ae = (size_t)Z * MA * N, be = (size_t)Z * MB * N;
    bf16_t *A, *B; float* G;
    CK(hipMalloc(&A, ae * 2)); CK(hipMalloc(&B, be * 2)); CK(hipMalloc(&G, (size_t)Z * MA * MB * 4));
    // image 0: random values on the first 2048 pixels, zero elsewhere (so the check can use a short host sum); other images constant
    CK(hipMemset(A, 0x3c, ae * 2)); CK(hipMemset(B, 0x3c, be * 2));
    CK(hipMemset(A, 0, (size_t)MA * N * 2)); CK(hipMemset(B, 0, (size_t)MB * N * 2));
    const int NP = 2048;
    std::vector<bf16_t> ha((size_t)MA * NP), hb((size_t)MB * NP);
    srand(3);
    for (auto& v : ha) v = f2bf((rand() % 200 - 100) / 100.f);
    for (auto& v : hb) v = f2bf((rand() % 200 - 100) / 100.f);
    for (int blk = 0; blk < MA / 32; ++blk) CK(hipMemcpy(A + (size_t)blk * N * 32, ha.data() + (size_t)blk * NP * 32, (size_t)NP * 32 * 2, hipMemcpyHostToDevice));
    for (int blk = 0; blk < MB / 32; ++blk) CK(hipMemcpy(B + (size_t)blk * N * 32, hb.data() + (size_t)blk * NP * 32, (size_t)NP * 32 * 2, hipMemcpyHostToDevice));
    float us;
    const char* tile = getenv("TILE") ? getenv("TILE") : "2x3";
    if (!strcmp(tile, "2x3")) us = run<2, 3>(A, B, G, MA, MB, N, Z, ppw);
    else if (!strcmp(tile, "1x3")) us = run<1, 3>(A, B, G, MA, MB, N, Z, ppw);
    else if (!strcmp(tile, "2x1")) us = run<2, 1>(A, B, G, MA, MB, N, Z, ppw);
    else us = run<2, 2>(A, B, G, MA, MB, N, Z, ppw);
    // check: G was accumulated 6 times (1 + 5 launches) after the memset
    std::vector<float> hg((size_t)MA * MB);
    CK(hipMemcpy(hg.data(), G, hg.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int a = 0; a < MA; a += 13) for (int b = 0; b < MB; b += 7) {
      double ref = 0;
      for (int p = 0; p < NP; ++p) ref += (double)bf2f(ha[((size_t)(a / 32) * NP + p) * 32 + (a & 31)]) * bf2f(hb[((size_t)(b / 32) * NP + p) * 32 + (b & 31)]);
      maxerr = fmax(maxerr, fabs(hg[(size_t)a * MB + b] / 6.0 - ref) / fmax(1.0, fabs(ref)));
    }
    const double gb = ((double)MA + MB) * N * Z * 2 / 1e9, tf = 2.0 * MA * MB * N * Z / 1e12;
    printf("blocked gram %s MA=%4d MB=%4d 256x256 x%d ppw %d: %8.1f us  %6.0f GB/s  %6.1f TF/s   max rel err %.3g\n", tile, MA, MB, Z, ppw, us, gb / us * 1e6,
           tf / us * 1e6, maxerr);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(G));
  }
  return 0;
}
