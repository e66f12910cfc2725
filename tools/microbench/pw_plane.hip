// Prototype (NOT part of the library): the structure of pw_blocked.hip - weights resident in LDS for the whole workgroup, every wave
// streaming its OWN 64-pixel tiles with no workgroup barrier - on the library's plane-major NCHW layout.  The two transposes the
// layout forces (X: [k][px] -> k-contiguous MFMA operand; Y: accumulator -> 128-byte channel rows) go through a small LDS patch that
// belongs to one wave (ds_read_b64_tr_b16 / 8-byte writes + 16-byte row reads), so waves still never wait for each other.
//     X: [z][K][N], Y: [z][M][N] bf16;   D[px][ch] += A[px][k] * B[k][ch],  A = X tile (via the patch), B = weights (LDS)
// "xres" (K <= 96): the X tile is turned into operand registers once, the wave walks all output channels 64 at a time.
// "stream" (M <= 96): accumulators for all M, X chunks of 32 k stream through the patch with register prefetch.
// Build: hipcc --offload-arch=gfx950 -O3 -o pw_plane pw_plane.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
static inline bf16_t f2bf(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t b) { unsigned u = (unsigned)b << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
__device__ inline unsigned pack2(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fff + ((ua >> 16) & 1); ub += 0x7fff + ((ub >> 16) & 1);
  return (ua >> 16) | (ub & 0xffff0000u);
}
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

constexpr int WROW = 40;    // weight row stride in LDS (32 k + pad), elements
constexpr int XS = 72;      // patch row stride (64 px + pad), elements
constexpr int PATCH = 32 * XS;   // elements: one 32-row x 64-pixel patch per wave (X chunk, or 32 output channels)

// raw global loads of one X chunk (32 k x 64 px): instruction i covers rows 8i + lane/8, 16 bytes (8 px) per lane
__device__ __forceinline__ void load_chunk(u32x4 (&raw)[4], const bf16_t* Xz, long N, int kb, long n0, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) raw[i] = *reinterpret_cast<const u32x4*>(Xz + (long)(kb * 32 + 8 * i + (lane >> 3)) * N + n0 + 8 * (lane & 7));
}
// chunk -> patch -> MFMA A operands a[nf] (k slots: element j<4 of lane group g is k = 4g+j, element j>=4 is k = 16+4g+(j-4))
__device__ __forceinline__ void chunk_to_frags(s16x8 (&a)[4], const u32x4 (&raw)[4], bf16_t* patch, int lane) {
  const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(&patch[(8 * i + (lane >> 3)) * XS + 8 * (lane & 7)]) = raw[i];
  wave_lds_sync();
  s16x4 lo[4], hi[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    lo[nf] = lds_tr_b16(&patch[(4 * g + qq) * XS + 16 * nf + 4 * pp]);
    hi[nf] = lds_tr_b16(&patch[(16 + 4 * g + qq) * XS + 16 * nf + 4 * pp]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])::"memory");
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) a[nf] = __builtin_shufflevector(lo[nf], hi[nf], 0, 1, 2, 3, 4, 5, 6, 7);
  wave_lds_sync();
}
// weight operand of fragment row `row` (channel), k-block kb, same k-slot permutation
__device__ __forceinline__ s16x8 w_frag(const bf16_t* Wl, int rows, int kb, int row, int g) {
  const bf16_t* wr = &Wl[((long)kb * rows + row) * WROW + 4 * g];
  const s16x4 lo = *reinterpret_cast<const s16x4*>(wr), hi = *reinterpret_cast<const s16x4*>(wr + 16);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// 32 output channels (two fragment columns of four pixel fragments) -> patch -> 128-byte channel rows in global memory
__device__ __forceinline__ void store_32ch(const f32x4 (&acc0)[4], const f32x4 (&acc1)[4], bf16_t* patch, bf16_t* Yz, long N, int m0, int M,
                                           long n0, int lane) {
  const int li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    *reinterpret_cast<u32x2*>(&patch[li * XS + 16 * nf + 4 * g]) = (u32x2){pack2(acc0[nf][0], acc0[nf][1]), pack2(acc0[nf][2], acc0[nf][3])};
    *reinterpret_cast<u32x2*>(&patch[(16 + li) * XS + 16 * nf + 4 * g]) = (u32x2){pack2(acc1[nf][0], acc1[nf][1]), pack2(acc1[nf][2], acc1[nf][3])};
  }
  wave_lds_sync();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3);
    const u32x4 v = *reinterpret_cast<const u32x4*>(&patch[row * XS + 8 * (lane & 7)]);
    if (m0 + row < M) *reinterpret_cast<u32x4*>(Yz + (long)(m0 + row) * N + n0 + 8 * (lane & 7)) = v;
  }
  wave_lds_sync();
}

__device__ __forceinline__ void stage_weights(bf16_t* Wl, const bf16_t* Wp, int M, int rows, int K, int m_base, int t, int nthreads) {
  const int kb_n = K / 32;
  for (int e = t; e < kb_n * rows * 4; e += nthreads) {
    const int piece = e & 3, row = (e >> 2) % rows, kb = (e >> 2) / rows;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m_base + row < M) v = *reinterpret_cast<const u32x4*>(Wp + (long)(m_base + row) * K + kb * 32 + piece * 8);
    *reinterpret_cast<u32x4*>(&Wl[((long)kb * rows + row) * WROW + piece * 8]) = v;
  }
}

template <int KB, int MW>
__global__ __launch_bounds__(64 * MW) void pw_plane_xres_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, bf16_t* __restrict__ Y,
                                                                int M, long N, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int K = 32 * KB;
  const int Mp = (M + 63) / 64 * 64;
  bf16_t* Wl = reinterpret_cast<bf16_t*>(lds);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, g = lane >> 4;
  bf16_t* patch = Wl + (long)KB * Mp * WROW + wv * PATCH;
  stage_weights(Wl, Wp, M, Mp, K, 0, t, 64 * MW);
  __syncthreads();
  const int z = blockIdx.z;
  const bf16_t* Xz = X + (long)z * K * N;
  bf16_t* Yz = Y + (long)z * M * N;
  const long tile0 = ((long)blockIdx.x * MW + wv) * tiles_per_wave;
  u32x4 raw[KB][4];
  if (tile0 * 64 < N) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) load_chunk(raw[kb], Xz, N, kb, tile0 * 64, lane);
  }
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const long n0 = (tile0 + tt) * 64;
    if (n0 >= N) break;
    s16x8 a[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) chunk_to_frags(a[kb], raw[kb], patch, lane);
    if (tt + 1 < tiles_per_wave && n0 + 64 < N) {           // next tile's X is in flight while this tile's channels are computed
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) load_chunk(raw[kb], Xz, N, kb, n0 + 64, lane);
    }
    for (int m0 = 0; m0 < Mp; m0 += 64) {
      f32x4 acc[4][4];   // [mf][nf]
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const s16x8 b = w_frag(Wl, Mp, kb, m0 + 16 * mf + li, g);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][nf], b, acc[mf][nf], 0, 0, 0);
        }
      store_32ch(acc[0], acc[1], patch, Yz, N, m0, M, n0, lane);
      store_32ch(acc[2], acc[3], patch, Yz, N, m0 + 32, M, n0, lane);
    }
  }
}

// M <= 16 MF: all output channels in accumulators, X streams through the patch one 32-k chunk at a time
template <int MF, int MW>
__global__ __launch_bounds__(64 * MW) void pw_plane_stream_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp, bf16_t* __restrict__ Y,
                                                                  int K, int M, long N, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int TM = 16 * MF;
  const int kb_n = K / 32;
  bf16_t* Wl = reinterpret_cast<bf16_t*>(lds);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, g = lane >> 4;
  bf16_t* patch = Wl + (long)kb_n * TM * WROW + wv * PATCH;
  const int m_base = blockIdx.y * TM;
  stage_weights(Wl, Wp, M, TM, K, m_base, t, 64 * MW);
  __syncthreads();
  const int z = blockIdx.z;
  const bf16_t* Xz = X + (long)z * K * N;
  bf16_t* Yz = Y + (long)z * M * N;
  const long tile0 = ((long)blockIdx.x * MW + wv) * tiles_per_wave;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const long n0 = (tile0 + tt) * 64;
    if (n0 >= N) break;
    f32x4 acc[MF][4];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 r0[4], r1[4];
    load_chunk(r0, Xz, N, 0, n0, lane);
    if (kb_n > 1) load_chunk(r1, Xz, N, 1, n0, lane);
    for (int kb = 0; kb < kb_n; kb += 2) {          // two chunks per trip: the raw registers are statically named
      {
        s16x8 a[4];
        chunk_to_frags(a, r0, patch, lane);
        if (kb + 2 < kb_n) load_chunk(r0, Xz, N, kb + 2, n0, lane);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const s16x8 b = w_frag(Wl, TM, kb, 16 * mf + li, g);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nf], b, acc[mf][nf], 0, 0, 0);
        }
      }
      if (kb + 1 < kb_n) {
        s16x8 a[4];
        chunk_to_frags(a, r1, patch, lane);
        if (kb + 3 < kb_n) load_chunk(r1, Xz, N, kb + 3, n0, lane);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const s16x8 b = w_frag(Wl, TM, kb + 1, 16 * mf + li, g);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nf], b, acc[mf][nf], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int mj = 0; mj < MF / 2; ++mj) store_32ch(acc[2 * mj], acc[2 * mj + 1], patch, Yz, N, m_base + 32 * mj, M, n0, lane);
  }
}

template <typename F> float time_it(F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipGetLastError()); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); const int it = 5;
  for (int i = 0; i < it; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / it * 1e3f;
}

template <int KB, int MW> float run_xres(const bf16_t* X, const bf16_t* W, bf16_t* Y, int M, long N, int Z, int tpw) {
  const int Mp = (M + 63) / 64 * 64;
  const size_t lds = ((size_t)KB * Mp * WROW + (size_t)MW * PATCH) * 2;
  if (lds > 160 * 1024) { printf("(LDS %zu too large) ", lds); return 0.f; }
  CK(hipFuncSetAttribute((const void*)pw_plane_xres_kernel<KB, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long tiles = N / 64;
  dim3 grid((unsigned)((tiles + (long)MW * tpw - 1) / ((long)MW * tpw)), 1, Z);
  return time_it([&]() { hipLaunchKernelGGL((pw_plane_xres_kernel<KB, MW>), grid, dim3(64 * MW), lds, 0, X, W, Y, M, N, tpw); });
}
template <int MF, int MW> float run_stream(const bf16_t* X, const bf16_t* W, bf16_t* Y, int K, int M, long N, int Z, int tpw) {
  const int TM = 16 * MF;
  const size_t lds = ((size_t)(K / 32) * TM * WROW + (size_t)MW * PATCH) * 2;
  if (lds > 160 * 1024) { printf("(LDS %zu too large) ", lds); return 0.f; }
  CK(hipFuncSetAttribute((const void*)pw_plane_stream_kernel<MF, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long tiles = N / 64;
  dim3 grid((unsigned)((tiles + (long)MW * tpw - 1) / ((long)MW * tpw)), (M + TM - 1) / TM, Z);
  return time_it([&]() { hipLaunchKernelGGL((pw_plane_stream_kernel<MF, MW>), grid, dim3(64 * MW), lds, 0, X, W, Y, K, M, N, tpw); });
}

int main() {
  const long N = 256 * 256; const int Z = getenv("Z") ? atoi(getenv("Z")) : 32;
  const int tpw = getenv("TPW") ? atoi(getenv("TPW")) : 4;
  const int mw = getenv("MW") ? atoi(getenv("MW")) : 8;
  struct Shape { int M, K; } shapes[] = {{510, 96}, {288, 96}, {96, 96}, {96, 288}, {96, 512}, {144, 64}, {64, 160}, {254, 64}, {192, 32}};
  for (auto sh : shapes) {
    const int M = sh.M, K = sh.K;
    const size_t xe = (size_t)Z * K * N, ye = (size_t)Z * M * N;
    bf16_t *X, *W, *Y;
    CK(hipMalloc(&X, xe * 2)); CK(hipMalloc(&Y, ye * 2)); CK(hipMalloc(&W, (size_t)M * K * 2));
    const int NP = 128;
    std::vector<bf16_t> hx((size_t)K * NP), hw((size_t)M * K);
    srand(1);
    for (auto& v : hw) v = f2bf((rand() % 200 - 100) / 100.f);
    for (auto& v : hx) v = f2bf((rand() % 200 - 100) / 100.f);
    CK(hipMemset(X, 0x3c, xe * 2));
    for (int k = 0; k < K; ++k) CK(hipMemcpy(X + (size_t)k * N, hx.data() + (size_t)k * NP, NP * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const bool xres = K <= 96 && M > 96;
    float us;
    if (xres) {
      if (mw == 8) us = K == 96 ? run_xres<3, 8>(X, W, Y, M, N, Z, tpw) : K == 64 ? run_xres<2, 8>(X, W, Y, M, N, Z, tpw) : run_xres<1, 8>(X, W, Y, M, N, Z, tpw);
      else us = K == 96 ? run_xres<3, 4>(X, W, Y, M, N, Z, tpw) : K == 64 ? run_xres<2, 4>(X, W, Y, M, N, Z, tpw) : run_xres<1, 4>(X, W, Y, M, N, Z, tpw);
    } else {
      if (M <= 64) us = mw == 8 ? run_stream<4, 8>(X, W, Y, K, M, N, Z, tpw) : run_stream<4, 4>(X, W, Y, K, M, N, Z, tpw);
      else us = mw == 8 ? run_stream<6, 8>(X, W, Y, K, M, N, Z, tpw) : run_stream<6, 4>(X, W, Y, K, M, N, Z, tpw);
    }
    std::vector<bf16_t> hy((size_t)M * NP);
    for (int m = 0; m < M; ++m) CK(hipMemcpy(hy.data() + (size_t)m * NP, Y + (size_t)m * N, NP * 2, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int m = 0; m < M; m += 7) for (int n = 0; n < NP; n += 5) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hw[(size_t)m * K + k]) * bf2f(hx[(size_t)k * NP + n]);
      maxerr = fmax(maxerr, fabs(bf2f(hy[(size_t)m * NP + n]) - ref) / fmax(1.0, fabs(ref)));
    }
    const double gb = ((double)K + M) * N * Z * 2 / 1e9, tf = 2.0 * M * K * N * Z / 1e12;
    printf("%s plane pw M=%4d K=%4d 256x256 x%d: %8.1f us  %6.0f GB/s  %6.1f TF/s   max rel err %.3g\n", xres ? "xres  " : "stream", M, K, Z, us,
           us > 0 ? gb / us * 1e6 : 0.0, us > 0 ? tf / us * 1e6 : 0.0, maxerr);
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(W));
  }
  return 0;
}
