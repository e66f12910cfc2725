// Prototype (NOT part of the library): the 1x1-conv GEMM on a CHANNEL-BLOCKED activation layout
//     X[z][K/32][N][32], Y[z][M/32][N][32]   (bf16; 32 consecutive channels of a pixel are 64 contiguous bytes)
// instead of the plane-major NCHW the library uses.  Question it answers: how much of the 55-80% roofline efficiency of
// pw_gemm is the layout?  In this layout a 16-pixel x 32-channel MFMA operand fragment is one contiguous KiB of global
// memory in exactly the register layout v_mfma_f32_16x16x32_bf16 wants (lane = pixel, 8 consecutive k), so X needs no LDS,
// no transpose and no barrier; the weights sit in LDS for the whole workgroup; outputs leave as 8-byte lane stores that
// tile whole 64-byte pixel records.
//     D[ch][px] += A[ch][k] * B[k][px],  A = weights (LDS), B = X fragment (global -> registers)
// Build: hipcc --offload-arch=gfx950 -O3 -o pw_blocked pw_blocked.hip      Run: ./pw_blocked
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
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

// MFMA row li of fragment mf <-> channel (within the workgroup's tile): pairs of fragments interleave in groups of four so that the
// accumulators of a pair hold eight consecutive channels per lane
__device__ inline int arow(int mf, int li) { return 32 * (mf >> 1) + 8 * (li >> 2) + 4 * (mf & 1) + (li & 3); }

// One workgroup = MW waves; the workgroup holds W[m0 .. m0+16*MF) x K in LDS; each wave streams 64-pixel tiles.
//   Wl layout: [kb][mf*16 + row][32 k] bf16, row stride 40 elements (bank padding)
template <int MF, int MW>
__global__ __launch_bounds__(64 * MW) void pw_blocked_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                             bf16_t* __restrict__ Y, int K, int M, long N, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  bf16_t* Wl = reinterpret_cast<bf16_t*>(lds);
  constexpr int TM = 16 * MF, WROW = 40;
  const int kb_n = K / 32;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z, m0 = blockIdx.y * TM;
  // weights: Wp is [M][K] bf16 row-major
  for (int e = t; e < kb_n * TM * 4; e += 64 * MW) {           // 16-byte pieces: 4 per (kb,row)
    const int piece = e & 3, row = (e >> 2) % TM, kb = (e >> 2) / TM;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m0 + row < M) v = *reinterpret_cast<const u32x4*>(Wp + (long)(m0 + row) * K + kb * 32 + piece * 8);
    *reinterpret_cast<u32x4*>(&Wl[(kb * TM + row) * WROW + piece * 8]) = v;
  }
  __syncthreads();
  const bf16_t* Xz = X + (long)z * K * N;
  bf16_t* Yz = Y + (long)z * (long)((M + 31) / 32 * 32) * N;
  const long tile0 = ((long)blockIdx.x * MW + wv) * tiles_per_wave;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const long n0 = (tile0 + tt) * 64;
    if (n0 >= N) break;
    f32x4 acc[MF][4];
#pragma unroll
    for (int a = 0; a < MF; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // B fragments of this tile for k-block kb: pixel frag pf -> 16 bytes at ((kb*N + n0 + 16 pf + li) * 32 + 8 g)
    u32x4 bcur[4], bnxt[4];
#pragma unroll
    for (int pf = 0; pf < 4; ++pf) bcur[pf] = *reinterpret_cast<const u32x4*>(Xz + ((long)0 * N + n0 + 16 * pf + li) * 32 + 8 * g);
    for (int kb = 0; kb < kb_n; ++kb) {
      if (kb + 1 < kb_n) {
#pragma unroll
        for (int pf = 0; pf < 4; ++pf)
          bnxt[pf] = *reinterpret_cast<const u32x4*>(Xz + ((long)(kb + 1) * N + n0 + 16 * pf + li) * 32 + 8 * g);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const s16x8 a = *reinterpret_cast<const s16x8*>(&Wl[(kb * TM + arow(mf, li)) * WROW + 8 * g]);
#pragma unroll
        for (int pf = 0; pf < 4; ++pf)
          acc[mf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(s16x8, bcur[pf]), acc[mf][pf], 0, 0, 0);
      }
#pragma unroll
      for (int pf = 0; pf < 4; ++pf) bcur[pf] = bnxt[pf];
    }
    // with the arow() permutation the MFMA pair (2j, 2j+1) leaves channels m0 + 32 j + 8 g + {0..7} of pixel n0 + 16 pf + li in
    // this lane: one 16-byte store, four lanes fill the pixel's 64-byte record
#pragma unroll
    for (int mj = 0; mj < MF / 2; ++mj) {
      const int ch = m0 + 32 * mj + 8 * g;
      if (ch >= M) continue;
#pragma unroll
      for (int pf = 0; pf < 4; ++pf) {
        u32x4 o = {pack2(acc[2 * mj][pf][0], acc[2 * mj][pf][1]), pack2(acc[2 * mj][pf][2], acc[2 * mj][pf][3]),
                   pack2(acc[2 * mj + 1][pf][0], acc[2 * mj + 1][pf][1]), pack2(acc[2 * mj + 1][pf][2], acc[2 * mj + 1][pf][3])};
        *reinterpret_cast<u32x4*>(Yz + ((long)(ch >> 5) * N + n0 + 16 * pf + li) * 32 + (ch & 31)) = o;
      }
    }
  }
}

template <int MF, int MW> float run(const bf16_t* X, const bf16_t* W, bf16_t* Y, int K, int M, long N, int Z, int tpw) {
  const int TM = 16 * MF;
  const size_t lds = (size_t)(K / 32) * TM * 40 * 2;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)pw_blocked_kernel<MF, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long tiles = N / 64;
  dim3 grid((unsigned)((tiles + (long)MW * tpw - 1) / ((long)MW * tpw)), (M + TM - 1) / TM, Z);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pw_blocked_kernel<MF, MW>), grid, dim3(64 * MW), lds, 0, X, W, Y, K, M, N, tpw);
  CK(hipGetLastError());
  CK(hipEventRecord(e0)); const int it = 5;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((pw_blocked_kernel<MF, MW>), grid, dim3(64 * MW), lds, 0, X, W, Y, K, M, N, tpw);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / it * 1e3f;
}

// Variant for K <= 96 (the write-heavy projections): the workgroup keeps ALL of W (M x K) in LDS, a wave loads its 64-pixel X tile into
// registers once (K/32 x 4 fragments) and walks the output channels 64 at a time - X is read exactly once, Y written exactly once.
template <int KB, int MW>
__global__ __launch_bounds__(64 * MW) void pw_blocked_xres_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wp,
                                                                  bf16_t* __restrict__ Y, int M, long N, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  bf16_t* Wl = reinterpret_cast<bf16_t*>(lds);
  constexpr int WROW = 40, K = 32 * KB;
  const int Mp = (M + 63) / 64 * 64;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z;
  for (int e = t; e < KB * Mp * 4; e += 64 * MW) {
    const int piece = e & 3, row = (e >> 2) % Mp, kb = (e >> 2) / Mp;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < M) v = *reinterpret_cast<const u32x4*>(Wp + (long)row * K + kb * 32 + piece * 8);
    *reinterpret_cast<u32x4*>(&Wl[(kb * Mp + row) * WROW + piece * 8]) = v;
  }
  __syncthreads();
  const bf16_t* Xz = X + (long)z * K * N;
  bf16_t* Yz = Y + (long)z * (long)((M + 31) / 32 * 32) * N;
  const long tile0 = ((long)blockIdx.x * MW + wv) * tiles_per_wave;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const long n0 = (tile0 + tt) * 64;
    if (n0 >= N) break;
    u32x4 xb[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int pf = 0; pf < 4; ++pf) xb[kb][pf] = *reinterpret_cast<const u32x4*>(Xz + ((long)kb * N + n0 + 16 * pf + li) * 32 + 8 * g);
    for (int m0 = 0; m0 < Mp; m0 += 64) {
      f32x4 acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const s16x8 a = *reinterpret_cast<const s16x8*>(&Wl[(kb * Mp + m0 + arow(mf, li)) * WROW + 8 * g]);
#pragma unroll
          for (int pf = 0; pf < 4; ++pf)
            acc[mf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(s16x8, xb[kb][pf]), acc[mf][pf], 0, 0, 0);
        }
#pragma unroll
      for (int mj = 0; mj < 2; ++mj) {
        const int ch = m0 + 32 * mj + 8 * g;
        if (ch >= M) continue;
#pragma unroll
        for (int pf = 0; pf < 4; ++pf) {
          u32x4 o = {pack2(acc[2 * mj][pf][0], acc[2 * mj][pf][1]), pack2(acc[2 * mj][pf][2], acc[2 * mj][pf][3]),
                     pack2(acc[2 * mj + 1][pf][0], acc[2 * mj + 1][pf][1]), pack2(acc[2 * mj + 1][pf][2], acc[2 * mj + 1][pf][3])};
          *reinterpret_cast<u32x4*>(Yz + ((long)(ch >> 5) * N + n0 + 16 * pf + li) * 32 + (ch & 31)) = o;
        }
      }
    }
  }
}

template <int KB, int MW> float run_xres(const bf16_t* X, const bf16_t* W, bf16_t* Y, int M, long N, int Z, int tpw) {
  const int Mp = (M + 63) / 64 * 64;
  const size_t lds = (size_t)KB * Mp * 40 * 2;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)pw_blocked_xres_kernel<KB, MW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long tiles = N / 64;
  dim3 grid((unsigned)((tiles + (long)MW * tpw - 1) / ((long)MW * tpw)), 1, Z);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pw_blocked_xres_kernel<KB, MW>), grid, dim3(64 * MW), lds, 0, X, W, Y, M, N, tpw);
  CK(hipGetLastError());
  CK(hipEventRecord(e0)); const int it = 5;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((pw_blocked_xres_kernel<KB, MW>), grid, dim3(64 * MW), lds, 0, X, W, Y, M, N, tpw);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / it * 1e3f;
}

int main() {
  const long N = 256 * 256; const int Z = 32;
  struct Shape { int M, K; } shapes[] = {{510, 96}, {288, 96}, {96, 96}, {96, 288}, {96, 512}, {144, 64}, {64, 160}, {254, 64}, {192, 32}};
  for (auto sh : shapes) {
    const int M = sh.M, K = sh.K, Mp = (M + 31) / 32 * 32;
    size_t xe = (size_t)Z * K * N, ye = (size_t)Z * Mp * N;
    bf16_t *X, *W, *Y;
    CK(hipMalloc(&X, xe * 2)); CK(hipMalloc(&Y, ye * 2)); CK(hipMalloc(&W, (size_t)M * K * 2));
    std::vector<bf16_t> hx((size_t)K * 4096), hw((size_t)M * K);
    srand(1);
    for (auto& v : hw) v = f2bf((rand() % 200 - 100) / 100.f);
    // fill X: first image, first 4096 pixels of every k-block with known values (for the check), the rest with a pattern via memset
    CK(hipMemset(X, 0x3c, xe * 2));
    for (int kb = 0; kb < K / 32; ++kb) {
      for (long n = 0; n < 128; ++n) for (int c = 0; c < 32; ++c) hx[((size_t)kb * 128 + n) * 32 + c] = f2bf((rand() % 200 - 100) / 100.f);
      CK(hipMemcpy(X + ((size_t)kb * N) * 32, hx.data() + (size_t)kb * 128 * 32, 128 * 32 * 2, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    float us;
    const double gb = ((double)K + M) * N * Z * 2 / 1e9, tf = 2.0 * M * K * N * Z / 1e12;
    const char* form = getenv("FORM") ? getenv("FORM") : "auto";
    const int mw = getenv("MW") ? atoi(getenv("MW")) : 8, tpw = getenv("TPW") ? atoi(getenv("TPW")) : 4;
    const bool xres = (!strcmp(form, "auto") && K <= 96 && M > 96) || !strcmp(form, "xres");
    if (xres && K % 32 == 0 && K <= 96) {
      if (mw == 16) us = K == 96 ? run_xres<3, 16>(X, W, Y, M, N, Z, tpw) : K == 64 ? run_xres<2, 16>(X, W, Y, M, N, Z, tpw) : run_xres<1, 16>(X, W, Y, M, N, Z, tpw);
      else us = K == 96 ? run_xres<3, 8>(X, W, Y, M, N, Z, tpw) : K == 64 ? run_xres<2, 8>(X, W, Y, M, N, Z, tpw) : run_xres<1, 8>(X, W, Y, M, N, Z, tpw);
    } else if (K <= 96) us = run<8, 4>(X, W, Y, K, M, N, Z, 8);  // 128-channel tile, weights <= 30 KB
    else us = run<6, 8>(X, W, Y, K, M, N, Z, tpw);               // 96-channel tile, 8 waves share up to 120 KB of weights
    printf("%s ", xres ? "xres  " : "stream");
    // check pixel 0..127 of image 0
    std::vector<bf16_t> hy((size_t)Mp / 32 * 128 * 32);
    double maxerr = 0;
    for (int mb = 0; mb < Mp / 32; ++mb) CK(hipMemcpy(hy.data() + (size_t)mb * 128 * 32, Y + ((size_t)mb * N) * 32, 128 * 32 * 2, hipMemcpyDeviceToHost));
    for (int m = 0; m < M; m += 7) for (long n = 0; n < 128; n += 5) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hw[(size_t)m * K + k]) * bf2f(hx[((size_t)(k / 32) * 128 + n) * 32 + (k & 31)]);
      const double got = bf2f(hy[((size_t)(m / 32) * 128 + n) * 32 + (m & 31)]);
      maxerr = fmax(maxerr, fabs(got - ref) / fmax(1.0, fabs(ref)));
    }
    printf("blocked pw M=%4d K=%4d 256x256 x%d: %8.1f us  %6.0f GB/s  %6.1f TF/s   max rel err %.3g\n", M, K, Z, us, gb / us * 1e6, tf / us * 1e6, maxerr);
    CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(W));
  }
  return 0;
}
