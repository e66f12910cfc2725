// Probe of the gfx950 fp8 conversion / MFMA semantics the fp8 projection path relies on (run once on the GPU box):
//   v_cvt_scalef32_pk_fp8_bf16: does it divide or multiply by the scale, does it saturate, which byte gets which element;
//   v_mfma_f32_16x16x32_fp8_fp8: k-slot order against the bf16 form.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void cvt_probe(const float* in, float scale, unsigned* raw, float* back, int n, int ovfl) {
  const int i = threadIdx.x;
  if (ovfl) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);   // MODE.FP16_OVFL = 1: clamp instead of NaN
  if (i >= n) return;
  bf16x2 a = {(__bf16)in[2 * i], (__bf16)in[2 * i + 1]};
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, a, scale, false);
  const unsigned u = (unsigned)__builtin_bit_cast(int, r);
  raw[i] = u;
  f32x2 f = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, false);
  back[2 * i] = f[0]; back[2 * i + 1] = f[1];
}
__device__ long to_f8(s16x8 v, float scale) {
  s16x2 lo = {0, 0}, hi = {0, 0};
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, (s16x2){v[0], v[1]}), scale, false);
  lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2, (s16x2){v[2], v[3]}), scale, true);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, (s16x2){v[4], v[5]}), scale, false);
  hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2, (s16x2){v[6], v[7]}), scale, true);
  return (long)(unsigned long)(unsigned)__builtin_bit_cast(int, lo) | ((long)__builtin_bit_cast(int, hi) << 32);
}
// one wave: A[16][32], B[16][32] (bf16, row-major) -> D = A B^T with the bf16 MFMA and with the fp8 MFMA on converted fragments
__global__ void mfma_probe(const __bf16* A, const __bf16* B, float* d16, float* d8, float sa, float sb) {
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  s16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = __builtin_bit_cast(short, A[li * 32 + 8 * g + e]);
    b[e] = __builtin_bit_cast(short, B[li * 32 + 8 * g + e]);
  }
  f32x4 z = {0, 0, 0, 0};
  f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a),
                                                      __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b), z, 0, 0, 0);
  f32x4 r8 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(to_f8(a, sa), to_f8(b, sb), z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    d16[(4 * g + r) * 16 + li] = r16[r];
    d8[(4 * g + r) * 16 + li] = r8[r] * sa * sb;
  }
}
int main() {
  const int n = 8;
  float h[16] = {1.f, 2.f, 0.3f, -0.7f, 448.f, 500.f, 1000.f, -3000.f, 0.001f, 0.002f, 0.0156f, 0.01f, 3.3f, 100.f, 240.f, 17.f};
  float *din, *dback; unsigned* draw;
  hipMalloc(&din, 64); hipMalloc(&dback, 64); hipMalloc(&draw, 32);
  hipMemcpy(din, h, 64, hipMemcpyHostToDevice);
  for (float sc : {1.f, 4.f, 0.25f, 3.f, -1.f}) {
    const int ovfl = sc < 0;
    if (ovfl) sc = 1.f;
    cvt_probe<<<1, 64>>>(din, sc, draw, dback, n, ovfl);
    float hb[16]; unsigned hr[8];
    hipMemcpy(hb, dback, 64, hipMemcpyDeviceToHost); hipMemcpy(hr, draw, 32, hipMemcpyDeviceToHost);
    printf("scale %g%s:\n", sc, ovfl ? " with MODE.FP16_OVFL" : "");
    for (int i = 0; i < n; ++i) printf("  in %10.4f %10.4f -> raw %08x -> %10.4f %10.4f\n", h[2 * i], h[2 * i + 1], hr[i], hb[2 * i], hb[2 * i + 1]);
  }
  __bf16 hA[512], hB[512];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = (__bf16)((rand() % 2001 - 1000) / 250.f); hB[i] = (__bf16)((rand() % 2001 - 1000) / 2000.f); }
  __bf16 *dA, *dB; float *d16, *d8;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&d16, 1024); hipMalloc(&d8, 1024);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  mfma_probe<<<1, 64>>>(dA, dB, d16, d8, 1.f / 64, 1.f / 512);
  float r16[256], r8[256];
  hipMemcpy(r16, d16, 1024, hipMemcpyDeviceToHost); hipMemcpy(r8, d8, 1024, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int i = 0; i < 256; ++i) { num += (r16[i] - r8[i]) * (r16[i] - r8[i]); den += r16[i] * r16[i]; }
  printf("mfma fp8 vs bf16: rel rms %.4f   (first row: %f %f | %f %f)\n", sqrt(num / den), r16[0], r8[0], r16[1], r8[1]);
  return 0;
}
