// Device helpers shared by the fused block kernels (fused_gdfn.hip, fused_mdta.hip): MFMA operand plumbing on gfx950.
#pragma once
#include "common.h"

namespace mi {
namespace fz {

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, delivered column-major.
// Lane 4q+p of the group supplies the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q.
// (EXEC must be all ones; address 8-byte aligned.)  Results are inline-asm outputs: tie them through fz_lds_wait*.
__device__ __forceinline__ s16x4 tr_b16(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(uintptr_t)p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait(s16x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void lds_wait(s16x4& a, s16x4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}
__device__ __forceinline__ void lds_wait(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
__device__ __forceinline__ void lds_wait(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : : "memory");
}
__device__ __forceinline__ void lds_wait(s16x4& a, s16x4& b, s16x4& c, s16x4& d, s16x4& e, s16x4& f, s16x4& g, s16x4& h) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory");
}
__device__ __forceinline__ s16x8 cat8(s16x4 lo, s16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// LDS operations of one wave execute in issue order; this only keeps the compiler from reordering across the point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float bf_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf_s(short s) { return __uint_as_float(((unsigned int)(unsigned short)s) << 16); }
__device__ __forceinline__ unsigned int pack_bf2(float a, float b) {
  const bf16 x = (bf16)a, y = (bf16)b;
  return (unsigned int)__builtin_bit_cast(u16, x) | ((unsigned int)__builtin_bit_cast(u16, y) << 16);
}
// two fp32 -> one dword of two bf16 in ONE v_cvt_pk_bf16_f32 (pack_bf2's two scalar casts compile to a convert + an SDWA or)
__device__ __forceinline__ unsigned int pk_bf2(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ short bf_bits(float a) { return (short)__builtin_bit_cast(u16, (bf16)a); }

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_prev_lane(float v) { return dpp_mov<0x138>(v); }  // wave_shr:1  lane i <- lane i-1
__device__ __forceinline__ float from_next_lane(float v) { return dpp_mov<0x130>(v); }  // wave_shl:1  lane i <- lane i+1

__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

}  // namespace fz
}  // namespace mi
