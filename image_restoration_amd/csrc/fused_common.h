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
// f(integral_constant<int, I>) for I = 0 .. N-1, unrolled at compile time (register arrays indexed by I stay in registers)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// wait for four transposed reads with up to KEEP YOUNGER LDS reads left in flight (LDS operations of a wave complete in issue order)
template <int KEEP> __device__ __forceinline__ void lds_wait_keep(s16x4& a, s16x4& b, s16x4& c, s16x4& d) {
  static_assert(KEEP >= 0 && KEEP <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(KEEP) : "memory");
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
__device__ __forceinline__ unsigned int pack_bf2(float a, float b) { return cvt_pk_bf16(a, b); }
// two fp32 -> one dword of two bf16 in ONE v_cvt_pk_bf16_f32 (pack_bf2's two scalar casts compile to a convert + an SDWA or)
__device__ __forceinline__ unsigned int pk_bf2(float a, float b) { return cvt_pk_bf16(a, b); }
__device__ __forceinline__ short bf_bits(float a) { return (short)__builtin_bit_cast(u16, (bf16)a); }

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_prev_lane(float v) { return dpp_mov<0x138>(v); }  // wave_shr:1  lane i <- lane i-1
__device__ __forceinline__ float from_next_lane(float v) { return dpp_mov<0x130>(v); }  // wave_shl:1  lane i <- lane i+1

__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

}  // namespace fz

// ---- tile staging shared by the persistent fused kernels (fused_mdta.hip, fused_gdfn.hip): K supplies C, NT, HR, VPR, TW, BODY,
// HPXP, NE, NBV, NBN, NEN, PLANE
using namespace fz;
// The next tile's raw x (tile + halo) travels global -> registers while the current tile's v chunks are computed, and
// registers -> LDS once the current tile is done: the HBM latency of the staging hides behind compute.
template <typename K> struct FmStage { u32x4 raw[K::NBN]; u16 rawe[K::NEN]; };

template <typename K>
__device__ __forceinline__ void fm_stage_load(FmStage<K>& st, const bf16* xb, int tt, int x0, int y0, int H, int W, int64_t HW) {
#pragma unroll
  for (int n = 0; n < K::NBN; ++n) {
    const int idx = tt + K::NT * n;
    const int cc = idx / (K::HR * K::VPR), rem = idx - cc * (K::HR * K::VPR), r = rem / K::VPR, u = rem % K::VPR;
    const int Y = y0 - 1 + r;
    st.raw[n] = (u32x4){0u, 0u, 0u, 0u};
    if (idx < K::NBV && Y >= 0 && Y < H)
      st.raw[n] = *reinterpret_cast<const u32x4*>(xb + (int64_t)cc * HW + (int64_t)Y * W + x0 + 8 * u);
  }
#pragma unroll
  for (int n = 0; n < K::NEN; ++n) {
    const int idx = tt + K::NT * n;
    const int cc = idx / K::NE, k = idx - cc * K::NE;
    st.rawe[n] = 0;
    if (idx < K::C * K::NE && k < 2 * K::HR) {
      const int side = k >= K::HR ? 1 : 0, r = k - side * K::HR;
      const int Y = y0 - 1 + r, X = side ? x0 + K::TW : x0 - 1;
      if (Y >= 0 && Y < H && X >= 0 && X < W)
        st.rawe[n] = reinterpret_cast<const u16*>(xb)[(int64_t)cc * HW + (int64_t)Y * W + X];
    }
  }
}
template <typename K>
__device__ __forceinline__ void fm_stage_store(const FmStage<K>& st, bf16* S, int tt) {
#pragma unroll
  for (int n = 0; n < K::NBN; ++n) {
    const int idx = tt + K::NT * n;
    const int cc = idx / (K::HR * K::VPR), rem = idx - cc * (K::HR * K::VPR), r = rem / K::VPR, u = rem % K::VPR;
    if (idx < K::NBV) *reinterpret_cast<u32x4*>(&S[cc * K::PLANE + r * K::TW + 8 * u]) = st.raw[n];
  }
#pragma unroll
  for (int n = 0; n < K::NEN; ++n) {
    const int idx = tt + K::NT * n;
    const int cc = idx / K::NE, k = idx - cc * K::NE;
    if (idx < K::C * K::NE) reinterpret_cast<u16*>(S)[cc * K::PLANE + K::BODY + k] = st.rawe[n];
  }
}

// Diagnostic build path (MI_FM_DEBUG & 0x1000): shader-clock stamps around the phases, summed per wave and written as floats to
// the `mean` buffer ([workgroup][wave][8]: stage + barrier, LayerNorm, GEMM1, barrier wait, Gram, conv, whole kernel, tiles).
// Stamps serialise what the real kernel overlaps: read the SHARES, never the run time of such a run.
__device__ __forceinline__ unsigned long long fm_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

// byte offset of 4-channel group `g4` (0..3) of pixel record r: the two 16-byte halves of a record are swapped on every other
// group of four records, which takes the GEMM1 stores (16 lanes = 16 consecutive records, 8 bytes each) from 4-way to 2-way bank
// conflicts and leaves the conv's 16-byte operand reads conflict-free
__device__ __forceinline__ int fm4_rec(int r, int g4) { return r * 32 + ((((g4 >> 1) ^ (r >> 2)) & 1) << 4) + ((g4 & 1) << 3); }

}  // namespace mi
