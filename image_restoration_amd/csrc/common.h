// Shared device/host helpers for the gfx950 Restormer-block kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi_restore.h"

namespace mi {

typedef __bf16 bf16;
typedef unsigned short u16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// MFMA operand policy of the bf16 kernels that can also run on fp8 operands (pw_gemm wave forms, fused GDFN).
// bf16: fragments go to v_mfma_f32_16x16x32_bf16 as they are.  fp8 (mi_pw_desc.f8, mi_gdfn_fused_fwd_f8):
// the SAME fragments (8 bf16 along k per lane) are divided by a power-of-two scale and rounded to OCP e4m3 in registers - four
// v_cvt_scalef32_pk_fp8_bf16 per fragment, element e to byte e for A and B alike, so the k-slot order carries over - and go to
// v_mfma_f32_16x16x32_fp8_fp8.  X is converted once per tile, W once per use (it stays bf16 in LDS).  The conversion returns NaN
// past +-448 unless MODE.FP16_OVFL is set (measured: tools/microbench/f8_probe.hip): the fp8 kernels set it, so a scale that is
// too small saturates instead of poisoning the image.
template <bool F8> struct MfmaOp;
template <> struct MfmaOp<false> {
  using Frag = s16x8;
  static __device__ __forceinline__ void enter() {}
  static __device__ __forceinline__ Frag cvt(const s16x8 v, float) { return v; }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaOp<true> {
  using Frag = long;
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ void enter() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }   // MODE.FP16_OVFL = 1
  static __device__ __forceinline__ Frag cvt(const s16x8 v, float scale) {
    s16x2_t lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2_t, (s16x2_t){v[0], v[1]}), scale, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_bit_cast(bf16x2_t, (s16x2_t){v[2], v[3]}), scale, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2_t, (s16x2_t){v[4], v[5]}), scale, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_bit_cast(bf16x2_t, (s16x2_t){v[6], v[7]}), scale, true);
    return (long)(unsigned long)(unsigned)__builtin_bit_cast(int, lo) | ((long)__builtin_bit_cast(int, hi) << 32);
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0); }
};

// ---- A/B switches (environment variables) ---------------------------------------------------------------------------------
// Every MI_* switch the launch planners consult is read ONCE, into a table, the first time the library needs one (round-2
// verdict: 27 getenv() calls per launch on a launch-bound path, not thread-safe against setenv from another thread).  Values
// are copied, so the pointers stay valid whatever happens to the environment.  mi_env_reload() (C-ABI) re-reads them - for
// tests and A/B tools that flip a switch inside one process; call it with no launch planner running on another thread.
#define MI_ENV_LIST(X)                                                                                                       \
  X(MI_BT_WIDE) X(MI_BT_DEBUG) X(MI_DW_LDS) X(MI_FG_CFG) X(MI_FG_DEBUG) X(MI_FG_NOXCD) X(MI_GRAM_FOLD) X(MI_GRAM_WANT)      \
  X(MI_GRAM_RECT) X(MI_GRAM_LDS) X(MI_GRAM_STREAM_ALL) X(MI_LN_FORM) X(MI_CO_STREAM) X(MI_ATTN_DQK_SPLIT)    \
  X(MI_PW_DMA) X(MI_PW_TM_EVEN) X(MI_PW_WAVE) X(MI_PW_CHUNKED) X(MI_PW_XWIDE) X(MI_PW_WAVE_WIDE) X(MI_PW_DIRECT)             \
  X(MI_PW_WAVE_TPW) X(MI_PW_XCD) X(MI_PW_TPB) X(MI_PW_B16) X(MI_FM_DEBUG) X(MI_FM_CFG) X(MI_NO_FUSED_MDTA) X(MI_NO_PW_LDS) X(MI_PW_LDS)
enum EnvId {
#define MI_ENV_ENUM(n) E_##n,
  MI_ENV_LIST(MI_ENV_ENUM)
#undef MI_ENV_ENUM
  E_ENV_COUNT
};
const char* env_get(int id);   // the variable's value as the process had it at load / last mi_env_reload(), or nullptr
#define MI_ENV(n) ::mi::env_get(::mi::E_##n)

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
#define MI_CHECK_ARG(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      mi::set_error(__VA_ARGS__);              \
      return MI_ERR_ARG;                       \
    }                                          \
  } while (0)
#define MI_CHECK_HIP(expr)                                                          \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      mi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return MI_ERR_HIP;                                                            \
    }                                                                               \
  } while (0)
#define MI_LAUNCH_CHECK() MI_CHECK_HIP(hipGetLastError())
#define MI_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != MI_OK) return rc_; \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline size_t dtype_size(int dt) { return dt == MI_BF16 ? 2 : 4; }

// Bump allocator over a caller-provided blob (saved-for-backward / workspace carving).
struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
  template <typename U = void>
  U* take(size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return base ? reinterpret_cast<U*>(base + o) : nullptr;
  }
};

// ---- device helpers ---------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned int bits16) { return __uint_as_float(bits16 << 16); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float from(float v) { return v; }
};
template <> struct Cvt<bf16> {
  static __device__ __forceinline__ bf16 from(float v) { return (bf16)v; }  // RNE, NaN-preserving (v_cvt_pk_bf16_f32)
};

// two fp32 -> one dword holding two bf16 (RNE, NaN-preserving) in ONE v_cvt_pk_bf16_f32: two scalar casts + shift/or compile
// to a convert plus an SDWA or per pair, twice the vector instructions in every bf16 epilogue
__device__ __forceinline__ unsigned int cvt_pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { *p = Cvt<T>::from(v); }

// Cache policy of the once-read plane accesses (the Vec loads below, the row loads of dwstream.hip, the X / residual loads of
// pw_gemm.hip).  Non-temporal LOADS: measured per kernel at bs 32 (profiles/r04_d_nontemporal_ab.txt): dwconv_gate_fwd -7.6 %,
// pw_gemm -1.4 %, LayerNorm -4..-9 %, mdta_av -6 %; the kernels whose operands other workgroups re-read (gram +12 %, bwd_tail
// +20 %) keep the default policy and do not use these macros.  Non-temporal STORES measured neutral (-DMI_NT_ST for the A/B).
#ifndef MI_NT_ST
#define MI_NT_ST 1
#endif
#if MI_NT_ST
#define MI_STREAM_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define MI_STREAM_ST(ptr, val) (*(ptr) = (val))
#endif
#ifndef MI_NT_LD
#define MI_NT_LD 1
#endif
#if MI_NT_LD
#define MI_STREAM_LD(ptr) __builtin_nontemporal_load(ptr)
#define MI_STREAM_DMA_AUX 2
#else
#define MI_STREAM_LD(ptr) (*(ptr))
#define MI_STREAM_DMA_AUX 0
#endif

// Vector access of V consecutive elements (V*sizeof(T) in {4,8,16} bytes, pointer aligned to it).
template <typename T, int V> struct Vec;
template <> struct Vec<float, 1> {
  static __device__ __forceinline__ void ld(const float* p, float* o) { o[0] = p[0]; }
  static __device__ __forceinline__ void st(float* p, const float* v) { p[0] = v[0]; }
};
template <> struct Vec<float, 2> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    f32x2 t = *reinterpret_cast<const f32x2*>(p); o[0] = t[0]; o[1] = t[1];
  }
  static __device__ __forceinline__ void st(float* p, const float* v) {
    f32x2 t = {v[0], v[1]}; *reinterpret_cast<f32x2*>(p) = t;
  }
};
template <> struct Vec<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float* o) {
    f32x4 t = MI_STREAM_LD(reinterpret_cast<const f32x4*>(p)); o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  }
  static __device__ __forceinline__ void st(float* p, const float* v) {
    f32x4 t = {v[0], v[1], v[2], v[3]}; MI_STREAM_ST(reinterpret_cast<f32x4*>(p), t);
  }
};
template <> struct Vec<bf16, 1> {
  static __device__ __forceinline__ void ld(const bf16* p, float* o) { o[0] = (float)p[0]; }
  static __device__ __forceinline__ void st(bf16* p, const float* v) { p[0] = (bf16)v[0]; }
};
template <> struct Vec<bf16, 2> {
  static __device__ __forceinline__ void ld(const bf16* p, float* o) {
    unsigned int t = *reinterpret_cast<const unsigned int*>(p);
    o[0] = bf16_bits_to_f32(t & 0xffffu); o[1] = bf16_bits_to_f32(t >> 16);
  }
  static __device__ __forceinline__ void st(bf16* p, const float* v) { *reinterpret_cast<unsigned int*>(p) = cvt_pk_bf16(v[0], v[1]); }
};
template <> struct Vec<bf16, 4> {
  static __device__ __forceinline__ void ld(const bf16* p, float* o) {
    u32x2 t = MI_STREAM_LD(reinterpret_cast<const u32x2*>(p));
    o[0] = bf16_bits_to_f32(t[0] & 0xffffu); o[1] = bf16_bits_to_f32(t[0] >> 16);
    o[2] = bf16_bits_to_f32(t[1] & 0xffffu); o[3] = bf16_bits_to_f32(t[1] >> 16);
  }
  static __device__ __forceinline__ void st(bf16* p, const float* v) {
    u32x2 t;
#pragma unroll
    for (int i = 0; i < 2; ++i) t[i] = cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    MI_STREAM_ST(reinterpret_cast<u32x2*>(p), t);
  }
};
template <> struct Vec<bf16, 8> {
  static __device__ __forceinline__ void ld(const bf16* p, float* o) {
    u32x4 t = MI_STREAM_LD(reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = bf16_bits_to_f32(t[i] & 0xffffu); o[2 * i + 1] = bf16_bits_to_f32(t[i] >> 16); }
  }
  static __device__ __forceinline__ void st(bf16* p, const float* v) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    MI_STREAM_ST(reinterpret_cast<u32x4*>(p), t);
  }
};

// Wave-wide sum on the VALU (DPP row shifts + row broadcasts; no LDS-pipe ds_bpermute): lane 63 ends up with the
// total, which is returned wave-uniform.
__device__ __forceinline__ float wave_sum(float v) {
#define MI_DPP_ADD(ctrl, rmask)                                                                                   \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  MI_DPP_ADD(0x111, 0xf);  // row_shr:1
  MI_DPP_ADD(0x112, 0xf);  // row_shr:2
  MI_DPP_ADD(0x114, 0xf);  // row_shr:4
  MI_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of every 16-lane row holds the row sum
  MI_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  MI_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
#undef MI_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {   // same DPP ladder as wave_sum (no LDS-pipe ds_bpermute round trips)
#define MI_DPP_MAX(ctrl, rmask)                                                                                    \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                     ctrl, rmask, 0xf, false)))
  MI_DPP_MAX(0x111, 0xf);
  MI_DPP_MAX(0x112, 0xf);
  MI_DPP_MAX(0x114, 0xf);
  MI_DPP_MAX(0x118, 0xf);
  MI_DPP_MAX(0x142, 0xa);
  MI_DPP_MAX(0x143, 0xc);
#undef MI_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// erf-form GELU and its derivative (Restormer.py:91: F.gelu default = erf form).
// erf is evaluated with Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 round-off level): one reciprocal,
// one exp and five FMAs instead of the ~35-instruction branchy libm erff; exp(-x^2/2) is shared with the Gaussian
// density of the derivative.  gelu_parts returns cdf = Phi(x) and pdf = phi(x).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float q = 0.5f * poly * e;          // = 0.5 * erfc(|x|/sqrt2)
  cdf = x >= 0.f ? 1.0f - q : q;
  pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_erf(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return x * cdf;
}
// Forward gate of the bf16 kernels: x sigmoid(1.6 x (1 + 0.0435 x^2)) - the tanh form of GELU with its two constants refitted
// to the erf form (max |difference| 2.9e-4 over the reals, 1/27 of a bf16 step at 1): 5 vector instructions + exp2 + rcp
// instead of 14 + exp + rcp, in kernels that are bound by their vector instruction count.  The fp32 (parity) instantiations and
// every backward keep the erf form (gelu_erf / gelu_parts); MI_GELU_EXACT=1 at build time keeps it everywhere.
template <typename T> __device__ __forceinline__ float gelu_fwd(float x) { return gelu_erf(x); }
#ifndef MI_GELU_EXACT
template <> __device__ __forceinline__ float gelu_fwd<bf16>(float x) {
  const float u = x * (-2.3083120f - 0.1004116f * x * x);            // -2 log2(e) 0.8 (x + 0.0435 x^3)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
}
#endif
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float cdf, pdf;
  gelu_parts(x, cdf, pdf);
  return cdf + x * pdf;
}

#endif  // __HIPCC__

// ---- optional per-kernel HIP-event profiler (off by default; bench.py's roofline pass turns it on) ----
enum KernelId {
  K_LN_FWD = 0, K_LN_BWD, K_DW_FWD, K_DW_GATE_FWD, K_DW_BWD_DATA, K_DW_GATE_BWD_DATA, K_DW_WGRAD, K_PW_GEMM, K_GRAM,
  K_GRAM_REDUCE, K_ATTN_FOLD, K_ATTN_BWD_SMALL, K_REDUCE_ROWS, K_CHAN_SUM, K_ADAMW, K_CAST, K_L1, K_PW_PACK, K_GAP,
  K_IM2COL, K_COL2IM, K_GDFN_FUSED_FWD, K_GDFN_FUSED_BWD, K_MDTA_FUSED_A, K_FUSED_PACK, K_MOE_ROUTE, K_CIRCCONV, K_EWISE, K_CONV3X3, K_GRAM_QK, K_PW_AV, K_BWD_TAIL, K_BWD_TAIL_FIN, K_ADAIR, K_COUNT
};
// Brackets one kernel launch with two events on ITS stream and books its algorithmic bytes / flops.
struct ProfScope {
  hipStream_t st; int kid; bool on;
  ProfScope(hipStream_t stream, int kernel_id, double bytes, double flops);
  ~ProfScope();
};

// ---- deferred parameter-gradient reductions (util.hip; C-ABI mi_deferred_*) ------------------------------------------------
// Every weight gradient ends in a fixed-order sum of partial rows into the gradient buffer.  As separate launches those sums were
// ~500 kernels of 5-15 us per training step.  While a caller-lent arena is active (mi_deferred_begin) a producer that ACCUMULATES
// into its gradient buffer may put its partials into the arena (deferred_take) and record the sum as a job (deferred_reduce_rows)
// instead of launching it; mi_deferred_flush runs all recorded jobs in ONE table-driven launch (a fixed order per job:
// bitwise reproducible).  Nothing may read such a gradient before the flush - true for the trainer's main_grad buffers, which
// are read by the all-reduce / optimizer only; producers whose caller reads the result at once (accumulate == 0) never defer.
// Two recorded jobs may target the same gradient (micro-batches under no_sync(), a parameter used twice in one backward): a job
// whose output was already recorded joins the next GENERATION, and a flush runs one launch per generation, in order.
// While `st` is being captured into a HIP graph nothing is deferred (the flush stages its table from host memory).
float* deferred_take(size_t nfloats, hipStream_t st);   // arena space for partials, or nullptr (inactive / full / capturing)
bool deferred_owns(const void* p);
// true: recorded (part must come from deferred_take); false: not recorded, the caller launches the reduction itself
bool deferred_reduce_rows(const float* part, float* out, int64_t rows, int64_t cols, int64_t part_ld, int accumulate, float scale,
                          float* out2 = nullptr, int64_t split = 0);

// generic small kernels implemented in util.hip, used by several modules
constexpr int REDUCE_GROUPS = 32;
int launch_reduce_rows(const float* part, float* out, int64_t rows, int64_t cols, int64_t part_ld,
                       int accumulate, float scale, hipStream_t st, float* tmp = nullptr, float* out2 = nullptr,
                       int64_t split = 0);  // out2: columns [split, cols) are written to out2[0 .. cols-split)

}  // namespace mi
