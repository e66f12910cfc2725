// Pointwise (1x1-conv) GEMM on N-contiguous NCHW planes, MFMA tiles for gfx950.
//   Y[z][m][n] = sum_k W_z(m,k) X[z][k][n] (+bias[m]) (+R[z][m][n])
// Orientation: the MFMA "A" operand is X^T (rows = pixels), "B" is W^T (cols = output channels), so the
// accumulator holds 4 consecutive pixels of one output channel per lane.  X is staged [k][n] exactly as it lies in
// HBM (coalesced 16-byte loads) and transposed for free on the LDS read (ds_read_b64_tr_b16).  Weights are first
// re-packed by a tiny kernel into the exact LDS image the main kernel wants ([m-tile][k-chunk][TM][row stride],
// activation dtype, zero padded, transposed if the caller passed W^T), so staging a weight chunk is a straight
// 16-byte copy (measured: per-tile scalar fp32 staging cost more issue slots than everything else in the kernel).
// The epilogue stages accumulators through LDS and writes whole 128-byte row segments, 16 bytes per lane.
// bf16 activations use v_mfma_f32_16x16x32_bf16, fp32 activations the exact v_mfma_f32_16x16x4_f32.
// Kernel families, in the order pw_launch tries them:
//   pw_gemm_wave_xres / pw_gemm_wave_stream  bf16, rows of whole 64-pixel tiles: weights resident in LDS, every wave streams its own
//                                            pixel tiles through a private transpose patch - no workgroup barriers (the default)
//   pw_gemm_res                              bf16, K <= 128: weight-resident persistent workgroups sharing each X chunk (MI_PW_WAVE=0)
//   pw_gemm                                  any dtype / alignment / ragged pixel counts: chunked through LDS (MI_PW_CHUNKED=1)
//   pw_gemm_dma                              opt-in LDS-DMA ring (MI_PW_DMA=1)
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "internal.h"

namespace mi {

constexpr int PW_TN = 64;  // pixels per workgroup tile
constexpr int PW_KC = 32;  // k per staged chunk
constexpr int PW_XS = 80;  // LDS row stride (elements) of the X chunk: conflict-free tr-reads / b32 reads

template <typename T> struct PwRow { static constexpr int WS_ROW = std::is_same<T, float>::value ? 34 : 40; };

__device__ __forceinline__ s16x4 lds_tr_b16(const void* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(uintptr_t)p;  // LDS aperture: low 32 bits are the LDS byte address
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// LDS operations of one wave execute in issue order; this only keeps the compiler from reordering across the point
// (wave-private LDS exchange without a workgroup barrier).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int TM> struct PwImg {
  static constexpr int WS_ROW = PwRow<T>::WS_ROW;
  static constexpr int W_BYTES = TM * WS_ROW * (int)sizeof(T);
  static constexpr int W_PAD = (W_BYTES + 4095) / 4096 * 4096;   // chunk image stride: whole KiB per wave for LDS-DMA
  static constexpr int W_PAD_ELEMS = W_PAD / (int)sizeof(T);
};
constexpr int PW_ZERO_BYTES = 256;  // zero block at the head of the workspace: LDS-DMA source for out-of-range rows/cols

// ---- weight re-pack: fp32 W(m,k) (any strides) -> T image [slice][m_tile][k_chunk][chunk stride] of [TM][WS_ROW] ----
struct PackJob {          // one weight matrix -> one packed image (device-visible: the batched refresh reads a table of these)
  const float* w; int64_t w_bs, w_gs, w_sm, w_sk;
  unsigned char* ws;      // [zero block][packed image]
  int M, K, tm, k_chunks, groups_w, chunk_elems, m_fast, dtype, slices;
  int64_t slice_elems;
};

template <typename T>
__device__ __forceinline__ void pw_pack_slice(const PackJob& j, int slice, int64_t first, int64_t stride) {
  constexpr int WS_ROW = PwRow<T>::WS_ROW;
  T* out = reinterpret_cast<T*>(j.ws + PW_ZERO_BYTES);
  const int sb = slice / j.groups_w, sg = slice - sb * j.groups_w;
  const float* wz = j.w + sb * j.w_bs + sg * j.w_gs;
  T* oz = out + (int64_t)slice * j.slice_elems;
  const int tm = j.tm, k_chunks = j.k_chunks;
  const int m_tiles = (int)(j.slice_elems / ((int64_t)k_chunks * j.chunk_elems));
  const int64_t total = (int64_t)m_tiles * k_chunks * tm * PW_KC;
  for (int64_t e = first; e < total; e += stride) {
    // m_fast: consecutive threads walk m (coalesced for W^T callers), else k
    int64_t r = e;
    int kk, mm;
    if (j.m_fast) { mm = (int)(r % tm); r /= tm; kk = (int)(r % PW_KC); r /= PW_KC; }
    else { kk = (int)(r % PW_KC); r /= PW_KC; mm = (int)(r % tm); r /= tm; }
    const int kc = (int)(r % k_chunks), mt = (int)(r / k_chunks);
    const int m = mt * tm + mm, k = kc * PW_KC + kk;
    const float v = (m < j.M && k < j.K) ? wz[(int64_t)m * j.w_sm + (int64_t)k * j.w_sk] : 0.f;
    oz[((int64_t)mt * k_chunks + kc) * j.chunk_elems + mm * WS_ROW + kk] = Cvt<T>::from(v);
  }
}

// ---- weight re-pack: fp32 W(m,k) (any strides) -> T image [slice][m_tile][k_chunk][chunk stride] of [TM][WS_ROW] ----
template <typename T>
__global__ __launch_bounds__(256) void pw_pack_kernel(PackJob j) {
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < PW_ZERO_BYTES / 4) reinterpret_cast<float*>(j.ws)[threadIdx.x] = 0.f;
  pw_pack_slice<T>(j, blockIdx.y, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}
// every cached weight of the model in ONE launch (mi_pw_cache_refresh): blockIdx.y = table entry
__global__ __launch_bounds__(256) void pw_pack_table_kernel(const PackJob* __restrict__ table) {
  const PackJob j = table[blockIdx.y];
  if (blockIdx.x == 0 && threadIdx.x < PW_ZERO_BYTES / 4) reinterpret_cast<float*>(j.ws)[threadIdx.x] = 0.f;
  for (int slice = 0; slice < j.slices; ++slice) {
    if (j.dtype == MI_F32) pw_pack_slice<float>(j, slice, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
    else pw_pack_slice<bf16>(j, slice, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
  }
}

struct PwG {
  PwK k;
  const unsigned char* ws;  // [zero block][packed weights]
  int64_t wp_slice;         // elements per packed slice
  int wp_per_batch, wp_per_group, k_chunks;
  // LayerNorm of X applied on the way in (xres form only): gamma / beta [K], statistics out [batch][n] (or null)
  const float* ln_w; const float* ln_b; float* ln_mean; float* ln_rstd; int ln_mode;   // 0 none, 1 WithBias, 2 BiasFree
  int xcd_map;              // stream form: XCD-aware workgroup order (see the kernel)
  int w_direct;             // wave forms: stage the weights straight from the fp32 matrix k.w (per-image weights: no pack launch)
  const bf16* wb16; int64_t wb16_sm;   // ... or from their bf16 copy (mi_pw_desc.w_b16): 16-byte copies, what the packed path's staging costs
  float f8_sx, f8_sw;       // fp8 operand forms: X / f8_sx and W / f8_sw are rounded to e4m3, the accumulator is scaled by their product
};

// X chunk addressing in LDS.  Register-staged image: [32][PW_XS] padded rows.  LDS-DMA image: [32][64] unpadded rows
// (a wave's 1 KiB DMA piece = whole rows) with the 16-byte pieces of row r XOR-permuted by ((r>>1)&3)<<1 so that the
// bf16 transposed reads stay bank-conflict free (the same permutation is applied to the DMA source address).
template <typename T, bool DMA> struct PwXAddr {
  static __device__ __forceinline__ int at(int r, int n) {
    if (!DMA) return r * PW_XS + n;
    if (sizeof(T) == 4) return r * PW_TN + n;
    return r * PW_TN + ((((n >> 3) ^ (((r >> 1) & 3) << 1))) << 3) + (n & 7);
  }
};

// Which 16-channel fragments of a TM-row tile a wave owns.  TM = 64 MF splits evenly (MF each); TM = 96 gives the four
// waves 2, 2, 1, 1 fragments and TM = 48 gives 1, 1, 1, 0: a 96- or 48-channel matrix then stages, multiplies and stores 25%
// less than when padded to 128 / 64 (these kernels are instruction-issue-bound, profiles/r01_x_pmc_pw_gemm_bs32.txt).
template <int MF, int TM> __device__ __forceinline__ void pw_wave_rows(int wv, int& mw, int& nfr) {
  if (TM == 64 * MF) { mw = wv * MF * 16; nfr = MF; }
  else if (TM == 96) { mw = wv < 2 ? wv * 32 : 64 + (wv - 2) * 16; nfr = wv < 2 ? 2 : 1; }
  else { mw = wv < 3 ? wv * 16 : 0; nfr = wv < 3 ? 1 : 0; }   // TM == 48 (the idle wave points at valid rows)
}

// one 32-deep chunk of MFMAs: acc[nf][mf] += X^T(pixels 16nf.., k) * W^T(k, channels mw+16mf..)
template <typename T, int MF, bool DMA, int XS = PW_XS>
__device__ __forceinline__ void pw_chunk_mma(const T* Xs, const T* Ws, f32x4 (*acc)[MF], int mw, int li, int g,
                                             int nfr = MF) {
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr int WS_ROW = PwRow<T>::WS_ROW;
  struct XA {
    static __device__ __forceinline__ int at(int r, int n) { return DMA ? PwXAddr<T, DMA>::at(r, n) : r * XS + n; }
  };
  if constexpr (F32) {
#pragma unroll
    for (int ks = 0; ks < PW_KC / 4; ++ks) {
      const int kk = 4 * ks + g;
      float a[4], b[MF];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) a[nf] = Xs[XA::at(kk, 16 * nf + li)];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) b[mf] = mf < nfr ? Ws[(mw + 16 * mf + li) * WS_ROW + kk] : 0.f;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          if (mf < nfr) acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nf], b[mf], acc[nf][mf], 0, 0, 0);
    }
  } else {
    // k slots: element j<4 of lane group g is k = 4g+j, element j>=4 is k = 16+4g+(j-4), for A and B alike.
    // The tr-reads are inline asm, so their results are tied through the s_waitcnt statement below:
    // every consumer is data-dependent on the wait and cannot be scheduled ahead of it.
    const int qq = li >> 2, pp = li & 3;
    s16x4 alo[4], ahi[4], blo[MF], bhi[MF];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      alo[nf] = lds_tr_b16(&Xs[XA::at(4 * g + qq, 16 * nf + 4 * pp)]);
      ahi[nf] = lds_tr_b16(&Xs[XA::at(16 + 4 * g + qq, 16 * nf + 4 * pp)]);
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const T* wr = &Ws[(mw + 16 * (mf < nfr ? mf : 0) + li) * WS_ROW + 4 * g];   // (an unowned fragment re-reads an owned row)
      blo[mf] = *reinterpret_cast<const s16x4*>(wr);
      bhi[mf] = *reinterpret_cast<const s16x4*>(wr + 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(alo[0]), "+v"(alo[1]), "+v"(alo[2]), "+v"(alo[3]), "+v"(ahi[0]), "+v"(ahi[1]), "+v"(ahi[2]),
                   "+v"(ahi[3])
                 :
                 : "memory");
    s16x8 a[4], b[MF];
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) a[nf] = __builtin_shufflevector(alo[nf], ahi[nf], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) b[mf] = __builtin_shufflevector(blo[mf], bhi[mf], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
        if (mf < nfr) acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nf], b[mf], acc[nf][mf], 0, 0, 0);
  }
}

// Default-cache-policy vector accesses for the weight-resident / chunked kernel forms below: they serve the deep levels (C >= 192:
// tensors of 100-330 MB at bs 32 that the producer has just left in the Infinity Cache and the next kernel reads at once), where the
// non-temporal policy of Vec<>::ld / ::st measured slower (C = 384 project_in 131 us against 102: profiles/r04_d_nontemporal_ab.txt).
template <typename T, int V> __device__ __forceinline__ void ld_cached(const T* p, float* o) {
  if constexpr (std::is_same<T, float>::value) {
    static_assert(V == 4, "fp32: 16-byte vectors");
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = t[i];
  } else {
    static_assert(V == 8, "bf16: 16-byte vectors");
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = bf16_bits_to_f32(t[i] & 0xffffu); o[2 * i + 1] = bf16_bits_to_f32(t[i] >> 16); }
  }
}
template <typename T, int V> __device__ __forceinline__ void st_cached(T* p, const float* v) {
  if constexpr (std::is_same<T, float>::value) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
  } else {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = cvt_pk_bf16(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<u32x4*>(p) = t;
  }
}

// Epilogue: lane holds pixels n0+16nf+4g+{0..3} of output channel m0+mw+16mf+li — 8 bytes per lane scattered over
// 16 rows.  The rows are staged through LDS (one 16-row fp32 slab per wave at a time, so the only rounding is the final
// store) and written as whole 128-byte row segments, 16 bytes per lane, with bias and residual added on the way.
template <typename T, int MF>
__device__ __forceinline__ void pw_epilogue(const PwK& p, float* lds_f32, const f32x4 (*acc)[MF], int zb, int zg, int m0,
                                            int64_t n0, int mw, int lane, int wv, int nfr = MF) {
  constexpr bool F32 = std::is_same<T, float>::value;
  const int li = lane & 15, g = lane >> 4;
  T* yz = (T*)p.y + zb * p.y_bs + zg * p.y_gs;
  const T* rz = p.r ? (const T*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  const float* bz = p.bias ? p.bias + zg * p.bias_gs : nullptr;
  constexpr int EPV = F32 ? 4 : 8;        // pixels per 16-byte vector of the output type
  constexpr int OS = PW_TN + 4;           // fp32 slab row stride (floats): 272 bytes
  constexpr int LPR = PW_TN / EPV;        // lanes per row on the way out
  constexpr int RPI = 64 / LPR;           // rows per store instruction
  float* slab = lds_f32 + wv * 16 * OS;
  __syncthreads();                        // every wave is done reading the last chunk
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const bool own = mf < nfr;            // wave-uniform; a wave without this fragment only keeps the barriers company
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      float o[4] = {acc[nf][mf][0], acc[nf][mf][1], acc[nf][mf][2], acc[nf][mf][3]};
      if (own) Vec<float, 4>::st(&slab[li * OS + 16 * nf + 4 * g], o);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16 / RPI; ++it) {
      const int row = it * RPI + lane / LPR, col = (lane % LPR) * EPV;
      const int m = m0 + mw + 16 * mf + row;
      const int64_t n = n0 + col;
      if (own && m < p.m && n < p.n) {
        float o[EPV];
#pragma unroll
        for (int v = 0; v < EPV / 4; ++v) Vec<float, 4>::ld(&slab[row * OS + col + 4 * v], o + 4 * v);
        const float bv = bz ? bz[m] : 0.f;
#pragma unroll
        for (int j = 0; j < EPV; ++j) o[j] += bv;
        const int64_t off = (int64_t)m * p.n + n;
        if (p.vec_ok) {
          if (rz) {
            float rr[EPV];
            ld_cached<T, EPV>(rz + off, rr);
#pragma unroll
            for (int j = 0; j < EPV; ++j) o[j] += rr[j];
          }
          st_cached<T, EPV>(yz + off, o);
        } else {
#pragma unroll
          for (int j = 0; j < EPV; ++j)
            if (n + j < p.n) st1(yz + off + j, o[j] + (rz ? ld1(rz + off + j) : 0.f));
        }
      }
    }
    __syncthreads();
  }
}

constexpr int PW_SLAB_BYTES = 4 * 16 * (PW_TN + 4) * (int)sizeof(float);  // epilogue: 4 waves x 16 rows x fp32

// ---- register-staged form: any alignment / ragged pixel counts; one chunk of prefetch ----
template <typename T, int MF, int TM = 64 * MF>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwG q) {
  const PwK& p = q.k;
  constexpr bool F32 = std::is_same<T, float>::value;
  using IM = PwImg<T, TM>;
  constexpr int X_ELEMS = PW_KC * PW_XS, W_ELEMS = TM * IM::WS_ROW;
  constexpr int STAGE_BYTES = (X_ELEMS + W_ELEMS) * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE_BYTES > PW_SLAB_BYTES ? STAGE_BYTES : PW_SLAB_BYTES];
  T* const Xs = reinterpret_cast<T*>(lds_raw);
  T* const Ws = Xs + X_ELEMS;

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z, zb = z / p.groups, zg = z - zb * p.groups;
  const int64_t n0 = (int64_t)blockIdx.x * PW_TN;
  const int m0 = blockIdx.y * TM;
  const int ktot = p.k1 + p.k2;
  const int nchunks = q.k_chunks;

  const T* x1 = (const T*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  const T* x2 = p.x2 ? (const T*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : nullptr;
  const int wslice = (q.wp_per_batch ? zb : 0) * (q.wp_per_group ? p.groups : 1) + (q.wp_per_group ? zg : 0);
  const T* wpk = reinterpret_cast<const T*>(q.ws + PW_ZERO_BYTES) + (int64_t)wslice * q.wp_slice +
                 (int64_t)blockIdx.y * nchunks * IM::W_PAD_ELEMS;

  constexpr int XV = F32 ? 2 : 1;                             // 16-byte vectors per thread for its 8 X elements
  constexpr int W_VECS = IM::W_BYTES / 16;                    // 16-byte vectors per weight chunk image
  constexpr int WV = (W_VECS + 255) / 256;
  u32x4 xreg[XV], wreg[WV];
  const int xr_row = t >> 3, xr_col = (t & 7) * 8;

  auto load_stage = [&](int chunk) {
    const int k = chunk * PW_KC + xr_row;
    const T* row = nullptr;
    if (k < p.k1) row = x1 + (int64_t)k * p.n;
    else if (k < ktot) row = x2 + (int64_t)(k - p.k1) * p.n;
    const int64_t n = n0 + xr_col;
    constexpr int EPVX = 8 / XV;  // elements per 16-byte vector
#pragma unroll
    for (int v = 0; v < XV; ++v) {
      const int64_t nn = n + v * EPVX;
      if (row && p.vec_ok && nn < p.n) {  // vec_ok: n % EPVX == 0 for every row, so the vector is all-in or all-out
        xreg[v] = *reinterpret_cast<const u32x4*>(row + nn);
      } else {
        __attribute__((aligned(16))) T tmp[EPVX];
#pragma unroll
        for (int j = 0; j < EPVX; ++j) tmp[j] = (row && nn + j < p.n) ? row[nn + j] : Cvt<T>::from(0.f);
        xreg[v] = *reinterpret_cast<u32x4*>(tmp);
      }
    }
    const u32x4* wc = reinterpret_cast<const u32x4*>(wpk + (int64_t)chunk * IM::W_PAD_ELEMS);
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int vid = t + 256 * i;
      if (vid < W_VECS) wreg[i] = wc[vid];
    }
  };
  auto write_stage = [&]() {
#pragma unroll
    for (int v = 0; v < XV; ++v)
      *reinterpret_cast<u32x4*>(&Xs[xr_row * PW_XS + xr_col + v * (8 / XV)]) = xreg[v];
    u32x4* wd = reinterpret_cast<u32x4*>(Ws);
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int vid = t + 256 * i;
      if (vid < W_VECS) wd[vid] = wreg[i];
    }
  };

  f32x4 acc[4][MF];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int mw, nfr;                  // this wave's first m inside the tile, and how many 16-row fragments it owns
  pw_wave_rows<MF, TM>(wv, mw, nfr);

  load_stage(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    write_stage();
    __syncthreads();
    if (c + 1 < nchunks) load_stage(c + 1);
    pw_chunk_mma<T, MF, false>(Xs, Ws, acc, mw, li, g, nfr);
  }
  pw_epilogue<T, MF>(p, reinterpret_cast<float*>(lds_raw), acc, zb, zg, m0, n0, mw, lane, wv, nfr);
}

// ---- weight-resident form (bf16, K <= 128, vector-aligned) ------------------------------------------------------------------
// SQ counters on the chunked kernel above (profiles/r01_x_pmc_pw_gemm_bs32.txt): waves live ~370 instructions, parked on
// s_waitcnt / s_barrier 62% of the time; VALU 8%, LDS 5%.  A tile there is a chain of dependent round trips - X chunk,
// next X chunk, ..., residual - behind ~13 workgroup barriers, and every tile re-stages the weight chunks (10 KB each,
// more bytes than the X chunk they multiply).  Here a workgroup keeps the whole (m-tile, z) weight image in LDS and walks
// pixel tiles: ALL K chunks of the next tile's X and the residual of the current one are loaded into registers while the
// current tile is multiplied, so a tile costs one exposed round trip at most, three barriers, and no weight traffic.
constexpr int PWR_MAXC = 4;                                       // K chunks held (K <= 128)
template <int MF, int TM = 64 * MF>
__global__ __launch_bounds__(256) void pw_gemm_res_kernel(PwG q, int n_tiles, int chunk_stride_elems) {
  using T = bf16;
  const PwK& p = q.k;
  constexpr int WS_ROW = PwRow<T>::WS_ROW;
  constexpr int WC_ELEMS = TM * WS_ROW;                           // one weight chunk in LDS (packed back to back)
  constexpr int XC_ELEMS = PW_KC * PW_XS;                         // one X chunk
  constexpr int OS = PW_TN + 4;
  constexpr int SLAB_ELEMS = 4 * 16 * OS * 2;                     // fp32 slabs of the four waves, in T units
  constexpr int XBUF_ELEMS = PWR_MAXC * XC_ELEMS > SLAB_ELEMS ? PWR_MAXC * XC_ELEMS : SLAB_ELEMS;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  const int nchunks = q.k_chunks;
  T* const Wl = reinterpret_cast<T*>(lds_dyn);                    // [nchunks][TM][WS_ROW]
  T* const Xl = Wl + nchunks * WC_ELEMS;                          // [nchunks][32][PW_XS]; the epilogue slabs alias it

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z, zb = z / p.groups, zg = z - zb * p.groups;
  const int m0 = blockIdx.y * TM;
  int mw, nfr;
  pw_wave_rows<MF, TM>(wv, mw, nfr);
  const int ktot = p.k1 + p.k2;
  {  // weight image: straight 16-byte copies from the packed workspace / cache
    const int wslice = (q.wp_per_batch ? zb : 0) * (q.wp_per_group ? p.groups : 1) + (q.wp_per_group ? zg : 0);
    const T* wpk = reinterpret_cast<const T*>(q.ws + PW_ZERO_BYTES) + (int64_t)wslice * q.wp_slice +
                   (int64_t)blockIdx.y * nchunks * chunk_stride_elems;
    constexpr int VPC = WC_ELEMS / 8;
    for (int v = t; v < nchunks * VPC; v += 256) {
      const int c = v / VPC, o = v - c * VPC;
      reinterpret_cast<u32x4*>(Wl)[v] = reinterpret_cast<const u32x4*>(wpk + (int64_t)c * chunk_stride_elems)[o];
    }
  }
  const T* x1 = (const T*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  const T* x2 = p.x2 ? (const T*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : nullptr;
  T* yz = (T*)p.y + zb * p.y_bs + zg * p.y_gs;
  const T* rz = p.r ? (const T*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  const float* bz = p.bias ? p.bias + zg * p.bias_gs : nullptr;

  const int xr_row = t >> 3, xr_col = (t & 7) * 8;                // this thread's 16-byte piece of every X chunk
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto load_x = [&](int tile, u32x4* r) {
    const int64_t n = (int64_t)tile * PW_TN + xr_col;
    const bool in = tile < n_tiles && n < p.n;
#pragma unroll
    for (int c = 0; c < PWR_MAXC; ++c) {
      const int k = c * PW_KC + xr_row;
      const T* row = k < p.k1 ? x1 + (int64_t)k * p.n : x2 + (int64_t)(k - p.k1) * p.n;
      r[c] = (in && c < nchunks && k < ktot) ? MI_STREAM_LD(reinterpret_cast<const u32x4*>(row + n)) : zero4;
    }
  };
  // epilogue geometry: lane reads row (it*8 + lane/8) of its wave's 16-row slab, 8 pixels at (lane%8)*8
  const int e_row = lane >> 3, e_col = (lane & 7) * 8;
  auto load_res = [&](int tile, u32x4 (*r)[2]) {
    const int64_t n = (int64_t)tile * PW_TN + e_col;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int m = m0 + mw + 16 * mf + it * 8 + e_row;
        r[mf][it] = (rz && mf < nfr && tile < n_tiles && m < p.m && n < p.n) ? MI_STREAM_LD(reinterpret_cast<const u32x4*>(rz + (int64_t)m * p.n + n)) : zero4;
      }
  };

  u32x4 xr[PWR_MAXC], rr[MF][2];
  load_x(blockIdx.x, xr);
  float* slab = reinterpret_cast<float*>(Xl) + wv * 16 * OS;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
    for (int c = 0; c < PWR_MAXC; ++c)
      if (c < nchunks) *reinterpret_cast<u32x4*>(&Xl[c * XC_ELEMS + xr_row * PW_XS + xr_col]) = xr[c];
    __syncthreads();                                   // X of this tile (and, the first time, the weights) are in place
    load_res(tile, rr);                                // in flight during the multiply
    load_x(tile + gridDim.x, xr);                      // the next tile's whole K
    f32x4 acc[4][MF];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < MF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunks; ++c) pw_chunk_mma<T, MF, false>(Xl + c * XC_ELEMS, Wl + c * WC_ELEMS, acc, mw, li, g, nfr);
    __syncthreads();                                   // everyone is done reading X: the slabs may overwrite it
    const int64_t n0 = (int64_t)tile * PW_TN;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      if (mf >= nfr) continue;                         // wave-uniform: this wave does not own a second fragment
      wave_lds_sync();
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        float o[4] = {acc[nf][mf][0], acc[nf][mf][1], acc[nf][mf][2], acc[nf][mf][3]};
        Vec<float, 4>::st(&slab[li * OS + 16 * nf + 4 * g], o);
      }
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + e_row;
        const int m = m0 + mw + 16 * mf + row;
        const int64_t n = n0 + e_col;
        if (m < p.m && n < p.n) {
          float o[8], res[8];
          Vec<float, 4>::ld(&slab[row * OS + e_col], o);
          Vec<float, 4>::ld(&slab[row * OS + e_col + 4], o + 4);
          const float bv = bz ? bz[m] : 0.f;
          const u32x4 rv = rr[mf][it];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            res[2 * j] = bf16_bits_to_f32(rv[j] & 0xffffu);
            res[2 * j + 1] = bf16_bits_to_f32(rv[j] >> 16);
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += bv + res[j];   // zeros when there is no residual
          Vec<T, 8>::st(yz + (int64_t)m * p.n + n, o);
        }
      }
    }
    __syncthreads();                                   // slabs are dead: the next tile's X may be written
  }
}

// ---- LDS-DMA form (16-byte aligned rows): global_load_lds_dwordx4 straight into a 3-deep ring of chunk images, two
// chunks in flight behind a counted s_waitcnt vmcnt and ONE raw s_barrier per chunk; no staging registers, no ds_write.
template <typename T, int MF>
__global__ __launch_bounds__(256) void pw_gemm_dma_kernel(PwG q) {
  const PwK& p = q.k;
  constexpr bool F32 = std::is_same<T, float>::value;
  constexpr int TM = 64 * MF;
  using IM = PwImg<T, TM>;
  constexpr int X_BYTES = PW_KC * PW_TN * (int)sizeof(T);      // 4 KiB (bf16) / 8 KiB (fp32): whole KiB per wave
  constexpr int NX = X_BYTES / 4096;                           // X DMA instructions per wave per chunk
  constexpr int NW = IM::W_PAD / 4096;                         // W DMA instructions per wave per chunk
  constexpr int NI = NX + NW;
  constexpr int BUF_BYTES = X_BYTES + IM::W_PAD;
  constexpr int NBUF = 3;
  static_assert(NBUF * BUF_BYTES >= PW_SLAB_BYTES, "ring must cover the epilogue slabs");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int z = blockIdx.z, zb = z / p.groups, zg = z - zb * p.groups;
  const int64_t n0 = (int64_t)blockIdx.x * PW_TN;
  const int m0 = blockIdx.y * TM;
  const int ktot = p.k1 + p.k2;
  const int nchunks = q.k_chunks;

  const T* x1 = (const T*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  const T* x2 = p.x2 ? (const T*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : nullptr;
  const int wslice = (q.wp_per_batch ? zb : 0) * (q.wp_per_group ? p.groups : 1) + (q.wp_per_group ? zg : 0);
  const unsigned char* wpk = q.ws + PW_ZERO_BYTES +
                             ((int64_t)wslice * q.wp_slice + (int64_t)blockIdx.y * nchunks * IM::W_PAD_ELEMS) * sizeof(T);
  const unsigned char* zero_src = q.ws;  // 256 zero bytes

  // X piece geometry of this lane: rows of (64 * sizeof(T)) bytes, ROWS_PI rows per 1 KiB piece
  constexpr int ROW_BYTES = PW_TN * (int)sizeof(T);
  constexpr int ROWS_PI = 1024 / ROW_BYTES;        // 8 (bf16) / 4 (fp32)
  constexpr int PCS = ROW_BYTES / 16;              // 16-byte pieces per row: 8 / 16
  const int xrow_in_piece = lane / PCS, xpc = lane % PCS;

  auto issue = [&](int chunk) {
    unsigned char* buf = lds_dyn + (chunk % NBUF) * BUF_BYTES;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int piece = wv * NX + j;                         // 1 KiB piece index inside the X image
      const int r = piece * ROWS_PI + xrow_in_piece;         // row of the chunk (0..31)
      const int k = chunk * PW_KC + r;
      const int src_pc = F32 ? xpc : (xpc ^ (((r >> 1) & 3) << 1));  // source-side swizzle (bf16)
      const int64_t n = n0 + (int64_t)src_pc * (16 / (int)sizeof(T));
      const unsigned char* src = zero_src + (lane & 15) * 16;
      if (n < p.n) {
        if (k < p.k1) src = reinterpret_cast<const unsigned char*>(x1 + (int64_t)k * p.n + n);
        else if (k < ktot) src = reinterpret_cast<const unsigned char*>(x2 + (int64_t)(k - p.k1) * p.n + n);
      }
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, MI_STREAM_DMA_AUX);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int piece = wv * NW + j;
      const unsigned char* src = wpk + (int64_t)chunk * IM::W_PAD + piece * 1024 + lane * 16;
      __builtin_amdgcn_global_load_lds((const void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + X_BYTES + piece * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][MF];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int mw = wv * MF * 16;

  issue(0);
  if (nchunks > 1) issue(1);
  for (int c = 0; c < nchunks; ++c) {
    // this wave's pieces of chunk c have landed once at most the NI pieces of chunk c+1 are still outstanding
    if (c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everyone's pieces of chunk c are in; everyone is done reading buffer (c-1)%3
    if (c + 2 < nchunks) issue(c + 2);
    const T* Xs = reinterpret_cast<const T*>(lds_dyn + (c % NBUF) * BUF_BYTES);
    const T* Ws = reinterpret_cast<const T*>(lds_dyn + (c % NBUF) * BUF_BYTES + X_BYTES);
    pw_chunk_mma<T, MF, true>(Xs, Ws, acc, mw, li, g);
  }
  pw_epilogue<T, MF>(p, reinterpret_cast<float*>(lds_dyn), acc, zb, zg, m0, n0, mw, lane, wv);
}

// ---- wave-owned forms (bf16, 64-pixel-aligned rows) ---------------------------------------------------------------------------
// The kernels above share every X chunk between the four waves of a workgroup, so a tile is a chain of workgroup barriers.  A
// prototype on a channel-blocked layout (tools/microbench/pw_blocked.hip, profiles/r01_zz_pw_blocked_layout.txt) showed what
// that costs: with the weights resident in LDS and every wave streaming its OWN pixel tiles - nothing a wave waits for except
// its own loads - the same GEMMs ran at 5.0-5.6 TB/s instead of 3.8.  Most of that survives on plane-major NCHW
// (tools/microbench/pw_plane.hip): the two transposes the layout forces (X chunk [k][px] -> k-contiguous MFMA operands; accumulators
// -> 128-byte channel rows) go through a 4.5 KB LDS patch that belongs to ONE wave, so they need no workgroup barrier either.
//   xres   (K <= 96, M > 96):  the whole weight matrix sits in LDS; a wave turns its X tile into operand registers once and walks
//                              the output channels 32 at a time - X is read once, Y written once.
//   stream (M <= 96, K <= 512): accumulators for every output channel; X streams through the patch one 32-k chunk at a time behind
//                              two chunks of register prefetch.
constexpr int PWW_MW = 8;                       // waves per workgroup
constexpr int PWW_XS = 72;                      // patch row stride (elements): 64 pixels + pad, conflict-free transposed reads
constexpr int PWW_PATCH = PW_KC * PWW_XS;       // bf16 elements per wave: one X chunk, or the fp32 slab of one 16-channel fragment
static_assert(PWW_PATCH * 2 >= 16 * (PW_TN + 4) * 4, "the patch must hold a 16-row fp32 slab");

struct PwwX {   // the X operand of slice z
  const bf16* x1; const bf16* x2; int k1, ktot; int64_t n;
};
// raw global loads of one X chunk (32 k x 64 px): instruction i covers rows 8i + lane/8, 16 bytes (8 pixels) per lane
__device__ __forceinline__ void pww_load_chunk(u32x4 (&raw)[4], const PwwX& x, int kb, int64_t n0, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = kb * PW_KC + 8 * i + (lane >> 3);
    const int kc = k < x.ktot ? k : 0;          // rows past K read row 0 and are zeroed on the way into the patch
    const bf16* row = kc < x.k1 ? x.x1 + (int64_t)kc * x.n : x.x2 + (int64_t)(kc - x.k1) * x.n;
    raw[i] = MI_STREAM_LD(reinterpret_cast<const u32x4*>(row + n0 + 8 * (lane & 7)));
  }
}
// chunk -> patch -> MFMA A operands (same k-slot order as pw_chunk_mma: element j<4 of lane group g is k = 4g+j, j>=4 is 16+4g+j-4)
__device__ __forceinline__ void pww_chunk_to_frags(s16x8 (&a)[4], const u32x4 (&raw)[4], bf16* patch, int kb, int ktot, int lane) {
  const int li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * i + (lane >> 3);
    *reinterpret_cast<u32x4*>(&patch[r * PWW_XS + 8 * (lane & 7)]) = kb * PW_KC + r < ktot ? raw[i] : zero4;
  }
  wave_lds_sync();
  s16x4 lo[4], hi[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    lo[nf] = lds_tr_b16(&patch[(4 * g + qq) * PWW_XS + 16 * nf + 4 * pp]);
    hi[nf] = lds_tr_b16(&patch[(16 + 4 * g + qq) * PWW_XS + 16 * nf + 4 * pp]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
               :
               : "memory");
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) a[nf] = __builtin_shufflevector(lo[nf], hi[nf], 0, 1, 2, 3, 4, 5, 6, 7);
  wave_lds_sync();
}
// B operand: 16 output channels; wr already points at this lane's row (li) and k offset (4g)
__device__ __forceinline__ s16x8 pww_w_frag(const bf16* wr) {
  const s16x4 lo = *reinterpret_cast<const s16x4*>(wr), hi = *reinterpret_cast<const s16x4*>(wr + 16);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// (MFMA operand policy - bf16 fragments as they are, or converted to fp8 e4m3 in registers: MfmaOp<F8> in common.h)
template <bool F8> using PwwOp = MfmaOp<F8>;
// packed weight image of this workgroup's slice -> LDS, chunk images back to back ([tile][chunk][tm][WS_ROW])
__device__ __forceinline__ void pww_stage_weights(bf16* Wl, const bf16* wpk, int images, int tm, int chunk_stride_elems, int t) {
  const int vpc = tm * PwRow<bf16>::WS_ROW / 8;
  for (int v = t; v < images * vpc; v += 64 * PWW_MW) {
    const int c = v / vpc, o = v - c * vpc;
    reinterpret_cast<u32x4*>(Wl)[v] = reinterpret_cast<const u32x4*>(wpk + (int64_t)c * chunk_stride_elems)[o];
  }
}
// Per-image weights (MDTA's project_out . softmax product and its transposes, the q / k gradient matrices) change with every call,
// so a packed image of them is never reused: with MI_PW_DIRECT=1 the wave-owned kernels read such a matrix straight from fp32 and
// round it to bf16 on the way into LDS - same values as the pack kernel, one launch less per GEMM, and measured slower (see
// pw_launch): kept as an A/B switch.  Chunk images [img][tm][WS_ROW], img = mt * nk + kc.
__device__ __forceinline__ void pww_stage_weights_f32(bf16* Wl, const float* __restrict__ w, int64_t sm, int64_t sk, int M, int K,
                                                      int mt0, int n_mt, int nk, int tm, int t) {
  constexpr int WS_ROW = PwRow<bf16>::WS_ROW;
  const int per = tm * PW_KC, total = n_mt * nk * per;
  for (int e = t; e < total; e += 64 * PWW_MW) {
    const int img = e / per, r = e - img * per;
    int mm, kk;
    if (sk == 1) { kk = r & (PW_KC - 1); mm = r / PW_KC; }           // consecutive threads follow the unit stride of the matrix
    else { mm = r % tm; kk = r / tm; }
    const int mt = img / nk, kc = img - mt * nk;
    const int m = (mt0 + mt) * tm + mm, k = kc * PW_KC + kk;
    const float v = (m < M && k < K) ? w[(int64_t)m * sm + (int64_t)k * sk] : 0.f;
    Wl[(img * tm + mm) * WS_ROW + kk] = (bf16)v;
  }
}
// The same from the bf16 copy a producer wrote beside its fp32 matrix (mi_pw_desc.w_b16: row-major, unit k stride, everything a
// multiple of 8): whole 16-byte vectors, i.e. exactly the loads and LDS stores pww_stage_weights spends on a packed image.
__device__ __forceinline__ void pww_stage_weights_b16(bf16* Wl, const bf16* __restrict__ w, int64_t sm, int M, int K, int mt0, int n_mt,
                                                      int nk, int tm, int t) {
  constexpr int WS_ROW = PwRow<bf16>::WS_ROW;
  const int per = tm * (PW_KC / 8), total = n_mt * nk * per;
  for (int e = t; e < total; e += 64 * PWW_MW) {
    const int img = e / per, r = e - img * per, mm = r / (PW_KC / 8), kv = r - mm * (PW_KC / 8);
    const int mt = img / nk, kc = img - mt * nk;
    const int m = (mt0 + mt) * tm + mm, k = kc * PW_KC + kv * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (m < M && k < K) v = *reinterpret_cast<const u32x4*>(w + (int64_t)m * sm + k);
    *reinterpret_cast<u32x4*>(&Wl[(img * tm + mm) * WS_ROW + kv * 8]) = v;
  }
}
struct PwwOut { bf16* y; const bf16* r; const float* bias; int m; int64_t n; bf16* y2; int split; };
// output row m: rows past `split` belong to the second output tensor (mi_pw_desc.y_split: two results of one pass over X)
__device__ __forceinline__ bf16* pww_out_row(const PwwOut& o, int m) {
  return (o.split > 0 && m >= o.split) ? o.y2 + (int64_t)(m - o.split) * o.n : o.y + (int64_t)m * o.n;
}
__device__ __forceinline__ void pww_load_res(u32x4 (&rr)[2], const PwwOut& o, int mbase, int64_t n0, int lane) {
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int m = mbase + it * 8 + (lane >> 3);
    rr[it] = (o.r && m < o.m) ? MI_STREAM_LD(reinterpret_cast<const u32x4*>(o.r + (int64_t)m * o.n + n0 + 8 * (lane & 7))) : zero4;
  }
}
// one 16-channel accumulator fragment (four pixel fragments) -> fp32 slab in the wave's patch -> + bias + residual -> 128-byte rows
__device__ __forceinline__ void pww_store_frag(const f32x4 (&acc)[4], const u32x4 (&rr)[2], float* slab, const PwwOut& o, int mbase,
                                               int64_t n0, int lane) {
  constexpr int OS = PW_TN + 4;
  const int li = lane & 15, g = lane >> 4, e_row = lane >> 3, e_col = (lane & 7) * 8;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    float v[4] = {acc[nf][0], acc[nf][1], acc[nf][2], acc[nf][3]};
    Vec<float, 4>::st(&slab[li * OS + 16 * nf + 4 * g], v);
  }
  wave_lds_sync();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int row = it * 8 + e_row, m = mbase + row;
    if (m < o.m) {
      float v[8];
      Vec<float, 4>::ld(&slab[row * OS + e_col], v);
      Vec<float, 4>::ld(&slab[row * OS + e_col + 4], v + 4);
      const float bv = o.bias ? o.bias[m] : 0.f;
      const u32x4 rv = rr[it];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[2 * j] += bv + bf16_bits_to_f32(rv[j] & 0xffffu);
        v[2 * j + 1] += bv + bf16_bits_to_f32(rv[j] >> 16);
      }
      Vec<bf16, 8>::st(pww_out_row(o, m) + n0 + e_col, v);
    }
  }
  wave_lds_sync();
}

// the same without a residual: bias is added in registers, the fragments are rounded once to bf16 and two of them (32 channels)
// share one trip through the patch - half the LDS bytes and half the trips of the fp32 slab
template <int NFR>
__device__ __forceinline__ void pww_store_bf16(const f32x4 (&acc0)[4], const f32x4 (&acc1)[4], bf16* patch, const PwwOut& o, int mbase,
                                               int64_t n0, int lane) {
  const int li = lane & 15, g = lane >> 4, e_row = lane >> 3, e_col = (lane & 7) * 8;
  const float b0 = (o.bias && mbase + li < o.m) ? o.bias[mbase + li] : 0.f;
  const float b1 = (NFR > 1 && o.bias && mbase + 16 + li < o.m) ? o.bias[mbase + 16 + li] : 0.f;
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    float v[4] = {acc0[nf][0] + b0, acc0[nf][1] + b0, acc0[nf][2] + b0, acc0[nf][3] + b0};
    Vec<bf16, 4>::st(&patch[li * PWW_XS + 16 * nf + 4 * g], v);
    if (NFR > 1) {
      float u[4] = {acc1[nf][0] + b1, acc1[nf][1] + b1, acc1[nf][2] + b1, acc1[nf][3] + b1};
      Vec<bf16, 4>::st(&patch[(16 + li) * PWW_XS + 16 * nf + 4 * g], u);
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int it = 0; it < 2 * NFR; ++it) {
    const int row = it * 8 + e_row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(&patch[row * PWW_XS + e_col]);
    if (mbase + row < o.m) MI_STREAM_ST(reinterpret_cast<u32x4*>(pww_out_row(o, mbase + row) + n0 + e_col), v);
  }
  wave_lds_sync();
}

// LayerNorm over the channels of one 64-pixel X tile while it sits in the raw load registers (instruction i of chunk kb holds
// row kb*32 + 8i + lane/8, pixels 8 (lane & 7) .. +7): per-pixel sums run over the lane's rows, then over the 8 lanes that share
// (lane & 7).  Same arithmetic as ln.hip (two-pass variance, (v - mu) * rstd * gamma + beta).  lnp: gamma[32 KB] | beta[32 KB] in LDS.
template <int KB>
__device__ __forceinline__ void pww_ln_inplace(u32x4 (&raw)[KB][4], int ktot, const float* lnp, int mode, float* mean_out,
                                               float* rstd_out, int lane) {
  const int r0 = lane >> 3;
  float s[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (kb * PW_KC + 8 * i + r0 < ktot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[2 * j] += bf16_bits_to_f32(raw[kb][i][j] & 0xffffu);
          s[2 * j + 1] += bf16_bits_to_f32(raw[kb][i][j] >> 16);
        }
      }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] += __shfl_xor(s[j], 8);
    s[j] += __shfl_xor(s[j], 16);
    s[j] += __shfl_xor(s[j], 32);
  }
  const float inv = 1.0f / (float)ktot;
#pragma unroll
  for (int j = 0; j < 8; ++j) { mu[j] = s[j] * inv; s[j] = 0.f; }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (kb * PW_KC + 8 * i + r0 < ktot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d0 = bf16_bits_to_f32(raw[kb][i][j] & 0xffffu) - mu[2 * j];
          const float d1 = bf16_bits_to_f32(raw[kb][i][j] >> 16) - mu[2 * j + 1];
          s[2 * j] += d0 * d0;
          s[2 * j + 1] += d1 * d1;
        }
      }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[j] += __shfl_xor(s[j], 8);
    s[j] += __shfl_xor(s[j], 16);
    s[j] += __shfl_xor(s[j], 32);
    rs[j] = 1.0f / sqrtf(s[j] * inv + 1e-5f);
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = kb * PW_KC + 8 * i + r0;
      if (k < ktot) {
        const float g = lnp[k], b = lnp[KB * PW_KC + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = bf16_bits_to_f32(raw[kb][i][j] & 0xffffu), v1 = bf16_bits_to_f32(raw[kb][i][j] >> 16);
          const float o0 = mode == 1 ? (v0 - mu[2 * j]) * rs[2 * j] * g + b : v0 * rs[2 * j] * g;
          const float o1 = mode == 1 ? (v1 - mu[2 * j + 1]) * rs[2 * j + 1] * g + b : v1 * rs[2 * j + 1] * g;
          raw[kb][i][j] = cvt_pk_bf16(o0, o1);
        }
      }
    }
  if (mean_out && r0 == 0) {
    float* mo = mean_out + 8 * (lane & 7);
    float* ro = rstd_out + 8 * (lane & 7);
    *reinterpret_cast<f32x4*>(mo) = (f32x4){mu[0], mu[1], mu[2], mu[3]};
    *reinterpret_cast<f32x4*>(mo + 4) = (f32x4){mu[4], mu[5], mu[6], mu[7]};
    *reinterpret_cast<f32x4*>(ro) = (f32x4){rs[0], rs[1], rs[2], rs[3]};
    *reinterpret_cast<f32x4*>(ro + 4) = (f32x4){rs[4], rs[5], rs[6], rs[7]};
  }
}

template <int KB, bool F8>
__global__ __launch_bounds__(64 * PWW_MW) void pw_gemm_wave_xres_kernel(PwG q, int m_tiles, int tiles_per_wave, int chunk_stride_elems) {
  using Op = PwwOp<F8>;
  Op::enter();
  const PwK& p = q.k;
  constexpr int WS_ROW = PwRow<bf16>::WS_ROW, TM = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  bf16* const Wl = reinterpret_cast<bf16*>(lds_dyn);                 // [m_tiles][KB][64][WS_ROW]
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: the wave's patch, tile range and their addresses stay out of the VGPRs
  bf16* const patch = Wl + (int64_t)m_tiles * KB * TM * WS_ROW + wv * PWW_PATCH;
  const int z = blockIdx.z, zb = z / p.groups, zg = z - zb * p.groups;
  const int wslice = (q.wp_per_batch ? zb : 0) * (q.wp_per_group ? p.groups : 1) + (q.wp_per_group ? zg : 0);
  if (q.w_direct && q.wb16)
    pww_stage_weights_b16(Wl, q.wb16 + zb * p.w_bs + zg * p.w_gs, q.wb16_sm, p.m, p.k1 + p.k2, 0, m_tiles, KB, TM, t);
  else if (q.w_direct)
    pww_stage_weights_f32(Wl, p.w + zb * p.w_bs + zg * p.w_gs, p.w_sm, p.w_sk, p.m, p.k1 + p.k2, 0, m_tiles, KB, TM, t);
  else
    pww_stage_weights(Wl, reinterpret_cast<const bf16*>(q.ws + PW_ZERO_BYTES) + (int64_t)wslice * q.wp_slice, m_tiles * KB, TM,
                      chunk_stride_elems, t);
  float* const lnp = reinterpret_cast<float*>(Wl + (int64_t)m_tiles * KB * TM * WS_ROW + PWW_MW * PWW_PATCH);   // gamma | beta
  if (q.ln_mode) {
    for (int i = t; i < KB * PW_KC; i += 64 * PWW_MW) {
      lnp[i] = i < p.k1 ? q.ln_w[i] : 0.f;
      lnp[KB * PW_KC + i] = (i < p.k1 && q.ln_b) ? q.ln_b[i] : 0.f;
    }
  }
  __syncthreads();                                                   // the only workgroup barrier
  PwwX x;
  x.x1 = (const bf16*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  x.x2 = p.x2 ? (const bf16*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : x.x1;
  x.k1 = p.k1; x.ktot = p.k1 + p.k2; x.n = p.n;
  PwwOut o;
  o.y = (bf16*)p.y + zb * p.y_bs + zg * p.y_gs;
  o.y2 = p.y2 ? (bf16*)p.y2 + zb * p.y2_bs + zg * p.y2_gs : nullptr; o.split = p.y_split;
  o.r = p.r ? (const bf16*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  o.bias = p.bias ? p.bias + zg * p.bias_gs : nullptr;
  o.m = p.m; o.n = p.n;
  const int64_t n_tiles = p.n / PW_TN;
  const int64_t tile0 = ((int64_t)blockIdx.x * PWW_MW + wv) * tiles_per_wave;
  u32x4 raw[KB][4];
  if (tile0 < n_tiles) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) pww_load_chunk(raw[kb], x, kb, tile0 * PW_TN, lane);
  }
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int64_t tile = tile0 + tt;
    if (tile >= n_tiles) break;
    const int64_t n0 = tile * PW_TN;
    int ln = lane;                                                   // opaque copy of the lane id: see pw_gemm_wave_stream_kernel
    asm volatile("" : "+v"(ln));
    const int li_ = ln & 15, g_ = ln >> 4;
    if (q.ln_mode)
      pww_ln_inplace<KB>(raw, x.ktot, lnp, q.ln_mode, q.ln_mean ? q.ln_mean + zb * p.n + n0 : nullptr,
                         q.ln_rstd ? q.ln_rstd + zb * p.n + n0 : nullptr, ln);
    typename Op::Frag a[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      s16x8 a16[4];
      pww_chunk_to_frags(a16, raw[kb], patch, kb, x.ktot, ln);
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) a[kb][nf] = Op::cvt(a16[nf], q.f8_sx);
    }
    if (tt + 1 < tiles_per_wave && tile + 1 < n_tiles) {             // the next tile's X flies while this tile's channels are computed
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) pww_load_chunk(raw[kb], x, kb, n0 + PW_TN, ln);
    }
    for (int mh = 0; mh < 2 * m_tiles; ++mh) {                       // 32 output channels per trip
      const int mbase = 32 * mh;
      if (mbase >= p.m) break;
      const bf16* wt = Wl + (int64_t)(mh >> 1) * KB * TM * WS_ROW + ((mh & 1) * 32 + li_) * WS_ROW + 4 * g_;
      f32x4 acc[2][4];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[f][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const typename Op::Frag b = Op::cvt(pww_w_frag(wt + (kb * TM + 16 * f) * WS_ROW), q.f8_sw);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[f][nf] = Op::mma(a[kb][nf], b, acc[f][nf]);
        }
      if (F8) {
        const float os = q.f8_sx * q.f8_sw;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[f][nf] *= os;
      }
      if (!o.r) {
        pww_store_bf16<2>(acc[0], acc[1], patch, o, mbase, n0, ln);
      } else {                                                       // (no model shape takes this branch: wide outputs carry no residual)
        u32x4 rr[2];
        pww_load_res(rr, o, mbase, n0, ln);
        pww_store_frag(acc[0], rr, reinterpret_cast<float*>(patch), o, mbase, n0, ln);
        pww_load_res(rr, o, mbase + 16, n0, ln);
        pww_store_frag(acc[1], rr, reinterpret_cast<float*>(patch), o, mbase + 16, n0, ln);
      }
    }
  }
}

template <int MF, bool F8>
__global__ __launch_bounds__(64 * PWW_MW) void pw_gemm_wave_stream_kernel(PwG q, int tiles_per_wave, int chunk_stride_elems) {
  using Op = PwwOp<F8>;
  Op::enter();
  const PwK& p = q.k;
  constexpr int WS_ROW = PwRow<bf16>::WS_ROW, TM = 16 * MF;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  bf16* const Wl = reinterpret_cast<bf16*>(lds_dyn);                 // [k_chunks][TM][WS_ROW]
  const int nchunks = q.k_chunks;
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);   // scalar: the wave's patch, tile range and their addresses stay out of the VGPRs
  bf16* const patch = Wl + (int64_t)nchunks * TM * WS_ROW + wv * PWW_PATCH;
  // Wide outputs: one workgroup per 16 MF-channel tile, and every tile streams the workgroup's X range again.  Workgroups are
  // handed to the 8 XCDs round-robin in launch order, so with the plain (x, y, z) order the m-tiles of one X range land on
  // different L2s.  xcd_map: XCD j instead takes the j-th contiguous eighth of the list ordered (image, pixel range, m-tile)
  // with the m-tile fastest - all m-tiles of an X range run back to back on ONE L2 and only the first of them goes to HBM.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (q.xcd_map) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned per = gridDim.x * gridDim.y * gridDim.z / 8u;
    const unsigned w = lin / 8u + per * (lin % 8u);
    by = (int)(w % gridDim.y);
    const unsigned u = w / gridDim.y;
    bx = (int)(u % gridDim.x);
    bz = (int)(u / gridDim.x);
  }
  const int z = bz, zb = z / p.groups, zg = z - zb * p.groups;
  const int wslice = (q.wp_per_batch ? zb : 0) * (q.wp_per_group ? p.groups : 1) + (q.wp_per_group ? zg : 0);
  const int m0 = by * TM;
  if (q.w_direct && q.wb16)
    pww_stage_weights_b16(Wl, q.wb16 + zb * p.w_bs + zg * p.w_gs, q.wb16_sm, p.m, p.k1 + p.k2, by, 1, nchunks, TM, t);
  else if (q.w_direct)
    pww_stage_weights_f32(Wl, p.w + zb * p.w_bs + zg * p.w_gs, p.w_sm, p.w_sk, p.m, p.k1 + p.k2, by, 1, nchunks, TM, t);
  else
    pww_stage_weights(Wl, reinterpret_cast<const bf16*>(q.ws + PW_ZERO_BYTES) + (int64_t)wslice * q.wp_slice +
                              (int64_t)by * nchunks * chunk_stride_elems, nchunks, TM, chunk_stride_elems, t);
  __syncthreads();
  PwwX x;
  x.x1 = (const bf16*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  x.x2 = p.x2 ? (const bf16*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : x.x1;
  x.k1 = p.k1; x.ktot = p.k1 + p.k2; x.n = p.n;
  PwwOut o;
  o.y = (bf16*)p.y + zb * p.y_bs + zg * p.y_gs;
  o.y2 = p.y2 ? (bf16*)p.y2 + zb * p.y2_bs + zg * p.y2_gs : nullptr; o.split = p.y_split;
  o.r = p.r ? (const bf16*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  o.bias = p.bias ? p.bias + zg * p.bias_gs : nullptr;
  o.m = p.m; o.n = p.n;
  const int64_t n_tiles = p.n / PW_TN;
  const int64_t tile0 = ((int64_t)bx * PWW_MW + wv) * tiles_per_wave;
  const bf16* wl = Wl + li * WS_ROW + 4 * g;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int64_t tile = tile0 + tt;
    if (tile >= n_tiles) break;
    const int64_t n0 = tile * PW_TN;
    int ln = lane;                                                   // opaque copy: the per-lane address terms below are loop-invariant,
    asm volatile("" : "+v"(ln));                                     // and hoisted out of the tile loop they cost registers this kernel lacks
    f32x4 acc[MF][4];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 r0[4], r1[4];                                              // two chunks in flight, statically named
    pww_load_chunk(r0, x, 0, n0, ln);
    if (nchunks > 1) pww_load_chunk(r1, x, 1, n0, ln);
    for (int kb = 0; kb < nchunks; kb += 2) {
      {
        s16x8 a16[4];
        pww_chunk_to_frags(a16, r0, patch, kb, x.ktot, ln);
        if (kb + 2 < nchunks) pww_load_chunk(r0, x, kb + 2, n0, ln);
        typename Op::Frag a[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) a[nf] = Op::cvt(a16[nf], q.f8_sx);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const typename Op::Frag b = Op::cvt(pww_w_frag(wl + (kb * TM + 16 * mf) * WS_ROW), q.f8_sw);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = Op::mma(a[nf], b, acc[mf][nf]);
        }
      }
      if (kb + 1 < nchunks) {
        s16x8 a16[4];
        pww_chunk_to_frags(a16, r1, patch, kb + 1, x.ktot, ln);
        if (kb + 3 < nchunks) pww_load_chunk(r1, x, kb + 3, n0, ln);
        typename Op::Frag a[4];
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) a[nf] = Op::cvt(a16[nf], q.f8_sx);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const typename Op::Frag b = Op::cvt(pww_w_frag(wl + ((kb + 1) * TM + 16 * mf) * WS_ROW), q.f8_sw);
#pragma unroll
          for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = Op::mma(a[nf], b, acc[mf][nf]);
        }
      }
    }
    if (F8) {
      const float os = q.f8_sx * q.f8_sw;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] *= os;
    }
    if (!o.r) {
#pragma unroll
      for (int mj = 0; mj < MF / 2; ++mj)
        if (m0 + 32 * mj < p.m) pww_store_bf16<2>(acc[2 * mj], acc[2 * mj + 1], patch, o, m0 + 32 * mj, n0, ln);
      if (MF & 1)
        if (m0 + 16 * (MF - 1) < p.m) pww_store_bf16<1>(acc[MF - 1], acc[MF - 1], patch, o, m0 + 16 * (MF - 1), n0, ln);
    } else {
      u32x4 rr[2], rn[2];                                            // residual rows: one fragment ahead
      pww_load_res(rr, o, m0, n0, ln);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if (m0 + 16 * mf >= p.m) break;
        if (mf + 1 < MF) pww_load_res(rn, o, m0 + 16 * (mf + 1), n0, ln);
        pww_store_frag(acc[mf], rr, reinterpret_cast<float*>(patch), o, m0 + 16 * mf, n0, ln);
        rr[0] = rn[0]; rr[1] = rn[1];
      }
    }
  }
}

// ---- X-resident, W streamed (wave form 3): the wide projections of the C = 128 .. 192 levels --------------------------------------
// The streaming form above re-reads X once per 96-channel tile, and its waves use each 4 KB X chunk for 24 MFMAs only: at K = 192,
// M = 576 / 1020 that is ~50 TB/s of L2 traffic at full MFMA rate, and the kernel sits at 15-20 % of the MFMA peak on L2 bandwidth
// (an XCD-aware tile order did not move it: profiles/r02_i_pw_xcd_map_ab.txt).  Here a wave keeps the MFMA fragments of its 64
// pixels for ALL of K in registers (KB x 16 VGPRs) and the workgroup walks the output channels 64 at a time: the weight image of
// one 64-row tile (KB chunk images, 30 KB at K = 192) is staged into LDS by all eight waves - double-buffered, the next tile's
// loads in flight during this tile's MFMAs, one barrier per tile - and every wave multiplies it with its resident X.  X is read
// once; the weights come from L2 once per workgroup.  Measured at bs 32 (profiles/r02_j_pw_xwide_deep_shapes.txt): 576 x 192 75 -> 59 us,
// 1020 x 192 141 -> 105 us, 510 x 192 (W^T) 74 -> 53 us - now bound by the output writes (3.0-3.4 TB/s of the 4.45 TB/s write roof).
// A 32-pixel-per-wave variant for K = 193 .. 384 (fragments still 96 VGPRs) was built, parity-tested and measured: no gain at the latent
// level (1152 x 384 66 -> 71 us, 2042 x 384 122 -> 112 us) - each weight fragment read from LDS then feeds only 2 MFMAs - and was dropped.
constexpr int PWX_SR = 64;   // output channels per staged weight slab (32 - two workgroups per CU by LDS, twice the barriers - measured 5-12 % slower)
template <int KB, bool F8, bool LN>   // LN: LayerNorm on load (a template parameter: as a run-time branch its registers spilled the plain path)
__global__ __launch_bounds__(64 * PWW_MW) void pw_gemm_wave_xwide_kernel(PwG q, int n_slabs, int slabs_per_wg, int chunk_stride_elems) {
  using Op = PwwOp<F8>;
  Op::enter();
  const PwK& p = q.k;
  constexpr int WS_ROW = PwRow<bf16>::WS_ROW, TM = 64;
  constexpr int NF = 4, SR = PWX_SR, TNW = PW_TN;                        // pixel fragments per wave, channels per staged slab, pixels per wave
  constexpr int VPC = SR * WS_ROW / 8;                               // 16-byte vectors per chunk image inside one slab
  constexpr int VPS = KB * VPC;
  constexpr int NT = 64 * PWW_MW;
  constexpr int VPT = (VPS + NT - 1) / NT;
  constexpr int SLAB = KB * SR * WS_ROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  bf16* const Wl = reinterpret_cast<bf16*>(lds_dyn);                 // [2][KB][SR][WS_ROW]
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  bf16* const patch = Wl + 2 * SLAB + wv * PWW_PATCH;
  const int z = blockIdx.z, zb = z / p.groups, zg = z - zb * p.groups;
  const int wslice = (q.wp_per_batch ? zb : 0) * (q.wp_per_group ? p.groups : 1) + (q.wp_per_group ? zg : 0);
  const bf16* const wpk = reinterpret_cast<const bf16*>(q.ws + PW_ZERO_BYTES) + (int64_t)wslice * q.wp_slice;
  const int s0 = blockIdx.y * slabs_per_wg;
  const int s1 = s0 + slabs_per_wg < n_slabs ? s0 + slabs_per_wg : n_slabs;
  u32x4 wr[VPT];
  const float* const wf = p.w + zb * p.w_bs + zg * p.w_gs;           // (per-image weights: staged straight from fp32, see w_store)
  auto w_load = [&](int sl) {                                        // slab sl = rows [SR sl, SR sl + SR) of the packed 64-row tiles
    if (q.w_direct) return;
    int tt = t;
    asm volatile("" : "+v"(tt));
    const int mt = sl * SR / TM, r0 = sl * SR - mt * TM;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tt + NT * i;
      if (v < VPS) {
        const int c = v / VPC, o = v - c * VPC;
        wr[i] = reinterpret_cast<const u32x4*>(wpk + ((int64_t)mt * KB + c) * chunk_stride_elems + r0 * WS_ROW)[o];
      }
    }
  };
  auto w_store = [&](int buf, int sl) {
    if (q.w_direct) {
      if (q.wb16) pww_stage_weights_b16(Wl + buf * SLAB, q.wb16 + zb * p.w_bs + zg * p.w_gs, q.wb16_sm, p.m, p.k1 + p.k2, sl, 1, KB, SR, t);
      else pww_stage_weights_f32(Wl + buf * SLAB, wf, p.w_sm, p.w_sk, p.m, p.k1 + p.k2, sl, 1, KB, SR, t);
      return;
    }
    int tt = t;
    asm volatile("" : "+v"(tt));
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tt + NT * i;
      if (v < VPS) reinterpret_cast<u32x4*>(Wl + buf * SLAB)[v] = wr[i];
    }
  };
  if constexpr (!LN) w_load(s0);                                     // (LN form: after the tile is normalised - its registers are needed there)
  float* const lnp = reinterpret_cast<float*>(Wl + 2 * SLAB + PWW_MW * PWW_PATCH);   // gamma | beta (LayerNorm on load)
  if constexpr (LN) {
    for (int i = t; i < KB * PW_KC; i += NT) {
      lnp[i] = i < p.k1 ? q.ln_w[i] : 0.f;
      lnp[KB * PW_KC + i] = (i < p.k1 && q.ln_b) ? q.ln_b[i] : 0.f;
    }
    __syncthreads();
  }

  PwwX x;
  x.x1 = (const bf16*)p.x1 + zb * p.x1_bs + zg * p.x1_gs;
  x.x2 = p.x2 ? (const bf16*)p.x2 + zb * p.x2_bs + zg * p.x2_gs : x.x1;
  x.k1 = p.k1; x.ktot = p.k1 + p.k2; x.n = p.n;
  PwwOut o;
  o.y = (bf16*)p.y + zb * p.y_bs + zg * p.y_gs;
  o.y2 = p.y2 ? (bf16*)p.y2 + zb * p.y2_bs + zg * p.y2_gs : nullptr; o.split = p.y_split;
  o.r = p.r ? (const bf16*)p.r + zb * p.r_bs + zg * p.r_gs : nullptr;
  o.bias = p.bias ? p.bias + zg * p.bias_gs : nullptr;
  o.m = p.m; o.n = p.n;
  const int64_t tile = (int64_t)blockIdx.x * PWW_MW + wv;
  const bool valid = tile < p.n / TNW;                               // (waves past the plane still stage weights and meet the barriers)
  const int64_t n0 = tile * TNW;
  typename Op::Frag a[KB][NF];
  if (valid) {
    u32x4 raw[KB][NF];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      pww_load_chunk(raw[kb], x, kb, n0, lane);
    }
    if constexpr (LN)                                                // LayerNorm over K while the tile sits in the load registers
      pww_ln_inplace<KB>(raw, x.ktot, lnp, q.ln_mode, q.ln_mean ? q.ln_mean + zb * p.n + n0 : nullptr,
                         q.ln_rstd ? q.ln_rstd + zb * p.n + n0 : nullptr, lane);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      s16x8 a16[NF];
      pww_chunk_to_frags(a16, raw[kb], patch, kb, x.ktot, lane);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) a[kb][nf] = Op::cvt(a16[nf], q.f8_sx);
    }
  }
  if constexpr (LN) w_load(s0);
  w_store(0, s0);
  __syncthreads();
  for (int sl = s0; sl < s1; ++sl) {
    const int buf = (sl - s0) & 1;
    if (sl + 1 < s1) w_load(sl + 1);
    if (valid) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int li_ = ln & 15, g_ = ln >> 4;
#pragma unroll 1
      for (int h = 0; h < SR / 32; ++h) {                            // 32 output channels per trip
        const int mbase = sl * SR + 32 * h;
        if (mbase >= p.m) break;
        const bf16* wt = Wl + buf * SLAB + (32 * h + li_) * WS_ROW + 4 * g_;
        f32x4 acc[2][NF];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[f][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const typename Op::Frag b = Op::cvt(pww_w_frag(wt + (kb * SR + 16 * f) * WS_ROW), q.f8_sw);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[f][nf] = Op::mma(a[kb][nf], b, acc[f][nf]);
          }
        if (F8) {
          const float os = q.f8_sx * q.f8_sw;
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[f][nf] *= os;
        }
        if (!o.r) {
          pww_store_bf16<2>(acc[0], acc[1], patch, o, mbase, n0, ln);
        } else {
          u32x4 rr[2];
          pww_load_res(rr, o, mbase, n0, ln);
          pww_store_frag(acc[0], rr, reinterpret_cast<float*>(patch), o, mbase, n0, ln);
          pww_load_res(rr, o, mbase + 16, n0, ln);
          pww_store_frag(acc[1], rr, reinterpret_cast<float*>(patch), o, mbase + 16, n0, ln);
        }
      }
    }
    if (sl + 1 < s1) w_store(buf ^ 1, sl + 1);
    __syncthreads();
  }
}

struct PwPlan { int tm, m_tiles, k_chunks, slices, per_batch, per_group, chunk_elems, wave; int64_t slice_elems; size_t bytes; };
constexpr size_t PWW_LDS_MAX = 160 * 1024;

static bool pw_vec_ok(const mi_pw_desc* d) {
  const int64_t vec = d->dtype == MI_BF16 ? 8 : 4;
  bool ok = (d->n % vec == 0) && aligned16(d->x1) && aligned16(d->x2) && aligned16(d->y) && aligned16(d->r);
  ok = ok && d->x1_bs % vec == 0 && d->x1_gs % vec == 0 && d->x2_bs % vec == 0 && d->x2_gs % vec == 0;
  ok = ok && d->y_bs % vec == 0 && d->y_gs % vec == 0 && d->r_bs % vec == 0 && d->r_gs % vec == 0;
  return ok;
}


static PwPlan pw_plan(const mi_pw_desc* d, bool allow_wave = true) {
  PwPlan pl;
  // m-tile 128 when its padding stays within 25% of the 64-granular minimum, else 64.  (256-row tiles measured 5-8%
  // slower on the wide GDFN shapes: fewer resident workgroups; profiles/r01_n_pw_tile_staging_ab.log)
  // m-tile: every m-tile streams all of X again and writes its (padded) rows of Y, so the rows moved per pixel are
  // m_tiles * (K + tm); 96- and 48-row tiles keep 96- / 48- / 144- / 288-channel matrices from being padded by a third
  // (ties go to the larger tile; the LDS-DMA kernel only has 64 / 128).
  {
    const bool dma = MI_ENV(MI_PW_DMA) != nullptr || MI_ENV(MI_PW_TM_EVEN) != nullptr;   // (the latter: A/B switch)
    const int ktot = d->k1 + d->k2;
    int64_t best_cost = 0;
    pl.tm = 0;
    for (int tm : {128, 96, 64, 48}) {
      if (dma && (tm == 96 || tm == 48)) continue;
      // the uneven tiles only where they fit exactly (M = 48 / 96 / 288): a half-empty 96-row tile measured slower than
      // the 64-row tiling it would replace (M = 144: 291 vs 268 us; profiles/r01_zz_pw_tm96_bs32.log)
      if (tm == 96 && !(d->m <= 96 || d->m % 96 == 0)) continue;
      if (tm == 48 && d->m > 48) continue;
      const int64_t cost = (int64_t)cdiv(d->m, tm) * (ktot + tm);
      if (!pl.tm || cost < best_cost) { pl.tm = tm; best_cost = cost; }
    }
  }
  pl.k_chunks = cdiv(d->k1 + d->k2, PW_KC);
  // wave-owned forms (1: xres, 2: stream) where their LDS budget holds; they fix the packed tile height
  pl.wave = 0;
  {
    const char* e = MI_ENV(MI_PW_WAVE);
    const bool off = (e && e[0] == '0') || MI_ENV(MI_PW_DMA) || MI_ENV(MI_PW_CHUNKED);
    if (allow_wave && !off && d->dtype == MI_BF16 && d->n % PW_TN == 0 && pw_vec_ok(d)) {
      const size_t patches = (size_t)PWW_MW * PWW_PATCH * sizeof(bf16), row = PwRow<bf16>::WS_ROW * sizeof(bf16);
      const char* xw = MI_ENV(MI_PW_XWIDE);                          // A/B switch
      if (d->m > 96 && pl.k_chunks <= 3 && (size_t)pl.k_chunks * cdiv(d->m, 64) * 64 * row + patches + 768 <= PWW_LDS_MAX) {
        pl.wave = 1; pl.tm = 64;
      } else if (d->m >= 256 && pl.k_chunks >= 4 && pl.k_chunks <= 6 && !(xw && xw[0] == '0')) {   // (192 x 192 measured better on the streaming form)
        pl.wave = 3; pl.tm = 64;                                       // X-resident, W streamed (K = 97 .. 192, wide outputs)
      } else {
        // stream: one 96- / 64- / 48-channel tile per workgroup; wide outputs tile M (every tile streams X again, like the
        // chunked kernel), preferring the tile height that moves the fewest rows and fits beside the patches
        int best = 0;
        int64_t best_cost = 0;
        for (int tm : {96, 64, 48}) {
          if (d->m <= 48 && tm != 48) continue;
          if (d->m <= 64 && tm == 96) continue;
          if ((size_t)pl.k_chunks * tm * row + patches > PWW_LDS_MAX) continue;
          const int64_t cost = (int64_t)cdiv(d->m, tm) * (d->k1 + d->k2 + tm);
          if (!best || cost < best_cost) { best = tm; best_cost = cost; }
        }
        const char* w = MI_ENV(MI_PW_WAVE_WIDE);
        if (best && (d->m <= 96 || !(w && w[0] == '0'))) { pl.wave = 2; pl.tm = best; }
      }
    }
  }
  pl.m_tiles = cdiv(d->m, pl.tm);
  pl.per_batch = d->w_bs != 0;
  pl.per_group = d->w_gs != 0;
  pl.slices = (pl.per_batch ? d->batch : 1) * (pl.per_group ? d->groups : 1);
  const int ws_row = d->dtype == MI_F32 ? PwRow<float>::WS_ROW : PwRow<bf16>::WS_ROW;
  const size_t es = dtype_size(d->dtype);
  pl.chunk_elems = (int)(align_up((size_t)pl.tm * ws_row * es, 4096) / es);
  pl.slice_elems = (int64_t)pl.m_tiles * pl.k_chunks * pl.chunk_elems;
  pl.bytes = align_up(PW_ZERO_BYTES + (size_t)pl.slices * pl.slice_elems * es, 256);
  return pl;
}

template <typename T, int MF>
static int pw_launch_dma(const PwG& q, dim3 grid, hipStream_t st) {
  using IM = PwImg<T, 64 * MF>;
  constexpr int lds = 3 * (PW_KC * PW_TN * (int)sizeof(T) + IM::W_PAD);
  if (lds > 64 * 1024)
    MI_CHECK_HIP(hipFuncSetAttribute((const void*)pw_gemm_dma_kernel<T, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((pw_gemm_dma_kernel<T, MF>), grid, dim3(256), lds, st, q);
  return MI_OK;
}

// ---- opt-in packed-weight cache (mi_pw_cache_*) -----------------------------------------------------------------------------
// Every static 1x1 weight is packed once per use, i.e. ~260 seven-microsecond launches per Restormer training step whose
// inputs only change at the optimizer step.  A caller that owns the weights' lifetime (the trainer) can lend the library
// a device buffer: packed images then live there, keyed by (pointer, shape, strides, tiling), and ONE launch
// (mi_pw_cache_refresh, right after the optimizer step) re-packs them all.  Disabled by default: without it every call
// packs into its own workspace and the library keeps no state.
struct PwCache {
  std::mutex mu;
  unsigned char* base = nullptr;
  size_t bytes = 0, used = 0, table_bytes = 0;
  std::vector<PackJob> jobs;
  bool dirty = false;   // host table newer than the device copy
  bool valid = false;   // packed images match the weights (set by refresh, cleared by invalidate / new entries)
  const char* lo = nullptr;   // only weights inside [lo, hi) are cached: the caller's parameter storage.  A temporary
  const char* hi = nullptr;   // (a permuted copy, say) may reuse an address with other values and must never hit.
};
static PwCache g_pwc;
constexpr size_t PWC_TABLE_ENTRIES = 4096;

static bool same_job(const PackJob& a, const PackJob& b) {
  return a.w == b.w && a.w_bs == b.w_bs && a.w_gs == b.w_gs && a.w_sm == b.w_sm && a.w_sk == b.w_sk && a.M == b.M &&
         a.K == b.K && a.tm == b.tm && a.dtype == b.dtype && a.groups_w == b.groups_w && a.slices == b.slices &&
         a.chunk_elems == b.chunk_elems;
}
// Returns the cached image of this job if it is current; otherwise registers the job (so that the next refresh covers
// it) and returns null: the caller then packs into its own workspace as usual.
static const unsigned char* pw_cache_lookup(const PackJob& j, size_t image_bytes, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_pwc.mu);
  if (!g_pwc.base || j.w_bs != 0) return nullptr;                 // per-image weights are never cached
  if ((const char*)j.w < g_pwc.lo || (const char*)j.w >= g_pwc.hi) return nullptr;   // not in the registered parameter storage
  for (const PackJob& e : g_pwc.jobs)
    if (same_job(e, j)) return g_pwc.valid && !g_pwc.dirty ? e.ws : nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;  // no new entries mid-capture
  const size_t need = align_up(image_bytes, 256);
  if (g_pwc.jobs.size() >= PWC_TABLE_ENTRIES || g_pwc.used + need > g_pwc.bytes) return nullptr;      // full: stay uncached
  PackJob e = j;
  e.ws = g_pwc.base + g_pwc.used;
  g_pwc.used += need;
  g_pwc.jobs.push_back(e);
  g_pwc.dirty = true;
  return nullptr;
}

template <typename T>
static int pw_launch(const mi_pw_desc* d, const PwK& k, const PwPlan& pl, void* ws, hipStream_t st) {
  PackJob job;
  job.w = d->w; job.w_bs = d->w_bs; job.w_gs = d->w_gs; job.w_sm = d->w_sm; job.w_sk = d->w_sk;
  job.ws = (unsigned char*)ws;
  job.M = d->m; job.K = d->k1 + d->k2; job.tm = pl.tm; job.k_chunks = pl.k_chunks;
  job.groups_w = pl.per_group ? d->groups : 1; job.chunk_elems = pl.chunk_elems; job.m_fast = d->w_sk != 1 ? 1 : 0;
  job.dtype = d->dtype; job.slices = pl.slices; job.slice_elems = pl.slice_elems;
  // MI_PW_DIRECT=1 (A/B switch, off): per-image weights on a wave-owned form without a packed image - the kernels stage them from
  // fp32.  Measured SLOWER (profiles/r02_m_per_image_weights_direct_ab.txt: 165.0 vs 157.2 ms per step): every one of the ~2000
  // workgroups of such a GEMM repeats the scalar fp32 -> bf16 walk that one 7 us pack launch does once.
  // Round 4: a producer that also wrote the matrices in bf16 (mi_pw_desc.w_b16) makes the direct staging free - 16-byte copies, as
  // the packed path's own staging - and the pack launch (183 per training step) goes away.  MI_PW_B16=0 keeps the pack.
  const bool b16ok = pl.wave != 0 && d->w_bs != 0 && std::is_same<T, bf16>::value && d->w_b16 && aligned16(d->w_b16) &&
                     d->w_b16_sm % 8 == 0 && d->w_bs % 8 == 0 && d->w_gs % 8 == 0 && (d->k1 + d->k2) % 8 == 0 && !d->f8 &&
                     !(MI_ENV(MI_PW_B16) && atoi(MI_ENV(MI_PW_B16)) == 0);
  const bool direct = b16ok || (pl.wave != 0 && d->w_bs != 0 && std::is_same<T, bf16>::value && MI_ENV(MI_PW_DIRECT));
  const unsigned char* cached = direct ? nullptr : pw_cache_lookup(job, pl.bytes, st);
  if (!cached && !direct) {  // re-pack the weights of every slice (and refresh the zero block)
    const int64_t total = (int64_t)pl.m_tiles * pl.k_chunks * pl.tm * PW_KC;
    int gx = cdiv(total, 256);
    if (gx > 1024) gx = 1024;
    ProfScope ps(st, K_PW_PACK, 4.0 * d->m * (d->k1 + d->k2) * pl.slices + (double)pl.bytes, 0.0);
    hipLaunchKernelGGL((pw_pack_kernel<T>), dim3(gx, pl.slices), dim3(256), 0, st, job);
    MI_LAUNCH_CHECK();
  }
  PwG q;
  q.k = k; q.ws = cached ? cached : (const unsigned char*)ws; q.wp_slice = pl.slice_elems; q.wp_per_batch = pl.per_batch;
  q.wp_per_group = pl.per_group; q.k_chunks = pl.k_chunks;
  q.ln_w = d->ln_w; q.ln_b = d->ln_b; q.ln_mean = d->ln_mean; q.ln_rstd = d->ln_rstd; q.ln_mode = d->ln_mode;
  q.f8_sx = d->f8_sx; q.f8_sw = d->f8_sw;
  q.xcd_map = 0;
  q.w_direct = direct ? 1 : 0;
  q.wb16 = b16ok ? (const bf16*)d->w_b16 : nullptr; q.wb16_sm = d->w_b16_sm;
  if (d->y_split) {
    MI_CHECK_ARG(pl.wave != 0 && (std::is_same<T, bf16>::value), "pw_gemm: a split output needs a wave-owned bf16 form (mi_pw_gemm_split_ok)");
    MI_CHECK_ARG(d->y2 && !d->r && d->y_split > 0 && d->y_split < d->m && aligned16(d->y2) && d->y2_bs % 8 == 0 && d->y2_gs % 8 == 0,
                 "pw_gemm: split output: y2 (16-byte aligned), 0 < y_split < m, no residual");
  }
  if (d->f8) {
    MI_CHECK_ARG(pl.wave != 0 && (std::is_same<T, bf16>::value), "pw_gemm: fp8 operands need a wave-owned bf16 form (mi_pw_gemm_f8_ok)");
    MI_CHECK_ARG(d->f8_sx > 0.f && d->f8_sw > 0.f, "pw_gemm: fp8 operand scales must be positive (powers of two)");
  }
  if (d->ln_mode) {
    MI_CHECK_ARG((pl.wave == 1 || (pl.wave == 3 && pl.k_chunks <= 4)) && d->k2 == 0 && d->groups == 1 && d->ln_w && (d->ln_mode == 2 || d->ln_b) &&
                     (d->ln_mode == 1 || d->ln_mode == 2) && (d->ln_mean == nullptr) == (d->ln_rstd == nullptr),
                 "pw_gemm: LayerNorm-on-load needs an X-resident form (bf16; 96 < M with K <= 96, or 256 <= M with K <= 128; one K panel, one group; "
                 "mi_pw_gemm_ln_ok)");
  }
  dim3 grid(cdiv(k.n, PW_TN), pl.m_tiles, d->batch * k.groups), block(256);
  if (grid.y > 65535 || grid.z > 65535) { set_error("pw_gemm: grid too large"); return MI_ERR_ARG; }
  const double Z = (double)d->batch * k.groups, kt = k.k1 + k.k2;
  // the forward attention product  out = M_b . v (+ x): per-image weights, one group, not transposed (modules.hip attn_core_fwd)
  const bool is_av = d->w_bs != 0 && k.groups == 1 && d->w_sk == 1 && k.r != nullptr;
  ProfScope ps(st, is_av ? K_PW_AV : K_PW_GEMM, (kt + k.m + (k.r ? k.m : 0)) * (double)k.n * Z * sizeof(T) + 4.0 * k.m * kt,
               2.0 * k.m * kt * (double)k.n * Z);
  // The LDS-DMA ring measured 5-12% SLOWER than register staging on every Restormer shape (K is 2-16 chunks, so the
  // per-tile prologue and epilogue dominate and its 3 x chunk LDS footprint halves the resident workgroups).  It stays
  // opt-in (MI_PW_DMA=1) as the base of a persistent cross-tile pipeline; tests run it through the same parity cases.
  const bool dma = MI_ENV(MI_PW_DMA) != nullptr;
  if (pl.wave) {
    if constexpr (std::is_same<T, bf16>::value) {
      const int64_t n_tiles = k.n / PW_TN;
      const size_t row = PwRow<bf16>::WS_ROW * sizeof(bf16), patches = (size_t)PWW_MW * PWW_PATCH * sizeof(bf16);
      const size_t wbytes = (size_t)(pl.wave == 1 ? pl.m_tiles : 1) * pl.k_chunks * pl.tm * row;
      // pixel tiles per wave: 2, or 4 to amortise a weight image near the LDS limit (510 x 96 at 256^2, bs 32: 535 vs 565 us;
      // the smaller images lose 2-3% at 4), while at least ~4 workgroups per CU remain
      int64_t tpw = wbytes > 96 * 1024 ? 4 : 2;
      const int64_t par = n_tiles * grid.z * (pl.wave == 2 ? pl.m_tiles : 1) / ((int64_t)PWW_MW * 256 * 4);
      if (tpw > par) tpw = par;
      if (const char* e = MI_ENV(MI_PW_WAVE_TPW)) tpw = atoi(e);
      if (tpw < 1) tpw = 1;
      if (tpw > 8) tpw = 8;
      dim3 wgrid((unsigned)cdiv(n_tiles, tpw * PWW_MW), pl.wave == 2 ? pl.m_tiles : 1, grid.z), wblock(64 * PWW_MW);
      size_t lds = wbytes + patches + (d->ln_mode ? (size_t)2 * pl.k_chunks * PW_KC * sizeof(float) : 0);
      int n_slabs = 0, slabs_per = 0;
      if (pl.wave == 3) {                                              // one 64-pixel tile per wave; M split only to fill the chip
        const int sr = PWX_SR;
        wgrid.x = (unsigned)cdiv(n_tiles, PWW_MW);
        n_slabs = cdiv(d->m, sr);
        const int64_t wgs = (int64_t)wgrid.x * grid.z;
        const int64_t split = wgs >= 200 ? 1 : std::min<int64_t>(n_slabs, cdiv(256, wgs));
        slabs_per = (int)cdiv(n_slabs, split);
        wgrid.y = (unsigned)cdiv(n_slabs, slabs_per);
        lds = (size_t)2 * pl.k_chunks * sr * row + patches + (d->ln_mode ? (size_t)2 * pl.k_chunks * PW_KC * sizeof(float) : 0);
      }
      {
        const char* e = MI_ENV(MI_PW_XCD);                           // A/B switch
        const int min_tiles = e ? atoi(e) : 0;   // off by default: measured no gain (profiles/r02_i_pw_xcd_map_ab.txt)
        const uint64_t total = (uint64_t)wgrid.x * wgrid.y * wgrid.z;
        q.xcd_map = (pl.wave == 2 && min_tiles > 0 && (int)wgrid.y >= min_tiles && total % 8 == 0 && total < (1ull << 31)) ? 1 : 0;
      }
#define PWW_LAUNCH(KERNEL, ...)                                                                                              \
  do {                                                                                                                       \
    if (lds > 64 * 1024) MI_CHECK_HIP(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(KERNEL, wgrid, wblock, lds, st, q, __VA_ARGS__);                                                      \
  } while (0)
      if (pl.wave == 3) {
#define PWX_CASE(KB_, LNOK_)                                                                                                   \
  if (pl.k_chunks == KB_) {                                                                                                    \
    if (LNOK_ && d->f8 && d->ln_mode) PWW_LAUNCH((pw_gemm_wave_xwide_kernel<KB_, true, LNOK_>), n_slabs, slabs_per, pl.chunk_elems);  \
    else if (d->f8) PWW_LAUNCH((pw_gemm_wave_xwide_kernel<KB_, true, false>), n_slabs, slabs_per, pl.chunk_elems);             \
    else if (LNOK_ && d->ln_mode) PWW_LAUNCH((pw_gemm_wave_xwide_kernel<KB_, false, LNOK_>), n_slabs, slabs_per, pl.chunk_elems);     \
    else PWW_LAUNCH((pw_gemm_wave_xwide_kernel<KB_, false, false>), n_slabs, slabs_per, pl.chunk_elems);                       \
  }
        PWX_CASE(4, true) else PWX_CASE(5, false) else PWX_CASE(6, false)
#undef PWX_CASE
      } else if (pl.wave == 1 && !d->f8) {
        if (pl.k_chunks == 1) PWW_LAUNCH((pw_gemm_wave_xres_kernel<1, false>), pl.m_tiles, (int)tpw, pl.chunk_elems);
        else if (pl.k_chunks == 2) PWW_LAUNCH((pw_gemm_wave_xres_kernel<2, false>), pl.m_tiles, (int)tpw, pl.chunk_elems);
        else PWW_LAUNCH((pw_gemm_wave_xres_kernel<3, false>), pl.m_tiles, (int)tpw, pl.chunk_elems);
      } else if (pl.wave == 1) {
        if (pl.k_chunks == 1) PWW_LAUNCH((pw_gemm_wave_xres_kernel<1, true>), pl.m_tiles, (int)tpw, pl.chunk_elems);
        else if (pl.k_chunks == 2) PWW_LAUNCH((pw_gemm_wave_xres_kernel<2, true>), pl.m_tiles, (int)tpw, pl.chunk_elems);
        else PWW_LAUNCH((pw_gemm_wave_xres_kernel<3, true>), pl.m_tiles, (int)tpw, pl.chunk_elems);
      } else if (!d->f8) {
        if (pl.tm == 96) PWW_LAUNCH((pw_gemm_wave_stream_kernel<6, false>), (int)tpw, pl.chunk_elems);
        else if (pl.tm == 64) PWW_LAUNCH((pw_gemm_wave_stream_kernel<4, false>), (int)tpw, pl.chunk_elems);
        else PWW_LAUNCH((pw_gemm_wave_stream_kernel<3, false>), (int)tpw, pl.chunk_elems);
      } else {
        if (pl.tm == 96) PWW_LAUNCH((pw_gemm_wave_stream_kernel<6, true>), (int)tpw, pl.chunk_elems);
        else if (pl.tm == 64) PWW_LAUNCH((pw_gemm_wave_stream_kernel<4, true>), (int)tpw, pl.chunk_elems);
        else PWW_LAUNCH((pw_gemm_wave_stream_kernel<3, true>), (int)tpw, pl.chunk_elems);
      }
#undef PWW_LAUNCH
    }
  } else if (k.vec_ok && dma) {
    if (pl.tm == 128) MI_TRY((pw_launch_dma<T, 2>(q, grid, st)));
    else MI_TRY((pw_launch_dma<T, 1>(q, grid, st)));
  } else if (std::is_same<T, bf16>::value && k.vec_ok && pl.k_chunks <= PWR_MAXC && k.m > 64 && !MI_ENV(MI_PW_CHUNKED)) {
    // (M <= 64 stays chunked: 48 x 48 is a tie and 48 x 127 loses 13%; profiles/r01_y_pw_resident_bs32.log)
    // weight-resident persistent kernel (K <= 128).  Things that did NOT help the chunked kernel beyond the Infinity Cache
    // (bs 32; profiles/r01_v_pw_*bs32*.log): a 128 x 128 tile, also persistent and cross-tile pipelined (-10..25%: 2-3
    // resident workgroups per CU instead of 5); two chunks of prefetch (-3..10%); an XCD-aware tile map running the
    // m-tiles of a pixel tile together on one L2 (-15..35%: 4x more channel rows over 4x narrower pixel spans in flight).
    const int n_tiles = (int)cdiv(k.n, PW_TN);
    const int wc = pl.tm * PwRow<bf16>::WS_ROW;
    const int xbuf = std::max(PWR_MAXC * PW_KC * PW_XS, 4 * 16 * (PW_TN + 4) * 2);
    const size_t lds = ((size_t)pl.k_chunks * wc + xbuf) * sizeof(bf16);
    // pixel tiles per workgroup: enough to amortise the weight load, while >= 2048 workgroups remain
    int64_t tpb = (int64_t)n_tiles * pl.m_tiles * grid.z / 2048;
    if (const char* e = MI_ENV(MI_PW_TPB)) tpb = atoi(e);
    if (tpb < 1) tpb = 1;
    if (tpb > 32) tpb = 32;
    dim3 rgrid((unsigned)cdiv(n_tiles, tpb), pl.m_tiles, grid.z);
    if (pl.tm == 128) {
      if (lds > 64 * 1024) MI_CHECK_HIP(hipFuncSetAttribute((const void*)pw_gemm_res_kernel<2, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((pw_gemm_res_kernel<2, 128>), rgrid, block, lds, st, q, n_tiles, pl.chunk_elems);
    } else if (pl.tm == 96) {
      hipLaunchKernelGGL((pw_gemm_res_kernel<2, 96>), rgrid, block, lds, st, q, n_tiles, pl.chunk_elems);
    } else if (pl.tm == 64) {
      hipLaunchKernelGGL((pw_gemm_res_kernel<1, 64>), rgrid, block, lds, st, q, n_tiles, pl.chunk_elems);
    } else {
      hipLaunchKernelGGL((pw_gemm_res_kernel<1, 48>), rgrid, block, lds, st, q, n_tiles, pl.chunk_elems);
    }
  } else {
    if (pl.tm == 128) hipLaunchKernelGGL((pw_gemm_kernel<T, 2, 128>), grid, block, 0, st, q);
    else if (pl.tm == 96) hipLaunchKernelGGL((pw_gemm_kernel<T, 2, 96>), grid, block, 0, st, q);
    else if (pl.tm == 64) hipLaunchKernelGGL((pw_gemm_kernel<T, 1, 64>), grid, block, 0, st, q);
    else hipLaunchKernelGGL((pw_gemm_kernel<T, 1, 48>), grid, block, 0, st, q);
  }
  MI_LAUNCH_CHECK();
  return MI_OK;
}

static int pw_check(const mi_pw_desc* d) {
  MI_CHECK_ARG(d && d->x1 && d->w && d->y, "pw_gemm: null pointer");
  MI_CHECK_ARG(d->m > 0 && d->n > 0 && d->k1 > 0 && d->k2 >= 0 && d->batch > 0 && d->groups > 0, "pw_gemm: bad shape");
  MI_CHECK_ARG((d->k2 == 0) == (d->x2 == nullptr), "pw_gemm: x2/k2 mismatch");
  MI_CHECK_ARG(d->dtype == MI_F32 || d->dtype == MI_BF16, "pw_gemm: bad dtype %d", d->dtype);
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" size_t mi_pw_gemm_workspace(const mi_pw_desc* d) {
  if (!d || d->m <= 0 || d->k1 <= 0 || d->k2 < 0 || d->batch <= 0 || d->groups <= 0) return 0;
  // the wave-owned forms tile the packed image differently; which form runs depends on pointer alignment at call time, and module
  // entry points size their workspaces before they see the pointers: cover both
  const size_t a = pw_plan(d).bytes, b = pw_plan(d, false).bytes;
  const size_t c = (d->dtype == MI_BF16 && d->k1 > 128) ? align_up(pw_lds_pack_bytes(d), 256) : 0;   // the LDS-tiled kernel's image
  const size_t ab = a > b ? a : b;
  return ab > c ? ab : c;
}

extern "C" int mi_pw_gemm(const mi_pw_desc* d, void* ws, void* stream) {
  MI_TRY(pw_check(d));
  MI_CHECK_ARG(ws && aligned16(ws), "pw_gemm: workspace missing or not 16-byte aligned");
  if (pw_lds_ok(d)) return pw_lds_launch(d, ws, (hipStream_t)stream);     // deep K: the LDS-tiled kernel (pw_lds.hip)
  PwK k;
  k.x1 = d->x1; k.x1_bs = d->x1_bs; k.x1_gs = d->x1_gs; k.k1 = d->k1;
  k.x2 = d->x2; k.x2_bs = d->x2_bs; k.x2_gs = d->x2_gs; k.k2 = d->k2;
  k.w = d->w; k.w_bs = d->w_bs; k.w_gs = d->w_gs; k.w_sm = d->w_sm; k.w_sk = d->w_sk;
  k.bias = d->bias; k.bias_gs = d->bias_gs;
  k.r = d->r; k.r_bs = d->r_bs; k.r_gs = d->r_gs;
  k.y = d->y; k.y_bs = d->y_bs; k.y_gs = d->y_gs;
  k.m = d->m; k.n = d->n; k.groups = d->groups;
  k.y2 = d->y2; k.y2_bs = d->y2_bs; k.y2_gs = d->y2_gs; k.y_split = d->y_split;
  const bool ok = pw_vec_ok(d);
  k.vec_ok = ok ? 1 : 0;
  const PwPlan pl = pw_plan(d);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == MI_F32) return pw_launch<float>(d, k, pl, ws, st);
  return pw_launch<bf16>(d, k, pl, ws, st);
}

extern "C" int mi_pw_gemm_ln_ok(const mi_pw_desc* d) {
  if (!d || pw_check(d) != MI_OK || d->k2 != 0 || d->groups != 1) return 0;
  // the two X-resident forms; the W-streamed one up to K = 128 only: at K = 129 .. 192 the tile (96 load registers) plus the
  // statistics spill (60 / 252 bytes of scratch per lane) and the separate LayerNorm kernel is the better choice
  const PwPlan pl = pw_plan(d);
  return (pl.wave == 1 || (pl.wave == 3 && pl.k_chunks <= 4)) ? 1 : 0;
}

extern "C" int mi_pw_gemm_split_ok(const mi_pw_desc* d) {
  if (!d || pw_check(d) != MI_OK || d->dtype != MI_BF16 || d->r) return 0;
  return pw_plan(d).wave != 0 ? 1 : 0;
}

extern "C" int mi_pw_gemm_f8_ok(const mi_pw_desc* d) {
  if (!d || pw_check(d) != MI_OK || d->dtype != MI_BF16) return 0;
  return pw_plan(d).wave != 0 ? 1 : 0;
}

extern "C" int mi_pw_cache_enable(void* buf, size_t bytes, const void* params_lo, const void* params_hi) {
  std::lock_guard<std::mutex> lk(g_pwc.mu);
  g_pwc.jobs.clear();
  g_pwc.valid = false; g_pwc.dirty = false; g_pwc.used = 0; g_pwc.base = nullptr; g_pwc.bytes = 0;
  g_pwc.lo = (const char*)params_lo; g_pwc.hi = (const char*)params_hi;
  if (!buf) return MI_OK;  // disable
  g_pwc.table_bytes = align_up(PWC_TABLE_ENTRIES * sizeof(PackJob), 256);
  MI_CHECK_ARG(aligned16(buf) && bytes > g_pwc.table_bytes + 4096, "pw_cache_enable: buffer too small or misaligned");
  g_pwc.base = (unsigned char*)buf; g_pwc.bytes = bytes; g_pwc.used = g_pwc.table_bytes;  // the job table sits at the head
  return MI_OK;
}

extern "C" int mi_pw_cache_invalidate(void) {
  std::lock_guard<std::mutex> lk(g_pwc.mu);
  g_pwc.valid = false;
  return MI_OK;
}

// 1 while the cache holds weights it has not packed yet (registered since the last refresh) or was invalidated
extern "C" int mi_pw_cache_pending(void) {
  std::lock_guard<std::mutex> lk(g_pwc.mu);
  return (g_pwc.base && !g_pwc.jobs.empty() && (g_pwc.dirty || !g_pwc.valid)) ? 1 : 0;
}

extern "C" int mi_pw_cache_refresh(void* stream) {
  std::lock_guard<std::mutex> lk(g_pwc.mu);
  if (!g_pwc.base || g_pwc.jobs.empty()) return MI_OK;
  hipStream_t st = (hipStream_t)stream;
  if (g_pwc.dirty) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    MI_CHECK_ARG(!(hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone),
                 "pw_cache_refresh: new weights were registered; refresh once outside stream capture first");
    MI_CHECK_HIP(hipMemcpyAsync(g_pwc.base, g_pwc.jobs.data(), g_pwc.jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice, st));
    MI_CHECK_HIP(hipStreamSynchronize(st));  // the host vector may be reallocated by a later registration
    g_pwc.dirty = false;
  }
  int64_t most = 0;
  double bytes = 0.0;
  for (const PackJob& j : g_pwc.jobs) {
    const int64_t total = j.slice_elems / j.chunk_elems * j.tm * PW_KC;
    if (total > most) most = total;
    bytes += 4.0 * j.M * j.K * j.slices + (double)j.slices * j.slice_elems * dtype_size(j.dtype);
  }
  int gx = cdiv(most, 256 * 4);
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  ProfScope ps(st, K_PW_PACK, bytes, 0.0);
  hipLaunchKernelGGL(pw_pack_table_kernel, dim3(gx, (unsigned)g_pwc.jobs.size()), dim3(256), 0, st, (const PackJob*)g_pwc.base);
  MI_LAUNCH_CHECK();
  g_pwc.valid = true;
  return MI_OK;
}
