// Training-data and evaluation kernels around the network:
//   * mi_patch_batch   the sample pipeline of AIOTrainDataset.__getitem__ for the denoise tasks
//                      (MoCE-IR-main/src/data/dataset_utils.py:156-165; degradation_utils.py:21-24; image_utils.py random_augmentation):
//                      crop a P x P patch out of a decoded uint8 HWC image, apply one of the 8 dihedral augmentations, add
//                      sigma * N(0,1) on the uint8 grid (clip to [0,255], truncate to uint8), ToTensor (/255, CHW) - one
//                      launch for the whole batch, straight into the network's activation dtype, from a device-resident
//                      pool of decoded images.  No DataLoader worker touches pixels.
//   * mi_psnr_ssim     per-image PSNR (10 log10(1/MSE) on values clipped to [0,1]) and SSIM as the reference computes them
//                      through scikit-image (AdaIR-main/utils/val_utils.py:50-64: data_range 1, 7x7 uniform window,
//                      K1 0.01, K2 0.03, sample covariance, border of 3 pixels excluded, channels averaged).
#include <math.h>

#include "internal.h"

namespace mi {

struct PatchArgs {
  const unsigned char* pool; const int64_t* src_off; const int* src_h; const int* src_w;
  const int* sample; const int* top; const int* left; const int* mode; const float* sigma; const float* noise;
  void* clean; void* degraded; int B, P, dtype;
};

// source coordinates (row, col) inside the un-augmented patch of output pixel (y, x); modes = image_utils.data_augmentation
__device__ __forceinline__ void aug_src(int mode, int y, int x, int P, int& sy, int& sx) {
  const int q = P - 1;
  switch (mode) {
    case 0: sy = y; sx = x; break;              // original
    case 1: sy = q - y; sx = x; break;          // flipud
    case 2: sy = x; sx = q - y; break;          // rot90 (counter-clockwise)
    case 3: sy = x; sx = y; break;              // rot90 + flipud  (= transpose)
    case 4: sy = q - y; sx = q - x; break;      // rot180
    case 5: sy = y; sx = q - x; break;          // rot180 + flipud (= fliplr)
    case 6: sy = q - x; sx = y; break;          // rot270
    default: sy = q - x; sx = q - y; break;     // rot270 + flipud (= anti-transpose)
  }
}

template <typename T>
__global__ __launch_bounds__(256) void patch_batch_kernel(PatchArgs a) {
  const int64_t per = (int64_t)3 * a.P * a.P, total = per * a.B;
  T* clean = reinterpret_cast<T*>(a.clean);
  T* degr = reinterpret_cast<T*>(a.degraded);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / per);
    int64_t r = i - (int64_t)b * per;
    const int c = (int)(r / ((int64_t)a.P * a.P));
    r -= (int64_t)c * a.P * a.P;
    const int y = (int)(r / a.P), x = (int)(r - (int64_t)y * a.P);
    int sy, sx;
    aug_src(a.mode[b], y, x, a.P, sy, sx);
    const int s = a.sample[b];
    const int64_t src = a.src_off[s] + ((int64_t)(a.top[b] + sy) * a.src_w[s] + (a.left[b] + sx)) * 3 + c;
    const float v = (float)a.pool[src];
    if (clean) clean[i] = Cvt<T>::from(v / 255.0f);                        // ToTensor: .div(255)
    if (degr) {
      // fp64 like numpy's  clean + noise * sigma  (an fp32 product lands on the other side of an integer ~1e-4 of the time)
      double d = (double)v + (double)a.noise[i] * (double)a.sigma[b];
      d = fmin(fmax(d, 0.0), 255.0);
      degr[i] = Cvt<T>::from((float)trunc(d) / 255.0f);                    // np.clip(...).astype(np.uint8) truncates
    }
  }
}

// ---------------------------------------------------------------------------------------------------- PSNR / SSIM
constexpr int SS_T = 32;            // output tile
constexpr int SS_R = 3;             // window radius (7 x 7)
constexpr int SS_L = SS_T + 2 * SS_R;

// blockIdx: (tile x, tile y, plane).  part[plane][tile][0] = sum of squared differences over the tile (all pixels),
// part[..][1] = sum of the SSIM map over the tile's interior pixels (>= 3 from every border)
template <typename T>
__global__ __launch_bounds__(256) void psnr_ssim_kernel(const T* __restrict__ xr, const T* __restrict__ yr, float* __restrict__ part,
                                                        int H, int W) {
  __shared__ float xs[SS_L][SS_L + 1], ys[SS_L][SS_L + 1];
  __shared__ float red[2][4];
  const int t = threadIdx.x;
  const int64_t plane = blockIdx.z;
  const T* xp = xr + plane * (int64_t)H * W;
  const T* yp = yr + plane * (int64_t)H * W;
  const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
  for (int i = t; i < SS_L * SS_L; i += 256) {
    const int r = i / SS_L, c = i - r * SS_L;
    const int Y = y0 + r - SS_R, X = x0 + c - SS_R;
    const bool in = Y >= 0 && Y < H && X >= 0 && X < W;
    xs[r][c] = in ? fminf(fmaxf(to_f32(xp[(int64_t)Y * W + X]), 0.f), 1.f) : 0.f;
    ys[r][c] = in ? fminf(fmaxf(to_f32(yp[(int64_t)Y * W + X]), 0.f), 1.f) : 0.f;
  }
  __syncthreads();
  float se = 0.f, ss = 0.f;
  for (int i = t; i < SS_T * SS_T; i += 256) {
    const int r = i / SS_T, c = i - r * SS_T;
    const int Y = y0 + r, X = x0 + c;
    if (Y >= H || X >= W) continue;
    const float d = xs[r + SS_R][c + SS_R] - ys[r + SS_R][c + SS_R];
    se += d * d;
    if (Y < SS_R || Y >= H - SS_R || X < SS_R || X >= W - SS_R) continue;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = 0; dy < 7; ++dy)
#pragma unroll
      for (int dx = 0; dx < 7; ++dx) {
        const float a = xs[r + dy][c + dx], b = ys[r + dy][c + dx];
        sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
      }
    const float n = 49.f, cn = n / (n - 1.f);                               // sample covariance
    const float ux = sx / n, uy = sy / n;
    const float vx = cn * (sxx / n - ux * ux), vy = cn * (syy / n - uy * uy), vxy = cn * (sxy / n - ux * uy);
    const float C1 = 1e-4f, C2 = 9e-4f;                                     // (0.01 R)^2, (0.03 R)^2 with R = 1
    ss += ((2.f * ux * uy + C1) * (2.f * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
  }
  se = wave_sum(se); ss = wave_sum(ss);
  if ((t & 63) == 0) { red[0][t >> 6] = se; red[1][t >> 6] = ss; }
  __syncthreads();
  if (t == 0) {
    const int64_t tile = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, ntiles = (int64_t)gridDim.x * gridDim.y;
    part[(plane * ntiles + tile) * 2 + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[(plane * ntiles + tile) * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// one thread per image: fixed-order sums over its planes and tiles -> PSNR, SSIM
__global__ void psnr_ssim_finish_kernel(const float* __restrict__ part, float* __restrict__ psnr, float* __restrict__ ssim, int B,
                                        int C, int64_t ntiles, int H, int W) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double se = 0.0, ss = 0.0;
  for (int64_t i = 0; i < (int64_t)C * ntiles; ++i) { se += part[((int64_t)b * C * ntiles + i) * 2]; ss += part[((int64_t)b * C * ntiles + i) * 2 + 1]; }
  const double mse = se / ((double)C * H * W);
  psnr[b] = (float)(10.0 * log10(1.0 / mse));
  ssim[b] = (float)(ss / ((double)C * (H - 2 * SS_R) * (W - 2 * SS_R)));
}

}  // namespace mi

using namespace mi;

extern "C" int mi_patch_batch(const unsigned char* pool, const int64_t* src_off, const int* src_h, const int* src_w,
                              const int* sample, const int* top, const int* left, const int* mode, const float* sigma,
                              const float* noise, void* clean, void* degraded, int B, int P, int dtype, void* stream) {
  MI_CHECK_ARG(pool && src_off && src_h && src_w && sample && top && left && mode && B > 0 && P > 0, "patch_batch: bad arguments");
  MI_CHECK_ARG(clean || degraded, "patch_batch: nothing to write");
  MI_CHECK_ARG(!degraded || (sigma && noise), "patch_batch: the degraded output needs sigma and noise");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "patch_batch: bad dtype %d", dtype);
  PatchArgs a{pool, src_off, src_h, src_w, sample, top, left, mode, sigma, noise, clean, degraded, B, P, dtype};
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)3 * P * P * B;
  int blocks = cdiv(total, 256 * 4);
  if (blocks > 16384) blocks = 16384;
  ProfScope ps(st, K_CAST, (double)total * (1.0 + 4.0 + 2.0 * dtype_size(dtype)), 4.0 * total);
  if (dtype == MI_F32) hipLaunchKernelGGL((patch_batch_kernel<float>), dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((patch_batch_kernel<bf16>), dim3(blocks), dim3(256), 0, st, a);
  MI_LAUNCH_CHECK();
  return MI_OK;
}

extern "C" size_t mi_psnr_ssim_workspace(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return align_up((size_t)B * C * cdiv(H, SS_T) * cdiv(W, SS_T) * 2 * sizeof(float), 256);
}

extern "C" int mi_psnr_ssim(const void* restored, const void* clean, float* psnr, float* ssim, int B, int C, int H, int W,
                            int dtype, void* ws, void* stream) {
  MI_CHECK_ARG(restored && clean && psnr && ssim && ws && B > 0 && C > 0, "psnr_ssim: bad arguments");
  MI_CHECK_ARG(H >= 7 && W >= 7, "psnr_ssim: images must be at least 7 x 7 (the SSIM window)");
  MI_CHECK_ARG(dtype == MI_F32 || dtype == MI_BF16, "psnr_ssim: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(W, SS_T), cdiv(H, SS_T), (unsigned)((int64_t)B * C));
  MI_CHECK_ARG((int64_t)B * C < 65536 * 16 && grid.y < 65536, "psnr_ssim: grid too large");
  const double px = (double)B * C * H * W;
  ProfScope ps(st, K_L1, 2.0 * px * dtype_size(dtype), 260.0 * px);
  if (dtype == MI_F32) hipLaunchKernelGGL((psnr_ssim_kernel<float>), grid, dim3(256), 0, st, (const float*)restored, (const float*)clean, (float*)ws, H, W);
  else hipLaunchKernelGGL((psnr_ssim_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)restored, (const bf16*)clean, (float*)ws, H, W);
  MI_LAUNCH_CHECK();
  hipLaunchKernelGGL(psnr_ssim_finish_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, (const float*)ws, psnr, ssim, B, C,
                     (int64_t)grid.x * grid.y, H, W);
  MI_LAUNCH_CHECK();
  return MI_OK;
}
