// Internal launchers shared between translation units (not part of the C-ABI).
#pragma once
#include "common.h"

namespace mi {
int launch_attn_fold(const float* graw, const float* ss, const float* temperature, const float* wo, float* P, float* A,
                     float* nrm, float* M, int B, int C, int heads, hipStream_t st);
int launch_attn_bwd_small(const float* dM, const float* A, const float* P, const float* nrm, const float* temperature,
                          const float* wo, float* dwo_part, float* dtemp_part, float* wdq, float* wdk, int B, int C,
                          int heads, hipStream_t st);
size_t chan_sum_workspace(int C, int64_t N);
int launch_chan_sum(const void* x, float* out, int B, int C, int64_t N, int dtype, int accumulate, void* ws, hipStream_t st);
}  // namespace mi
