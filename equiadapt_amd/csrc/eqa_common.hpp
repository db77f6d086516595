// Shared by the translation units of libeqa_hip.so (gfx950 / CDNA4, wave64).  C ABI: include/eqa_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "eqa_hip.h"

namespace eqa {

constexpr int kThreads = 256;  // block size of every kernel that does not say otherwise
constexpr int kXcd = 8;        // MI355X: 8 XCDs, block b runs on XCD b % 8, each XCD has a private L2

// window sums (pooling.hip) and the Winograd output transform that emits them directly (winograd.hip)
constexpr int kMaxWinK = 8;
constexpr int kWsMaxBorder = kMaxWinK - 1;  // k - 1 <= 7
constexpr int kFinCh = 32;                  // channels per block of the finalize kernel

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH; }

// pooling.hip: part (B, nseg, C, 1 + 2(k-1)) row segments -> S (B, C, k, k) fp64; used by eqa_window_sums_nhwc and by the
// Winograd output transform fused with the window sums
int launch_window_sums_nhwc_finalize(const float* part, double* S, int B, int C, int k, int nseg, hipStream_t stream, int sub = 1);

}  // namespace eqa

using namespace eqa;
