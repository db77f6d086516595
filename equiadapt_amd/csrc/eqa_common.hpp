// Shared by the translation units of libeqa_hip.so (gfx950 / CDNA4, wave64).  C ABI: include/eqa_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "eqa_hip.h"

namespace eqa {

constexpr int kThreads = 256;  // block size of every kernel that does not say otherwise
constexpr int kXcd = 8;        // MI355X: 8 XCDs, block b runs on XCD b % 8, each XCD has a private L2

// window sums (pooling.hip) and the Winograd output transform that emits them directly (winograd.hip)
constexpr int kMaxWinK = 10;                // (9: the reference tutorial's ESCNN canonicalizer, kernel_size = 9)
constexpr int kWsMaxBorder = kMaxWinK - 1;  // k - 1 <= 9: register arrays of the kernels instantiated for 8 < k <= 10
constexpr int kWsSmallBorder = 7;           // ... and of the instantiation every k <= 8 keeps (its register / LDS budget unchanged)
constexpr int kFinCh = 32;                  // channels per block of the finalize kernel

// Sum over the 64 lanes of a wave, returned to every lane (wave-uniform: it comes back through a scalar register).
// PRECONDITION: all 64 lanes active (EXEC all ones) -- the total is read from lane 63 after the row broadcasts, and an inactive
// lane would contribute a stale register; every call site is wave-uniform control flow (no early return, no divergent branch
// around it).  The row_bcast15 / row_bcast31 controls exist on gfx9 / CDNA only, which is the one target of this library.
// Data-parallel-primitive adds instead of __shfl_xor: the shuffles compile to ds_bpermute_b32, a six-deep chain of LDS-crossbar
// round trips per sum; the DPP modifiers ride on the v_add itself.  Inside a row of 16 lanes: quad swaps, then the two row
// mirrors; across rows: row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3; lane 63 holds the total.
#define EQA_DPP_ADD(v, ctrl, row_mask) \
  (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (row_mask), 0xf, false))
__device__ __forceinline__ float wave_sum_f(float v) {
  EQA_DPP_ADD(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  EQA_DPP_ADD(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  EQA_DPP_ADD(v, 0x141, 0xf);  // row_half_mirror
  EQA_DPP_ADD(v, 0x140, 0xf);  // row_mirror
  EQA_DPP_ADD(v, 0x142, 0xa);  // row_bcast15 -> rows 1, 3 (masked rows add the `old` operand, 0)
  EQA_DPP_ADD(v, 0x143, 0xc);  // row_bcast31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef EQA_DPP_ADD


// (i % Q, i / Q) along a grid-stride loop without a division per trip.  The loops used `i % Q` on a size_t index: a 64-bit division
// is ~100 vector instructions -- more than the 16 bytes of data a trip moves are worth.  One 32-bit division pair at the start (the
// first index and the stride are below 2^32 for every launch in this library: <= 2^20 threads), additions with a carry afterwards.
struct QuadWalk {
  unsigned q, dq, Q;
  size_t row, drow;
  __device__ __forceinline__ QuadWalk(size_t i0, size_t stride, int Q_) {
    Q = (unsigned)Q_;
    const unsigned i = (unsigned)i0, st = (unsigned)stride;
    row = i / Q;
    q = i - (unsigned)row * Q;
    drow = st / Q;
    dq = st - (unsigned)drow * Q;
  }
  __device__ __forceinline__ void next() {
    q += dq;
    row += drow;
    if (q >= Q) {
      q -= Q;
      ++row;
    }
  }
};

extern int g_vn_kernel_choice;  // pointcloud.hip; eqa_set_option key 1
extern int g_cgemm_bf16_form;   // cgemm3m_bf16.hip; eqa_set_option key 2

// Counter-based hash for the dropout mask of the canonicalization network's hidden blocks (one draw per element, reproducible from
// (seed, element index)): shared by batchnorm.hip (which writes / recomputes the mask) and pooling.hip (the window sums that
// consume a hidden block without its output ever being written).  i = index of the channel QUAD in the channels-last map.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
// ALWAYS: the caller knows drop_threshold != 0 -- no (uniform, but scheduling-region-splitting) branch per element
template <bool ALWAYS = false>
__device__ __forceinline__ void dropout_keep_quad(size_t i, uint32_t drop_threshold, uint32_t seed, bool (&keep)[4]) {
  const uint32_t base = mix32((uint32_t)i * 0x9E3779B1u + seed) ^ (uint32_t)(i >> 32);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool kept = mix32(base + (uint32_t)k * 0x632BE5ABu) >= drop_threshold;
    keep[k] = ALWAYS ? kept : (!drop_threshold || kept);
  }
}
inline uint32_t dropout_threshold(float drop_p) {
  return drop_p > 0.0f ? (uint32_t)std::min<double>((double)drop_p * 4294967296.0, 4294967295.0) : 0u;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH; }

// Kernels that need more dynamic LDS than a launch's default 64 KB: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the
// CURRENT device only, so the answer is cached per (kernel, device ordinal) -- a function-local `static const bool` would set it on
// the device of the first call and leave every other GPU of the process with launch failures.
inline bool allow_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, bool> seen;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(kernel, dev);
  const auto it = seen.find(key);
  if (it != seen.end()) return it->second;
  const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  seen[key] = ok;
  return ok;
}

// pooling.hip: part (B, nseg, C, 1 + 2(k-1)) row segments -> S (B, C, k, k) fp64; used by eqa_window_sums_nhwc and by the
// Winograd output transform fused with the window sums
int launch_window_sums_nhwc_finalize(const float* part, double* S, int B, int C, int k, int nseg, hipStream_t stream, int sub = 1);

}  // namespace eqa

using namespace eqa;
