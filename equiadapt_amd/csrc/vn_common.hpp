// Shared device code of the VNSmall kernels (pointcloud.hip: fused eval forward; vnsmall_train.hip: training passes).
#pragma once
#include "eqa_common.hpp"

namespace {

constexpr int kVnC = 21, kVnK = 20, kVnThreads = 128, kVnParams = 1310;
#ifndef EQA_VN_MIN_BLOCKS
#define EQA_VN_MIN_BLOCKS 3  // waves per SIMD the register allocation must allow (measured 2: 481 k, 3: 531 k, 4: 416 k clouds/s)
#endif
constexpr int kVnQueue = 12;  // pending kNN candidates per thread (LDS, 8 bytes each)
constexpr float kVnEps = 1e-6f;

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ float dot3(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// VN batch-norm (eval): q * BN(|q| + EPS) / (|q| + EPS);  then direction-gated ReLU with slope 0
__device__ __forceinline__ V3 vn_bn(V3 q, float scale, float shift) {
  const float n = sqrtf(dot3(q, q)) + kVnEps;
  const float r = (n * scale + shift) / n;
  return v3(q.x * r, q.y * r, q.z * r);
}
__device__ __forceinline__ V3 vn_relu(V3 q, const V3& d) {
  const float dp = dot3(q, d);
  if (dp < 0.0f) {
    const float t = dp / (dot3(d, d) + kVnEps);
    q.x -= t * d.x; q.y -= t * d.y; q.z -= t * d.z;
  }
  return q;
}

// A cloud staged in LDS as float4 (x, y, z, |p|^2): one ds_read_b128 per kNN candidate.  x: (3, N) of one cloud.
__device__ __forceinline__ void vn_stage_cloud(const float* __restrict__ xb, int N, int Npad, float4* pts, int tid) {
  for (int i = tid; i < Npad; i += kVnThreads) {
    if (i < N) {
      const float a = xb[i], c = xb[N + i], d = xb[2 * (size_t)N + i];
      pts[i] = make_float4(a, c, d, a * a + c * c + d * d);  // torch.sum(x**2, dim=1)
    } else {
      pts[i] = make_float4(0.f, 0.f, 0.f, INFINITY);  // padding: value -inf, never selected
    }
  }
}

// The k = 20 nearest neighbours of the point (ctr, cn = |ctr|^2) among pts[0, Npad): indices in bi, best first.
// queue: this thread's slot 0 of the pending-candidate queue in LDS (slot s at queue[s * kVnThreads]).
__device__ __forceinline__ void vn_knn(const float4* pts, int Npad, float2* queue, const V3& ctr, float cn, int (&bi)[kVnK]) {
  // ---- kNN: k largest of  -|xj|^2 + 2 xi.xj - |xi|^2  (the reference's expansion, equivariant_networks.py:28-30),
  // kept sorted (descending) in registers; strict '>' so that the earlier index wins ties.
  // The sorted insertion is a ~100-instruction chain that the whole wave executes whenever ANY of its 64 points
  // accepts a candidate -- which is true for ~870 of the 1024 candidates although each point accepts only ~93 (first
  // version: 181 k VALU instructions per wave, ~145 k of them here, 11 % of the lanes doing useful work).  So a
  // candidate that beats the point's current 20th score is only APPENDED to a small per-thread queue in LDS (3
  // instructions), and the queues are drained -- in index order, each entry re-tested against the then-current
  // threshold, i.e. the same result as immediate insertion -- when any lane's queue could overflow on the next group:
  // ~185 chain executions per wave instead of ~870.
  float bv[kVnK];
#pragma unroll
  for (int t = 0; t < kVnK; ++t) { bv[t] = -INFINITY; bi[t] = 0; }
  auto insert = [&](float val, int j) {
    if (val > bv[kVnK - 1]) {
      float cv = val;
      int ci = j;
#pragma unroll
      for (int t = 0; t < kVnK; ++t) {
        const bool sw = cv > bv[t];
        const float tv = bv[t];
        const int ti = bi[t];
        bv[t] = sw ? cv : tv;
        bi[t] = sw ? ci : ti;
        cv = sw ? tv : cv;
        ci = sw ? ti : ci;
      }
    }
  };
  auto score = [&](const float4& p) {
    const float inner = -2.0f * (ctr.x * p.x + ctr.y * p.y + ctr.z * p.z);
    return (-p.w - inner) - cn;
  };
  int cnt = 0;
  auto drain = [&]() {
#pragma unroll
    for (int s = 0; s < kVnQueue; ++s) {
      if (s < cnt) {
        const float2 e = queue[s * kVnThreads];
        insert(e.x, __float_as_int(e.y));
      }
    }
    cnt = 0;
  };
  auto offer = [&](float v, int j) {
    if (v > bv[kVnK - 1]) {
      queue[cnt * kVnThreads] = make_float2(v, __int_as_float(j));
      ++cnt;
    }
  };
  for (int j = 0; j < Npad; j += 4) {
    const float4 p0 = pts[j], p1 = pts[j + 1], p2 = pts[j + 2], p3 = pts[j + 3];
    offer(score(p0), j);
    offer(score(p1), j + 1);
    offer(score(p2), j + 2);
    offer(score(p3), j + 3);
    if (__any(cnt > kVnQueue - 4)) drain();  // wave-uniform
  }
  drain();
}

}  // namespace
