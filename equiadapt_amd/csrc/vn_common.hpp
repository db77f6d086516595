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

// Backward of one VN layer output  y = gate(u * nbn, d),  u = q / n,  n = |q| + EPS,  nbn = n * scale + shift  (scale = gamma *
// rstd, shift = beta - mean * scale; vector_neuron_layers.py:251-273, :303-324): forward values and the gradients that do not
// need batch-wide sums.
struct VnGrad {
  V3 q, u, d, g_qn, g_d;  // pre-norm vector, its direction q / n, gate direction, dL/d(normalised vector), dL/d(gate)
  float nr, nbn, g_nbn;   // n = |q| + EPS, batch-normalised norm, dL/d(nbn)
};
__device__ __forceinline__ VnGrad vn_gate_grad(const V3& q, const V3& d, float scale, float shift, const V3& g_out) {
  VnGrad r;
  r.q = q;
  r.nr = sqrtf(dot3(r.q, r.q)) + kVnEps;
  const float inv_n = 1.0f / r.nr;
  r.u = v3(r.q.x * inv_n, r.q.y * inv_n, r.q.z * inv_n);
  r.nbn = r.nr * scale + shift;
  const V3 qn = v3(r.u.x * r.nbn, r.u.y * r.nbn, r.u.z * r.nbn);
  r.d = d;
  const float dp = dot3(qn, r.d);
  if (dp >= 0.0f) {  // kept as is
    r.g_qn = g_out;
    r.g_d = v3(0.f, 0.f, 0.f);
  } else {           // out = qn - alpha d, alpha = <qn, d> / (|d|^2 + EPS)
    const float rr = 1.0f / (dot3(r.d, r.d) + kVnEps);
    const float alpha = dp * rr;
    const float g_alpha = -dot3(g_out, r.d);
    const float ga_r = g_alpha * rr;
    r.g_qn = v3(g_out.x + ga_r * r.d.x, g_out.y + ga_r * r.d.y, g_out.z + ga_r * r.d.z);
    r.g_d = v3(-alpha * g_out.x + ga_r * (qn.x - 2.0f * alpha * r.d.x), -alpha * g_out.y + ga_r * (qn.y - 2.0f * alpha * r.d.y),
               -alpha * g_out.z + ga_r * (qn.z - 2.0f * alpha * r.d.z));
  }
  r.g_nbn = dot3(r.g_qn, r.u);
  return r;
}
// The same layer without a gate (VNBatchNorm alone): y = u * nbn
__device__ __forceinline__ VnGrad vn_norm_grad(const V3& q, float scale, float shift, const V3& g_out) {
  VnGrad r;
  r.q = q;
  r.nr = sqrtf(dot3(q, q)) + kVnEps;
  const float inv_n = 1.0f / r.nr;
  r.u = v3(q.x * inv_n, q.y * inv_n, q.z * inv_n);
  r.nbn = r.nr * scale + shift;
  r.d = v3(0.f, 0.f, 0.f);
  r.g_qn = g_out;
  r.g_d = r.d;
  r.g_nbn = dot3(g_out, r.u);
  return r;
}
// dL/dq once the batch sums are known: through the direction u = q / n and through the norm,
//   g_n = gamma rstd (g_nbn - m1 - nhat m2)   (m1 = sum g_nbn / M, m2 = sum g_nbn nhat / M; both 0 with running statistics)
//   g_q = (g_u - u <g_u, q> / |q|) / n + g_n q / |q|,   g_u = g_qn * nbn
// sc = gamma * rstd.  Idle threads (duplicates of the last point) must not pick up the batch terms -m1 - nhat m2.
__device__ __forceinline__ V3 vn_norm_input_grad(const VnGrad& r, float sc, float mu, float rs, float m1, float m2, bool active) {
  const float nhat = (r.nr - mu) * rs;
  const float g_n = active ? sc * (r.g_nbn - m1 - nhat * m2) : 0.0f;
  const float qlen = fmaxf(r.nr - kVnEps, 1e-30f);
  const V3 g_u = v3(r.g_qn.x * r.nbn, r.g_qn.y * r.nbn, r.g_qn.z * r.nbn);
  const float proj = dot3(g_u, r.q) / qlen;
  const float inv_n = 1.0f / r.nr, gq = g_n / qlen;
  return v3((g_u.x - r.u.x * proj) * inv_n + gq * r.q.x, (g_u.y - r.u.y * proj) * inv_n + gq * r.q.y,
            (g_u.z - r.u.z * proj) * inv_n + gq * r.q.z);
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
