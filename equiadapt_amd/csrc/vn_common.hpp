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

// Square root and reciprocal of the VN layers.  The correctly rounded forms (sqrtf, IEEE division) expand to ~10 instructions
// each and made up a third of the per-(edge, channel) work of conv_pos (76 VALU instructions, two divisions and a root); the
// hardware's v_sqrt_f32 / v_rcp_f32 are one instruction and 1 ulp: the layer outputs move by ~1e-7 relative, two orders below
// the 1e-5 point-cloud tolerance and below the fp32 noise of the reference's own evaluation (6e-5 at 8 x 1024
// points in training mode, tools/diag/vn_train_noise.py).  -DEQA_VN_IEEE_MATH=1 restores the correctly rounded forms.
#ifndef EQA_VN_IEEE_MATH
#define EQA_VN_IEEE_MATH 0
#endif
__device__ __forceinline__ float vn_sqrt(float x) { return EQA_VN_IEEE_MATH ? sqrtf(x) : __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float vn_rcp(float x) { return EQA_VN_IEEE_MATH ? 1.0f / x : __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float vn_div(float a, float b) { return EQA_VN_IEEE_MATH ? a / b : a * __builtin_amdgcn_rcpf(b); }

// VN batch-norm (eval): q * BN(|q| + EPS) / (|q| + EPS);  then direction-gated ReLU with slope 0
__device__ __forceinline__ V3 vn_bn(V3 q, float scale, float shift) {
  const float n = vn_sqrt(dot3(q, q)) + kVnEps;
  const float r = vn_div(n * scale + shift, n);
  return v3(q.x * r, q.y * r, q.z * r);
}
__device__ __forceinline__ V3 vn_relu(V3 q, const V3& d) {
  const float dp = dot3(q, d);
  if (dp < 0.0f) {
    const float t = vn_div(dp, dot3(d, d) + kVnEps);
    q.x -= t * d.x; q.y -= t * d.y; q.z -= t * d.z;
  }
  return q;
}
// the same without a branch (straight-line code for the unrolled edge loops: some lane of a wave takes the branch anyway)
__device__ __forceinline__ V3 vn_relu_sel(V3 q, const V3& d) {
  const float dp = dot3(q, d);
  const float t = dp < 0.0f ? vn_div(dp, dot3(d, d) + kVnEps) : 0.0f;
  return v3(q.x - t * d.x, q.y - t * d.y, q.z - t * d.z);
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
  r.nr = vn_sqrt(dot3(r.q, r.q)) + kVnEps;
  const float inv_n = vn_rcp(r.nr);
  r.u = v3(r.q.x * inv_n, r.q.y * inv_n, r.q.z * inv_n);
  r.nbn = r.nr * scale + shift;
  const V3 qn = v3(r.u.x * r.nbn, r.u.y * r.nbn, r.u.z * r.nbn);
  r.d = d;
  const float dp = dot3(qn, r.d);
  if (dp >= 0.0f) {  // kept as is
    r.g_qn = g_out;
    r.g_d = v3(0.f, 0.f, 0.f);
  } else {           // out = qn - alpha d, alpha = <qn, d> / (|d|^2 + EPS)
    const float rr = vn_rcp(dot3(r.d, r.d) + kVnEps);
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
  r.nr = vn_sqrt(dot3(q, q)) + kVnEps;
  const float inv_n = vn_rcp(r.nr);
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
  const float inv_q = vn_rcp(qlen);
  const float proj = dot3(g_u, r.q) * inv_q;
  const float inv_n = vn_rcp(r.nr), gq = g_n * inv_q;
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

// ------------------------------------------------------------------------------------------------------------------------
// Four lanes per point ("quad" kernels).  One thread per point leaves a batch of 64 clouds with one wave per SIMD, each wave a
// single dependent instruction stream (215 k clouds/s at B = 64 against 555 k at B = 2048).  Here the four lanes of a DPP quad share
// a point: lane q scans the candidates j = q (mod 4) and the point's sorted neighbour list is DISTRIBUTED over the quad -- lane q
// holds ranks [SEG q, SEG q + SEG) -- so an insertion is a SEG-step chain per lane instead of a 4 SEG-step one, the threshold
// (the list's last entry, lane 3) is exact for all four scanners, there is no merge step, and at the end every lane already
// owns its share of the point's edges.  SEG = 5 serves k <= 20, SEG = 8 serves k <= 32; a runtime k < 4 SEG uses the ranks < k.
//
// Insertion of a candidate (v, j) that all four lanes see (vector_neuron kNN = k largest scores, strict '>': on equal scores the
// entry inserted first stays in front):  with pv = the OLD last entry of the previous lane's segment (+inf for lane 0),
//   v > pv : the candidate lands before this lane's segment, whose entries all move down by one: the incoming element is pv;
//   else   : the incoming element is the candidate itself, placed in front of the first entry it beats (if any).
// From the insertion point on every entry moves down, so what leaves a segment is always its old last entry, which is exactly
// what the next lane takes in.  (Emulated against a stable sort in tools/knn_quad_model.py.)
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kVnQThreads = 256;                  // 64 points per block
constexpr int kVnQPts = kVnQThreads / 4;
// Measured (tools/kbench_vn.py, B = 64 / 2048 clouds of 1024 points, mean pooling, us): 12 slots + 3 waves per SIMD 146 / 3339;
// 8 slots (37.8 KB of LDS per block: four blocks per CU) + 4 waves 137 / 3107; skipping empty insertion rounds: kNN kernel 82 -> 73.
#ifndef EQA_VN_QSLOTS
#define EQA_VN_QSLOTS 8
#endif
#ifndef EQA_VN_QUAD_WAVES
#define EQA_VN_QUAD_WAVES 4   // waves per SIMD the mean-pooling kernel's register allocation must allow
#endif
#ifndef EQA_VN_DRAIN_SKIP
#define EQA_VN_DRAIN_SKIP 1   // 1: skip a quad lane's insertion round when no lane of the wave has an entry in it
#endif
constexpr int kVnQSlots = EQA_VN_QSLOTS;          // pending candidates per lane (LDS, 8 bytes each)

template <int CTRL>
__device__ __forceinline__ float vn_dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int vn_dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
constexpr int kQuadPrev = 0x90;  // quad_perm [0,0,1,2]: lane q reads lane q - 1 (lane 0 itself)
// sum over the four lanes of a quad, returned to all of them
__device__ __forceinline__ float vn_quad_sum(float v) {
  v += vn_dpp_f<0xB1>(v);  // [1,0,3,2]
  v += vn_dpp_f<0x4E>(v);  // [2,3,0,1]
  return v;
}

// cloud -> LDS as float4 (x, y, z, |p|^2), padded with never-selected points to a multiple of 16
__device__ __forceinline__ void vn_stage_cloud_quad(const float* __restrict__ xb, int N, int Npad, float4* pts, int tid) {
  for (int i = tid; i < Npad; i += kVnQThreads) {
    if (i < N) {
      const float a = xb[i], c = xb[N + i], d = xb[2 * (size_t)N + i];
      pts[i] = make_float4(a, c, d, a * a + c * c + d * d);
    } else {
      pts[i] = make_float4(0.f, 0.f, 0.f, INFINITY);
    }
  }
}

// The point's 4 SEG nearest neighbours, distributed: on return lane `sub` of the quad holds ranks [SEG sub, SEG sub + SEG) in
// (bv, bi), best first.  queue: LDS, [kVnQSlots][kVnQThreads] float2, this BLOCK's base.
template <int SEG>
__device__ __forceinline__ void vn_knn_quad(const float4* pts, int Npad, float2* queue, const V3& ctr, float cn, int tid,
                                            float (&bv)[SEG], int (&bi)[SEG]) {
  const int sub = tid & 3;
#pragma unroll
  for (int t = 0; t < SEG; ++t) { bv[t] = -INFINITY; bi[t] = 0; }
  auto score = [&](const float4& p) {
    const float inner = -2.0f * (ctr.x * p.x + ctr.y * p.y + ctr.z * p.z);
    return (-p.w - inner) - cn;
  };
  auto insert = [&](float v, int j) {   // v, j uniform over the quad; v = -inf: no candidate
    float pv = vn_dpp_f<kQuadPrev>(bv[SEG - 1]);
    const int pi = vn_dpp_i<kQuadPrev>(bi[SEG - 1]);
    pv = sub == 0 ? INFINITY : pv;
    const bool take = v > pv;
    const float cv = take ? pv : v;
    const int ci = take ? pi : j;
    // g[t]: the incoming element goes in front of entry t.  The segment is sorted, so g is monotone (the "sticky" flag of the
    // serial chain for free) and every entry's new value is a two-level select of OLD values: entry t keeps itself, takes the
    // incoming element (g[t] and not g[t-1]) or takes entry t-1 -- SEG independent selects instead of a SEG-deep chain.
    bool g[SEG];
#pragma unroll
    for (int t = 0; t < SEG; ++t) g[t] = take | (cv > bv[t]);   // '|', not '||': no control flow
#pragma unroll
    for (int t = SEG - 1; t >= 1; --t) {
      const float uv = g[t - 1] ? bv[t - 1] : cv;
      const int ui = g[t - 1] ? bi[t - 1] : ci;
      bv[t] = g[t] ? uv : bv[t];
      bi[t] = g[t] ? ui : bi[t];
    }
    bv[0] = g[0] ? cv : bv[0];
    bi[0] = g[0] ? ci : bi[0];
  };
  float thr = -INFINITY;   // the list's last entry (lane 3's last): a candidate must beat it
  int cnt = 0;
  float2* const myq = queue + tid;
  const float2* const quadq = queue + (tid & ~3);
  auto drain = [&]() {
#pragma unroll 1
    for (int s = 0; s < kVnQSlots; ++s) {
      if (!__any(s < cnt)) break;   // wave-uniform
      const float2 e0 = quadq[s * kVnQThreads], e1 = quadq[s * kVnQThreads + 1], e2 = quadq[s * kVnQThreads + 2],
                   e3 = quadq[s * kVnQThreads + 3];
      const int c0 = vn_dpp_i<0x00>(cnt), c1 = vn_dpp_i<0x55>(cnt), c2 = vn_dpp_i<0xAA>(cnt), c3 = vn_dpp_i<0xFF>(cnt);
#if EQA_VN_DRAIN_SKIP
      if (__any(s < c0)) insert(s < c0 ? e0.x : -INFINITY, __float_as_int(e0.y));
      if (__any(s < c1)) insert(s < c1 ? e1.x : -INFINITY, __float_as_int(e1.y));
      if (__any(s < c2)) insert(s < c2 ? e2.x : -INFINITY, __float_as_int(e2.y));
      if (__any(s < c3)) insert(s < c3 ? e3.x : -INFINITY, __float_as_int(e3.y));
#else
      insert(s < c0 ? e0.x : -INFINITY, __float_as_int(e0.y));
      insert(s < c1 ? e1.x : -INFINITY, __float_as_int(e1.y));
      insert(s < c2 ? e2.x : -INFINITY, __float_as_int(e2.y));
      insert(s < c3 ? e3.x : -INFINITY, __float_as_int(e3.y));
#endif
    }
    cnt = 0;
    thr = vn_dpp_f<0xFF>(bv[SEG - 1]);
  };
  auto offer = [&](float v, int j) {
    if (v > thr) {
      myq[cnt * kVnQThreads] = make_float2(v, __int_as_float(j));
      ++cnt;
    }
  };
  for (int j0 = 0; j0 < Npad; j0 += 16) {
    const int j = j0 + sub;
    const float4 p0 = pts[j], p1 = pts[j + 4], p2 = pts[j + 8], p3 = pts[j + 12];
    offer(score(p0), j);
    offer(score(p1), j + 4);
    offer(score(p2), j + 8);
    offer(score(p3), j + 12);
    if (__any(cnt > kVnQSlots - 4)) drain();  // wave-uniform
  }
  drain();
}

}  // namespace
