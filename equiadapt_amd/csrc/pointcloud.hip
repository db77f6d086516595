// libeqa_hip.so, part 5 of 5 -- point clouds and n-body: fused kNN + VNSmall forward (P1, P2), Gram-Schmidt (P3), SO(3) action
// (P4), modified Gram-Schmidt and the rigid action on rows (n-body E(3)).  C ABI: include/eqa_hip.h.  HISTORY.md section 3.3, 3.5.
#include "eqa_common.hpp"
#include "vn_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// P4 / P3: SO(3) action on point clouds, batched Gram-Schmidt
// ------------------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kThreads) void so3_rotate_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                             float* __restrict__ y, int N, int transpose) {
  const int b = blockIdx.y;
  const float* Rb = R + (size_t)b * 9;
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = transpose ? Rb[(k % 3) * 3 + k / 3] : Rb[k];
  const float* xb = x + (size_t)b * 3 * N;
  float* yb = y + (size_t)b * 3 * N;
  if (VEC) {
    const int n4 = N >> 2;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n4) return;
    const float4 p0 = reinterpret_cast<const float4*>(xb)[k];
    const float4 p1 = reinterpret_cast<const float4*>(xb + N)[k];
    const float4 p2 = reinterpret_cast<const float4*>(xb + 2 * (size_t)N)[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // same accumulation order as a k-ordered dot product: (m0*x0 + m1*x1) + m2*x2
      float4 o;
      o.x = m[r * 3] * p0.x + m[r * 3 + 1] * p1.x + m[r * 3 + 2] * p2.x;
      o.y = m[r * 3] * p0.y + m[r * 3 + 1] * p1.y + m[r * 3 + 2] * p2.y;
      o.z = m[r * 3] * p0.z + m[r * 3 + 1] * p1.z + m[r * 3 + 2] * p2.z;
      o.w = m[r * 3] * p0.w + m[r * 3 + 1] * p1.w + m[r * 3 + 2] * p2.w;
      reinterpret_cast<float4*>(yb + (size_t)r * N)[k] = o;
    }
  } else {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= N) return;
    const float p0 = xb[k], p1 = xb[N + k], p2 = xb[2 * (size_t)N + k];
#pragma unroll
    for (int r = 0; r < 3; ++r) yb[(size_t)r * N + k] = m[r * 3] * p0 + m[r * 3 + 1] * p1 + m[r * 3 + 2] * p2;
  }
}

// classical Gram-Schmidt of the three rows of p -> o (both 9 floats; used by gram_schmidt_kernel and by the fused tail below)
__device__ __forceinline__ void gram_schmidt_rows(const float* p, float* o);

__global__ __launch_bounds__(kThreads) void gram_schmidt_kernel(const float* __restrict__ v, float* __restrict__ out, int B) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  gram_schmidt_rows(v + (size_t)b * 9, out + (size_t)b * 9);
}

__device__ __forceinline__ void gram_schmidt_rows(const float* p, float* o) {
  // The three steps of common/utils.py:22-51 (no epsilon, no handedness fix) evaluated in fp64 and rounded once: 9 numbers per
  // cloud, so the cost is nil, and the step has no epsilon -- it amplifies rounding by the conditioning of the three vectors
  // (cond 540 on one of the bench's eight clouds, 1,659 among 64: an fp32 evaluation, this one or the reference's, then sits
  // 4e-5 .. 6e-5 from the exact frame; tools/cfg4_budget.py).  In fp64 the frame is the exact Gram-Schmidt of the vectors the
  // network kernel produced, to the last fp32 bit, so the only error left in R is the network's own, amplified.
  double a0 = p[0], a1 = p[1], a2 = p[2], b0 = p[3], b1 = p[4], b2 = p[5], c0 = p[6], c1 = p[7], c2 = p[8];
  double n = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  a0 /= n; a1 /= n; a2 /= n;
  const double d = b0 * a0 + b1 * a1 + b2 * a2;
  b0 -= d * a0; b1 -= d * a1; b2 -= d * a2;
  n = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
  b0 /= n; b1 /= n; b2 /= n;
  const double d1 = c0 * a0 + c1 * a1 + c2 * a2;
  const double d2 = c0 * b0 + c1 * b1 + c2 * b2;  // classical GS: both projections use the ORIGINAL c
  c0 = c0 - d1 * a0 - d2 * b0;
  c1 = c1 - d1 * a1 - d2 * b1;
  c2 = c2 - d1 * a2 - d2 * b2;
  n = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
  c0 /= n; c1 /= n; c2 /= n;
  o[0] = (float)a0; o[1] = (float)a1; o[2] = (float)a2; o[3] = (float)b0; o[4] = (float)b1; o[5] = (float)b2;
  o[6] = (float)c0; o[7] = (float)c1; o[8] = (float)c2;
}

// Backward of the classical Gram-Schmidt above: g:(B,3,3) = dL/d(e1,e2,e3) -> gv:(B,3,3) = dL/d(v1,v2,v3).  With e = u / |u|,
// dL/du = (g - e <e, g>) / |u|;  u3 = v3 - <v3,e1> e1 - <v3,e2> e2 and u2 = v2 - <v2,e1> e1 feed gradients back into e1, e2:
//   d/de of -<v,e> e against gu:  -(<v,e> gu + <gu,e> v).
// (As 100 element-wise autograd launches on (B,3) tensors this was the largest host cost of a point-cloud training step.)
__global__ __launch_bounds__(kThreads) void gram_schmidt_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                   float* __restrict__ gv, int B) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  const float* p = v + (size_t)b * 9;
  const float* q = g + (size_t)b * 9;
  const V3 v1 = v3(p[0], p[1], p[2]), v2 = v3(p[3], p[4], p[5]), vv3 = v3(p[6], p[7], p[8]);
  auto scaled = [](const V3& a, float s) { return v3(a.x * s, a.y * s, a.z * s); };
  auto sub = [](const V3& a, const V3& c) { return v3(a.x - c.x, a.y - c.y, a.z - c.z); };
  auto add = [](const V3& a, const V3& c) { return v3(a.x + c.x, a.y + c.y, a.z + c.z); };
  // forward, keeping the lengths
  const float n1 = sqrtf(dot3(v1, v1));
  const V3 e1 = scaled(v1, 1.0f / n1);
  const float c = dot3(v2, e1);
  const V3 u2 = sub(v2, scaled(e1, c));
  const float n2 = sqrtf(dot3(u2, u2));
  const V3 e2 = scaled(u2, 1.0f / n2);
  const float a1 = dot3(vv3, e1), a2 = dot3(vv3, e2);
  const V3 u3 = sub(sub(vv3, scaled(e1, a1)), scaled(e2, a2));
  const float n3 = sqrtf(dot3(u3, u3));
  const V3 e3 = scaled(u3, 1.0f / n3);
  // backward
  V3 g1 = v3(q[0], q[1], q[2]), g2 = v3(q[3], q[4], q[5]);
  const V3 g3 = v3(q[6], q[7], q[8]);
  const V3 gu3 = scaled(sub(g3, scaled(e3, dot3(e3, g3))), 1.0f / n3);
  const float s31 = dot3(gu3, e1), s32 = dot3(gu3, e2);
  const V3 gv3 = sub(sub(gu3, scaled(e1, s31)), scaled(e2, s32));
  g1 = sub(g1, add(scaled(gu3, a1), scaled(vv3, s31)));
  g2 = sub(g2, add(scaled(gu3, a2), scaled(vv3, s32)));
  const V3 gu2 = scaled(sub(g2, scaled(e2, dot3(e2, g2))), 1.0f / n2);
  const float s21 = dot3(gu2, e1);
  const V3 gv2 = sub(gu2, scaled(e1, s21));
  g1 = sub(g1, add(scaled(gu2, c), scaled(v2, s21)));
  const V3 gv1 = scaled(sub(g1, scaled(e1, dot3(e1, g1))), 1.0f / n1);
  float* o = gv + (size_t)b * 9;
  o[0] = gv1.x; o[1] = gv1.y; o[2] = gv1.z; o[3] = gv2.x; o[4] = gv2.y; o[5] = gv2.z; o[6] = gv3.x; o[7] = gv3.y; o[8] = gv3.z;
}

// ------------------------------------------------------------------------------------------------
// P1 + P2: fused VNSmall forward (eval mode, mean pooling): kNN graph -> cross edge features -> VN linear / VN batch-norm
// / direction-gated ReLU (3->21) -> mean over neighbours -> (21->21) + VN batch-norm -> (21->4) -> mean over points.
// Reference: pointcloud/canonicalization_networks/equivariant_networks.py:15-76 (knn, get_graph_feature_cross),
// :128-150 (VNSmall.forward), vector_neuron_layers.py:251-273, :303-324.
// The reference materialises ten (B,21,3,N,k) tensors (5 MB per cloud each); here a cloud is 12 KB in LDS and every
// intermediate lives in registers: one thread = one point, its k nearest neighbours kept as a sorted register list
// while it streams over the cloud (LDS broadcast reads), then its k edges are pushed through the layers one by one.
// Packed parameter buffer (floats), eval-mode batch-norms pre-folded to scale/shift of the vector NORM:
//   [0,63) pos.Wf(21x3)  [63,126) pos.Wd  [126,147) pos.bn scale  [147,168) pos.bn shift
//   [168,609) c1.Wf(21x21)  [609,1050) c1.Wd  [1050,1071) c1.bn scale  [1071,1092) c1.bn shift
//   [1092,1113) bn1 scale  [1113,1134) bn1 shift
//   [1134,1218) c2.Wf(4x21)  [1218,1302) c2.Wd  [1302,1306) c2.bn scale  [1306,1310) c2.bn shift
//   pooling = "max" only: [1310,1751) pool.Wd(21x21)
// MAXPOOL (VNMaxPool, vector_neuron_layers.py:349-364): per point and channel the edge with the largest <q_c, (W_p q)_c> is kept
// instead of the mean over the edges -- the first such edge, like torch.max; W_p mixes all 21 channels of an edge, so the edge's
// 21 vectors stay in registers until its 21 scores exist (two waves per SIMD instead of three).
// ------------------------------------------------------------------------------------------------
template <bool MAXPOOL>
__global__ __launch_bounds__(kVnThreads, MAXPOOL ? 2 : EQA_VN_MIN_BLOCKS) void vnsmall_fwd_kernel(const float* __restrict__ x,
                                                                                                const float* __restrict__ prm,
                                                                                                float* __restrict__ partial, int N,
                                                                                                int nblk) {
  extern __shared__ __attribute__((aligned(16))) float vn_smem[];
  float4* pts = reinterpret_cast<float4*>(vn_smem);  // [Npad] (x, y, z, |p|^2): one ds_read_b128 per candidate
  __shared__ float s_part[kVnThreads / 64][12];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int Npad = (N + 3) & ~3;
  vn_stage_cloud(x + (size_t)b * 3 * N, N, Npad, pts, tid);
  __syncthreads();
  const int n = blockIdx.x * kVnThreads + tid;
  const bool active = n < N;
  const float4 c4 = pts[active ? n : N - 1];
  const V3 ctr = v3(c4.x, c4.y, c4.z);
  int bi[kVnK];
  vn_knn(pts, Npad, reinterpret_cast<float2*>(vn_smem + 4 * Npad) + tid, ctr, c4.w, bi);

  // ---- conv_pos on the k edges + mean over neighbours
  V3 pooled[kVnC];
  float best[MAXPOOL ? kVnC : 1];
#pragma unroll
  for (int c = 0; c < kVnC; ++c) pooled[c] = v3(0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < (MAXPOOL ? kVnC : 1); ++c) best[c] = 0.f;
  const float* Wf = prm;
  const float* Wd = prm + 63;
  const float* bsc = prm + 126;
  const float* bsh = prm + 147;
#pragma unroll 1
  for (int t = 0; t < kVnK; ++t) {
    // compiler barrier: otherwise the 168 loop-invariant scalar weights are hoisted out of the loop and, for want of
    // SGPRs, parked in VGPRs for its whole duration (the kernel then needs > 256 VGPRs: one wave per SIMD)
    asm volatile("" ::: "memory");
    // (dynamic t: pick the t-th neighbour index through a select chain, the list lives in registers)
    int j = bi[0];
#pragma unroll
    for (int u = 1; u < kVnK; ++u) j = (t == u) ? bi[u] : j;
    const float4 nb4 = pts[j];
    const V3 nb = v3(nb4.x, nb4.y, nb4.z);
    const V3 f0 = v3(nb.x - ctr.x, nb.y - ctr.y, nb.z - ctr.z);                                   // neighbour - centre
    const V3 f2 = v3(nb.y * ctr.z - nb.z * ctr.y, nb.z * ctr.x - nb.x * ctr.z, nb.x * ctr.y - nb.y * ctr.x);  // nbr x ctr
    V3 qe[MAXPOOL ? kVnC : 1];
#pragma unroll
    for (int c = 0; c < kVnC; ++c) {
      const float a0 = Wf[c * 3], a1 = Wf[c * 3 + 1], a2 = Wf[c * 3 + 2];
      const float d0 = Wd[c * 3], d1 = Wd[c * 3 + 1], d2 = Wd[c * 3 + 2];
      V3 q = v3(a0 * f0.x + a1 * ctr.x + a2 * f2.x, a0 * f0.y + a1 * ctr.y + a2 * f2.y, a0 * f0.z + a1 * ctr.z + a2 * f2.z);
      const V3 d = v3(d0 * f0.x + d1 * ctr.x + d2 * f2.x, d0 * f0.y + d1 * ctr.y + d2 * f2.y, d0 * f0.z + d1 * ctr.z + d2 * f2.z);
      q = vn_relu(vn_bn(q, bsc[c], bsh[c]), d);
      if (MAXPOOL) {
        qe[c] = q;
      } else {
        pooled[c].x += q.x; pooled[c].y += q.y; pooled[c].z += q.z;
      }
    }
    if (MAXPOOL) {
      const float* Wp = prm + kVnParams;
#pragma unroll
      for (int c = 0; c < kVnC; ++c) {
        if (c % 3 == 0) asm volatile("" ::: "memory");     // keeps the 441 scalar weights of W_p from being hoisted out of the edge loop
        V3 dp = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < kVnC; ++a) {
          const float w = Wp[c * kVnC + a];
          dp.x += w * qe[a].x; dp.y += w * qe[a].y; dp.z += w * qe[a].z;
        }
        const float sc = qe[c].x * dp.x + qe[c].y * dp.y + qe[c].z * dp.z;
        const bool take = t == 0 || sc > best[c];          // strict: the first maximal edge wins, as torch.max does
        best[c] = take ? sc : best[c];
        pooled[c].x = take ? qe[c].x : pooled[c].x;
        pooled[c].y = take ? qe[c].y : pooled[c].y;
        pooled[c].z = take ? qe[c].z : pooled[c].z;
      }
    }
  }
  if (!MAXPOOL) {
    const float inv_k = 1.0f / (float)kVnK;
#pragma unroll
    for (int c = 0; c < kVnC; ++c) { pooled[c].x *= inv_k; pooled[c].y *= inv_k; pooled[c].z *= inv_k; }
  }

  // ---- conv1 (21->21) + its VN-BN + ReLU, then bn1; every output channel is folded into conv2's (21->4) two linear
  // maps as soon as it exists, so the 21 x 3 intermediate never has to be held in registers
  V3 q2[4], d2[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) { q2[o] = v3(0.f, 0.f, 0.f); d2[o] = v3(0.f, 0.f, 0.f); }
  {
    const float* W1f = prm + 168;
    const float* W1d = prm + 609;
    const float* s1 = prm + 1050;
    const float* t1 = prm + 1071;
    const float* s2 = prm + 1092;
    const float* t2 = prm + 1113;
    const float* W2f = prm + 1134;
    const float* W2d = prm + 1218;
#pragma unroll 1  // rolled: fully unrolled (882 scalar weights in flight) the kernel needs > 256 VGPRs
    for (int c = 0; c < kVnC; ++c) {
      asm volatile("" ::: "memory");
      V3 q = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 0.f);
#pragma unroll
      for (int a = 0; a < kVnC; ++a) {
        const float wf = W1f[c * kVnC + a], wd = W1d[c * kVnC + a];
        q.x += wf * pooled[a].x; q.y += wf * pooled[a].y; q.z += wf * pooled[a].z;
        d.x += wd * pooled[a].x; d.y += wd * pooled[a].y; d.z += wd * pooled[a].z;
      }
      q = vn_relu(vn_bn(q, s1[c], t1[c]), d);
      const V3 hc = vn_bn(q, s2[c], t2[c]);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float wf = W2f[o * kVnC + c], wd = W2d[o * kVnC + c];
        q2[o].x += wf * hc.x; q2[o].y += wf * hc.y; q2[o].z += wf * hc.z;
        d2[o].x += wd * hc.x; d2[o].y += wd * hc.y; d2[o].z += wd * hc.z;
      }
    }
  }
  // ---- conv2's VN-BN + ReLU -> this point's contribution to the mean over points
  float outv[12];
  {
    const float* s3 = prm + 1302;
    const float* t3 = prm + 1306;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const V3 q = vn_relu(vn_bn(q2[c], s3[c], t3[c]), d2[c]);
      outv[c * 3] = active ? q.x : 0.f;
      outv[c * 3 + 1] = active ? q.y : 0.f;
      outv[c * 3 + 2] = active ? q.z : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const float sres = wave_sum_f(outv[c]);
    if ((tid & 63) == 0) s_part[tid >> 6][c] = sres;
  }
  __syncthreads();
  if (tid < 12) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < kVnThreads / 64; ++w) acc += s_part[w][tid];
    partial[((size_t)b * nblk + blockIdx.x) * 12 + tid] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// The same network with FOUR LANES PER POINT (vn_common.hpp, "quad" kernels) and any k <= 4 SEG: the quad scans the candidates
// together (distributed sorted list), each lane pushes its share of the point's edges (ranks SEG sub + t < k) through conv_pos,
// the pooled features are combined over the quad by DPP (sum, or first-maximum for VNMaxPool), conv1 / bn1 / conv2 are split
// over the quad by OUTPUT channel (lane `sub` takes c = sub, sub + 4, ...; those weights are lane-dependent, so they come from a
// re-packed copy in LDS: rows padded to 24 floats for ds_read_b128, rows 21..23 zero so that the idle sixth trip of lanes 1..3
// contributes exact zeros), and lane `sub` finishes output channel `sub`.  4 x the waves of the one-thread-per-point kernel at
// the same total instruction count: a batch of 64 clouds fills 4 waves per SIMD instead of one.
// LDS floats: [4 Npad] cloud | [2 * 12 * 256] queue | tail parameters:
//   W1f [24][24] | W1d [24][24] | (s1, t1, s2, t2) [24][4] | W2T [24][8] = (W2f[0..3][c], W2d[0..3][c]) | (s3, t3) [4][2]
// ------------------------------------------------------------------------------------------------
constexpr int kVnTailLds = 2 * 24 * 24 + 24 * 4 + 24 * 8 + 8;   // 1448 floats

template <int SEG, bool MAXPOOL>
__global__ __launch_bounds__(kVnQThreads, MAXPOOL ? 2 : EQA_VN_QUAD_WAVES) void vnsmall_fwd_quad_kernel(const float* __restrict__ x,
                                                                                         const float* __restrict__ prm,
                                                                                         float* __restrict__ partial, int N, int k,
                                                                                         int nblk) {
  extern __shared__ __attribute__((aligned(16))) float vn_smem[];
  const int b = blockIdx.y, tid = threadIdx.x, sub = tid & 3;
  const int Npad = (N + 15) & ~15;
  float4* pts = reinterpret_cast<float4*>(vn_smem);
  float2* queue = reinterpret_cast<float2*>(vn_smem + 4 * Npad);
  float* tl = vn_smem + 4 * Npad + 2 * kVnQSlots * kVnQThreads;
  __shared__ float s_part[kVnQThreads / 64][12];
  vn_stage_cloud_quad(x + (size_t)b * 3 * N, N, Npad, pts, tid);
  // tail parameters, re-packed (see above)
  for (int i = tid; i < kVnTailLds; i += kVnQThreads) {
    float v = 0.f;
    if (i < 2 * 576) {                       // W1f | W1d rows
      const int m = i / 576, r = (i - m * 576) / 24, a = i % 24;
      if (r < kVnC && a < kVnC) v = prm[(m ? 609 : 168) + r * kVnC + a];
    } else if (i < 2 * 576 + 96) {           // (s1, t1, s2, t2)[c]
      const int c = (i - 1152) >> 2, w = (i - 1152) & 3;
      if (c < kVnC) v = prm[(w == 0 ? 1050 : w == 1 ? 1071 : w == 2 ? 1092 : 1113) + c];
    } else if (i < 2 * 576 + 96 + 192) {     // W2T[c][0..3] = W2f[o][c], [4..7] = W2d[o][c]
      const int c = (i - 1248) >> 3, w = (i - 1248) & 7;
      if (c < kVnC) v = prm[(w < 4 ? 1134 : 1218) + (w & 3) * kVnC + c];
    } else {                                 // (s3, t3)[o]
      const int o = (i - 1440) >> 1, w = (i - 1440) & 1;
      v = prm[(w ? 1306 : 1302) + o];
    }
    tl[i] = v;
  }
  __syncthreads();
  const int n = blockIdx.x * kVnQPts + (tid >> 2);
  const bool active = n < N;
  const float4 c4 = pts[active ? n : N - 1];
  const V3 ctr = v3(c4.x, c4.y, c4.z);
  float bv[SEG];
  int bi[SEG];
  vn_knn_quad<SEG>(pts, Npad, queue, ctr, c4.w, tid, bv, bi);

  // ---- conv_pos on this lane's edges (ranks SEG * sub + t < k)
  V3 pooled[kVnC];
  float best[MAXPOOL ? kVnC : 1];
#pragma unroll
  for (int c = 0; c < kVnC; ++c) pooled[c] = v3(0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < (MAXPOOL ? kVnC : 1); ++c) best[c] = -INFINITY;
  const float* Wf = prm;
  const float* Wd = prm + 63;
  const float* bsc = prm + 126;
  const float* bsh = prm + 147;
  if (!MAXPOOL) {
    // mean pooling: channel-outer, edge-inner.  The lane's SEG edges are set up once (6 registers each); a channel's 8 scalars are
    // then loaded once for all of them (edge-outer: once per edge, with a scalar-cache wait per channel), the SEG dependent
    // root / reciprocal chains of a channel are independent of each other, and only one accumulator triple is live at a time.
    V3 f0[SEG], f2[SEG];
    float m[SEG];
#pragma unroll
    for (int t = 0; t < SEG; ++t) {
      const float4 nb4 = pts[bi[t]];
      const V3 nb = v3(nb4.x, nb4.y, nb4.z);
      f0[t] = v3(nb.x - ctr.x, nb.y - ctr.y, nb.z - ctr.z);                                               // neighbour - centre
      f2[t] = v3(nb.y * ctr.z - nb.z * ctr.y, nb.z * ctr.x - nb.x * ctr.z, nb.x * ctr.y - nb.y * ctr.x);  // neighbour x centre
      m[t] = SEG * sub + t < k ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < kVnC; ++c) {
      asm volatile("" ::: "memory");   // one channel at a time: without it the scheduler interleaves all 21 and spills
      const float a0 = Wf[c * 3], a1 = Wf[c * 3 + 1], a2 = Wf[c * 3 + 2];
      const float d0 = Wd[c * 3], d1 = Wd[c * 3 + 1], d2 = Wd[c * 3 + 2];
      const float sc = bsc[c], sh = bsh[c];
      const V3 qc = v3(a1 * ctr.x, a1 * ctr.y, a1 * ctr.z), dc = v3(d1 * ctr.x, d1 * ctr.y, d1 * ctr.z);  // the centre's share: same for all edges
      V3 acc = v3(0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < SEG; ++t) {
        V3 q = v3(a0 * f0[t].x + qc.x + a2 * f2[t].x, a0 * f0[t].y + qc.y + a2 * f2[t].y, a0 * f0[t].z + qc.z + a2 * f2[t].z);
        const V3 d = v3(d0 * f0[t].x + dc.x + d2 * f2[t].x, d0 * f0[t].y + dc.y + d2 * f2[t].y, d0 * f0[t].z + dc.z + d2 * f2[t].z);
        q = vn_relu_sel(vn_bn(q, sc, sh), d);
        acc.x += m[t] * q.x; acc.y += m[t] * q.y; acc.z += m[t] * q.z;
      }
      pooled[c] = acc;
    }
  } else {
#pragma unroll 1
  for (int t = 0; t < SEG; ++t) {
    if (!__any(SEG * sub + t < k)) break;   // wave-uniform
    asm volatile("" ::: "memory");          // keep the weights in the scalar cache, not hoisted into VGPRs (see above)
    int j = bi[0];
#pragma unroll
    for (int u = 1; u < SEG; ++u) j = (t == u) ? bi[u] : j;
    const bool valid = SEG * sub + t < k;
    const float m = valid ? 1.0f : 0.0f;
    const float4 nb4 = pts[j];
    const V3 nb = v3(nb4.x, nb4.y, nb4.z);
    const V3 f0 = v3(nb.x - ctr.x, nb.y - ctr.y, nb.z - ctr.z);
    const V3 f2 = v3(nb.y * ctr.z - nb.z * ctr.y, nb.z * ctr.x - nb.x * ctr.z, nb.x * ctr.y - nb.y * ctr.x);
    V3 qe[MAXPOOL ? kVnC : 1];
#pragma unroll
    for (int c = 0; c < kVnC; ++c) {
      const float a0 = Wf[c * 3], a1 = Wf[c * 3 + 1], a2 = Wf[c * 3 + 2];
      const float d0 = Wd[c * 3], d1 = Wd[c * 3 + 1], d2 = Wd[c * 3 + 2];
      V3 q = v3(a0 * f0.x + a1 * ctr.x + a2 * f2.x, a0 * f0.y + a1 * ctr.y + a2 * f2.y, a0 * f0.z + a1 * ctr.z + a2 * f2.z);
      const V3 d = v3(d0 * f0.x + d1 * ctr.x + d2 * f2.x, d0 * f0.y + d1 * ctr.y + d2 * f2.y, d0 * f0.z + d1 * ctr.z + d2 * f2.z);
      q = vn_relu(vn_bn(q, bsc[c], bsh[c]), d);
      if (MAXPOOL) {
        qe[c] = q;
      } else {
        pooled[c].x += m * q.x; pooled[c].y += m * q.y; pooled[c].z += m * q.z;
      }
    }
    if (MAXPOOL) {
      const float* Wp = prm + kVnParams;
#pragma unroll
      for (int c = 0; c < kVnC; ++c) {
        if (c % 3 == 0) asm volatile("" ::: "memory");
        V3 dp = v3(0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < kVnC; ++a) {
          const float w = Wp[c * kVnC + a];
          dp.x += w * qe[a].x; dp.y += w * qe[a].y; dp.z += w * qe[a].z;
        }
        const float sc = qe[c].x * dp.x + qe[c].y * dp.y + qe[c].z * dp.z;
        const bool take = valid && sc > best[c];   // strict: the first maximal edge (lowest rank) wins, as torch.max does
        best[c] = take ? sc : best[c];
        pooled[c].x = take ? qe[c].x : pooled[c].x;
        pooled[c].y = take ? qe[c].y : pooled[c].y;
        pooled[c].z = take ? qe[c].z : pooled[c].z;
      }
    }
  }
  }
  // ---- the point's pooled features in all four lanes
  if (MAXPOOL) {
    // lane `sub` holds the ranks below lane sub + 1's: on equal scores the lower lane's pick stands
#pragma unroll
    for (int c = 0; c < kVnC; ++c) {
      {
        const float os = vn_dpp_f<0xB1>(best[c]);
        const float ox = vn_dpp_f<0xB1>(pooled[c].x), oy = vn_dpp_f<0xB1>(pooled[c].y), oz = vn_dpp_f<0xB1>(pooled[c].z);
        const bool take = (sub & 1) ? (os >= best[c]) : (os > best[c]);
        best[c] = take ? os : best[c];
        pooled[c].x = take ? ox : pooled[c].x; pooled[c].y = take ? oy : pooled[c].y; pooled[c].z = take ? oz : pooled[c].z;
      }
      {
        const float os = vn_dpp_f<0x4E>(best[c]);
        const float ox = vn_dpp_f<0x4E>(pooled[c].x), oy = vn_dpp_f<0x4E>(pooled[c].y), oz = vn_dpp_f<0x4E>(pooled[c].z);
        const bool take = (sub & 2) ? (os >= best[c]) : (os > best[c]);
        pooled[c].x = take ? ox : pooled[c].x; pooled[c].y = take ? oy : pooled[c].y; pooled[c].z = take ? oz : pooled[c].z;
      }
    }
  } else {
    const float inv_k = 1.0f / (float)k;
#pragma unroll
    for (int c = 0; c < kVnC; ++c) {
      pooled[c].x = vn_quad_sum(pooled[c].x) * inv_k;
      pooled[c].y = vn_quad_sum(pooled[c].y) * inv_k;
      pooled[c].z = vn_quad_sum(pooled[c].z) * inv_k;
    }
  }

  // ---- conv1 + its VN-BN + ReLU + bn1 for the output channels c = sub, sub + 4, ..., folded into conv2's two maps
  V3 q2[4], d2[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) { q2[o] = v3(0.f, 0.f, 0.f); d2[o] = v3(0.f, 0.f, 0.f); }
#pragma unroll 1
  for (int ci = 0; ci < 6; ++ci) {
    const int c = sub + 4 * ci;   // <= 23; rows 21..23 are zero
    const float4* rf = reinterpret_cast<const float4*>(tl + c * 24);
    const float4* rd = reinterpret_cast<const float4*>(tl + 576 + c * 24);
    V3 q = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int a4 = 0; a4 < 6; ++a4) {
      const float4 wf = rf[a4], wd = rd[a4];
      const float wfa[4] = {wf.x, wf.y, wf.z, wf.w}, wda[4] = {wd.x, wd.y, wd.z, wd.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int a = a4 * 4 + u;
        if (a < kVnC) {
          q.x += wfa[u] * pooled[a].x; q.y += wfa[u] * pooled[a].y; q.z += wfa[u] * pooled[a].z;
          d.x += wda[u] * pooled[a].x; d.y += wda[u] * pooled[a].y; d.z += wda[u] * pooled[a].z;
        }
      }
    }
    const float4 st = *reinterpret_cast<const float4*>(tl + 1152 + c * 4);
    q = vn_relu(vn_bn(q, st.x, st.y), d);
    const V3 hc = vn_bn(q, st.z, st.w);
    const float4 w2f = *reinterpret_cast<const float4*>(tl + 1248 + c * 8), w2d = *reinterpret_cast<const float4*>(tl + 1248 + c * 8 + 4);
    const float wfo[4] = {w2f.x, w2f.y, w2f.z, w2f.w}, wdo[4] = {w2d.x, w2d.y, w2d.z, w2d.w};
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      q2[o].x += wfo[o] * hc.x; q2[o].y += wfo[o] * hc.y; q2[o].z += wfo[o] * hc.z;
      d2[o].x += wdo[o] * hc.x; d2[o].y += wdo[o] * hc.y; d2[o].z += wdo[o] * hc.z;
    }
  }
  // ---- conv2's VN-BN + ReLU: lane `sub` finishes output channel `sub`
  V3 qs = v3(0.f, 0.f, 0.f), ds = v3(0.f, 0.f, 0.f);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const float qx = vn_quad_sum(q2[o].x), qy = vn_quad_sum(q2[o].y), qz = vn_quad_sum(q2[o].z);
    const float dx = vn_quad_sum(d2[o].x), dy = vn_quad_sum(d2[o].y), dz = vn_quad_sum(d2[o].z);
    if (sub == o) { qs = v3(qx, qy, qz); ds = v3(dx, dy, dz); }
  }
  const V3 qo = vn_relu(vn_bn(qs, tl[1440 + 2 * sub], tl[1441 + 2 * sub]), ds);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const bool mine = active && sub == o;
    const float sx = wave_sum_f(mine ? qo.x : 0.f), sy = wave_sum_f(mine ? qo.y : 0.f), sz = wave_sum_f(mine ? qo.z : 0.f);
    if ((tid & 63) == 0) { s_part[tid >> 6][o * 3] = sx; s_part[tid >> 6][o * 3 + 1] = sy; s_part[tid >> 6][o * 3 + 2] = sz; }
  }
  __syncthreads();
  if (tid < 12) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < kVnQThreads / 64; ++w) acc += s_part[w][tid];
    partial[((size_t)b * nblk + blockIdx.x) * 12 + tid] = acc;
  }
}

// (B, nblk, 12) partial sums -> (B,3,3): mean over the N points, first three of the four output channels
__global__ void vnsmall_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int B, int nblk, float inv_n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 9) return;
  const int b = i / 9, c = i - b * 9;
  float acc = 0.f;
  for (int k = 0; k < nblk; ++k) acc += partial[((size_t)b * nblk + k) * 12 + c];
  out[i] = acc * inv_n;
}

// The whole tail of the eval-mode point-cloud canonicalizer in ONE launch per batch (one block per cloud): the partial sums of the
// fused VNSmall kernel -> the network's (3, 3) output vectors (vnsmall_finalize_kernel's sums) -> their Gram-Schmidt frame
// (gram_schmidt_kernel's arithmetic; equiadapt/common/utils.py:22-51) -> the canonical cloud y = R x (so3_rotate_kernel's;
// pointcloud/canonicalization/continuous_group.py:74-79).  As three launches behind the network kernel these cost three kernel
// boundaries of a 0.12 ms step (ModelNet40-shaped batches of 64 clouds).
__global__ __launch_bounds__(kThreads) void vnsmall_canon_tail_kernel(const float* __restrict__ partial, const float* __restrict__ x,
                                                                     float* __restrict__ vec, float* __restrict__ R,
                                                                     float* __restrict__ y, int N, int nblk, float inv_n) {
  __shared__ float s_v[9], s_r[9];
  const int b = blockIdx.x;
  if (threadIdx.x < 9) {
    float acc = 0.f;
    for (int k = 0; k < nblk; ++k) acc += partial[((size_t)b * nblk + k) * 12 + threadIdx.x];
    acc *= inv_n;
    s_v[threadIdx.x] = acc;
    vec[(size_t)b * 9 + threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) gram_schmidt_rows(s_v, s_r);
  __syncthreads();
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = s_r[k];
  if (threadIdx.x < 9) R[(size_t)b * 9 + threadIdx.x] = m[threadIdx.x];
  const float* xb = x + (size_t)b * 3 * N;
  float* yb = y + (size_t)b * 3 * N;
  if ((N & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0)) {
    for (int k = threadIdx.x; k < (N >> 2); k += kThreads) {
      const float4 p0 = reinterpret_cast<const float4*>(xb)[k];
      const float4 p1 = reinterpret_cast<const float4*>(xb + N)[k];
      const float4 p2 = reinterpret_cast<const float4*>(xb + 2 * (size_t)N)[k];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float4 o;
        o.x = m[r * 3] * p0.x + m[r * 3 + 1] * p1.x + m[r * 3 + 2] * p2.x;
        o.y = m[r * 3] * p0.y + m[r * 3 + 1] * p1.y + m[r * 3 + 2] * p2.y;
        o.z = m[r * 3] * p0.z + m[r * 3 + 1] * p1.z + m[r * 3 + 2] * p2.z;
        o.w = m[r * 3] * p0.w + m[r * 3 + 1] * p1.w + m[r * 3 + 2] * p2.w;
        reinterpret_cast<float4*>(yb + (size_t)r * N)[k] = o;
      }
    }
  } else {
    for (int k = threadIdx.x; k < N; k += kThreads) {
      const float p0 = xb[k], p1 = xb[N + k], p2 = xb[2 * (size_t)N + k];
#pragma unroll
      for (int r = 0; r < 3; ++r) yb[(size_t)r * N + k] = m[r * 3] * p0 + m[r * 3 + 1] * p1 + m[r * 3 + 2] * p2;
    }
  }
}

// (f).4 -- n-body E(3) canonicalization: modified Gram-Schmidt and the per-row rigid action
// (nbody/canonicalization/euclidean_group.py:87-157).  Tiny tensors (nodes x 3): one thread per row.
__global__ __launch_bounds__(kThreads) void modified_gram_schmidt_kernel(const float* __restrict__ v, float* __restrict__ out, int B) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  const float* p = v + (size_t)b * 9;
  // fp64 inside, rounded once (see gram_schmidt_rows)
  double a0 = p[0], a1 = p[1], a2 = p[2], b0 = p[3], b1 = p[4], b2 = p[5], c0 = p[6], c1 = p[7], c2 = p[8];
  double n = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  a0 /= n; a1 /= n; a2 /= n;
  double d = b0 * a0 + b1 * a1 + b2 * a2;
  b0 -= d * a0; b1 -= d * a1; b2 -= d * a2;
  n = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
  b0 /= n; b1 /= n; b2 /= n;
  d = c0 * a0 + c1 * a1 + c2 * a2;
  c0 -= d * a0; c1 -= d * a1; c2 -= d * a2;
  d = c0 * b0 + c1 * b1 + c2 * b2;  // modified GS: second projection uses the UPDATED third vector
  c0 -= d * b0; c1 -= d * b1; c2 -= d * b2;
  n = sqrt(c0 * c0 + c1 * c1 + c2 * c2);
  c0 /= n; c1 /= n; c2 /= n;
  float* o = out + (size_t)b * 9;
  o[0] = (float)a0; o[1] = (float)a1; o[2] = (float)a2; o[3] = (float)b0; o[4] = (float)b1; o[5] = (float)b2;
  o[6] = (float)c0; o[7] = (float)c1; o[8] = (float)c2;
}

// mode 0: out = x R + t            (invert_canonicalization :126-137; t may be NULL)
// mode 1: out = x R^T - t R^T      (canonicalize :108-124, the two products subtracted as the reference does)
__global__ __launch_bounds__(kThreads) void rigid_rows_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                             const float* __restrict__ t, float* __restrict__ out, int M, int mode) {
  const int m = blockIdx.x * kThreads + threadIdx.x;
  if (m >= M) return;
  const float* r = R + (size_t)m * 9;
  const float x0 = x[(size_t)m * 3], x1 = x[(size_t)m * 3 + 1], x2 = x[(size_t)m * 3 + 2];
  const float t0 = t ? t[(size_t)m * 3] : 0.f, t1 = t ? t[(size_t)m * 3 + 1] : 0.f, t2 = t ? t[(size_t)m * 3 + 2] : 0.f;
  float* o = out + (size_t)m * 3;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (mode == 0) {
      o[j] = (x0 * r[j] + x1 * r[3 + j] + x2 * r[6 + j]) + (j == 0 ? t0 : j == 1 ? t1 : t2);
    } else {
      const float a = x0 * r[3 * j] + x1 * r[3 * j + 1] + x2 * r[3 * j + 2];
      const float b = t0 * r[3 * j] + t1 * r[3 * j + 1] + t2 * r[3 * j + 2];
      o[j] = t ? a - b : a;
    }
  }
}

}  // namespace

// eqa_set_option key 1: 0 = choose the VNSmall forward kernel by size (default), 1 = always one thread per point (k = 20 only),
// 2 = always four lanes per point
int eqa::g_vn_kernel_choice = 0;
#ifndef EQA_VN_SINGLE_MIN_POINTS
#define EQA_VN_SINGLE_MIN_POINTS (1LL << 20)   // B * N from which the one-thread-per-point kernel is preferred for max pooling
#endif

extern "C" {

int eqa_so3_rotate(const float* x, const float* R, float* y, int B, int N, int transpose, void* stream) {
  if (B == 0 && N > 0) return EQA_OK;
  if (!x || !R || !y || B < 0 || N <= 0) return EQA_ERR_INVALID_ARG;
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (N % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL((so3_rotate_kernel<true>), dim3(((N >> 2) + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st, x, R, y, N, transpose);
  else
    hipLaunchKernelGGL((so3_rotate_kernel<false>), dim3((N + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st, x, R, y, N, transpose);
  return launch_status();
}

int64_t eqa_vnsmall_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (int64_t)B * ((N + kVnQPts - 1) / kVnQPts) * 12 * (int64_t)sizeof(float);   // the finer of the two block sizes
}

static int vnsmall_main(const float* x, const float* params, void* workspace, int B, int N, int k, int pooling, void* stream, int* nblk_out);

int eqa_vnsmall_fwd(const float* x, const float* params, float* out, void* workspace, int B, int N, int k, int pooling,
                    void* stream) {
  if (!out) return EQA_ERR_INVALID_ARG;
  int nblk = 0;
  const int rc = vnsmall_main(x, params, workspace, B, N, k, pooling, stream, &nblk);
  if (rc != EQA_OK || B == 0) return rc;
  hipLaunchKernelGGL(vnsmall_finalize_kernel, dim3((B * 9 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, out, B,
                     nblk, 1.0f / (float)N);
  return launch_status();
}

int eqa_vnsmall_canonicalize(const float* x, const float* params, float* vectors, float* R, float* y, void* workspace, int B, int N, int k,
                             int pooling, void* stream) {
  if (B == 0 && N > 0) return EQA_OK;     // (empty tensors have null data pointers)
  if (!vectors || !R || !y) return EQA_ERR_INVALID_ARG;
  int nblk = 0;
  const int rc = vnsmall_main(x, params, workspace, B, N, k, pooling, stream, &nblk);
  if (rc != EQA_OK || B == 0) return rc;
  hipLaunchKernelGGL(vnsmall_canon_tail_kernel, dim3(B), dim3(kThreads), 0, (hipStream_t)stream, (const float*)workspace, x, vectors, R, y,
                     N, nblk, 1.0f / (float)N);
  return launch_status();
}

static int vnsmall_main(const float* x, const float* params, void* workspace, int B, int N, int k, int pooling, void* stream, int* nblk_out) {
  if (!x || !params || !workspace || B < 0 || N <= 0 || k <= 0) return EQA_ERR_INVALID_ARG;
  if (k > 32 || (pooling != 0 && pooling != 1) || N < k) return EQA_ERR_UNSUPPORTED;  // pooling 0 mean, 1 max
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  hipStream_t st = (hipStream_t)stream;
  // one thread per point: fewer, longer instruction streams.  Mean pooling: the quad kernel is ahead at every batch size (B = 64:
  // 137 vs 250 us, B = 2048: 3.11 vs 3.21 ms); max pooling (two waves per SIMD either way): ahead only once the batch alone fills
  // the chip (B = 2048: 5.2 vs 6.0 ms; B = 256: 0.91 vs 0.84)
  // (its LDS -- 16 B per point + the 12 KB queue -- must fit the default 64 KB of a launch: clouds beyond 3,328 points always take
  // the quad kernel, whatever the batch; only the FORCED choice reports them as unsupported)
  const size_t lds_single = (size_t)4 * ((N + 3) & ~3) * sizeof(float) + (size_t)kVnThreads * kVnQueue * sizeof(float2);
  if (g_vn_kernel_choice == 1 && k == kVnK && lds_single > 64 * 1024) return EQA_ERR_UNSUPPORTED;
  const bool single = k == kVnK && lds_single <= 64 * 1024 &&
                      (g_vn_kernel_choice == 1 || (g_vn_kernel_choice == 0 && pooling == 1 && (long long)B * N >= EQA_VN_SINGLE_MIN_POINTS));
  int nblk;
  if (single) {
    const size_t lds = lds_single;
    nblk = (N + kVnThreads - 1) / kVnThreads;
    if (pooling == 1)
      hipLaunchKernelGGL(vnsmall_fwd_kernel<true>, dim3(nblk, B), dim3(kVnThreads), lds, st, x, params, (float*)workspace, N, nblk);
    else
      hipLaunchKernelGGL(vnsmall_fwd_kernel<false>, dim3(nblk, B), dim3(kVnThreads), lds, st, x, params, (float*)workspace, N, nblk);
  } else {
    const size_t lds = ((size_t)4 * ((N + 15) & ~15) + 2 * kVnQSlots * kVnQThreads + kVnTailLds) * sizeof(float);
    if (lds > 128 * 1024) return EQA_ERR_UNSUPPORTED;
    nblk = (N + kVnQPts - 1) / kVnQPts;
    const dim3 grid(nblk, B), blk(kVnQThreads);
    float* ws = (float*)workspace;
    // clouds beyond ~2,500 points need more dynamic LDS than the default 64 KB limit of a launch
#define EQA_VN_QUAD_LAUNCH(SEG_, MAX_)                                                                                              \
  do {                                                                                                                             \
    if (lds > 64 * 1024 && !allow_dynamic_lds((const void*)vnsmall_fwd_quad_kernel<SEG_, MAX_>, 128 * 1024))                       \
      return EQA_ERR_UNSUPPORTED;                                                                                                  \
    hipLaunchKernelGGL((vnsmall_fwd_quad_kernel<SEG_, MAX_>), grid, blk, lds, st, x, params, ws, N, k, nblk);                       \
  } while (0)
    if (k <= 20) {
      if (pooling == 1) EQA_VN_QUAD_LAUNCH(5, true); else EQA_VN_QUAD_LAUNCH(5, false);
    } else {
      if (pooling == 1) EQA_VN_QUAD_LAUNCH(8, true); else EQA_VN_QUAD_LAUNCH(8, false);
    }
#undef EQA_VN_QUAD_LAUNCH
  }
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  *nblk_out = nblk;
  return EQA_OK;
}

int eqa_modified_gram_schmidt(const float* v, float* out, int B, void* stream) {
  if (B == 0) return EQA_OK;
  if (!v || !out || B < 0) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(modified_gram_schmidt_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, v, out, B);
  return launch_status();
}

int eqa_rigid_rows(const float* x, const float* R, const float* t, float* out, int M, int mode, void* stream) {
  if (M == 0) return EQA_OK;
  if (!x || !R || !out || M < 0 || (mode != 0 && mode != 1)) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rigid_rows_kernel, dim3((M + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, x, R, t, out, M, mode);
  return launch_status();
}

int eqa_gram_schmidt_bwd(const float* v, const float* grad_out, float* grad_v, int B, void* stream) {
  if (B == 0) return EQA_OK;
  if (!v || !grad_out || !grad_v || B < 0) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gram_schmidt_bwd_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, v, grad_out,
                     grad_v, B);
  return launch_status();
}

int eqa_gram_schmidt(const float* v, float* out, int B, void* stream) {
  if (B == 0) return EQA_OK;
  if (!v || !out || B < 0) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gram_schmidt_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, v, out, B);
  return launch_status();
}

}  // extern "C"
