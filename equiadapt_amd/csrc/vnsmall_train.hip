// libeqa_hip.so, part 7 -- training passes of VNSmall's first block (P1 + the conv_pos half of P2): kNN graph -> cross edge
// features -> VNLinearLeakyReLU(3 -> 21, slope 0) with TRAINING-mode VN batch-norm -> mean over the k neighbours
// (reference: pointcloud/canonicalization_networks/equivariant_networks.py:15-76, :128-140; vector_neuron_layers.py:251-273,
// :303-324).  Op by op the block materialises ~150 element-wise passes over (B, 21, 3, N, k) tensors (330 MB each at B = 64):
// 14 of the 15.9 ms of a training step.  Here nothing of that size exists: every pass re-derives the edge features from the
// 12 KB cloud in LDS and the (B, N, k) neighbour indices, one thread per point:
//   eqa_vn_knn                 neighbour indices (the queued insertion of the fused eval kernel)
//   eqa_vn_convpos_stats       per-channel sums of n and n^2, n = |W_f f| + EPS, over all edges  -> batch statistics
//   eqa_vn_convpos_fwd         q = W_f f; q <- q / n * (n * scale + shift); gate by d = W_d f; mean over k  -> (B, 21, 3, N)
//   eqa_vn_convpos_bwd_reduce  per-channel sums of g and g * nhat (g = dL/d BN output): d beta, d gamma, batch-norm terms
//   eqa_vn_convpos_bwd_apply   d W_f, d W_d (126 partial sums per block)
// No gradient w.r.t. the point coordinates (the cloud is data).  C ABI: include/eqa_hip.h.
#include "vn_common.hpp"

namespace {

// edge features of (centre, neighbour): [neighbour - centre, centre, neighbour x centre]
struct VnEdge {
  V3 f0, f1, f2;
};
__device__ __forceinline__ VnEdge vn_edge(const V3& ctr, const float4& nb4) {
  const V3 nb = v3(nb4.x, nb4.y, nb4.z);
  VnEdge e;
  e.f0 = v3(nb.x - ctr.x, nb.y - ctr.y, nb.z - ctr.z);
  e.f1 = ctr;
  e.f2 = v3(nb.y * ctr.z - nb.z * ctr.y, nb.z * ctr.x - nb.x * ctr.z, nb.x * ctr.y - nb.y * ctr.x);
  return e;
}
__device__ __forceinline__ V3 vn_mix(const float* __restrict__ w, const VnEdge& e) {  // w[0..2]: one output channel
  return v3(w[0] * e.f0.x + w[1] * e.f1.x + w[2] * e.f2.x, w[0] * e.f0.y + w[1] * e.f1.y + w[2] * e.f2.y,
            w[0] * e.f0.z + w[1] * e.f1.z + w[2] * e.f2.z);
}

// block-wide sum of NV per-thread values -> out[0..NV) (deterministic: shuffles, then a fixed-order sum over the waves)
template <int NV>
__device__ __forceinline__ void vn_block_sum(float (&v)[NV], float* s_red /* [waves][NV] */, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();  // the scratch may still be read from a previous call
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum_f(v[i]);
    if (lane == 0) s_red[wave * NV + i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NV; i += kVnThreads) {
    float s = 0.f;
    for (int w = 0; w < kVnThreads / 64; ++w) s += s_red[w * NV + i];
    out[i] = s;
  }
}

// The same in two halves for sums formed inside a loop: every wave parks its NV sums at s_red[wave][at + i] (no barrier), and after
// the loop one barrier and a fixed-order sum over the waves writes all `total` values.
template <int NV>
__device__ __forceinline__ void vn_wave_part(const float (&v)[NV], float* s_red, int total, int at) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = wave_sum_f(v[i]);
    if (lane == 0) s_red[wave * total + at + i] = s;
  }
}
__device__ __forceinline__ void vn_block_finish(const float* s_red, int total, float* __restrict__ out) {
  __syncthreads();
  for (int i = threadIdx.x; i < total; i += kVnThreads) {
    float s = 0.f;
    for (int w = 0; w < kVnThreads / 64; ++w) s += s_red[w * total + i];
    out[i] = s;
  }
}

// Four lanes per point (vn_common.hpp): the quad's distributed sorted list IS the neighbour list, lane `sub` stores the ranks
// [SEG sub, SEG sub + SEG) below k.  (One thread per point this was the largest kernel of a training step at B = 64: 157 us.)
template <int SEG>
__global__ __launch_bounds__(kVnQThreads, 4) void vn_knn_kernel(const float* __restrict__ x, int32_t* __restrict__ idx, int N, int k) {
  extern __shared__ __attribute__((aligned(16))) float vn_smem[];
  float4* pts = reinterpret_cast<float4*>(vn_smem);
  const int b = blockIdx.y, tid = threadIdx.x, sub = tid & 3;
  const int Npad = (N + 15) & ~15;
  vn_stage_cloud_quad(x + (size_t)b * 3 * N, N, Npad, pts, tid);
  __syncthreads();
  const int n = blockIdx.x * kVnQPts + (tid >> 2);
  const bool active = n < N;
  const float4 c4 = pts[active ? n : N - 1];
  float bv[SEG];
  int bi[SEG];
  vn_knn_quad<SEG>(pts, Npad, reinterpret_cast<float2*>(vn_smem + 4 * Npad), v3(c4.x, c4.y, c4.z), c4.w, tid, bv, bi);
  if (active) {
    int32_t* o = idx + ((size_t)b * N + n) * k;
#pragma unroll
    for (int t = 0; t < SEG; ++t)
      if (SEG * sub + t < k) o[SEG * sub + t] = bi[t];
  }
}

// Four lanes per point, five of its twenty edges each (kVnSplit x kVnEdges = kVnK): with one thread per point a batch of 64
// clouds is 65 k threads = one wave per SIMD walking 20 edges x 21 channels of dependent arithmetic (the four passes took
// 0.05 + 0.13 + 0.15 + 0.19 ms); split, there are four waves per SIMD and a quarter of the chain each.  Sums over a point's
// edges: the block sums already cover all lanes; the forward pass adds the four lanes of a quad with two DPP quad permutes.
// Any k <= 32: lane `sub` takes the edges [E sub, min(E sub + E, k)), E = ceil(k / 4); the kernels are instantiated for E <= 5
// (k <= 20; k = 20 is E = 5 exactly, every lane busy) and E <= 8 (k <= 32).
constexpr int kVnSplit = 4, kVnPts = kVnThreads / kVnSplit;

// common prologue of the four conv_pos passes: cloud in LDS, this thread's point and its share of the neighbour list
#define VN_PASS_PROLOGUE                                                                           \
  extern __shared__ __attribute__((aligned(16))) float vn_smem[];                                  \
  float4* pts = reinterpret_cast<float4*>(vn_smem);                                                \
  const int b = blockIdx.y, tid = threadIdx.x;                                                     \
  const int Npad = (N + 3) & ~3;                                                                   \
  vn_stage_cloud(x + (size_t)b * 3 * N, N, Npad, pts, tid);                                        \
  __syncthreads();                                                                                 \
  const int n = blockIdx.x * kVnPts + tid / kVnSplit, sub = tid % kVnSplit;                        \
  const bool active = n < N;                                                                       \
  const float4 c4 = pts[active ? n : N - 1];                                                       \
  const V3 ctr = v3(c4.x, c4.y, c4.z);                                                             \
  const int edges = (k + kVnSplit - 1) / kVnSplit;                                                 \
  const int cnt = min(max(k - sub * edges, 0), edges);                                             \
  const int32_t* nbr = idx + ((size_t)b * N + (active ? n : N - 1)) * k + min(sub * edges, k - 1);

template <int EMAX>
__global__ __launch_bounds__(kVnThreads) void vn_convpos_stats_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                                     const float* __restrict__ Wf, float* __restrict__ partial,
                                                                     int N, int k) {
  __shared__ float s_red[(kVnThreads / 64) * 2 * kVnC];
  VN_PASS_PROLOGUE
  // channel-outer, edge-inner (as the fused eval kernel): the lane's edges are set up once, a channel's three weights are loaded
  // once for all of them, and the EMAX norm chains of a channel are independent
  VnEdge e[EMAX];
  float live[EMAX];
#pragma unroll
  for (int t = 0; t < EMAX; ++t) {
    e[t] = vn_edge(ctr, pts[nbr[t < cnt ? t : 0]]);
    live[t] = active && t < cnt ? 1.0f : 0.0f;
  }
  float acc[2 * kVnC];
#pragma unroll
  for (int c = 0; c < kVnC; ++c) {
    asm volatile("" ::: "memory");  // one channel at a time (see pointcloud.hip)
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int t = 0; t < EMAX; ++t) {
      const V3 q = vn_mix(Wf + 3 * c, e[t]);
      const float nr = live[t] * (vn_sqrt(dot3(q, q)) + kVnEps);
      s0 += nr;
      s1 += nr * nr;
    }
    acc[2 * c] = s0;
    acc[2 * c + 1] = s1;
  }
  vn_block_sum<2 * kVnC>(acc, s_red, partial + ((size_t)b * gridDim.x + blockIdx.x) * (2 * kVnC));
}

template <int EMAX>
__global__ __launch_bounds__(kVnThreads) void vn_convpos_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                                   const float* __restrict__ Wf, const float* __restrict__ Wd,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   float* __restrict__ pooled, int N, int k) {
  VN_PASS_PROLOGUE
  VnEdge e[EMAX];
  float m[EMAX];
#pragma unroll
  for (int t = 0; t < EMAX; ++t) {
    e[t] = vn_edge(ctr, pts[nbr[t < cnt ? t : 0]]);
    m[t] = t < cnt ? 1.0f : 0.0f;   // (k = 20: every lane has its five edges and m == 1 throughout)
  }
  V3 acc[kVnC];
#pragma unroll
  for (int c = 0; c < kVnC; ++c) {
    asm volatile("" ::: "memory");  // one channel at a time
    const float sc = scale[c], sh = shift[c];
    V3 a = v3(0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < EMAX; ++t) {
      const V3 q = vn_relu_sel(vn_bn(vn_mix(Wf + 3 * c, e[t]), sc, sh), vn_mix(Wd + 3 * c, e[t]));
      a.x += m[t] * q.x; a.y += m[t] * q.y; a.z += m[t] * q.z;
    }
    acc[c] = a;
  }
  // the four lanes of a point: butterfly over the quad, every lane ends up with the point's sums; lane `sub` stores channels
  // c = sub, sub + 4, ...
  auto quad_sum = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));  // [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));  // [2,3,0,1]
    return v;
  };
  const float inv_k = 1.0f / (float)k;
  float* o = pooled + (size_t)b * kVnC * 3 * N + (active ? n : N - 1);  // (B, 21, 3, N)
#pragma unroll
  for (int c = 0; c < kVnC; ++c) {
    const float sx = quad_sum(acc[c].x), sy = quad_sum(acc[c].y), sz = quad_sum(acc[c].z);
    if (active && (c % kVnSplit) == sub) {
      o[(size_t)(3 * c) * N] = sx * inv_k;
      o[(size_t)(3 * c + 1) * N] = sy * inv_k;
      o[(size_t)(3 * c + 2) * N] = sz * inv_k;
    }
  }
}

// One (edge, channel) of the block: mix the edge features, then the shared layer gradient (vn_common.hpp)
__device__ __forceinline__ VnGrad vn_edge_grad(const float* __restrict__ wf, const float* __restrict__ wd, const VnEdge& e,
                                               float scale, float shift, const V3& g_out) {
  return vn_gate_grad(vn_mix(wf, e), vn_mix(wd, e), scale, shift, g_out);
}

template <int EMAX>
__global__ __launch_bounds__(kVnThreads) void vn_convpos_bwd_reduce_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                                          const float* __restrict__ Wf, const float* __restrict__ Wd,
                                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                          const float* __restrict__ gpool, float* __restrict__ partial,
                                                                          int N, int k) {
  __shared__ float s_red[(kVnThreads / 64) * 2 * kVnC];   // per wave, all channels: ONE barrier after the channel loop, not two per channel
  VN_PASS_PROLOGUE
  const float inv_k = active ? 1.0f / (float)k : 0.0f;  // idle threads contribute nothing
  const float* gp = gpool + (size_t)b * kVnC * 3 * N + (active ? n : N - 1);
  int nb[EMAX];
#pragma unroll
  for (int t = 0; t < EMAX; ++t) nb[t] = nbr[t < cnt ? t : 0];
  float* out = partial + ((size_t)b * gridDim.x + blockIdx.x) * (2 * kVnC);
  // channel by channel (the output gradient of a channel is the same for all k edges of the point): two accumulators live
#pragma unroll 1
  for (int c = 0; c < kVnC; ++c) {
    const V3 g_out = v3(gp[(size_t)(3 * c) * N] * inv_k, gp[(size_t)(3 * c + 1) * N] * inv_k, gp[(size_t)(3 * c + 2) * N] * inv_k);
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c];
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < EMAX; ++t) {
      if (t < cnt) {   // (k = 20, EMAX = 5: always true)
        const VnGrad r = vn_edge_grad(Wf + 3 * c, Wd + 3 * c, vn_edge(ctr, pts[nb[t]]), sc, sh, g_out);
        acc[0] += r.g_nbn;
        acc[1] += r.g_nbn * (r.nr - mu) * rs;
      }
    }
    vn_wave_part<2>(acc, s_red, 2 * kVnC, 2 * c);
  }
  vn_block_finish(s_red, 2 * kVnC, out);
}

// dW_f[c][i] = sum <g_q, f_i>, dW_d[c][i] = sum <g_d, f_i>;  g_q = dL/dq through the direction u = q/n and through the norm:
//   g_n = gamma rstd (g_nbn - m1 - nhat m2)   (m1 = sum g_nbn / M, m2 = sum g_nbn nhat / M; both 0 with running statistics)
//   g_q = (g_u - u <g_u, q> / |q|) / n + g_n q / |q|,   g_u = g_qn * nbn
template <int EMAX>
__global__ __launch_bounds__(kVnThreads) void vn_convpos_bwd_apply_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                                         const float* __restrict__ Wf, const float* __restrict__ Wd,
                                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                         const float* __restrict__ m1, const float* __restrict__ m2,
                                                                         const float* __restrict__ gpool, float* __restrict__ partial,
                                                                         int N, int k) {
  __shared__ float s_red[(kVnThreads / 64) * 6 * kVnC];
  VN_PASS_PROLOGUE
  const float inv_k = active ? 1.0f / (float)k : 0.0f;
  const float* gp = gpool + (size_t)b * kVnC * 3 * N + (active ? n : N - 1);
  int nb[EMAX];
#pragma unroll
  for (int t = 0; t < EMAX; ++t) nb[t] = nbr[t < cnt ? t : 0];
  float* out = partial + ((size_t)b * gridDim.x + blockIdx.x) * (6 * kVnC);  // [c][W_f 0..2, W_d 0..2]
#pragma unroll 1
  for (int c = 0; c < kVnC; ++c) {
    const V3 g_out = v3(gp[(size_t)(3 * c) * N] * inv_k, gp[(size_t)(3 * c + 1) * N] * inv_k, gp[(size_t)(3 * c + 2) * N] * inv_k);
    const float sc = scale[c], sh = shift[c], mu = mean[c], rs = rstd[c], mm1 = m1[c], mm2 = m2[c];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < EMAX; ++t) {
      if (t < cnt) {
        const VnEdge e = vn_edge(ctr, pts[nb[t]]);
        const VnGrad r = vn_edge_grad(Wf + 3 * c, Wd + 3 * c, e, sc, sh, g_out);
        const V3 g_q = vn_norm_input_grad(r, sc, mu, rs, mm1, mm2, active);
        acc[0] += dot3(g_q, e.f0);
        acc[1] += dot3(g_q, e.f1);
        acc[2] += dot3(g_q, e.f2);
        acc[3] += dot3(r.g_d, e.f0);
        acc[4] += dot3(r.g_d, e.f1);
        acc[5] += dot3(r.g_d, e.f2);
      }
    }
    vn_wave_part<6>(acc, s_red, 6 * kVnC, 6 * c);
  }
  vn_block_finish(s_red, 6 * kVnC, out);
}

inline int vn_check(const void* x, const void* idx, int B, int N, int k, size_t& lds, bool knn) {
  if (B < 0 || N <= 0 || k <= 0) return EQA_ERR_INVALID_ARG;
  if (N < k || k > 32) return EQA_ERR_UNSUPPORTED;
  lds = knn ? ((size_t)4 * ((N + 15) & ~15) + 2 * kVnQSlots * kVnQThreads) * sizeof(float) : (size_t)4 * ((N + 3) & ~3) * sizeof(float);
  // the kNN kernel raises its dynamic-LDS limit (eqa_vn_knn); the four passes hold only the cloud and stay inside the default 64 KB
  if (lds > (knn ? 128 : 64) * 1024 || B > 65535) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  if (!x || !idx) return EQA_ERR_INVALID_ARG;
  return 1;  // go
}

}  // namespace

extern "C" {

// blocks per cloud of the four conv_pos passes (32 points per block: four lanes per point); the kNN kernel keeps one thread per point
int eqa_vn_blocks(int N) { return N <= 0 ? 0 : (N + kVnPts - 1) / kVnPts; }
static int vn_knn_blocks(int N) { return (N + kVnQPts - 1) / kVnQPts; }

int eqa_vn_knn(const float* x, int32_t* idx, int B, int N, int k, void* stream) {
  size_t lds;
  const int rc = vn_check(x, idx, B, N, k, lds, true);
  if (rc != 1) return rc;
  const dim3 grid(vn_knn_blocks(N), B);
  if (lds > 64 * 1024 && !allow_dynamic_lds(k <= 20 ? (const void*)vn_knn_kernel<5> : (const void*)vn_knn_kernel<8>, 128 * 1024))
    return EQA_ERR_UNSUPPORTED;
  if (k <= 20)
    hipLaunchKernelGGL(vn_knn_kernel<5>, grid, dim3(kVnQThreads), lds, (hipStream_t)stream, x, idx, N, k);
  else
    hipLaunchKernelGGL(vn_knn_kernel<8>, grid, dim3(kVnQThreads), lds, (hipStream_t)stream, x, idx, N, k);
  return launch_status();
}

#define VN_LAUNCH_E(kernel, ...)                                                                                          \
  do {                                                                                                                    \
    if (k <= 20)                                                                                                          \
      hipLaunchKernelGGL(kernel<5>, dim3(eqa_vn_blocks(N), B), dim3(kVnThreads), lds, (hipStream_t)stream, __VA_ARGS__);  \
    else                                                                                                                  \
      hipLaunchKernelGGL(kernel<8>, dim3(eqa_vn_blocks(N), B), dim3(kVnThreads), lds, (hipStream_t)stream, __VA_ARGS__);  \
  } while (0)

int eqa_vn_convpos_stats(const float* x, const int32_t* idx, const float* Wf, float* partial, int B, int N, int k, void* stream) {
  size_t lds;
  const int rc = vn_check(x, idx, B, N, k, lds, false);
  if (rc != 1) return rc;
  if (!Wf || !partial) return EQA_ERR_INVALID_ARG;
  VN_LAUNCH_E(vn_convpos_stats_kernel, x, idx, Wf, partial, N, k);
  return launch_status();
}

int eqa_vn_convpos_fwd(const float* x, const int32_t* idx, const float* Wf, const float* Wd, const float* scale, const float* shift,
                       float* pooled, int B, int N, int k, void* stream) {
  size_t lds;
  const int rc = vn_check(x, idx, B, N, k, lds, false);
  if (rc != 1) return rc;
  if (!Wf || !Wd || !scale || !shift || !pooled) return EQA_ERR_INVALID_ARG;
  VN_LAUNCH_E(vn_convpos_fwd_kernel, x, idx, Wf, Wd, scale, shift, pooled, N, k);
  return launch_status();
}

int eqa_vn_convpos_bwd_reduce(const float* x, const int32_t* idx, const float* Wf, const float* Wd, const float* scale,
                              const float* shift, const float* mean, const float* rstd, const float* gpool, float* partial, int B,
                              int N, int k, void* stream) {
  size_t lds;
  const int rc = vn_check(x, idx, B, N, k, lds, false);
  if (rc != 1) return rc;
  if (!Wf || !Wd || !scale || !shift || !mean || !rstd || !gpool || !partial) return EQA_ERR_INVALID_ARG;
  VN_LAUNCH_E(vn_convpos_bwd_reduce_kernel, x, idx, Wf, Wd, scale, shift, mean, rstd, gpool, partial, N, k);
  return launch_status();
}

int eqa_vn_convpos_bwd_apply(const float* x, const int32_t* idx, const float* Wf, const float* Wd, const float* scale,
                             const float* shift, const float* mean, const float* rstd, const float* m1, const float* m2,
                             const float* gpool, float* partial, int B, int N, int k, void* stream) {
  size_t lds;
  const int rc = vn_check(x, idx, B, N, k, lds, false);
  if (rc != 1) return rc;
  if (!Wf || !Wd || !scale || !shift || !mean || !rstd || !m1 || !m2 || !gpool || !partial) return EQA_ERR_INVALID_ARG;
  VN_LAUNCH_E(vn_convpos_bwd_apply_kernel, x, idx, Wf, Wd, scale, shift, mean, rstd, m1, m2, gpool, partial, N, k);
  return launch_status();
}

}  // extern "C"
