// libeqa_hip.so, part 9 -- training passes of VNSmall's tail (the rest of P2): on the pooled (B, 21, 3, N) features of the first
// block,  conv1 = VNLinearLeakyReLU(21 -> 21, slope 0)  ->  bn1 = VNBatchNorm(21)  ->  conv2 = VNLinearLeakyReLU(21 -> 4, slope 0)
// ->  dropout  ->  mean over the points  (reference: pointcloud/canonicalization_networks/equivariant_networks.py:141-150;
// vector_neuron_layers.py:251-273, :303-324), all three batch-norms in TRAINING mode (statistics over the B*N points).
// Op by op this is ~170 element-wise launches on (B, 21, 3, N) tensors plus four weight-gradient GEMMs of the shape
// (21 x 196608) . (196608 x 21), for which the library picks one 32 x 32 tile per CU-less grid: 1.5 of the 4.2 ms of a B = 64
// training step.  Here every pass recomputes the chain from the 63 floats of a point held in registers (5 kFLOP: nothing),
// one thread per point, one wave per block:
//   pass 0..2   per-channel sums of n, n^2 of conv1 / bn1 / conv2     -> eqa_vn_bn_finalize -> scale, shift, mean, rstd (+ running
//               statistics, as nn.BatchNorm1d updates them)
//   pass 3      sum over the block's points of dropout(conv2 output)  -> (blocks, 12) partials of the mean
//   pass 4      sums of g, g * nhat of conv2's batch-norm             -> eqa_vn_bn_bwd_finalize -> d beta, d gamma, m1, m2
//   pass 5      the same for bn1, and d W of conv2 (rows of the block staged in LDS, 168 dot products over the 64 points)
//   pass 6      the same for conv1's batch-norm, and d W_d of conv1 (441 dot products per block)
//   pass 7      d W_f of conv1 and the gradient w.r.t. the pooled features
// Partials are per block and summed by the caller in a fixed order (deterministic).  C ABI: include/eqa_hip.h.
#include "vn_common.hpp"

namespace {

constexpr int kTailThreads = 64;   // one wave: block sums are shuffles, the LDS phase needs a single barrier
constexpr int kTailC2 = 4;         // conv2's output channels (12 // 3)
constexpr int kTailPitch = 68;     // floats per staged row: 64 points + 4 (16-byte rows; 16 lanes on distinct bank quads)
constexpr int kTailStat = 128;     // floats per batch-norm in `stat`: scale[32] shift[32] mean[32] rstd[32]
constexpr int kTailRed = 64;       // floats per batch-norm in `red`: m1[32] m2[32]
constexpr int kTailW1 = kVnC * kVnC, kTailW2 = kTailC2 * kVnC;
constexpr int kTailWeights = 2 * kTailW1 + 2 * kTailW2;  // W_f1 | W_d1 | W_f2 | W_d2

struct TailArgs {
  const float* P;      // (B, 21, 3, N) pooled features
  const float* W;      // kTailWeights
  const float* stat;   // [3][kTailStat]
  const float* red;    // [3][kTailRed]
  const float* mask;   // (B, 4, 3, N) dropout factors (0 or 1 / (1 - p)), or null
  const float* gout;   // (B, 4, 3) gradient of the mean (backward passes)
  float* partial;      // (blocks, partial_floats(pass))
  float* gP;           // (B, 21, 3, N), pass 7
  int N;
};

__host__ __device__ constexpr int tail_partial_floats(int pass) {
  return pass <= 1 ? 2 * kVnC : pass == 2 ? 2 * kTailC2 : pass == 3 ? 3 * kTailC2 : pass == 4 ? 2 * kTailC2
         : pass == 5 ? 2 * kVnC + 2 * kTailW2 : pass == 6 ? 2 * kVnC + kTailW1 : kTailW1;
}

__device__ __forceinline__ V3 tail_mix(const float* __restrict__ w, const V3 (&X)[kVnC]) {  // w: one output row, 21 inputs
  V3 p = v3(0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < kVnC; ++i) {
    p.x += w[i] * X[i].x; p.y += w[i] * X[i].y; p.z += w[i] * X[i].z;
  }
  return p;
}
__device__ __forceinline__ void axpy3(V3& a, float w, const V3& v) { a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; }

// dot products of staged rows: out[k] = sum_a sum_pt G[(g_row(k)) * 3 + a][pt] * Y[(y_row(k)) * 3 + a][pt]
__device__ __forceinline__ float tail_rows_dot(const float* __restrict__ G, const float* __restrict__ Y) {
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float4* g4 = reinterpret_cast<const float4*>(G + a * kTailPitch);
    const float4* y4 = reinterpret_cast<const float4*>(Y + a * kTailPitch);
#pragma unroll 4
    for (int j = 0; j < kTailThreads / 4; ++j) {
      const float4 g = g4[j], y = y4[j];
      acc += g.x * y.x + g.y * y.y + g.z * y.z + g.w * y.w;
    }
  }
  return acc;
}

// The pointers are kernel arguments of their own (const __restrict__), not members of a struct: only then does the compiler
// prove the weights and statistics read-only and fetch them with scalar loads (as struct members every weight was a per-lane
// global_load inside the channel loop: pass 7 took 127 us instead of 45).
template <int PASS>
__global__ __launch_bounds__(kTailThreads) void vn_tail_kernel(const float* __restrict__ aP, const float* __restrict__ aW,
                                                              const float* __restrict__ a_stat, const float* __restrict__ a_red,
                                                              const float* __restrict__ a_mask, const float* __restrict__ a_gout,
                                                              float* __restrict__ a_partial, float* __restrict__ a_gP, int N) {
  constexpr int kRows = PASS == 5 ? 3 * (kVnC + 2 * kTailC2) : PASS >= 6 ? 3 * 2 * kVnC : 1;  // 34 KB: four blocks per CU
  __shared__ __attribute__((aligned(16))) float rows[kRows * kTailPitch];
  const int b = blockIdx.y, lane = threadIdx.x;
  const int n = blockIdx.x * kTailThreads + lane;
  const bool active = n < N;
  const int nn = active ? n : N - 1;
  const float* __restrict__ Wf1 = aW;
  const float* __restrict__ Wd1 = aW + kTailW1;
  const float* __restrict__ Wf2 = aW + 2 * kTailW1;
  const float* __restrict__ Wd2 = aW + 2 * kTailW1 + kTailW2;
  const float* __restrict__ st1 = a_stat;
  const float* __restrict__ st2 = a_stat + kTailStat;
  const float* __restrict__ st3 = a_stat + 2 * kTailStat;
  float* __restrict__ out = a_partial + ((size_t)b * gridDim.x + blockIdx.x) * tail_partial_floats(PASS);

  V3 X[kVnC];
  {
    const float* p = aP + (size_t)b * kVnC * 3 * N + nn;
#pragma unroll
    for (int i = 0; i < kVnC; ++i) X[i] = v3(p[(size_t)(3 * i) * N], p[(size_t)(3 * i + 1) * N], p[(size_t)(3 * i + 2) * N]);
  }

  // ---- forward chain, channel by channel; conv2's eight mixes accumulate as its inputs appear
  V3 p3[kTailC2], d3[kTailC2];
#pragma unroll
  for (int o = 0; o < kTailC2; ++o) p3[o] = d3[o] = v3(0.f, 0.f, 0.f);
#pragma unroll 1
  for (int c = 0; c < kVnC; ++c) {
    asm volatile("" ::: "memory");  // weights stay scalar loads inside the loop (not hoisted into VGPRs)
    const V3 p1 = tail_mix(Wf1 + c * kVnC, X);
    if constexpr (PASS == 0) {
      const float nr = active ? sqrtf(dot3(p1, p1)) + kVnEps : 0.f;
      const float s0 = wave_sum_f(nr), s1 = wave_sum_f(nr * nr);
      if (lane == 0) { out[2 * c] = s0; out[2 * c + 1] = s1; }
      continue;
    }
    const V3 y1 = vn_relu(vn_bn(p1, st1[c], st1[32 + c]), tail_mix(Wd1 + c * kVnC, X));
    if constexpr (PASS == 1) {
      const float m = active ? sqrtf(dot3(y1, y1)) + kVnEps : 0.f;
      const float s0 = wave_sum_f(m), s1 = wave_sum_f(m * m);
      if (lane == 0) { out[2 * c] = s0; out[2 * c + 1] = s1; }
      continue;
    }
    const V3 y2 = vn_bn(y1, st2[c], st2[32 + c]);
#pragma unroll
    for (int o = 0; o < kTailC2; ++o) {
      axpy3(p3[o], Wf2[o * kVnC + c], y2);
      axpy3(d3[o], Wd2[o * kVnC + c], y2);
    }
  }
  if constexpr (PASS <= 1) return;
  if constexpr (PASS == 2) {
#pragma unroll
    for (int o = 0; o < kTailC2; ++o) {
      const float nr = active ? sqrtf(dot3(p3[o], p3[o])) + kVnEps : 0.f;
      const float s0 = wave_sum_f(nr), s1 = wave_sum_f(nr * nr);
      if (lane == 0) { out[2 * o] = s0; out[2 * o + 1] = s1; }
    }
    return;
  }
  // dropout factors of this point's 12 outputs (0 for idle threads: they add nothing to the mean and receive no gradient)
  float keep[3 * kTailC2];
#pragma unroll
  for (int j = 0; j < 3 * kTailC2; ++j)
    keep[j] = !active ? 0.f : a_mask ? a_mask[((size_t)b * 3 * kTailC2 + j) * N + nn] : 1.f;
  if constexpr (PASS == 3) {
#pragma unroll
    for (int o = 0; o < kTailC2; ++o) {
      const V3 y3 = vn_relu(vn_bn(p3[o], st3[o], st3[32 + o]), d3[o]);
      const float s0 = wave_sum_f(y3.x * keep[3 * o]), s1 = wave_sum_f(y3.y * keep[3 * o + 1]), s2 = wave_sum_f(y3.z * keep[3 * o + 2]);
      if (lane == 0) { out[3 * o] = s0; out[3 * o + 1] = s1; out[3 * o + 2] = s2; }
    }
    return;
  }

  // ---- backward.  mean over N, then dropout: g_y3 = gout / N * keep
  if constexpr (PASS >= 4) {
    const float inv_N = 1.0f / (float)N;
    const float* __restrict__ rd1 = a_red;
    const float* __restrict__ rd2 = a_red + kTailRed;
    const float* __restrict__ rd3 = a_red + 2 * kTailRed;
    V3 gp3[kTailC2], gd3[kTailC2];
#pragma unroll
    for (int o = 0; o < kTailC2; ++o) {
      const float* g = a_gout + ((size_t)b * kTailC2 + o) * 3;
      const V3 g_y3 = v3(g[0] * inv_N * keep[3 * o], g[1] * inv_N * keep[3 * o + 1], g[2] * inv_N * keep[3 * o + 2]);
      const VnGrad r = vn_gate_grad(p3[o], d3[o], st3[o], st3[32 + o], g_y3);
      if constexpr (PASS == 4) {
        const float s0 = wave_sum_f(r.g_nbn), s1 = wave_sum_f(r.g_nbn * (r.nr - st3[64 + o]) * st3[96 + o]);
        if (lane == 0) { out[2 * o] = s0; out[2 * o + 1] = s1; }
      } else {
        gp3[o] = vn_norm_input_grad(r, st3[o], st3[64 + o], st3[96 + o], rd3[o], rd3[32 + o], active);
        gd3[o] = r.g_d;
      }
    }
    if constexpr (PASS == 4) return;

    if constexpr (PASS == 5) {  // rows: y2 (63) | g_p3 (12) | g_d3 (12)
#pragma unroll
      for (int o = 0; o < kTailC2; ++o) {
        float* r0 = rows + (3 * kVnC + 3 * o) * kTailPitch + lane;
        r0[0] = gp3[o].x; r0[kTailPitch] = gp3[o].y; r0[2 * kTailPitch] = gp3[o].z;
        float* r1 = r0 + 3 * kTailC2 * kTailPitch;
        r1[0] = gd3[o].x; r1[kTailPitch] = gd3[o].y; r1[2 * kTailPitch] = gd3[o].z;
      }
    }
    V3 gX[kVnC];
    if constexpr (PASS == 7) {
#pragma unroll
      for (int i = 0; i < kVnC; ++i) gX[i] = v3(0.f, 0.f, 0.f);
    }
#pragma unroll 1
    for (int c = 0; c < kVnC; ++c) {
      asm volatile("" ::: "memory");
      const V3 p1 = tail_mix(Wf1 + c * kVnC, X), d1 = tail_mix(Wd1 + c * kVnC, X);
      const V3 y1 = vn_relu(vn_bn(p1, st1[c], st1[32 + c]), d1);
      V3 g_y2 = v3(0.f, 0.f, 0.f);
#pragma unroll
      for (int o = 0; o < kTailC2; ++o) {
        axpy3(g_y2, Wf2[o * kVnC + c], gp3[o]);
        axpy3(g_y2, Wd2[o * kVnC + c], gd3[o]);
      }
      const VnGrad rb = vn_norm_grad(y1, st2[c], st2[32 + c], g_y2);
      if constexpr (PASS == 5) {
        const float s0 = wave_sum_f(rb.g_nbn), s1 = wave_sum_f(rb.g_nbn * (rb.nr - st2[64 + c]) * st2[96 + c]);
        if (lane == 0) { out[2 * c] = s0; out[2 * c + 1] = s1; }
        float* r0 = rows + 3 * c * kTailPitch + lane;  // y2 = u * nbn; idle threads stage zeros
        const float f = active ? rb.nbn : 0.f;
        r0[0] = rb.u.x * f; r0[kTailPitch] = rb.u.y * f; r0[2 * kTailPitch] = rb.u.z * f;
        continue;
      }
      const V3 g_y1 = vn_norm_input_grad(rb, st2[c], st2[64 + c], st2[96 + c], rd2[c], rd2[32 + c], active);
      const VnGrad ra = vn_gate_grad(p1, d1, st1[c], st1[32 + c], g_y1);
      if constexpr (PASS == 6) {  // rows: g_d1 (63) | X (63)  (the gate's gradient needs no batch sums of this layer)
        const float s0 = wave_sum_f(ra.g_nbn), s1 = wave_sum_f(ra.g_nbn * (ra.nr - st1[64 + c]) * st1[96 + c]);
        if (lane == 0) { out[2 * c] = s0; out[2 * c + 1] = s1; }
        float* r0 = rows + 3 * c * kTailPitch + lane;
        r0[0] = ra.g_d.x; r0[kTailPitch] = ra.g_d.y; r0[2 * kTailPitch] = ra.g_d.z;
        continue;
      }
      if constexpr (PASS == 7) {  // rows: g_p1 (63) | X (63)
        const V3 g_p1 = vn_norm_input_grad(ra, st1[c], st1[64 + c], st1[96 + c], rd1[c], rd1[32 + c], active);
        float* r0 = rows + 3 * c * kTailPitch + lane;
        r0[0] = g_p1.x; r0[kTailPitch] = g_p1.y; r0[2 * kTailPitch] = g_p1.z;
#pragma unroll
        for (int i = 0; i < kVnC; ++i) {
          axpy3(gX[i], Wf1[c * kVnC + i], g_p1);
          axpy3(gX[i], Wd1[c * kVnC + i], ra.g_d);
        }
      }
    }
    if constexpr (PASS == 5) {
      __syncthreads();
      float* o2 = out + 2 * kVnC;  // d W_f2 (4 x 21) | d W_d2 (4 x 21)
      for (int k = lane; k < 2 * kTailW2; k += kTailThreads) {
        const int mat = k / kTailW2, r = k - mat * kTailW2, o = r / kVnC, c = r - o * kVnC;
        o2[k] = tail_rows_dot(rows + (3 * kVnC + 3 * (mat * kTailC2 + o)) * kTailPitch, rows + 3 * c * kTailPitch);
      }
    }
    if constexpr (PASS >= 6) {  // pass 6: d W_d1 (21 x 21) behind the 42 sums; pass 7: d W_f1 (21 x 21) and g_pooled
      float* gp = a_gP + (size_t)b * kVnC * 3 * N + n;
#pragma unroll
      for (int i = 0; i < kVnC; ++i) {
        float* r2 = rows + (3 * kVnC + 3 * i) * kTailPitch + lane;
        const float f = active ? 1.f : 0.f;
        r2[0] = X[i].x * f; r2[kTailPitch] = X[i].y * f; r2[2 * kTailPitch] = X[i].z * f;
        if constexpr (PASS == 7) {
          if (active) {
            gp[(size_t)(3 * i) * N] = gX[i].x; gp[(size_t)(3 * i + 1) * N] = gX[i].y; gp[(size_t)(3 * i + 2) * N] = gX[i].z;
          }
        }
      }
      __syncthreads();
      float* ow = PASS == 6 ? out + 2 * kVnC : out;
      for (int k = lane; k < kTailW1; k += kTailThreads) {
        const int c = k / kVnC, i = k - c * kVnC;
        ow[k] = tail_rows_dot(rows + 3 * c * kTailPitch, rows + (3 * kVnC + 3 * i) * kTailPitch);
      }
    }
  }
}

// sum over the blocks j = slice, slice + 16, ... of one column of the partials; eight loads in flight (one at a time the loop
// is a chain of L2 round trips: 18 us for 1024 blocks)
__device__ __forceinline__ double tail_column_sum(const float* __restrict__ col, int nblk, int stride, int slice) {
  double acc = 0.0;
  int j = slice;
  for (; j + 7 * 16 < nblk; j += 8 * 16) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = col[(size_t)(j + 16 * u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (double)t[u];
  }
  for (; j < nblk; j += 16) acc += (double)col[(size_t)j * stride];
  return acc;
}

// partial:(nblk, stride) block sums (sum n at 2c, sum n^2 at 2c+1) -> stat = scale | shift | mean | rstd of one batch-norm over M
// samples, and the running statistics updated like nn.BatchNorm1d in training mode (unbiased variance, momentum).
__global__ __launch_bounds__(1024) void vn_bn_finalize_kernel(const float* __restrict__ partial, int nblk, int stride, int C, double M,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             long long* __restrict__ num_batches_tracked, float momentum, float eps,
                                                             float* __restrict__ stat) {
  __shared__ double s[16][64];
  const int v = threadIdx.x & 63, slice = threadIdx.x >> 6;
  s[slice][v] = v < 2 * C ? tail_column_sum(partial + v, nblk, stride, slice) : 0.0;
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < 16; ++k) { s0 += s[k][2 * c]; s1 += s[k][2 * c + 1]; }
    const double mean = s0 / M;
    double var = s1 / M - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float meanf = (float)mean, varf = (float)var;
    const float rstd = 1.0f / sqrtf(varf + eps);
    const float scale = gamma[c] * rstd;
    stat[c] = scale;
    stat[32 + c] = beta[c] - meanf * scale;
    stat[64 + c] = meanf;
    stat[96 + c] = rstd;
    if (running_mean) {
      const float unbiased = (float)(var * (M / (M > 1.0 ? M - 1.0 : 1.0)));
      running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * meanf;
      running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  if (threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
}

// partial:(nblk, stride) block sums (sum g at 2c, sum g * nhat at 2c+1) -> grads = d beta[32] | d gamma[32], red = m1[32] | m2[32]
__global__ __launch_bounds__(1024) void vn_bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int stride, int C, double M,
                                                                 float* __restrict__ grads, float* __restrict__ red) {
  __shared__ double s[16][64];
  const int v = threadIdx.x & 63, slice = threadIdx.x >> 6;
  s[slice][v] = v < 2 * C ? tail_column_sum(partial + v, nblk, stride, slice) : 0.0;
  __syncthreads();
  if (threadIdx.x < 2 * C) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += s[k][threadIdx.x];
    const int c = threadIdx.x >> 1, which = threadIdx.x & 1;
    grads[32 * which + c] = (float)t;
    red[32 * which + c] = (float)(t / M);
  }
}

template <int PASS>
int launch_tail(const TailArgs& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(vn_tail_kernel<PASS>, dim3((a.N + kTailThreads - 1) / kTailThreads, B), dim3(kTailThreads), 0, st, a.P, a.W, a.stat,
                     a.red, a.mask, a.gout, a.partial, a.gP, a.N);
  return launch_status();
}

}  // namespace

extern "C" {

int eqa_vn_tail_blocks(int N) { return N <= 0 ? 0 : (N + kTailThreads - 1) / kTailThreads; }

int eqa_vn_tail_partial_floats(int pass) { return pass < 0 || pass > 7 ? 0 : tail_partial_floats(pass); }

int eqa_vn_tail_pass(int pass, const float* pooled, const float* weights, const float* stat, const float* red, const float* mask,
                     const float* gout, float* partial, float* g_pooled, int B, int N, void* stream) {
  if (pass < 0 || pass > 7 || B < 0 || N <= 0) return EQA_ERR_INVALID_ARG;
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  if (!pooled || !weights || !partial || (pass >= 1 && !stat) || (pass >= 4 && !gout) || (pass >= 5 && !red) || (pass == 7 && !g_pooled))
    return EQA_ERR_INVALID_ARG;
  const TailArgs a{pooled, weights, stat, red, mask, gout, partial, g_pooled, N};
  hipStream_t st = (hipStream_t)stream;
  switch (pass) {
    case 0: return launch_tail<0>(a, B, st);
    case 1: return launch_tail<1>(a, B, st);
    case 2: return launch_tail<2>(a, B, st);
    case 3: return launch_tail<3>(a, B, st);
    case 4: return launch_tail<4>(a, B, st);
    case 5: return launch_tail<5>(a, B, st);
    case 6: return launch_tail<6>(a, B, st);
    default: return launch_tail<7>(a, B, st);
  }
}

int eqa_vn_bn_finalize(const float* partial, int nblk, int stride, int C, long long M, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* stat,
                       void* stream) {
  if (!partial || !gamma || !beta || !stat || nblk <= 0 || C <= 0 || M <= 0 || stride < 2 * C || (running_mean && !running_var))
    return EQA_ERR_INVALID_ARG;
  if (C > 32) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(vn_bn_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nblk, stride, C, (double)M, gamma, beta,
                     running_mean, running_var, num_batches_tracked, momentum, eps, stat);
  return launch_status();
}

int eqa_vn_bn_bwd_finalize(const float* partial, int nblk, int stride, int C, long long M, float* grads, float* red, void* stream) {
  if (!partial || !grads || !red || nblk <= 0 || C <= 0 || M <= 0 || stride < 2 * C) return EQA_ERR_INVALID_ARG;
  if (C > 32) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(vn_bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, nblk, stride, C, (double)M, grads, red);
  return launch_status();
}

}  // extern "C"
