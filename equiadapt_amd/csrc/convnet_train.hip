// libeqa_hip.so, part 16 -- ConvNetwork (row I10) in TRAINING: the stride-2 convolutions' filter and data gradients on the fp32
// matrix cores, and the batch-norm (batch statistics) + activation blocks of the encoder and the head, forward and backward.
// C ABI: include/eqa_hip.h.
//
// Reference: equiadapt/images/canonicalization_networks/custom_nonequivariant_networks.py:44-80 -- per layer
// Conv2d(k, stride 2, padding 0 / 1) -> BatchNorm2d -> GELU; head BatchNorm1d -> Dropout1d(0.5) -> ReLU -> Linear.  The optimised
// canonicalizer trains it on G group views per image (tutorial cell 26/30: 2048 views of 64 x 64 per step, twice).  Through the
// framework that was MIOpen's implicit-GEMM forward / backward-data / backward-weights kernels (34 % of the loop) and ATen's
// batch-norm kernels (28 %).  Here (forward convolution: eqa_conv_s2 of smallconv.hip, with gelu = 0):
//   conv_s2_wgrad_*   dW[co][ci][u][v] = sum over output pixels of dz[p][co] * x[window(p) + (u, v)][ci].  The reduction index of
//                     the 16x16x4 matrix instruction is the PIXEL: A[i = co][k = pixel] = dz, B[k = pixel][j = ci] = x at one tap.
//                     A wave owns one filter row u and a run of output pixels; its K * (Cout / 16) * (Cin / 16) accumulator tiles go
//                     to a workspace, summed over the runs in a fixed order by conv_s2_wgrad_reduce_kernel (deterministic, no atomics).
//   conv_s2_dgrad_*   dx[pixel][ci] = sum over (tap, co) of dz[.][co] * W[co][ci][tap] -- only taps of the pixel's own parity reach
//                     a stride-2 output, so the input pixels are walked one parity class (row parity, column parity) at a time: inside a
//                     class the gradient is a dense stride-1 correlation of dz with the sub-sampled filter, the forward kernel's loop
//                     with the roles of Cin and Cout exchanged.
//   bn_act_*          y = rowscale[p] * act(scale[c] z + shift[c]) and its backward in two passes (per-channel sums of g and g * zhat
//                     as fp64 partials per block, then dz); act = exact GELU (encoder) or ReLU (head, rowscale = Dropout1d's per-row
//                     factor).  The batch statistics themselves: eqa_bn_stats_nhwc (batchnorm.hip).
#include "eqa_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOob = 0x7ffffff0u;   // a buffer offset outside every buffer: the load returns zero
// ok ? off : kOob without a branch: the compiler turns the ternary into an exec-mask region when `off` is a few instructions of
// arithmetic, and a region boundary between matrix instructions makes it copy their accumulators between the register files
__device__ __forceinline__ unsigned off_or_oob(bool ok, unsigned off) {
  const unsigned m = 0u - (unsigned)ok;
  return (off & m) | (kOob & ~m);
}

__device__ __forceinline__ float act_fwd(int act, float h) {
  return act == 0 ? 0.5f * h * (1.0f + erff(h * 0.70710678118654752440f)) : fmaxf(h, 0.0f);
}
__device__ __forceinline__ float act_grad(int act, float h) {
  if (act != 0) return h > 0.0f ? 1.0f : 0.0f;
  // d/dh [h Phi(h)] = Phi(h) + h phi(h)
  return 0.5f * (1.0f + erff(h * 0.70710678118654752440f)) + h * 0.39894228040143267794f * __expf(-0.5f * h * h);
}

// ---------------------------------------------------------------------------------------------------------------------------
// batch-norm (given per-channel scale / shift) + activation (+ per-row factor), channels-last (npix, C), C % 4 == 0
// ---------------------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(kThreads) void bn_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ rowscale,
                                                             float* __restrict__ y, size_t nquad, int Q) {
  const size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x, stride = (size_t)gridDim.x * kThreads;
  QuadWalk w(i0, stride, Q);
  for (size_t i = i0; i < nquad; i += stride, w.next()) {
    const int q = (int)w.q;
    const float4 v = reinterpret_cast<const float4*>(z)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float r = rowscale ? rowscale[w.row] : 1.0f;
    reinterpret_cast<float4*>(y)[i] = make_float4(r * act_fwd(ACT, v.x * sc.x + sh.x), r * act_fwd(ACT, v.y * sc.y + sh.y),
                                                  r * act_fwd(ACT, v.z * sc.z + sh.z), r * act_fwd(ACT, v.w * sc.w + sh.w));
  }
}

// partial[(blk * C + c) * 2 + {0, 1}] = sum, sum of squares of z over the block's pixels (bn_stats_nhwc_kernel of batchnorm.hip with the
// block's pixel count a launch parameter: the head's (2048, 1152) matrix needs 8-pixel blocks to fill the chip, the encoder's maps 256)
__global__ __launch_bounds__(kThreads) void bn_act_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, size_t npix, int C,
                                                               int pix_per_block) {
  __shared__ float4 s_red[2][kThreads];
  const int Q = C >> 2;
  const int lanes = min(Q, kThreads);
  const int rows = kThreads / lanes;
  const int q0 = threadIdx.x % lanes, r0 = threadIdx.x / lanes;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = min(npix, p0 + pix_per_block);
  for (int qb = 0; qb < Q; qb += lanes) {
    const int q = qb + q0;
    const bool has_q = q < Q;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
    // SHIFTED sums: the fp32 accumulators hold sum (x - K) and sum (x - K)^2 with K = the block's first pixel of the
    // channel, and the block's sum / sum of squares of x are put together from them in fp64 below.  Plain fp32 sums of x^2 lose
    // ~1e-7 mean^2 / var of the variance to cancellation for a channel whose mean is large against its spread (the head's
    // BatchNorm1d sits behind a GELU: means well away from 0); shifted, the loss is ~1e-7 of the variance itself.
    const float4 K = (has_q && p0 < p1) ? reinterpret_cast<const float4*>(x)[p0 * Q + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 < rows && has_q) {
      for (size_t p = p0 + r0; p < p1; p += rows) {
        float4 v = reinterpret_cast<const float4*>(x)[p * Q + q];
        v.x -= K.x; v.y -= K.y; v.z -= K.z; v.w -= K.w;
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        ss.x += v.x * v.x; ss.y += v.y * v.y; ss.z += v.z * v.z; ss.w += v.w * v.w;
      }
    }
    __syncthreads();
    s_red[0][threadIdx.x] = s;
    s_red[1][threadIdx.x] = ss;
    __syncthreads();
    if (r0 == 0 && has_q) {
      for (int r = 1; r < rows; ++r) {
        const float4 a = s_red[0][r * lanes + q0], b = s_red[1][r * lanes + q0];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        ss.x += b.x; ss.y += b.y; ss.z += b.z; ss.w += b.w;
      }
      double* o = partial + ((size_t)blockIdx.x * C + 4 * q) * 2;
      const double n = (double)(p1 - p0);
      auto put = [&](int k, float sd, float sdd, float kk) {      // sum x = sum d + n K;  sum x^2 = sum d^2 + 2 K sum d + n K^2
        o[2 * k] = (double)sd + n * (double)kk;
        o[2 * k + 1] = (double)sdd + 2.0 * (double)kk * (double)sd + n * (double)kk * (double)kk;
      };
      put(0, s.x, ss.x, K.x);
      put(1, s.y, ss.y, K.y);
      put(2, s.z, ss.z, K.z);
      put(3, s.w, ss.w, K.w);
    }
  }
}

// Sum of the per-block partials of 4 channels (block = 4 channels x 64 slices of the block index; slices combined in order: the
// result does not depend on the launch) -> out[0..1] on the threads of slice 0.
constexpr int kBnFinCh = 4, kBnFinSlices = kThreads / kBnFinCh;
__device__ __forceinline__ void sum_partials(const double* __restrict__ partial, int nblk, int C, int c0, double (&out)[2],
                                             double (*s_acc)[kBnFinCh][2]) {
  const int cl = threadIdx.x & (kBnFinCh - 1), slice = threadIdx.x / kBnFinCh;
  const int c = c0 + cl;
  double a0 = 0.0, a1 = 0.0;
  if (c < C)
    for (int b = slice; b < nblk; b += kBnFinSlices) {
      const double* p = partial + ((size_t)b * C + c) * 2;
      a0 += p[0];
      a1 += p[1];
    }
  s_acc[slice][cl][0] = a0;
  s_acc[slice][cl][1] = a1;
  __syncthreads();
  out[0] = out[1] = 0.0;
  if (slice == 0)
    for (int s = 0; s < kBnFinSlices; ++s) {
      out[0] += s_acc[s][cl][0];
      out[1] += s_acc[s][cl][1];
    }
}

// Forward finalize: batch mean / biased variance from the partial sums, folded with the affine parameters, and the module's running
// statistics updated as nn.BatchNorm*d does (momentum < 0: cumulative average with the already incremented batch counter).
__global__ __launch_bounds__(kThreads) void bn_act_finalize_kernel(const double* __restrict__ partial, int nblk, int C, double npix,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, double eps,
                                                                  double momentum, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, float* __restrict__ scale,
                                                                  float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ double s_acc[kBnFinSlices][kBnFinCh][2];
  double s[2];
  const int c0 = blockIdx.x * kBnFinCh;
  sum_partials(partial, nblk, C, c0, s, s_acc);
  const int c = c0 + (threadIdx.x & (kBnFinCh - 1));
  if (threadIdx.x >= kBnFinCh || c >= C) return;
  const double m = s[0] / npix;
  const double var = fmax(s[1] / npix - m * m, 0.0);
  const double r = 1.0 / sqrt(var + eps);
  const double sc = (double)gamma[c] * r;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - m * sc);
  mean[c] = (float)m;
  rstd[c] = (float)r;
  if (running_mean) {
    const double unbiased = var * (npix / fmax(npix - 1.0, 1.0));
    running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
  }
}

// Backward finalize: dgamma = sum(g zhat), dbeta = sum(g), and the three per-channel coefficients of eqa_bn_act_bwd_apply
__global__ __launch_bounds__(kThreads) void bn_act_bwd_finalize_kernel(const double* __restrict__ partial, int nblk, int C, double npix,
                                                                      const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      float* __restrict__ gscale, float* __restrict__ m1, float* __restrict__ m2) {
  __shared__ double s_acc[kBnFinSlices][kBnFinCh][2];
  double s[2];
  const int c0 = blockIdx.x * kBnFinCh;
  sum_partials(partial, nblk, C, c0, s, s_acc);
  const int c = c0 + (threadIdx.x & (kBnFinCh - 1));
  if (threadIdx.x >= kBnFinCh || c >= C) return;
  dbeta[c] = (float)s[0];
  dgamma[c] = (float)s[1];
  m1[c] = (float)(s[0] / npix);
  m2[c] = (float)(s[1] / npix);
  gscale[c] = gamma[c] * rstd[c];
}

// g = gy * rowscale * act'(scale z + shift);  partial[(blk * C + c) * 2 + {0, 1}] = sum of g, sum of g * zhat over the block's pixels
template <int ACT>
__global__ __launch_bounds__(kThreads) void bn_act_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ z,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    const float* __restrict__ rowscale, double* __restrict__ partial,
                                                                    size_t npix, int C, int pix_per_block) {
  __shared__ float4 s_red[2][kThreads];
  const int Q = C >> 2;
  const int lanes = min(Q, kThreads);
  const int rows = kThreads / lanes;
  const int q0 = threadIdx.x % lanes, r0 = threadIdx.x / lanes;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = min(npix, p0 + pix_per_block);
  for (int qb = 0; qb < Q; qb += lanes) {      // uniform trip count: every thread reaches both barriers of every trip
    const int q = qb + q0;
    const bool has_q = q < Q;
    const int qc = has_q ? q : 0;
    const float4 sc = reinterpret_cast<const float4*>(scale)[qc], sh = reinterpret_cast<const float4*>(shift)[qc];
    const float4 mu = reinterpret_cast<const float4*>(mean)[qc], rs = reinterpret_cast<const float4*>(rstd)[qc];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sx = s;
    if (r0 < rows && has_q) {
      for (size_t p = p0 + r0; p < p1; p += rows) {
        const size_t i = p * Q + q;
        const float4 g4 = reinterpret_cast<const float4*>(gy)[i];
        const float4 v = reinterpret_cast<const float4*>(z)[i];
        const float r = rowscale ? rowscale[p] : 1.0f;
        const float g0 = g4.x * r * act_grad(ACT, v.x * sc.x + sh.x), g1 = g4.y * r * act_grad(ACT, v.y * sc.y + sh.y);
        const float g2 = g4.z * r * act_grad(ACT, v.z * sc.z + sh.z), g3 = g4.w * r * act_grad(ACT, v.w * sc.w + sh.w);
        s.x += g0; s.y += g1; s.z += g2; s.w += g3;
        sx.x += g0 * (v.x - mu.x) * rs.x; sx.y += g1 * (v.y - mu.y) * rs.y;
        sx.z += g2 * (v.z - mu.z) * rs.z; sx.w += g3 * (v.w - mu.w) * rs.w;
      }
    }
    __syncthreads();
    s_red[0][threadIdx.x] = s;
    s_red[1][threadIdx.x] = sx;
    __syncthreads();
    if (r0 == 0 && has_q) {
      for (int r = 1; r < rows; ++r) {
        const float4 a = s_red[0][r * lanes + q0], b = s_red[1][r * lanes + q0];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        sx.x += b.x; sx.y += b.y; sx.z += b.z; sx.w += b.w;
      }
      double* o = partial + ((size_t)blockIdx.x * C + 4 * q) * 2;
      o[0] = s.x; o[1] = sx.x; o[2] = s.y; o[3] = sx.y; o[4] = s.z; o[5] = sx.z; o[6] = s.w; o[7] = sx.w;
    }
  }
}

// dz = gscale[c] * (g - m1[c] - zhat * m2[c]),  gscale = gamma * rstd, m1 = sum(g) / n, m2 = sum(g zhat) / n
template <int ACT>
__global__ __launch_bounds__(kThreads) void bn_act_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ z,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ rowscale, const float* __restrict__ gscale,
                                                                   const float* __restrict__ m1, const float* __restrict__ m2,
                                                                   float* __restrict__ dz, size_t nquad, int Q) {
  const size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x, stride = (size_t)gridDim.x * kThreads;
  QuadWalk w(i0, stride, Q);
  for (size_t i = i0; i < nquad; i += stride, w.next()) {
    const int q = (int)w.q;
    const float4 g4 = reinterpret_cast<const float4*>(gy)[i];
    const float4 v = reinterpret_cast<const float4*>(z)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], rs = reinterpret_cast<const float4*>(rstd)[q];
    const float4 gs = reinterpret_cast<const float4*>(gscale)[q], a1 = reinterpret_cast<const float4*>(m1)[q],
                 a2 = reinterpret_cast<const float4*>(m2)[q];
    const float r = rowscale ? rowscale[w.row] : 1.0f;
    float4 o;
    o.x = gs.x * (g4.x * r * act_grad(ACT, v.x * sc.x + sh.x) - a1.x - (v.x - mu.x) * rs.x * a2.x);
    o.y = gs.y * (g4.y * r * act_grad(ACT, v.y * sc.y + sh.y) - a1.y - (v.y - mu.y) * rs.y * a2.y);
    o.z = gs.z * (g4.z * r * act_grad(ACT, v.z * sc.z + sh.z) - a1.z - (v.z - mu.z) * rs.z * a2.z);
    o.w = gs.w * (g4.w * r * act_grad(ACT, v.w * sc.w + sh.w) - a1.w - (v.w - mu.w) * rs.w * a2.w);
    reinterpret_cast<float4*>(dz)[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// filter gradient of the stride-2 convolution
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kWgIter = 16;       // output pixels per loop trip of a wave (four k-slots x four matrix instructions)

// the output pixel's (image, row, column) kept by increments: one division per wave start, carries afterwards
struct PixPos {
  int b, oy, ox;
};
__device__ __forceinline__ PixPos pix_pos(long p, int OH, int OW) {
  PixPos r;                           // 32-bit divisions: p < 2^29 (dz is at most 2^31 bytes)
  const unsigned q = (unsigned)p, row = q / (unsigned)OW;
  r.ox = (int)(q - row * (unsigned)OW);
  r.b = (int)(row / (unsigned)OH);
  r.oy = (int)(row - (unsigned)r.b * (unsigned)OH);
  return r;
}
// branch-free (a loop with a data-dependent trip count between two groups of matrix instructions splits the block they live in, and
// the accumulators carried across the split get copied between the register files): quotients by OW / OH through their 32-bit
// reciprocals m = floor(2^32 / d) + 1, exact for operands below 2^16
struct PixStep {
  unsigned m_ow, m_oh, one_ow, one_oh;     // one_*: all ones when the divisor is 1 (its reciprocal does not fit 32 bits: quotient = operand)
  int OH, OW;
};
__device__ __forceinline__ PixStep pix_step(int OH, int OW) {
  return PixStep{0xffffffffu / (unsigned)OW + 1u, 0xffffffffu / (unsigned)OH + 1u, 0u - (unsigned)(OW == 1), 0u - (unsigned)(OH == 1), OH, OW};
}
__device__ __forceinline__ void pix_advance(PixPos& r, int step, const PixStep& g) {
  const unsigned x = (unsigned)(r.ox + step);
  const unsigned cx = (__umulhi(x, g.m_ow) & ~g.one_ow) | (x & g.one_ow);
  r.ox = (int)(x - cx * (unsigned)g.OW);
  const unsigned y = (unsigned)r.oy + cx;
  const unsigned cy = (__umulhi(y, g.m_oh) & ~g.one_oh) | (y & g.one_oh);
  r.oy = (int)(y - cy * (unsigned)g.OH);
  r.b += (int)cy;
}

// NHWC input, Cin = 16 CH.  x:(B,H,W,Cin), dz:(B,OH,OW,Cout).  grid (pixel runs, K / ROWS filter-row groups, Cout / 16 channel
// chunks): a wave owns ROWS filter rows (all K where the K * K * CH accumulator tiles fit the register file: dz is then read once).
// ws:(runs, K, K, Cout, Cin) partial filter gradients.
template <int K, int CH, int PAD, int ROWS>
__global__ __launch_bounds__(kThreads) void conv_s2_wgrad_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                     float* __restrict__ ws, int Cout, int H, int W, int OH, int OW, long P,
                                                                     int pix_per_wave, unsigned x_bytes, unsigned dz_bytes) {
  constexpr int Cin = 16 * CH;
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const long run = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int u0 = blockIdx.y * ROWS, n = blockIdx.z;
  if (run * pix_per_wave >= P) return;
  // 32-bit pixel indices (P < 2^29: dz is at most 2^31 bytes): with 64-bit ones the compiler put the address arithmetic of a dead
  // slot under a branch, and accumulators carried across that branch were copied between the register files around every matrix
  // instruction (148 + 148 v_accvgpr copies per 100 matrix instructions)
  const int p_begin = (int)(run * pix_per_wave);
  const int p_end = (int)min(P, (long)p_begin + pix_per_wave);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, dz_bytes, 0x00020000);
  f32x4 acc[ROWS][K][CH];
#pragma unroll
  for (int ur = 0; ur < ROWS; ++ur)
#pragma unroll
    for (int v = 0; v < K; ++v)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[ur][v][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // slot s of this lane = pixel p + 4 s + kq of the trip starting at p
  PixPos pos[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) pos[s] = pix_pos(p_begin + 4 * s + kq, OH, OW);
  const PixStep pstep = pix_step(OH, OW);
#pragma unroll 1
  for (int p = p_begin; p < p_end; p += kWgIter) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ps = p + 4 * s + kq;
      const bool live = ps < p_end;
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, off_or_oob(live, (unsigned)(ps * Cout + 16 * n + j) * 4u), 0, 0));
      const int iy0 = 2 * pos[s].oy - PAD + u0, ix0 = 2 * pos[s].ox - PAD;
      const unsigned base0 = (unsigned)(((pos[s].b * H + iy0) * W + ix0) * Cin + j) * 4u;
#pragma unroll
      for (int ur = 0; ur < ROWS; ++ur) {
        const bool row_ok = live && (PAD == 0 || (iy0 + ur >= 0 && iy0 + ur < H));
        float b[K][CH];
#pragma unroll
        for (int v = 0; v < K; ++v) {
          const bool ok = row_ok && (PAD == 0 || (ix0 + v >= 0 && ix0 + v < W));
#pragma unroll
          for (int c = 0; c < CH; ++c)
            b[v][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off_or_oob(ok, base0 + (unsigned)((ur * W + v) * Cin + 16 * c) * 4u), 0, 0));
        }
#pragma unroll
        for (int v = 0; v < K; ++v)
#pragma unroll
          for (int c = 0; c < CH; ++c) acc[ur][v][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[v][c], acc[ur][v][c], 0, 0, 0);
      }
      pix_advance(pos[s], kWgIter, pstep);
    }
  }
  // D[i = 4 kq + r][j]: co = 16 n + 4 kq + r, ci = 16 c + j
#pragma unroll
  for (int ur = 0; ur < ROWS; ++ur) {
    float* o = ws + (((size_t)run * K + u0 + ur) * K) * (size_t)(Cout * Cin);
#pragma unroll
    for (int v = 0; v < K; ++v)
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(size_t)v * (Cout * Cin) + (16 * n + 4 * kq + r) * Cin + 16 * c + j] = acc[ur][v][c][r];
  }
}

// The same for Cin = 16 with the taps fetched four at a time: lane (j, kq) asks for the 16 bytes x[pixel kq][row][tap j / 4]
// [channels 4 (j % 4) .. + 3] -- the 16 lanes of a pixel cover four consecutive taps (256 contiguous bytes) -- and matrix instruction e
// of the group uses element e of every lane: D_e[co][j] = dW[co][ci = 4 (j % 4) + e][v = v0 + j / 4].  Four taps cost ONE wave load
// instead of four (the kernel is bound by the number of wave loads the vector L1 takes, 26 per 25 matrix instructions before);
// the accumulators just hold the filter in a permuted order, undone by the store.  K = 5: taps 0..3 by one such load, tap 4 by a
// dword load (j = channel, as before); K = 7: taps 0..3 and 3..6 by two loads, the second with its first tap masked (lanes j < 4).
// Launched with ROWS = 1 (a wave per filter row): with all five rows in one wave the register allocator rotates the 25 accumulator
// tiles through the register files every trip (391 v_accvgpr copies) and the kernel is slower than the dword form (132 vs 92 us);
// a row per wave runs the second layer as fast as that form and the third in 46 us instead of 66.
template <int K, int PAD, int ROWS>
__global__ __launch_bounds__(kThreads) void conv_s2_wgrad_nhwc16_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                       float* __restrict__ ws, int Cout, int H, int W, int OH, int OW,
                                                                       long P, int pix_per_wave, unsigned x_bytes, unsigned dz_bytes) {
  static_assert(K == 5 || K == 7, "taps four at a time: 5 = 4 + 1, 7 = 4 + (1 masked +) 3");
  constexpr int Cin = 16;
  constexpr int NQ = K == 5 ? 1 : 2;           // 16-byte tap groups per filter row
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int jt = j >> 2, jc = j & 3;           // this lane's tap inside a group, its channel quad
  const long run = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int u0 = blockIdx.y * ROWS, n = blockIdx.z;
  if (run * pix_per_wave >= P) return;
  const int p_begin = (int)(run * pix_per_wave);
  const int p_end = (int)min(P, (long)p_begin + pix_per_wave);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, dz_bytes, 0x00020000);
  f32x4 accq[ROWS][NQ][4];       // [row][tap group][e]
  f32x4 acc1[ROWS];              // K = 5: tap 4
#pragma unroll
  for (int ur = 0; ur < ROWS; ++ur) {
    acc1[ur] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NQ; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) accq[ur][g][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  PixPos pos[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) pos[s] = pix_pos(p_begin + 4 * s + kq, OH, OW);
  const PixStep pstep = pix_step(OH, OW);
#pragma unroll 1
  for (int p = p_begin; p < p_end; p += kWgIter) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ps = p + 4 * s + kq;
      const bool live = ps < p_end;
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, off_or_oob(live, (unsigned)(ps * Cout + 16 * n + j) * 4u), 0, 0));
      const int iy0 = 2 * pos[s].oy - PAD + u0, ix0 = 2 * pos[s].ox - PAD;
      const unsigned pix0 = (unsigned)((pos[s].b * H + iy0) * W + ix0);       // pixel index of tap (row u0, column 0)
#pragma unroll
      for (int ur = 0; ur < ROWS; ++ur) {
        const bool row_ok = live && (PAD == 0 || (iy0 + ur >= 0 && iy0 + ur < H));
        f32x4 q[NQ];
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
          const int v = 3 * g + jt;                                            // group 0: taps 0..3; group 1 (K = 7): taps 3..6
          const bool ok = row_ok && (g == 0 || jt != 0) && (PAD == 0 || (ix0 + v >= 0 && ix0 + v < W));
          q[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               xr, off_or_oob(ok, ((pix0 + (unsigned)(ur * W + v)) * Cin + 4 * jc) * 4u), 0, 0));
        }
        float b4 = 0.f;
        if (K == 5) {
          const bool ok = row_ok && (PAD == 0 || (ix0 + 4 >= 0 && ix0 + 4 < W));
          b4 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off_or_oob(ok, ((pix0 + (unsigned)(ur * W + 4)) * Cin + j) * 4u), 0, 0));
        }
#pragma unroll
        for (int g = 0; g < NQ; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) accq[ur][g][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, q[g][e], accq[ur][g][e], 0, 0, 0);
        if (K == 5) acc1[ur] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b4, acc1[ur], 0, 0, 0);
      }
      pix_advance(pos[s], kWgIter, pstep);
    }
  }
  // D_e[i = 4 kq + r][j]: co = 16 n + 4 kq + r, ci = 4 jc + e, v = 3 g + jt (group 1's jt = 0 column is zero: tap 3 belongs to group 0)
#pragma unroll
  for (int ur = 0; ur < ROWS; ++ur) {
    float* o = ws + (((size_t)run * K + u0 + ur) * K) * (size_t)(Cout * Cin);
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
      const int v = 3 * g + jt;
      if (g == 1 && jt == 0) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)     // the lane's four channels of (tap v, output channel 16 n + 4 kq + r): one 16-byte store
        *reinterpret_cast<f32x4*>(o + (size_t)v * (Cout * Cin) + (16 * n + 4 * kq + r) * Cin + 4 * jc) =
            f32x4{accq[ur][g][0][r], accq[ur][g][1][r], accq[ur][g][2][r], accq[ur][g][3][r]};
    }
    if (K == 5) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(size_t)4 * (Cout * Cin) + (16 * n + 4 * kq + r) * Cin + j] = acc1[ur][r];
    }
  }
}

// planar input with Cin <= 4 (the first layer: NCHW views).  The 16 columns of a B tile are the (channel, filter row) pairs
// jj = ci * K + u of one filter COLUMN v (Cin * K <= 16 NJ); a wave owns all K filter columns, so dz -- the large operand of this layer
// -- is read once.  x:(B,Cin,H,W), dz:(B,OH,OW,Cout).  ws:(runs, K [v], Cout, 16 NJ [jj]).  grid (pixel runs, 1, Cout / 16).
template <int K, int NJ, int PAD>
__global__ __launch_bounds__(kThreads) void conv_s2_wgrad_planar_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                       float* __restrict__ ws, int Cin, int Cout, int H, int W, int OH,
                                                                       int OW, long P, int pix_per_wave, unsigned x_bytes,
                                                                       unsigned dz_bytes) {
  const int lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const long run = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  if (run * pix_per_wave >= P) return;
  const int p_begin = (int)(run * pix_per_wave);          // 32-bit indices: see conv_s2_wgrad_nhwc_kernel
  const int p_end = (int)min(P, (long)p_begin + pix_per_wave);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, dz_bytes, 0x00020000);
  // this lane's columns: jj = 16 t + j -> (ci, u): a lane owns one filter ROW of one channel, so the K taps it needs for a pixel are
  // K consecutive floats of the plane -- one 16-byte load (+ one 4-byte / a second 16-byte load) instead of K dword gathers per
  // pixel group (the kernel is bound by the number of wave loads: 6 per 5 matrix instructions before, 3 now)
  int lci[NJ], lu[NJ];
  bool lhas[NJ];
#pragma unroll
  for (int t = 0; t < NJ; ++t) {
    const int jj = 16 * t + j;
    lhas[t] = jj < Cin * K;
    lci[t] = lhas[t] ? jj / K : 0;
    lu[t] = lhas[t] ? jj - lci[t] * K : 0;
  }
  f32x4 acc[K][NJ];       // [v][t]
#pragma unroll
  for (int v = 0; v < K; ++v)
#pragma unroll
    for (int t = 0; t < NJ; ++t) acc[v][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  PixPos pos[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) pos[s] = pix_pos(p_begin + 4 * s + kq, OH, OW);
  const PixStep pstep = pix_step(OH, OW);
#pragma unroll 1
  for (int p = p_begin; p < p_end; p += kWgIter) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ps = p + 4 * s + kq;
      const bool live = ps < p_end;
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, off_or_oob(live, (unsigned)(ps * Cout + 16 * n + j) * 4u), 0, 0));
      const int iy0 = 2 * pos[s].oy - PAD, ix0 = 2 * pos[s].ox - PAD;
      float b[K][NJ];
#pragma unroll
      for (int t = 0; t < NJ; ++t) {
        const int iy = iy0 + lu[t];
        const bool row_ok = live && lhas[t] && (PAD == 0 || (iy >= 0 && iy < H));
        const unsigned off0 = (unsigned)(((pos[s].b * Cin + lci[t]) * H + iy) * W + ix0) * 4u;
        if (PAD == 0 && K >= 4) {
          const unsigned o = off_or_oob(row_ok, off0);
          const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, o, 0, 0));
#pragma unroll
          for (int v = 0; v < 4; ++v) b[v][t] = q[v];
          if (K == 5) {
            b[K == 5 ? 4 : 0][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o + 16, 0, 0));   // (index: no out-of-range subscript in the K = 3 instantiation's dead branch)
          } else if (K == 7) {       // taps 3 .. 6: a second 16-byte load ending exactly at the row's last tap
            const f32x4 q2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, o + 12, 0, 0));
#pragma unroll
            for (int v = 4; v < 7; ++v) b[v][t] = q2[v - 3];
          }
        } else {
#pragma unroll
          for (int v = 0; v < K; ++v) {
            const bool ok = row_ok && (PAD == 0 || (ix0 + v >= 0 && ix0 + v < W));
            b[v][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off_or_oob(ok, off0 + (unsigned)v * 4u), 0, 0));
          }
        }
      }
      pix_advance(pos[s], kWgIter, pstep);
#pragma unroll
      for (int v = 0; v < K; ++v)
#pragma unroll
        for (int t = 0; t < NJ; ++t) acc[v][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[v][t], acc[v][t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int v = 0; v < K; ++v) {
    float* o = ws + ((size_t)run * K + v) * (size_t)(Cout * 16 * NJ);
#pragma unroll
    for (int t = 0; t < NJ; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(16 * n + 4 * kq + r) * (16 * NJ) + 16 * t + j] = acc[v][t][r];
  }
}

// dw[co][ci][u][v] = sum over the runs of the partials (fp32 partials of <= pix_per_wave pixels each, fp64 across), in two launches so
// that the whole chip reads the workspace (one launch over per_run / 64 blocks is 20 blocks at the first layer: 53 us of latency):
//   partial: block (64 consecutive elements of the per-run block, chunk of kRedChunk runs); its four waves take every fourth run of
//            the chunk, the four slices are combined in order -> part2 (chunks, per_run) fp64
//   final:   a thread per element adds the chunks in order and writes dw in the framework's order.
// The result does not depend on the launch.   nhwc: ws (runs, K, K, Cout, Cin);   planar: ws (runs, K [v], Cout, JJ) with jj = ci * K + u
constexpr int kRedChunk = 128;
__global__ __launch_bounds__(kThreads) void conv_s2_wgrad_partial_kernel(const float* __restrict__ ws, double* __restrict__ part2, int runs,
                                                                        size_t per_run) {
  __shared__ double s_acc[4][64];
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + el;   // element of ws's per-run block (coalesced reads)
  const int r0 = (int)blockIdx.y * kRedChunk, r1 = min(runs, r0 + kRedChunk);
  double acc = 0.0;
  if (e < per_run) {
    int r = r0 + slice;
    for (; r + 12 < r1; r += 16) {     // four independent loads in flight
      const float a0 = ws[(size_t)r * per_run + e], a1 = ws[(size_t)(r + 4) * per_run + e], a2 = ws[(size_t)(r + 8) * per_run + e],
                  a3 = ws[(size_t)(r + 12) * per_run + e];
      acc += (double)a0;
      acc += (double)a1;
      acc += (double)a2;
      acc += (double)a3;
    }
    for (; r < r1; r += 4) acc += (double)ws[(size_t)r * per_run + e];
  }
  s_acc[slice][el] = acc;
  __syncthreads();
  if (slice != 0 || e >= per_run) return;
  part2[(size_t)blockIdx.y * per_run + e] = ((s_acc[0][el] + s_acc[1][el]) + s_acc[2][el]) + s_acc[3][el];
}

__global__ __launch_bounds__(kThreads) void conv_s2_wgrad_reduce_kernel(const double* __restrict__ part2, float* __restrict__ dw, int chunks,
                                                                       int K, int Cout, int Cin, int JJ) {
  const int n_out = Cout * Cin * K * K;
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  const size_t per_run = JJ ? (size_t)K * Cout * JJ : (size_t)n_out;
  if (e >= per_run) return;
  double acc = 0.0;
  for (int c = 0; c < chunks; ++c) acc += part2[(size_t)c * per_run + e];
  int co, ci, u, v;
  if (JJ) {
    const int jj = (int)(e % JJ);
    co = (int)((e / JJ) % Cout);
    v = (int)(e / ((size_t)JJ * Cout));
    if (jj >= Cin * K) return;
    ci = jj / K;
    u = jj - ci * K;
  } else {
    ci = (int)(e % Cin);
    co = (int)((e / Cin) % Cout);
    v = (int)((e / ((size_t)Cin * Cout)) % K);
    u = (int)(e / ((size_t)Cin * Cout * K));
  }
  dw[((co * Cin + ci) * K + u) * K + v] = (float)acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// data gradient of the stride-2 convolution, NHWC: dz:(B,OH,OW,Cout = 16 CC) -> dx:(B,H,W,Cin = 16 NCI)
// One parity class (PY, PX) of input pixels: iy = PY + 2 yh, ix = PX + 2 xh.  Taps of the class: u = U0 + 2 a, v = V0 + 2 c with
// U0 = (PY + PAD) & 1; the output pixel is oy = yh + (PY + PAD - U0) / 2 - a.  wd:(K*K taps, CC, NCI, 4 [kq], 16 [j = ci], 4 [s])
// holds W[co = 16 cc + 4 kq + s][ci = 16 n + j][u][v] (ops.pack_conv_s2_dgrad_weights).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kDgTiles = 4;

template <int K, int CC, int NCI, int PAD, int PY, int PX>
__device__ __forceinline__ void conv_s2_dgrad_class(const float* __restrict__ dz, const float* __restrict__ wd, float* __restrict__ dx, int B,
                                                    int H, int W, int OH, int OW, unsigned dz_bytes, long wave) {
  constexpr int Cout = 16 * CC, Cin = 16 * NCI;
  constexpr int U0 = (PY + PAD) & 1, V0 = (PX + PAD) & 1;
  constexpr int TA = (K - U0 + 1) / 2, TC = (K - V0 + 1) / 2;       // taps per axis in this class
  constexpr int OY0 = (PY + PAD - U0) / 2, OX0 = (PX + PAD - V0) / 2;
  const int Hc = (H - PY + 1) / 2, Wc = (W - PX + 1) / 2;            // class pixels per column / row
  if (Hc <= 0 || Wc <= 0) return;
  // 32-bit index arithmetic (the host checks B * H * W < 2^31): a 64-bit division is ~100 instructions, and the first version spent
  // 60 of them per wave (three per tile up front, three per stored row) next to 64-144 matrix instructions
  const unsigned Q = (unsigned)B * (unsigned)Hc * (unsigned)Wc;
  const unsigned q0 = (unsigned)wave * (16 * kDgTiles);
  if (q0 >= Q) return;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, dz_bytes, 0x00020000);
  int yh[kDgTiles], xh[kDgTiles], bb[kDgTiles];
  int opix[kDgTiles];                                                 // this lane's pixel of the tile: its index in dx (pixels)
#pragma unroll
  for (int t = 0; t < kDgTiles; ++t) {
    const unsigned q = min(q0 + 16 * t + i, Q - 1);
    const unsigned row = q / (unsigned)Wc;
    xh[t] = (int)(q - row * (unsigned)Wc);
    bb[t] = (int)(row / (unsigned)Hc);
    yh[t] = (int)(row - (unsigned)bb[t] * (unsigned)Hc);
    opix[t] = (bb[t] * H + PY + 2 * yh[t]) * W + PX + 2 * xh[t];
  }
  f32x4 acc[kDgTiles][NCI];
#pragma unroll
  for (int t = 0; t < kDgTiles; ++t)
#pragma unroll
    for (int n = 0; n < NCI; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* wl = reinterpret_cast<const f32x4*>(wd) + lane;
#pragma unroll 1
  for (int a = 0; a < TA; ++a) {
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const int tap = (U0 + 2 * a) * K + V0 + 2 * c;
#pragma unroll
      for (int cc = 0; cc < CC; ++cc) {
        f32x4 av[kDgTiles], bv[NCI];
#pragma unroll
        for (int t = 0; t < kDgTiles; ++t) {
          const int oy = yh[t] + OY0 - a, ox = xh[t] + OX0 - c;
          const bool ok = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
          const unsigned off = (unsigned)(((bb[t] * OH + oy) * OW + ox) * Cout + 16 * cc + 4 * kq) * 4u;
          av[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, ok ? off : kOob, 0, 0));
        }
#pragma unroll
        for (int n = 0; n < NCI; ++n) bv[n] = wl[((tap * CC + cc) * NCI + n) * 64];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < kDgTiles; ++t)
#pragma unroll
            for (int n = 0; n < NCI; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][s], bv[n][s], acc[t][n], 0, 0, 0);
      }
    }
  }
  // D[i = 4 rq + r][j]: pixel q0 + 16 t + 4 rq + r, channel 16 n + j; that pixel's place in dx is held by lane 4 rq + r
  const int jj = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int t = 0; t < kDgTiles; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pix = __builtin_amdgcn_ds_bpermute((4 * rq + r) * 4, opix[t]);
      if (q0 + 16 * t + 4 * rq + r >= Q) continue;
      float* o = dx + (size_t)pix * Cin;
#pragma unroll
      for (int n = 0; n < NCI; ++n) o[16 * n + jj] = acc[t][n][r];
    }
}

template <int K, int CC, int NCI, int PAD>
__global__ __launch_bounds__(kThreads) void conv_s2_dgrad_nhwc_kernel(const float* __restrict__ dz, const float* __restrict__ wd,
                                                                     float* __restrict__ dx, int B, int H, int W, int OH, int OW,
                                                                     unsigned dz_bytes) {
  const long wave = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  switch (blockIdx.y) {   // block-uniform
    case 0: conv_s2_dgrad_class<K, CC, NCI, PAD, 0, 0>(dz, wd, dx, B, H, W, OH, OW, dz_bytes, wave); break;
    case 1: conv_s2_dgrad_class<K, CC, NCI, PAD, 0, 1>(dz, wd, dx, B, H, W, OH, OW, dz_bytes, wave); break;
    case 2: conv_s2_dgrad_class<K, CC, NCI, PAD, 1, 0>(dz, wd, dx, B, H, W, OH, OW, dz_bytes, wave); break;
    default: conv_s2_dgrad_class<K, CC, NCI, PAD, 1, 1>(dz, wd, dx, B, H, W, OH, OW, dz_bytes, wave); break;
  }
}

int wgrad_pix_per_wave(long P) {
  // ~2048 runs of >= 256 pixels (a multiple of the trip): the chip wants several waves per SIMD, but every run costs one partial
  // filter in the workspace and one term of the reduction (swept with the round-5 kernels, tools/kbench_convnet.py: 1024 / 2048 /
  // 4096 / 8192 runs = 168 / 122 / 126 / 138 us at the first layer, 93 / 83 / 92 / 115 at the second)
#ifndef EQA_WG_RUNS
#define EQA_WG_RUNS 2048
#endif
#ifndef EQA_WG_MINPIX
#define EQA_WG_MINPIX 256
#endif
  long per = (P + EQA_WG_RUNS - 1) / EQA_WG_RUNS;
  per = std::max((long)EQA_WG_MINPIX, std::min(16384L, per));
  return (int)((per + kWgIter - 1) / kWgIter * kWgIter);
}

}  // namespace

extern "C" {

int eqa_bn_act_fwd(const float* z, const float* scale, const float* shift, const float* rowscale, float* y, int64_t npix, int C, int act,
                   void* stream) {
  if (npix < 0 || C <= 0 || (act != 0 && act != 1)) return EQA_ERR_INVALID_ARG;
  if (npix == 0) return EQA_OK;
  if (!z || !scale || !shift || !y) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || (((uintptr_t)z | (uintptr_t)y | (uintptr_t)scale | (uintptr_t)shift) & 15)) return EQA_ERR_UNSUPPORTED;
  const size_t nquad = (size_t)npix * (C >> 2);
  const unsigned blocks = (unsigned)std::min<size_t>((nquad + kThreads - 1) / kThreads, 256 * 16);
  if (act == 0)
    hipLaunchKernelGGL(bn_act_fwd_kernel<0>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, z, scale, shift, rowscale, y, nquad, C >> 2);
  else
    hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, z, scale, shift, rowscale, y, nquad, C >> 2);
  return launch_status();
}

// pixels per block of the reductions: ~2048 blocks on the chip -- 8 pixels per block for the head's (2048, 1152) matrix, 1024 for the
// first layer's 1.8 M pixels (fewer partials for the finalize kernels to sum)
static int bn_act_pix_per_block(int64_t npix) {
  int64_t per = (npix + 2047) / 2048;
  per = std::max<int64_t>(8, std::min<int64_t>(1024, per));
  return (int)((per + 7) / 8 * 8);
}

int64_t eqa_bn_act_partial_blocks(int64_t npix) {
  if (npix <= 0) return 0;
  const int per = bn_act_pix_per_block(npix);
  return (npix + per - 1) / per;
}

int eqa_bn_act_stats(const float* z, double* partial, int64_t npix, int C, void* stream) {
  if (npix < 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (npix == 0) return EQA_OK;
  if (!z || !partial) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || ((uintptr_t)z & 15)) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(bn_act_stats_kernel, dim3((unsigned)eqa_bn_act_partial_blocks(npix)), dim3(kThreads), 0, (hipStream_t)stream, z, partial,
                     (size_t)npix, C, bn_act_pix_per_block(npix));
  return launch_status();
}

int eqa_bn_act_finalize(const double* partial, int64_t npix, int C, const float* gamma, const float* beta, double eps, double momentum,
                        float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd, void* stream) {
  if (npix <= 0 || C <= 0 || !partial || !gamma || !beta || !scale || !shift || !mean || !rstd || (!running_mean) != (!running_var))
    return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_act_finalize_kernel, dim3((C + kBnFinCh - 1) / kBnFinCh), dim3(kThreads), 0, (hipStream_t)stream, partial,
                     (int)eqa_bn_act_partial_blocks(npix), C, (double)npix, gamma, beta, eps, momentum, running_mean, running_var, scale, shift,
                     mean, rstd);
  return launch_status();
}

int eqa_bn_act_bwd_finalize(const double* partial, int64_t npix, int C, const float* gamma, const float* rstd, float* dgamma, float* dbeta,
                            float* gscale, float* m1, float* m2, void* stream) {
  if (npix <= 0 || C <= 0 || !partial || !gamma || !rstd || !dgamma || !dbeta || !gscale || !m1 || !m2) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_act_bwd_finalize_kernel, dim3((C + kBnFinCh - 1) / kBnFinCh), dim3(kThreads), 0, (hipStream_t)stream, partial,
                     (int)eqa_bn_act_partial_blocks(npix), C, (double)npix, gamma, rstd, dgamma, dbeta, gscale, m1, m2);
  return launch_status();
}

int eqa_bn_act_bwd_reduce(const float* gy, const float* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                          const float* rowscale, double* partial, int64_t npix, int C, int act, void* stream) {
  if (npix < 0 || C <= 0 || (act != 0 && act != 1)) return EQA_ERR_INVALID_ARG;
  if (npix == 0) return EQA_OK;
  if (!gy || !z || !scale || !shift || !mean || !rstd || !partial) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || (((uintptr_t)gy | (uintptr_t)z | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd) & 15))
    return EQA_ERR_UNSUPPORTED;
  const unsigned blocks = (unsigned)eqa_bn_act_partial_blocks(npix);
  if (act == 0)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<0>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, gy, z, scale, shift, mean, rstd,
                       rowscale, partial, (size_t)npix, C, bn_act_pix_per_block(npix));
  else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<1>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, gy, z, scale, shift, mean, rstd,
                       rowscale, partial, (size_t)npix, C, bn_act_pix_per_block(npix));
  return launch_status();
}

int eqa_bn_act_bwd_apply(const float* gy, const float* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                         const float* rowscale, const float* gscale, const float* m1, const float* m2, float* dz, int64_t npix, int C,
                         int act, void* stream) {
  if (npix < 0 || C <= 0 || (act != 0 && act != 1)) return EQA_ERR_INVALID_ARG;
  if (npix == 0) return EQA_OK;
  if (!gy || !z || !scale || !shift || !mean || !rstd || !gscale || !m1 || !m2 || !dz) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || (((uintptr_t)gy | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd |
                   (uintptr_t)gscale | (uintptr_t)m1 | (uintptr_t)m2) & 15))
    return EQA_ERR_UNSUPPORTED;
  const size_t nquad = (size_t)npix * (C >> 2);
  const unsigned blocks = (unsigned)std::min<size_t>((nquad + kThreads - 1) / kThreads, 256 * 16);
  if (act == 0)
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<0>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, gy, z, scale, shift, mean, rstd,
                       rowscale, gscale, m1, m2, dz, nquad, C >> 2);
  else
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<1>, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, gy, z, scale, shift, mean, rstd,
                       rowscale, gscale, m1, m2, dz, nquad, C >> 2);
  return launch_status();
}

int eqa_conv_s2_wgrad_supported(int Cin, int Cout, int K, int pad, int planar) {
  if (!eqa_conv_s2_supported(Cin, Cout, K, pad, planar)) return 0;
  if (planar) return Cin * K <= 32;
  return 1;
}

int64_t eqa_conv_s2_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int K, int pad, int planar) {
  if (!eqa_conv_s2_wgrad_supported(Cin, Cout, K, pad, planar) || B <= 0 || H + 2 * pad < K || W + 2 * pad < K) return 0;
  const long OH = (H + 2 * pad - K) / 2 + 1, OW = (W + 2 * pad - K) / 2 + 1, P = (long)B * OH * OW;
  const long per = wgrad_pix_per_wave(P), runs = (P + per - 1) / per;
  const long per_run = planar ? (long)K * Cout * (Cin * K <= 16 ? 16 : 32) : (long)K * K * Cout * Cin;
  const long chunks = (runs + kRedChunk - 1) / kRedChunk;
  return ((runs * per_run * 4 + 15) & ~15L) + chunks * per_run * 8;      // the runs' fp32 partials, then the chunks' fp64 ones
}

int eqa_conv_s2_wgrad(const float* x, const float* dz, float* dw, void* workspace, int B, int Cin, int H, int W, int Cout, int K, int pad,
                      int planar, void* stream) {
  if (!dw || B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return EQA_ERR_INVALID_ARG;
  if (!eqa_conv_s2_wgrad_supported(Cin, Cout, K, pad, planar)) return EQA_ERR_UNSUPPORTED;
  if (H + 2 * pad < K || W + 2 * pad < K) return EQA_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  // an empty batch (its tensors may be null pointers): the filter gradient of nothing is zero
  if (B == 0) return hipMemsetAsync(dw, 0, (size_t)Cout * Cin * K * K * 4, st) == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
  if (!x || !dz || !workspace) return EQA_ERR_INVALID_ARG;
  const int OH = (H + 2 * pad - K) / 2 + 1, OW = (W + 2 * pad - K) / 2 + 1;
  const long P = (long)B * OH * OW;
  const size_t xbytes = (size_t)B * Cin * H * W * 4, gbytes = (size_t)P * Cout * 4;
  if (xbytes > 0x7fffffe0ULL || gbytes > 0x7fffffe0ULL || (((uintptr_t)x | (uintptr_t)dz | (uintptr_t)dw | (uintptr_t)workspace) & 15))
    return EQA_ERR_UNSUPPORTED;
  const int per = wgrad_pix_per_wave(P);
  const long runs = (P + per - 1) / per;
  const unsigned gx = (unsigned)((runs + kThreads / 64 - 1) / (kThreads / 64));
  float* ws = static_cast<float*>(workspace);
  const unsigned xb = (unsigned)xbytes, gb = (unsigned)gbytes;
  int jj = 0;
  if (planar) {
    const int nj = Cin * K <= 16 ? 1 : 2;
    jj = 16 * nj;
    const dim3 grid(gx, 1, Cout / 16);
#define EQA_WGP(KK, NJ, PD)                                                                                                        \
  hipLaunchKernelGGL((conv_s2_wgrad_planar_kernel<KK, NJ, PD>), grid, dim3(kThreads), 0, st, x, dz, ws, Cin, Cout, H, W, OH, OW, P, per, xb, gb)
#define EQA_WGP_K(KK)                                                                                                              \
  do {                                                                                                                             \
    if (nj == 1) { if (pad) EQA_WGP(KK, 1, 1); else EQA_WGP(KK, 1, 0); }                                                          \
    else { if (pad) EQA_WGP(KK, 2, 1); else EQA_WGP(KK, 2, 0); }                                                                  \
  } while (0)
    if (K == 7) EQA_WGP_K(7); else if (K == 5) EQA_WGP_K(5); else EQA_WGP_K(3);
#undef EQA_WGP_K
#undef EQA_WGP
  } else {
    const int ch = Cin / 16;
    // all K filter rows in one wave where K * K * CH accumulator tiles (4 registers each) leave room: K = 3 always, K = 5 at 16 channels
#define EQA_WGN(KK, CH, PD, RW)                                                                                                    \
  hipLaunchKernelGGL((conv_s2_wgrad_nhwc_kernel<KK, CH, PD, RW>), dim3(gx, KK / RW, Cout / 16), dim3(kThreads), 0, st, x, dz, ws, Cout, H, W, \
                     OH, OW, P, per, xb, gb)
#define EQA_WGN16(KK, PD, RW)                                                                                                      \
  hipLaunchKernelGGL((conv_s2_wgrad_nhwc16_kernel<KK, PD, RW>), dim3(gx, KK / RW, Cout / 16), dim3(kThreads), 0, st, x, dz, ws, Cout, H, W, OH, \
                     OW, P, per, xb, gb)
#define EQA_WGN_K(KK, R1, R2, R4)                                                                                                  \
  do {                                                                                                                             \
    if (ch == 1 && KK == 5) { if (pad) EQA_WGN16(5, 1, 1); else EQA_WGN16(5, 0, 1); }                                             \
    else if (ch == 1 && KK == 7) { if (pad) EQA_WGN16(7, 1, 1); else EQA_WGN16(7, 0, 1); }                                        \
    else if (ch == 1) { if (pad) EQA_WGN(KK, 1, 1, R1); else EQA_WGN(KK, 1, 0, R1); }                                             \
    else if (ch == 2) { if (pad) EQA_WGN(KK, 2, 1, R2); else EQA_WGN(KK, 2, 0, R2); }                                             \
    else { if (pad) EQA_WGN(KK, 4, 1, R4); else EQA_WGN(KK, 4, 0, R4); }                                                          \
  } while (0)
    if (K == 7) EQA_WGN_K(7, 1, 1, 1); else if (K == 5) EQA_WGN_K(5, 5, 1, 1); else EQA_WGN_K(3, 3, 3, 3);
#undef EQA_WGN_K
#undef EQA_WGN16
#undef EQA_WGN
  }
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  const size_t per_run = planar ? (size_t)K * Cout * jj : (size_t)K * K * Cout * Cin;
  const int chunks = (int)((runs + kRedChunk - 1) / kRedChunk);
  double* part2 = reinterpret_cast<double*>(static_cast<char*>(workspace) + (((size_t)runs * per_run * 4 + 15) & ~(size_t)15));
  hipLaunchKernelGGL(conv_s2_wgrad_partial_kernel, dim3((unsigned)((per_run + 63) / 64), (unsigned)chunks), dim3(kThreads), 0, st, ws, part2,
                     (int)runs, per_run);
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  hipLaunchKernelGGL(conv_s2_wgrad_reduce_kernel, dim3((unsigned)((per_run + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, part2, dw,
                     chunks, K, Cout, Cin, jj);
  return launch_status();
}

int eqa_conv_s2_dgrad_supported(int Cin, int Cout, int K, int pad) { return eqa_conv_s2_supported(Cin, Cout, K, pad, 0); }

int eqa_conv_s2_dgrad(const float* dz, const float* wd, float* dx, int B, int Cin, int H, int W, int Cout, int K, int pad, void* stream) {
  if (B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return EQA_ERR_INVALID_ARG;
  if (!eqa_conv_s2_dgrad_supported(Cin, Cout, K, pad)) return EQA_ERR_UNSUPPORTED;
  if (H + 2 * pad < K || W + 2 * pad < K) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if (!dz || !wd || !dx) return EQA_ERR_INVALID_ARG;
  const int OH = (H + 2 * pad - K) / 2 + 1, OW = (W + 2 * pad - K) / 2 + 1;
  const size_t gbytes = (size_t)B * OH * OW * Cout * 4, xbytes = (size_t)B * H * W * Cin * 4;
  if (gbytes > 0x7fffffe0ULL || xbytes > 0x7fffffffffULL || (size_t)B * H * W > 0x7fffffffULL ||
      (((uintptr_t)dz | (uintptr_t)wd | (uintptr_t)dx) & 15))
    return EQA_ERR_UNSUPPORTED;
  const long Qmax = (long)B * ((H + 1) / 2) * ((W + 1) / 2);        // the largest parity class
  const long waves = (Qmax + 16 * kDgTiles - 1) / (16 * kDgTiles);
  const dim3 grid((unsigned)((waves + kThreads / 64 - 1) / (kThreads / 64)), 4);
  hipStream_t st = (hipStream_t)stream;
  const unsigned gb = (unsigned)gbytes;
  const int cc = Cout / 16, nci = Cin / 16;
#define EQA_DG(KK, CC_, NCI_, PD) \
  hipLaunchKernelGGL((conv_s2_dgrad_nhwc_kernel<KK, CC_, NCI_, PD>), grid, dim3(kThreads), 0, st, dz, wd, dx, B, H, W, OH, OW, gb)
#define EQA_DG_K(KK)                                                                                                               \
  do {                                                                                                                             \
    if (nci == 1 && cc == 1) { if (pad) EQA_DG(KK, 1, 1, 1); else EQA_DG(KK, 1, 1, 0); }                                          \
    else if (nci == 1 && cc == 2) { if (pad) EQA_DG(KK, 2, 1, 1); else EQA_DG(KK, 2, 1, 0); }                                     \
    else if (nci == 2 && cc == 2) { if (pad) EQA_DG(KK, 2, 2, 1); else EQA_DG(KK, 2, 2, 0); }                                     \
    else if (nci == 2 && cc == 4) { if (pad) EQA_DG(KK, 4, 2, 1); else EQA_DG(KK, 4, 2, 0); }                                     \
    else { if (pad) EQA_DG(KK, 4, 4, 1); else EQA_DG(KK, 4, 4, 0); }                                                              \
  } while (0)
  if (K == 7) EQA_DG_K(7); else if (K == 5) EQA_DG_K(5); else EQA_DG_K(3);
#undef EQA_DG_K
#undef EQA_DG
  return launch_status();
}

}  // extern "C"
