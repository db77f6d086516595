// libeqa_hip.so, part 6 -- training-mode InnerBatchNorm + ReLU + PointwiseDropout of the canonicalization network
// (reference: escnn_networks.py:67-85 builds every hidden block as R2Conv -> InnerBatchNorm(momentum 0.9) -> ReLU ->
// PointwiseDropout(0.5)) on channels-last maps, forward and backward.  C ABI: include/eqa_hip.h.
//
// The op-by-op form costs ~20 full-tensor passes per block and direction (measured: 60 of the 114 ms of a training step of
// the headline net at B = 256); fused it is: statistics (1 read), apply (1 read, 1 write), backward reduce (3 reads),
// backward apply (3 reads, 1 write).  All four are HBM streams over (pixels, C) with float4 lanes along the channels.
#include "eqa_common.hpp"

namespace {

constexpr int kBnPixPerBlock = 256;  // pixels per block of the two reductions: (N / 256) x C x 2 fp64 partials

// (the dropout mask's counter-based hash, mix32 / dropout_keep_quad: eqa_common.hpp)

// partial[(blk * C + c) * 2 + {0, 1}] = sum, sum of squares over the block's pixels (fp32 inside a block of <= 256 pixels,
// fp64 across blocks by the caller: deterministic)
__global__ __launch_bounds__(kThreads) void bn_stats_nhwc_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                                size_t npix, int C) {
  __shared__ float4 s_red[2][kThreads];
  const int Q = C >> 2;                       // channel quads
  const int lanes = min(Q, kThreads);         // threads along the channels
  const int rows = kThreads / lanes;          // pixels handled concurrently
  const int q0 = threadIdx.x % lanes, r0 = threadIdx.x / lanes;
  const size_t p0 = (size_t)blockIdx.x * kBnPixPerBlock;
  const size_t p1 = min(npix, p0 + kBnPixPerBlock);
  for (int qb = 0; qb < Q; qb += lanes) {      // uniform trip count: every thread reaches both barriers of every trip
    const int q = qb + q0;
    const bool has_q = q < Q;                   // (C / 4 > 256 and not a multiple of 256: the last trip is partial)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), ss = s;
    if (r0 < rows && has_q) {
      for (size_t p = p0 + r0; p < p1; p += rows) {
        const float4 v = reinterpret_cast<const float4*>(x)[p * Q + q];
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
      o[0] = s.x; o[1] = ss.x; o[2] = s.y; o[3] = ss.y; o[4] = s.z; o[5] = ss.z; o[6] = s.w; o[7] = ss.w;
    }
  }
}

// y = dropout(relu(x * scale[c] + shift[c])): kept elements are multiplied by inv_keep, the others become 0
__global__ __launch_bounds__(kThreads) void bn_relu_dropout_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                       const float* __restrict__ shift, float* __restrict__ y,
                                                                       size_t nquad, int Q, uint32_t drop_threshold,
                                                                       float inv_keep, uint32_t seed) {
  const size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x, stride = (size_t)gridDim.x * kThreads;
  QuadWalk w(i0, stride, Q);
  for (size_t i = i0; i < nquad; i += stride, w.next()) {
    const int q = (int)w.q;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    float o[4] = {fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f), fmaxf(v.z * sc.z + sh.z, 0.f),
                  fmaxf(v.w * sc.w + sh.w, 0.f)};
    if (drop_threshold) {
      const uint32_t base = mix32((uint32_t)i * 0x9E3779B1u + seed) ^ (uint32_t)(i >> 32);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = mix32(base + (uint32_t)k * 0x632BE5ABu) >= drop_threshold ? o[k] * inv_keep : 0.f;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// "Kept and positive" of one channel quad, recomputed exactly as bn_relu_dropout_nhwc_kernel decided it (same expression for
// the pre-activation, same hash): lets the backward passes skip reading y (one 8.67 MB map per image less in each).
__device__ __forceinline__ void bn_live_quad(const float4& v, const float4& sc, const float4& sh, size_t i, uint32_t drop_threshold,
                                             uint32_t seed, bool (&live)[4]) {
  const float o[4] = {fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f), fmaxf(v.z * sc.z + sh.z, 0.f),
                      fmaxf(v.w * sc.w + sh.w, 0.f)};
  bool keep[4];
  dropout_keep_quad(i, drop_threshold, seed, keep);
#pragma unroll
  for (int k = 0; k < 4; ++k) live[k] = o[k] > 0.f && keep[k];
}

// The gradient of a hidden block whose output feeds only the window sums of the linearised last layer (pooling.hip) depends on a
// pixel only through the CLASS of its row and of its column (border index < nb, interior, bottom / right border): TABLE mode reads
// gy[b][y][x][c] = table[b][cls(y)][cls(x)][c] from the (B, 2nb+1, 2nb+1, C) table instead of an expanded (B, H, W, C) map that
// one kernel would write and two would read (2.2 GB each way at the headline shape).
struct WsGradGeom {
  int H, W, nb;   // map size, k - 1
};
__device__ __forceinline__ int ws_cls(int i, int n, int nb) { return i < nb ? i : (i >= n - nb ? i - (n - nb) + nb + 1 : nb); }
__device__ __forceinline__ size_t ws_table_quad(size_t p, int q, int Q, const WsGradGeom& g) {   // pixel index -> quad index in the table
  const unsigned hw = (unsigned)g.H * (unsigned)g.W, pp = (unsigned)p;      // B * H * W < 2^31: checked by the host; 32-bit divisions
  const unsigned b = pp / hw, r = pp - b * hw;
  const unsigned y = r / (unsigned)g.W, x = r - y * (unsigned)g.W;
  const unsigned T = 2 * g.nb + 1;
  return ((size_t)(b * T + ws_cls((int)y, g.H, g.nb)) * T + ws_cls((int)x, g.W, g.nb)) * Q + q;
}

// g = gy * (y > 0 ? inv_keep : 0)  (gradient at the batch-norm output);  partial sums of g and g * xhat per channel
template <bool RECOMPUTE, bool TABLE = false>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_nhwc_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                     const float* __restrict__ x, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, float inv_keep,
                                                                     double* __restrict__ partial, size_t npix, int C,
                                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                                     uint32_t drop_threshold, uint32_t seed, WsGradGeom geom) {
  __shared__ float4 s_red[2][kThreads];
  const int Q = C >> 2;
  const int lanes = min(Q, kThreads);
  const int rows = kThreads / lanes;
  const int q0 = threadIdx.x % lanes, r0 = threadIdx.x / lanes;
  const size_t p0 = (size_t)blockIdx.x * kBnPixPerBlock;
  const size_t p1 = min(npix, p0 + kBnPixPerBlock);
  for (int qb = 0; qb < Q; qb += lanes) {      // uniform trip count (see bn_stats_nhwc_kernel)
    const int q = qb + q0;
    const bool has_q = q < Q;
    const int qc = has_q ? q : 0;
    const float4 mu = reinterpret_cast<const float4*>(mean)[qc], rs = reinterpret_cast<const float4*>(rstd)[qc];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sx = s;
    if (r0 < rows && has_q) {
      for (size_t p = p0 + r0; p < p1; p += rows) {
        const size_t i = p * Q + q;
        const float4 g4 = reinterpret_cast<const float4*>(gy)[TABLE ? ws_table_quad(p, q, Q, geom) : i];
        const float4 x4 = reinterpret_cast<const float4*>(x)[i];
        bool live[4];
        if (RECOMPUTE) {
          bn_live_quad(x4, reinterpret_cast<const float4*>(scale)[q], reinterpret_cast<const float4*>(shift)[q], i, drop_threshold, seed,
                       live);
        } else {
          const float4 y4 = reinterpret_cast<const float4*>(y)[i];
          live[0] = y4.x > 0.f; live[1] = y4.y > 0.f; live[2] = y4.z > 0.f; live[3] = y4.w > 0.f;
        }
        const float g0 = live[0] ? g4.x * inv_keep : 0.f, g1 = live[1] ? g4.y * inv_keep : 0.f;
        const float g2 = live[2] ? g4.z * inv_keep : 0.f, g3 = live[3] ? g4.w * inv_keep : 0.f;
        s.x += g0; s.y += g1; s.z += g2; s.w += g3;
        sx.x += g0 * (x4.x - mu.x) * rs.x; sx.y += g1 * (x4.y - mu.y) * rs.y;
        sx.z += g2 * (x4.z - mu.z) * rs.z; sx.w += g3 * (x4.w - mu.w) * rs.w;
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

// dx = a[c] * (g - b[c] - xhat * d[c]),  xhat = (x - mean[c]) * rstd[c],  g as above.
// (a = gamma * rstd, b = sum(g) / n, d = sum(g xhat) / n per FIELD, expanded to channels by the caller; with running
// statistics -- eval mode under autograd -- b = d = 0.)
template <bool RECOMPUTE, bool TABLE = false>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_nhwc_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                    const float* __restrict__ x, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ a,
                                                                    const float* __restrict__ b, const float* __restrict__ d,
                                                                    float inv_keep, float* __restrict__ dx, size_t nquad, int Q,
                                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                                    uint32_t drop_threshold, uint32_t seed, WsGradGeom geom) {
  const size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x, stride = (size_t)gridDim.x * kThreads;
  QuadWalk w(i0, stride, Q);
  for (size_t i = i0; i < nquad; i += stride, w.next()) {
    const int q = (int)w.q;
    const float4 g4 = reinterpret_cast<const float4*>(gy)[TABLE ? ws_table_quad(w.row, q, Q, geom) : i];
    const float4 x4 = reinterpret_cast<const float4*>(x)[i];
    bool live[4];
    if (RECOMPUTE) {
      bn_live_quad(x4, reinterpret_cast<const float4*>(scale)[q], reinterpret_cast<const float4*>(shift)[q], i, drop_threshold, seed, live);
    } else {
      const float4 y4 = reinterpret_cast<const float4*>(y)[i];
      live[0] = y4.x > 0.f; live[1] = y4.y > 0.f; live[2] = y4.z > 0.f; live[3] = y4.w > 0.f;
    }
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], rs = reinterpret_cast<const float4*>(rstd)[q];
    const float4 a4 = reinterpret_cast<const float4*>(a)[q], b4 = reinterpret_cast<const float4*>(b)[q];
    const float4 d4 = reinterpret_cast<const float4*>(d)[q];
    float4 o;
    o.x = a4.x * ((live[0] ? g4.x * inv_keep : 0.f) - b4.x - (x4.x - mu.x) * rs.x * d4.x);
    o.y = a4.y * ((live[1] ? g4.y * inv_keep : 0.f) - b4.y - (x4.y - mu.y) * rs.y * d4.y);
    o.z = a4.z * ((live[2] ? g4.z * inv_keep : 0.f) - b4.z - (x4.z - mu.z) * rs.z * d4.z);
    o.w = a4.w * ((live[3] ? g4.w * inv_keep : 0.f) - b4.w - (x4.w - mu.w) * rs.w * d4.w);
    reinterpret_cast<float4*>(dx)[i] = o;
  }
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline uint32_t drop_threshold(float drop_p) { return dropout_threshold(drop_p); }
inline unsigned stream_blocks(size_t nquad) { return (unsigned)std::min<size_t>((nquad + kThreads - 1) / kThreads, 256 * 32); }

}  // namespace

extern "C" {

int64_t eqa_bn_partial_blocks(int64_t n_pixels) { return n_pixels <= 0 ? 0 : (n_pixels + kBnPixPerBlock - 1) / kBnPixPerBlock; }

int eqa_bn_stats_nhwc(const float* x, double* partial, int64_t n_pixels, int C, void* stream) {
  if (n_pixels < 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (n_pixels == 0) return EQA_OK;
  if (!x || !partial) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || !aligned16(x)) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(bn_stats_nhwc_kernel, dim3((unsigned)eqa_bn_partial_blocks(n_pixels)), dim3(kThreads), 0, (hipStream_t)stream, x,
                     partial, (size_t)n_pixels, C);
  return launch_status();
}

int eqa_bn_relu_dropout_nhwc(const float* x, const float* scale, const float* shift, float* y, int64_t n_pixels, int C,
                             float drop_p, uint32_t seed, void* stream) {
  if (n_pixels < 0 || C <= 0 || !(drop_p >= 0.0f && drop_p < 1.0f)) return EQA_ERR_INVALID_ARG;
  if (n_pixels == 0) return EQA_OK;
  if (!x || !scale || !shift || !y) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || !aligned16(x) || !aligned16(y) || !aligned16(scale) || !aligned16(shift)) return EQA_ERR_UNSUPPORTED;
  const size_t nquad = (size_t)n_pixels * (C >> 2);
  const uint32_t thr = drop_threshold(drop_p);
  hipLaunchKernelGGL(bn_relu_dropout_nhwc_kernel, dim3(stream_blocks(nquad)), dim3(kThreads), 0, (hipStream_t)stream, x, scale, shift,
                     y, nquad, C >> 2, thr, 1.0f / (1.0f - drop_p), seed);
  return launch_status();
}

int eqa_bn_bwd_reduce_nhwc(const float* gy, const float* y, const float* x, const float* mean, const float* rstd, float drop_p,
                           double* partial, int64_t n_pixels, int C, const float* scale, const float* shift, uint32_t seed,
                           void* stream) {
  if (n_pixels < 0 || C <= 0 || !(drop_p >= 0.0f && drop_p < 1.0f)) return EQA_ERR_INVALID_ARG;
  if (n_pixels == 0) return EQA_OK;
  if (!gy || !x || !mean || !rstd || !partial || (!y && (!scale || !shift))) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || !aligned16(gy) || !aligned16(y) || !aligned16(x) || !aligned16(mean) || !aligned16(rstd) || !aligned16(scale) ||
      !aligned16(shift))
    return EQA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)eqa_bn_partial_blocks(n_pixels));
  const float inv_keep = 1.0f / (1.0f - drop_p);
  if (y)
    hipLaunchKernelGGL(bn_bwd_reduce_nhwc_kernel<false>, grid, dim3(kThreads), 0, (hipStream_t)stream, gy, y, x, mean, rstd, inv_keep,
                       partial, (size_t)n_pixels, C, scale, shift, 0u, seed, WsGradGeom{0, 0, 0});
  else
    hipLaunchKernelGGL(bn_bwd_reduce_nhwc_kernel<true>, grid, dim3(kThreads), 0, (hipStream_t)stream, gy, y, x, mean, rstd, inv_keep,
                       partial, (size_t)n_pixels, C, scale, shift, drop_threshold(drop_p), seed, WsGradGeom{0, 0, 0});
  return launch_status();
}

int eqa_bn_bwd_apply_nhwc(const float* gy, const float* y, const float* x, const float* mean, const float* rstd, const float* a,
                          const float* b, const float* d, float drop_p, float* dx, int64_t n_pixels, int C, const float* scale,
                          const float* shift, uint32_t seed, void* stream) {
  if (n_pixels < 0 || C <= 0 || !(drop_p >= 0.0f && drop_p < 1.0f)) return EQA_ERR_INVALID_ARG;
  if (n_pixels == 0) return EQA_OK;
  if (!gy || !x || !mean || !rstd || !a || !b || !d || !dx || (!y && (!scale || !shift))) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || !aligned16(gy) || !aligned16(y) || !aligned16(x) || !aligned16(dx) || !aligned16(mean) || !aligned16(rstd) ||
      !aligned16(a) || !aligned16(b) || !aligned16(d) || !aligned16(scale) || !aligned16(shift))
    return EQA_ERR_UNSUPPORTED;
  const size_t nquad = (size_t)n_pixels * (C >> 2);
  const float inv_keep = 1.0f / (1.0f - drop_p);
  if (y)
    hipLaunchKernelGGL(bn_bwd_apply_nhwc_kernel<false>, dim3(stream_blocks(nquad)), dim3(kThreads), 0, (hipStream_t)stream, gy, y, x, mean,
                       rstd, a, b, d, inv_keep, dx, nquad, C >> 2, scale, shift, 0u, seed, WsGradGeom{0, 0, 0});
  else
    hipLaunchKernelGGL(bn_bwd_apply_nhwc_kernel<true>, dim3(stream_blocks(nquad)), dim3(kThreads), 0, (hipStream_t)stream, gy, y, x, mean,
                       rstd, a, b, d, inv_keep, dx, nquad, C >> 2, scale, shift, drop_threshold(drop_p), seed, WsGradGeom{0, 0, 0});
  return launch_status();
}

// The two backward passes of a hidden block whose output feeds only the window sums (eqa_window_sums_nhwc_act): the upstream
// gradient is the (B, 2k-1, 2k-1, C) class table of the window sums' backward, read in place of an expanded (B, H, W, C) map;
// "kept and positive" is recomputed from x, scale, shift and the seed.  x:(B,H,W,C).
int eqa_bn_bwd_reduce_nhwc_wsgrad(const float* table, const float* x, const float* mean, const float* rstd, float drop_p,
                                  double* partial, int B, int H, int W, int C, int k, const float* scale, const float* shift,
                                  uint32_t seed, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || !(drop_p >= 0.0f && drop_p < 1.0f)) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if (!table || !x || !mean || !rstd || !partial || !scale || !shift) return EQA_ERR_INVALID_ARG;
  if (H < 2 * k - 1 || W < 2 * k - 1) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || !aligned16(table) || !aligned16(x) || !aligned16(mean) || !aligned16(rstd) || !aligned16(scale) || !aligned16(shift) ||
      (size_t)B * H * W > 0x7fffffffULL)
    return EQA_ERR_UNSUPPORTED;
  const int64_t npix = (int64_t)B * H * W;
  hipLaunchKernelGGL((bn_bwd_reduce_nhwc_kernel<true, true>), dim3((unsigned)eqa_bn_partial_blocks(npix)), dim3(kThreads), 0,
                     (hipStream_t)stream, table, (const float*)nullptr, x, mean, rstd, 1.0f / (1.0f - drop_p), partial, (size_t)npix, C, scale,
                     shift, drop_threshold(drop_p), seed, WsGradGeom{H, W, k - 1});
  return launch_status();
}

int eqa_bn_bwd_apply_nhwc_wsgrad(const float* table, const float* x, const float* mean, const float* rstd, const float* a,
                                 const float* b, const float* d, float drop_p, float* dx, int B, int H, int W, int C, int k,
                                 const float* scale, const float* shift, uint32_t seed, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || !(drop_p >= 0.0f && drop_p < 1.0f)) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if (!table || !x || !mean || !rstd || !a || !b || !d || !dx || !scale || !shift) return EQA_ERR_INVALID_ARG;
  if (H < 2 * k - 1 || W < 2 * k - 1) return EQA_ERR_INVALID_ARG;
  if ((C & 3) || !aligned16(table) || !aligned16(x) || !aligned16(dx) || !aligned16(mean) || !aligned16(rstd) || !aligned16(a) ||
      !aligned16(b) || !aligned16(d) || !aligned16(scale) || !aligned16(shift) || (size_t)B * H * W > 0x7fffffffULL)
    return EQA_ERR_UNSUPPORTED;
  const size_t nquad = (size_t)B * H * W * (C >> 2);
  hipLaunchKernelGGL((bn_bwd_apply_nhwc_kernel<true, true>), dim3(stream_blocks(nquad)), dim3(kThreads), 0, (hipStream_t)stream, table,
                     (const float*)nullptr, x, mean, rstd, a, b, d, 1.0f / (1.0f - drop_p), dx, nquad, C >> 2, scale, shift,
                     drop_threshold(drop_p), seed, WsGradGeom{H, W, k - 1});
  return launch_status();
}

}  // extern "C"
