// libeqa_hip.so, part 2 of 5 -- reductions of the canonicalization network's feature map: group pooling + argmax (I3, I4),
// window sums of the linearised last convolution (NCHW and channels-last), their GEMV, bias + ReLU.  C ABI: include/eqa_hip.h.
#include "eqa_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// I3 + I4: group pooling (mean over channels and space per group slot) and orientation argmax
// ------------------------------------------------------------------------------------------------

constexpr int kPoolSplitTarget = 2048;  // blocks wanted in flight for the streaming pass


// grid (splits, B).  Block (b, s) streams channels [s*cps, (s+1)*cps) of image b: each wave takes whole
// (channel, group) planes of HW contiguous floats with float4 loads, reduces them with shuffles and
// adds the plane sum to its own per-group accumulator in LDS (fp64: the cross-plane sum is the long one).
template <bool VEC>
__global__ __launch_bounds__(kThreads) void group_pool_partial_kernel(const float* __restrict__ feat,
                                                                     double* __restrict__ partial, int Cf, int G,
                                                                     int HW, int cps, int splits) {
  extern __shared__ __attribute__((aligned(16))) double acc[];  // [4 waves][G]
  const int s = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < 4 * G; k += kThreads) acc[k] = 0.0;
  __syncthreads();
  const int c_lo = s * cps, c_hi = min(c_lo + cps, Cf);
  const int planes = (c_hi - c_lo) * G;
  const float* base = feat + ((size_t)b * Cf + c_lo) * G * HW;
  for (int p = wave; p < planes; p += 4) {
    const float* pl = base + (size_t)p * HW;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (VEC) {
      const float4* p4 = reinterpret_cast<const float4*>(pl);
      const int n4 = HW >> 2;
      for (int k = lane; k < n4; k += 64) {
        const float4 t = p4[k];
        v0 += t.x; v1 += t.y; v2 += t.z; v3 += t.w;
      }
    } else {
      for (int k = lane; k < HW; k += 64) v0 += pl[k];
    }
    const float tot = wave_sum_f((v0 + v1) + (v2 + v3));
    if (lane == 0) acc[wave * G + (p % G)] += (double)tot;
  }
  __syncthreads();
  if (tid < G) partial[((size_t)b * splits + s) * G + tid] = (acc[tid] + acc[G + tid]) + (acc[2 * G + tid] + acc[3 * G + tid]);
}

// one wave per image: lanes g < G own one orientation each; argmax by butterfly shuffles with
// (value, index) pairs, smaller index winning ties == torch.argmax's first-maximum rule.
__device__ __forceinline__ void wave_argmax_store(float v, int g, int G, int32_t* out) {
  int idx = (g < G) ? g : 0x7fffffff;
  float val = (g < G) ? v : -INFINITY;
  // NaN handling as torch: a NaN is "greater" than everything; first NaN wins.
  bool isn = (g < G) && (v != v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(val, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    const bool on = __shfl_xor((int)isn, o, 64) != 0;
    bool take;
    if (on != isn) take = on;
    else if (on) take = oi < idx;
    else take = (ov > val) || (ov == val && oi < idx);
    if (take) { val = ov; idx = oi; isn = on; }
  }
  if (g == 0) *out = idx;
}

__global__ __launch_bounds__(kThreads) void group_pool_finalize_kernel(const double* __restrict__ partial,
                                                                      float* __restrict__ act,
                                                                      int32_t* __restrict__ gidx, int B, int G,
                                                                      int splits, double inv_count) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  for (int g0 = 0; g0 < G; g0 += 64) {  // G <= 64 in every supported group; loop keeps it general for act
    const int g = g0 + lane;
    float a = 0.f;
    if (g < G) {
      double sum = 0.0;
      for (int s = 0; s < splits; ++s) sum += partial[((size_t)b * splits + s) * G + g];
      a = (float)(sum * inv_count);
      act[(size_t)b * G + g] = a;
    }
    if (G <= 64 && gidx) wave_argmax_store(a, g, G, gidx + b);
  }
}

__global__ __launch_bounds__(kThreads) void group_argmax_kernel(const float* __restrict__ act, int32_t* __restrict__ gidx,
                                                               int B, int G) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float a = lane < G ? act[(size_t)b * G + lane] : 0.f;
  wave_argmax_store(a, lane, G, gidx + b);
}

int pool_splits(int B, int Cf) {
  int s = (kPoolSplitTarget + B - 1) / B;
  if (s > Cf) s = Cf;
  if (s < 1) s = 1;
  return s;
}

// ------------------------------------------------------------------------------------------------
// Window sums: the exact linear shortcut for "last convolution -> group mean".
//   mean_{o,y,x} conv(h, W)[o,g,y,x] = (1/count) * sum_{c,u,v} (sum_o W[(o,g),c,u,v]) * S[c,u,v] + mean(bias),
//   S[c,u,v] = sum_{y<OH, x<OW} act(h[c, y+u, x+v]),  OH = H-k+1, OW = W-k+1,  act = relu?(scale*h + shift).
// One block per (image, channel) plane: the plane is read from HBM exactly once (float4), transformed and parked in
// LDS; column totals and the k-1 leading / trailing rows give the k row-window sums per column, then the same trick
// across columns gives the k*k outputs.  fp64 accumulation (the consumer compares orientations by tiny margins).
// HBM-bound: H*W*4 bytes per plane in, k*k*8 bytes out.
// ------------------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kThreads) void window_sums_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu,
                                                              double* __restrict__ out, int C, int H, int W, int k) {
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  const int HW = H * W;
  float* sp = ws_smem;                                                   // [H*W] transformed plane
  double* cs = reinterpret_cast<double*>(ws_smem + ((HW + 3) & ~3));      // [k][W] row-window sums per column
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
  const float* p = x + (size_t)plane * HW;
  const int tid = threadIdx.x;
  auto act = [&](float v) {
    v = v * sc + sh;
    return (relu && v < 0.0f) ? 0.0f : v;
  };
  if (VEC) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float4* s4 = reinterpret_cast<float4*>(sp);
    for (int i = tid; i < (HW >> 2); i += kThreads) {
      float4 t = p4[i];
      t.x = act(t.x); t.y = act(t.y); t.z = act(t.z); t.w = act(t.w);
      s4[i] = t;
    }
  } else {
    for (int i = tid; i < HW; i += kThreads) sp[i] = act(p[i]);
  }
  __syncthreads();
  const int OH = H - k + 1, OW = W - k + 1;
  // column x: total over rows, minus the u leading and the (k-1-u) trailing rows -> rows [u, u+OH)
  for (int xcol = tid; xcol < W; xcol += kThreads) {
    double tot = 0.0;
    for (int y = 0; y < H; ++y) tot += (double)sp[y * W + xcol];
    double pre = 0.0;
    for (int u = 0; u < k; ++u) {
      double suf = 0.0;
      for (int y = u + OH; y < H; ++y) suf += (double)sp[y * W + xcol];
      cs[u * W + xcol] = tot - pre - suf;
      pre += (double)sp[u * W + xcol];
    }
  }
  __syncthreads();
  // output (u, v): columns [v, v+OW) of row-window u.  One thread per output; k*k <= 64 of them.
  if (tid < k * k) {
    const int u = tid / k, v = tid - u * k;
    double acc = 0.0;
    for (int xcol = v; xcol < v + OW; ++xcol) acc += cs[u * W + xcol];
    out[(size_t)plane * (k * k) + tid] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Channels-last (NHWC) companions: MIOpen's fp32 implicit-GEMM convolutions run natively in NHWC, so the inference path
// of the canonicalization network stays in that layout end to end (no NCHW<->NHWC transposes).
//  * bias_relu_nhwc_kernel: x[p][c] = max(x[p][c] + bias[c], 0) in place, float4 over channels.
//  * window_sums_nhwc: same S[b,c,u,v] as window_sums_kernel, for a (B,H,W,C) buffer.  Rows are cut into segments:
//    each of the 2(k-1) border rows alone, the interior in bands.  A segment kernel streams its rows once (a wave reads
//    one pixel's channels = contiguous floats per instruction) and emits, per channel, the segment total and the sums of
//    the k-1 leading / trailing columns.  Every window sum is then total - excluded rows - excluded columns + their
//    intersections (inclusion-exclusion), assembled per (image, channel) in fp64 by a small finalize kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bias_relu_nhwc_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                                 size_t n_vec, int C4) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  const size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x;
  QuadWalk w(i0, stride, C4);
  for (size_t i = i0; i < n_vec; i += stride, w.next()) {
    const int c4 = (int)w.q;
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 b = reinterpret_cast<const float4*>(bias)[c4];
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    reinterpret_cast<float4*>(x)[i] = v;
  }
}


// Backward of the window sums (training): a pixel receives the sum of dS over the windows that contain it, which depends
// only on the CLASS of its row and of its column -- border index i < nb, interior (class nb), or bottom / right border
// nb + 1 + (i - (n - nb)).  table:(B, 2nb+1, 2nb+1, C) holds one gradient per class pair (built by the caller from dS, a
// few KB per image); this kernel expands it to dx:(B,H,W,C): a pure streaming write, one block per (image, row), float4
// over channels.  (As two torch index lookups the expansion took 1.6 ms of the 53 ms training step; 2.0 GB written.)
__global__ __launch_bounds__(kThreads) void window_sums_bwd_expand_kernel(const float* __restrict__ table, float* __restrict__ dx,
                                                                         int H, int W, int C4, int nb) {
  const int y = blockIdx.x % H;
  const size_t b = blockIdx.x / H;
  const int T = 2 * nb + 1;
  auto cls = [nb](int i, int n) { return i < nb ? i : (i >= n - nb ? i - (n - nb) + nb + 1 : nb); };
  const float4* trow = reinterpret_cast<const float4*>(table) + (b * T + cls(y, H)) * (size_t)T * C4;
  float4* orow = reinterpret_cast<float4*>(dx) + (b * H + y) * (size_t)W * C4;
  for (int i = threadIdx.x; i < W * C4; i += kThreads) {
    const int x = i / C4, c4 = i - x * C4;
    orow[i] = trow[cls(x, W) * C4 + c4];
  }
}

// segment s of image b: rows [seg_y0(s), seg_y1(s)).  Output per (b, s, c): 1 + 2(k-1) floats
//   [0] total, [1 + j] column j, [1 + (k-1) + j] column W-k+1+j      (j < k-1), all over the segment's rows.
__device__ __forceinline__ void ws_segment_rows(int s, int H, int k, int nbands, int& y0, int& y1) {
  const int nb = k - 1;
  if (s < nb) { y0 = s; y1 = s + 1; return; }                       // top border rows
  if (s < 2 * nb) { y0 = H - nb + (s - nb); y1 = y0 + 1; return; }  // bottom border rows
  const int lo = nb, hi = H - nb;                                   // interior rows, cut into nbands bands
  const int band = s - 2 * nb, rows = hi - lo;
  y0 = lo + (int)(((long long)rows * band) / nbands);
  y1 = lo + (int)(((long long)rows * (band + 1)) / nbands);
}

// drop_threshold != 0: act also carries the dropout of a training-mode hidden block (mask = the hash batchnorm.hip uses, on the
// element's quad index in the (B, H, W, C / 4) map; kept elements x inv_keep) -- the block's output is consumed here without ever
// being written (eqa_window_sums_nhwc_act).
// RELU / DROP are template arguments: as run-time flags they were (uniform) branches inside the load-and-activate step, and the
// wait-count pass then put s_waitcnt vmcnt(0) behind EVERY load -- one load in flight per wave, 2.7-3.0 TB/s (seen in the ISA only).
template <bool RELU, bool DROP, int MAXB = kWsSmallBorder>   // MAXB: border columns held in registers (k - 1 <= MAXB)
__global__ __launch_bounds__(kThreads) void window_sums_nhwc_segment_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                           const float* __restrict__ shift, int /*relu*/,
                                                                           float* __restrict__ part, int C, int H, int W, int k,
                                                                           int nbands, uint32_t drop_threshold, float inv_keep,
                                                                           uint32_t seed) {
  __shared__ float4 s_tot[4][64];  // only the totals need a cross-wave sum; a border column has ONE owner wave
  const int s = blockIdx.x, b = blockIdx.y;
  const int nseg = gridDim.x;
  int y0, y1;
  ws_segment_rows(s, H, k, nbands, y0, y1);
  const int nb = k - 1, nval = 1 + 2 * nb;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Q = C >> 2;  // channel quads
  // fewer than 64 channel quads (C = 32: 8): a wave instruction covers P = 64 / Q pixels instead of leaving 64 - Q lanes idle
  // (the CIFAR-shaped configuration ran this kernel at 0.95 TB/s with 8 of 64 lanes loading); the P pixel groups of a lane's
  // quad are added up by a butterfly before the totals leave the wave
  const int P = (Q < 64 && (64 % Q) == 0) ? 64 / Q : 1;
  const int psub = P > 1 ? lane / Q : 0;
  const float4* xb = reinterpret_cast<const float4*>(x + (size_t)b * H * W * C);
  for (int q0 = 0; q0 < Q; q0 += 64) {
    const int q = P > 1 ? lane % Q : q0 + lane;
    const bool on = q < Q;
    const int qq = on ? q : Q - 1;
    const float4 sc = scale ? reinterpret_cast<const float4*>(scale)[qq] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? reinterpret_cast<const float4*>(shift)[qq] : make_float4(0.f, 0.f, 0.f, 0.f);
    // Load and activation are two steps so that a trip can request all its pixels before it touches the first: `raw` issues the
    // load, `act` makes the value opaque (left alone, the compiler sinks the load of a component under the dropout mask's select:
    // a dword load inside an exec-masked branch, waited for on the spot) and applies affine map, ReLU and dropout mask.
    typedef float ws_f4 __attribute__((ext_vector_type(4)));
    auto raw = [&](int y, int xc) { return reinterpret_cast<const ws_f4*>(xb)[((size_t)y * W + xc) * Q + qq]; };
    auto act = [&](ws_f4 r, int y, int xc) {
      asm volatile("" : "+v"(r));
      float4 v = make_float4(r[0], r[1], r[2], r[3]);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if constexpr (DROP) {
        bool keep[4];
        dropout_keep_quad<true>((size_t)b * H * W * Q + ((size_t)y * W + xc) * Q + qq, drop_threshold, seed, keep);
        v.x = keep[0] ? v.x * inv_keep : 0.f; v.y = keep[1] ? v.y * inv_keep : 0.f;
        v.z = keep[2] ? v.z * inv_keep : 0.f; v.w = keep[3] ? v.w * inv_keep : 0.f;
      }
      return v;
    };
    auto ld = [&](int y, int xc) { return act(raw(y, xc), y, xc); };
    float4 acc[1 + 2 * MAXB];
#pragma unroll
    for (int i = 0; i < 1 + 2 * MAXB; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int xs = 4 * P;   // pixels covered by the block's four waves per step
    int y = y0;
    if (W <= xs && nb == 0) {
      // rows no wider than one step of the block and no border columns (the CIFAR-shaped map in front of a 1 x 1 tail: 28 pixels,
      // P = 8): the loop below would have ONE load in flight per thread and row, a round trip per row (3.0 TB/s at 8192 images).
      // Five rows' loads go out together; the sums are taken row by row in the same order.
      const int xc = wave * P + psub;
      for (; y < y1; y += 5) {   // ~10 rows per band: two trips
        ws_f4 rr[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) rr[u] = raw(min(y + u, y1 - 1), min(xc, W - 1));
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const float4 r = act(rr[u], y + u, xc);
          if (xc < W && y + u < y1) { acc[0].x += r.x; acc[0].y += r.y; acc[0].z += r.z; acc[0].w += r.w; }
        }
      }
    }
    for (; y < y1; ++y) {
      // the 4 waves take pixels x = wave, wave+4, ...: each load instruction reads one pixel's channels, contiguous
      float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
      int xc = wave * P + psub;
      for (; xc + 3 * xs < W; xc += 4 * xs) {   // four pixels requested together; the additions in the order of two trips of the loop below
        const ws_f4 r0 = raw(y, xc), r1 = raw(y, xc + xs), r2 = raw(y, xc + 2 * xs), r3 = raw(y, xc + 3 * xs);
        const float4 a = act(r0, y, xc), c2 = act(r1, y, xc + xs), a2 = act(r2, y, xc + 2 * xs), c3 = act(r3, y, xc + 3 * xs);
        t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w;
        t1.x += c2.x; t1.y += c2.y; t1.z += c2.z; t1.w += c2.w;
        t0.x += a2.x; t0.y += a2.y; t0.z += a2.z; t0.w += a2.w;
        t1.x += c3.x; t1.y += c3.y; t1.z += c3.z; t1.w += c3.w;
      }
      for (; xc + xs < W; xc += 2 * xs) {
        const ws_f4 r0 = raw(y, xc), r1 = raw(y, xc + xs);
        const float4 a = act(r0, y, xc), c2 = act(r1, y, xc + xs);
        t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w;
        t1.x += c2.x; t1.y += c2.y; t1.z += c2.z; t1.w += c2.w;
      }
      if (xc < W) { const float4 a = ld(y, xc); t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w; }
      acc[0].x += t0.x + t1.x; acc[0].y += t0.y + t1.y; acc[0].z += t0.z + t1.z; acc[0].w += t0.w + t1.w;
      // border columns (static j, wave-uniform owner): re-read from L1
#pragma unroll
      for (int j = 0; j < MAXB; ++j) {
        if (j < nb && psub == 0) {   // (one pixel: the lanes of pixel group 0 own it)
          if ((j & 3) == wave) { const float4 a = ld(y, j); acc[1 + j].x += a.x; acc[1 + j].y += a.y; acc[1 + j].z += a.z; acc[1 + j].w += a.w; }
          const int xr = W - nb + j;
          if ((xr & 3) == wave) {
            const float4 a = ld(y, xr);
            acc[1 + MAXB + j].x += a.x; acc[1 + MAXB + j].y += a.y;
            acc[1 + MAXB + j].z += a.z; acc[1 + MAXB + j].w += a.w;
          }
        }
      }
    }
    auto put = [&](int i, const float4& v) {  // value index i of this lane's 4 channels
      float* o = part + ((((size_t)b * nseg + s) * Q + q) * 4) * nval + i;
      o[0] = v.x; o[nval] = v.y; o[2 * nval] = v.z; o[3 * nval] = v.w;
    };
    if (P > 1) {   // totals of the P pixel groups -> the lanes of group 0 (fixed order: deterministic)
      for (int off = 32; off >= Q; off >>= 1) {
        acc[0].x += __shfl_down(acc[0].x, off); acc[0].y += __shfl_down(acc[0].y, off);
        acc[0].z += __shfl_down(acc[0].z, off); acc[0].w += __shfl_down(acc[0].w, off);
      }
    }
    const bool owner = on && psub == 0;
    __syncthreads();
    s_tot[wave][lane] = acc[0];
#pragma unroll
    for (int j = 0; j < MAXB; ++j) {
      if (j < nb && owner) {
        if ((j & 3) == wave) put(1 + j, acc[1 + j]);
        if (((W - nb + j) & 3) == wave) put(1 + nb + j, acc[1 + MAXB + j]);
      }
    }
    __syncthreads();
    if (wave == 0 && owner) {
      const float4 a = s_tot[0][lane], b2 = s_tot[1][lane], c2 = s_tot[2][lane], d = s_tot[3][lane];
      put(0, make_float4((a.x + b2.x) + (c2.x + d.x), (a.y + b2.y) + (c2.y + d.y), (a.z + b2.z) + (c2.z + d.z), (a.w + b2.w) + (c2.w + d.w)));
    }
  }
}

// part: (B, nseg, C, nval) -> out (B, C, k, k) fp64.  Block = one image x 32 channels.  For a fixed (image, segment) the
// block's 32 channels x nval values are ONE contiguous run of part, so thread t owns element t of that run (channel t / nval,
// value t % nval) and walks the segment axis: every load instruction is a contiguous row (the first version gave each
// thread a whole channel, 36-byte lane stride: 0.26 ms for 207 MB; this one: see HISTORY.md).  Fixed order, deterministic.
// Then one thread per channel assembles the k*k window sums from the totals and the 2(k-1) border-row segments.
constexpr int kFinThreads = 320;   // k = 5: a block's run is 32 channels x 9 values = 288 elements -- one trip of 320 threads, not two of 256

// `sub` > 1: every segment arrives in `sub` pieces (one per tile column of the producer), laid out as consecutive segments;
// the pieces of a border row are added up first (in fp32, as a single producer thread would have).
template <int KW, int MAXB = kWsSmallBorder>   // the window size as a template argument (0: any): the assembly below then unrolls into independent LDS reads
__global__ __launch_bounds__(kFinThreads) void window_sums_nhwc_finalize_kernel(const float* __restrict__ part, double* __restrict__ out,
                                                                            int B, int C, int k_rt, int nseg, int sub) {
  constexpr int kFinVals = 1 + 2 * MAXB;
  const int k = KW ? KW : k_rt;
  __shared__ double s_tot[kFinCh * kFinVals];
  __shared__ float s_brd[2 * MAXB][kFinCh * kFinVals];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * kFinCh;
  const int nch = min(kFinCh, C - c0);
  const int nb = k - 1, nval = 1 + 2 * nb;
  const int run = nch * nval;  // contiguous elements of this block in one (image, segment) row
  const float* base = part + ((size_t)b * nseg * C + c0) * nval;
  const size_t seg_stride = (size_t)C * nval;
  for (int e = threadIdx.x; e < run; e += blockDim.x) {
    const float* q = base + e;
    double acc = 0.0;
    int s = 0;
    if (sub <= 2) {
      // border-row segments are needed individually as well.  All their pieces are requested before the first is used (the
      // rolled form paid one round trip per piece, 16 in a row at k = 5: most of the kernel's 39 us for 85 MB)
      float v[2 * MAXB][2];
#pragma unroll
      for (int i = 0; i < 2 * MAXB; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) v[i][t] = (i < 2 * nb && t < sub) ? q[(size_t)(i * sub + t) * seg_stride] : 0.0f;
#pragma unroll
      for (int i = 0; i < 2 * MAXB; ++i) {
        if (i < 2 * nb) {
          const float w = sub == 2 ? v[i][0] + v[i][1] : v[i][0];
          s_brd[i][e] = w;
          acc += (double)w;
        }
      }
      s = 2 * nb;
    } else {
      for (; s < 2 * nb; ++s) {
        float v = q[(size_t)(s * sub) * seg_stride];
        for (int t = 1; t < sub; ++t) v += q[(size_t)(s * sub + t) * seg_stride];
        s_brd[s][e] = v;
        acc += (double)v;
      }
    }
    s *= sub;
    for (; s + 8 <= nseg; s += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = q[(size_t)(s + j) * seg_stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (double)v[j];
    }
    if (s < nseg) {   // the last < 8 pieces: requested together as well (same order of additions)
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s + j < nseg ? q[(size_t)(s + j) * seg_stride] : 0.0f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s + j < nseg) acc += (double)v[j];
    }
    s_tot[e] = acc;
  }
  __syncthreads();
  // One (channel, window) pair per thread and trip (one thread per channel doing its k*k windows in turn left 224 of the
  // block's 256 threads idle through ~1100 conditional fp64 adds: 110 of the kernel's 124 us).
  // window (u, v) keeps rows [u, H-nb+u) and columns [v, W-nb+v): it excludes top border rows r < u, bottom border rows
  // r >= u (of the nb bottom rows), left border columns j < v and right border columns j >= v
  // (Every block reaches this point at about the same time -- the grid is one wave of blocks and the loads above are bandwidth-bound
  // -- so nothing overlaps the assembly: as a rolled chain of 24 dependent LDS reads + fp64 adds per window it was 16 of the
  // kernel's 36 us.  With k known the reads of a window are independent instructions; the additions keep their order.)
  const int kk = k * k;
#ifdef EQA_FIN_NOPHASE2   // experiment: how much of the kernel is the window assembly
  if (threadIdx.x < nch * kk && s_tot[0] == 12345.0) out[0] = 1.0;
  if (true) return;
#endif
  for (int it = threadIdx.x; it < nch * kk; it += blockDim.x) {
    const int cl = it / kk, uv = it - cl * kk;
    const int u = uv / k, v = uv - u * k;
    if constexpr (KW > 0) {
      constexpr int NB = KW - 1;
      float p0[NB], pc[NB][NB];
      double tc[NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) p0[r] = s_brd[r < u ? r : NB + r][cl * nval];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int i = j < v ? 1 + j : 1 + NB + j;
        tc[j] = s_tot[cl * nval + i];
#pragma unroll
        for (int r = 0; r < NB; ++r) pc[j][r] = s_brd[r < u ? r : NB + r][cl * nval + i];
      }
      double a = s_tot[cl * nval];
#pragma unroll
      for (int r = 0; r < NB; ++r) a -= (double)p0[r];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        a -= tc[j];
#pragma unroll
        for (int r = 0; r < NB; ++r) a += (double)pc[j][r];
      }
      out[((size_t)b * C + c0) * kk + it] = a;
    } else {
      auto P = [&](int s, int i) { return (double)s_brd[s][cl * nval + i]; };
      double a = s_tot[cl * nval];
      for (int r = 0; r < nb; ++r) a -= r < u ? P(r, 0) : P(nb + r, 0);
      for (int j = 0; j < nb; ++j) {
        const int i = j < v ? 1 + j : 1 + nb + j;  // the excluded column of this pair: left border j < v, right border j >= v
        a -= s_tot[cl * nval + i];
        for (int r = 0; r < nb; ++r) a += r < u ? P(r, i) : P(nb + r, i);  // excluded rows x excluded columns were subtracted twice
      }
      out[((size_t)b * C + c0) * kk + it] = a;
    }
  }
}

// The GEMV that follows the window sums (pooling.window_sums_to_activations; reference: the last convolution followed by
// torch.mean over (C, H', W') -- escnn_networks.py:115, custom_equivariant_networks.py:91):
//   act[b][e] = float( scale * sum_j S[b][j] * Wm[e][j] + shift ),  S fp64 (B, K), Wm fp64 (E, K), E <= 16.
// One block per image, fixed-order tree reduction (deterministic).  rocBLAS' dgemm takes 0.2 ms for this 256 x 6400 x 8 shape.
constexpr int kGemvMaxE = 16;

template <int THREADS, int EMAX, int COLS>   // EMAX >= E; COLS columns per thread and trip
__global__ __launch_bounds__(THREADS) void sums_gemv_kernel(const double* __restrict__ S, const double* __restrict__ Wm,
                                                            float* __restrict__ act, int K, int E, double scale, double shift) {
  __shared__ double s_red[THREADS / 64][kGemvMaxE];
  const int b = blockIdx.x;
  const double* sb = S + (size_t)b * K;
  double acc[kGemvMaxE];
#pragma unroll
  for (int e = 0; e < kGemvMaxE; ++e) acc[e] = 0.0;
  // COLS columns per trip: their (1 + E) x COLS loads are in flight together (one column per trip was latency-bound: 84 us for
  // 13 MB of S and a weight matrix that sits in L2; five per trip with a one-column tail loop: 20 us at K = 6400 = 25 columns
  // per thread; nine per trip for E <= 8, the tail inside the predicated batch: three trips).  A thread adds its columns in
  // ascending order whatever the batching: same bits.
  for (int j = threadIdx.x; j < K; j += COLS * THREADS) {
    double s[COLS], w[COLS][EMAX];
#pragma unroll
    for (int u = 0; u < COLS; ++u) {
      const bool in = j + u * THREADS < K;
      s[u] = in ? sb[j + u * THREADS] : 0.0;
#pragma unroll
      for (int e = 0; e < EMAX; ++e) w[u][e] = (in && e < E) ? Wm[(size_t)e * K + j + u * THREADS] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < COLS; ++u)
#pragma unroll
      for (int e = 0; e < EMAX; ++e)
        if (e < E && j + u * THREADS < K) acc[e] += s[u] * w[u][e];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < kGemvMaxE; ++e) {
    double v = acc[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < E) {
    double v = 0.0;
    for (int w = 0; w < THREADS / 64; ++w) v += s_red[w][threadIdx.x];
    act[(size_t)b * E + threadIdx.x] = (float)(v * scale + shift);
  }
}


// ---- Optimized canonicalizer, inference tail (I8 / I10): embeddings of the G views -> (B, G) activations -------------------
// z = relu(h * scale[d] + shift[d]) on (R, D) rows: BatchNorm1d (eval, folded) + ReLU of ConvNetwork's head
// (custom_nonequivariant_networks.py:62-67) in one pass
__global__ __launch_bounds__(kThreads) void affine_relu_rows_kernel(const float* __restrict__ h, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float* __restrict__ z, size_t n_vec,
                                                                   int Dq) {
  const size_t i0 = (size_t)blockIdx.x * kThreads + threadIdx.x, stride = (size_t)gridDim.x * kThreads;
  QuadWalk w(i0, stride, Dq);
  for (size_t i = i0; i < n_vec; i += stride, w.next()) {
    const int q = (int)w.q;
    const float4 v = reinterpret_cast<const float4*>(h)[i];
    const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
    reinterpret_cast<float4*>(z)[i] = make_float4(fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f),
                                                  fmaxf(v.z * sc.z + sh.z, 0.f), fmaxf(v.w * sc.w + sh.w, 0.f));
  }
}

// act[b][g] = cosine_similarity(ref, v[g * B + b]) = sum(ref / max(|ref|, eps) * v / max(|v|, eps)) -- torch's formula -- for the
// element-major (G * B, V) embeddings of the orbit (discrete_group.py:475-481: cosine_similarity, reshape(G, -1).T): one wave
// per embedding, lanes over V, wavefront-shuffle reductions; replaces ~10 element-wise launches and the transposed copy.
__global__ __launch_bounds__(kThreads) void cosine_group_activations_kernel(const float* __restrict__ v, const float* __restrict__ ref,
                                                                           float* __restrict__ act, int B, int G, int V, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);      // row of v: g * B + b
  if (r >= G * B) return;
  float rr = 0.f, vv = 0.f;
  for (int k = lane; k < V; k += 64) {
    const float a = ref[k], x = v[(size_t)r * V + k];
    rr += a * a;
    vv += x * x;
  }
  rr = wave_sum_f(rr);
  vv = wave_sum_f(vv);
  const float inv_r = 1.0f / fmaxf(sqrtf(rr), eps), inv_v = 1.0f / fmaxf(sqrtf(vv), eps);
  float dot = 0.f;
  for (int k = lane; k < V; k += 64) dot += (ref[k] * inv_r) * (v[(size_t)r * V + k] * inv_v);
  dot = wave_sum_f(dot);
  if (lane == 0) act[(size_t)(r % B) * G + r / B] = dot;
}

}  // namespace

int eqa::launch_window_sums_nhwc_finalize(const float* part, double* S, int B, int C, int k, int nseg, hipStream_t stream, int sub) {
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  // nseg counts the pieces: rows (or row groups) x sub
  const dim3 fgrid((C + kFinCh - 1) / kFinCh, B);
  // a block's run is <= 32 channels x (2k - 1) values: 320 threads take k = 5's 288 in one trip, smaller runs keep 256 (more blocks per CU)
  const dim3 fblock(std::min(C, kFinCh) * (2 * k - 1) > kThreads ? kFinThreads : kThreads);
  if (k == 5)
    hipLaunchKernelGGL(window_sums_nhwc_finalize_kernel<5>, fgrid, fblock, 0, stream, part, S, B, C, k, nseg, sub);
  else if (k == 3)
    hipLaunchKernelGGL(window_sums_nhwc_finalize_kernel<3>, fgrid, fblock, 0, stream, part, S, B, C, k, nseg, sub);
  else if (k - 1 <= kWsSmallBorder)
    hipLaunchKernelGGL(window_sums_nhwc_finalize_kernel<0>, fgrid, fblock, 0, stream, part, S, B, C, k, nseg, sub);
  else
    hipLaunchKernelGGL((window_sums_nhwc_finalize_kernel<0, kWsMaxBorder>), fgrid, fblock, 0, stream, part, S, B, C, k, nseg, sub);
  return launch_status();
}

namespace {

// Backward of the k * k shifted-window sums as a CLASS TABLE (escnn_networks._window_grad_table): a pixel's gradient depends only on
// the class of its row and of its column -- the k - 1 top / left border indices, the interior (representative: index k - 1), the
// k - 1 bottom / right ones -- and equals the sum of dS over the windows that contain it, a RECTANGLE of the (k, k) array:
//     table[b][t][s][c] = sum_{u in U(t)} sum_{v in V(s)} dS[b][c][u][v],   U(t) = [max(0, rep_t - (H - k)), min(k - 1, rep_t)].
// As an einsum over fp64 masks this was two batched fp64 GEMMs of (B C) tiny matrices through the library (295 + 103 us per step of
// the reference tutorial's first loop).  Here: a block = one image x 64 channels, dS[b][c0..c0+63] staged in LDS (coalesced),
// thread = channel: 2-D inclusive prefix sums in place, then every table entry from four corners; stores are 256-byte runs.
constexpr int kWgtCh = 64;
__global__ __launch_bounds__(kWgtCh) void window_grad_table_kernel(const double* __restrict__ dS, float* __restrict__ table, int C, int H,
                                                                  int W, int k) {
  extern __shared__ double s_p[];                  // [kWgtCh][k * k + 1]
  const int kk = k * k, stride = kk + 1;
  const int b = blockIdx.y, c0 = blockIdx.x * kWgtCh, nch = min(kWgtCh, C - c0);
  const double* src = dS + ((size_t)b * C + c0) * kk;
  for (int i = threadIdx.x; i < nch * kk; i += kWgtCh) s_p[(i / kk) * stride + i % kk] = src[i];
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= nch) return;
  double* p = s_p + c * stride;
  for (int u = 0; u < k; ++u)
    for (int v = 1; v < k; ++v) p[u * k + v] += p[u * k + v - 1];
  for (int u = 1; u < k; ++u)
    for (int v = 0; v < k; ++v) p[u * k + v] += p[(u - 1) * k + v];
  const int nb = k - 1, T = 2 * nb + 1;
  auto range = [&](int t, int n, int& lo, int& hi) {      // windows u with u <= rep <= u + (n - k)
    const int rep = t <= nb ? t : n - nb + (t - nb - 1);
    lo = max(0, rep - (n - k));
    hi = min(k - 1, rep);
  };
  auto rect = [&](int u0, int u1, int v0, int v1) -> double {
    if (u0 > u1 || v0 > v1) return 0.0;
    double r = p[u1 * k + v1];
    if (u0 > 0) r -= p[(u0 - 1) * k + v1];
    if (v0 > 0) r -= p[u1 * k + v0 - 1];
    if (u0 > 0 && v0 > 0) r += p[(u0 - 1) * k + v0 - 1];
    return r;
  };
  float* out = table + (size_t)b * T * T * C + c0 + c;
  for (int t = 0; t < T; ++t) {
    int u0, u1;
    range(t, H, u0, u1);
    for (int s = 0; s < T; ++s) {
      int v0, v1;
      range(s, W, v0, v1);
      out[((size_t)t * T + s) * C] = (float)rect(u0, u1, v0, v1);
    }
  }
}

// Backward of act = scale * S . Wm^T (sums_gemv_kernel): dS[b][j] = scale * sum_e dact[b][e] Wm[e][j]  (fp64, E <= 16)
__global__ __launch_bounds__(kThreads) void sums_gemv_bwd_ds_kernel(const float* __restrict__ dact, const double* __restrict__ Wm,
                                                                   double* __restrict__ dS, int K, int E, double scale) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= K) return;
  double acc = 0.0;
  for (int e = 0; e < E; ++e) acc += (double)dact[(size_t)b * E + e] * Wm[(size_t)e * K + j];
  dS[(size_t)b * K + j] = scale * acc;
}

// ... and dWm[e][j] = scale * sum_b dact[b][e] S[b][j]: thread = column j, blockIdx.y = a slice of the batch; the slices' partials
// (nslice, E, K) are summed in order by the second kernel (deterministic)
template <int MAXE>
__global__ __launch_bounds__(kThreads) void sums_gemv_bwd_dw_kernel(const float* __restrict__ dact, const double* __restrict__ S,
                                                                   double* __restrict__ partial, int B, int K, int E, int per_slice) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  const int b0 = blockIdx.y * per_slice, b1 = min(B, b0 + per_slice);
  if (j >= K) return;
  double acc[MAXE];
#pragma unroll
  for (int e = 0; e < MAXE; ++e) acc[e] = 0.0;
  for (int b = b0; b < b1; ++b) {
    const double sv = S[(size_t)b * K + j];
#pragma unroll
    for (int e = 0; e < MAXE; ++e)
      if (e < E) acc[e] += (double)dact[(size_t)b * E + e] * sv;
  }
#pragma unroll
  for (int e = 0; e < MAXE; ++e)
    if (e < E) partial[((size_t)blockIdx.y * E + e) * K + j] = acc[e];
}
__global__ __launch_bounds__(kThreads) void sums_gemv_bwd_dw_reduce_kernel(const double* __restrict__ partial, double* __restrict__ dW,
                                                                          size_t n, int nslice, double scale) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int s = 0; s < nslice; ++s) acc += partial[(size_t)s * n + i];
  dW[i] = scale * acc;
}

constexpr int kGemvBwdSlices = 16;

}  // namespace

extern "C" {

int64_t eqa_group_pool_workspace_bytes(int B, int Cf, int G, int HW) {
  (void)HW;
  if (B <= 0 || Cf <= 0 || G <= 0) return 0;
  return (int64_t)B * pool_splits(B, Cf) * G * (int64_t)sizeof(double);
}

int eqa_group_pool_argmax(const float* feat, float* act, int32_t* gidx, void* workspace, int B, int Cf, int G, int HW,
                          void* stream) {
  if (B == 0) return EQA_OK;
  if (!feat || !act || !workspace || B < 0 || Cf <= 0 || G <= 0 || HW <= 0) return EQA_ERR_INVALID_ARG;
  if (gidx && G > 64) return EQA_ERR_UNSUPPORTED;
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int splits = pool_splits(B, Cf);
  const int cps = (Cf + splits - 1) / splits;
  const int used = (Cf + cps - 1) / cps;  // splits that own at least one channel
  double* partial = (double*)workspace;
  const bool vec = (HW % 4 == 0) && (((uintptr_t)feat & 15) == 0);
  const size_t lds = (size_t)4 * G * sizeof(double);
  if (vec)
    hipLaunchKernelGGL((group_pool_partial_kernel<true>), dim3(used, B), dim3(kThreads), lds, st, feat, partial, Cf, G, HW, cps, used);
  else
    hipLaunchKernelGGL((group_pool_partial_kernel<false>), dim3(used, B), dim3(kThreads), lds, st, feat, partial, Cf, G, HW, cps, used);
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  const double inv_count = 1.0 / ((double)Cf * (double)HW);
  hipLaunchKernelGGL(group_pool_finalize_kernel, dim3((B + 3) / 4), dim3(kThreads), 0, st, partial, act, gidx, B, G, used, inv_count);
  return launch_status();
}

int eqa_group_argmax(const float* act, int32_t* gidx, int B, int G, void* stream) {
  if (B == 0 && G > 0 && G <= 64) return EQA_OK;
  if (!act || !gidx || B < 0 || G <= 0) return EQA_ERR_INVALID_ARG;
  if (G > 64) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(group_argmax_kernel, dim3((B + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream, act, gidx, B, G);
  return launch_status();
}

int eqa_window_sums(const float* x, const float* scale, const float* shift, int relu, double* out, int B, int C, int H,
                    int W, int k, void* stream) {
  if (!x || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || k > H || k > W) return EQA_ERR_INVALID_ARG;
  if (k > kMaxWinK) return EQA_ERR_UNSUPPORTED;
  const size_t lds = (size_t)((H * W + 3) & ~3) * sizeof(float) + (size_t)k * W * sizeof(double);
  if (lds > 64 * 1024 || (long long)B * C > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;  // plane must fit one block's LDS
  if (B == 0) return EQA_OK;
  const bool vec = ((H * W) % 4 == 0) && (((uintptr_t)x & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL((window_sums_kernel<true>), dim3((unsigned)(B * C)), dim3(kThreads), lds, st, x, scale, shift, relu, out, C, H, W, k);
  else
    hipLaunchKernelGGL((window_sums_kernel<false>), dim3((unsigned)(B * C)), dim3(kThreads), lds, st, x, scale, shift, relu, out, C, H, W, k);
  return launch_status();
}

int eqa_window_sums_gemv(const double* S, const double* Wm, float* act, int B, int K, int E, double scale, double shift,
                         void* stream) {
  if (B < 0 || K <= 0 || E <= 0) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if (!S || !Wm || !act) return EQA_ERR_INVALID_ARG;
  if (E > kGemvMaxE) return EQA_ERR_UNSUPPORTED;
  // (1024 threads per image -- one trip instead of five -- was measured: 74 us against 20, the fp64 operand sets spill)
  // columns per thread and trip: as many as a thread has, up to nine (a batch wider than the row wastes predicated instructions,
  // which shows where the grid is many waves of blocks: the CIFAR-shaped batch of 8192 images with K = 800)
  const int per_thread = (K + kThreads - 1) / kThreads;
  if (E <= 8 && per_thread > 5)
    hipLaunchKernelGGL((sums_gemv_kernel<kThreads, 8, 9>), dim3(B), dim3(kThreads), 0, (hipStream_t)stream, S, Wm, act, K, E, scale, shift);
  else if (E <= 8 && per_thread > 2)
    hipLaunchKernelGGL((sums_gemv_kernel<kThreads, 8, 5>), dim3(B), dim3(kThreads), 0, (hipStream_t)stream, S, Wm, act, K, E, scale, shift);
  else if (E <= 8)
    hipLaunchKernelGGL((sums_gemv_kernel<kThreads, 8, 2>), dim3(B), dim3(kThreads), 0, (hipStream_t)stream, S, Wm, act, K, E, scale, shift);
  else
    hipLaunchKernelGGL((sums_gemv_kernel<kThreads, kGemvMaxE, 5>), dim3(B), dim3(kThreads), 0, (hipStream_t)stream, S, Wm, act, K, E, scale,
                       shift);
  return launch_status();
}

int eqa_affine_relu_rows(const float* h, const float* scale, const float* shift, float* z, int64_t rows, int D, void* stream) {
  if (!h || !scale || !shift || !z || rows < 0 || D <= 0) return EQA_ERR_INVALID_ARG;
  if (D % 4 != 0 || (((uintptr_t)h | (uintptr_t)z | (uintptr_t)scale | (uintptr_t)shift) & 15)) return EQA_ERR_UNSUPPORTED;
  if (rows == 0) return EQA_OK;
  const size_t n_vec = (size_t)rows * (D / 4);
  const unsigned blocks = (unsigned)std::min<size_t>((n_vec + kThreads - 1) / kThreads, 8192);
  hipLaunchKernelGGL(affine_relu_rows_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, h, scale, shift, z, n_vec, D / 4);
  return launch_status();
}

int eqa_cosine_group_activations(const float* v, const float* ref, float* act, int B, int G, int V, float eps, void* stream) {
  if (!v || !ref || !act || B < 0 || G <= 0 || V <= 0) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if ((int64_t)B * G > 0x7fffffff) return EQA_ERR_UNSUPPORTED;
  const int rows = B * G, per = kThreads / 64;
  hipLaunchKernelGGL(cosine_group_activations_kernel, dim3((rows + per - 1) / per), dim3(kThreads), 0, (hipStream_t)stream, v, ref, act, B,
                     G, V, eps);
  return launch_status();
}

int eqa_bias_relu_nhwc(float* x, const float* bias, int64_t n_pixels, int C, void* stream) {
  if (!x || !bias || n_pixels < 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (C % 4 != 0 || (((uintptr_t)x | (uintptr_t)bias) & 15)) return EQA_ERR_UNSUPPORTED;
  if (n_pixels == 0) return EQA_OK;
  const size_t n_vec = (size_t)n_pixels * (C / 4);
  const unsigned blocks = (unsigned)((n_vec + kThreads - 1) / kThreads < 8192 ? (n_vec + kThreads - 1) / kThreads : 8192);
  hipLaunchKernelGGL(bias_relu_nhwc_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, x, bias, n_vec, C / 4);
  return launch_status();
}

int eqa_window_sums_bwd_expand_nhwc(const float* table, float* dx, int B, int H, int W, int C, int k, void* stream) {
  if (!table || !dx || B < 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || H < 2 * k - 1 || W < 2 * k - 1) return EQA_ERR_INVALID_ARG;
  if (C % 4 != 0 || (((uintptr_t)table | (uintptr_t)dx) & 15) || (size_t)B * H > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  hipLaunchKernelGGL(window_sums_bwd_expand_kernel, dim3((unsigned)((size_t)B * H)), dim3(kThreads), 0, (hipStream_t)stream, table,
                     dx, H, W, C / 4, k - 1);
  return launch_status();
}

static int ws_nhwc_bands(int H, int k) {
  const int interior = H - 2 * (k - 1);
  int nb = (interior + 9) / 10;  // ~10 rows per band
  if (nb < 1) nb = 1;
  return nb;
}

int64_t eqa_window_sums_nhwc_workspace_bytes(int B, int C, int H, int k) {
  if (B <= 0 || C <= 0 || H <= 0 || k <= 0) return 0;
  const int nseg = 2 * (k - 1) + ws_nhwc_bands(H, k);
  return (int64_t)B * nseg * C * (1 + 2 * (k - 1)) * (int64_t)sizeof(float);
}

static int window_sums_nhwc_impl(const float* x, const float* scale, const float* shift, int relu, float drop_p, uint32_t seed, double* out,
                                 void* workspace, int B, int C, int H, int W, int k, void* stream);

int eqa_window_sums_nhwc(const float* x, const float* scale, const float* shift, int relu, double* out, void* workspace,
                         int B, int C, int H, int W, int k, void* stream) {
  return window_sums_nhwc_impl(x, scale, shift, relu, 0.0f, 0u, out, workspace, B, C, H, W, k, stream);
}

int eqa_window_sums_nhwc_act(const float* x, const float* scale, const float* shift, int relu, float drop_p, uint32_t seed, double* out,
                             void* workspace, int B, int C, int H, int W, int k, void* stream) {
  if (!(drop_p >= 0.0f && drop_p < 1.0f)) return EQA_ERR_INVALID_ARG;
  return window_sums_nhwc_impl(x, scale, shift, relu, drop_p, seed, out, workspace, B, C, H, W, k, stream);
}

static int window_sums_nhwc_impl(const float* x, const float* scale, const float* shift, int relu, float drop_p, uint32_t seed, double* out,
                                 void* workspace, int B, int C, int H, int W, int k, void* stream) {
  if (!x || !out || !workspace || B < 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return EQA_ERR_INVALID_ARG;
  // needs disjoint top / bottom (left / right) border sets and at least one interior row
  if (k > kMaxWinK || C % 4 != 0 || H < 2 * (k - 1) + 1 || W < 2 * (k - 1) + 1 || B > 65535) return EQA_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (scale && (((uintptr_t)scale) & 15)) || (shift && (((uintptr_t)shift) & 15))) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nbands = ws_nhwc_bands(H, k);
  const int nseg = 2 * (k - 1) + nbands;
  const uint32_t thr = dropout_threshold(drop_p);
#define EQA_WS_SEG(R_, D_)                                                                                                     \
  do {                                                                                                                         \
    if (k - 1 <= kWsSmallBorder)                                                                                               \
      hipLaunchKernelGGL((window_sums_nhwc_segment_kernel<R_, D_>), dim3(nseg, B), dim3(kThreads), 0, st, x, scale, shift, relu, \
                         (float*)workspace, C, H, W, k, nbands, thr, 1.0f / (1.0f - drop_p), seed);                            \
    else                                                                                                                       \
      hipLaunchKernelGGL((window_sums_nhwc_segment_kernel<R_, D_, kWsMaxBorder>), dim3(nseg, B), dim3(kThreads), 0, st, x, scale, \
                         shift, relu, (float*)workspace, C, H, W, k, nbands, thr, 1.0f / (1.0f - drop_p), seed);               \
  } while (0)
  if (relu && thr) EQA_WS_SEG(true, true);
  else if (relu) EQA_WS_SEG(true, false);
  else if (thr) EQA_WS_SEG(false, true);
  else EQA_WS_SEG(false, false);
#undef EQA_WS_SEG
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  return eqa::launch_window_sums_nhwc_finalize((const float*)workspace, out, B, C, k, nseg, st, 1);
}

int eqa_window_grad_table(const double* dS, float* table, int B, int C, int H, int W, int k, void* stream) {
  if (B < 0 || C <= 0 || k < 1 || H < k || W < k) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if (!dS || !table) return EQA_ERR_INVALID_ARG;
  if (k > kMaxWinK || B > 65535 || H < 2 * (k - 1) + 1 || W < 2 * (k - 1) + 1) return EQA_ERR_UNSUPPORTED;
  const size_t lds = (size_t)kWgtCh * (k * k + 1) * sizeof(double);
  hipLaunchKernelGGL(window_grad_table_kernel, dim3((C + kWgtCh - 1) / kWgtCh, B), dim3(kWgtCh), lds, (hipStream_t)stream, dS, table, C, H, W,
                     k);
  return launch_status();
}

int64_t eqa_window_sums_gemv_bwd_workspace_bytes(int K, int E) { return K <= 0 || E <= 0 ? 0 : (int64_t)kGemvBwdSlices * E * K * 8; }

int eqa_window_sums_gemv_bwd(const float* dact, const double* Wm, const double* S, double* dS, double* dWm, void* workspace, int B, int K,
                             int E, double scale, void* stream) {
  if (B < 0 || K <= 0 || E <= 0) return EQA_ERR_INVALID_ARG;
  if (E > kGemvMaxE) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    if (dWm && hipMemsetAsync(dWm, 0, (size_t)E * K * 8, st) != hipSuccess) return EQA_ERR_LAUNCH;
    return EQA_OK;
  }
  if (!dact || (dS && !Wm) || (dWm && (!S || !workspace))) return EQA_ERR_INVALID_ARG;
  const unsigned gx = (unsigned)((K + kThreads - 1) / kThreads);
  if (dS) {
    if (B > 65535) return EQA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sums_gemv_bwd_ds_kernel, dim3(gx, B), dim3(kThreads), 0, st, dact, Wm, dS, K, E, scale);
    if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  }
  if (dWm) {
    const int per = (B + kGemvBwdSlices - 1) / kGemvBwdSlices;
    const int nslice = (B + per - 1) / per;
    if (E <= 8)
      hipLaunchKernelGGL((sums_gemv_bwd_dw_kernel<8>), dim3(gx, nslice), dim3(kThreads), 0, st, dact, S, (double*)workspace, B, K, E, per);
    else
      hipLaunchKernelGGL((sums_gemv_bwd_dw_kernel<kGemvMaxE>), dim3(gx, nslice), dim3(kThreads), 0, st, dact, S, (double*)workspace, B, K, E,
                         per);
    if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
    const size_t n = (size_t)E * K;
    hipLaunchKernelGGL(sums_gemv_bwd_dw_reduce_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       (const double*)workspace, dWm, n, nslice, scale);
  }
  return launch_status();
}

}  // extern "C"
