// libeqa_hip.so, part 3 of 5 -- Winograd F(2x2,5x5) / F(4x4,5x5) input and output transforms for the 5x5 group
// convolutions (I2a).  C ABI: include/eqa_hip.h.  Design notes: HISTORY.md section 3.4.
#include "eqa_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// Winograd F(m x m, 5x5) transforms, m = 2 or 4, for the 5x5 regular->regular layers of the canonicalization network
// (inference, channels-last).  Cook-Toom with N = m + 4 points per axis:
//   m = 2: {0, 1, -1, 2, -2, inf}             36 multiplies per 2x2 outputs = 9 per output   (direct: 25)
//   m = 4: {0, 1, -1, 2, -2, 1/2, -1/2, inf}  64 multiplies per 4x4 outputs = 4 per output
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A
// B^T and A^T have only small integers / dyadic fractions (exact in fp32); G is rational and is applied to the static
// filters offline in fp64 (images/canonicalization_networks/winograd.py).  The channel contraction of the N*N
// transformed planes is a plain strided-batched fp32 GEMM (library); these kernels are the HBM-bound ends:
//   input  : x (nimg, H, W, C) -> V (tiles, N*N, C),  tile (ty,tx) = rows m*ty..m*ty+N-1, cols m*tx..m*tx+N-1
//   output : M (tiles, N*N, C) -> y (nimg, OH, OW, C) = [relu](A^T M A + bias),  OH = H-4, OW = W-4 (multiples of m)
// fp32 error against an fp64 convolution, 256 channels: m = 2: 7e-6 of max|y|, m = 4: 9e-6 (direct fp32: 3e-7).
//
// Layout: V and M are tile-major -- every tile owns one contiguous N*N*C run, and plane `a` is the strided matrix
// V[:, a, :] (row stride N*N*C) the batched GEMM consumes.  (Measured for m = 2 against N*N dense planes: input
// transform 1236 vs 1383 us, GEMM and output equal.)
// One thread = one channel of a tile: all accesses are contiguous over channels.
// ------------------------------------------------------------------------------------------------
// t = B^T d
__device__ __forceinline__ void wino_bt(const float (&d)[6], float (&t)[6]) {
  t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
  t[1] = 4.0f * (d[1] + d[2]) - (d[3] + d[4]);
  t[2] = 4.0f * (d[2] - d[1]) + (d[3] - d[4]);
  t[3] = 2.0f * (d[3] - d[1]) + (d[4] - d[2]);
  t[4] = 2.0f * (d[1] - d[3]) + (d[4] - d[2]);
  t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
__device__ __forceinline__ void wino_bt(const float (&d)[8], float (&t)[8]) {
  const float e1 = d[2] + d[6] - 4.25f * d[4], o1 = d[1] + d[5] - 4.25f * d[3];
  const float e2 = 0.25f * d[2] - 1.25f * d[4] + d[6], o2 = 0.5f * d[1] - 2.5f * d[3] + 2.0f * d[5];
  const float e3 = 4.0f * d[2] - 5.0f * d[4] + d[6], o3 = 2.0f * d[1] - 2.5f * d[3] + 0.5f * d[5];
  t[0] = (d[6] - d[0]) + 5.25f * (d[2] - d[4]);
  t[1] = e1 + o1;
  t[2] = e1 - o1;
  t[3] = e2 + o2;
  t[4] = e2 - o2;
  t[5] = e3 + o3;
  t[6] = e3 - o3;
  t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
}
// y = A^T m
__device__ __forceinline__ void wino_at(const float (&m)[6], float (&y)[2]) {
  y[0] = (m[0] + m[1]) + (m[2] + m[3]) + m[4];
  y[1] = (m[1] - m[2]) + 2.0f * (m[3] - m[4]) + m[5];
}
__device__ __forceinline__ void wino_at(const float (&m)[8], float (&y)[4]) {
  const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4], s3 = m[5] + m[6], d3 = m[5] - m[6];
  y[0] = (m[0] + s1) + (s2 + s3);
  y[1] = d1 + 2.0f * d2 + 0.5f * d3;
  y[2] = s1 + 4.0f * s2 + 0.25f * s3;
  y[3] = (d1 + m[7]) + 8.0f * d2 + 0.125f * d3;
}

// t = A d, A the transpose of A^T above (adjoint of the output transform: dM = A dY A^T, used for the filter gradient)
__device__ __forceinline__ void wino_a(const float (&d)[2], float (&t)[6]) {
  t[0] = d[0];
  t[1] = d[0] + d[1];
  t[2] = d[0] - d[1];
  t[3] = d[0] + 2.0f * d[1];
  t[4] = d[0] - 2.0f * d[1];
  t[5] = d[1];
}
__device__ __forceinline__ void wino_a(const float (&d)[4], float (&t)[8]) {
  const float e = d[0] + d[2], o = d[1] + d[3];
  const float e2 = d[0] + 4.0f * d[2], o2 = 2.0f * d[1] + 8.0f * d[3];
  const float e3 = d[0] + 0.25f * d[2], o3 = 0.5f * d[1] + 0.125f * d[3];
  t[0] = d[0];
  t[1] = e + o;
  t[2] = e - o;
  t[3] = e2 + o2;
  t[4] = e2 - o2;
  t[5] = e3 + o3;
  t[6] = e3 - o3;
  t[7] = d[3];
}

#ifdef EQA_WABL_NOLOAD
#define WINO_LD(ptr, k) ((float)(threadIdx.x + (k)))
#else
#define WINO_LD(ptr, k) (*(ptr))
#endif
#ifdef EQA_WABL_NOSTORE
#define WINO_ST(ptr, v) do { if ((v) == 1.2345e-30f) *(ptr) = (v); } while (0)
#elif defined(EQA_WINO_NO_NT)
#define WINO_ST(ptr, v) (*(ptr) = (v))
#else
// V is written once and next read by the GEMM after 8 GB of other traffic: non-temporal stores keep it out of the caches
// (measured 2016 vs 2060 us for the 256-image launch)
#define WINO_ST(ptr, v) __builtin_nontemporal_store((v), (ptr))
#endif
#ifndef EQA_WINO_BLOCKS
#define EQA_WINO_BLOCKS 4096
#endif
// M is read exactly once: streaming (non-temporal) loads, measured 1485 vs 1588 us for output + window sums at 256 images
#ifdef EQA_WINO_NO_NT
#define WINO_LDM(ptr) (*(ptr))
#else
#define WINO_LDM(ptr) __builtin_nontemporal_load(ptr)
#endif
// Optional on-the-fly activation of the INPUT: d = relu(x + in_bias[c]) (the previous layer's folded bias / batch-norm
// and ReLU), which removes a separate pass over the previous feature map.
__device__ __forceinline__ float wino_act(float v, float ib, int in_relu) {
  v += ib;
  return (in_relu && v < 0.0f) ? 0.0f : v;
}

// One block = a strip of consecutive tiles in one tile row x 256 channels (thread = channel).  V = B^T d B is evaluated
// as (B^T d) B: the vertical transform u[:, col] = B^T d[:, col] depends only on the input column, so it is computed
// once per column and the N-column window slides by m per tile: m*N loads per tile instead of N*N.  The next tile's
// columns are requested before the current tile's N*N stores are issued (the memory pipeline is in-order per CU).
// PAD: the logical input is x zero-padded by `pad` pixels on every side (the input gradient of the convolution is a
// forward convolution of the padded output gradient); out-of-range taps read as 0 and are not activated.
template <int N, bool PAD>
__global__ __launch_bounds__(kThreads) void winograd_k5_input_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                                    const float* __restrict__ in_bias, int in_relu,
                                                                    int H, int W, int C, int TY, int TX, int nstrip,
                                                                    int strip_len, size_t nwork, int pad) {
  constexpr int MT = N - 4;
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  // XCD-aware order: block b runs on XCD b % 8, so XCD k is given the contiguous work range [k*nwork/8, (k+1)*nwork/8)
  // (bijective form for nwork % 8 != 0).  Input pixels are shared by overlapping tiles; with neighbouring strips on one
  // XCD the overlap is served by that XCD's L2 instead of being re-fetched from HBM (measured, m = 2, tiles dealt
  // round-robin: 3.6 GB fetched per 0.55 GB of input; with this order 0.55 GB).
  const size_t bid = blockIdx.x;
  const size_t q8 = nwork / kXcd, r8 = nwork % kXcd, xcd = bid % kXcd;
  const size_t work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / kXcd;  // (img*TY + ty)*nstrip + s
  const int s = (int)(work % nstrip);
  const size_t r = work / nstrip;  // img*TY + ty
  const int ty = (int)(r % TY);
  const size_t img = r / TY;
  const int tx0 = s * strip_len;
  const int tx1 = min(TX, tx0 + strip_len);
  // (with PAD the pointer may start before the image; it is only dereferenced for in-range rows / columns)
  const int gy0 = MT * ty - (PAD ? pad : 0), gx0 = MT * tx0 - (PAD ? pad : 0);
  const float* p = x + ((ptrdiff_t)(img * H) + gy0) * (ptrdiff_t)W * C + (ptrdiff_t)gx0 * C + c;
  const float ib = in_bias ? in_bias[c] : 0.0f;
  // PAD: out-of-range taps load the clamped pixel and select 0 afterwards.  (Row and column are the same for the whole
  // block, so `in ? load : 0` compiled to a scalar branch around every single load -- no two loads in flight together:
  // 4.1 ms for the 96x96 padded gradient against 1.8 ms for the unpadded 92x92 forward.)  The clamped row offsets are
  // computed once per block, the clamped column offset once per column: one add per tap.
  const float* ximg = x + (ptrdiff_t)(img * H) * (ptrdiff_t)W * C + c;
  unsigned roff[N];
  bool rin[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int yy = min(max(gy0 + i, 0), H - 1);
    roff[i] = PAD ? (unsigned)yy * (unsigned)W * (unsigned)C : 0u;  // launch_wino_input: H * W * C < 2^31 with PAD
    rin[i] = yy == gy0 + i;
  }
  auto tap = [&](const float* q, int i, int k, int gx_base, int tag) -> float {
    if (PAD) {
      const int xx = min(max(gx_base + k, 0), W - 1);
      const float v = wino_act(WINO_LD(ximg + (roff[i] + (unsigned)xx * (unsigned)C), tag), ib, in_relu);
      return (rin[i] && xx == gx_base + k) ? v : 0.0f;
    }
    return wino_act(WINO_LD(q + ((ptrdiff_t)i * W + k) * C, tag), ib, in_relu);
  };
  float u[N][N];  // u[i][k] = (B^T d)[i][window column k]
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float d[N], t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = tap(p, i, k, gx0, i * N + k);
    wino_bt(d, t);
#pragma unroll
    for (int i = 0; i < N; ++i) u[i][k] = t[i];
  }
  float* vout = V + (r * TX + tx0) * (size_t)(N * N) * C + c;
  for (int tx = tx0; tx < tx1; ++tx) {
    float nx[MT][N];
    const bool more = tx + 1 < tx1;  // uniform
    if (more) {
      const int off = MT * (tx - tx0) + N;
      const float* pn = p + (ptrdiff_t)off * C;
#pragma unroll
      for (int k = 0; k < MT; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) nx[k][i] = tap(pn, i, k, gx0 + off, i * MT + k + tx);  // activated here
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float o[N];
      wino_bt(u[i], o);
#pragma unroll
      for (int j = 0; j < N; ++j) WINO_ST(vout + (size_t)(i * N + j) * C, o[j]);
    }
    vout += (size_t)(N * N) * C;
    if (more) {
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < N - MT; ++k) u[i][k] = u[i][k + MT];
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        float d[N], t[N];
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = nx[k][i];
        wino_bt(d, t);
#pragma unroll
        for (int i = 0; i < N; ++i) u[i][N - MT + k] = t[i];
      }
    }
  }
}

// o = A^T M A (+ bias, ReLU) of one tile and channel; mp -> M[tile][0][c]
template <int N>
__device__ __forceinline__ void wino_tile_out(const float* __restrict__ mp, int C, float b, int relu, float (&o)[N - 4][N - 4]) {
  constexpr int MT = N - 4;
  float sc[MT][N];  // sc[r][j] = (A^T M)[r][j]
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float m[N], y[MT];
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = WINO_LDM(mp + (size_t)(i * N + j) * C);
    wino_at(m, y);
#pragma unroll
    for (int r = 0; r < MT; ++r) sc[r][j] = y[r];
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
    wino_at(sc[r], o[r]);
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      o[r][q] += b;
      if (relu) o[r][q] = fmaxf(o[r][q], 0.0f);
    }
  }
}

template <int N>
__global__ __launch_bounds__(kThreads) void winograd_k5_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                                     int relu, float* __restrict__ y, int OH, int OW, int C,
                                                                     int TY, int TX) {
  constexpr int MT = N - 4;
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t tile = blockIdx.x;
  const int tx = (int)(tile % TX);
  const size_t r = tile / TX;
  const int ty = (int)(r % TY);
  const size_t img = r / TY;
  float o[MT][MT];
  wino_tile_out<N>(M + tile * (size_t)(N * N) * C + c, C, bias ? bias[c] : 0.0f, relu, o);
  float* q = y + ((img * OH + MT * ty) * (size_t)OW + MT * tx) * C + c;
#pragma unroll
  for (int rr = 0; rr < MT; ++rr)
#pragma unroll
    for (int qq = 0; qq < MT; ++qq) q[((size_t)rr * OW + qq) * C] = o[rr][qq];
}

// Adjoint of the output transform (training: filter gradient dU[a] = V[:, a]^T dM[:, a]):  dM = A dY A^T per tile,
// dY:(nimg, OH, OW, C) -> dM:(tiles, N*N, C).  Tiles do not overlap on the output side: m*m loads, N*N stores per thread.
template <int N>
__global__ __launch_bounds__(kThreads) void winograd_k5_output_adjoint_kernel(const float* __restrict__ dY, float* __restrict__ dM,
                                                                             int OH, int OW, int C, int TY, int TX) {
  constexpr int MT = N - 4;
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t tile = blockIdx.x;
  const int tx = (int)(tile % TX);
  const size_t r = tile / TX;
  const int ty = (int)(r % TY);
  const size_t img = r / TY;
  const float* p = dY + ((img * OH + MT * ty) * (size_t)OW + MT * tx) * C + c;
  float w[N][MT];  // w = A d, column by column
#pragma unroll
  for (int q = 0; q < MT; ++q) {
    float d[MT], t[N];
#pragma unroll
    for (int rr = 0; rr < MT; ++rr) d[rr] = p[((size_t)rr * OW + q) * C];
    wino_a(d, t);
#pragma unroll
    for (int i = 0; i < N; ++i) w[i][q] = t[i];
  }
  float* o = dM + tile * (size_t)(N * N) * C + c;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t[N];
    wino_a(w[i], t);
#pragma unroll
    for (int j = 0; j < N; ++j) o[(size_t)(i * N + j) * C] = t[j];
  }
}

// Output transform fused with the window-sum segments of the NEXT (last, linearised) layer: instead of writing the
// (nimg, OH, OW, C) activation and re-reading it, every border row and every interior tile row becomes one "segment" in
// the format of window_sums_nhwc_finalize_kernel -- per channel [row total, first NB columns, last NB columns] -- with the
// segment order that kernel expects: rows 0..NB-1, rows OH-NB..OH-1, then the interior.  One block = one tile row (m
// output rows) of one image x 256 channels, looping over the TX tiles.  NB = k_last - 1 must be a multiple of m.
template <int N, int NB>
__global__ __launch_bounds__(kThreads) void winograd_k5_output_sums_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                                          int relu, float* __restrict__ part, int OH, int OW,
                                                                          int C, int TY, int TX, int nseg) {
  constexpr int MT = N - 4;
  static_assert(NB % MT == 0, "border width must be whole tiles");
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const int ty = blockIdx.x % TY;
  const size_t img = blockIdx.x / TY;
  const float b = bias ? bias[c] : 0.0f;
  constexpr int NV = 1 + 2 * NB;
  float acc[MT][NV];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[r][i] = 0.0f;
  const float* mrow = M + (img * TY + ty) * (size_t)TX * (N * N) * C + c;
  constexpr int HB = NB / MT;  // border tiles per side
  // left border tiles (static column indices), interior, right border tiles
#pragma unroll
  for (int t = 0; t < HB; ++t) {
    float o[MT][MT];
    wino_tile_out<N>(mrow + (size_t)t * (N * N) * C, C, b, relu, o);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int q = 0; q < MT; ++q) { acc[r][0] += o[r][q]; acc[r][1 + MT * t + q] += o[r][q]; }
  }
  for (int tx = HB; tx < TX - HB; ++tx) {
    float o[MT][MT];
    wino_tile_out<N>(mrow + (size_t)tx * (N * N) * C, C, b, relu, o);
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      float a = o[r][0];
#pragma unroll
      for (int q = 1; q < MT; ++q) a += o[r][q];
      acc[r][0] += a;
    }
  }
#pragma unroll
  for (int t = 0; t < HB; ++t) {
    float o[MT][MT];
    wino_tile_out<N>(mrow + (size_t)(TX - HB + t) * (N * N) * C, C, b, relu, o);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int q = 0; q < MT; ++q) { acc[r][0] += o[r][q]; acc[r][1 + NB + MT * t + q] += o[r][q]; }
  }
  if (ty >= HB && ty < TY - HB) {
    // interior tile row: only the sum over its rows is ever needed (the finalize kernel wants the border ROWS one by
    // one, everything else as a total) -- one segment instead of m, a third of the partial buffer at 88 rows
    float* o = part + ((img * nseg + 2 * NB + (ty - HB)) * (size_t)C + c) * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float a = acc[0][i];
#pragma unroll
      for (int r = 1; r < MT; ++r) a += acc[r][i];
      o[i] = a;
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
    const int y = MT * ty + r;
    const int seg = y < NB ? y : NB + (y - (OH - NB));
    float* o = part + ((img * nseg + seg) * (size_t)C + c) * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = acc[r][i];
  }
}

template <int N>
int launch_wino_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                      void* stream, int pad = 0) {
  constexpr int MT = N - 4;
  const int Hl = H + 2 * pad, Wl = W + 2 * pad;  // logical (zero-padded) input
  if (!x || !V || nimg < 0 || pad < 0 || H <= 0 || W <= 0 || Hl < N || Wl < N || C <= 0) return EQA_ERR_INVALID_ARG;
  if (((Hl - 4) % MT) || ((Wl - 4) % MT)) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int TY = (Hl - 4) / MT, TX = (Wl - 4) / MT;
  // strips: enough blocks to fill 256 CUs x their resident blocks, long enough that the N*N-load prologue is amortised
  // (m*N further loads per tile); measured for m = 2 at 64 x 92 x 92 x 256: whole rows 1112 us, 8-tile strips 1289 us
  const int cb = (C + kThreads - 1) / kThreads;
  const size_t rows = (size_t)nimg * TY;
  int nstrip = (int)((EQA_WINO_BLOCKS + rows * cb - 1) / (rows * cb));
  nstrip = std::max(1, std::min(nstrip, std::max(1, TX / 4)));
  const int strip_len = (TX + nstrip - 1) / nstrip;
  nstrip = (TX + strip_len - 1) / strip_len;
  const size_t nwork = rows * nstrip;
  if (nwork > 0x7fffffffULL || rows * TX > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  if (pad && (size_t)H * W * C > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  if (pad)
    hipLaunchKernelGGL((winograd_k5_input_kernel<N, true>), dim3((unsigned)nwork, cb), dim3(kThreads), 0, (hipStream_t)stream,
                       x, V, in_bias, in_relu, H, W, C, TY, TX, nstrip, strip_len, nwork, pad);
  else
    hipLaunchKernelGGL((winograd_k5_input_kernel<N, false>), dim3((unsigned)nwork, cb), dim3(kThreads), 0, (hipStream_t)stream,
                       x, V, in_bias, in_relu, H, W, C, TY, TX, nstrip, strip_len, nwork, 0);
  return launch_status();
}

template <int N>
int launch_wino_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                       void* stream) {
  constexpr int MT = N - 4;
  if (!M || !y || nimg < 0 || OH < MT || OW < MT || C <= 0) return EQA_ERR_INVALID_ARG;
  if ((OH % MT) || (OW % MT)) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int TY = OH / MT, TX = OW / MT;
  const size_t tiles = (size_t)nimg * TY * TX;
  if (tiles > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((winograd_k5_output_kernel<N>), dim3((unsigned)tiles, (C + kThreads - 1) / kThreads), dim3(kThreads),
                     0, (hipStream_t)stream, M, bias, relu, y, OH, OW, C, TY, TX);
  return launch_status();
}

template <int N>
int launch_wino_output_adjoint(const float* dY, float* dM, int nimg, int OH, int OW, int C, void* stream) {
  constexpr int MT = N - 4;
  if (!dY || !dM || nimg < 0 || OH < MT || OW < MT || C <= 0) return EQA_ERR_INVALID_ARG;
  if ((OH % MT) || (OW % MT)) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int TY = OH / MT, TX = OW / MT;
  const size_t tiles = (size_t)nimg * TY * TX;
  if (tiles > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((winograd_k5_output_adjoint_kernel<N>), dim3((unsigned)tiles, (C + kThreads - 1) / kThreads),
                     dim3(kThreads), 0, (hipStream_t)stream, dY, dM, OH, OW, C, TY, TX);
  return launch_status();
}

template <int N>
int launch_wino_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg, int OH,
                            int OW, int C, int k_next, void* stream) {
  constexpr int MT = N - 4;
  if (!M || !S || !workspace || nimg < 0 || OH < MT || OW < MT || C <= 0 || k_next <= 0) return EQA_ERR_INVALID_ARG;
  const int nb = k_next - 1;
  // border width in whole tiles, disjoint borders with an interior, same limits as eqa_window_sums_nhwc
  if ((OH % MT) || (OW % MT) || (nb != 4 && nb != 2) || (nb % MT) || OH < 2 * nb + MT || OW < 2 * nb + MT || k_next > kMaxWinK)
    return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  if (nimg > 65535) return EQA_ERR_UNSUPPORTED;
  const int TY = OH / MT, TX = OW / MT;
  if ((size_t)nimg * TY > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((size_t)nimg * TY), (C + kThreads - 1) / kThreads);
  const int nseg = 2 * nb + (TY - 2 * (nb / MT));  // the 2 nb border rows one by one + one segment per interior tile row
  float* part = (float*)workspace;
  if (nb == 4) {
    hipLaunchKernelGGL((winograd_k5_output_sums_kernel<N, 4>), grid, dim3(kThreads), 0, st, M, bias, relu, part, OH, OW, C, TY, TX, nseg);
  } else {
    if constexpr (MT == 2)
      hipLaunchKernelGGL((winograd_k5_output_sums_kernel<N, 2>), grid, dim3(kThreads), 0, st, M, bias, relu, part, OH, OW, C, TY, TX, nseg);
  }
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  return launch_window_sums_nhwc_finalize((const float*)workspace, S, nimg, C, k_next, nseg, st);
}

}  // namespace

extern "C" {

int eqa_winograd_f2k5_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            void* stream) {
  return launch_wino_input<6>(x, V, in_bias, in_relu, nimg, H, W, C, stream);
}

int eqa_winograd_f4k5_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            void* stream) {
  return launch_wino_input<8>(x, V, in_bias, in_relu, nimg, H, W, C, stream);
}

int eqa_winograd_f2k5_input_padded(const float* x, float* V, int nimg, int H, int W, int C, int pad, void* stream) {
  return launch_wino_input<6>(x, V, nullptr, 0, nimg, H, W, C, stream, pad);
}
int eqa_winograd_f4k5_input_padded(const float* x, float* V, int nimg, int H, int W, int C, int pad, void* stream) {
  return launch_wino_input<8>(x, V, nullptr, 0, nimg, H, W, C, stream, pad);
}

int eqa_winograd_f2k5_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                             void* stream) {
  return launch_wino_output<6>(M, bias, relu, y, nimg, OH, OW, C, stream);
}

int eqa_winograd_f4k5_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                             void* stream) {
  return launch_wino_output<8>(M, bias, relu, y, nimg, OH, OW, C, stream);
}

int eqa_winograd_f2k5_output_adjoint(const float* dY, float* dM, int nimg, int OH, int OW, int C, void* stream) {
  return launch_wino_output_adjoint<6>(dY, dM, nimg, OH, OW, C, stream);
}
int eqa_winograd_f4k5_output_adjoint(const float* dY, float* dM, int nimg, int OH, int OW, int C, void* stream) {
  return launch_wino_output_adjoint<8>(dY, dM, nimg, OH, OW, C, stream);
}

int64_t eqa_winograd_f2k5_output_sums_workspace_bytes(int nimg, int OH, int C, int k_next) {
  if (nimg <= 0 || OH <= 0 || C <= 0 || k_next <= 0) return 0;
  return (int64_t)nimg * OH * C * (1 + 2 * (k_next - 1)) * (int64_t)sizeof(float);
}

int eqa_winograd_f2k5_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg,
                                  int OH, int OW, int C, int k_next, void* stream) {
  return launch_wino_output_sums<6>(M, bias, relu, S, workspace, nimg, OH, OW, C, k_next, stream);
}

int eqa_winograd_f4k5_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg,
                                  int OH, int OW, int C, int k_next, void* stream) {
  return launch_wino_output_sums<8>(M, bias, relu, S, workspace, nimg, OH, OW, C, k_next, stream);
}

}  // extern "C"
