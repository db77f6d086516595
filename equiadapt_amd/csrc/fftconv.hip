// libeqa_hip.so, part 8 -- 5x5 stride-1 group convolutions in inference as an overlap-save FFT convolution (I2a).
// C ABI: include/eqa_hip.h.  Design notes: HISTORY.md section 3.4.
//
// Winograd F(4x4,5x5) needs 4 multiplies per output and a 4x expansion of the activations (V, M: 8.1 GB each at the
// headline shape); its library GEMM runs at the clock-limited fp32 roofline, so only fewer multiplies help.  A 48x48
// real FFT tile gives 44x44 outputs -- two tiles cover the 88 output rows / columns exactly -- with 48 x 25 complex
// frequencies: 4 real multiplies per complex one make 2.48 per output, and the spectra are only 1.24x the activations.
// Per frequency f = (ky, kx) the channel contraction is a complex matrix product, done by the GEMM library as a REAL one
// through [Ar | Ai] . [[Br, Bi], [-Bi, Br]] = [Cr | Ci]  (A: tiles x 2 Cin, B: 2 Cin x 2 Cout, batched over the stored frequencies).
// conv2d is a cross-correlation: B holds conj(FFT(filter)) / 48^2, computed once per weight version on the host side.
// fp32 throughout; error vs an fp64 convolution 2-4e-7 of max|y| (Winograd F(4,5): 9e-6).
//
// Default: the two FUSED kernels further down (row pass, LDS, column pass in one block).  Their two-pass ancestors stay as
// the path for channel counts that are not a multiple of 16 and as the ablation (EQA_FFT_TWO_PASS=1):
// four streaming kernels, one thread per channel (channels-last: every load and store instruction of a wave is one
// contiguous run of channels), one 48-point transform per thread held in registers (fft48.inc, generated, 819 flops):
//   rows_fwd   x (nimg,H,W,C) -> T (nimg,H,TX,25,2,C): real rows of 48 pixels (tile columns 44 tx .. 44 tx + 47, zero
//              beyond W), previous layer's bias / ReLU applied while loading
//   cols_fwd   T -> V (F, M, 2C), F = 1154 stored frequencies (below): columns of 48 rows (44 ty .. 44 ty + 47, zero beyond H), M = nimg*TY*TX tiles
//   [ batched GEMM by the caller: Mo[f] = V[f] . B[f] ]
//   cols_inv   Mo (F, M, 2C) -> T2 (nimg,OH,TX,25,2,C): inverse over ky, rows 44..47 of a tile (circular wrap) dropped
//   rows_inv   T2 -> y (nimg,OH,OW,C) = [relu](. + bias), or -> the window-sum segments of the next (last, linearised)
//              layer in the format of window_sums_nhwc_finalize_kernel (one segment per output row)
#include <cstdlib>

#include "eqa_common.hpp"

namespace {

#include "fft_common.inc"

// Mo in HBM: complex numbers, re and im interleaved, read once with 8-byte non-temporal loads
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fft_ldg2(const float2* p, float& re, float& im) {
  const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
  re = v[0];
  im = v[1];
}

// One column of Mo (48 ky of one kx, tile and channel) for the inverse transform.  Edge columns store rows 0..24 only; their
// rows 25..47 are the conjugates of rows 23..1.  Written so that the mirror costs no per-row select: rows >= 25 are addressed as
// base2 + (ky - 25) * step2 with per-thread constants (forward or backward through the stored rows) and their imaginary parts
// get a per-thread sign mask.
__device__ __forceinline__ void fft_load_column(const float2* __restrict__ p, size_t fpitch, bool edge, float (&re)[kFftN],
                                                float (&im)[kFftN]) {
#pragma unroll
  for (int ky = 0; ky < kFftH; ++ky) fft_ldg2(p + ky * fpitch, re[ky], im[ky]);
  const float2* base2 = edge ? p + (size_t)(kFftH - 2) * fpitch : p + (size_t)kFftH * fpitch;
  const ptrdiff_t step2 = edge ? -(ptrdiff_t)fpitch : (ptrdiff_t)fpitch;
  const unsigned flip = edge ? 0x80000000u : 0u;
#pragma unroll
  for (int ky = kFftH; ky < kFftN; ++ky) {
    fft_ldg2(base2 + (ky - kFftH) * step2, re[ky], im[ky]);
    im[ky] = __uint_as_float(__float_as_uint(im[ky]) ^ flip);
  }
}

// (The two-pass kernels below take the tile's output size O = 49 - kernel size as a template argument: 44 for the 5 x 5 layers
// -- where they are the fallback of the fused kernels further down -- and 46 / 42 / 40 for 3 x 3 / 7 x 7 / 9 x 9, which run
// through them only: eqa_fft48_* entry points, round 4.)
template <int O>
__global__ __launch_bounds__(kThreads) void fft48_rows_fwd_kernel(const float* __restrict__ x, float* __restrict__ T,
                                                                 const float* __restrict__ in_bias, int in_relu, int H, int W,
                                                                 int C, int TX, int win) {
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const int xt = blockIdx.x % TX;
  const size_t row = blockIdx.x / TX;  // img * H + y
  const float ib = in_bias ? in_bias[c] : 0.0f;
  const float* p = x + (row * W + (size_t)O * xt) * C + c;
  const int nvalid = min(win, W - O * xt);  // uniform; win = 48: activation tiles (overlap 4), 44: gradient tiles (disjoint)
  float re[kFftN], ore[kFftH], oim[kFftH];
#pragma unroll
  for (int j = 0; j < kFftN; ++j) {
    // columns beyond the image: the clamped pixel is loaded and replaced by 0 (a uniform `j < nvalid ? load : 0` would
    // become a scalar branch around every load)
    float v = p[(size_t)min(j, nvalid - 1) * C] + ib;
    v = in_relu ? fmaxf(v, 0.0f) : v;
    re[j] = j < nvalid ? v : 0.0f;
  }
  fft48_r2c(re, ore, oim);  // half spectrum of a real row: 488 operations (the complex transform with a zero imaginary part: ~610)
  float* o = T + ((row * TX + xt) * kFftH) * 2 * (size_t)C + c;
#pragma unroll
  for (int k = 0; k < kFftH; ++k) {
    o[(size_t)(2 * k) * C] = ore[k];
    o[(size_t)(2 * k + 1) * C] = oim[k];
  }
}

template <int O>
__global__ __launch_bounds__(kThreads) void fft48_cols_fwd_kernel(const float* __restrict__ T, float* __restrict__ V, int H, int C,
                                                                 int TY, int TX, size_t M, size_t m0, int G, int win) {
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const int kx = blockIdx.x % kFftH;
  const size_t m = blockIdx.x / kFftH;  // (img * TY + ty) * TX + tx, img counted inside this chunk of images
  const int tx = (int)(m % TX);
  const int ty = (int)((m / TX) % TY);
  const size_t img = m / ((size_t)TX * TY);
  const int y0 = O * ty;
  const int nvalid = min(win, H - y0);  // uniform
  const size_t pitch = (size_t)TX * kFftH * 2 * C;  // one image row of T
  const float* p = T + ((img * H + y0) * TX + tx) * (size_t)kFftH * 2 * C + (size_t)(2 * kx) * C + c;
  float re[kFftN], im[kFftN], ore[kFftN], oim[kFftN];
#pragma unroll
  for (int i = 0; i < kFftN; ++i) {
    const size_t off = (size_t)min(i, nvalid - 1) * pitch;
    const float a = p[off], b = p[off + C];
    re[i] = i < nvalid ? a : 0.0f;
    im[i] = i < nvalid ? b : 0.0f;
  }
  fft48(re, im, ore, oim);
  // rows of V: [Re x G | Im x G] per group of G channels (G = 16: what a block of the fused kernel owns; G = 1: interleaved)
  float* o = V + ((size_t)fft_f0(kx) * M + m0 + m) * 2 * (size_t)C + (c / G) * 2 * G + c % G;
  const size_t fpitch = (size_t)fft_fstep(kx) * M * 2 * C;  // from ky to ky + 1
  const int nky = fft_nky(kx);
#pragma unroll
  for (int ky = 0; ky < kFftN; ++ky) {
    if (ky < nky) {
      __builtin_nontemporal_store(ore[ky], o + ky * fpitch);
      __builtin_nontemporal_store(oim[ky], o + ky * fpitch + G);
    }
  }
}

// FULL (overlap-add, the input gradient): all 48 rows are results, stored tile by tile: T2 (nimg, TY, 48, TX, 25, 2, C)
template <bool FULL, int O>
__global__ __launch_bounds__(kThreads) void fft48_cols_inv_kernel(const float* __restrict__ Mo, float* __restrict__ T2, int OH, int C,
                                                                 int TY, int TX, size_t M, size_t m0) {
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const int kx = blockIdx.x % kFftH;
  const size_t m = blockIdx.x / kFftH;
  const int tx = (int)(m % TX);
  const int ty = (int)((m / TX) % TY);
  const size_t img = m / ((size_t)TX * TY);
  const float2* p = reinterpret_cast<const float2*>(Mo) + ((size_t)fft_f0(kx) * M + m0 + m) * (size_t)C + c;
  const size_t fpitch = (size_t)fft_fstep(kx) * M * C;
  float re[kFftN], im[kFftN], ore[kFftN], oim[kFftN];
  fft_load_column(p, fpitch, fft_edge(kx), re, im);
  fft48(im, re, oim, ore);  // inverse: real and imaginary parts swapped in and out (1 / 48^2 is in the filter spectra)
  const size_t pitch = (size_t)TX * kFftH * 2 * C;
  if (FULL) {
    float* o = T2 + (((img * TY + ty) * kFftN) * TX + tx) * (size_t)kFftH * 2 * C + (size_t)(2 * kx) * C + c;
#pragma unroll
    for (int i = 0; i < kFftN; ++i) {
      o[i * pitch] = ore[i];
      o[i * pitch + C] = oim[i];
    }
    return;
  }
  const int y0 = O * ty;
  const int nrows = min(O, OH - y0);  // uniform; rows 44..47 of the tile are the circular wrap-around
  float* o = T2 + ((img * OH + y0) * TX + tx) * (size_t)kFftH * 2 * C + (size_t)(2 * kx) * C + c;
#pragma unroll
  for (int i = 0; i < O; ++i) {
    if (i < nrows) {
      o[i * pitch] = ore[i];
      o[i * pitch + C] = oim[i];
    }
  }
}

// Input gradient (training), row pass with overlap-add.  A 44 x 44 output-gradient tile convolved with the 5 x 5 filters is a
// 48 x 48 block of the input gradient (44 + 5 - 1: the circular transform wraps nothing); blocks of neighbouring tiles overlap
// by 4.  Thread = (image, input row y, channel): the row spectra of the (at most two) tile rows that reach y are added before
// the row transform (it is linear), the 4 overlapping columns of consecutive tile columns are carried in registers: every
// input-gradient element is written once, in a fixed order.
template <int O>
__global__ __launch_bounds__(kThreads) void fft48_rows_inv_add_kernel(const float* __restrict__ T2, float* __restrict__ dx, int H, int W,
                                                                     int C, int TY, int TX) {
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t row = blockIdx.x;  // img * H + y
  const int y = (int)(row % H);
  const size_t img = row / H;
  const int t1 = min(y / O, TY - 1), i1 = y - O * t1;  // i1 <= 47 because H <= 44 TY + 4
  const bool two = i1 < kFftN - O && t1 > 0;               // rows 44..47 of the tile row above reach y as well
  const size_t pitch = (size_t)TX * kFftH * 2 * C;
  const float* p1 = T2 + ((img * TY + t1) * kFftN + i1) * pitch + c;
  const float* p0 = two ? T2 + ((img * TY + t1 - 1) * kFftN + i1 + O) * pitch + c : p1;
  float carry[kFftN - O];
#pragma unroll
  for (int j = 0; j < kFftN - O; ++j) carry[j] = 0.0f;
  float* o = dx + row * (size_t)W * C + c;
  for (int tx = 0; tx < TX; ++tx) {
    float re[kFftH], im[kFftH], ore[kFftN];
#pragma unroll
    for (int k = 0; k < kFftH; ++k) {
      const size_t off = ((size_t)tx * kFftH + k) * 2 * C;
      const float a = p1[off], b = p1[off + C], a0 = p0[off], b0 = p0[off + C];
      re[k] = two ? a + a0 : a;
      im[k] = two ? b + b0 : b;
    }
    ifft48_c2r(re, im, ore);  // real output from the stored half of the spectrum (468 operations; the complex transform: 819)
    const int x0 = O * tx;
#pragma unroll
    for (int j = 0; j < kFftN - O; ++j) ore[j] += carry[j];
    const bool last = tx == TX - 1;  // uniform
#pragma unroll
    for (int j = 0; j < kFftN; ++j) {
      if ((j < O || last) && x0 + j < W) o[(size_t)(x0 + j) * C] = ore[j];
    }
#pragma unroll
    for (int j = 0; j < kFftN - O; ++j) carry[j] = ore[O + j];
  }
}

// NB = k_next - 1 border columns on each side are needed one by one for the window sums; 0: plain output
template <int NB, int O>
__global__ __launch_bounds__(kThreads) void fft48_rows_inv_kernel(const float* __restrict__ T2, const float* __restrict__ bias, int relu,
                                                                 float* __restrict__ out, int OH, int OW, int C, int TX,
                                                                 size_t img0) {
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t row = blockIdx.x;  // img * OH + y inside this chunk of images (T2 is chunk-local, `out` is not)
  const int y = (int)(row % OH);
  const size_t img = img0 + row / OH;
  const float b = bias ? bias[c] : 0.0f;
  constexpr int NV = 1 + 2 * NB;
  float acc[NV > 1 ? NV : 1];
#pragma unroll
  for (int i = 0; i < (NV > 1 ? NV : 1); ++i) acc[i] = 0.0f;
  for (int tx = 0; tx < TX; ++tx) {
    const float* p = T2 + ((row * TX + tx) * kFftH) * 2 * (size_t)C + c;
    float re[kFftH], im[kFftH], ore[kFftN];
#pragma unroll
    for (int k = 0; k < kFftH; ++k) {
      re[k] = p[(size_t)(2 * k) * C];
      im[k] = p[(size_t)(2 * k + 1) * C];
    }
    ifft48_c2r(re, im, ore);  // real output from the stored half of the spectrum
    const int x0 = O * tx;
    const int ncols = min(O, OW - x0);  // uniform
    if (NB == 0) {
      float* o = out + ((img * OH + y) * OW + x0) * (size_t)C + c;
#pragma unroll
      for (int j = 0; j < O; ++j) {
        if (j < ncols) {
          const float v = ore[j] + b;
          o[(size_t)j * C] = relu ? fmaxf(v, 0.0f) : v;
        }
      }
    } else {
      float tot = 0.0f;
#pragma unroll
      for (int j = 0; j < O; ++j) {
        float v = ore[j] + b;
        v = relu ? fmaxf(v, 0.0f) : v;
        v = j < ncols ? v : 0.0f;
        tot += v;
        const int xx = x0 + j;  // uniform
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          if (xx == q) acc[1 + q] += v;
          if (xx == OW - NB + q) acc[1 + NB + q] += v;
        }
      }
      acc[0] += tot;
    }
  }
  if (NB > 0) {
    // one segment per output row, in the order window_sums_nhwc_finalize_kernel expects: rows 0..NB-1, OH-NB..OH-1, interior
    const int seg = y < NB ? y : (y >= OH - NB ? NB + (y - (OH - NB)) : 2 * NB + (y - NB));
    float* o = out + ((img * OH + seg) * (size_t)C + c) * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = acc[i];
  }
}

// Row and column pass of the forward transform in ONE kernel: a block owns one 48 x 48 tile x 16 channels, the row
// spectra (48 x 25 complex x 16 channels = 154 KB) go through LDS instead of through a 2.4 GB intermediate in HBM that is
// written and read straight back (the two-pass form moves 9.5 GB and runs at the copy rate: 1.75 ms; this one moves 4.7).
// 768 threads: phase 1, thread (y, c) transforms row y; phase 2, thread (kx, c), 400 of them, transforms column kx.
// A wave reads 4 pixels x 16 channels = 4 runs of 64 bytes; the 16 channel groups of a tile are dealt to the same XCD so
// that the other halves of the 128-byte lines come out of its L2.
#ifdef EQA_FFT_CLOCK
// Debug build: shader-cycle stamps of one block of the fused inverse (thread 0, a column-role thread): [0] loads issued and
// returned, [1] column transform + LDS writes, [2] wait at the barrier, [3] row transform + epilogue, [4] window-sum pieces.
__device__ unsigned long long g_fft_clock[32];  // [0..4] inverse, [8..12] forward, [16..21] pipeline producer, [24..29] consumer
#define FFT_CLOCK_BEGIN() const bool clk_on = blockIdx.x == gridDim.x / 2 && threadIdx.x == 0; unsigned long long clk_t = __builtin_readcyclecounter()
#define FFT_CLOCK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if (clk_on) g_fft_clock[i] = n_ - clk_t; clk_t = n_; } while (0)
#define FFT_CLOCK_LOADS() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FFT_CLOCK(0); } while (0)
#define FFT_CLOCK_USE(v, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(v) : "memory"); FFT_CLOCK(i); } while (0)
// pipeline kernel: sums over the items of the middle block (thread 0 of the producers / of the consumers)
#define PIPE_CLOCK_BEGIN(tid0) const bool pclk_on = blockIdx.x == gridDim.x / 2 && threadIdx.x == (tid0); unsigned long long pclk_t = __builtin_readcyclecounter(); unsigned long long pclk_s[6] = {0, 0, 0, 0, 0, 0}
#define PIPE_CLOCK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pclk_s[i] += n_ - pclk_t; pclk_t = n_; } while (0)
#define PIPE_CLOCK_WAIT(i) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PIPE_CLOCK(i); } while (0)
#define PIPE_CLOCK_END(base) do { if (pclk_on) for (int i_ = 0; i_ < 6; ++i_) g_fft_clock[(base) + i_] = pclk_s[i_]; } while (0)
#else
#define FFT_CLOCK_BEGIN() do { } while (0)
#define FFT_CLOCK(i) do { } while (0)
#define FFT_CLOCK_LOADS() do { } while (0)
#define FFT_CLOCK_USE(v, i) do { } while (0)
#define PIPE_CLOCK_BEGIN(tid0) do { } while (0)
#define PIPE_CLOCK(i) do { } while (0)
#define PIPE_CLOCK_WAIT(i) do { } while (0)
#define PIPE_CLOCK_END(base) do { } while (0)
#endif
constexpr int kFusCh = 16;
constexpr int kFusThreads = kFftN * kFusCh;                       // 768
constexpr int kFusKxPitch = kFftN * 2 * kFusCh + kFusCh;          // floats per kx slab (+16: slabs start 16 banks apart)
constexpr int kFusLds = kFftH * kFusKxPitch;                      // 38,800 floats = 155,200 bytes

template <int O = kFftO>   // outputs per tile = the tile stride (44: 5 x 5 filters; 46 / 42 / 40: eqa_fft48_* for 3 / 7 / 9)
__global__ __launch_bounds__(kFusThreads) void fft48_fwd_fused_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                                      const float* __restrict__ in_bias, int in_relu, int H, int W,
                                                                      int C, int TY, int TX, size_t M, unsigned nwork, int win,
                                                                      unsigned x_bytes, unsigned v_bytes, int x_grouped) {
  extern __shared__ float lds[];
  FFT_CLOCK_BEGIN();  // forward stamps: [8] loads, [9] row transform + LDS writes, [10] barrier, [11] LDS reads + column transform, [12] stores issued
  // XCD-aware order: consecutive work items (the channel groups of one tile) on one XCD
  const unsigned bid = blockIdx.x;
  const unsigned q8 = nwork / kXcd, r8 = nwork % kXcd, xcd = bid % kXcd;
  const unsigned work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / kXcd;
  const int ngrp = C / kFusCh;
  const int grp = work % ngrp;
  const size_t m = work / ngrp;  // (img * TY + ty) * TX + tx
  const int tx = (int)(m % TX);
  const int ty = (int)((m / TX) % TY);
  const size_t img = m / ((size_t)TX * TY);
  const int cl = threadIdx.x % kFusCh;
  const int c = grp * kFusCh + cl;
  {
    const int y = threadIdx.x / kFusCh;  // 0..47
    const int gy = O * ty + y;
    const int nvalid = min(win, W - O * tx);  // uniform
    const float ib = in_bias ? in_bias[c] : 0.0f;
    // x_grouped: the map is stored (img, channel group of 16, y, x, 16) -- a tile row of this block's group is one 3 KB run, and
    // the two 64-byte halves of a cache line are asked for by consecutive loads of the same lanes.  Channels-last, the other half
    // of every line belongs to the neighbouring channel group, i.e. to another block: every line was requested twice
    // (TCP_TCC_READ_REQ 37.7 M for 18.4 M lines, profiles/r02/pmc_memory_path.md).
    const float* p = x_grouped ? x + ((((img * ngrp + grp) * H + min(gy, H - 1)) * W + (size_t)O * tx) * kFusCh) + cl
                               : x + ((img * H + min(gy, H - 1)) * W + (size_t)O * tx) * C + c;
    const int xstep = x_grouped ? kFusCh : C;
    const bool row_in = gy < H && y < win;
    float re[kFftN], ore[kFftH], oim[kFftH];
    if (nvalid == kFftN && !in_bias && !in_relu) {  // uniform: a full-width tile of a plain map (the headline case) needs no per-pixel work
      if (x_bytes) {
        // the 48 pixels of a row are C floats apart for every lane: buffer loads with that step as a scalar offset, no vector
        // address arithmetic (x_bytes = 0: the map does not fit a 32-bit offset)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
        const unsigned voff = (unsigned)((size_t)(p - x) * 4);
#pragma unroll
        for (int j = 0; j < kFftN; ++j) re[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, voff, (unsigned)(j * xstep * 4), 0));
      } else {
#pragma unroll
        for (int j = 0; j < kFftN; ++j) re[j] = p[(size_t)j * xstep];
      }
      if (!row_in) {
#pragma unroll
        for (int j = 0; j < kFftN; ++j) re[j] = 0.0f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < kFftN; ++j) {
        float v = p[(size_t)min(j, nvalid - 1) * xstep] + ib;
        v = in_relu ? fmaxf(v, 0.0f) : v;
        re[j] = (row_in && j < nvalid) ? v : 0.0f;
      }
    }
    FFT_CLOCK_USE(re[0], 8);
    fft48_r2c(re, ore, oim);
    float* o = lds + (y * 2) * kFusCh + cl;
#pragma unroll
    for (int k = 0; k < kFftH; ++k) {
      o[k * kFusKxPitch] = ore[k];
      o[k * kFusKxPitch + kFusCh] = oim[k];
    }
  }
  FFT_CLOCK(9);
  __syncthreads();
  FFT_CLOCK(10);
  if (threadIdx.x < kFftH * kFusCh) {
    const int kx = threadIdx.x / kFusCh;
    const float* q = lds + kx * kFusKxPitch + cl;
    float re[kFftN], im[kFftN], ore[kFftN], oim[kFftN];
#pragma unroll
    for (int i = 0; i < kFftN; ++i) {
      re[i] = q[(i * 2) * kFusCh];
      im[i] = q[(i * 2 + 1) * kFusCh];
    }
    fft48(re, im, ore, oim);
    // [Re x 16 | Im x 16] per channel group: the block's 128 bytes of a (frequency, tile) row are one run.  (Measured against
    // interleaved complex with 8-byte stores: 1.18 vs 1.42 ms; the inverse prefers 8-byte loads of interleaved complex.)
    float* o = V + ((size_t)fft_f0(kx) * M + m) * 2 * (size_t)C + grp * 2 * kFusCh + cl;
    const size_t fpitch = (size_t)fft_fstep(kx) * M * 2 * C;
    const int nky = fft_nky(kx);
    FFT_CLOCK_USE(ore[0], 11);
    if (v_bytes && !__any(fft_edge(kx))) {
      // a wave without edge columns: every lane steps by the same 23 * M * 2C floats from one ky to the next -> buffer stores
      // with that step as a scalar offset (non-temporal: aux 2)
      const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(V, 0, v_bytes, 0x00020000);
      const unsigned voff = (unsigned)((size_t)(o - V) * 4);
      const unsigned step = (unsigned)((size_t)kFftInner * M * 2 * C * 4);      // (from constants: provably wave-uniform)
#pragma unroll
      for (int ky = 0; ky < kFftN; ++ky) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ore[ky]), vr, voff, ky * step, 2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oim[ky]), vr, voff + kFusCh * 4, ky * step, 2);
      }
    } else {
#pragma unroll
      for (int ky = 0; ky < kFftN; ++ky) {
        if (ky < nky) {
          __builtin_nontemporal_store(ore[ky], o + ky * fpitch);
          __builtin_nontemporal_store(oim[ky], o + ky * fpitch + kFusCh);
        }
      }
    }
  }
  FFT_CLOCK(12);
}

// The fused forward transform as a persistent producer / consumer pipeline (round 3; the mirror image of fft48_inv_pipe_kernel
// further down).  Counters on the one-block-per-item kernel above (profiles/r03/pmc_memory_path.md): 52 cache lines per CU in flight,
// but each waits 1,538 cycles, twice a streaming kernel's latency -- the kernel's 2.5 GB of reads queue behind its own 2.5 GB of
// writes, which a block issues in one burst at the end of its life while no other block of the CU is loading.  Here the block's
// waves are specialised and walk the work items together:
//   row waves (6, thread (row ry, channel): rows ry and ry + 24):  (wait for the 96 loads of item i) two real-input transforms ->
//        LDS | barrier A | issue the 96 loads of item i+1 | barrier B
//   column waves (6, thread (column, channel)):  barrier A | read the column from LDS | barrier B | transform, 96 stores
// so an item's stores and the next item's loads are in the memory system at the same time, and the column waves' arithmetic
// overlaps the row waves'.  24 x 16 column tasks for 25 columns: the two purely real columns kx = 0 and kx = 24 (spectra of real
// rows at the DC and Nyquist bins) ride in ONE complex transform, z = c0 + i c24, and are separated afterwards:
//   C0[k] = (Z[k] + conj(Z[-k])) / 2,   C24[k] = (Z[k] - conj(Z[-k])) / 2i        (k = 0..24: the rows the edge columns store).
// 12 waves = 3 per SIMD, 162 VGPRs, no scratch.  Only for full-width tiles of a plain channel-group-major map (the headline case).
// MEASURED (B = 256, inside the step): 1.066 ms against the one-block-per-item kernel's 0.957 -- correct (same tests) but SLOWER,
// so it is opt-in (EQA_FFT_FWD_PIPE=1) and the kernel above stays the default.  Why: an item's stores and the next item's loads
// do overlap now, but with ONE LDS buffer the chain load(i+1) -> row transforms -> A -> column read -> B is serial, and during
// the row transforms (6 waves: ~7 k cycles) plus the column read nothing new is requested: ~35 k cycles per item against 31 k.
// Hiding that needs the NEXT item's rows in flight while this item's are transformed: 2 x 96 loaded registers per row thread,
// which 168 do not hold next to the transform (the inverse pipeline holds ONE incoming column per producer).  Also measured:
// issuing the loads after B instead of before (1.08 ms), priority for the row waves (1.16 ms).
constexpr int kFwdRowT = 6 * 64, kFwdColT = 6 * 64, kFwdPipeThreads = kFwdRowT + kFwdColT;

__global__ __launch_bounds__(kFwdPipeThreads) void fft48_fwd_pipe_kernel(const float* __restrict__ x, float* __restrict__ V, int H, int W,
                                                                        int C, int TY, int TX, size_t M, unsigned nwork, int win,
                                                                        unsigned x_bytes, unsigned v_bytes) {
  extern __shared__ float lds[];
  const unsigned nblk = gridDim.x;               // a multiple of the XCD count (or < 8): virtual block v runs on XCD v % 8
  const int ngrp = C / kFusCh;
  const unsigned q8 = nwork / kXcd, r8 = nwork % kXcd;
  auto work_of = [&](unsigned v) {
    const unsigned xcd = v % kXcd;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + v / kXcd;
  };
  if (threadIdx.x < kFwdRowT) {
    // ---------------------------------------------------------------- row waves
    const int ry = threadIdx.x / kFusCh, cl = threadIdx.x % kFusCh;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
    float r0[kFftN], r1[kFftN];
    // item -> byte offsets of this thread's two rows (out of range = zero row: rows beyond the map or beyond the tile's window;
    // 0xffffe000 + the largest scalar offset does not wrap, and the launcher keeps the map below it)
#define EQA_FWD_ISSUE(v_)                                                                                                     \
  do {                                                                                                                        \
    const unsigned work_ = work_of(v_);                                                                                       \
    const unsigned grp_ = work_ % ngrp, m_ = work_ / ngrp;                                                                    \
    const unsigned tx_ = m_ % TX, ty_ = (m_ / TX) % TY, img_ = m_ / (TX * TY);                                                \
    const int gy0_ = kFftO * (int)ty_ + ry, gy1_ = gy0_ + 24;                                                                 \
    const bool in0_ = (v_) < nwork && gy0_ < H && ry < win, in1_ = (v_) < nwork && gy1_ < H && ry + 24 < win;                 \
    const size_t plane_ = ((size_t)img_ * ngrp + grp_) * H;                                                                   \
    const unsigned o0_ = in0_ ? (unsigned)((((plane_ + gy0_) * W + (size_t)kFftO * tx_) * kFusCh + cl) * 4) : 0xffffe000u;   \
    const unsigned o1_ = in1_ ? (unsigned)((((plane_ + gy1_) * W + (size_t)kFftO * tx_) * kFusCh + cl) * 4) : 0xffffe000u;   \
    _Pragma("unroll") for (int j = 0; j < kFftN; ++j) {                                                                      \
      r0[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o0_, (unsigned)(j * kFusCh * 4), 0));        \
      r1[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, o1_, (unsigned)(j * kFusCh * 4), 0));        \
    }                                                                                                                         \
  } while (0)
    unsigned v = blockIdx.x;
    EQA_FWD_ISSUE(v);
    float* const o0 = lds + (ry * 2) * kFusCh + cl;
    float* const o1 = lds + ((ry + 24) * 2) * kFusCh + cl;
    for (; v < nwork; v += nblk) {
      float ore[kFftH], oim[kFftH];
      fft48_r2c(r0, ore, oim);
#pragma unroll
      for (int k = 0; k < kFftH; ++k) {
        o0[k * kFusKxPitch] = ore[k];
        o0[k * kFusKxPitch + kFusCh] = oim[k];
      }
      fft48_r2c(r1, ore, oim);
#pragma unroll
      for (int k = 0; k < kFftH; ++k) {
        o1[k * kFusKxPitch] = ore[k];
        o1[k * kFusKxPitch + kFusCh] = oim[k];
      }
      __syncthreads();                                   // A: the item's row spectra are in LDS
      EQA_FWD_ISSUE(v + nblk);
      asm volatile("" ::: "memory");                     // the loads stay on this side of the barrier
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                   // B: the column waves have read their columns
    }
#undef EQA_FWD_ISSUE
    return;
  }
  // ------------------------------------------------------------------ column waves
  const int t = threadIdx.x - kFwdRowT;
  const int kc = t / kFusCh, cl = t % kFusCh;
  const bool packed = kc == kFftInner;                   // the last column task: kx = 0 and kx = 24 in one transform
  const int kx = packed ? 0 : kc + 1;
  const float* const qre = lds + kx * kFusKxPitch + cl;
  // imaginary parts: the column's own (interior columns); the REAL parts of column 24 (packed task)
  const float* const qim = packed ? lds + (kFftH - 1) * kFusKxPitch + cl : qre + kFusCh;
  const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(V, 0, v_bytes, 0x00020000);
  const unsigned step_in = (unsigned)((size_t)kFftInner * M * 2 * C * 4);       // interior columns: 23 frequencies per ky
  const unsigned step_ed = (unsigned)((size_t)2 * M * 2 * C * 4);               // edge columns: 2 per ky
  for (unsigned v = blockIdx.x; v < nwork; v += nblk) {
    const unsigned work = work_of(v);
    const unsigned grp = work % ngrp, m = work / ngrp;
    float re[kFftN], im[kFftN], ore[kFftN], oim[kFftN];
    __syncthreads();                                     // A
#pragma unroll
    for (int i = 0; i < kFftN; ++i) {
      re[i] = qre[(i * 2) * kFusCh];
      im[i] = qim[(i * 2) * kFusCh];
    }
    __syncthreads();                                     // B: LDS may be overwritten (the column is in registers)
    fft48(re, im, ore, oim);
    // [Re x 16 | Im x 16] per channel group and (frequency, tile): Re at +0, Im at +64 bytes
    const unsigned col = (unsigned)(((size_t)m * 2 * C + grp * 2 * kFusCh + cl) * 4);
    if (!__any(packed)) {                                // wave-uniform: waves without the packed task
      const unsigned voff = (unsigned)((size_t)(kx - 1) * M * 2 * C * 4) + col;
#pragma unroll
      for (int ky = 0; ky < kFftN; ++ky) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ore[ky]), vr, voff, ky * step_in, 2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oim[ky]), vr, voff + kFusCh * 4, ky * step_in, 2);
      }
    } else {
      if (!packed) {
        const unsigned voff = (unsigned)((size_t)(kx - 1) * M * 2 * C * 4) + col;
#pragma unroll
        for (int ky = 0; ky < kFftN; ++ky) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ore[ky]), vr, voff, ky * step_in, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, oim[ky]), vr, voff + kFusCh * 4, ky * step_in, 2);
        }
      } else {
        // frequencies 1104 + 2 ky (kx = 0) and 1104 + 2 ky + 1 (kx = 24), ky = 0..24
        const unsigned v0 = (unsigned)((size_t)(kFftN * kFftInner) * M * 2 * C * 4) + col;
        const unsigned v24 = v0 + (unsigned)((size_t)M * 2 * C * 4);
#pragma unroll
        for (int ky = 0; ky < kFftH; ++ky) {
          const int kn = (kFftN - ky) % kFftN;
          const float a = ore[ky], b = oim[ky], c2 = ore[kn], d = oim[kn];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 0.5f * (a + c2)), vr, v0, ky * step_ed, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 0.5f * (b - d)), vr, v0 + kFusCh * 4, ky * step_ed, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 0.5f * (b + d)), vr, v24, ky * step_ed, 2);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 0.5f * (c2 - a)), vr, v24 + kFusCh * 4, ky * step_ed, 2);
        }
      }
    }
  }
}

// Window-sum pieces of one output row of a tile: [sum of the 44 columns, the NB leftmost columns of the map, the NB
// rightmost].  The leftmost columns live in tile column 0 at j = 0..NB-1.  The rightmost start at j = a (uniform, any
// value): they are pulled out of the register array by a shift network on the bits of a -- 81 selects on 6 scalar
// conditions instead of a compare-and-add per column and border (352).
template <int NB>
__device__ __forceinline__ void fft_row_pieces(const float (&ore)[kFftN], int relu, int ncols, bool first_col, int a,
                                               float (&acc)[1 + 2 * NB]) {
  // `ore` already carries the bias (added to the row's DC bin before the transform).  relu, ncols, a are block-uniform: the
  // common cases -- a full-width tile, the right border either outside this tile or exactly at its end -- take branches without
  // per-column selects (the kernel is bound by its vector instruction count: profiles/r02/pmc_sq_counters.md).
  float v[kFftO];
  if (relu) {
#pragma unroll
    for (int j = 0; j < kFftO; ++j) v[j] = fmaxf(ore[j], 0.0f);
  } else {
#pragma unroll
    for (int j = 0; j < kFftO; ++j) v[j] = ore[j];
  }
  if (ncols < kFftO) {
#pragma unroll
    for (int j = 0; j < kFftO; ++j) v[j] = j < ncols ? v[j] : 0.0f;
  }
  float total = 0.0f;
#pragma unroll
  for (int j = 0; j < kFftO; ++j) total += v[j];
  acc[0] = total;
#pragma unroll
  for (int q = 0; q < NB; ++q) acc[1 + q] = first_col ? v[q] : 0.0f;
  constexpr int PAD = NB - 1;
  const int ap = a + PAD;  // offset of the first wanted column in the PAD-shifted array of the general case
  if (ap < 0 || ap >= kFftO + PAD) {            // the map's right border is not in this tile
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[1 + NB + q] = 0.0f;
  } else if (a == kFftO - NB) {                 // ... or ends exactly with it (88 = 2 x 44: the headline)
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[1 + NB + q] = v[kFftO - NB + q];
  } else {                                      // anywhere: shift network on the bits of its position
    constexpr int EXT = 64 + NB;
    float w[EXT];
#pragma unroll
    for (int i = 0; i < EXT; ++i) w[i] = (i >= PAD && i < PAD + kFftO) ? v[i - PAD] : 0.0f;
#pragma unroll
    for (int bit = 32; bit >= 1; bit >>= 1) {
      const bool on = (ap & bit) != 0;
#pragma unroll
      for (int i = 0; i < bit + NB - 1; ++i) w[i] = on ? w[i + bit] : w[i];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[1 + NB + q] = w[q];
  }
}

// The inverse counterpart: column pass (thread (kx, c)), LDS, row pass (thread (y, c), y < 44) and the epilogue; Mo is read
// as interleaved complex with 8-byte loads (128 bytes per (frequency, tile) row and block).  NB > 0: the window-sum pieces of this tile's 44 output columns go to segment (row, tile column); the pieces of a
// row are put together by window_sums_nhwc_finalize_kernel (sub = TX).
template <int NB, int CH, int O = kFftO>   // O: outputs per tile (window-sum pieces, NB > 0: 44 only)
__global__ __launch_bounds__(kFftN * CH) void fft48_inv_fused_kernel(const float* __restrict__ Mo, const float* __restrict__ bias,
                                                                      int relu, float* __restrict__ out, int OH, int OW, int C, int TY,
                                                                      int TX, size_t M, unsigned nwork, unsigned mo_bytes) {
  static_assert(NB == 0 || O == kFftO, "the window-sum epilogue is written for 44-output tiles");
  extern __shared__ float lds[];
  FFT_CLOCK_BEGIN();
  constexpr int kPitch = kFftN * 2 * CH + CH;  // floats per kx slab
  const unsigned bid = blockIdx.x;
  const unsigned q8 = nwork / kXcd, r8 = nwork % kXcd, xcd = bid % kXcd;
  const unsigned work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / kXcd;
  const int ngrp = C / CH;
  const int grp = work % ngrp;
  const size_t m = work / ngrp;
  const int tx = (int)(m % TX);
  const int ty = (int)((m / TX) % TY);
  const size_t img = m / ((size_t)TX * TY);
  const int cl = threadIdx.x % CH;
  const int c = grp * CH + cl;
  if (threadIdx.x < kFftH * CH) {
    const int kx = threadIdx.x / CH;
#ifdef EQA_FFT_SAMETILE  // experiment: every block reads tile 0 (L2-resident) -- the kernel without its HBM wait
    const size_t m_ld = 0;
#else
    const size_t m_ld = m;
#endif
    const float2* p = reinterpret_cast<const float2*>(Mo) + ((size_t)fft_f0(kx) * M + m_ld) * (size_t)C + c;
    const size_t fpitch = (size_t)fft_fstep(kx) * M * C;
    float re[kFftN], im[kFftN], ore[kFftN], oim[kFftN];
    const bool edge = fft_edge(kx);
    if (mo_bytes && !__any(edge)) {
      // a wave without edge columns (5 of the 7): all its lanes step through the 48 frequencies of their column by the same
      // 23 * M * C complex numbers -> buffer loads with that step as a SCALAR offset, no vector address arithmetic at all
      // (mo_bytes = 0: the spectra do not fit a 32-bit offset, pointer arithmetic as in the edge waves)
      const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Mo), 0, mo_bytes, 0x00020000);
      const unsigned voff = (unsigned)((((size_t)fft_f0(kx) * M + m_ld) * (size_t)C + c) * 8);
      const unsigned step = (unsigned)(kFftInner * M * (size_t)C * 8);
#pragma unroll
      for (int ky = 0; ky < kFftN; ++ky) {
        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(mr, voff, ky * step, 2));   // aux 2: non-temporal
        re[ky] = t[0];
        im[ky] = t[1];
      }
    } else {
      fft_load_column(p, fpitch, edge, re, im);
    }
    FFT_CLOCK_LOADS();
    fft48(im, re, oim, ore);
    float* q = lds + kx * kPitch + cl;
#pragma unroll
    for (int i = 0; i < O; ++i) {  // rows O..47: the circular wrap-around
      q[(i * 2) * CH] = ore[i];
      q[(i * 2 + 1) * CH] = oim[i];
    }
  }
  FFT_CLOCK(1);
  __syncthreads();
  FFT_CLOCK(2);
  const int y = threadIdx.x / CH;
  const int gy = O * ty + y;
  const bool valid = y < O && gy < OH;
  constexpr int NV = 1 + 2 * NB;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0f;
  if (valid) {
    const float* q = lds + (y * 2) * CH + cl;
    float re[kFftH], im[kFftH], ore[kFftN];
#pragma unroll
    for (int k = 0; k < kFftH; ++k) {
      re[k] = q[k * kPitch];
      im[k] = q[k * kPitch + CH];
    }
    // the bias rides on the row's DC bin: the (unnormalised) inverse adds re[0] to every output
    const float b = bias ? bias[c] : 0.0f;
    re[0] += b;
    ifft48_c2r(re, im, ore);  // real output from the stored half of the spectrum (468 operations; the complex transform: 819)
    const int x0 = O * tx;
    const int ncols = min(O, OW - x0);  // uniform
    if constexpr (NB == 0) {
      float* o = out + ((img * OH + gy) * OW + x0) * (size_t)C + c;
#pragma unroll
      for (int j = 0; j < O; ++j) {
        if (j < ncols) o[(size_t)j * C] = relu ? fmaxf(ore[j], 0.0f) : ore[j];
      }
    } else {
      fft_row_pieces<NB>(ore, relu, ncols, tx == 0, OW - NB - x0, acc);
    }
  }
  FFT_CLOCK(3);
  if (NB > 0) {
    // Window-sum pieces.  Segments (the order window_sums_nhwc_finalize_kernel expects): the NB top rows, the NB bottom rows,
    // then ONE per tile row for its interior rows -- those are only ever needed as a sum, which the block forms here in a
    // fixed order (LDS is free once every thread has read its row spectrum) instead of writing 44 pieces per tile column.
    const int nseg = 2 * NB + TY;
    const bool border = gy < NB || gy >= OH - NB;
    if (valid && border) {
      const int seg = gy < NB ? gy : NB + (gy - (OH - NB));
      float* o = out + (((img * nseg + seg) * TX + tx) * (size_t)C + c) * NV;
#pragma unroll
      for (int i = 0; i < NV; ++i) o[i] = acc[i];
    }
    // a wave holds 64 / CH consecutive rows of the same CH channels: those are summed by wavefront shuffles first, one LDS slot
    // per wave instead of one per row (44 -> 11 terms in the serial sum below, a quarter of the LDS traffic)
    constexpr int kRowsPerWave = 64 / CH;
    static_assert(kFftO % kRowsPerWave == 0 || CH > 16, "the rows of a wave are either all below 44 or all above");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float v = (valid && !border) ? acc[i] : 0.0f;
#pragma unroll
      for (int o = CH; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      acc[i] = v;
    }
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    constexpr int kWaves = (kFftO * CH + 63) / 64;      // waves that hold output rows
    if ((threadIdx.x & 63) < CH && wv < kWaves) {
#pragma unroll
      for (int i = 0; i < NV; ++i) lds[(wv * NV + i) * CH + cl] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < NV * CH) {
      const int i = threadIdx.x / CH;
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < kWaves; ++r) t += lds[(r * NV + i) * CH + cl];
      // tile rows made of border rows only (OH <= 2 NB + ...) contribute an all-zero piece: harmless
      out[(((img * nseg + 2 * NB + ty) * TX + tx) * (size_t)C + c) * NV + i] = t;
    }
  }
  FFT_CLOCK(4);
}

// The fused inverse as a persistent producer / consumer pipeline.  PMC (profiles/r02/pmc_memory_path.md): the fused kernel above
// keeps only 34 cache lines per CU in flight on average (the group action: 69, the forward transform: 75) because a block loads
// for 43 % of its life and computes for the rest, and with 155 KB of LDS no second block can fill the gap.  Here the block's
// waves are specialised and walk the work items together:
//   producers (7 waves, thread (kx, c)): column transform of item i -> LDS | barrier A | issue the 48 loads of item i+1 |
//                                       barrier B | (wait for the loads) column transform of item i+1 ...
//   consumers (5 waves, thread (row, c), 20 rows per pass, 3 passes): barrier A | read row, transform, epilogue | ... | read last row |
//                                       barrier B | transform, epilogue, pieces ...
// so the loads of the next item are in flight while the consumers work on this one, and the producers' registers (they only
// hold the incoming column) cost nothing meanwhile.  12 waves = 3 per SIMD, 168 VGPRs each (with a sixth consumer wave the limit is
// 128 and the producer loop spills its incoming column).
constexpr int kPipeProd = 7 * 64, kPipeCons = 5 * 64, kPipeThreads = kPipeProd + kPipeCons;
constexpr int kPipePasses = (kFftO * 16 + kPipeCons - 1) / kPipeCons;  // 20 rows of 16 channels per pass: 3 passes (20 + 20 + 4)

// STATS (NB == 0, training: an InnerBatchNorm follows): every consumer thread also sums the values (and their squares) it stores
// for its channel; per item, a consumer wave's 4 rows are folded by shuffles and its 16 lanes write one fp64 partial row
//   stats[((tile * 5 + consumer wave) * C + c) * 2 + {0, 1}]
// -- the statistics pass over the finished map (eqa_bn_stats_nhwc, 0.45 ms at the headline shape) is not needed.
template <int NB, int CH, bool STATS = false>
__global__ __launch_bounds__(kPipeThreads) void fft48_inv_pipe_kernel(const float* __restrict__ Mo, const float* __restrict__ bias,
                                                                     int relu, float* __restrict__ out, int OH, int OW, int C, int TY,
                                                                     int TX, size_t M, unsigned nwork, unsigned mo_bytes,
                                                                     double* __restrict__ stats) {
  static_assert(!STATS || NB == 0, "statistics are taken of the full map");
  extern __shared__ float lds[];
  constexpr int kPitch = kFftN * 2 * CH + CH;  // floats per kx slab
  constexpr int NV = 1 + 2 * NB;
  constexpr int kPassRows = kPipeCons / CH;      // 20 rows per consumer pass
  const unsigned nblk = gridDim.x;               // a multiple of the XCD count (or < 8): virtual block v runs on XCD v % 8 = blockIdx % 8
  const int ngrp = C / CH;
  const unsigned q8 = nwork / kXcd, r8 = nwork % kXcd;
  auto work_of = [&](unsigned v) {
    const unsigned xcd = v % kXcd;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + v / kXcd;
  };
  if (threadIdx.x < kPipeProd) {
    // ---------------------------------------------------------------- producers
    // the 48 idle lanes of the seventh wave duplicate its column kx = 24 (same loads, same values to the same LDS words): the
    // loop body stays straight-line code
    const int kx = min((int)threadIdx.x / CH, kFftH - 1);
    const int cl = threadIdx.x % CH;
    const bool edge = fft_edge(kx);
    // One load path for every lane: buffer loads (32-bit offsets: the launcher checks that the spectra fit) at
    // base + kk * lstep, lstep = this lane's frequency step.  Edge columns store rows 0..24 only: their rows ky > 24 are the
    // conjugates of rows 48 - ky.  Past the last item the base is out of range: no memory access, zeros.
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Mo), 0, mo_bytes, 0x00020000);
    unsigned lstep = (unsigned)((size_t)fft_fstep(kx) * M * C * 8);
    const unsigned flip = edge ? 0x80000000u : 0u;
    const unsigned col0 = (unsigned)((size_t)fft_f0(kx) * M * C * 8) + (unsigned)cl * 8;
    float re[kFftN], im[kFftN];
    // lstep is made opaque at every issue: otherwise its 48 multiples are hoisted out of the item loop as loop invariants and
    // the incoming column is spilled to make room for them
#ifndef EQA_PIPE_EARLY
#define EQA_PIPE_EARLY 16   // loads of the next item issued BEFORE barrier A (see the loop below)
#endif
#define EQA_PIPE_ISSUE_RANGE(v_, K0_, K1_)                                                                                    \
  do {                                                                                                                       \
    asm volatile("" : "+v"(lstep));                                                                                          \
    const unsigned work_ = work_of(v_);                                                                                      \
    const unsigned item_ = (unsigned)(((size_t)(work_ / ngrp) * C + (size_t)(work_ % ngrp) * CH) * 8);                       \
    const unsigned base_ = (v_) < nwork ? col0 + item_ : 0xfffffff0u;                                                        \
    _Pragma("unroll") for (int ky = (K0_); ky < (K1_); ++ky) {                                                               \
      const unsigned kk_ = ky > kFftH - 1 ? (edge ? (unsigned)(kFftN - ky) : (unsigned)ky) : (unsigned)ky;                   \
      const unsigned off_ = (v_) < nwork ? base_ + kk_ * lstep : 0xfffffff0u;                                                \
      const f32x2 t_ = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(mr, off_, 0, 2));                      \
      re[ky] = t_[0];                                                                                                        \
      im[ky] = ky > kFftH - 1 ? __uint_as_float(__float_as_uint(t_[1]) ^ flip) : t_[1];                                      \
    }                                                                                                                        \
    /* every offset right in front of its load (left alone, the scheduler forms all the offsets first: as many more registers) */ \
    _Pragma("unroll") for (int ky = (K0_); ky < (K1_); ++ky) {                                                               \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                                     \
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                     \
    }                                                                                                                        \
  } while (0)
#define EQA_PIPE_ISSUE(v_) EQA_PIPE_ISSUE_RANGE(v_, 0, kFftN)
    unsigned v = blockIdx.x;
    EQA_PIPE_ISSUE(v);
    float* const q = lds + kx * kPitch + cl;
    PIPE_CLOCK_BEGIN(0);  // producer stamps: [16] wait for the loads, [17] column transform + LDS writes, [18] wait at A, [19] issue, [20] wait at B
    for (; v < nwork; v += nblk) {
      PIPE_CLOCK_WAIT(0);
      float ore[kFftN], oim[kFftN];
      fft48(im, re, oim, ore);
#pragma unroll
      for (int i = 0; i < kFftO; ++i) {  // rows 44..47: the circular wrap-around
        q[(i * 2) * CH] = ore[i];
        q[(i * 2 + 1) * CH] = oim[i];
      }
      PIPE_CLOCK(1);
      // The column's registers are free as soon as its transform is in LDS, and the producers used to wait ~2.4 k cycles at A for
      // the consumers' epilogue of the previous item with nothing in flight: the first EQA_PIPE_EARLY loads of the next item go out
      // in that window (more would delay the producers' arrival at A -- the memory pipe throttles the issue -- and the consumers
      // with it), the rest behind A as before.
      __builtin_amdgcn_sched_barrier(0);
      EQA_PIPE_ISSUE_RANGE(v + nblk, 0, EQA_PIPE_EARLY);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                   // A: the item's column transforms are in LDS
      PIPE_CLOCK(2);
      EQA_PIPE_ISSUE_RANGE(v + nblk, EQA_PIPE_EARLY, kFftN);
      asm volatile("" ::: "memory");                     // the loads stay on this side of the barrier
      __builtin_amdgcn_sched_barrier(0);
      PIPE_CLOCK(3);
      __syncthreads();                                   // B: the consumers have read their last row
      PIPE_CLOCK(4);
    }
    PIPE_CLOCK_END(16);
#undef EQA_PIPE_ISSUE
#undef EQA_PIPE_ISSUE_RANGE
    return;
  }
  // ------------------------------------------------------------------ consumers
  const int t = threadIdx.x - kPipeProd;
  const int r = t / CH, cl = t % CH;
  const unsigned cons_wave = __builtin_amdgcn_readfirstlane((unsigned)t >> 6);   // wave-uniform: lives in a scalar register
  PIPE_CLOCK_BEGIN(kPipeProd);  // consumer stamps: [24] wait at A, [25] passes before the last LDS read, [26] wait at B, [27] last pass + pieces
  for (unsigned v = blockIdx.x; v < nwork; v += nblk) {
    const unsigned work = work_of(v);
    const int grp = work % ngrp;
    const size_t m = work / ngrp;
    const int tx = (int)(m % TX);
    const int ty = (int)((m / TX) % TY);
    const size_t img = m / ((size_t)TX * TY);
    const int c = grp * CH + cl;
    const float b = bias ? bias[c] : 0.0f;
    const int x0 = kFftO * tx;
    const int ncols = min(kFftO, OW - x0);  // uniform
    const int nseg = 2 * NB + (kPipeCons / 64) * TY;   // border rows one by one, then one interior piece per (tile row, consumer wave)
    float tot[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) tot[i] = 0.0f;
    float st_s = 0.0f, st_q = 0.0f;                      // STATS: this thread's (<= 3 rows x 44) stored values, summed
    __syncthreads();                                     // A
    PIPE_CLOCK(0);
#pragma unroll 1
    for (int pass = 0; pass < kPipePasses; ++pass) {
      const int y = pass * kPassRows + r;
      const int gy = kFftO * ty + y;
      const bool valid = y < kFftO && gy < OH;
      float re[kFftH], im[kFftH], ore[kFftN];
      if (valid) {
        const float* q = lds + (y * 2) * CH + cl;
#pragma unroll
        for (int k = 0; k < kFftH; ++k) {
          re[k] = q[k * kPitch];
          im[k] = q[k * kPitch + CH];
        }
      }
      if (pass == kPipePasses - 1) {
        PIPE_CLOCK(1);
        __syncthreads();                                 // B: LDS may be overwritten (the values are in registers)
        PIPE_CLOCK(2);
      }
      if (valid) {
        re[0] += b;  // the bias rides on the row's DC bin: the (unnormalised) inverse adds re[0] to every output
        ifft48_c2r(re, im, ore);
        if constexpr (NB == 0) {
          float* o = out + ((img * OH + gy) * OW + x0) * (size_t)C + c;
#pragma unroll
          for (int j = 0; j < kFftO; ++j) {
            if (j < ncols) {
              const float w = relu ? fmaxf(ore[j], 0.0f) : ore[j];
              o[(size_t)j * C] = w;
              if (STATS) {
                st_s += w;
                st_q = fmaf(w, w, st_q);
              }
            }
          }
        } else {
          float acc[NV];
          fft_row_pieces<NB>(ore, relu, ncols, tx == 0, OW - NB - x0, acc);
          if (gy < NB || gy >= OH - NB) {
            const int seg = gy < NB ? gy : NB + (gy - (OH - NB));
            float* o = out + (((img * nseg + seg) * TX + tx) * (size_t)C + c) * NV;
#pragma unroll
            for (int i = 0; i < NV; ++i) o[i] = acc[i];
          } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) tot[i] += acc[i];
          }
        }
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int o = CH; o < 64; o <<= 1) {
        st_s += __shfl_xor(st_s, o, 64);
        st_q += __shfl_xor(st_q, o, 64);
      }
      if ((t & 63) < CH) {
        double* o = stats + ((m * (kPipeCons / 64) + cons_wave) * C + c) * 2;
        o[0] = (double)st_s;
        o[1] = (double)st_q;
      }
    }
    if constexpr (NB > 0) {
      // a wave holds 64 / CH rows of the same CH channels: summed by shuffles, one interior piece per wave and tile row
      const int cw = t >> 6;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float s_ = tot[i];
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) s_ += __shfl_xor(s_, o, 64);
        tot[i] = s_;
      }
      if ((t & 63) < CH) {
        float* o = out + (((img * nseg + 2 * NB + (kPipeCons / 64) * ty + cw) * TX + tx) * (size_t)C + c) * NV;
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = tot[i];
      }
    }
    PIPE_CLOCK(3);
  }
  PIPE_CLOCK_END(24);
}

// Filter spectra for the batched GEMM: bank (Cout, Cin, 5, 5) -> B (F, 2 Cin, 2 Cout), B[f] = [[Br, Bi], [-Bi, Br]] with
// Br + i Bi = conj(FFT48x48(filter))[ky][kx] / 48^2 = sum_{u,v} w[u][v] (cos t + i sin t) / 2304, t = 2 pi (ky u + kx v) / 48.
// Rows follow the rows of V ([Re x G | Im x G] per group of G input channels), columns the rows of Mo (interleaved complex).
// One thread per (ci, co) keeps its 25 taps in registers and walks the frequencies (fp64 accumulation, twiddles from a
// 48-entry table): 0.3 ms for 256 x 256 filters, against 13.6 ms for the same through torch.fft + concatenations -- cheap
// enough to run every training step.
// `sgn` = +1: the correlation form above (forward pass); -1: FFT(filter) itself, for the convolution of the input gradient.
template <int KS>
__global__ __launch_bounds__(kThreads) void fft48_filter_spectra_kernel(const float* __restrict__ bank, float* __restrict__ B, int Cout,
                                                                       int Cin, int G, float sgn) {
  __shared__ double tw_c[kFftN], tw_s[kFftN];
  if (threadIdx.x < kFftN) {
    const double t = 6.283185307179586476925286766559 * threadIdx.x / kFftN;
    tw_c[threadIdx.x] = cos(t);
    tw_s[threadIdx.x] = sin(t);
  }
  __syncthreads();
  const int co = blockIdx.y * kThreads + threadIdx.x;
  const int ci = blockIdx.x;
  if (co >= Cout) return;
  double w[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) w[i] = bank[((size_t)co * Cin + ci) * (KS * KS) + i];
  const int r0 = (ci / G) * 2 * G + ci % G, r1 = r0 + G;
  const size_t fstride = (size_t)2 * Cin * 2 * Cout;
  float2* o0 = reinterpret_cast<float2*>(B + (size_t)r0 * 2 * Cout) + co;
  float2* o1 = reinterpret_cast<float2*>(B + (size_t)r1 * 2 * Cout) + co;
  constexpr double inv = 1.0 / (kFftN * kFftN);
  // separable: S_u(kx) = sum_v w[u][v] e^{i t kx v} once per kx, then sum_u e^{i t ky u} S_u for the 48 ky
  // (25 x (50 + 48 x 20) multiply-adds per filter instead of 1200 x 50 in the direct form, and a fifth of the table look-ups)
  for (int kx = 0; kx < kFftH; ++kx) {
    double sr[KS], si[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      sr[u] = 0.0;
      si[u] = 0.0;
#pragma unroll
      for (int v = 0; v < KS; ++v) {
        const int t = (kx * v) % kFftN;
        sr[u] += w[u * KS + v] * tw_c[t];
        si[u] += w[u * KS + v] * tw_s[t];
      }
    }
    const int nky = fft_nky(kx), f0 = fft_f0(kx), fstep = fft_fstep(kx);
    for (int ky = 0; ky < nky; ++ky) {
      double br = 0.0, bi = 0.0;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const int t = (ky * u) % kFftN;
        const double c = tw_c[t], sn = tw_s[t];
        br += c * sr[u] - sn * si[u];
        bi += c * si[u] + sn * sr[u];
      }
      const float fr = (float)(br * inv), fi = sgn * (float)(bi * inv);
      const size_t f = (size_t)(f0 + ky * fstep) * (fstride / 2);  // in float2
      o0[f] = make_float2(fr, fi);
      o1[f] = make_float2(-fi, fr);
    }
  }
}

// The same filter spectra in the operand order of the hand-written 3-multiplication complex GEMM (cgemm3m.hip):
// B3 (F, S = Cin/16, Cout/32, 3 [Br | Bi | Br + Bi], 2 [b], 64 [lane = 32 h + j], 4 [t]) with k = 16 s + 8 b + 4 h + t the input
// channel and 32 c + j the output channel: a wave's B fragment of one MFMA k-block is one contiguous 1 KB run.  Thread = (output
// channel, 4 consecutive input channels = t) for one kx: three 16-byte stores per frequency, fully coalesced over the lanes.
// Br + Bi is the correctly rounded sum of the two STORED floats (fp64 add of the rounded values), so that
// Ci = (Ar + Ai)(Br + Bi) - Ar Br - Ai Bi cancels against exactly the Br, Bi the other two products see.
template <int KS>
__global__ __launch_bounds__(kThreads) void fft48_filter_spectra3m_kernel(const float* __restrict__ bank, float* __restrict__ B3, int Cout,
                                                                         int Cin, float sgn) {
  __shared__ double tw_c[kFftN], tw_s[kFftN];
  if (threadIdx.x < kFftN) {
    const double t = 6.283185307179586476925286766559 * threadIdx.x / kFftN;
    tw_c[threadIdx.x] = cos(t);
    tw_s[threadIdx.x] = sin(t);
  }
  __syncthreads();
  const int co = blockIdx.y * kThreads + threadIdx.x;
  const int cq = blockIdx.x;                      // quad of input channels 4 cq .. 4 cq + 3
  const int kx = blockIdx.z;
  if (co >= Cout) return;
  constexpr double inv = 1.0 / (kFftN * kFftN);
  // S_u(kx) = sum_v w[u][v] e^{i t kx v} for the 4 filters of this thread
  double sr[4][KS], si[4][KS];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float* w = bank + ((size_t)co * Cin + 4 * cq + c) * (KS * KS);
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      sr[c][u] = 0.0;
      si[c][u] = 0.0;
#pragma unroll
      for (int v = 0; v < KS; ++v) {
        const int t = (kx * v) % kFftN;
        const double wv = w[u * KS + v];
        sr[c][u] += wv * tw_c[t];
        si[c][u] += wv * tw_s[t];
      }
    }
  }
  const int S = Cin / 16;
  const int k0 = 4 * cq, s = k0 / 16, b = (k0 % 16) / 8, h = (k0 % 8) / 4;
  const int c32 = co / 32, lane = 32 * h + (co % 32);
  const size_t per_f = (size_t)Cin * Cout * 3;
  float4* o = reinterpret_cast<float4*>(B3 + (((((size_t)s * (Cout / 32) + c32) * 3) * 2 + b) * 64 + lane) * 4);   // part 0
  const size_t part = (size_t)2 * 64;             // float4 between the parts
  (void)S;
  const int nky = fft_nky(kx), f0 = fft_f0(kx), fstep = fft_fstep(kx);
  for (int ky = 0; ky < nky; ++ky) {
    float fr[4], fi[4], fs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double br = 0.0, bi = 0.0;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const int t = (ky * u) % kFftN;
        const double cs = tw_c[t], sn = tw_s[t];
        br += cs * sr[c][u] - sn * si[c][u];
        bi += cs * si[c][u] + sn * sr[c][u];
      }
      fr[c] = (float)(br * inv);
      fi[c] = sgn * (float)(bi * inv);
      fs[c] = (float)((double)fr[c] + (double)fi[c]);
    }
    float4* of = o + (size_t)(f0 + ky * fstep) * (per_f / 4);
    of[0] = make_float4(fr[0], fr[1], fr[2], fr[3]);
    of[part] = make_float4(fi[0], fi[1], fi[2], fi[3]);
    of[2 * part] = make_float4(fs[0], fs[1], fs[2], fs[3]);
  }
}

// Filter gradient in the frequency domain (training).  With G = the spectra of the output-gradient tiles (44 x 44, zero-padded
// to 48: eqa_fft48k5_grad_transform) and V those of the input tiles, D[f] = V[f]^T . G[f] (real form, one batched GEMM over
// the tiles) holds  Dr = D[re ci][re co] + D[im ci][im co],  Di = D[im ci][re co] - D[re ci][im co]  of  X_f^T conj(G_f), and
//   dW[co][ci][u][v] = 1/48^2 sum_f wgt(f) (cos t Dr - sin t Di),  t = 2 pi (ky u + kx v) / 48,
// wgt = 2 for the stored frequencies whose conjugate partner is not stored, 1 for the self-conjugate ones -- the correlation theorem; no
// wrap-around because a 44-wide gradient tile shifted by up to 4 stays inside the 48-wide input tile.
// PACKED: D3 (F, Cin, 2, Cout) as eqa_fft48k5_wgrad3m writes it -- Dr | Di per input channel, plain channel order.
template <bool PACKED, int KS>
__global__ __launch_bounds__(kThreads) void fft48_filter_grad_kernel(const float* __restrict__ D, float* __restrict__ dbank, int Cout,
                                                                    int Cin, int Gin, int Gout) {
  __shared__ double tw_c[kFftN], tw_s[kFftN];
  if (threadIdx.x < kFftN) {
    const double t = 6.283185307179586476925286766559 * threadIdx.x / kFftN;
    tw_c[threadIdx.x] = cos(t);
    tw_s[threadIdx.x] = sin(t);
  }
  __syncthreads();
  const int co = blockIdx.y * kThreads + threadIdx.x;
  const int ci = blockIdx.x;
  if (co >= Cout) return;
  const int r0 = (ci / Gin) * 2 * Gin + ci % Gin, r1 = r0 + Gin;
  const int c0 = (co / Gout) * 2 * Gout + co % Gout, c1 = c0 + Gout;
  const size_t ld = (size_t)2 * Cout, fstride = (size_t)2 * Cin * (PACKED ? (size_t)Cout : ld);
  const size_t p_dr = ((size_t)2 * ci) * Cout + co, p_di = p_dr + Cout;   // PACKED
  double acc[KS * KS];
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) acc[i] = 0.0;
  // separable, like the spectra kernel: A_u(kx) = sum_ky e^{i t ky u} D(ky, kx), then dW[u][v] += Re(e^{i t kx v} A_u(kx))
  for (int kx = 0; kx < kFftH; ++kx) {
    const bool edge = fft_edge(kx);
    const int nky = fft_nky(kx), f0 = fft_f0(kx), fstep = fft_fstep(kx);
    double ar[KS], ai[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) { ar[u] = 0.0; ai[u] = 0.0; }
    // one block per CU and one thread per filter: the loop is a chain of round trips unless several frequencies are requested
    // together (0.56 ms for 1.26 GB at one ky per trip).  Eight per trip; the sums run over ky in the same order.
    constexpr int kKyBatch = 8;
    for (int ky0 = 0; ky0 < nky; ky0 += kKyBatch) {
      float q[kKyBatch][PACKED ? 2 : 4];
#pragma unroll
      for (int b = 0; b < kKyBatch; ++b) {
        const float* d = D + (size_t)(f0 + min(ky0 + b, nky - 1) * fstep) * fstride;
        if (PACKED) {
          q[b][0] = d[p_dr];
          q[b][1] = d[p_di];
        } else {
          q[b][0] = d[r0 * ld + c0];
          q[b][1] = d[r1 * ld + c1];
          q[b][PACKED ? 0 : 2] = d[r1 * ld + c0];
          q[b][PACKED ? 1 : 3] = d[r0 * ld + c1];
        }
      }
#pragma unroll
      for (int b = 0; b < kKyBatch; ++b) {
        const int ky = ky0 + b;
        if (ky < nky) {
          // weight 2 for every stored frequency whose conjugate partner is not stored; 1 for the four self-conjugate ones
          const double wgt = (edge && (ky == 0 || ky == kFftH - 1)) ? 1.0 : 2.0;
          const double dr = PACKED ? wgt * (double)q[b][0] : wgt * ((double)q[b][0] + (double)q[b][1]);
          const double di = PACKED ? wgt * (double)q[b][1] : wgt * ((double)q[b][PACKED ? 0 : 2] - (double)q[b][PACKED ? 1 : 3]);
#pragma unroll
          for (int u = 0; u < KS; ++u) {
            const int t = (ky * u) % kFftN;
            const double c = tw_c[t], sn = tw_s[t];
            ar[u] += c * dr - sn * di;
            ai[u] += c * di + sn * dr;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int v = 0; v < KS; ++v) {
        const int t = (kx * v) % kFftN;
        acc[u * KS + v] += tw_c[t] * ar[u] - tw_s[t] * ai[u];
      }
  }
  constexpr double inv = 1.0 / (kFftN * kFftN);
  float* o = dbank + ((size_t)co * Cin + ci) * (KS * KS);
#pragma unroll
  for (int i = 0; i < KS * KS; ++i) o[i] = (float)(acc[i] * inv);
}

// The same reduction with one thread per (ci, co, filter ROW u = blockIdx.z): KS x the blocks and 1 / KS of the accumulators.  With
// few channels the kernel above is a handful of blocks of long serial threads (64 x 64 channels at k = 9: 64 blocks, 81 fp64
// accumulators per thread, 0.52 ms -- 5 % of the reference tutorial's training step); the KS rows re-read D from L2.  Same sums
// in the same order per element, so both forms give identical filters.
template <bool PACKED, int KS>
__global__ __launch_bounds__(kThreads) void fft48_filter_grad_rows_kernel(const float* __restrict__ D, float* __restrict__ dbank, int Cout,
                                                                         int Cin, int Gin, int Gout) {
  __shared__ double tw_c[kFftN], tw_s[kFftN];
  if (threadIdx.x < kFftN) {
    const double t = 6.283185307179586476925286766559 * threadIdx.x / kFftN;
    tw_c[threadIdx.x] = cos(t);
    tw_s[threadIdx.x] = sin(t);
  }
  __syncthreads();
  const int co = blockIdx.y * kThreads + threadIdx.x;
  const int ci = blockIdx.x;
  const int u = blockIdx.z;
  if (co >= Cout) return;
  const int r0 = (ci / Gin) * 2 * Gin + ci % Gin, r1 = r0 + Gin;
  const int c0 = (co / Gout) * 2 * Gout + co % Gout, c1 = c0 + Gout;
  const size_t ld = (size_t)2 * Cout, fstride = (size_t)2 * Cin * (PACKED ? (size_t)Cout : ld);
  const size_t p_dr = ((size_t)2 * ci) * Cout + co, p_di = p_dr + Cout;
  double acc[KS];
#pragma unroll
  for (int v = 0; v < KS; ++v) acc[v] = 0.0;
  for (int kx = 0; kx < kFftH; ++kx) {
    const bool edge = fft_edge(kx);
    const int nky = fft_nky(kx), f0 = fft_f0(kx), fstep = fft_fstep(kx);
    double ar = 0.0, ai = 0.0;
    constexpr int kKyBatch = 8;
    for (int ky0 = 0; ky0 < nky; ky0 += kKyBatch) {
      float q[kKyBatch][PACKED ? 2 : 4];
#pragma unroll
      for (int b = 0; b < kKyBatch; ++b) {
        const float* d = D + (size_t)(f0 + min(ky0 + b, nky - 1) * fstep) * fstride;
        if (PACKED) {
          q[b][0] = d[p_dr];
          q[b][1] = d[p_di];
        } else {
          q[b][0] = d[r0 * ld + c0];
          q[b][1] = d[r1 * ld + c1];
          q[b][PACKED ? 0 : 2] = d[r1 * ld + c0];
          q[b][PACKED ? 1 : 3] = d[r0 * ld + c1];
        }
      }
#pragma unroll
      for (int b = 0; b < kKyBatch; ++b) {
        const int ky = ky0 + b;
        if (ky < nky) {
          const double wgt = (edge && (ky == 0 || ky == kFftH - 1)) ? 1.0 : 2.0;
          const double dr = PACKED ? wgt * (double)q[b][0] : wgt * ((double)q[b][0] + (double)q[b][1]);
          const double di = PACKED ? wgt * (double)q[b][1] : wgt * ((double)q[b][PACKED ? 0 : 2] - (double)q[b][PACKED ? 1 : 3]);
          const int t = (ky * u) % kFftN;
          const double c = tw_c[t], sn = tw_s[t];
          ar += c * dr - sn * di;
          ai += c * di + sn * dr;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < KS; ++v) {
      const int t = (kx * v) % kFftN;
      acc[v] += tw_c[t] * ar - tw_s[t] * ai;
    }
  }
  constexpr double inv = 1.0 / (kFftN * kFftN);
  float* o = dbank + ((size_t)co * Cin + ci) * (KS * KS) + u * KS;
#pragma unroll
  for (int v = 0; v < KS; ++v) o[v] = (float)(acc[v] * inv);
}

int fft_dims_ok(int nimg, int H, int W, int C) { return nimg >= 0 && H >= 5 && W >= 5 && C > 0; }

#ifndef EQA_FFT_INV_CH
#define EQA_FFT_INV_CH 16
#endif
constexpr int kInvCh = EQA_FFT_INV_CH;  // channels per block of the fused inverse (8, two blocks per CU: 1.54 ms; 16: 1.25 ms)
int fft_group_in(int C) { return C % kFusCh == 0 ? kFusCh : 1; }

}  // namespace

// The row pass writes an intermediate the column pass reads straight back.  Both run on chunks of images small enough for
// that intermediate (9.4 MB per 92 x 92 x 256 image) to stay in the 256 MB Infinity Cache.
#ifndef EQA_FFT_CHUNK_BYTES
#define EQA_FFT_CHUNK_BYTES (96ull << 20)
#endif
static int fft_chunk_images(int nimg, int rows, int TX, int C) {
  const size_t per_img = (size_t)rows * TX * kFftH * 2 * C * sizeof(float);
  size_t n = EQA_FFT_CHUNK_BYTES / (per_img ? per_img : 1);
  if (n < 1) n = 1;
  return (int)(n < (size_t)nimg ? n : (size_t)nimg);
}

// stats != nullptr (NB == 0 only): the pipeline's STATS form, or EQA_ERR_UNSUPPORTED where the pipeline does not apply;
// stats_rows != nullptr: no launch, *stats_rows = the partial rows that form writes (0: it does not apply)
template <int NB>
static int fft_output_impl(const float* Mo, float* T2, const float* bias, int relu, float* out, int nimg, int OH, int OW, int C,
                           hipStream_t st, int* fused, double* stats = nullptr, int64_t* stats_rows = nullptr) {
  const int TY = (OH + kFftO - 1) / kFftO, TX = (OW + kFftO - 1) / kFftO;
  const size_t M = (size_t)nimg * TY * TX;
  if (stats_rows) *stats_rows = 0;
  if ((size_t)nimg * OH > 0x7fffffffULL || M * kFftH > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  *fused = 0;
  static const bool two_pass = getenv("EQA_FFT_TWO_PASS") != nullptr;
  // persistent producer / consumer pipeline (EQA_FFT_INV_PIPE=0: one block per work item); with window-sum pieces the buffer
  // must hold 2 NB + 6 TY segments per image (it has OH)
  static const bool pipe_on = []() { const char* e = getenv("EQA_FFT_INV_PIPE"); return !(e && e[0] == '0'); }();
  if (pipe_on && kInvCh == 16 && C % kInvCh == 0 && M * (C / kInvCh) <= 0x7fffffffULL && !two_pass &&
      (size_t)kFftF * fft_pitch(M) * C * 8 <= 0xffffff00ULL && (NB == 0 || 2 * NB + (kPipeCons / 64) * TY <= OH)) {
    constexpr int lds_bytes = kFftH * (kFftN * 2 * kInvCh + kInvCh) * (int)sizeof(float);
    const bool lds_ok = allow_dynamic_lds((const void*)fft48_inv_pipe_kernel<NB, kInvCh>, lds_bytes);
    static const int n_cu = []() { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    if (lds_ok) {
      const unsigned nwork = (unsigned)(M * (C / kInvCh));
      const unsigned blocks = nwork < (unsigned)n_cu ? nwork : (unsigned)(n_cu / kXcd * kXcd);
      const size_t mo_total = (size_t)kFftF * fft_pitch(M) * C * 8;
      if constexpr (NB == 0) {
        if (stats || stats_rows) {
          if (!allow_dynamic_lds((const void*)fft48_inv_pipe_kernel<0, kInvCh, true>, lds_bytes)) return EQA_ERR_UNSUPPORTED;
          if (stats_rows) { *stats_rows = (int64_t)M * (kPipeCons / 64); return EQA_OK; }
          hipLaunchKernelGGL((fft48_inv_pipe_kernel<0, kInvCh, true>), dim3(blocks), dim3(kPipeThreads), lds_bytes, st, Mo, bias, relu, out,
                             OH, OW, C, TY, TX, fft_pitch(M), nwork, mo_total <= 0xfffffff0ULL ? (unsigned)mo_total : 0u, stats);
          *fused = 3;
          return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
        }
      }
      hipLaunchKernelGGL((fft48_inv_pipe_kernel<NB, kInvCh>), dim3(blocks), dim3(kPipeThreads), lds_bytes, st, Mo, bias, relu, out, OH,
                         OW, C, TY, TX, fft_pitch(M), nwork, mo_total <= 0xfffffff0ULL ? (unsigned)mo_total : 0u, (double*)nullptr);
      *fused = 3;
      return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
    }
    (void)hipGetLastError();
  }
  if (stats || stats_rows) return EQA_ERR_UNSUPPORTED;
  if (C % kInvCh == 0 && M * (C / kInvCh) <= 0x7fffffffULL && !two_pass) {
    constexpr int lds_bytes = kFftH * (kFftN * 2 * kInvCh + kInvCh) * (int)sizeof(float);
    const bool lds_ok = allow_dynamic_lds((const void*)fft48_inv_fused_kernel<NB, kInvCh>, lds_bytes);
    if (lds_ok) {
      const unsigned nwork = (unsigned)(M * (C / kInvCh));
      const size_t mo_total = (size_t)kFftF * fft_pitch(M) * C * 8;      // bytes of Mo; 0 to the kernel = beyond 32-bit offsets
      hipLaunchKernelGGL((fft48_inv_fused_kernel<NB, kInvCh>), dim3(nwork), dim3(kFftN * kInvCh), lds_bytes, st, Mo, bias, relu, out, OH,
                         OW, C, TY, TX, fft_pitch(M), nwork, mo_total <= 0xfffffff0ULL ? (unsigned)mo_total : 0u);
      *fused = 1;
      return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
    }
    (void)hipGetLastError();
  }
  const unsigned cb = (C + kThreads - 1) / kThreads;
  const int chunk = fft_chunk_images(nimg, OH, TX, C);
  for (int i0 = 0; i0 < nimg; i0 += chunk) {
    const int n = std::min(chunk, nimg - i0);
    hipLaunchKernelGGL((fft48_cols_inv_kernel<false, kFftO>), dim3((unsigned)((size_t)n * TY * TX * kFftH), cb), dim3(kThreads), 0, st, Mo, T2, OH, C,
                       TY, TX, fft_pitch(M), (size_t)i0 * TY * TX);
    hipLaunchKernelGGL((fft48_rows_inv_kernel<NB, kFftO>), dim3((unsigned)((size_t)n * OH), cb), dim3(kThreads), 0, st, T2, bias, relu, out, OH,
                       OW, C, TX, (size_t)i0);
  }
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

extern "C" {
#ifdef EQA_FFT_CLOCK
int eqa_debug_fft_clock(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft_clock), sizeof(g_fft_clock)) == hipSuccess ? 0 : -1; }
#endif

int eqa_fft48k5_group(int C, int side) {
  if (C <= 0 || (side != 0 && side != 1)) return EQA_ERR_INVALID_ARG;
  return side == 0 ? fft_group_in(C) : 1;
}

int eqa_fft48k5_frequencies(void) { return kFftF; }

int64_t eqa_fft48k5_tiles(int n) { return n <= 4 ? 0 : (n - 4 + kFftO - 1) / kFftO; }

int64_t eqa_fft48k5_tile_pitch(int64_t tiles) { return tiles <= 0 ? 0 : (int64_t)fft_pitch((size_t)tiles); }

int eqa_fft48k5_filter_spectra(const float* bank, float* B, int Cout, int Cin, int correlate, void* stream) {
  if (!bank || !B || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (((uintptr_t)B & 7) || Cin > 65535) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fft48_filter_spectra_kernel<5>, dim3(Cin, (Cout + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, bank,
                     B, Cout, Cin, fft_group_in(Cin), correlate ? 1.0f : -1.0f);
  return launch_status();
}

int eqa_fft48k5_filter_spectra3m(const float* bank, float* B3, int Cout, int Cin, int correlate, void* stream) {
  if (!bank || !B3 || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (((uintptr_t)B3 & 15) || Cin % 32 || Cout % 64 || Cin / 4 > 65535) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fft48_filter_spectra3m_kernel<5>, dim3(Cin / 4, (Cout + kThreads - 1) / kThreads, kFftH), dim3(kThreads), 0,
                     (hipStream_t)stream, bank, B3, Cout, Cin, correlate ? 1.0f : -1.0f);
  return launch_status();
}

int eqa_fft48k5_input_grad(const float* Cg, float* T2, float* dx, int nimg, int H, int W, int C, void* stream) {
  if (!Cg || !T2 || !dx || nimg < 0 || H < 5 || W < 5 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  const int OH = H - 4, OW = W - 4;
  const int TY = (OH + kFftO - 1) / kFftO, TX = (OW + kFftO - 1) / kFftO;
  const size_t M = (size_t)nimg * TY * TX;
  if ((size_t)nimg * H > 0x7fffffffULL || M * kFftH > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned cb = (C + kThreads - 1) / kThreads;
  // chunks of images, as in the other two-pass paths (T2 holds 48 rows per tile row here)
  const int chunk = fft_chunk_images(nimg, TY * kFftN, TX, C);
  for (int i0 = 0; i0 < nimg; i0 += chunk) {
    const int n = std::min(chunk, nimg - i0);
    hipLaunchKernelGGL((fft48_cols_inv_kernel<true, kFftO>), dim3((unsigned)((size_t)n * TY * TX * kFftH), cb), dim3(kThreads), 0, st, Cg, T2, OH,
                       C, TY, TX, fft_pitch(M), (size_t)i0 * TY * TX);
    hipLaunchKernelGGL(fft48_rows_inv_add_kernel<kFftO>, dim3((unsigned)((size_t)n * H), cb), dim3(kThreads), 0, st, T2,
                       dx + (size_t)i0 * H * W * C, H, W, C, TY, TX);
  }
  return launch_status();
}

int64_t eqa_fft48k5_workspace_bytes(int nimg, int rows, int cols, int C) {
  if (nimg <= 0 || rows <= 0 || cols <= 0 || C <= 0) return 0;
  const int TX = (cols + kFftO - 1) / kFftO;  // callers pass the OUTPUT width (input width - 4) for either direction
  return (int64_t)fft_chunk_images(nimg, rows, TX, C) * rows * TX * kFftH * 2 * C * (int64_t)sizeof(float);
}

static int fft_forward_impl(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            int TY, int TX, int win, hipStream_t st, int x_grouped = 0) {
  const size_t M = (size_t)nimg * TY * TX;
  if ((size_t)nimg * H * TX > 0x7fffffffULL || M * kFftH > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  static const bool two_pass = getenv("EQA_FFT_TWO_PASS") != nullptr;  // ablation switch: the unfused passes
  if (C % kFusCh == 0 && M * (C / kFusCh) <= 0x7fffffffULL && !two_pass) {
    const bool lds_ok = allow_dynamic_lds((const void*)fft48_fwd_fused_kernel<kFftO>, kFusLds * 4);
    if (lds_ok) {
      const unsigned nwork = (unsigned)(M * (C / kFusCh));
      const size_t xb = (size_t)nimg * H * W * C * 4, vb = (size_t)kFftF * fft_pitch(M) * 2 * C * 4;     // 0 = beyond 32-bit offsets
      // the pipelined form: a plain channel-group-major map, every tile full-width, 32-bit offsets, enough items to keep 256 blocks busy
      const char* pipe_env = getenv("EQA_FFT_FWD_PIPE");      // "1": opt in to the pipelined form (read per call: tests toggle it)
      const bool pipe_off = !(pipe_env != nullptr && pipe_env[0] == '1');
      const bool pipe_lds_ok = allow_dynamic_lds((const void*)fft48_fwd_pipe_kernel, kFusLds * 4);
      const bool full_width = W >= kFftN && (W - kFftN) % kFftO == 0 && TX == (W - kFftN) / kFftO + 1;
      if (!pipe_off && pipe_lds_ok && x_grouped && !in_bias && !in_relu && full_width && xb <= 0xffffe000ULL && vb <= 0xfffffff0ULL &&
          nwork >= 2048 && (size_t)nimg * (C / kFusCh) * H <= 0x7fffffffULL) {
        const unsigned nblk = 256;   // one persistent block per CU
        hipLaunchKernelGGL(fft48_fwd_pipe_kernel, dim3(nblk), dim3(kFwdPipeThreads), kFusLds * sizeof(float), st, x, V, H, W, C, TY, TX,
                           fft_pitch(M), nwork, win, (unsigned)xb, (unsigned)vb);
        return launch_status();
      }
      hipLaunchKernelGGL(fft48_fwd_fused_kernel<kFftO>, dim3(nwork), dim3(kFusThreads), kFusLds * sizeof(float), st, x, V, in_bias, in_relu, H,
                         W, C, TY, TX, fft_pitch(M), nwork, win, xb <= 0xfffffff0ULL ? (unsigned)xb : 0u,
                         vb <= 0xfffffff0ULL ? (unsigned)vb : 0u, x_grouped);
      return launch_status();
    }
    (void)hipGetLastError();
  }
  if (x_grouped) return EQA_ERR_UNSUPPORTED;  // only the fused kernel reads the grouped layout
  const unsigned cb = (C + kThreads - 1) / kThreads;
  const int chunk = fft_chunk_images(nimg, H, TX, C);
  for (int i0 = 0; i0 < nimg; i0 += chunk) {
    const int n = std::min(chunk, nimg - i0);
    hipLaunchKernelGGL(fft48_rows_fwd_kernel<kFftO>, dim3((unsigned)((size_t)n * H * TX), cb), dim3(kThreads), 0, st,
                       x + (size_t)i0 * H * W * C, T, in_bias, in_relu, H, W, C, TX, win);
    hipLaunchKernelGGL(fft48_cols_fwd_kernel<kFftO>, dim3((unsigned)((size_t)n * TY * TX * kFftH), cb), dim3(kThreads), 0, st, T, V, H, C,
                       TY, TX, fft_pitch(M), (size_t)i0 * TY * TX, fft_group_in(C), win);
  }
  return launch_status();
}

int eqa_fft48k5_input(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                      void* stream) {
  if (!x || !T || !V || !fft_dims_ok(nimg, H, W, C)) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  return fft_forward_impl(x, T, V, in_bias, in_relu, nimg, H, W, C, (int)eqa_fft48k5_tiles(H), (int)eqa_fft48k5_tiles(W), kFftN,
                          (hipStream_t)stream);
}

int eqa_fft48k5_input_grouped_supported(int C) {
  static const bool two_pass = getenv("EQA_FFT_TWO_PASS") != nullptr;
  return C > 0 && C % kFusCh == 0 && !two_pass;
}

int eqa_fft48k5_input_grouped(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                              void* stream) {
  if (!x || !T || !V || !fft_dims_ok(nimg, H, W, C)) return EQA_ERR_INVALID_ARG;
  if (!eqa_fft48k5_input_grouped_supported(C)) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  return fft_forward_impl(x, T, V, in_bias, in_relu, nimg, H, W, C, (int)eqa_fft48k5_tiles(H), (int)eqa_fft48k5_tiles(W), kFftN,
                          (hipStream_t)stream, 1);
}

int eqa_fft48k5_grad_transform(const float* dy, float* T, float* G, int nimg, int OH, int OW, int C, void* stream) {
  if (!dy || !T || !G || nimg < 0 || OH <= 0 || OW <= 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  return fft_forward_impl(dy, T, G, nullptr, 0, nimg, OH, OW, C, (OH + kFftO - 1) / kFftO, (OW + kFftO - 1) / kFftO, kFftO,
                          (hipStream_t)stream);
}

int eqa_fft48k5_filter_grad(const float* D, float* dbank, int Cout, int Cin, void* stream) {
  if (!D || !dbank || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (Cin > 65535) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((fft48_filter_grad_kernel<false, 5>), dim3(Cin, (Cout + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, D,
                     dbank, Cout, Cin, fft_group_in(Cin), fft_group_in(Cout));
  return launch_status();
}

int eqa_fft48k5_filter_grad3m(const float* D, float* dbank, int Cout, int Cin, void* stream) {
  if (!D || !dbank || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (Cin > 65535 || Cin % kFusCh || Cout % kFusCh) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((fft48_filter_grad_kernel<true, 5>), dim3(Cin, (Cout + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, D,
                     dbank, Cout, Cin, kFusCh, kFusCh);
  return launch_status();
}

int eqa_fft48k5_output(const float* Mo, float* T2, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                       void* stream) {
  if (!Mo || !T2 || !y || nimg < 0 || OH <= 0 || OW <= 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  int fused = 0;
  const int rc = fft_output_impl<0>(Mo, T2, bias, relu, y, nimg, OH, OW, C, (hipStream_t)stream, &fused);
  return rc != EQA_OK ? rc : launch_status();
}

int64_t eqa_fft48k5_output_stats_rows(int nimg, int OH, int OW, int C) {
  if (nimg <= 0 || OH <= 0 || OW <= 0 || C <= 0) return 0;
  int fused = 0;
  int64_t rows = 0;
  const int rc = fft_output_impl<0>(nullptr, nullptr, nullptr, 0, nullptr, nimg, OH, OW, C, nullptr, &fused, nullptr, &rows);
  return rc == EQA_OK ? rows : 0;
}

int eqa_fft48k5_output_stats(const float* Mo, float* T2, float* y, double* partial, int nimg, int OH, int OW, int C, void* stream) {
  if (!Mo || !T2 || !y || !partial || nimg < 0 || OH <= 0 || OW <= 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  int fused = 0;
  const int rc = fft_output_impl<0>(Mo, T2, nullptr, 0, y, nimg, OH, OW, C, (hipStream_t)stream, &fused, partial);
  return rc != EQA_OK ? rc : launch_status();
}

int eqa_fft48k5_output_sums(const float* Mo, float* T2, const float* bias, int relu, double* S, void* workspace, int nimg,
                            int OH, int OW, int C, int k_next, void* stream) {
  if (!Mo || !T2 || !S || !workspace || nimg < 0 || OH <= 0 || OW <= 0 || C <= 0 || k_next <= 0) return EQA_ERR_INVALID_ARG;
  const int nb = k_next - 1;
  if ((nb != 4 && nb != 2) || OH < 2 * nb + 1 || OW < 2 * nb + 1 || k_next > kMaxWinK || nimg > 65535) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;  // (nimg, OH, TX, C, 1 + 2 nb) floats
  int fused = 0;
  const int rc = nb == 4 ? fft_output_impl<4>(Mo, T2, bias, relu, part, nimg, OH, OW, C, st, &fused)
                         : fft_output_impl<2>(Mo, T2, bias, relu, part, nimg, OH, OW, C, st, &fused);
  if (rc != EQA_OK) return rc;
  // fused path: 2 nb border rows + one segment per tile row, each in `sub` = TX pieces; two-pass path: one per output row
  const int sub = fused ? (OW + kFftO - 1) / kFftO : 1;
  // interior pieces per tile row: 1 (one block per item), 5 (pipeline: one per consumer wave)
  const int per_row = fused == 3 ? kPipeCons / 64 : fused;
  const int nseg = fused ? 2 * nb + per_row * ((OH + kFftO - 1) / kFftO) : OH;
  return eqa::launch_window_sums_nhwc_finalize(part, S, nimg, C, k_next, nseg * sub, st, sub);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------------
// Any odd kernel size 3 .. 9 (round 4): the same overlap-save scheme with 48 x 48 tiles of O = 49 - k outputs.  The forward, gradient
// and output transforms use the fused kernels instantiated for O where C % 16 == 0 (two-pass kernels otherwise and for the input
// gradient); no pipelined inverse and no window-sum epilogue off k = 5: the caller runs eqa_window_sums_nhwc on the map.  The reference's kernel_size is a free constructor argument (escnn_networks.py:19-44): its tutorial trains k = 9, its own test
// k = 3.  The multiply count per output falls with k^2: 1154 x 3 real products per 40 x 40 outputs and channel pair at k = 9 against
// 81 in the direct form (37x fewer; tiles that fit the map badly give some of it back).  The per-frequency channel contraction is
// k-independent: eqa_fft48k5_cgemm3m / _wgrad3m / torch.bmm on buffers of eqa_fft48k5_tile_pitch rows, as for k = 5.
// ------------------------------------------------------------------------------------------------------------------------------
namespace {

template <int KS>
struct FftK {
  static constexpr int O = kFftN + 1 - KS;

  static int forward(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C, int TY, int TX,
                     int win, hipStream_t st) {
    const size_t M = (size_t)nimg * TY * TX;
    if ((size_t)nimg * H * TX > 0x7fffffffULL || M * kFftH > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
    static const bool two_pass = getenv("EQA_FFT_TWO_PASS") != nullptr;
    if (C % kFusCh == 0 && M * (C / kFusCh) <= 0x7fffffffULL && !two_pass &&
        allow_dynamic_lds((const void*)fft48_fwd_fused_kernel<O>, kFusLds * 4)) {
      // row pass, LDS, column pass in one block per (tile, 16 channels): the spectra are written once and nothing is read back
      const unsigned nwork = (unsigned)(M * (C / kFusCh));
      const size_t xb = (size_t)nimg * H * W * C * 4, vb = (size_t)kFftF * fft_pitch(M) * 2 * C * 4;
      hipLaunchKernelGGL(fft48_fwd_fused_kernel<O>, dim3(nwork), dim3(kFusThreads), kFusLds * sizeof(float), st, x, V, in_bias, in_relu, H,
                         W, C, TY, TX, fft_pitch(M), nwork, win, xb <= 0xfffffff0ULL ? (unsigned)xb : 0u,
                         vb <= 0xfffffff0ULL ? (unsigned)vb : 0u, 0);
      return launch_status();
    }
    const unsigned cb = (C + kThreads - 1) / kThreads;
    const int chunk = fft_chunk_images(nimg, H, TX, C);
    for (int i0 = 0; i0 < nimg; i0 += chunk) {
      const int n = std::min(chunk, nimg - i0);
      hipLaunchKernelGGL(fft48_rows_fwd_kernel<O>, dim3((unsigned)((size_t)n * H * TX), cb), dim3(kThreads), 0, st,
                         x + (size_t)i0 * H * W * C, T, in_bias, in_relu, H, W, C, TX, win);
      hipLaunchKernelGGL(fft48_cols_fwd_kernel<O>, dim3((unsigned)((size_t)n * TY * TX * kFftH), cb), dim3(kThreads), 0, st, T, V, H, C,
                         TY, TX, fft_pitch(M), (size_t)i0 * TY * TX, fft_group_in(C), win);
    }
    return launch_status();
  }

  static int output(const float* Mo, float* T2, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C, hipStream_t st) {
    const int TY = (OH + O - 1) / O, TX = (OW + O - 1) / O;
    const size_t M = (size_t)nimg * TY * TX;
    if ((size_t)nimg * OH > 0x7fffffffULL || M * kFftH > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
    static const bool two_pass = getenv("EQA_FFT_TWO_PASS") != nullptr;
    constexpr int lds_bytes = kFftH * (kFftN * 2 * kInvCh + kInvCh) * (int)sizeof(float);
    if (C % kInvCh == 0 && M * (C / kInvCh) <= 0x7fffffffULL && !two_pass &&
        allow_dynamic_lds((const void*)fft48_inv_fused_kernel<0, kInvCh, O>, lds_bytes)) {
      const unsigned nwork = (unsigned)(M * (C / kInvCh));
      const size_t mo_total = (size_t)kFftF * fft_pitch(M) * C * 8;
      hipLaunchKernelGGL((fft48_inv_fused_kernel<0, kInvCh, O>), dim3(nwork), dim3(kFftN * kInvCh), lds_bytes, st, Mo, bias, relu, y, OH, OW, C,
                         TY, TX, fft_pitch(M), nwork, mo_total <= 0xfffffff0ULL ? (unsigned)mo_total : 0u);
      return launch_status();
    }
    const unsigned cb = (C + kThreads - 1) / kThreads;
    const int chunk = fft_chunk_images(nimg, OH, TX, C);
    for (int i0 = 0; i0 < nimg; i0 += chunk) {
      const int n = std::min(chunk, nimg - i0);
      hipLaunchKernelGGL((fft48_cols_inv_kernel<false, O>), dim3((unsigned)((size_t)n * TY * TX * kFftH), cb), dim3(kThreads), 0, st, Mo, T2,
                         OH, C, TY, TX, fft_pitch(M), (size_t)i0 * TY * TX);
      hipLaunchKernelGGL((fft48_rows_inv_kernel<0, O>), dim3((unsigned)((size_t)n * OH), cb), dim3(kThreads), 0, st, T2, bias, relu, y, OH,
                         OW, C, TX, (size_t)i0);
    }
    return launch_status();
  }

  static int input_grad(const float* Cg, float* T2, float* dx, int nimg, int H, int W, int C, hipStream_t st) {
    const int OH = H - (KS - 1), OW = W - (KS - 1);
    const int TY = (OH + O - 1) / O, TX = (OW + O - 1) / O;
    const size_t M = (size_t)nimg * TY * TX;
    if ((size_t)nimg * H > 0x7fffffffULL || M * kFftH > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
    const unsigned cb = (C + kThreads - 1) / kThreads;
    const int chunk = fft_chunk_images(nimg, TY * kFftN, TX, C);
    for (int i0 = 0; i0 < nimg; i0 += chunk) {
      const int n = std::min(chunk, nimg - i0);
      hipLaunchKernelGGL((fft48_cols_inv_kernel<true, O>), dim3((unsigned)((size_t)n * TY * TX * kFftH), cb), dim3(kThreads), 0, st, Cg, T2,
                         OH, C, TY, TX, fft_pitch(M), (size_t)i0 * TY * TX);
      hipLaunchKernelGGL(fft48_rows_inv_add_kernel<O>, dim3((unsigned)((size_t)n * H), cb), dim3(kThreads), 0, st, T2,
                         dx + (size_t)i0 * H * W * C, H, W, C, TY, TX);
    }
    return launch_status();
  }

  static int spectra(const float* bank, float* B, int Cout, int Cin, float sgn, hipStream_t st) {
    hipLaunchKernelGGL(fft48_filter_spectra_kernel<KS>, dim3(Cin, (Cout + kThreads - 1) / kThreads), dim3(kThreads), 0, st, bank, B, Cout,
                       Cin, fft_group_in(Cin), sgn);
    return launch_status();
  }
  static int spectra3m(const float* bank, float* B3, int Cout, int Cin, float sgn, hipStream_t st) {
    hipLaunchKernelGGL(fft48_filter_spectra3m_kernel<KS>, dim3(Cin / 4, (Cout + kThreads - 1) / kThreads, kFftH), dim3(kThreads), 0, st, bank,
                       B3, Cout, Cin, sgn);
    return launch_status();
  }
  static int filter_grad(const float* D, float* dbank, int Cout, int Cin, bool packed, hipStream_t st) {
    // fewer than ~2 blocks per CU in the one-thread-per-filter form: one thread per filter ROW instead
    if ((size_t)Cin * ((Cout + kThreads - 1) / kThreads) < 512) {
      const dim3 grid(Cin, (Cout + kThreads - 1) / kThreads, KS);
      if (packed)
        hipLaunchKernelGGL((fft48_filter_grad_rows_kernel<true, KS>), grid, dim3(kThreads), 0, st, D, dbank, Cout, Cin, kFusCh, kFusCh);
      else
        hipLaunchKernelGGL((fft48_filter_grad_rows_kernel<false, KS>), grid, dim3(kThreads), 0, st, D, dbank, Cout, Cin, fft_group_in(Cin),
                           fft_group_in(Cout));
      return launch_status();
    }
    if (packed)
      hipLaunchKernelGGL((fft48_filter_grad_kernel<true, KS>), dim3(Cin, (Cout + kThreads - 1) / kThreads), dim3(kThreads), 0, st, D, dbank,
                         Cout, Cin, kFusCh, kFusCh);
    else
      hipLaunchKernelGGL((fft48_filter_grad_kernel<false, KS>), dim3(Cin, (Cout + kThreads - 1) / kThreads), dim3(kThreads), 0, st, D, dbank,
                         Cout, Cin, fft_group_in(Cin), fft_group_in(Cout));
    return launch_status();
  }
};

inline bool fft_ksize_ok(int k) { return k == 3 || k == 5 || k == 7 || k == 9; }
inline int fft_tiles_k(int n, int k) { return n < k ? 0 : (n - (k - 1) + (kFftN + 1 - k) - 1) / (kFftN + 1 - k); }

#define EQA_FFT_K(ksize, expr)                 \
  switch (ksize) {                             \
    case 3: { using K_ = FftK<3>; return expr; } \
    case 5: { using K_ = FftK<5>; return expr; } \
    case 7: { using K_ = FftK<7>; return expr; } \
    case 9: { using K_ = FftK<9>; return expr; } \
    default: return EQA_ERR_UNSUPPORTED;       \
  }

}  // namespace

extern "C" {

int eqa_fft48_supported(int ksize) { return fft_ksize_ok(ksize) ? 1 : 0; }

int64_t eqa_fft48_tiles(int n, int ksize) { return fft_ksize_ok(ksize) ? fft_tiles_k(n, ksize) : 0; }

int64_t eqa_fft48_workspace_bytes(int nimg, int rows, int cols, int C, int ksize) {
  if (nimg <= 0 || rows <= 0 || cols <= 0 || C <= 0 || !fft_ksize_ok(ksize)) return 0;
  const int O = kFftN + 1 - ksize;
  const int TX = (cols + O - 1) / O;     // callers pass the OUTPUT width (input width - (k - 1)) for either direction
  return (int64_t)fft_chunk_images(nimg, rows, TX, C) * rows * TX * kFftH * 2 * C * (int64_t)sizeof(float);
}

int eqa_fft48_filter_spectra(const float* bank, float* B, int Cout, int Cin, int ksize, int correlate, void* stream) {
  if (!bank || !B || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (((uintptr_t)B & 7) || Cin > 65535) return EQA_ERR_UNSUPPORTED;
  EQA_FFT_K(ksize, K_::spectra(bank, B, Cout, Cin, correlate ? 1.0f : -1.0f, (hipStream_t)stream));
}

int eqa_fft48_filter_spectra3m(const float* bank, float* B3, int Cout, int Cin, int ksize, int correlate, void* stream) {
  if (!bank || !B3 || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (((uintptr_t)B3 & 15) || Cin % 32 || Cout % 64 || Cin / 4 > 65535) return EQA_ERR_UNSUPPORTED;
  EQA_FFT_K(ksize, K_::spectra3m(bank, B3, Cout, Cin, correlate ? 1.0f : -1.0f, (hipStream_t)stream));
}

int eqa_fft48_input(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C, int ksize,
                    void* stream) {
  if (!x || !T || !V || nimg < 0 || C <= 0 || !fft_ksize_ok(ksize) || H < ksize || W < ksize) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  EQA_FFT_K(ksize, K_::forward(x, T, V, in_bias, in_relu, nimg, H, W, C, fft_tiles_k(H, ksize), fft_tiles_k(W, ksize), kFftN,
                               (hipStream_t)stream));
}

int eqa_fft48_grad_transform(const float* dy, float* T, float* G, int nimg, int OH, int OW, int C, int ksize, void* stream) {
  if (!dy || !T || !G || nimg < 0 || OH <= 0 || OW <= 0 || C <= 0 || !fft_ksize_ok(ksize)) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  EQA_FFT_K(ksize, K_::forward(dy, T, G, nullptr, 0, nimg, OH, OW, C, (OH + K_::O - 1) / K_::O, (OW + K_::O - 1) / K_::O, K_::O,
                               (hipStream_t)stream));
}

int eqa_fft48_output(const float* Mo, float* T2, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C, int ksize,
                     void* stream) {
  if (!Mo || !T2 || !y || nimg < 0 || OH <= 0 || OW <= 0 || C <= 0 || !fft_ksize_ok(ksize)) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  EQA_FFT_K(ksize, K_::output(Mo, T2, bias, relu, y, nimg, OH, OW, C, (hipStream_t)stream));
}

int eqa_fft48_input_grad(const float* Cg, float* T2, float* dx, int nimg, int H, int W, int C, int ksize, void* stream) {
  if (!Cg || !T2 || !dx || nimg < 0 || C <= 0 || !fft_ksize_ok(ksize) || H < ksize || W < ksize) return EQA_ERR_INVALID_ARG;
  if (nimg == 0) return EQA_OK;
  EQA_FFT_K(ksize, K_::input_grad(Cg, T2, dx, nimg, H, W, C, (hipStream_t)stream));
}

int eqa_fft48_filter_grad(const float* D, float* dbank, int Cout, int Cin, int ksize, int packed, void* stream) {
  if (!D || !dbank || Cout <= 0 || Cin <= 0) return EQA_ERR_INVALID_ARG;
  if (Cin > 65535 || (packed && (Cin % kFusCh || Cout % kFusCh))) return EQA_ERR_UNSUPPORTED;
  EQA_FFT_K(ksize, K_::filter_grad(D, dbank, Cout, Cin, packed != 0, (hipStream_t)stream));
}

}  // extern "C"
