// libeqa_hip.so, part 11 -- the strided convolutions of ConvNetwork (row I10: the encoder the optimised canonicalizer scores its
// G group views with) in inference, on the fp32 matrix cores.  C ABI: include/eqa_hip.h.
//
// Reference: equiadapt/images/canonicalization_networks/custom_nonequivariant_networks.py:44-57 -- Conv2d(k, stride 2, padding 0 or
// 1) -> BatchNorm2d -> GELU per layer, 16-32 channels on 128 x 128 inputs: 7-20 MFLOP per image and layer.  The vendor library
// runs them at 20-26 TFLOP/s (0.21 / 0.19 / 0.06 ms for the three layers of the segmentation config at 256 views); they are
// an implicit GEMM with N = 16 or 32 -- exactly the N of v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: an fmaf chain, no
// reduced precision).
//
// As in cgemm3m.hip the matrix pipe is slow enough (32 cycles per 16x16x4 MFMA and SIMD) for the operands to come straight
// from global memory / L2 into registers: no LDS, no barriers, every wave on its own.  A wave owns 4 tiles of 16 consecutive
// output pixels (any 16 of the flattened (image, y, x) index: every lane keeps its own pixel's input offset) x all output channels.
// k-slot mapping of the MFMA (A[i = lane & 15][k = lane >> 4]):
//   NHWC input, Cin % 16 == 0: lane (i, kq) loads 16 bytes = channels 4 kq .. 4 kq + 3 of tap (u, v) of its pixel; MFMA s of the
//        tap's 16-channel chunk multiplies the channels {s, 4 + s, 8 + s, 12 + s}; the weights are packed to match
//        (wp[tap][chunk][n][kq][j][s]), a lane's B operand for the 4 MFMAs is one 16-byte load.
//   planar input with Cin <= 4 (the first layer: the orbit kernel's NCHW views): k-slot kq = input channel (zero weights for
//        kq >= Cin), one 4-byte load and one MFMA per tap.
// Epilogue: + bias (the eval-mode batch-norm is folded into weights and bias by the caller), exact GELU (erf), NHWC store: the
// accumulator layout (lane: channel j, registers: pixels 4 rq .. 4 rq + 3) gives 64-byte runs per pixel.
// Padding 1: taps that fall outside the image get an out-of-range buffer offset, which the hardware reads as zero.
#include "eqa_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTiles = 2;  // pixel tiles per wave (swept 2 / 4 / 8 at the tutorial's shapes: 77 / 81 / 95 us at the second layer, 29 / 37 / 37 at the third)

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int NCO>
__device__ __forceinline__ void conv_epilogue(const f32x4 (&acc)[kTiles][NCO], const float* __restrict__ bias, int gelu,
                                              float* __restrict__ y, long p0, long P, int Cout, int lane) {
  const int j = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int n = 0; n < NCO; ++n) {
    const float b = bias ? bias[16 * n + j] : 0.0f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long p = p0 + 16 * t + 4 * rq + r;
        float v = acc[t][n][r] + b;
        v = gelu ? gelu_erf(v) : v;
        if (p < P) y[p * Cout + 16 * n + j] = v;
      }
  }
}

// NHWC input, Cin = 16 * CHUNKS.  x:(B,H,W,Cin) -> y:(B,OH,OW,16 NCO); wp:(K*K, CHUNKS, NCO, 4 [kq], 16 [j], 4 [s])
template <int K, int CHUNKS, int NCO, int PAD>
__global__ __launch_bounds__(kThreads) void conv_s2_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, int gelu, float* __restrict__ y, int H,
                                                                int W, int OH, int OW, long P, unsigned x_bytes) {
  constexpr int Cin = 16 * CHUNKS;
  const int lane = threadIdx.x & 63;
  const long p0 = ((long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * (16 * kTiles);
  if (p0 >= P) return;
  const int i = lane & 15, kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  // The matrix instruction wants lane 16 kq + i to hold pixel i's channels 4 kq .. 4 kq + 3 -- neighbouring lanes on neighbouring
  // PIXELS, 16 bytes each from a different 64-byte request: loaded that way the kernel ran at a quarter of the vector L1's rate
  // (66 us of loads against 34 us of matrix instructions at the tutorial's second layer).  So the LOAD uses the other order --
  // lane 4 p + q fetches pixel p's channel quad q: four neighbouring lanes = one pixel's 64 contiguous bytes -- and four
  // ds_bpermute per loaded vector bring lane 16 kq + i the data of lane 4 i + kq.
  const int lp = lane >> 2, lq = lane & 3;
  const int perm = 4 * (4 * i + kq);                                   // byte address of the source lane
  // per tile: byte offset of tap (0, 0) of the LOAD lane's pixel (+ its 4 channels), and for PAD > 0 which taps exist
  unsigned base[kTiles];
  int oy2[kTiles], ox2[kTiles];
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    const unsigned p = (unsigned)min(p0 + 16 * t + lp, P - 1);        // 32-bit divisions (P < 2^31: checked by the host)
    const unsigned row = p / (unsigned)OW;
    const int ox = (int)(p - row * (unsigned)OW);
    const long b = row / (unsigned)OH;
    const int oy = (int)(row - (unsigned)b * (unsigned)OH);
    oy2[t] = 2 * oy - PAD;
    ox2[t] = 2 * ox - PAD;
    base[t] = (unsigned)((((b * H + oy2[t]) * W + ox2[t]) * Cin + 4 * lq) * 4);
  }
  f32x4 acc[kTiles][NCO];
#pragma unroll
  for (int t = 0; t < kTiles; ++t)
#pragma unroll
    for (int n = 0; n < NCO; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* wl = reinterpret_cast<const f32x4*>(wp) + lane;     // [tap][chunk][n] x 64 lanes
#pragma unroll 1
  for (int u = 0; u < K; ++u) {
#pragma unroll
    for (int v = 0; v < K; ++v) {
#pragma unroll
      for (int c = 0; c < CHUNKS; ++c) {
        f32x4 a[kTiles], b[NCO];
        const unsigned tap_off = (unsigned)(((u * W + v) * Cin + 16 * c) * 4);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
          unsigned off = base[t] + tap_off;
          if (PAD > 0) {
            const int iy = oy2[t] + u, ix = ox2[t] + v;
            off = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? off : 0x7ffffff0u;     // outside the buffer: reads as zero
          }
          const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
#pragma unroll
          for (int s = 0; s < 4; ++s) a[t][s] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm, (int)raw[s]));
        }
#pragma unroll
        for (int n = 0; n < NCO; ++n) b[n] = wl[(((u * K + v) * CHUNKS + c) * NCO + n) * 64];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < kTiles; ++t)
#pragma unroll
            for (int n = 0; n < NCO; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][s], b[n][s], acc[t][n], 0, 0, 0);
      }
    }
  }
  conv_epilogue<NCO>(acc, bias, gelu, y, p0, P, 16 * NCO, lane);
}

// planar input, Cin <= 4.  x:(B,Cin,H,W) -> y:(B,OH,OW,16 NCO); wp:(K*K, NCO, 4 [kq], 16 [j]), zero for kq >= Cin
template <int K, int NCO, int PAD>
__global__ __launch_bounds__(kThreads) void conv_s2_planar_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                  const float* __restrict__ bias, int gelu, float* __restrict__ y, int Cin,
                                                                  int H, int W, int OH, int OW, long P, unsigned x_bytes) {
  const int lane = threadIdx.x & 63;
  const long p0 = ((long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * (16 * kTiles);
  if (p0 >= P) return;
  const int i = lane & 15, kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  const bool has_ch = kq < Cin;
  unsigned base[kTiles];
  int oy2[kTiles], ox2[kTiles];
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    const unsigned p = (unsigned)min(p0 + 16 * t + i, P - 1);
    const unsigned row = p / (unsigned)OW;
    const int ox = (int)(p - row * (unsigned)OW);
    const long b = row / (unsigned)OH;
    const int oy = (int)(row - (unsigned)b * (unsigned)OH);
    oy2[t] = 2 * oy - PAD;
    ox2[t] = 2 * ox - PAD;
    base[t] = has_ch ? (unsigned)((((b * Cin + kq) * H + oy2[t]) * W + ox2[t]) * 4) : 0x7ffffff0u;   // missing channel: zero
  }
  f32x4 acc[kTiles][NCO];
#pragma unroll
  for (int t = 0; t < kTiles; ++t)
#pragma unroll
    for (int n = 0; n < NCO; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wl = wp + lane;
  if (PAD == 0) {
    // Without padding the K taps of a filter row are K consecutive floats of the plane: one 16-byte load + one 4-byte load (K = 5),
    // two 16-byte loads (K = 7) per tile and filter row instead of K dword gathers -- the kernel was bound by the number of wave loads the vector L1
    // takes (25 x 4 dword gathers per wave at K = 5: 58 of its 117 us), not by bytes.
#pragma unroll      // all filter rows in one scheduling region: the next row's loads go out behind this row's matrix instructions (105 -> 83 us)
    for (int u = 0; u < K; ++u) {
      float a[kTiles][K];
#pragma unroll
      for (int t = 0; t < kTiles; ++t) {
        const unsigned off = has_ch ? base[t] + (unsigned)(u * W * 4) : 0x7ffffff0u;
        if (K >= 4) {
          const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
#pragma unroll
          for (int v = 0; v < 4; ++v) a[t][v] = q[v];
        }
        if (K == 7) {                                  // taps 3 .. 6: a second 16-byte load ending exactly at the row's last tap
          const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off + 12, 0, 0));
#pragma unroll
          for (int v = 4; v < 7; ++v) a[t][v] = q[v - 3];
        } else if (K == 5) {
          a[t][4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off + 16, 0, 0));
        } else {
#pragma unroll
          for (int v = 0; v < K; ++v) a[t][v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off + 4 * v, 0, 0));
        }
      }
#pragma unroll
      for (int v = 0; v < K; ++v) {
        float b[NCO];
#pragma unroll
        for (int n = 0; n < NCO; ++n) b[n] = wl[((u * K + v) * NCO + n) * 64];
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
#pragma unroll
          for (int n = 0; n < NCO; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][v], b[n], acc[t][n], 0, 0, 0);
      }
    }
  } else {
#pragma unroll 1
    for (int u = 0; u < K; ++u) {
#pragma unroll
      for (int v = 0; v < K; ++v) {
        float a[kTiles], b[NCO];
        const unsigned tap_off = (unsigned)((u * W + v) * 4);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) {
          unsigned off = base[t] + tap_off;
          const int iy = oy2[t] + u, ix = ox2[t] + v;
          off = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? off : 0x7ffffff0u;
          off = has_ch ? off : 0x7ffffff0u;
          a[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
        }
#pragma unroll
        for (int n = 0; n < NCO; ++n) b[n] = wl[((u * K + v) * NCO + n) * 64];
#pragma unroll
        for (int t = 0; t < kTiles; ++t)
#pragma unroll
          for (int n = 0; n < NCO; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[n], acc[t][n], 0, 0, 0);
      }
    }
  }
  conv_epilogue<NCO>(acc, bias, gelu, y, p0, P, 16 * NCO, lane);
}

template <int K, int PAD>
int launch_nhwc(const float* x, const float* wp, const float* bias, int gelu, float* y, int Cin, int Cout, int H, int W, int OH, int OW,
                long P, unsigned xb, hipStream_t st) {
  const dim3 grid((unsigned)((P + 16 * kTiles * (kThreads / 64) - 1) / (16 * kTiles * (kThreads / 64))));
#define EQA_SC(CH, NC)                                                                                                                 \
  hipLaunchKernelGGL((conv_s2_nhwc_kernel<K, CH, NC, PAD>), grid, dim3(kThreads), 0, st, x, wp, bias, gelu, y, H, W, OH, OW, P, xb)
  const int ch = Cin / 16, nc = Cout / 16;
  if (ch == 1 && nc == 1) EQA_SC(1, 1);
  else if (ch == 1 && nc == 2) EQA_SC(1, 2);
  else if (ch == 2 && nc == 2) EQA_SC(2, 2);
  else if (ch == 2 && nc == 4) EQA_SC(2, 4);
  else if (ch == 4 && nc == 4) EQA_SC(4, 4);
  else return EQA_ERR_UNSUPPORTED;
#undef EQA_SC
  return launch_status();
}

template <int K, int PAD>
int launch_planar(const float* x, const float* wp, const float* bias, int gelu, float* y, int Cin, int Cout, int H, int W, int OH, int OW,
                  long P, unsigned xb, hipStream_t st) {
  const dim3 grid((unsigned)((P + 16 * kTiles * (kThreads / 64) - 1) / (16 * kTiles * (kThreads / 64))));
  if (Cout == 16)
    hipLaunchKernelGGL((conv_s2_planar_kernel<K, 1, PAD>), grid, dim3(kThreads), 0, st, x, wp, bias, gelu, y, Cin, H, W, OH, OW, P, xb);
  else if (Cout == 32)
    hipLaunchKernelGGL((conv_s2_planar_kernel<K, 2, PAD>), grid, dim3(kThreads), 0, st, x, wp, bias, gelu, y, Cin, H, W, OH, OW, P, xb);
  else
    return EQA_ERR_UNSUPPORTED;
  return launch_status();
}

}  // namespace

extern "C" {

int eqa_conv_s2_supported(int Cin, int Cout, int K, int pad, int planar) {
  if (K != 3 && K != 5 && K != 7) return 0;
  if (pad != 0 && pad != 1) return 0;
  if (planar) return Cin >= 1 && Cin <= 4 && (Cout == 16 || Cout == 32);
  const int ch = Cin / 16, nc = Cout / 16;
  if (Cin % 16 || Cout % 16) return 0;
  return (ch == 1 && (nc == 1 || nc == 2)) || (ch == 2 && (nc == 2 || nc == 4)) || (ch == 4 && nc == 4);
}

int eqa_conv_s2(const float* x, const float* wp, const float* bias, int gelu, float* y, int B, int Cin, int H, int W, int Cout, int K,
                int pad, int planar, void* stream) {
  if (!x || !wp || !y || B < 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return EQA_ERR_INVALID_ARG;
  if (!eqa_conv_s2_supported(Cin, Cout, K, pad, planar)) return EQA_ERR_UNSUPPORTED;
  if (H + 2 * pad < K || W + 2 * pad < K) return EQA_ERR_INVALID_ARG;  // frame smaller than the kernel (torch raises too)
  const int OH = (H + 2 * pad - K) / 2 + 1, OW = (W + 2 * pad - K) / 2 + 1;
  if (B == 0) return EQA_OK;
  const size_t xbytes = (size_t)B * Cin * H * W * 4;
  if (xbytes > 0x7fffffe0ULL || (((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y) & 15)) return EQA_ERR_UNSUPPORTED;
  const long P = (long)B * OH * OW;
  if (P > 0x7fffffffL) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const unsigned xb = (unsigned)xbytes;
#define EQA_SC_K(KK)                                                                                                                  \
  return planar ? (pad ? launch_planar<KK, 1>(x, wp, bias, gelu, y, Cin, Cout, H, W, OH, OW, P, xb, st)                              \
                       : launch_planar<KK, 0>(x, wp, bias, gelu, y, Cin, Cout, H, W, OH, OW, P, xb, st))                             \
                : (pad ? launch_nhwc<KK, 1>(x, wp, bias, gelu, y, Cin, Cout, H, W, OH, OW, P, xb, st)                                \
                       : launch_nhwc<KK, 0>(x, wp, bias, gelu, y, Cin, Cout, H, W, OH, OW, P, xb, st))
  if (K == 7) { EQA_SC_K(7); }
  if (K == 5) { EQA_SC_K(5); }
  EQA_SC_K(3);
#undef EQA_SC_K
}

}  // extern "C"
