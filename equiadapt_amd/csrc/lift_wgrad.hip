// Filter gradient of the lifting convolution (I2a training, escnn_networks.py:48-66 -> e2cnn R2Conv(3 -> 32 x 8, k5); the
// reference gets it from autograd's convolution-weight-gradient).  fp32 on the matrix cores, deterministic.
//
//   dW[co][ci][ky][kx] = sum over (n, oy, ox) of dy[n][oy][ox][co] * x[n][oy+ky][ox+kx][ci]       (both channels-last)
//
// As a GEMM the reduction runs over the 2.2 M output positions with a 80 x 256 result (R = KH*KW*Cin = 75 rows, padded to
// 5 tiles of 16): 83 GFLOP -> 0.65 ms of v_mfma_f32_16x16x4_f32 at the headline shape, beside 2.2 GB of dy read once
// (0.45 ms at 5 TB/s).  MIOpen's weight-gradient solvers take 3.1 ms for it (two igemm_wrw launches).
//
// Layout of the work: a block = 4 waves, one per SIMD, wave w owning output channels [64 (4 cb + w), +64).  The block walks
// units (n, oy) = one output row; the KH input rows under it (one contiguous run of KH*W*Cin floats) are staged in LDS,
// double buffered, next unit's run and next unit's dy registers requested before the current unit's MFMAs.  Per k-step
// (4 consecutive ox) a lane loads ONE 16-byte piece of dy -- channels 4 n .. 4 n + 3 of position 4 s + lane / 16 -- which
// is the B operand of 4 MFMAs (column n of channel tile j holds channel 4 n + j), and reads 5 patch values from LDS (A
// operand: row i of tile t is filter tap r = 16 t + i, at x-run offset ky * W * Cin + Cin * ox + r % (KW * Cin); taps
// r >= R read a zeroed strip).  20 MFMAs (640 cycles) per 1 global load and 5 LDS reads.
// Every wave leaves its partial (80 x 64) in `part`; lift_wgrad_reduce_kernel adds the partials of a channel group in a
// fixed order and writes the bank in the framework's (Cout, Cin, KH, KW) order.
#include "eqa_common.hpp"

namespace {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kWgThreads = 256;
constexpr int kWgRT = 5;            // 16-row tiles of the tap index
constexpr int kWgRun = 2048;        // floats per staged input run (KH * W * Cin), at most
constexpr int kWgZero = 512;        // zeroed strip for the padding taps (>= 4 * Cin * KS)
constexpr int kWgXr = kWgRun / kWgThreads;

template <int KS>
__global__ __launch_bounds__(kWgThreads) void lift_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ part, int N, int H, int W, int Cin, int OH,
                                                                     int Cout, int KH, int KW, int ncb) {
  __shared__ float xs[2 * kWgRun + kWgZero];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x % ncb, bi = blockIdx.x / ncb, nb = gridDim.x / ncb;
  const int OW = 4 * KS;
  const int R = KH * KW * Cin, rowf = W * Cin, run = KH * rowf;
  const int units = N * OH;
  for (int i = tid; i < kWgZero; i += kWgThreads) xs[2 * kWgRun + i] = 0.0f;
  // A operand: this lane's tap per tile
  const int li = lane & 15, kk = lane >> 4;
  int a0[kWgRT], a1[kWgRT];
#pragma unroll
  for (int t = 0; t < kWgRT; ++t) {
    const int r = 16 * t + li;
    const int off = (r / (KW * Cin)) * rowf + Cin * kk + r % (KW * Cin);
    a0[t] = r < R ? off : 2 * kWgRun;
    a1[t] = r < R ? kWgRun + off : 2 * kWgRun;
  }
  const int co = 64 * (4 * cb + wave) + 4 * li;  // first of this lane's 4 channels
  wg_f32x4 acc[kWgRT][4];
#pragma unroll
  for (int t = 0; t < kWgRT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = wg_f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  auto x_base = [&](int u) { return ((size_t)(u / OH) * H + (u % OH)) * (size_t)rowf; };
  auto dy_base = [&](int u) { return ((size_t)u * OW + kk) * (size_t)Cout + co; };
  float xr[kWgXr];
  wg_f32x4 dc[KS], dn[KS];
  int u = bi;
  if (u < units) {
    const float* px = x + x_base(u);
#pragma unroll
    for (int i = 0; i < kWgXr; ++i) xr[i] = tid + kWgThreads * i < run ? px[tid + kWgThreads * i] : 0.0f;
    const float* pd = dy + dy_base(u);
#pragma unroll
    for (int s = 0; s < KS; ++s) dc[s] = *reinterpret_cast<const wg_f32x4*>(pd + (size_t)4 * s * Cout);
#pragma unroll
    for (int i = 0; i < kWgXr; ++i) xs[tid + kWgThreads * i] = xr[i];
  }
  __syncthreads();
  int cur = 0;
  for (; u < units; u += nb) {
    const int un = u + nb < units ? u + nb : u;  // the last unit is requested again: no branch around the loads
    const float* px = x + x_base(un);
#pragma unroll
    for (int i = 0; i < kWgXr; ++i) xr[i] = tid + kWgThreads * i < run ? px[tid + kWgThreads * i] : 0.0f;
    const float* pd = dy + dy_base(un);
#pragma unroll
    for (int s = 0; s < KS; ++s) dn[s] = *reinterpret_cast<const wg_f32x4*>(pd + (size_t)4 * s * Cout);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float a[kWgRT];
#pragma unroll
      for (int t = 0; t < kWgRT; ++t) a[t] = xs[(cur ? a1[t] : a0[t]) + 4 * Cin * s];
#pragma unroll
      for (int t = 0; t < kWgRT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], dc[s][j], acc[t][j], 0, 0, 0);
    }
    float* nx = xs + (cur ? 0 : kWgRun);
#pragma unroll
    for (int i = 0; i < kWgXr; ++i) nx[tid + kWgThreads * i] = xr[i];
#pragma unroll
    for (int s = 0; s < KS; ++s) dc[s] = dn[s];
    cur ^= 1;
    __syncthreads();
  }
  // partial of this wave: (80 taps, 64 channels), channel 4 n + j of the group in column n of tile j
  float* o = part + (((size_t)(4 * cb + wave) * nb + bi) * (16 * kWgRT)) * 64 + 4 * li;
#pragma unroll
  for (int t = 0; t < kWgRT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 16 * t + 4 * kk + e;
      *reinterpret_cast<wg_f32x4*>(o + (size_t)r * 64) = wg_f32x4{acc[t][0][e], acc[t][1][e], acc[t][2][e], acc[t][3][e]};
    }
}

// dbank (Cout, Cin, KH, KW) = the partials of each 64-channel group added in block order.
__global__ __launch_bounds__(kThreads) void lift_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbank, int nb,
                                                                    int Cin, int Cout, int KH, int KW) {
  const int R = KH * KW * Cin;
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= R * Cout) return;
  const int co = idx % Cout, r = idx / Cout;
  const float* p = part + ((size_t)(co / 64) * nb * (16 * kWgRT) + r) * 64 + co % 64;
  float s = 0.0f;
  for (int b = 0; b < nb; ++b) s += p[(size_t)b * (16 * kWgRT) * 64];
  const int ci = r % Cin, kx = (r / Cin) % KW, ky = r / (Cin * KW);
  dbank[(((size_t)co * Cin + ci) * KH + ky) * KW + kx] = s;
}

int wg_blocks(int units, int ncb) {
  return std::max(1, std::min(units, 256 / ncb));  // one 4-wave block per CU (the 23-step variant needs > 256 registers)
}

template <int KS>
void wg_launch(const float* x, const float* dy, float* part, int N, int H, int W, int Cin, int OH, int Cout, int KH, int KW, int ncb,
               int nb, hipStream_t st) {
  hipLaunchKernelGGL(lift_wgrad_mfma_kernel<KS>, dim3(nb * ncb), dim3(kWgThreads), 0, st, x, dy, part, N, H, W, Cin, OH, Cout, KH, KW,
                     ncb);
}

}  // namespace

extern "C" {

int eqa_lift_conv_wgrad_supported(int N, int H, int W, int Cin, int Cout, int KH, int KW) {
  if (N <= 0 || H < KH || W < KW || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0) return 0;
  const int OW = W - KW + 1, KS = OW / 4;
  if (OW % 4 || Cout % 256 || KH * KW * Cin > 16 * kWgRT || KH * W * Cin > kWgRun || 4 * Cin * KS > kWgZero) return 0;
  if ((size_t)N * (H - KH + 1) > 0x7fffffffULL) return 0;
  return KS == 7 || KS == 11 || KS == 15 || KS == 23 || KS == 31;
}

int64_t eqa_lift_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW) {
  if (!eqa_lift_conv_wgrad_supported(N, H, W, Cin, Cout, KH, KW)) return 0;
  const int ncb = Cout / 256;
  return (int64_t)(Cout / 64) * wg_blocks(N * (H - KH + 1), ncb) * (16 * kWgRT) * 64 * (int64_t)sizeof(float);
}

int eqa_lift_conv_wgrad_nhwc(const float* x, const float* dy, void* workspace, float* dbank, int N, int H, int W, int Cin, int Cout,
                             int KH, int KW, void* stream) {
  if (!x || !dy || !workspace || !dbank) return EQA_ERR_INVALID_ARG;
  if (!eqa_lift_conv_wgrad_supported(N, H, W, Cin, Cout, KH, KW)) return EQA_ERR_UNSUPPORTED;
  if (((uintptr_t)dy & 15) || ((uintptr_t)workspace & 15)) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int OH = H - KH + 1, KS = (W - KW + 1) / 4, ncb = Cout / 256;
  const int nb = wg_blocks(N * OH, ncb);
  float* part = (float*)workspace;
  switch (KS) {
    case 7: wg_launch<7>(x, dy, part, N, H, W, Cin, OH, Cout, KH, KW, ncb, nb, st); break;
    case 11: wg_launch<11>(x, dy, part, N, H, W, Cin, OH, Cout, KH, KW, ncb, nb, st); break;
    case 15: wg_launch<15>(x, dy, part, N, H, W, Cin, OH, Cout, KH, KW, ncb, nb, st); break;
    case 23: wg_launch<23>(x, dy, part, N, H, W, Cin, OH, Cout, KH, KW, ncb, nb, st); break;
    default: wg_launch<31>(x, dy, part, N, H, W, Cin, OH, Cout, KH, KW, ncb, nb, st); break;
  }
  const int total = KH * KW * Cin * Cout;
  hipLaunchKernelGGL(lift_wgrad_reduce_kernel, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, part, dbank, nb, Cin,
                     Cout, KH, KW);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

}  // extern "C"
