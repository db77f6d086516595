// Issue rate of v_mfma_f32_32x32x2_f32 with 1, 2 and 4 independent accumulator chains, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 80 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 1.2345f) out[0] = s;
}

template <int CHAINS>
void run(int blocks, const char* what) {
  float* d;
  hipMalloc(&d, 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = (double)iters * (80 / CHAINS) * CHAINS;
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  const double flops = mfma_per_wave * blocks * 4 * 4096.0;
  printf("%-28s chains %d: %7.3f ms  %6.1f TFLOP/s  -> %5.1f ns per MFMA per SIMD (= 64 cycles at %.2f GHz)\n", what, CHAINS, ms,
         flops / ms / 1e9, ms * 1e6 / (mfma_per_wave * waves_per_simd), 64.0 / (ms * 1e6 / (mfma_per_wave * waves_per_simd)));
  hipFree(d);
}

int main() {
  run<1>(256, "1 wave/SIMD"); run<2>(256, "1 wave/SIMD"); run<4>(256, "1 wave/SIMD");
  run<1>(512, "2 waves/SIMD"); run<2>(512, "2 waves/SIMD"); run<4>(512, "2 waves/SIMD");
  return 0;
}
