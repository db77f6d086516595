// Micro-benchmark: the generated 48-point transforms on float (one transform per lane) against a 2-vector of floats (two
// independent transforms per lane on v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Same number of transforms in both launches.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -I equiadapt_amd/csrc tools/micro/fft_pk.hip -o build_variants/fft_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#include "fft48.inc"

template <typename T>
__global__ __launch_bounds__(256) void fft_loop(const T* __restrict__ in, T* __restrict__ out, int reps, int which) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  T re[48], im[48], ore[48], oim[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) { re[i] = in[(t * 96 + i)]; im[i] = in[(t * 96 + 48 + i)]; }
  for (int r = 0; r < reps; ++r) {
    if (which == 0) {
      fft48(re, im, ore, oim);
#pragma unroll
      for (int i = 0; i < 48; ++i) { re[i] = ore[i] * 0.02f; im[i] = oim[i] * 0.02f; }
    } else {
      T hr[25], hi[25];
#pragma unroll
      for (int i = 0; i < 25; ++i) { hr[i] = re[i]; hi[i] = im[i]; }
      ifft48_c2r(hr, hi, ore);
#pragma unroll
      for (int i = 0; i < 48; ++i) re[i] = ore[i] * 0.02f;
    }
  }
#pragma unroll
  for (int i = 0; i < 48; ++i) { out[(t * 96 + i)] = re[i]; out[(t * 96 + 48 + i)] = im[i]; }
}

int main() {
  const int transforms = 256 * 1024;  // scalar threads; the 2-vector launch uses half as many
  const int reps = 256;
  float *in, *out;
  hipMalloc(&in, (size_t)transforms * 96 * 4);
  hipMalloc(&out, (size_t)transforms * 96 * 4);
  std::vector<float> h((size_t)transforms * 96);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int vec = 0; vec < 2; ++vec) {
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        if (vec == 0)
          hipLaunchKernelGGL(fft_loop<float>, dim3(transforms / 256), dim3(256), 0, 0, in, out, reps, which);
        else
          hipLaunchKernelGGL(fft_loop<f32x2>, dim3(transforms / 512), dim3(256), 0, 0, (const f32x2*)in, (f32x2*)out, reps, which);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      const double ops = (which == 0 ? 819.0 + 96 : 468.0 + 48) * (double)transforms * reps;
      printf("%s %s: %.3f ms  %.2f T lane-op/s\n", which == 0 ? "fft48     " : "ifft48_c2r", vec ? "2-vector" : "scalar  ", best, ops / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
