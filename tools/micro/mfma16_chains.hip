#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
__device__ unsigned long long g_cyc[4];
template <int NCH, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int rounds) {
  const int lane = threadIdx.x & 63;
  h8 a[5], b[5]; b8 ab[5], bb[5];
  for (int i = 0; i < 5; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(0.01f * (i + j + lane)); b[i][j] = (_Float16)(0.02f * (i + j)); ab[i][j] = (__bf16)(0.01f * (i + j + lane)); bb[i][j] = (__bf16)(0.02f * (i + j)); }
  f32x4 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (BF) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bb[(s + c) % 5], acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[(s + c) % 5], acc[c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0; for (int c = 0; c < NCH; ++c) sum += acc[c][0] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc[0] = t1 - t0;
}
template <int NCH, bool BF> void run(const char* nm, float* d) {
  const int rounds = 2000;
  k<NCH, BF><<<256, 256>>>(d, rounds);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_cyc), 8);
  printf("%-40s %d chains: %.1f cycles per instruction\n", nm, NCH, (double)c / (rounds * 5.0 * NCH));
}
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4);
  run<1, false>("v_mfma_f32_16x16x32_f16", d); run<2, false>("v_mfma_f32_16x16x32_f16", d); run<3, false>("v_mfma_f32_16x16x32_f16", d);
  run<4, false>("v_mfma_f32_16x16x32_f16", d); run<6, false>("v_mfma_f32_16x16x32_f16", d);
  run<1, true>("v_mfma_f32_16x16x32_bf16", d); run<3, true>("v_mfma_f32_16x16x32_bf16", d); run<6, true>("v_mfma_f32_16x16x32_bf16", d);
  return 0;
}
