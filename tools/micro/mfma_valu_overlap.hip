// Do matrix instructions and vector-ALU instructions of DIFFERENT waves overlap on one SIMD?  Blocks of 8 waves (one per CU): waves
// 0..3 (one per SIMD) run a matrix-instruction stream, waves 4..7 (their SIMD partners) a stream of independent v_fma_f32; each alone
// and both together.  fp32 matrix instruction (v_mfma_f32_16x16x4_f32) against the bf16 one (v_mfma_f32_16x16x32_bf16).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o tools/micro/_bin/mfma_valu_overlap && tools/micro/_bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ unsigned long long g_cyc[4];

template <int KIND>   // 0: fp32 16x16x4, 1: bf16 16x16x32
__global__ __launch_bounds__(512) void k(float* out, int rounds, int run_mfma, int run_valu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0;
  if (wave < 4) {
    if (!run_mfma) return;
    f32x4 a[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float w = 0.01f * lane, b = 0.02f * lane;
    bf16x8 wb, bb;
    for (int i = 0; i < 8; ++i) { wb[i] = (__bf16)(0.01f * (lane + i)); bb[i] = (__bf16)(0.02f * (lane + i)); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
      for (int t = 0; t < 20; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          a[j] = KIND == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, bb, a[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int j = 0; j < 3; ++j) s += a[j][0] + a[j][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc[0] = t1 - t0;
  } else {
    if (!run_valu) return;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (lane + i);
    const float m = 1.0001f, c = 0.0003f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
      for (int t = 0; t < 30; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 16; ++i) s += v[i];
    if (blockIdx.x == 0 && threadIdx.x == 256) g_cyc[1] = t1 - t0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(float* out, int m, int v, const char* what) {
  const int rounds = 1000;
  unsigned long long z[4] = {0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), z, sizeof(z));
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, 10, m, v);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, out, rounds, m, v);
  (void)hipDeviceSynchronize();
  unsigned long long cyc[4]; (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
  printf("%-46s", what);
  if (m) printf("  matrix wave: %6.1f ticks per instruction", cyc[0] / (rounds * 60.0));
  if (v) printf("  vector wave: %5.2f ticks per v_fma_f32", cyc[1] / (rounds * 480.0));
  printf("\n");
}
int main() {
  float* out; if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
  run<0>(out, 1, 0, "fp32 16x16x4 alone");
  run<0>(out, 0, 1, "v_fma_f32 alone");
  run<0>(out, 1, 1, "fp32 16x16x4 + v_fma_f32 on the SIMD partner");
  run<1>(out, 1, 0, "bf16 16x16x32 alone");
  run<1>(out, 1, 1, "bf16 16x16x32 + v_fma_f32 on the SIMD partner");
  return 0;
}
