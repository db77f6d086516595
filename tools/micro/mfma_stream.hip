// Why does a lone wave's stream of v_mfma_f32_16x16x4_f32 in csrc/lift_fft.hip run at 46-62 cycles per instruction when the pipe takes one
// per 32?  One block of W waves per CU (256 blocks), each wave runs R rounds of 57 matrix instructions in the kernel's order (19 steps
// x 3 independent accumulators) under variants; cycles per instruction from s_memtime on wave 0.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_stream.hip -o tools/micro/_bin/mfma_stream && tools/micro/_bin/mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_cyc[8];

// MODE 0: operands in registers, 3 accumulators step-major (16x16x4)
// MODE 1: same + one ds_read_b32 per instruction interleaved 5 steps ahead (as in lift_fft.hip)
// MODE 2: 32x32x2, 2 accumulators step-major, operands in registers (the unfused lifting kernel's instruction)
// MODE 3: 16x16x4, ONE accumulator chain (every instruction depends on its predecessor)
// MODE 4: 16x16x4, 6 accumulators step-major
template <int MODE>
__global__ __launch_bounds__(768) void k(float* out, int rounds, int nwaves) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0.001f * i;
  __syncthreads();
  if (wave >= nwaves) return;
  float w[19], b[3][19];
  for (int t = 0; t < 19; ++t) { w[t] = 0.01f * (t + lane); for (int j = 0; j < 3; ++j) b[j][t] = 0.02f * (t + j + lane); }
  f32x4 a4[6] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 a16[2] = {};
  const float* base = lds + lane * 3;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 19; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], b[j][t], a4[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j) b[j][t] = base[t * 16 + j * 48 + (r & 7)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 19; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], b[j][t], a4[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 5 < 19) {
#pragma unroll
          for (int j = 0; j < 3; ++j) b[j][t + 5] = base[(t + 5) * 16 + j * 48 + (r & 7)];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < 19; ++t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) a16[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[t], b[j][t], a16[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int t = 0; t < 19; ++t) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], b[j][t], a4[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 19; ++t) {
#pragma unroll
        for (int j = 0; j < 6; ++j) a4[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], b[j % 3][t], a4[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int j = 0; j < 6; ++j) s += a4[j][0] + a4[j][1] + a4[j][2] + a4[j][3];
  for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += a16[j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc[MODE] = t1 - t0;
}
template <int MODE>
void run(float* out, int nwaves, const char* what, int per_round, double flops_per) {
  const int rounds = 2000;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, out, 10, nwaves);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, out, rounds, nwaves);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  unsigned long long cyc[8]; (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
  const double n = (double)rounds * per_round;
  printf("%-58s waves/CU %2d: %6.1f ticks per instruction and wave  (%.3f ms; chip %.1f TFLOP/s)\n", what, nwaves, cyc[MODE] / n, ms,
         256.0 * nwaves * n * flops_per / (ms * 1e-3) / 1e12);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// bf16 16x16x32: VAR = 0 one A and one B register quad for every instruction; VAR = 1 twelve A quads and nine B quads in rotation (the
// operand pattern of the piece convolution in csrc/lift_fft.hip); 3 accumulators step-major either way
template <int VAR>
__global__ __launch_bounds__(768) void kb(float* out, int rounds, int nwaves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nwaves) return;
  bf16x8 A[12], B[9];
  for (int i = 0; i < 12; ++i) for (int e = 0; e < 8; ++e) A[i][e] = (__bf16)(0.01f * (lane + i + e));
  for (int i = 0; i < 9; ++i) for (int e = 0; e < 8; ++e) B[i][e] = (__bf16)(0.02f * (lane + 2 * i + e));
  f32x4 a4[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          a4[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(VAR ? A[3 * s + pr % 3] : A[0], VAR ? B[3 * j + pr / 2] : B[0], a4[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0;
  for (int j = 0; j < 3; ++j) sum += a4[j][0] + a4[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc[6 + VAR] = t1 - t0;
}
template <int VAR>
void runb(float* out, int nwaves, const char* what) {
  const int rounds = 2000;
  hipLaunchKernelGGL(kb<VAR>, dim3(256), dim3(768), 0, 0, out, 10, nwaves);
  hipLaunchKernelGGL(kb<VAR>, dim3(256), dim3(768), 0, 0, out, rounds, nwaves);
  (void)hipDeviceSynchronize();
  unsigned long long cyc[8]; (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
  printf("%-58s waves/CU %2d: %6.1f ticks per instruction and wave\n", what, nwaves, cyc[6 + VAR] / (rounds * 72.0));
}
int main() {
  float* out; if (hipMalloc(&out, 256 * 768 * 4) != hipSuccess) return 1;
  for (int nw : {4, 8, 12}) {
    run<0>(out, nw, "16x16x4, 3 accumulators step-major, register operands", 57, 2048.0);
    run<1>(out, nw, "16x16x4, 3 accumulators, + 1 ds_read_b32 per instruction", 57, 2048.0);
    run<2>(out, nw, "32x32x2, 2 accumulators step-major, register operands", 38, 4096.0);
    run<3>(out, nw, "16x16x4, ONE dependent chain", 57, 2048.0);
    run<4>(out, nw, "16x16x4, 6 accumulators step-major", 114, 2048.0);
  }
  for (int nw : {4, 8}) {
    runb<0>(out, nw, "bf16 16x16x32, 3 accumulators, ONE A / B register quad");
    runb<1>(out, nw, "bf16 16x16x32, 3 accumulators, 12 A x 9 B quads in rotation");
  }
  return 0;
}
