// Stand-alone timing of eqa_fft48k5_cgemm3m (csrc/cgemm3m.hip) for A/B experiments on the kernel:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I equiadapt_amd/csrc [-DEQA_CGEMM_<variant>] \
//         tools/micro/cgemm3m_bench.hip equiadapt_amd/csrc/cgemm3m.hip -o build_variants/cg_<variant>
//   build_variants/cg_<variant> [M=1024] [Cin=256] [Cout=256] [reps=20]
// Variants are compile-time hooks inside cgemm3m.hip (EQA_CGEMM_NOSTORE, EQA_CGEMM_SAMETILE, ...); the default build is the product kernel.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "eqa_hip.h"
#ifdef EQA_CG_INCLUDE_KERNEL
#include "../../equiadapt_amd/csrc/cgemm3m.hip"
#endif

// the two helpers cgemm3m.hip takes from fftconv.hip
extern "C" int eqa_fft48k5_frequencies(void) { return 1154; }
extern "C" int64_t eqa_fft48k5_tile_pitch(int64_t tiles) { return tiles <= 0 ? 0 : (tiles | 1); }

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f);     // uniform [-1, 1): random data, not zeros (the clock depends on it)
  }
}

int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 1024;
  const int Cin = argc > 2 ? atoi(argv[2]) : 256, Cout = argc > 3 ? atoi(argv[3]) : 256, reps = argc > 4 ? atoi(argv[4]) : 20;
  const int F = eqa_fft48k5_frequencies();
  const size_t pitch = (size_t)eqa_fft48k5_tile_pitch(M);
  const size_t nV = (size_t)F * pitch * 2 * Cin, nB = (size_t)F * Cin * Cout * 3, nM = (size_t)F * pitch * 2 * Cout;
  float *V, *B3, *Mo;
  hipMalloc(&V, nV * 4); hipMalloc(&B3, nB * 4); hipMalloc(&Mo, nM * 4);
  fill_kernel<<<4096, 256>>>(V, nV, 1u);
  fill_kernel<<<4096, 256>>>(B3, nB, 2u);
  hipDeviceSynchronize();
  for (int i = 0; i < 3; ++i) {
    const int st = eqa_fft48k5_cgemm3m(V, B3, Mo, M, Cin, Cout, nullptr);
    if (st) { printf("status %d\n", st); return 1; }
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0.f;
  for (int i = 0; i < reps; ++i) {
    hipEventRecord(e0, nullptr);
    eqa_fft48k5_cgemm3m(V, B3, Mo, M, Cin, Cout, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; sum += ms;
  }
  const double flop3 = 3.0 * 2.0 * F * (double)M * Cin * Cout;
#ifdef EQA_CGEMM_CLOCK
  {
    static unsigned long long h[1024 * 4];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cg_clock), sizeof(h));
    double tot = 0, mma = 0, epi = 0, tiles = 0, mx = 0;
    for (int w = 0; w < 1024; ++w) { tot += h[4 * w]; mma += h[4 * w + 1]; epi += h[4 * w + 2]; tiles += h[4 * w + 3]; mx = h[4 * w] > mx ? h[4 * w] : mx; }
    const double mf = 96.0 * (Cin / 16) * 64.0;      // MFMA issue cycles per tile
    printf("clock: per wave mean total %.0f (max %.0f) cycles, per tile: stage loop %.0f, epilogue %.0f, MFMA issue %.0f -> busy %.3f in the loop, %.3f overall; tiles/wave %.2f\n",
           tot / 1024, mx, mma / tiles, epi / tiles, mf, mf / (mma / tiles), mf * tiles / tot, tiles / 1024);
  }
#endif
  printf("M=%ld Cin=%d Cout=%d: mean %.3f ms, best %.3f ms, %.1f TFLOP/s (3M flops, mean), 4M-equivalent %.1f\n", M, Cin, Cout, sum / reps, best,
         flop3 / (sum / reps * 1e-3) / 1e12, flop3 * 4 / 3 / (sum / reps * 1e-3) / 1e12);
  return 0;
}
