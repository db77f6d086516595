// Would the channel contraction of the FFT convolution (csrc/cgemm3m.hip: fp32 MFMA, 0.85 of its 157 TFLOP/s peak, 57 % of the
// headline step) run faster on the bf16 matrix cores with fp32 operands split EXACTLY into three bf16 pieces (x = p1 + p2 + p3, each
// piece the next 8 significant bits, by truncation) and every product formed from the pieces?  Nine piece products reproduce the
// fp32 product exactly (fp32 accumulation, like v_mfma_f32_32x32x2_f32); six drop the three of relative size <= 2^-24.
// One K-stage of a wave's 64 x 64 complex tile in the 3-multiplication form: 48 fp32 A values per lane to split (VALU), B pieces
// ready-made, 12 accumulators x {6, 9} v_mfma_f32_32x32x16_bf16.  Prints shader cycles per stage against the fp32 form's 6144.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/bf16x3_rate.hip -o tools/micro/_bin/bf16x3_rate && tools/micro/_bin/bf16x3_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Pieces { u32x4 p[3]; };   // 8 values -> three packed bf16x8 operands

__device__ __forceinline__ Pieces split8(const float (&x)[8]) {
  Pieces o;
  unsigned r1[8], r2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = x[i];
    const float p1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffff0000u);
    const float b = a - p1;
    const float p2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xffff0000u);
    const float c = b - p2;
    r1[i] = __builtin_bit_cast(unsigned, b);
    r2[i] = __builtin_bit_cast(unsigned, c);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // high halves of two floats -> one dword (v_perm_b32)
    o.p[0][j] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[2 * j + 1]), __builtin_bit_cast(unsigned, x[2 * j]), 0x07060302u);
    o.p[1][j] = __builtin_amdgcn_perm(r1[2 * j + 1], r1[2 * j], 0x07060302u);
    o.p[2][j] = __builtin_amdgcn_perm(r2[2 * j + 1], r2[2 * j], 0x07060302u);
  }
  return o;
}

template <int TERMS, bool SPLIT>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float seed) {
  f32x16 acc[3][2][2];
  for (int t = 0; t < 3; ++t) for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[t][m][n][i] = 0.f;
  // B pieces: [part][n][piece]
  u32x4 bp[3][2][3];
  for (int t = 0; t < 3; ++t) for (int n = 0; n < 2; ++n) for (int p = 0; p < 3; ++p) for (int j = 0; j < 4; ++j)
    bp[t][n][p][j] = 0x3f803f80u + (threadIdx.x << 3) + t + n + p + j;
  float ar[2][8], ai[2][8];
  for (int m = 0; m < 2; ++m) for (int i = 0; i < 8; ++i) { ar[m][i] = seed + threadIdx.x * 0.37f + i + m; ai[m][i] = seed * 1.7f + threadIdx.x * 0.11f - i; }
  Pieces A[3][2];
  for (int t = 0; t < 3; ++t) for (int m = 0; m < 2; ++m) A[t][m] = split8(ar[m]);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (SPLIT) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float as[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // (stands for freshly loaded operands: values that change every stage)
          asm volatile("" : "+v"(ar[m][i]), "+v"(ai[m][i]));
          as[i] = ar[m][i] + ai[m][i];
        }
        A[0][m] = split8(ar[m]);
        A[1][m] = split8(ai[m]);
        A[2][m] = split8(as);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
          for (int pa = 0; pa < 3; ++pa)
#pragma unroll
            for (int pb = 0; pb < 3; ++pb) {
              if (TERMS == 6 && pa + pb > 2) continue;
              acc[t][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[t][m].p[pa]), __builtin_bit_cast(bf16x8, bp[t][n][pb]),
                                                                      acc[t][m][n], 0, 0, 0);
            }
        }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int t = 0; t < 3; ++t) for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) s += acc[t][m][n][i];
  if (s == 1.2345f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int TERMS, bool SPLIT>
void run(const char* what) {
  float* d; unsigned long long* c;
  (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 8);
  const int iters = 400;
  hipLaunchKernelGGL((k<TERMS, SPLIT>), dim3(256), dim3(256), 0, 0, d, c, 10, 1.f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<TERMS, SPLIT>), dim3(256), dim3(256), 0, 0, d, c, iters, 1.f);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h;
  (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const int nm = TERMS * 12;
  // fp32-equivalent flops of one stage: 3 real products x 64 x 64 x 16 x 2
  const double eq = 3.0 * 64 * 64 * 16 * 2;
  printf("%-44s %7.0f cycles per stage (%d MFMAs, %4.1f each; fp32 MFMA form: 6144)  %6.1f TFLOP/s fp32-equivalent on 1024 waves\n", what,
         (double)h / iters, nm, (double)h / iters / nm, eq * iters * 1024 / (ms * 1e-3) * 1e-12);
  (void)hipFree(d); (void)hipFree(c);
}

int main() {
  run<9, false>("9 products, no split (MFMA stream only)");
  run<9, true>("9 products + split of 48 values per stage");
  run<6, false>("6 products, no split");
  run<6, true>("6 products + split of 48 values per stage");
  return 0;
}
