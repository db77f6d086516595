// What does an instruction issued between two v_mfma_f32_32x32x2_f32 (64 cycles each) cost with ONE wave per SIMD?
// Two accumulator chains alternate; after every MFMA, F filler instructions of one kind.  Prints shader cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shadow.hip -o /tmp/mfma_shadow && /tmp/mfma_shadow
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { NONE, VALU, SALU, DSREAD, DSREAD128, DSWRITE128, VSTORE, NOP };

template <int KIND, int F>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, float a0, float b0) {
  __shared__ float lds[4096];
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  float v0 = a, v1 = b, v2 = a + b, v3 = a - b;
  f32x4 q = {a, b, a, b};
  unsigned s0 = iters, s1 = 3;
  lds[threadIdx.x] = a;
  __syncthreads();
  const unsigned ldsaddr = (threadIdx.x * 4) * 4;
  float* gp = out + 1024 + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 40; ++u) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (c == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
#pragma unroll
        for (int f = 0; f < F; ++f) {
          if (KIND == VALU) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f & 1 ? v0 : v1) : "v"(f & 1 ? v2 : v3));
          if (KIND == SALU) asm volatile("s_add_u32 %0, %0, %1" : "+s"(f & 1 ? s0 : s1) : "s"(3u) : "scc");
          if (KIND == DSREAD) asm volatile("ds_read_b32 %0, %1" : "=v"(f & 1 ? v0 : v1) : "v"(ldsaddr) : "memory");
          if (KIND == DSREAD128) asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(ldsaddr) : "memory");
          if (KIND == DSWRITE128) asm volatile("ds_write_b128 %0, %1" :: "v"(ldsaddr), "v"(q) : "memory");
          if (KIND == VSTORE) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gp), "v"(q) : "memory");
          if (KIND == NOP) asm volatile("s_nop 0");
        }
      }
    }
    if (KIND == DSREAD || KIND == DSREAD128 || KIND == DSWRITE128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == VSTORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v0 + v1 + q[0] + q[3] + (float)s0 + (float)s1;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 1.2345f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int F>
void run(const char* what) {
  float* d; unsigned long long* c;
  hipMalloc(&d, (1024 + 256 * 256 * 4) * 4 + 4096);
  hipMalloc(&c, 8);
  const int iters = 500;
  hipLaunchKernelGGL((k<KIND, F>), dim3(256), dim3(256), 0, 0, d, c, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<KIND, F>), dim3(256), dim3(256), 0, 0, d, c, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("%-22s x%d per MFMA: %6.1f cycles per MFMA (64 = free)\n", what, F, (double)h / (iters * 80.0));
  hipFree(d); hipFree(c);
}

int main() {
  run<NONE, 0>("nothing");
  run<VALU, 1>("v_max_f32"); run<VALU, 2>("v_max_f32"); run<VALU, 4>("v_max_f32"); run<VALU, 8>("v_max_f32");
  run<SALU, 1>("s_add_u32"); run<SALU, 2>("s_add_u32"); run<SALU, 4>("s_add_u32"); run<SALU, 8>("s_add_u32");
  run<NOP, 2>("s_nop 0"); run<NOP, 8>("s_nop 0");
  run<DSREAD, 1>("ds_read_b32"); run<DSREAD, 2>("ds_read_b32"); run<DSREAD, 4>("ds_read_b32");
  run<DSREAD128, 1>("ds_read_b128"); run<DSREAD128, 2>("ds_read_b128");
  run<DSWRITE128, 1>("ds_write_b128"); run<DSWRITE128, 2>("ds_write_b128");
  run<VSTORE, 1>("global_store_dwordx4");
  return 0;
}
