// What v_permlane16_swap / v_permlane32_swap do to a register pair (gfx950), and the 4 x 4 (register, 16-lane row) transpose built
// from them (csrc/lift_fft.hip: transpose4), printed: register k of lane l starts as 1000 k + l.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/permlane_swap.hip -o tools/micro/_bin/permlane_swap && tools/micro/_bin/permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
// (inline asm, the wait states of the "VALU write -> v_permlane read" hazard inside the string: chained through the builtin's
// two-element result, hipcc 7.2 folded the second element into the first -- every output register came out as register 0)
__device__ __forceinline__ void sw16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void sw32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__global__ void k(float* out, const float* in) {
  const unsigned l = threadIdx.x;
  float r[4];
  for (int i = 0; i < 4; ++i) r[i] = in[64 * i + l];
  sw16(r[0], r[1]); sw16(r[2], r[3]); sw32(r[0], r[2]); sw32(r[1], r[3]);
  for (int i = 0; i < 4; ++i) out[64 * i + l] = r[i];
}
int main() {
  float *d, *din; float h[256], hin[256];
  for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) hin[64 * i + l] = 1000.0f * i + l;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&din, sizeof(h));
  (void)hipMemcpy(din, hin, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, din);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 4; ++i) {
    printf("register %d:", i);
    for (int r = 0; r < 4; ++r) {
      printf("  row%d: %g..%g", r, h[64 * i + 16 * r], h[64 * i + 16 * r + 15]);
      for (int p = 0; p < 16; ++p) bad += h[64 * i + 16 * r + p] != 1000.0f * r + 16 * i + p;   // register i of row r <- register r of row i
    }
    printf("\n");
  }
  printf("transpose4 %s\n", bad ? "WRONG" : "ok: register k of row i holds what register i held in row k");
  return bad != 0;
}
