// Store-path rate of one CU for the spectrum layout of the fused forward transforms (csrc/lift_fft.hip): 6 waves per block, 256 persistent
// blocks, 64 items each, an item = 1154 frequency rows x 128 bytes ([Re x 16 | Im x 16] of one tile and channel group) at the real
// row pitch (1025 tiles x 2 KB).  Patterns, same bytes:
//   0  one dword per lane: 4 x 64-byte pieces per instruction, Re and Im pieces of a line by two instructions (rounds 2-5)
//   1  16 bytes per lane: 16 x 64-byte pieces per instruction, Re and Im by two instructions
//   2  16 bytes per lane: 8 x whole 128-byte lines per instruction
//   3  16 bytes per lane, tile-major toy layout: 1 KB contiguous per instruction (what a plain copy does)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o tools/micro/_bin/store_pattern && tools/micro/_bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kF = 1154, kM = 1025, kRow = 2048;   // bytes per (frequency, tile) row: 16 groups x 128
template <int PAT>
__global__ __launch_bounds__(384) void k(float* V, unsigned bytes, int items) {
  const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(V, 0, bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned grp = blockIdx.x % 16;
  const size_t rowb = (size_t)kM * kRow;
  for (int it = 0; it < items; ++it) {
    const unsigned m = blockIdx.x / 16 + 16 * it;
    const unsigned col = m * kRow + grp * 128;
    if (PAT == 0) {          // lane = (task q = lane / 16, channel c): 48 ky x (Re, Im); tasks = 4 per wave -> frequencies 23 ky + kc
      const int q = lane >> 4, c = lane & 15, kc = wave * 4 + q;
      for (int ky = 0; ky < 48; ++ky) {
        const unsigned off = (unsigned)((size_t)(23 * ky + kc) * rowb) + col + c * 4;
        __builtin_amdgcn_raw_buffer_store_b32(ky, vr, off, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b32(ky, vr, off, 64, 2);
      }
    } else if (PAT == 1) {   // lane = (ci = lane / 16, kxi, cq): 12 groups of 4 ky
      const int ci = lane >> 4, kxi = (lane >> 2) & 3, cq = lane & 3, kc = wave * 4 + kxi;
      for (int g = 0; g < 12; ++g) {
        const unsigned off = (unsigned)((size_t)(23 * (4 * g + ci) + kc) * rowb) + col + cq * 16;
        const u32x4 v = {1u, 2u, 3u, (unsigned)g};
        __builtin_amdgcn_raw_buffer_store_b128(v, vr, off, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, vr, off, 64, 2);
      }
    } else if (PAT == 2) {   // pairs of tasks: 8 lanes = one 128-byte line
      const int ci = lane >> 4, kxi = (lane >> 2) & 3, cq = lane & 3;
      const int half = kxi & 1, pairbase = wave * 4 + (kxi & 2);
      for (int g = 0; g < 12; ++g) {
        const u32x4 v = {1u, 2u, 3u, (unsigned)g};
        const unsigned off0 = (unsigned)((size_t)(23 * (4 * g + ci) + pairbase) * rowb) + col + half * 64 + cq * 16;
        __builtin_amdgcn_raw_buffer_store_b128(v, vr, off0, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, vr, off0 + (unsigned)rowb, 0, 2);
      }
    } else {                 // contiguous: 24 KB per wave and item
      const unsigned base = (unsigned)(((size_t)(blockIdx.x * (size_t)items + it) * 6 + wave) * 24576) + lane * 16;
      for (int g = 0; g < 24; ++g) {
        const u32x4 v = {1u, 2u, 3u, (unsigned)g};
        __builtin_amdgcn_raw_buffer_store_b128(v, vr, base, g * 1024, 2);
      }
    }
  }
}
int main() {
  const size_t bytes = (size_t)kF * kM * kRow;
  float* V; if (hipMalloc(&V, bytes) != hipSuccess) return 1;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int pat = 0; pat < 4; ++pat) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(a);
      if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(384), 0, 0, V, (unsigned)bytes, 64);
      if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(384), 0, 0, V, (unsigned)bytes, 64);
      if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(384), 0, 0, V, (unsigned)bytes, 64);
      if (pat == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(384), 0, 0, V, (unsigned)bytes, 64);
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    const double gb = 256.0 * 64 * 6 * 24576 / 1e9;
    printf("pattern %d: %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.1 GHz)\n", pat, best, gb / best, gb * 1e9 / (best * 1e-3) / 256 / 2.1e9);
  }
  return 0;
}
