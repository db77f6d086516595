// What the memory system delivers when every block gathers one piece of S bytes from each of F planes (the read pattern of
// the fused inverse FFT: F = 1154 frequency planes of (tiles x channels) complex numbers, a block owns 16 channels of a
// tile = 128 bytes per plane).  Blocks of 256 threads, 8-byte loads, 16 pieces in flight per thread group, nothing else.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/piece_read.hip -o /tmp/piece_read && /tmp/piece_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

// piece = S bytes = S/8 lanes; a block of 256 lanes covers 256*8/S planes per trip
template <int S>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ base, float* out, int F, size_t plane_floats, int pieces_per_plane) {
  constexpr int LANES = S / 8, PER = 256 / LANES;
  const int lane = threadIdx.x % LANES, sub = threadIdx.x / LANES;
  const size_t piece = (size_t)blockIdx.x % pieces_per_plane;
  const float* p = base + piece * (S / 4) + lane * 2;
  f32x2 acc = {0.f, 0.f};
  for (int f = sub; f < F; f += PER * 4) {
    f32x2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ff = f + u * PER;
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p + (size_t)(ff < F ? ff : f) * plane_floats));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] == 1.2345f) out[0] = acc[0];
}

template <int S>
void run(const float* d, float* o, int F, size_t plane_bytes, const char* what) {
  const int pieces = (int)(plane_bytes / S);  // one block per piece of a plane
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gather<S>, dim3(pieces), dim3(256), 0, 0, d, o, F, plane_bytes / 4 + 512, pieces);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather<S>, dim3(pieces), dim3(256), 0, 0, d, o, F, plane_bytes / 4 + 512, pieces);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%-28s pieces of %4d B: %7.3f ms  %6.0f GB/s\n", what, S, ms, (double)F * plane_bytes / ms / 1e6);
}

int main() {
  const int F = 1154;
  const size_t plane = 2u << 20;  // 1024 tiles x 256 channels x 8 bytes
  float *d, *o;
  hipMalloc(&d, (size_t)F * (plane + 2048) + 4096);
  hipMalloc(&o, 16);
  hipMemset(d, 0, (size_t)F * (plane + 2048) + 4096);
  run<64>(d, o, F, plane, "one block per piece,");
  run<128>(d, o, F, plane, "one block per piece,");
  run<256>(d, o, F, plane, "one block per piece,");
  run<512>(d, o, F, plane, "one block per piece,");
  run<2048>(d, o, F, plane, "one block per piece,");
  return 0;
}
