// Does the SHAPE of a block's footprint matter for a 1:1 read/write stream?  Plane copy of (B*3) planes of 224 x 224 floats with
// blocks that own TW x TH pixel tiles (float4 per thread, one or more rows of 128-byte lines), in the grid order of
// group_action_kernel (x fastest: 8 * tiles_x with the XCD in the low bits, then tiles_y, then image groups) or plain.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tile_copy.hip -o /tmp/tile_copy && /tmp/tile_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int S = 224, C = 3;

// block = 256 threads = TW x TH pixels x (256 * 4 / (TW * TH)) ... one float4 per thread and channel
template <int TW, int TH, bool NT, bool LDS, bool PLAIN = false>
__global__ __launch_bounds__(256) void tile_copy(const float* __restrict__ s, float* __restrict__ d, int B) {
  static_assert(TW * TH == 1024, "256 threads x 4 pixels");
  __shared__ f32x4 stage[LDS ? 3 * 256 : 1];
  constexpr int tiles_x = (S + TW - 1) / TW;
  // PLAIN: consecutive blocks = consecutive tiles of one image (the 8 XCDs share every image)
  const int n = PLAIN ? blockIdx.z * 8 + blockIdx.x / tiles_x : blockIdx.z * 8 + (blockIdx.x & 7);
  if (n >= B) return;
  const int tx = PLAIN ? blockIdx.x % tiles_x : blockIdx.x >> 3, ty = blockIdx.y;
  const int qpr = TW / 4;  // float4 per tile row
  const int r = threadIdx.x / qpr, q = threadIdx.x % qpr;
  const int y = ty * TH + r, x = tx * TW + 4 * q;
  if (y >= S || x >= S) return;
  const size_t base = (size_t)n * C * S * S + (size_t)y * S + x;
  f32x4 v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = *reinterpret_cast<const f32x4*>(s + base + (size_t)c * S * S);
  if (LDS) {  // through LDS behind a barrier, like a staged kernel
#pragma unroll
    for (int c = 0; c < C; ++c) stage[c * 256 + threadIdx.x] = v[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = stage[c * 256 + (threadIdx.x ^ 1)];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    f32x4* o = reinterpret_cast<f32x4*>(d + base + (size_t)c * S * S);
    if (NT) __builtin_nontemporal_store(v[c], o); else *o = v[c];
  }
}

__global__ __launch_bounds__(256) void flat_copy(const f32x4* __restrict__ s, f32x4* __restrict__ d, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}

template <typename F>
static float time_us(F launch, int iters = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}

// one plane per block: 1024 threads-worth of pixels = TW x TH of ONE channel; planes = B * 3
template <int TW, int TH>
__global__ __launch_bounds__(256) void plane_tile_copy(const float* __restrict__ s, float* __restrict__ d, int planes) {
  const int n = blockIdx.z * 8 + (blockIdx.x & 7);
  if (n >= planes) return;
  const int tx = blockIdx.x >> 3, ty = blockIdx.y;
  const int qpr = TW / 4;
  const int r = threadIdx.x / qpr, q = threadIdx.x % qpr;
  const int y = ty * TH + r, x = tx * TW + 4 * q;
  if (y >= S || x >= S) return;
  const size_t base = (size_t)n * S * S + (size_t)y * S + x;
  *reinterpret_cast<f32x4*>(d + base) = *reinterpret_cast<const f32x4*>(s + base);
}
// flat copy, three float4 per thread a plane apart (what a 3-channel block does), blocks in linear order
__global__ __launch_bounds__(256) void flat3_copy(const float* __restrict__ s, float* __restrict__ d, int B) {
  const size_t plane4 = (size_t)S * S / 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index inside one image plane, images along blockIdx.y
  if (i >= plane4) return;
  const f32x4* sp = reinterpret_cast<const f32x4*>(s) + (size_t)blockIdx.y * 3 * plane4 + i;
  f32x4* dp = reinterpret_cast<f32x4*>(d) + (size_t)blockIdx.y * 3 * plane4 + i;
  const f32x4 a = sp[0], b = sp[plane4], c = sp[2 * plane4];
  dp[0] = a; dp[plane4] = b; dp[2 * plane4] = c;
}

template <int TW, int TH, bool NT, bool LDS, bool PLAIN = false>
static void run(const float* s, float* d, int B, const char* what) {
  const dim3 grid(8 * ((S + TW - 1) / TW), (S + TH - 1) / TH, (B + 7) / 8);
  const float us = time_us([&] { tile_copy<TW, TH, NT, LDS, PLAIN><<<grid, 256>>>(s, d, B); });
  printf("B=%4d  %-44s %8.1f us  %6.2f TB/s\n", B, what, us, 2.0 * B * C * S * S * 4 / us * 1e-6);
}

int main() {
  for (int B : {1024}) {
    const size_t bytes = (size_t)B * C * S * S * 4;
    float *s, *d;
    (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes);
    (void)hipMemset(s, 1, bytes); (void)hipMemset(d, 0, bytes);
    const size_t n4 = bytes / 16;
    const float us = time_us([&] { flat_copy<<<(unsigned)((n4 + 255) / 256), 256>>>((const f32x4*)s, (f32x4*)d, n4); });
    printf("B=%4d  %-44s %8.1f us  %6.2f TB/s\n", B, "flat: one float4 per thread", us, 2.0 * bytes / us * 1e-6);
    run<32, 32, false, false>(s, d, B, "tiles 32 x 32");
    run<32, 32, false, true>(s, d, B, "tiles 32 x 32, through LDS + barrier");
    run<32, 32, false, false, true>(s, d, B, "tiles 32 x 32, plain block order");
    run<32, 32, false, true, true>(s, d, B, "tiles 32 x 32, plain order, LDS + barrier");
    {
      const dim3 grid(8 * 7, 7, (3 * B + 7) / 8);
      const float us = time_us([&] { plane_tile_copy<32, 32><<<grid, 256>>>(s, d, 3 * B); });
      printf("B=%4d  %-44s %8.1f us  %6.2f TB/s\n", B, "tiles 32 x 32 of ONE plane per block", us, 2.0 * bytes / us * 1e-6);
      const dim3 g3((S * S / 4 + 255) / 256, B);
      const float us3 = time_us([&] { flat3_copy<<<g3, 256>>>(s, d, B); });
      printf("B=%4d  %-44s %8.1f us  %6.2f TB/s\n", B, "flat, 3 planes per thread", us3, 2.0 * bytes / us3 * 1e-6);
    }
    run<32, 32, true, false>(s, d, B, "tiles 32 x 32, nt store");
    run<64, 16, false, false>(s, d, B, "tiles 64 x 16");
    run<128, 8, false, false>(s, d, B, "tiles 128 x 8");
    run<256, 4, false, false>(s, d, B, "tiles 256 x 4 (whole rows)");
    run<256, 4, true, false>(s, d, B, "tiles 256 x 4 (whole rows), nt store");
    run<256, 4, false, true>(s, d, B, "tiles 256 x 4 (whole rows), LDS + barrier");
    (void)hipFree(s); (void)hipFree(d);
  }
  return 0;
}
