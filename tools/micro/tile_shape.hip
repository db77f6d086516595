// Why does the tile stream of group_action_kernel reach 0.66 of the HBM peak on 224 x 224 planes and 0.57-0.59 on 1024 x 1024
// planes (config 5), where a flat copy reaches 0.67 on both?  Plain tile copy (no LDS, no arithmetic): one block of 256 threads per
// TW x TH tile of an image's three planes, grid = (8 * tiles_x, tiles_y, B / 8) like the product (blockIdx.x & 7 = image of the
// group = XCD), float4 per lane.  Variants: tile shape, the order the blocks of an XCD walk the tiles, planes per block.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tile_shape.hip -o tools/micro/_bin/tile_shape && tools/micro/_bin/tile_shape
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <functional>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ORDER 0: tile rows (x fastest)   1: tile columns (y fastest)   2: 4 x 4 super-tiles, rows inside
// ROT: read the tile from the transposed position (a right-angle rotation's window: rows of TH pixels at TW rows) -- reads only
template <int TW, int TH, int ORDER, bool ROT>
__global__ __launch_bounds__(256) void tile_copy(const float* __restrict__ s, float* __restrict__ d, int S, int tiles_x, int tiles_y) {
  const int xcd = blockIdx.x & 7;
  int lin = (int)blockIdx.y * tiles_x + (int)(blockIdx.x >> 3);
  int tx, ty;
  if (ORDER == 0) { ty = lin / tiles_x; tx = lin - ty * tiles_x; }
  else if (ORDER == 1) { tx = lin / tiles_y; ty = lin - tx * tiles_y; }
  else {
    const int sx = tiles_x >> 2;            // super-tiles per row
    const int st = lin >> 4, in = lin & 15;
    const int sy = st / sx;
    tx = (st - sy * sx) * 4 + (in & 3);
    ty = sy * 4 + (in >> 2);
  }
  const int n = (int)blockIdx.z * 8 + xcd;
  const size_t plane = (size_t)S * S;
  const float* sp = s + (size_t)n * 3 * plane;
  float* dp = d + (size_t)n * 3 * plane;
  constexpr int LPR = TW / 4;               // lanes per tile row
  constexpr int RPP = 256 / LPR;            // rows per pass
  const int q = threadIdx.x % LPR, r = threadIdx.x / LPR;
  f32x4 v[3][TH / RPP > 0 ? TH / RPP : 1];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < TH / RPP; ++k) {
      const int row = r + k * RPP;
      if (!ROT) v[c][k] = *reinterpret_cast<const f32x4*>(sp + c * plane + (size_t)(ty * TH + row) * S + tx * TW + 4 * q);
      else {
        // the transposed window: TW rows of TH pixels; lane (row, q) takes 4 pixels of window row (row * LPR + q) * 4 / TH ...
        const int idx = (row * LPR + q) * 4;                  // 0 .. TW * TH in steps of 4
        const int wr = idx / TH, wc = idx - wr * TH;          // window row (0 .. TW), column (0 .. TH)
        v[c][k] = *reinterpret_cast<const f32x4*>(sp + c * plane + (size_t)(tx * TW + wr) * S + ty * TH + wc);
      }
    }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < TH / RPP; ++k) {
      const int row = r + k * RPP;
      *reinterpret_cast<f32x4*>(dp + c * plane + (size_t)(ty * TH + row) * S + tx * TW + 4 * q) = v[c][k];
    }
}

static float time_us(const std::function<void()>& f, int reps = 30) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

template <int TW, int TH, int ORDER, bool ROT>
static void run(const char* what, float** rs, float** rd, int ring, int S, int B) {
  const int tiles_x = S / TW, tiles_y = S / TH;
  if (tiles_x * TW != S || tiles_y * TH != S || (ORDER == 2 && ((tiles_x & 3) || (tiles_y & 3))) || (ROT && S % TH)) return;
  int it = 0;
  const dim3 grid(8 * tiles_x, tiles_y, B / 8);
  const float us = time_us([&] { it = (it + 1) % ring; tile_copy<TW, TH, ORDER, ROT><<<grid, 256>>>(rs[it], rd[it], S, tiles_x, tiles_y); });
  const double bytes = 2.0 * B * 3 * S * S * 4;
  printf("  S=%4d B=%4d  %-44s %7.1f us  %6.0f GB/s  %.3f\n", S, B, what, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
}

int main() {
  for (int pass = 0; pass < 2; ++pass) {
    const int S = pass ? 1024 : 224, B = pass ? 32 : 672;
    const size_t bytes = (size_t)B * 3 * S * S * 4;
    const int ring = 3;
    float *rs[3], *rd[3];
    for (int i = 0; i < ring; ++i) { (void)hipMalloc(&rs[i], bytes); (void)hipMalloc(&rd[i], bytes); (void)hipMemset(rs[i], 1, bytes); }
    int it = 0;
    const float us = time_us([&] { it = (it + 1) % ring; (void)hipMemcpyAsync(rd[it], rs[it], bytes, hipMemcpyDeviceToDevice, 0); });
    printf("S=%d B=%d: hipMemcpy D2D %.1f us %.0f GB/s %.3f\n", S, B, us, 2.0 * bytes / us / 1e3, 2.0 * bytes / us / 1e3 / 8000);
    run<32, 32, 0, false>("32 x 32, rows of tiles", rs, rd, ring, S, B);
    run<32, 32, 1, false>("32 x 32, columns of tiles", rs, rd, ring, S, B);
    run<32, 32, 2, false>("32 x 32, 4 x 4 super-tiles", rs, rd, ring, S, B);
    run<64, 16, 0, false>("64 x 16, rows of tiles", rs, rd, ring, S, B);
    run<64, 16, 1, false>("64 x 16, columns of tiles", rs, rd, ring, S, B);
    run<128, 8, 0, false>("128 x 8, rows of tiles", rs, rd, ring, S, B);
    run<128, 8, 1, false>("128 x 8, columns of tiles", rs, rd, ring, S, B);
    run<256, 4, 0, false>("256 x 4, rows of tiles", rs, rd, ring, S, B);
    run<16, 64, 0, false>("16 x 64, rows of tiles", rs, rd, ring, S, B);
    run<32, 32, 0, true>("32 x 32 transposed window, rows of tiles", rs, rd, ring, S, B);
    run<32, 32, 1, true>("32 x 32 transposed window, columns of tiles", rs, rd, ring, S, B);
    run<64, 16, 0, true>("64 x 16 transposed window, rows of tiles", rs, rd, ring, S, B);
    run<64, 16, 1, true>("64 x 16 transposed window, columns", rs, rd, ring, S, B);
    for (int i = 0; i < ring; ++i) { (void)hipFree(rs[i]); (void)hipFree(rd[i]); }
  }
  return 0;
}
