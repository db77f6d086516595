// What does a 1:1 read/write stream reach on this part?  The ceiling group_action_kernel (154 MB in, 154 MB out at B = 256) is priced
// against in DESIGN 3.1.  Variants: one float4 per thread; persistent grid-stride with U float4 in flight per thread; non-temporal
// loads / stores; XCD-contiguous block order; read-only and write-only streams for reference.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/copy_bw.hip -o /tmp/copy_bw && /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_one(const f32x4* __restrict__ s, f32x4* __restrict__ d, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const f32x4 v = NTL ? __builtin_nontemporal_load(s + i) : s[i];
    if (NTS) __builtin_nontemporal_store(v, d + i); else d[i] = v;
  }
}

// persistent: block b walks chunks b, b + grid, ...; a chunk = 256 threads x U float4 (U loads in flight per thread)
template <int U, bool NTL, bool NTS, bool XCD>
__global__ __launch_bounds__(256) void copy_persist(const f32x4* __restrict__ s, f32x4* __restrict__ d, size_t n) {
  const size_t chunk = 256 * U;
  const size_t nchunk = n / chunk;
  size_t first = blockIdx.x, step = gridDim.x;
  size_t lo = 0, hi = nchunk;
  if (XCD) {  // XCD x (blockIdx & 7) owns a contiguous eighth of the buffer
    const size_t per = nchunk / 8;
    lo = (blockIdx.x & 7) * per; hi = lo + per;
    first = lo + (blockIdx.x >> 3); step = gridDim.x >> 3;
  }
  for (size_t c = first; c < hi; c += step) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(s + c * chunk + u * 256 + threadIdx.x) : s[c * chunk + u * 256 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NTS) __builtin_nontemporal_store(v[u], d + c * chunk + u * 256 + threadIdx.x); else d[c * chunk + u * 256 + threadIdx.x] = v[u];
    }
  }
}

__global__ __launch_bounds__(256) void read_only(const f32x4* __restrict__ s, float* __restrict__ out, size_t n) {
  f32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += s[i];
  if (a[0] + a[1] + a[2] + a[3] == 1.2345f) out[0] = 1.0f;
}
__global__ __launch_bounds__(256) void write_only(f32x4* __restrict__ d, size_t n) {
  const f32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = v;
}

template <typename F>
static float time_us(F launch, int iters = 30) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / iters;
}

int main() {
  for (size_t mb : {154ull, 616ull}) {
    const size_t bytes = mb * 1000 * 1000 / 4096 * 4096, n = bytes / 16;
    f32x4 *s, *d;
    hipMalloc(&s, bytes); hipMalloc(&d, bytes);
    hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
    auto rep = [&](const char* what, float us, double traffic) { printf("%4zu MB  %-46s %8.1f us  %6.2f TB/s\n", mb, what, us, traffic / us * 1e-6); };
    const double rw = 2.0 * bytes;
    const unsigned g1 = (unsigned)((n + 255) / 256);
    rep("one float4 / thread", time_us([&] { copy_one<false, false><<<g1, 256>>>(s, d, n); }), rw);
    rep("one float4 / thread, nt store", time_us([&] { copy_one<false, true><<<g1, 256>>>(s, d, n); }), rw);
    rep("one float4 / thread, nt load + store", time_us([&] { copy_one<true, true><<<g1, 256>>>(s, d, n); }), rw);
    for (unsigned bpc : {4u, 8u}) {
      char nm[96];
      snprintf(nm, sizeof nm, "persistent U=4, %u blocks/CU", bpc);
      rep(nm, time_us([&] { copy_persist<4, false, false, false><<<256 * bpc, 256>>>(s, d, n); }), rw);
      snprintf(nm, sizeof nm, "persistent U=8, %u blocks/CU", bpc);
      rep(nm, time_us([&] { copy_persist<8, false, false, false><<<256 * bpc, 256>>>(s, d, n); }), rw);
      snprintf(nm, sizeof nm, "persistent U=4, %u blocks/CU, nt store", bpc);
      rep(nm, time_us([&] { copy_persist<4, false, true, false><<<256 * bpc, 256>>>(s, d, n); }), rw);
      snprintf(nm, sizeof nm, "persistent U=4, %u blocks/CU, XCD-contiguous", bpc);
      rep(nm, time_us([&] { copy_persist<4, false, false, true><<<256 * bpc, 256>>>(s, d, n); }), rw);
      snprintf(nm, sizeof nm, "persistent U=8, %u blocks/CU, XCD-contiguous, nt", bpc);
      rep(nm, time_us([&] { copy_persist<8, true, true, true><<<256 * bpc, 256>>>(s, d, n); }), rw);
    }
    rep("hipMemcpyAsync d2d", time_us([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }), rw);
    rep("read only (sum), 8 blocks/CU", time_us([&] { read_only<<<2048, 256>>>(s, (float*)d, n); }), (double)bytes);
    rep("write only (fill), 8 blocks/CU", time_us([&] { write_only<<<2048, 256>>>(d, n); }), (double)bytes);
    hipFree(s); hipFree(d);
  }
  return 0;
}
