// Copy-model of group_action_kernel's structure: what does a 1:1 tile stream reach when the source window of a 32 x 32 tile is
// staged in LDS by direct-to-LDS DMA and the tile is gathered out of LDS (4 neighbours) and stored as float4 --
//   base : one block per tile, three planes per stage, stage -> wait -> barrier -> gather -> store (the product kernel's shape)
//   pipe : persistent blocks, a ring of NBUF one-plane windows, every wave issues its share of the DMA for stage st + D
//          before it gathers stage st (software pipeline, in-order vmcnt), one barrier per stage
//   split: persistent blocks, NDMA extra waves do nothing but DMA (they never store), the four others gather + store
// B images of 3 x 224 x 224 floats, window edge WB (33 = right-angle elements, 47 = 45 degrees).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pc_tile.hip -o tools/micro/_bin/pc_tile && tools/micro/_bin/pc_tile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
#ifndef S_EDGE
#define S_EDGE 224          // -DS_EDGE=1024: config 5's planes (run with B = 32)
#endif
constexpr int S = S_EDGE, C = 3, T = 32, LS = 47, TR = S / T, TPI = TR * TR;  // image edge, planes, tile edge, LDS row stride, tiles per row / image
constexpr int kPlaneLds = LS * LS;

struct Win { int n, ty, tx, x_lo, y_lo; };
__device__ __forceinline__ Win decode(int w, int xcd, int WB) {
  Win o;
  const int z = w / TPI, t = w - z * TPI;
  o.ty = t / TR; o.tx = t - o.ty * TR; o.n = z * 8 + xcd;
  o.x_lo = min(max(o.tx * T - (WB - 33) / 2, 0), S - WB);
  o.y_lo = min(max(o.ty * T - (WB - 33) / 2, 0), S - WB);
  return o;
}

// rows [first, first + step, ...) x NR of a WB x WB window of `nch` planes -> LDS at lds (row stride nch * LS, plane stride LS);
// lane = window column.  NR instructions x nch per call whatever WB (rows past the window clamp onto its last row).
template <int NR, int NCH>
__device__ __forceinline__ void dma_rows(const float* plane0, int x_lo, int y_lo, int WB, float* lds, int first, int step, int lane) {
  const unsigned col = (unsigned)(x_lo + min(lane, WB - 1)) * 4u;
  const char* p0 = reinterpret_cast<const char*>(plane0);
  const char* p1 = reinterpret_cast<const char*>(plane0 + (NCH > 1 ? S * S : 0)) - LS * 4;
  const char* p2 = reinterpret_cast<const char*>(plane0 + (NCH > 2 ? 2 * S * S : 0)) - 2 * LS * 4;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
  if (lane < LS) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int y = min(first + k * step, WB - 1);
      const unsigned voff = (unsigned)((y_lo + y) * S) * 4u + col;
      const unsigned lrow = (unsigned)(uintptr_t)(lptr_t)(lds + y * (NCH * LS));
      if (NCH == 1)
        asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]" ::[v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0) : "memory");
      else
        asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]\n\tglobal_load_lds_dword %[v], %[p1] offset:%[o1]\n\t"
                     "global_load_lds_dword %[v], %[p2] offset:%[o2]" ::[v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2),
                     [o1] "i"(LS * 4), [o2] "i"(2 * LS * 4) : "memory");
    }
  }
  asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
}

template <int NCH, int NT>
__device__ __forceinline__ void gather_store(const float* lds, int dx, int dy, float* dst_plane0, int ty, int tx, int tid, int WB) {
  const int r = tid >> 3, q = tid & 7;
  const int row0 = min(r + dy, WB - 2);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = row0 * (NCH * LS) + c * LS + min(4 * q + k + dx, WB - 2);
      v[k] = lds[idx] * 0.4f + lds[idx + 1] * 0.3f + lds[idx + NCH * LS] * 0.2f + lds[idx + NCH * LS + 1] * 0.1f;
    }
    f32x4* o = reinterpret_cast<f32x4*>(dst_plane0 + (size_t)c * S * S + (size_t)(ty * T + r) * S + tx * T + 4 * q);
    if (NT == 1) __builtin_nontemporal_store(v, o);
    else if (NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(v) : "memory");
    else if (NT == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(v) : "memory");
    else if (NT == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(o), "v"(v) : "memory");
    else *o = v;
  }
}

// base with (MASK) only the lanes of the 45-degree diamond |x - 23| + |y - 23| <= 24 of the 47 x 47 box requesting data (the rotated tile's
// preimage + its neighbour ring), and / or (CHAIN) the product's dependent start-up loads in front of the DMA: group index of the image
// -> its row of the element table -> window position
template <bool MASK, bool CHAIN, int SLEEP = 0>
__global__ __launch_bounds__(256) void base_tile_v(const float* __restrict__ s, float* __restrict__ d, int B, int WB,
                                                   const int* __restrict__ gidx, const float* __restrict__ theta) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.z * 8 + (blockIdx.x & 7);
  if (n >= B) return;
  const int tx = blockIdx.x >> 3, ty = blockIdx.y;
  int x_lo = min(max(tx * T - (WB - 33) / 2, 0), S - WB), y_lo = min(max(ty * T - (WB - 33) / 2, 0), S - WB);
  if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);   // a prologue of SLEEP x 64 cycles in front of the DMA issue
  if (CHAIN) {
    const int e = gidx[n];
    const float* th = theta + e * 6;
    x_lo += (int)(th[0] + th[1] + th[2]);   // the table holds zeros: same window, but only known after two dependent round trips
    y_lo += (int)(th[3] + th[4] + th[5]);
  }
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const unsigned col = (unsigned)(x_lo + min(lane, WB - 1)) * 4u;
    const float* plane0 = s + (size_t)n * C * S * S;
    const char* p0 = reinterpret_cast<const char*>(plane0);
    const char* p1 = reinterpret_cast<const char*>(plane0 + S * S) - LS * 4;
    const char* p2 = reinterpret_cast<const char*>(plane0 + 2 * S * S) - 2 * LS * 4;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    const int adx = abs(lane - 23);
    if (lane < LS) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int y = min(wave + k * 4, WB - 1);
        const unsigned voff = (unsigned)((y_lo + y) * S) * 4u + col;
        const unsigned lrow = (unsigned)(uintptr_t)(lptr_t)(smem + y * (3 * LS));
        if (!MASK || adx + abs(y - 23) <= 24)
          asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]\n\tglobal_load_lds_dword %[v], %[p1] offset:%[o1]\n\t"
                       "global_load_lds_dword %[v], %[p2] offset:%[o2]" ::[v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2),
                       [o1] "i"(LS * 4), [o2] "i"(2 * LS * 4) : "memory");
      }
    }
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  gather_store<3, false>(smem, tx * T - x_lo, ty * T - y_lo, d + (size_t)n * C * S * S, ty, tx, threadIdx.x, WB);
}

template <int NT, int LDF>   // LDF: DMA flavour 1 nt, 2 sc1, 3 sc0 sc1
__global__ __launch_bounds__(256) void base_tile_ld(const float* __restrict__ s, float* __restrict__ d, int B, int WB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.z * 8 + (blockIdx.x & 7);
  if (n >= B) return;
  const int tx = blockIdx.x >> 3, ty = blockIdx.y;
  const int x_lo = min(max(tx * T - (WB - 33) / 2, 0), S - WB), y_lo = min(max(ty * T - (WB - 33) / 2, 0), S - WB);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const unsigned col = (unsigned)(x_lo + min(lane, WB - 1)) * 4u;
    const float* plane0 = s + (size_t)n * C * S * S;
    const char* p0 = reinterpret_cast<const char*>(plane0);
    const char* p1 = reinterpret_cast<const char*>(plane0 + S * S) - LS * 4;
    const char* p2 = reinterpret_cast<const char*>(plane0 + 2 * S * S) - 2 * LS * 4;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    if (lane < LS) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int y = min(wave + k * 4, WB - 1);
        const unsigned voff = (unsigned)((y_lo + y) * S) * 4u + col;
        const unsigned lrow = (unsigned)(uintptr_t)(lptr_t)(smem + y * (3 * LS));
#define DMA3(FL) asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]" FL "\n\tglobal_load_lds_dword %[v], %[p1] offset:%[o1]" FL "\n\t" \
                       "global_load_lds_dword %[v], %[p2] offset:%[o2]" FL ::[v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), \
                       [o1] "i"(LS * 4), [o2] "i"(2 * LS * 4) : "memory")
        if (LDF == 1) DMA3(" nt"); else if (LDF == 2) DMA3(" sc1"); else DMA3(" sc0 sc1");
      }
    }
    asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  gather_store<3, NT>(smem, tx * T - x_lo, ty * T - y_lo, d + (size_t)n * C * S * S, ty, tx, threadIdx.x, WB);
}

template <int NT>
__global__ __launch_bounds__(256) void base_tile(const float* __restrict__ s, float* __restrict__ d, int B, int WB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = blockIdx.z * 8 + (blockIdx.x & 7);
  if (n >= B) return;
  const int tx = blockIdx.x >> 3, ty = blockIdx.y;
  const int x_lo = min(max(tx * T - (WB - 33) / 2, 0), S - WB), y_lo = min(max(ty * T - (WB - 33) / 2, 0), S - WB);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  dma_rows<12, 3>(s + (size_t)n * C * S * S, x_lo, y_lo, WB, smem, wave, 4, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  gather_store<3, NT>(smem, tx * T - x_lo, ty * T - y_lo, d + (size_t)n * C * S * S, ty, tx, threadIdx.x, WB);
}

#define WAITCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// persistent, one plane per stage, all four waves issue DMA and gather; D = NBUF - 1 stages of prefetch
template <int NBUF, bool NT>
__global__ __launch_bounds__(256) void pipe_tile(const float* __restrict__ s, float* __restrict__ d, int B, int WB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int D = NBUF - 1;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int items = (B / 8) * TPI;
  const int mine = slot < items ? (items - slot + nslot - 1) / nslot : 0;
  const int total = mine * C;
  auto issue = [&](int st) {
    const int it = st / C, c = st - it * C;
    const Win w = decode(slot + it * nslot, xcd, WB);
    dma_rows<12, 1>(s + ((size_t)w.n * C + c) * S * S, w.x_lo, w.y_lo, WB, smem + (st % NBUF) * kPlaneLds, wave, 4, lane);
  };
  for (int st = 0; st < D && st < total; ++st) issue(st);
  if (total > D) WAITCNT(12 * (D - 1)); else WAITCNT(0);
  __syncthreads();
  for (int st = 0; st < total; ++st) {
    const bool more = st + D < total;
    if (more) issue(st + D);
    const int it = st / C, c = st - it * C;
    const Win w = decode(slot + it * nslot, xcd, WB);
    gather_store<1, NT>(smem + (st % NBUF) * kPlaneLds, w.tx * T - w.x_lo, w.ty * T - w.y_lo, d + ((size_t)w.n * C + c) * S * S, w.ty, w.tx,
                        threadIdx.x, WB);
    if (more) WAITCNT(12 * (D - 1) + 1); else WAITCNT(0);   // in-order counter: everything older than the last D-1 stages' DMA + this store
    __syncthreads();
  }
}

// persistent, three planes per stage (the product kernel's stage), ring of NBUF, all waves issue + gather
template <int NBUF, bool NT>
__global__ __launch_bounds__(256) void pipe3_tile(const float* __restrict__ s, float* __restrict__ d, int B, int WB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int D = NBUF - 1;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int items = (B / 8) * TPI;
  const int total = slot < items ? (items - slot + nslot - 1) / nslot : 0;
  auto issue = [&](int st) {
    const Win w = decode(slot + st * nslot, xcd, WB);
    dma_rows<12, 3>(s + (size_t)w.n * C * S * S, w.x_lo, w.y_lo, WB, smem + (st % NBUF) * 3 * kPlaneLds, wave, 4, lane);
  };
  for (int st = 0; st < D && st < total; ++st) issue(st);
  if (total > D && D == 2) WAITCNT(36); else WAITCNT(0);
  __syncthreads();
  for (int st = 0; st < total; ++st) {
    const bool more = st + D < total;
    if (more) issue(st + D);
    const Win w = decode(slot + st * nslot, xcd, WB);
    gather_store<3, NT>(smem + (st % NBUF) * 3 * kPlaneLds, w.tx * T - w.x_lo, w.ty * T - w.y_lo, d + (size_t)w.n * C * S * S, w.ty, w.tx, threadIdx.x, WB);
    if (more && D == 2) WAITCNT(36 + 3); else if (more && D == 1) WAITCNT(3); else WAITCNT(0);
    __syncthreads();
  }
}

// persistent, role split: waves 4 .. 4 + NDMA - 1 only DMA (one plane per stage, ring of NBUF), waves 0-3 gather + store
template <int NDMA, int NBUF, bool NT>
__global__ __launch_bounds__(256 + 64 * NDMA) void split_tile(const float* __restrict__ s, float* __restrict__ d, int B, int WB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int D = NBUF - 1, NR = (LS + NDMA - 1) / NDMA;
  static_assert(NR * (D - 1) <= 63, "wait count");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int items = (B / 8) * TPI;
  const int mine = slot < items ? (items - slot + nslot - 1) / nslot : 0;
  const int total = mine * C;
  if (wave >= 4) {
    auto issue = [&](int st) {
      const int it = st / C, c = st - it * C;
      const Win w = decode(slot + it * nslot, xcd, WB);
      dma_rows<NR, 1>(s + ((size_t)w.n * C + c) * S * S, w.x_lo, w.y_lo, WB, smem + (st % NBUF) * kPlaneLds, wave - 4, NDMA, lane);
    };
    for (int st = 0; st < D && st < total; ++st) issue(st);
    if (total > D) WAITCNT(NR * (D - 1)); else WAITCNT(0);
    __syncthreads();
    for (int st = 0; st < total; ++st) {
      const bool more = st + D < total;
      if (more) { issue(st + D); WAITCNT(NR * (D - 1)); } else WAITCNT(0);
      __syncthreads();
    }
  } else {
    __syncthreads();
    for (int st = 0; st < total; ++st) {
      const int it = st / C, c = st - it * C;
      const Win w = decode(slot + it * nslot, xcd, WB);
      gather_store<1, NT>(smem + (st % NBUF) * kPlaneLds, w.tx * T - w.x_lo, w.ty * T - w.y_lo, d + ((size_t)w.n * C + c) * S * S, w.ty, w.tx,
                          threadIdx.x, WB);
      __syncthreads();
    }
  }
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
  hipEventDestroy(a); hipEventDestroy(b);
  return ms * 1000.0f / iters;
}

static const float* g_s; static float* g_d; static int g_B; static bool g_quick;
static void report(const char* what, int WB, float us) {
  printf("B=%4d WB=%2d  %-58s %8.1f us  %6.2f TB/s\n", g_B, WB, what, us, 2.0 * g_B * C * S * S * 4 / us * 1e-6);
  fflush(stdout);
}
static bool check(const char* what) {
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { printf("  !! %s: %s\n", what, hipGetErrorString(e)); return false; }
  return true;
}
// verify one variant against the base kernel's output (same arithmetic per pixel)
static float* g_ref;
static void verify(const char* what) {
  static float* h0 = nullptr; static float* h1 = nullptr;
  const size_t n = (size_t)64 * C * S * S;   // the first 64 images
  if (!h0) { h0 = (float*)malloc(n * 4); h1 = (float*)malloc(n * 4); }
  (void)hipMemcpy(h0, g_ref, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(h1, g_d, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) bad += h0[i] != h1[i];
  if (bad) printf("  !! %s: %zu of %zu values differ from the base kernel\n", what, bad, n);
}

template <int NBUF, bool NT> static void run_pipe(int WB, int per_cu) {
  const size_t lds = (size_t)NBUF * kPlaneLds * 4;
  (void)hipFuncSetAttribute((const void*)pipe_tile<NBUF, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipMemset(g_d, 0, (size_t)64 * C * S * S * 4);
  const float us = time_us([&] { pipe_tile<NBUF, NT><<<256 * per_cu, 256, lds>>>(g_s, g_d, g_B, WB); });
  char w[128]; snprintf(w, sizeof w, "pipe  1 plane/stage ring %d, %d blocks/CU%s", NBUF, per_cu, NT ? ", nt" : "");
  if (check(w)) { report(w, WB, us); verify(w); }
}
template <int NBUF, bool NT> static void run_pipe3(int WB, int per_cu) {
  const size_t lds = (size_t)NBUF * 3 * kPlaneLds * 4;
  (void)hipFuncSetAttribute((const void*)pipe3_tile<NBUF, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipMemset(g_d, 0, (size_t)64 * C * S * S * 4);
  const float us = time_us([&] { pipe3_tile<NBUF, NT><<<256 * per_cu, 256, lds>>>(g_s, g_d, g_B, WB); });
  char w[128]; snprintf(w, sizeof w, "pipe3 3 planes/stage ring %d, %d blocks/CU%s", NBUF, per_cu, NT ? ", nt" : "");
  if (check(w)) { report(w, WB, us); verify(w); }
}
template <int NDMA, int NBUF, bool NT> static void run_split(int WB, int per_cu) {
  const size_t lds = (size_t)NBUF * kPlaneLds * 4;
  (void)hipFuncSetAttribute((const void*)split_tile<NDMA, NBUF, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipMemset(g_d, 0, (size_t)64 * C * S * S * 4);
  const float us = time_us([&] { split_tile<NDMA, NBUF, NT><<<256 * per_cu, 256 + 64 * NDMA, lds>>>(g_s, g_d, g_B, WB); });
  char w[128]; snprintf(w, sizeof w, "split %d DMA wave(s), ring %d, %d blocks/CU%s", NDMA, NBUF, per_cu, NT ? ", nt" : "");
  if (check(w)) { report(w, WB, us); verify(w); }
}

int main(int argc, char** argv) {
  g_B = argc > 1 ? atoi(argv[1]) : 1024;
  g_quick = argc > 2;   // base variants only
  const size_t bytes = (size_t)g_B * C * S * S * 4;
  float *s, *d, *ref;
  (void)hipMalloc(&s, bytes); (void)hipMalloc(&d, bytes); (void)hipMalloc(&ref, (size_t)64 * C * S * S * 4);
  {  // non-trivial data
    float* h = (float*)malloc(bytes);
    uint32_t x = 12345u;
    for (size_t i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) * (1.0f / 16777216.0f); }
    (void)hipMemcpy(s, h, bytes, hipMemcpyHostToDevice);
    free(h);
  }
  g_s = s; g_d = d; g_ref = ref;
  if (argc > 2 && argv[2][0] == 'r') {
    // ring of 4 buffer pairs of B images (beyond the 256 MB Infinity Cache when B = 256), one launch per pair, cycled: what a launch
    // of the product costs inside the real step, including whatever the kernel boundary costs (end-of-kernel write-back of the
    // dirty L2 lines, ramp, tail)
    float *rs[4], *rd[4];
    for (int i = 0; i < 4; ++i) { (void)hipMalloc(&rs[i], bytes); (void)hipMalloc(&rd[i], bytes); (void)hipMemcpy(rs[i], s, bytes, hipMemcpyDeviceToDevice); }
    const dim3 grid(8 * TR, TR, g_B / 8);
    const size_t lds3 = (size_t)3 * kPlaneLds * 4;
    for (int WB : {33, 35, 47}) {     // (35: the window starts one pixel left of the tile, i.e. in the previous 128-byte line)
      int it = 0;
#define RING(NTV, what) { const float us = time_us([&] { it = (it + 1) & 3; base_tile<NTV><<<grid, 256, lds3>>>(rs[it], rd[it], g_B, WB); }, 40); if (check(what)) report(what, WB, us); }
      RING(0, "ring: base, plain stores");
      RING(1, "ring: base, nt stores");
      RING(2, "ring: base, sc1 stores (write-through)");
      RING(3, "ring: base, sc0 sc1 stores");
      RING(4, "ring: base, sc1 nt stores");
      RING(0, "ring: base, plain stores (again)");
#define RINGL(NTV, LDF, what) { const float us = time_us([&] { it = (it + 1) & 3; base_tile_ld<NTV, LDF><<<grid, 256, lds3>>>(rs[it], rd[it], g_B, WB); }, 40); if (check(what)) report(what, WB, us); }
      RINGL(0, 1, "ring: base, nt DMA loads, plain stores");
      RINGL(1, 1, "ring: base, nt DMA loads, nt stores");
      RINGL(4, 1, "ring: base, nt DMA loads, sc1 nt stores");
      RINGL(0, 2, "ring: base, sc1 DMA loads, plain stores");
      RINGL(0, 3, "ring: base, sc0 sc1 DMA loads, plain stores");
      RING(0, "ring: base, plain stores (third)");
      {
        static int* gi = nullptr; static float* th = nullptr;
        if (!gi) { (void)hipMalloc(&gi, g_B * 4); (void)hipMalloc(&th, 64 * 4); (void)hipMemset(gi, 0, g_B * 4); (void)hipMemset(th, 0, 64 * 4); }
#define RINGS(SL, what) { const float us = time_us([&] { it = (it + 1) & 3; base_tile_v<false, true, SL><<<grid, 256, lds3>>>(rs[it], rd[it], g_B, WB, gi, th); }, 40); if (check(what)) report(what, WB, us); }
        RINGS(0, "ring: base + chain, no sleep");
        RINGS(4, "ring: base + chain + 256-cycle prologue");
        RINGS(8, "ring: base + chain + 512-cycle prologue");
        RINGS(16, "ring: base + chain + 1024-cycle prologue");
        RINGS(32, "ring: base + chain + 2048-cycle prologue");
        RINGS(64, "ring: base + chain + 4096-cycle prologue");
      }
    }
    return 0;
  }
  for (int WB : {33, 47}) {
    const dim3 grid(8 * TR, TR, g_B / 8);
    const size_t lds3 = (size_t)3 * kPlaneLds * 4;
    float us = time_us([&] { base_tile<0><<<grid, 256, lds3>>>(s, d, g_B, WB); });
    if (check("base")) report("base: block per tile, 3 planes, stage-barrier-gather-store", WB, us);
    (void)hipMemcpy(ref, d, (size_t)64 * C * S * S * 4, hipMemcpyDeviceToDevice);
    us = time_us([&] { base_tile<1><<<grid, 256, lds3>>>(s, d, g_B, WB); });
    if (check("base nt")) report("base, nt store", WB, us);
    {
      static int* gi = nullptr; static float* th = nullptr;
      if (!gi) { (void)hipMalloc(&gi, g_B * 4); (void)hipMalloc(&th, 64 * 4); (void)hipMemset(gi, 0, g_B * 4); (void)hipMemset(th, 0, 64 * 4); }
      us = time_us([&] { base_tile_v<false, true><<<grid, 256, lds3>>>(s, d, g_B, WB, gi, th); });
      if (check("chain")) report("base + dependent start-up loads (gidx -> theta -> window)", WB, us);
      us = time_us([&] { base_tile_v<true, false><<<grid, 256, lds3>>>(s, d, g_B, WB, gi, th); });
      if (check("mask")) report("base, DMA lanes masked to the 45-degree diamond", WB, us);
      us = time_us([&] { base_tile_v<true, true><<<grid, 256, lds3>>>(s, d, g_B, WB, gi, th); });
      if (check("mask+chain")) report("base, diamond mask + start-up loads", WB, us);
    }
    if (g_quick) continue;
    run_pipe<2, false>(WB, 4); run_pipe<3, false>(WB, 4); run_pipe<4, false>(WB, 4); run_pipe<4, false>(WB, 3);
    run_pipe<5, false>(WB, 3); run_pipe<6, false>(WB, 3); run_pipe<4, false>(WB, 2); run_pipe<6, false>(WB, 2);
    run_pipe<3, false>(WB, 6); run_pipe<3, false>(WB, 8); run_pipe<2, false>(WB, 8);
    run_pipe<4, true>(WB, 4); run_pipe<3, true>(WB, 6);
    run_pipe3<2, false>(WB, 3); run_pipe3<2, false>(WB, 2); run_pipe3<3, false>(WB, 2); run_pipe3<2, true>(WB, 3);
    run_split<1, 2, false>(WB, 4); run_split<2, 2, false>(WB, 4); run_split<2, 3, false>(WB, 4); run_split<2, 3, false>(WB, 3);
    run_split<2, 3, false>(WB, 5); run_split<2, 3, true>(WB, 4); run_split<1, 2, false>(WB, 6);
  }
  return 0;
}
