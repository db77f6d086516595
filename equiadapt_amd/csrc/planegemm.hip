// libeqa_hip.so, part 13 -- the per-plane channel contraction of the Winograd convolution (I2a where the 48 x 48 FFT tiles do not
// fit: escnn_networks.py:67-91 of the reference at small feature maps) as a hand-written batched REAL GEMM on the fp32 matrix cores:
//     M[t][p][:] (Cout) = V[t][p][:] (Cin) . U[p] (Cin x Cout)          p < P planes (36 / 64), t < T tiles
// the real-valued sibling of fft_cgemm3m_kernel (cgemm3m.hip), which it follows in everything but the complex arithmetic and the
// parked epilogue: one wave per SIMD, no LDS, no barriers; every wave streams its MFMA operand fragments straight from global
// memory / L2 into registers one K-stage (16 channels) ahead; v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: exact fmaf
// chains).  The MFMA sums over k in any order, so a lane's 16 bytes are 4 consecutive channels of its row of V:
//   V fragment (the MFMA's B operand, pixel = column j): lane l = 32 h + j holds V[row j][plane p][16 s + 8 b + 4 h + t], t = 0..3
//   U fragment (the A operand, output channel = row i):  lane l = 32 h + i holds U[p][16 s + 8 b + 4 h + t][col i]; the host
//       packs U in exactly this order (ops.pack_plane_gemm_weights), so a wave's U loads are contiguous 1 KB runs
// and step (b, t) multiplies the k-pair {16 s + 8 b + t, 16 s + 8 b + 4 + t}.  With the channels as accumulator ROWS a lane ends
// up with 4 consecutive output channels of one tile row per register group: the epilogue is 4 sixteen-byte stores per 32 x 32
// accumulator, straight from the accumulator registers (V and M are (tile, plane, channel): a plane's rows are P * C floats
// apart, which the fragment loads and these stores absorb -- no transposed copy of either).
// Work: wave-tile = (plane, 64 tiles, 32 NT output channels); the planes are dealt to the XCDs (p mod 8; block b runs on XCD
// b mod 8) so that a plane's U (256 KB at 256 channels) is read from HBM once and served by that XCD's L2 to its row tiles.
// Through the GEMM library (strided-batched, one launch) this product ran at 136 TFLOP/s at 256 channels; it is the fallback
// convolution, so the aim here is "no library kernel on the path", not the last percent.
#include "eqa_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
struct PgOps {              // one K-stage of operands
  f32x4 v[2][2];            // [row subtile m][b]
  f32x4 u[NT][2];           // [column subtile n][b]
};

__device__ __forceinline__ f32x4 pg_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int NT>
__device__ __forceinline__ void pg_load(PgOps<NT>& o, __amdgpu_buffer_rsrc_t rv, __amdgpu_buffer_rsrc_t ru, unsigned voff0, unsigned voff1,
                                        unsigned uoff, unsigned sv, unsigned su) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    o.v[0][b] = pg_ld(rv, voff0 + 32 * b, sv);
    o.v[1][b] = pg_ld(rv, voff1 + 32 * b, sv);
#pragma unroll
    for (int n = 0; n < NT; ++n) o.u[n][b] = pg_ld(ru, uoff + (n * 2 + b) * 1024, su);
  }
}

template <int NT>
__device__ __forceinline__ void pg_mma(const PgOps<NT>& o, f32x16 (&acc)[2][NT]) {
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.u[n][b][t], o.v[m][b][t], acc[m][n], 0, 0, 0);
}

// V (T, P, Cin); Upk (P, Cin/16, Cout/32, 2, 64, 4); Mo (T, P, Cout)
template <int NT>
__global__ __launch_bounds__(256, 1) void plane_gemm_kernel(const float* __restrict__ V, const float* __restrict__ Upk, float* __restrict__ Mo,
                                                            int T, int P, int Cin, int Cout, int n_rt, int n_ct, int waves_per_xcd) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & (kXcd - 1);
  const int q = (blockIdx.x >> 3) * 4 + wave;           // this wave's slot among the XCD's waves
  const int S = Cin / 16;
  const int wpp = n_rt * n_ct;                           // wave-tiles per plane
  const int np_x = (P - xcd + kXcd - 1) / kXcd;          // planes of this XCD: xcd, xcd + 8, ...
  const int total = np_x * wpp;
  if (q >= total) return;
  const int j = lane & 31, h = lane >> 5;
  const size_t vrow = (size_t)P * Cin, mrow = (size_t)P * Cout;   // floats between consecutive tiles of one plane
  const unsigned u_stage_bytes = (unsigned)(Cout / 32) * 2 * 1024;  // bytes of one (plane, K-stage) of Upk
  const unsigned uoff = lane * 16;

  struct Tile {
    __amdgpu_buffer_rsrc_t rv, ru;
    unsigned su;       // scalar byte offset of the tile's first column subtile inside a K-stage of Upk
    int p, row0, ct;
  };
  auto locate = [&](int u) {
    Tile t;
    const int pi = u / wpp, r = u - pi * wpp;
    t.p = xcd + kXcd * pi;
    const int rt = r / n_ct;
    t.ct = r - rt * n_ct;
    t.row0 = rt * 64;
    const int rows = min(64, T - t.row0);
    // rows beyond T fall outside the descriptor: they read as zero (and their stores are dropped, below)
    t.rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V) + (size_t)t.row0 * vrow + (size_t)t.p * Cin, 0,
                                             (unsigned)(((size_t)(rows - 1) * vrow + Cin) * 4), 0x00020000);
    t.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Upk) + (size_t)t.p * S * (u_stage_bytes / 4), 0, (unsigned)S * u_stage_bytes,
                                             0x00020000);
    t.su = (unsigned)(t.ct * NT) * 2048u;
    return t;
  };
  const unsigned voff0 = (unsigned)((size_t)j * vrow + 4 * h) * 4u;      // this lane's row of the first 32-row subtile (< 2^32: host-checked)
  const unsigned voff1 = voff0 + 32u * (unsigned)vrow * 4u;

  Tile cur = locate(q);
  PgOps<NT> s0, s1;
  pg_load<NT>(s0, cur.rv, cur.ru, voff0, voff1, uoff, 0, cur.su);
  for (int u = q; u < total; u += waves_per_xcd) {
    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
    // the tile after this one (or this one again when it is the last: a harmless reload instead of a conditional load)
    const Tile nxt = locate(u + waves_per_xcd < total ? u + waves_per_xcd : u);
    for (int s = 0; s < S; s += 2) {
      // the scheduling barriers keep the next stage's loads inside this stage's MFMA stream (left alone, the compiler sinks them
      // to their first use and every stage starts with an exposed L2 / HBM round trip)
      pg_load<NT>(s1, cur.rv, cur.ru, voff0, voff1, uoff, (unsigned)(s + 1) * 64u, cur.su + (unsigned)(s + 1) * u_stage_bytes);
      __builtin_amdgcn_sched_barrier(0);
      pg_mma<NT>(s0, acc);
      __builtin_amdgcn_sched_barrier(0);
      const bool more = s + 2 < S;
      pg_load<NT>(s0, more ? cur.rv : nxt.rv, more ? cur.ru : nxt.ru, voff0, voff1, uoff, more ? (unsigned)(s + 2) * 64u : 0u,
                  more ? cur.su + (unsigned)(s + 2) * u_stage_bytes : nxt.su);
      __builtin_amdgcn_sched_barrier(0);
      pg_mma<NT>(s1, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: accumulator register e of lane (h, j) is output channel (e & 3) + 8 (e >> 2) + 4 h of its 32-channel subtile, tile
    // row j of its 32-row subtile: register group g = e >> 2 is 16 contiguous bytes of M
    {
      const int rows = min(64, T - cur.row0);
      const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(Mo + (size_t)cur.row0 * mrow + (size_t)cur.p * Cout + (size_t)cur.ct * (32 * NT),
                                                                          0, (unsigned)(((size_t)(rows - 1) * mrow + Cout - (size_t)cur.ct * (32 * NT)) * 4),
                                                                          0x00020000);
      const unsigned moff = (unsigned)((size_t)j * mrow + 4 * h) * 4u;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc[m][n][4 * g + k];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rm, moff + (unsigned)(32 * n + 8 * g) * 4u,
                                                   (unsigned)m * 32u * (unsigned)mrow * 4u, 0);
          }
    }
    cur = nxt;
  }
}

}  // namespace

extern "C" {

int eqa_plane_gemm_supported(int Cin, int Cout) { return Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0; }

int eqa_plane_gemm(const float* V, const float* Upk, float* M, long long T, int P, int Cin, int Cout, void* stream) {
  if (T < 0 || P <= 0 || Cin <= 0 || Cout <= 0) return EQA_ERR_INVALID_ARG;
  if (T > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;
  if (!eqa_plane_gemm_supported(Cin, Cout)) return EQA_ERR_UNSUPPORTED;
  if (T == 0) return EQA_OK;
  if (!V || !Upk || !M) return EQA_ERR_INVALID_ARG;
  if ((((uintptr_t)V | (uintptr_t)Upk | (uintptr_t)M) & 15)) return EQA_ERR_UNSUPPORTED;
  // 32-bit lane offsets inside a 64-row wave tile; the U panel of one plane behind one descriptor
  if ((size_t)64 * P * std::max(Cin, Cout) * 4 >= (1ULL << 31) || (size_t)Cin * Cout * 4 >= (1ULL << 31)) return EQA_ERR_UNSUPPORTED;
  const int NT = Cout % 64 == 0 ? 2 : 1;
  const long long n_rt = (T + 63) / 64;
  const int n_ct = Cout / (32 * NT);
  const long long per_xcd = ((P + kXcd - 1) / kXcd) * n_rt * n_ct;
  if (per_xcd > 0x7fffffffLL || n_rt > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;
  // persistent waves: one per SIMD, 128 per XCD; fewer when there is less work
  const int waves_per_xcd = (int)std::min<long long>(128, (per_xcd + 3) / 4 * 4);
  const dim3 grid((unsigned)(kXcd * (waves_per_xcd / 4)));
  hipStream_t st = (hipStream_t)stream;
  if (NT == 2)
    hipLaunchKernelGGL(plane_gemm_kernel<2>, grid, dim3(256), 0, st, V, Upk, M, (int)T, P, Cin, Cout, (int)n_rt, n_ct, waves_per_xcd);
  else
    hipLaunchKernelGGL(plane_gemm_kernel<1>, grid, dim3(256), 0, st, V, Upk, M, (int)T, P, Cin, Cout, (int)n_rt, n_ct, waves_per_xcd);
  return launch_status();
}

}  // extern "C"
