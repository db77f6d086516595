// libeqa_hip.so, part 10 -- the channel contraction of the overlap-save FFT convolution (I2a) as a hand-written batched
// COMPLEX GEMM on the fp32 matrix cores, 3-multiplication form.  C ABI: include/eqa_hip.h.  Design notes: HISTORY.md section 3.4.
//
// Per stored frequency f:  Mo[f] (M x Cout, complex) = V[f] (M x Cin, complex) . B[f] (Cin x Cout, complex).  Through the GEMM
// library this ran as a real [M x 2Cin].[2Cin x 2Cout] product -- 4 real multiplies per complex one -- at 135-138 TFLOP/s, i.e.
// at what the fp32 MFMA delivers at the clock the chip holds: only fewer multiplies can make it faster.  Here
//     T1 = Ar.Br    T2 = Ai.Bi    T3 = (Ar + Ai).(Br + Bi)          Cr = T1 - T2    Ci = T3 - T1 - T2
// three real products instead of four (-25 % MFMA work) at unchanged HBM traffic: Br + Bi is precomputed once per weight
// version (eqa_fft48k5_filter_spectra3m), Ar + Ai is formed in registers from the loaded operands (16 adds per 96 MFMAs).
//
// v_mfma_f32_32x32x2_f32 issues once per 64 cycles per SIMD and eats one operand register of A and of B each time: the
// matrix pipe is 16x slower than for bf16, so the operand traffic per MFMA cycle is tiny -- a wave's 64 x 64 (complex) tile needs
// 20 KB of operands per 6144 cycles of MFMA issue.  Hence NO LDS and NO barriers: every wave is its own pipeline (one wave per
// SIMD, up to 512 registers), loads its MFMA operand fragments straight from global memory / L2 into registers -- 16 bytes per
// lane, one K-stage (16 complex k) ahead, double-buffered -- and keeps three accumulator sets (T1, T2, T3: 192 registers).
// The MFMA sums over k in any order, so a lane's 16 bytes are simply 4 consecutive channels of its row of V:
//   A fragment: lane l = 32 h + i holds V[row i][channel 16 s + 8 b + 4 h + t], t = 0..3 (b = 0, 1: two loads per stage)
//   B fragment: lane l = 32 h + j holds B[k = 16 s + 8 b + 4 h + t][col j] -- B3 is stored in exactly this fragment order, so
//               a wave's B loads are contiguous 1 KB runs
// and MFMA step (b, t) multiplies the k-pair {16 s + 8 b + t, 16 s + 8 b + 4 + t}.
// Epilogue.  Stored straight from the accumulator layout a tile is 64 eight-byte stores per lane, and the CU's store path takes
// ~90 cycles per such instruction: 23 k cycles per tile during which the wave's matrix pipe idles (measured with
// tools/micro/cgemm3m_bench.hip: 4.08 ms with, 3.46 ms without the stores).  So a finished tile is only COMBINED
// (Cr = T1 - T2, Ci = T3 - T1 - T2) and parked in the wave's private 32 KB of LDS (64 ds_write_b64, ~1.5 k cycles); it leaves
// for HBM during the NEXT tile's MFMA stream, two whole 512-byte rows per ds_read_b128 + global_store_dwordx4 pair, 32 / S pairs
// per K-stage -- LDS and store instructions issue in the shadow of the 64-cycle MFMAs.
// (Tried and dropped: tile-major spectra (tile, 8 channels, frequency, 16 floats) so that the fused FFT kernels could run two
// 8-channel blocks per CU on contiguous runs: correct, but forward 1.03 -> 1.48 ms, inverse 0.96 -> 1.11 ms, this kernel 3.54 ->
// 3.64 ms -- DESIGN 3.4.)
// Work: wave-tile = (frequency, 64 rows, 64 complex columns); the 4 waves of a block take the column tiles of one (f, row tile)
// (they share the rows of V through L1 / L2); the frequencies are dealt to the XCDs (f mod 8, block b runs on XCD b mod 8) so that a
// frequency's B panel (0.79 MB at 256 channels) is read from HBM once and then served by that XCD's L2 to its 16 row tiles.
#include <type_traits>

#include "eqa_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef EQA_CGEMM_CLOCK   // experiment (tools/micro/cgemm3m_bench.hip): shader-cycle stamps per wave [total, stage loops, epilogues, tiles]
__device__ unsigned long long g_cg_clock[1024 * 4];
#define CG_STAMP(x) const unsigned long long x = __builtin_readcyclecounter()
#else
#define CG_STAMP(x)
#endif

constexpr int kTileM = 64, kTileN = 64;   // wave tile: rows (tiles of the FFT convolution) x complex output channels
constexpr int kStageK = 16;               // complex k per stage = one [Re x 16 | Im x 16] group of V

struct OperandSet {                       // one K-stage of MFMA operands: 80 registers
  f32x4 ar[2][2], ai[2][2];               // [m][b]
  f32x4 b[3][2][2];                       // [part r/i/s][n][b]
};

// Operand loads go through buffer descriptors: a wave-uniform descriptor (rebuilt per tile by scalar code) + a scalar byte offset
// (frequency / K-stage, advanced by scalar adds) + a 32-bit lane offset that is constant within a tile.  No vector address
// arithmetic in the MFMA stream (a VALU instruction there costs ~6 MFMA cycles), and rows beyond the buffer read as zero instead
// of needing a clamp.
struct StageAddr {
  __amdgpu_buffer_rsrc_t a, b;   // rows of V the tile reads; B3[f]
  unsigned sa, sb;               // scalar byte offsets of the stage
};

__device__ __forceinline__ f32x4 buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// aoff0 / aoff1: byte offset of this lane's row (the two 32-row subtiles) + 16 h; boff = 16 * lane.  A row's stage is one
// [Re x 16 | Im x 16] piece: k-block b at +32 b, Im at +64.  In the order the MFMAs consume them: all of b = 0, then b = 1.
__device__ __forceinline__ void load_stage(OperandSet& o, const StageAddr& at, unsigned aoff0, unsigned aoff1, unsigned boff) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    o.ar[0][b] = buf_ld(at.a, aoff0 + 32 * b, at.sa);
    o.ai[0][b] = buf_ld(at.a, aoff0 + 32 * b + 64, at.sa);
    o.ar[1][b] = buf_ld(at.a, aoff1 + 32 * b, at.sa);
    o.ai[1][b] = buf_ld(at.a, aoff1 + 32 * b + 64, at.sa);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int p = 0; p < 3; ++p) o.b[p][n][b] = buf_ld(at.b, boff + ((n * 3 + p) * 2 + b) * 1024, at.sb);
  }
}

// FIRST: the tile's first K-stage -- the first product into each accumulator starts from zero, so no accumulator is ever cleared
// (clearing them was 2 x 192 v_accvgpr_write per tile in front of the matrix stream: the compiler emitted the loop twice)
template <bool FIRST = false>
__device__ __forceinline__ void mma_stage(const OperandSet& o, f32x16 (&acc)[3][2][2]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const f32x4 as0 = o.ar[0][b] + o.ai[0][b], as1 = o.ar[1][b] + o.ai[1][b];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float ar = o.ar[m][b][t], ai = o.ai[m][b][t], as = (m ? as1 : as0)[t];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const bool fresh = FIRST && b == 0 && t == 0;
          acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, o.b[0][n][b][t], fresh ? zero : acc[0][m][n], 0, 0, 0);
          acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, o.b[1][n][b][t], fresh ? zero : acc[1][m][n], 0, 0, 0);
          acc[2][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, o.b[2][n][b][t], fresh ? zero : acc[2][m][n], 0, 0, 0);
        }
      }
    }
  }
}

constexpr int kLdsRowFloats = 2 * kTileN;                 // one parked row: 64 complex = 128 floats = 512 bytes
constexpr int kLdsWaveFloats = kTileM * kLdsRowFloats;    // 32 KB per wave

// Where a parked tile goes: a buffer over the tile's rows that lie inside M (rows beyond it fall outside num_records and are
// dropped by the hardware: no branch), the lane's byte offset for row pair 0, the step to the next pair, a scalar offset.
struct ParkedDst {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff, pair_bytes;
  unsigned soff;
};

__device__ __forceinline__ void store_pair(const ParkedDst& d, int p, f32x4 v) {
#ifdef EQA_CGEMM_NOSTORE      // experiment: no global stores (the value stays live through the asm)
  asm volatile("" ::"v"(v));
#else
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), d.rsrc, d.voff + p * d.pair_bytes, d.soff, 0);
#endif
}

// row pairs [p0, p1) of the parked tile -> Mo: lanes 0..31 carry row 2p, lanes 32..63 row 2p + 1, 16 bytes each
__device__ __forceinline__ void flush_rows(const float* lds_lane, const ParkedDst& d, int p0, int p1) {
  for (int p = p0; p < p1; ++p) store_pair(d, p, *reinterpret_cast<const f32x4*>(lds_lane + p * (2 * kLdsRowFloats)));
}

// One K-stage: request the next stage's operands, send NPAIR row pairs of the parked tile on their way, 96 MFMAs -- as ONE
// scheduling region whose instruction order is then pinned: the LDS reads first, one global load behind every third MFMA (a run of
// 20 loads would stall the in-order wave for ~300 cycles with the matrix pipe drained), the stores further down.
template <int NPAIR, bool FIRST = false>
__device__ __forceinline__ void run_stage(OperandSet& nxt, const OperandSet& cur, f32x16 (&acc)[3][2][2], const StageAddr& at,
                                          unsigned aoff0, unsigned aoff1, unsigned boff, const float* lds_lane, const ParkedDst& dst,
                                          int p0) {
  f32x4 park[NPAIR > 0 ? NPAIR : 1];
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) park[k] = *reinterpret_cast<const f32x4*>(lds_lane + (p0 + k) * (2 * kLdsRowFloats));
  load_stage(nxt, at, aoff0, aoff1, boff);
  mma_stage<FIRST>(cur, acc);
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) store_pair(dst, p0 + k, park[k]);
#ifndef EQA_CGEMM_NOPIN
  __builtin_amdgcn_sched_group_barrier(0x100, NPAIR, 0);            // DS reads
#pragma unroll
  for (int k = 0; k < 20; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);              // 3 MFMAs
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // 1 global load
  }
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);              // 1 global store
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 36 - 2 * NPAIR, 0);
#endif
}

// V (F, pitch, 2 Cin) rows [Re x 16 | Im x 16] per 16 channels; B3 (F, S, Cout/32, 3, 2, 64, 4); Mo (F, pitch, 2 Cout) interleaved
// complex.  NPAIR: row pairs of the parked tile flushed per K-stage: 32 / S where S divides 32, 0: any S (unpinned flush loop)
template <int NPAIR>
__global__ __launch_bounds__(256, 1) void fft_cgemm3m_kernel(const float* __restrict__ V, const float* __restrict__ B3,
                                                             float* __restrict__ Mo, int M, int pitch, int Cin, int Cout, int F,
                                                             int n_rt, int n_ct, int waves_per_xcd) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & (kXcd - 1);
  const int q = (blockIdx.x >> 3) * 4 + wave;          // this wave's slot among the XCD's waves
  const int S = Cin / kStageK;                          // stages (even: Cin % 32 == 0)
  const int wpf = n_rt * n_ct;                          // wave-tiles per frequency
  const int nf_x = (F - xcd + kXcd - 1) / kXcd;         // frequencies of this XCD: xcd, xcd + 8, ...
  const int total = nf_x * wpf;                         // < 2^31: checked by the host
  if (q >= total) return;
  const int i = lane & 31, h = lane >> 5;
  __shared__ __attribute__((aligned(16))) float lds_all[4 * kLdsWaveFloats];
  float* lds_w = lds_all + wave * kLdsWaveFloats;       // this wave's parking space for one finished tile
  const float* lds_lane = lds_w + h * kLdsRowFloats + i * 4;
  const size_t rowf = (size_t)2 * Cin, mo_row = (size_t)2 * Cout;           // floats per row of V / Mo
  const unsigned b_stage_bytes = (unsigned)(Cout / 32) * 3 * 2 * 64 * 4 * 4;                    // bytes per (f, stage) of B3
  const unsigned a_stage = 32u * 4u;                                        // bytes from one K-stage of a row to the next
  const unsigned boff = lane * 16;

  // a wave-tile u (index in this XCD's sequence): operand descriptors + scalar offsets at stage 0, the lane's row offsets, its
  // coordinates.  All sizes are < 2^32 bytes (host-checked).
  auto locate = [&](int u, StageAddr& at, unsigned& aoff0, unsigned& aoff1, int& f, int& row0, int& ct) {
#ifdef EQA_CGEMM_SAMETILE     // experiment (tools/micro/cgemm3m_bench.hip): every wave-tile reads tile 0 -- operands always cached
    u = 0;
#endif
    const int fi = u / wpf, r = u - fi * wpf;
    f = xcd + kXcd * fi;
    const int rt = r / n_ct;
    ct = r - rt * n_ct;
    row0 = rt * kTileM;
    at.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V) + (size_t)f * pitch * rowf, 0, (unsigned)((size_t)pitch * rowf * 4), 0x00020000);
    at.sa = 0;
    aoff0 = (unsigned)((size_t)(row0 + i) * rowf + 4 * h) * 4u;        // rows >= pitch fall outside the descriptor and read as 0
    aoff1 = aoff0 + 32 * (unsigned)rowf * 4u;
    at.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B3) + (size_t)f * S * (b_stage_bytes / 4), 0, (unsigned)S * b_stage_bytes, 0x00020000);
    at.sb = (unsigned)(2 * ct) * (3 * 2 * 64 * 4 * 4);
  };
  auto at_stage = [&](const StageAddr& t, int s) { return StageAddr{t.a, t.b, t.sa + s * a_stage, t.sb + s * b_stage_bytes}; };
  // destination of the tile (f, row0, ct) once it is parked
  auto parked = [&](int f, int row0, int ct) {
    const int rows = min(kTileM, M - row0);
    ParkedDst d;
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(Mo + ((size_t)f * pitch + row0) * mo_row, 0, (unsigned)(rows * mo_row * 4), 0x00020000);
    d.voff = (h * (int)mo_row + ct * kLdsRowFloats + i * 4) * 4;
    d.pair_bytes = 2 * (int)mo_row * 4;
    d.soff = 0;
    return d;
  };

  StageAddr at;
  unsigned aoff0, aoff1;
  int f, row0, ct;
  locate(q, at, aoff0, aoff1, f, row0, ct);
  ParkedDst dst = parked(f, row0, ct);
  dst.rsrc = __builtin_amdgcn_make_buffer_rsrc(Mo, 0, 0, 0x00020000);       // nothing parked yet: an empty buffer drops the stores
  OperandSet s0, s1;
  CG_STAMP(c_begin);
#ifdef EQA_CGEMM_CLOCK
  unsigned long long c_mma = 0, c_epi = 0, c_tiles = 0;
#endif
  load_stage(s0, at, aoff0, aoff1, boff);
  for (int u = q; u < total; u += waves_per_xcd) {
    CG_STAMP(c0);
    f32x16 acc[3][2][2];      // never cleared: the first stage's products start from zero (mma_stage<true>)
    // the tile after this one (or this one again when it is the last: a harmless reload instead of a conditional load)
    const int un = u + waves_per_xcd < total ? u + waves_per_xcd : u;
    StageAddr nat;
    unsigned naoff0, naoff1;
    int nf, nrow0, nct;
    locate(un, nat, naoff0, naoff1, nf, nrow0, nct);
    // two K-stages; the scheduling barriers keep the next stage's loads INSIDE this stage's MFMA stream: left alone, the compiler
    // sinks them to their first use (the next stage) to save registers and every stage starts with an exposed HBM round trip
    auto stage_pair = [&](int s, auto first_tag) {
      constexpr bool kFirst = decltype(first_tag)::value;
      const bool more = s + 2 < S;
      if (NPAIR > 0) {
        __builtin_amdgcn_sched_barrier(0);
        run_stage<NPAIR, kFirst>(s1, s0, acc, at_stage(at, s + 1), aoff0, aoff1, boff, lds_lane, dst, s * NPAIR);
        __builtin_amdgcn_sched_barrier(0);
        run_stage<NPAIR>(s0, s1, acc, more ? at_stage(at, s + 2) : nat, more ? aoff0 : naoff0, more ? aoff1 : naoff1, boff, lds_lane, dst,
                         (s + 1) * NPAIR);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        load_stage(s1, at_stage(at, s + 1), aoff0, aoff1, boff);
        flush_rows(lds_lane, dst, (32 * s) / S, (32 * (s + 1)) / S);
        __builtin_amdgcn_sched_barrier(0);
        mma_stage<kFirst>(s0, acc);
        __builtin_amdgcn_sched_barrier(0);
        load_stage(s0, more ? at_stage(at, s + 2) : nat, more ? aoff0 : naoff0, more ? aoff1 : naoff1, boff);
        flush_rows(lds_lane, dst, (32 * (s + 1)) / S, (32 * (s + 2)) / S);
        __builtin_amdgcn_sched_barrier(0);
        mma_stage(s1, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    stage_pair(0, std::true_type{});
    for (int s = 2; s < S; s += 2) stage_pair(s, std::false_type{});
    CG_STAMP(c1);
    // epilogue: Cr = T1 - T2, Ci = T3 - T1 - T2 into the wave's LDS tile [row][complex column]; accumulator register e of lane
    // (h, j) is row (e & 3) + 8 (e >> 2) + 4 h, column j of its 32 x 32 block.  (All of the previous tile's rows have left the
    // LDS: the stage loop above flushed its 32 row pairs; LDS operations of one wave execute in order.)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const float t1 = acc[0][m][n][e], t2 = acc[1][m][n][e], t3 = acc[2][m][n][e];
          f32x2v c;
          c[0] = t1 - t2;
          c[1] = t3 - t1 - t2;
          *reinterpret_cast<f32x2v*>(lds_w + r * kLdsRowFloats + (32 * n + i) * 2) = c;
        }
      }
    dst = parked(f, row0, ct);
    at = nat; aoff0 = naoff0; aoff1 = naoff1; f = nf; row0 = nrow0; ct = nct;
#ifdef EQA_CGEMM_CLOCK
    CG_STAMP(c2);
    c_mma += c1 - c0; c_epi += c2 - c1; ++c_tiles;
#endif
  }
  flush_rows(lds_lane, dst, 0, 32);       // the wave's last tile
#ifdef EQA_CGEMM_CLOCK
  if (lane == 0) {
    unsigned long long* o = g_cg_clock + (size_t)(blockIdx.x * 4 + wave) * 4;
    o[0] = __builtin_readcyclecounter() - c_begin; o[1] = c_mma; o[2] = c_epi; o[3] = c_tiles;
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// The filter gradient's contraction (training):  D[f] (Cin x Cout, complex) = V[f]^T . conj(G[f]),  summed over the M tiles --
// the same 3-multiplication trick with the TILES as the contraction axis:
//     P1 = Vr^T Gr    P2 = Vi^T Gi    P3 = (Vr - Vi)^T (Gr + Gi)          Dr = P1 + P2    Di = ViGr - VrGi = P1 - P2 - P3
// (through the GEMM library: the real [2Cin x M].[M x 2Cout] product, 4 multiplies per complex one, 4.7 ms at B = 256).
// Wave tile = 64 input x 64 output channels of one frequency, three 2 x 2 x (32 x 32) accumulator sets (192 registers); operands
// straight from global memory / L2 into registers, one stage (16 tiles) ahead, no LDS, no barriers.  Rows beyond M fall outside
// the buffer descriptor and read as zero.  The K loop is M / 2 steps of 12 MFMAs (393 k cycles at M = 1024): the epilogue (128
// 8-byte stores per lane) is noise and goes straight from the accumulators.
// D comes out compact: D3 (F, Cin, 2, Cout) = Dr | Di per input channel, plain channel order (fft48_filter_grad_kernel<PACKED>).
// (Writing Dr / Di into two quadrants of the library form's (F, 2Cin, 2Cout) layout and leaving the rest unwritten made every
// line of D half-written: the reader then took 1.1 ms instead of 0.6.)
// (First version: v_mfma_f32_16x16x4_f32 with 48 four-register accumulators and 16-byte loads -- the register allocator kept a
// third of them in AGPRs as spill slots and shuttled them through VGPRs around every MFMA: 316 copies per 96 MFMAs, 6.4 ms.  With
// twelve 16-register accumulators, as in the kernel above, the MFMAs take them in AGPR form.)
constexpr int kWgSteps = 8;   // MFMA k-steps (2 tiles each) per stage: operands are requested one stage = 96 MFMAs = 6144 cycles ahead
                              // (one step ahead, 768 cycles, is less than an L2 round trip under load: 7 ms instead of 3.5)
struct WgOps {
  f32x2v ar[kWgSteps], ai[kWgSteps], br[kWgSteps], bi[kWgSteps];
};

__device__ __forceinline__ f32x2v buf_ld2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

__device__ __forceinline__ void wg_load(WgOps& o, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, unsigned aoff, unsigned boff,
                                        unsigned sa, unsigned sb, unsigned step_a, unsigned step_b) {
#pragma unroll
  for (int k = 0; k < kWgSteps; ++k) {
    o.ar[k] = buf_ld2(ra, aoff, sa + k * step_a);
    o.ai[k] = buf_ld2(ra, aoff + 64, sa + k * step_a);
    o.br[k] = buf_ld2(rb, boff, sb + k * step_b);
    o.bi[k] = buf_ld2(rb, boff + 64, sb + k * step_b);
  }
}

__device__ __forceinline__ void wg_mma(const WgOps& o, f32x16 (&acc)[3][2][2]) {
#pragma unroll
  for (int k = 0; k < kWgSteps; ++k) {
    const f32x2v as = o.ar[k] - o.ai[k], bs = o.br[k] + o.bi[k];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[0][t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ar[k][t], o.br[k][u], acc[0][t][u], 0, 0, 0);
        acc[1][t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.ai[k][t], o.bi[k][u], acc[1][t][u], 0, 0, 0);
        acc[2][t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[t], bs[u], acc[2][t][u], 0, 0, 0);
      }
  }
}

// one stage (16 tiles, 96 MFMAs): the next stage's 32 loads spread through them (one behind every third)
__device__ __forceinline__ void wg_stage(WgOps& nxt, const WgOps& cur, f32x16 (&acc)[3][2][2], __amdgpu_buffer_rsrc_t ra,
                                         __amdgpu_buffer_rsrc_t rb, unsigned aoff, unsigned boff, unsigned sa, unsigned sb, unsigned step_a,
                                         unsigned step_b) {
  wg_load(nxt, ra, rb, aoff, boff, sa, sb, step_a, step_b);
  wg_mma(cur, acc);
#pragma unroll
  for (int k = 0; k < 4 * kWgSteps; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  }
}

// v_mfma_f32_32x32x2_f32: A[i][k], B[k][j] one register each, lane l = 32 k + i -- k = 2 consecutive tiles.  A lane's 8-byte load is
// 2 consecutive channels of its tile row; component t feeds MFMA row block t, whose 32 rows are the channels
// {16 (i / 8) + 2 (i % 8) + t} of the wave tile's 64: a permutation the epilogue undoes when it stores.
__global__ __launch_bounds__(256, 1) void fft_wgrad3m_kernel(const float* __restrict__ V, const float* __restrict__ G, float* __restrict__ D,
                                                             int M, int pitch, int Cin, int Cout, int F, int n_it, int n_ot,
                                                             int waves_per_xcd) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & (kXcd - 1);
  const int q = (blockIdx.x >> 3) * 4 + wave;
  const int wpf = n_it * n_ot;                          // wave tiles per frequency; neighbouring waves share the input-channel tile
  const int nf_x = (F - xcd + kXcd - 1) / kXcd;
  const int total = nf_x * wpf;
  const int i = lane & 31, h = lane >> 5;
  const size_t rowa = (size_t)2 * Cin, rowb = (size_t)2 * Cout;
  const unsigned step_a = 2u * (unsigned)rowa * 4u, step_b = 2u * (unsigned)rowb * 4u;   // 2 tiles further
  const int steps = (M + 1) / 2;
  for (int u = q; u < total; u += waves_per_xcd) {
    const int fi = u / wpf, r = u - fi * wpf;
    const int f = xcd + kXcd * fi;
    const int it = r / n_ot, ot = r - it * n_ot;
#ifdef EQA_WG_SAMEF   // experiment: every wave tile reads frequency 0 -- operands always cached
    const int fl = 0;
#else
    const int fl = f;
#endif
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V) + (size_t)fl * pitch * rowa, 0,
                                                                        (unsigned)((size_t)M * rowa * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G) + (size_t)fl * pitch * rowb, 0,
                                                                        (unsigned)((size_t)M * rowb * 4), 0x00020000);
    // this lane's 2 channels of tile row h: group i / 8 of the wave tile's four [Re x 16 | Im x 16] groups, floats 2 (i % 8) ..
    const unsigned aoff = (unsigned)((size_t)h * rowa + (size_t)it * 128 + (i >> 3) * 32 + (i & 7) * 2) * 4u;
    const unsigned boff = (unsigned)((size_t)h * rowb + (size_t)ot * 128 + (i >> 3) * 32 + (i & 7) * 2) * 4u;
    f32x16 acc[3][2][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[p][t][v][e] = 0.f;
    // (A third operand set -- two stages of lead -- spills; a one-dword touch of the lines of the stage after next changed
    // nothing: the 9 % to the always-cached rate, 3.82 vs 3.49 ms at M = 1024, are not first-touch latency.)
    WgOps s0, s1;
    wg_load(s0, ra, rb, aoff, boff, 0, 0, step_a, step_b);
    for (int s = 0; s < steps; s += 2 * kWgSteps) {   // steps past the last read beyond the descriptors: zeros, harmless
      __builtin_amdgcn_sched_barrier(0);
      wg_stage(s1, s0, acc, ra, rb, aoff, boff, (s + kWgSteps) * step_a, (s + kWgSteps) * step_b, step_a, step_b);
      __builtin_amdgcn_sched_barrier(0);
      wg_stage(s0, s1, acc, ra, rb, aoff, boff, (s + 2 * kWgSteps) * step_a, (s + 2 * kWgSteps) * step_b, step_a, step_b);
      __builtin_amdgcn_sched_barrier(0);
    }
    // accumulator register e of lane (h, j = i) in block (t, v): block row (e & 3) + 8 (e >> 2) + 4 h = input-channel position ir,
    // channel 16 (ir / 8) + 2 (ir % 8) + t; output channel 16 (j / 8) + 2 (j % 8) + v -> 2 consecutive output channels per 8-byte store,
    // the 32 lanes of a half wave one 256-byte run
    float* Df = D + (size_t)f * (2 * (size_t)Cin) * Cout;           // D3[f][ci][Dr | Di][co], plain channel order
    const size_t col = (size_t)ot * 64 + (i >> 3) * 16 + (i & 7) * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ir = (e & 3) + 8 * (e >> 2) + 4 * h;
        const size_t ci = (size_t)it * 64 + (ir >> 3) * 16 + (ir & 7) * 2 + t;
        f32x2v dr, di;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const float p1 = acc[0][t][v][e], p2 = acc[1][t][v][e], p3 = acc[2][t][v][e];
          dr[v] = p1 + p2;
          di[v] = p1 - p2 - p3;
        }
        *reinterpret_cast<f32x2v*>(Df + (2 * ci) * Cout + col) = dr;          // 32 lanes: 256 contiguous bytes
        *reinterpret_cast<f32x2v*>(Df + (2 * ci + 1) * Cout + col) = di;
      }
  }
}

}  // namespace

extern "C" {

int eqa_fft48k5_cgemm3m_supported(int Cin, int Cout) { return Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % kTileN == 0; }

int64_t eqa_fft48k5_spectra3m_floats(int Cin, int Cout) {
  if (!eqa_fft48k5_cgemm3m_supported(Cin, Cout)) return 0;
  return (int64_t)eqa_fft48k5_frequencies() * Cin * Cout * 3;
}

int eqa_fft48k5_cgemm3m(const float* V, const float* B3, float* Mo, int64_t M, int Cin, int Cout, void* stream) {
  if (!V || !B3 || !Mo || M < 0 || Cin <= 0 || Cout <= 0) return EQA_ERR_INVALID_ARG;
  if (M == 0) return EQA_OK;
  const int F = eqa_fft48k5_frequencies();
  // every descriptor range and lane offset must fit 32 bits: one frequency of V / Mo, one frequency of B3
  const int64_t fm_bytes = ((M | 1) + 64) * 2 * (int64_t)std::max(Cin, Cout) * 4;
  if (!eqa_fft48k5_cgemm3m_supported(Cin, Cout) || M > 0x3fffff || fm_bytes > 0x7fffffffLL || (int64_t)Cin * Cout * 3 * 4 > 0x7fffffffLL ||
      (((uintptr_t)V | (uintptr_t)B3 | (uintptr_t)Mo) & 15))
    return EQA_ERR_UNSUPPORTED;
  const int n_rt = (int)((M + kTileM - 1) / kTileM), n_ct = Cout / kTileN;
  // persistent: one block of 4 waves per CU (the register budget admits one wave per SIMD); 32 blocks per XCD
  const int blocks = 256;
  const int S = Cin / kStageK;
#define EQA_CG_LAUNCH(NP)                                                                                                        \
  hipLaunchKernelGGL(fft_cgemm3m_kernel<NP>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, V, B3, Mo, (int)M,                 \
                     (int)eqa_fft48k5_tile_pitch(M), Cin, Cout, F, n_rt, n_ct, (blocks / kXcd) * 4)
  switch (32 % S == 0 ? 32 / S : 0) {     // (16 pairs per stage -- 32 channels -- would need 64 registers to park them: dynamic form)
    case 8: EQA_CG_LAUNCH(8); break;      // 64 channels (the reference tutorial's 16 x C4): 4 stages, 8 row pairs leave per stage
    case 4: EQA_CG_LAUNCH(4); break;
    case 2: EQA_CG_LAUNCH(2); break;
    case 1: EQA_CG_LAUNCH(1); break;
    default: EQA_CG_LAUNCH(0); break;
  }
#undef EQA_CG_LAUNCH
  return launch_status();
}

int eqa_fft48k5_wgrad3m_supported(int Cin, int Cout) { return Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0; }

int eqa_fft48k5_wgrad3m(const float* V, const float* G, float* D, int64_t M, int Cin, int Cout, void* stream) {
  if (!V || !G || !D || M <= 0 || Cin <= 0 || Cout <= 0) return EQA_ERR_INVALID_ARG;
  const int F = eqa_fft48k5_frequencies();
  const int64_t fm_bytes = ((M | 1) + 8) * 2 * (int64_t)std::max(Cin, Cout) * 4;
  if (!eqa_fft48k5_wgrad3m_supported(Cin, Cout) || M > 0x3fffff || fm_bytes > 0x7fffffffLL ||
      (((uintptr_t)V | (uintptr_t)G | (uintptr_t)D) & 15))
    return EQA_ERR_UNSUPPORTED;
  const int n_it = Cin / 64, n_ot = Cout / 64;
  if ((int64_t)F * n_it * n_ot > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;
  const int blocks = 256;
  hipLaunchKernelGGL(fft_wgrad3m_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, V, G, D, (int)M, (int)eqa_fft48k5_tile_pitch(M),
                     Cin, Cout, F, n_it, n_ot, (blocks / kXcd) * 4);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

}  // extern "C"
