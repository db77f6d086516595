// libeqa_hip.so, part 11 -- the channel contraction of the overlap-save FFT convolution (cgemm3m.hip) on the bf16 matrix cores,
// EXACT in the sense the fp32 matrix instruction is: every fp32 operand is split without error into three bf16 pieces
//     x = p1 + p2 + p3,     p1 = the top 16 bits of x, p2 = the top 16 bits of x - p1, p3 = x - p1 - p2   (8 significant bits each:
//     3 x 8 = the 24 of an fp32 significand; truncation keeps the remainders' signs equal, so no bit is spent twice)
// and the product of two operands is the sum of the nine products of their pieces, each of which the matrix core forms exactly
// (8 x 8 significant bits) and accumulates in fp32 -- the same "exact products, fp32 accumulation" contract as
// v_mfma_f32_32x32x2_f32, in a different summation order.  v_mfma_f32_32x32x16_bf16 retires 16 k per 32 cycles where the fp32
// form retires 2 per 64: nine piece products cost 288 cycles per 16 k against 512.  C ABI: include/eqa_hip.h; HISTORY.md 3.8.
//
// TERMS = 9: all piece products (the default when this path is selected).  TERMS = 6: the three products of relative size
// <= 2^-24 (p2.p3, p3.p2, p3.p3) are left out -- an error of at most 2^-23 |a||b| per product, the size of the rounding a single
// fp32 multiply-add commits; opt-in (the caller passes terms = 6), reported beside the exact form, never silently.
//
// Same work decomposition, operand flow and epilogue as fft_cgemm3m_kernel: a wave-tile is (frequency, 64 rows, 64 complex
// columns), three accumulator sets (T1 = Ar.Br, T2 = Ai.Bi, T3 = (Ar + Ai).(Br + Bi)), one wave per SIMD, no LDS for operands,
// operands one K-stage (16 complex k) ahead, the finished tile parked in LDS and flushed under the next tile's MFMA stream.
// A (the spectra of the activations, fp32 in HBM) is split in registers: 48 values per lane and stage, ~6 VALU instructions each,
// in the shadow of the matrix instructions (tools/micro/bf16x3_rate.hip: 3925 cycles per stage with the split, 3465 without,
// 6144 in the fp32 form).  B (the filter spectra) is split once per weight version (eqa_fft48k5_spectra3m_split) into the
// fragment order of the bf16 instruction: (F, S, Cout / 32, 3 parts, 3 pieces, 64 lanes, 8 bf16).
// The instruction sums over k in any order, so a lane keeps the 16 + 16 bytes it loads today: slot e = 4 b + t of lane (i, h) is
// channel 16 s + 8 b + 4 h + t on both sides.
#include "eqa_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kTileM = 64, kTileN = 64;
constexpr int kStageK = 16;
constexpr unsigned kHi = 0xffff0000u;

// 8 fp32 values (two 16-byte loads) -> three packed operands of the bf16 instruction
struct Pieces {
  u32x4 p[3];
};
__device__ __forceinline__ Pieces split8(const f32x4 lo, const f32x4 hi) {
  Pieces o;
#ifdef EQA_CGEMM_NOSPLIT      // experiment: no split arithmetic (wrong values): what the kernel costs without its vector work
  o.p[0] = __builtin_bit_cast(u32x4, lo); o.p[1] = __builtin_bit_cast(u32x4, hi); o.p[2] = __builtin_bit_cast(u32x4, lo);
  return o;
#endif
  unsigned x0[8], x1[8], x2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = e < 4 ? lo[e & 3] : hi[e & 3];
    const float p1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & kHi);
    const float r1 = a - p1;                                       // exact: the low 16 bits of the significand
    const float p2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & kHi);
    const float r2 = r1 - p2;                                      // exact: at most 8 significant bits -> a bf16 as it stands
    x0[e] = __builtin_bit_cast(unsigned, a);
    x1[e] = __builtin_bit_cast(unsigned, r1);
    x2[e] = __builtin_bit_cast(unsigned, r2);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {                                    // the high halves of two floats -> one dword (v_perm_b32)
    o.p[0][j] = __builtin_amdgcn_perm(x0[2 * j + 1], x0[2 * j], 0x07060302u);
    o.p[1][j] = __builtin_amdgcn_perm(x1[2 * j + 1], x1[2 * j], 0x07060302u);
    o.p[2][j] = __builtin_amdgcn_perm(x2[2 * j + 1], x2[2 * j], 0x07060302u);
  }
  return o;
}

// Registers: 192 accumulators (AGPRs) leave 256 + 64 for everything else, and the register allocator shuttles operands between
// the two files once the architectural half overflows (first version: 576 v_accvgpr copies per 216 matrix instructions, no faster
// than the fp32 kernel).  So B is held by HALF stages -- the 36 registers of one 32-column half in use, the other half in flight --
// and a stage runs column half by column half: [n = 0: m = 0, m = 1] [n = 1: m = 0, m = 1] with the pieces of both row halves of A
// (72 registers) split once per stage.
struct ARaw {                    // the rows of one 32-row half of a K-stage of A as loaded: fp32, [b]
  f32x4 ar[2], ai[2];
};
struct BHalf {                   // one 32-column half of a K-stage of B: [part r/i/s][piece]
  u32x4 b[3][3];
};
struct APieces {
  Pieces a[2][3];                // [m][part]
};

struct StageAddr {
  __amdgpu_buffer_rsrc_t a, b;
  unsigned sa, sb;
};

__device__ __forceinline__ u32x4 buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

constexpr unsigned kBFrag = 64 * 16;    // bytes of one (part, piece) fragment of a 32-column tile

__device__ __forceinline__ void load_a(ARaw& o, const StageAddr& at, unsigned aoff) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    o.ar[b] = __builtin_bit_cast(f32x4, buf_ld(at.a, aoff + 32 * b, at.sa));
    o.ai[b] = __builtin_bit_cast(f32x4, buf_ld(at.a, aoff + 32 * b + 64, at.sa));
  }
}
__device__ __forceinline__ void load_b(BHalf& o, const StageAddr& at, unsigned boff, int n) {
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) o.b[p][q] = buf_ld(at.b, boff + ((n * 3 + p) * 3 + q) * kBFrag, at.sb);
}

__device__ __forceinline__ void split_a(APieces& P, const ARaw& r, int m) {
  P.a[m][0] = split8(r.ar[0], r.ar[1]);
  P.a[m][1] = split8(r.ai[0], r.ai[1]);
  P.a[m][2] = split8(r.ar[0] + r.ai[0], r.ar[1] + r.ai[1]);
}

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the matrix instructions of one (column half n, row half m): consecutive ones go to different accumulators, large products first
template <int TERMS>
__device__ __forceinline__ void mma_quarter(const APieces& P, const BHalf& B, f32x16 (&acc)[3][2][2], int m, int n) {
#pragma unroll
  for (int w = 0; w < 5; ++w)         // w = pa + pb
#pragma unroll
    for (int pa = 0; pa < 3; ++pa) {
      const int pb = w - pa;
      if (pb < 0 || pb > 2 || (TERMS == 6 && w > 2)) continue;
#pragma unroll
      for (int p = 0; p < 3; ++p) acc[p][m][n] = mma(P.a[m][p].p[pa], B.b[p][pb], acc[p][m][n]);
    }
}

constexpr int kLdsRowFloats = 2 * kTileN;
constexpr int kLdsWaveFloats = kTileM * kLdsRowFloats;

struct ParkedDst {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff, pair_bytes;
  unsigned soff;
};

__device__ __forceinline__ void store_pair(const ParkedDst& d, int p, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), d.rsrc, d.voff + p * d.pair_bytes, d.soff, 0);
}
__device__ __forceinline__ void flush_rows(const float* lds_lane, const ParkedDst& d, int p0, int p1) {
  for (int p = p0; p < p1; ++p) store_pair(d, p, *reinterpret_cast<const f32x4*>(lds_lane + p * (2 * kLdsRowFloats)));
}

// One K-stage = four quarters (column half n, row half m), 3 x TERMS matrix instructions each, every quarter one scheduling region
// in which the instruction order is pinned: the wave issues in order, so whatever is to run in the shadow of the matrix pipe has
// to sit BETWEEN two matrix instructions.
//   quarter (0,0): + split of THIS stage's row half 1 (its pieces are dead since the previous stage's last quarter)
//                  + requests: this stage's column half 1 of B, the next stage's row half 0 of A
//   quarter (1,0): + request: the next stage's row half 1 of A (its registers were read by the split just done)
//   quarter (0,1): + request: the next stage's column half 0 of B (half 0 was last read in the previous quarter); parked rows leave
//   quarter (1,1): + split of the NEXT stage's row half 0 (requested three quarters ago)
// so a B half is requested half a stage before its use, A a whole stage, and the 280 VALU instructions of a stage's splits are
// spread over two quarters.  On entry: P.a[0] = pieces of this stage's row half 0, raw1 = this stage's row half 1 as loaded,
// B0 = this stage's column half 0.
template <int NMMA, int NVALU, int NLOAD, int NSTORE, int NDS>
__device__ __forceinline__ void pin_quarter() {
#ifndef EQA_CGEMM_BF16_NOPIN
  if (NDS) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
  constexpr int kMem = NLOAD + NSTORE;
  constexpr int kValuPer = (NVALU + NMMA - 1) / NMMA;
#pragma unroll
  for (int k = 0; k < NMMA; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // one matrix instruction
    if (k < NLOAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);        // one load behind each of the first ones
    else if (k < kMem) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);    // ... then the stores
    if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);     // its share of the vector arithmetic
  }
#endif
}

template <int NPAIR, int TERMS>
__device__ __forceinline__ void run_stage(APieces& P, ARaw& raw0, ARaw& raw1, BHalf& B0, BHalf& B1, f32x16 (&acc)[3][2][2],
                                          const StageAddr& cur, const StageAddr& nxt, unsigned naoff0, unsigned naoff1, unsigned boff,
                                          const float* lds_lane, const ParkedDst& dst, int p0, const StageAddr& far, unsigned far_off,
                                          unsigned& touch) {
  constexpr int kQ = 3 * TERMS;          // matrix instructions per quarter
  constexpr int kSplit = 150;            // vector instructions of one row half's split (an upper bound for the pinning)
  f32x4 park[NPAIR > 0 ? NPAIR : 1];
  __builtin_amdgcn_sched_barrier(0);
#ifdef EQA_CGEMM_TOUCH     // experiment, measured SLOWER (3.46 vs 3.32 ms): not the missing look-ahead
  // A stage is 1.8 us of matrix instructions and A is requested three quarters of a stage ahead: less than an HBM round trip
  // under load.  One dword per lane (= per row of the tile: a row's stage is one 128-byte line) of the stage kPrefetch ahead pulls
  // those lines into L2 early; the value is never used (the previous stage's is retired here, a stage after its request).
  asm volatile("" ::"v"(touch));
  touch = __builtin_amdgcn_raw_buffer_load_b32(far.a, far_off, far.sa, 0);
#endif
  load_b(B1, cur, boff, 1);
  load_a(raw0, nxt, naoff0);
  split_a(P, raw1, 1);
  mma_quarter<TERMS>(P, B0, acc, 0, 0);
#ifdef EQA_CGEMM_TOUCH
  pin_quarter<kQ, kSplit, 14, 0, 0>();
#else
  pin_quarter<kQ, kSplit, 13, 0, 0>();
#endif
  __builtin_amdgcn_sched_barrier(0);
  load_a(raw1, nxt, naoff1);
  mma_quarter<TERMS>(P, B0, acc, 1, 0);
  pin_quarter<kQ, 0, 4, 0, 0>();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) park[k] = *reinterpret_cast<const f32x4*>(lds_lane + (p0 + k) * (2 * kLdsRowFloats));
  load_b(B0, nxt, boff, 0);
  mma_quarter<TERMS>(P, B1, acc, 0, 1);
#pragma unroll
  for (int k = 0; k < NPAIR; ++k) store_pair(dst, p0 + k, park[k]);
  pin_quarter<kQ, 0, 9, NPAIR, NPAIR>();
  __builtin_amdgcn_sched_barrier(0);
  split_a(P, raw0, 0);
  mma_quarter<TERMS>(P, B1, acc, 1, 1);
  pin_quarter<kQ, kSplit, 0, 0, 0>();
  __builtin_amdgcn_sched_barrier(0);
}

template <int NPAIR, int TERMS>
__global__ __launch_bounds__(256, 1) void fft_cgemm3m_bf16_kernel(const float* __restrict__ V, const uint16_t* __restrict__ Bp,
                                                                  float* __restrict__ Mo, int M, int pitch, int Cin, int Cout, int F,
                                                                  int n_rt, int n_ct, int waves_per_xcd) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & (kXcd - 1);
  const int q = (blockIdx.x >> 3) * 4 + wave;
  const int S = Cin / kStageK;
  const int wpf = n_rt * n_ct;
  const int nf_x = (F - xcd + kXcd - 1) / kXcd;
  const int total = nf_x * wpf;
  if (q >= total) return;
  const int i = lane & 31, h = lane >> 5;
  __shared__ __attribute__((aligned(16))) float lds_all[4 * kLdsWaveFloats];
  float* lds_w = lds_all + wave * kLdsWaveFloats;
  const float* lds_lane = lds_w + h * kLdsRowFloats + i * 4;
  const size_t rowf = (size_t)2 * Cin, mo_row = (size_t)2 * Cout;
  const unsigned b_stage_bytes = (unsigned)(Cout / 32) * 9 * kBFrag;        // bytes per (f, stage) of Bp
  const unsigned a_stage = 32u * 4u;
  const unsigned boff = lane * 16;

  auto locate = [&](int u, StageAddr& at, unsigned& aoff0, unsigned& aoff1, int& f, int& row0, int& ct, unsigned& toff) {
#ifdef EQA_CGEMM_SAMETILE     // experiment: every wave-tile reads tile 0 -- operands always cached
    u = 0;
#endif
    const int fi = u / wpf, r = u - fi * wpf;
    f = xcd + kXcd * fi;
    const int rt = r / n_ct;
    ct = r - rt * n_ct;
    row0 = rt * kTileM;
#ifdef EQA_CGEMM_SAME_A      // experiment: A always from (frequency 0, row tile 0)
    at.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V), 0, (unsigned)((size_t)pitch * rowf * 4), 0x00020000);
    at.sa = 0;
    aoff0 = (unsigned)((size_t)i * rowf + 4 * h) * 4u;
#else
    at.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V) + (size_t)f * pitch * rowf, 0, (unsigned)((size_t)pitch * rowf * 4), 0x00020000);
    at.sa = 0;
    aoff0 = (unsigned)((size_t)(row0 + i) * rowf + 4 * h) * 4u;
#endif
    aoff1 = aoff0 + 32 * (unsigned)rowf * 4u;
    toff = (unsigned)((size_t)(row0 + lane) * rowf) * 4u;                  // row `lane` of the tile: the prefetch touch
#ifdef EQA_CGEMM_SAME_B      // experiment: B always from (frequency 0, column tile 0)
    at.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp), 0, (unsigned)S * b_stage_bytes, 0x00020000);
    at.sb = 0;
#else
    at.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp) + (size_t)f * S * (b_stage_bytes / 2), 0, (unsigned)S * b_stage_bytes, 0x00020000);
    at.sb = (unsigned)(2 * ct) * (9 * kBFrag);
#endif
  };
  auto at_stage = [&](const StageAddr& t, int s) { return StageAddr{t.a, t.b, t.sa + s * a_stage, t.sb + s * b_stage_bytes}; };
  auto parked = [&](int f, int row0, int ct) {
    const int rows = min(kTileM, M - row0);
    ParkedDst d;
#ifdef EQA_CGEMM_NOSTORE     // experiment: an empty buffer drops the stores
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(Mo, 0, 0, 0x00020000);
#else
    d.rsrc = __builtin_amdgcn_make_buffer_rsrc(Mo + ((size_t)f * pitch + row0) * mo_row, 0, (unsigned)(rows * mo_row * 4), 0x00020000);
#endif
    d.voff = (h * (int)mo_row + ct * kLdsRowFloats + i * 4) * 4;
    d.pair_bytes = 2 * (int)mo_row * 4;
    d.soff = 0;
    return d;
  };

  StageAddr at;
  unsigned aoff0, aoff1;
  int f, row0, ct;
  unsigned toff;
  locate(q, at, aoff0, aoff1, f, row0, ct, toff);
  ParkedDst dst = parked(f, row0, ct);
  unsigned touch = 0;
  constexpr int kPrefetch = 3;             // stages between the touch of a row's line and the request of its operands
  dst.rsrc = __builtin_amdgcn_make_buffer_rsrc(Mo, 0, 0, 0x00020000);       // nothing parked yet: an empty buffer drops the stores
  ARaw raw0, raw1;
  APieces P;
  BHalf B0, B1;
  load_a(raw0, at, aoff0);
  load_a(raw1, at, aoff1);
  load_b(B0, at, boff, 0);
  split_a(P, raw0, 0);
  for (int u = q; u < total; u += waves_per_xcd) {
    f32x16 acc[3][2][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[p][m][n][e] = 0.f;
    const int un = u + waves_per_xcd < total ? u + waves_per_xcd : u;
    StageAddr nat;
    unsigned naoff0, naoff1, ntoff;
    int nf, nrow0, nct;
    locate(un, nat, naoff0, naoff1, nf, nrow0, nct, ntoff);
    for (int s = 0; s < S; ++s) {
      const bool more = s + 1 < S;
      if (NPAIR == 0) flush_rows(lds_lane, dst, (32 * s) / S, (32 * (s + 1)) / S);
      const bool far_here = s + kPrefetch < S;   // (S < kPrefetch: the touches run into the next tile's later stages -- harmless)
      run_stage<NPAIR, TERMS>(P, raw0, raw1, B0, B1, acc, at_stage(at, s), more ? at_stage(at, s + 1) : nat, more ? aoff0 : naoff0,
                              more ? aoff1 : naoff1, boff, lds_lane, dst, s * NPAIR,
                              far_here ? at_stage(at, s + kPrefetch) : at_stage(nat, s + kPrefetch - S), far_here ? toff : ntoff, touch);
    }
    // epilogue: Cr = T1 - T2, Ci = T3 - T1 - T2 into the wave's LDS tile (accumulator layout = that of the fp32 instruction)
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
    at = nat; aoff0 = naoff0; aoff1 = naoff1; f = nf; row0 = nrow0; ct = nct; toff = ntoff;
  }
  asm volatile("" ::"v"(touch));
  flush_rows(lds_lane, dst, 0, 32);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The BLOCK form (Cout a multiple of 128).  What bounds the wave form is not the matrix pipe but the vector L1: a 1-KB wave load
// returns every ~22 cycles per CU, and a wave-tile of 64 x 64 asks for 18 fragments of B and 8 loads of A per K-stage -- 104 loads
// per CU and stage = 2.3 k cycles against 2.3 k (six products) / 3.5 k (nine) of matrix instructions.  Here a block works on
// 128 rows x 128 columns of one frequency, wave w on columns 32 w .. 32 w + 31 of all 128 rows:
//   * A's K-stage is the same for the four waves: loaded and split ONCE per block (two rows per thread) and shared through LDS in
//     the fragment order of the instruction -- a half of the wave form's split arithmetic per matrix instruction, no A load per wave;
//   * B's K-stage is 9 fragments per wave instead of 18: 36 + 16 loads per CU and stage (1.1 k cycles of the L1).
//   LDS: three K-stages of A's pieces, 36 fragments [row quarter m][part r/i/s][piece] of 64 lanes x 16 bytes = 36 KB a stage.
//        Stage g + 2 is written while stage g is multiplied and stage g + 1 stands published: ONE barrier per stage, at the stage
//        boundary, and nothing is read right behind it (a stage's first fragments are read before the barrier that ends the
//        stage before: they were published a stage earlier).
//   splitter: thread (wave w, lane l) owns rows 16 w + (l >> 1 & 15) and + 64, k = 8 (l & 1) + 4 (l >> 5) + 0..3 of the stage:
//        four 16-byte loads (re, im of two rows), 24 values split, eighteen 8-byte LDS writes (lanes 0..31 of a write cover 512
//        contiguous bytes: no conflict).  The raw values of stage g + 4 are requested when those of g + 2 have been split.
//   B: a K-stage (9 fragments, 36 registers) a stage ahead in a second register set, requested in the stage's first quarter.
//   matrix instructions of a stage: four regions (row quarter m), each the three A pieces against every B piece: 27 (TERMS = 6:
//        18) instructions; a region reads the nine fragments it needs next from LDS behind its first nine matrix instructions.
//   the finished tile leaves straight from the accumulators: lane (column i, h) holds (Cr, Ci) of 16 rows per m -- 8-byte stores,
//        256 contiguous bytes per row.
constexpr int kBlkM = 128, kBlkN = 128;
constexpr int kFragBytes = 64 * 16;
constexpr int kABufs = 3;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// TERMS = 3: the fp16 form.  An fp32 operand x, scaled by a power of two into the top of the fp16 range, is split into TWO fp16
// pieces h1 = rn(x), h2 = rn(x - h1): 11 + 11 significant bits and a sign -> |x - h1 - h2| <= 2^-23 |x|, one fp32 ulp at worst (a third of one in rms); the
// product x y is taken as h1 k1 + h1 k2 + h2 k1 (each exact in v_mfma_f32_32x32x16_f16, fp32 accumulate; h2 k2 <= 2^-22 |x y| is
// left out).  Three matrix instructions per product instead of six, and -- fewer accumulator roundings -- CLOSER to an fp64
// product than the six bf16 pieces or the fp32 instruction (tools/micro/f16x2_gemm_check.hip, profiles/r06/f16x2_gemm_check.txt:
// rms 3.9e-8 / 5.1e-8 / 5.7e-8 on the parity test's magnitudes).  The price is the range: the caller hands over an upper bound of
// |V| (the DC bins of non-negative activations bound every other bin) and the filter spectra are scaled when they are split;
// values 2^16 below the bound keep full precision, smaller ones lose it gradually (fp16 subnormals: honoured by the instruction).
constexpr int pieces_of(int terms) { return terms == 3 ? 2 : 3; }
constexpr int a_stage_bytes(int terms) { return 12 * pieces_of(terms) * kFragBytes; }

template <int NP>
struct BSet {
  u32x4 b[3][NP];                 // [part][piece] of the wave's 32 columns
};
struct AFrags {
  u32x4 a[3];                     // [part] of one (row quarter, piece)
};
struct RawA {
  f32x4 re[2], im[2];             // rows r and r + 64
};
struct TileAt {                   // uniform: where a block tile's operands are
  __amdgpu_buffer_rsrc_t a, b;
  unsigned sb;
};

__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&o)[3]) {
  unsigned x[3][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float a = v[t];
    const float p1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & kHi);
    const float r1 = a - p1;
    const float p2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r1) & kHi);
    const float r2 = r1 - p2;
    x[0][t] = __builtin_bit_cast(unsigned, a);
    x[1][t] = __builtin_bit_cast(unsigned, r1);
    x[2][t] = __builtin_bit_cast(unsigned, r2);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    o[q][0] = __builtin_amdgcn_perm(x[q][1], x[q][0], 0x07060302u);
    o[q][1] = __builtin_amdgcn_perm(x[q][3], x[q][2], 0x07060302u);
  }
}
// two fp16 pieces of four (already scaled) values: v_cvt_pk_f16_f32 (round to nearest even), the remainder, again
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4h(const f32x4 v, u32x2 (&o)[2]) {
#ifdef EQA_BLK_NOSPLIT       // ablation: no split arithmetic (wrong values)
  o[0][0] = __builtin_bit_cast(unsigned, v[0]) & 0x3fff3fffu; o[0][1] = __builtin_bit_cast(unsigned, v[1]) & 0x3fff3fffu;
  o[1][0] = __builtin_bit_cast(unsigned, v[2]) & 0x3fff3fffu; o[1][1] = __builtin_bit_cast(unsigned, v[3]) & 0x3fff3fffu;
  return;
#endif
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    const f16x2v h = {(_Float16)a, (_Float16)b};
    const f16x2v l = {(_Float16)(a - (float)h[0]), (_Float16)(b - (float)h[1])};
    o[0][j] = __builtin_bit_cast(unsigned, h);
    o[1][j] = __builtin_bit_cast(unsigned, l);
  }
}
// one part (0 re, 1 im, 2 re + im) of row ROW's 4 k: split, NP 8-byte writes into the stage being built (row r + 64 is row
// quarter m + 2: 6 NP fragments further).  NP = 2: the values are scaled first (`scale`: a power of two).
template <int NP, int ROW, int PART>
__device__ __forceinline__ void split_part(const RawA& r, unsigned char* wr, const float scale) {
  const f32x4 v = PART == 0 ? r.re[ROW] : PART == 1 ? r.im[ROW] : r.re[ROW] + r.im[ROW];
  u32x2 o[NP];
  if constexpr (NP == 3) split4(v, o);
  else split4h(v * scale, o);
#pragma unroll
  for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x2*>(wr + (ROW * 6 * NP + PART * NP + q) * kFragBytes) = o[q];
}
template <int ROW>
__device__ __forceinline__ void load_raw(RawA& o, const TileAt& t, unsigned voff, unsigned row64, unsigned soff) {
  o.re[ROW] = __builtin_bit_cast(f32x4, buf_ld(t.a, voff + ROW * row64, soff));
  o.im[ROW] = __builtin_bit_cast(f32x4, buf_ld(t.a, voff + ROW * row64 + 64, soff));
}
template <int NP>
__device__ __forceinline__ void load_bset(BSet<NP>& o, const TileAt& t, unsigned voff, unsigned soff) {
#pragma unroll
  for (int k = 0; k < 3 * NP; ++k) o.b[k / NP][k % NP] = buf_ld(t.b, voff + k * kBFrag, t.sb + soff);
}
template <int NP, int M, int PA>
__device__ __forceinline__ void read_frags(AFrags& o, const unsigned char* rd) {
#ifdef EQA_BLK_NOLDSREAD     // ablation: the fragments of (row quarter 0, piece 0) stand for every one (wrong values)
  if (M != 0 || PA != 0) return;
#endif
#pragma unroll
  for (int p = 0; p < 3; ++p) o.a[p] = *reinterpret_cast<const u32x4*>(rd + ((M * 3 + p) * NP + PA) * kFragBytes);
}
template <int TERMS>
__device__ __forceinline__ f32x16 mma_t(const u32x4 a, const u32x4 b, const f32x16 c) {
  if constexpr (TERMS == 3) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return mma(a, b, c);
}
// FIRST: the tile's first K-stage -- the first product into an accumulator starts from zero (no accumulator is ever cleared)
template <int TERMS, int M, int PA, bool FIRST>
__device__ __forceinline__ void mma_rows(const AFrags& A, const BSet<pieces_of(TERMS)>& B, f32x16 (&acc)[3][4]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pb = 0; pb < pieces_of(TERMS); ++pb) {
    if ((TERMS == 6 && PA + pb > 2) || (TERMS == 3 && PA + pb > 1)) continue;
#pragma unroll
    for (int p = 0; p < 3; ++p) acc[p][M] = mma_t<TERMS>(A.a[p], B.b[p][pb], FIRST && PA == 0 && pb == 0 ? zero : acc[p][M]);
  }
}
// the order of a region's instructions: behind each matrix instruction its share of the LDS reads, loads, vector arithmetic, writes
template <int NMMA, int NDSR, int NLOAD, int NVALU, int NDSW>
__device__ __forceinline__ void pin_region() {
#ifndef EQA_CGEMM_BF16_NOPIN
  constexpr int kValuPer = (NVALU + NMMA - 1) / NMMA;
#pragma unroll
  for (int k = 0; k < NMMA; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (k < NDSR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (k < NLOAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, kValuPer, 0);
    if (k >= NMMA - NDSW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
  }
#endif
}

#if defined(EQA_BLK_CLOCK) && EQA_BLK_CLOCK >= 2
#define EQA_TICK(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(tk[i])); } while (0)
#else
#define EQA_TICK(i)
#endif

// One K-stage.  On entry: Bc = this stage's B, F0 = the fragments (row quarter 0, piece 0), raw = the raw values of stage g + 2
// (arrived); rd / rd_next = this lane's read address in this stage's / the next stage's LDS buffer, wr = its write address in
// stage g + 2's.  On exit: Bn = the next stage's B (requested), F0 = the next stage's (0, 0), raw = stage g + 4 (requested).
template <int TERMS, bool FIRST>
__device__ __forceinline__ void run_stage_block(const BSet<pieces_of(TERMS)>& Bc, BSet<pieces_of(TERMS)>& Bn, RawA& raw, AFrags& F0,
                                                AFrags& F1, AFrags& F2, AFrags& F3,
                                                f32x16 (&acc)[3][4], const unsigned char* rd, const unsigned char* rd_next,
                                                unsigned char* wr, const TileAt& tb, unsigned sb_off, unsigned b_voff, const TileAt& ta,
                                                unsigned sa_off, unsigned a_voff, unsigned row64, const float scale
#ifdef EQA_BLK_CLOCK
                                                , unsigned long long (&tot)[8]
#endif
                                                ) {
#ifdef EQA_BLK_CLOCK
  unsigned long long tk[8];
#endif
  constexpr int NP = pieces_of(TERMS);
  constexpr int kN = TERMS == 9 ? 27 : TERMS == 6 ? 18 : 9;   // matrix instructions of a region
  constexpr int kSplit = NP == 3 ? 32 : 20;                   // vector instructions of one split_part (an upper bound for the pinning)
  constexpr int kRd = 3 * NP;                                 // fragments a region reads: pieces 1.. of its row quarter, piece 0 of the next
  EQA_TICK(0);
  // region m = 0: B of the next stage; row r: re
  __builtin_amdgcn_sched_barrier(0);
  read_frags<NP, 0, 1>(F1, rd);
  if constexpr (NP == 3) read_frags<NP, 0, 2>(F2, rd);
  read_frags<NP, 1, 0>(F3, rd);
  load_bset<NP>(Bn, tb, b_voff, sb_off);
  split_part<NP, 0, 0>(raw, wr, scale);
  mma_rows<TERMS, 0, 0, FIRST>(F0, Bc, acc); mma_rows<TERMS, 0, 1, FIRST>(F1, Bc, acc);
  if constexpr (NP == 3) mma_rows<TERMS, 0, 2, FIRST>(F2, Bc, acc);
  pin_region<kN, kRd, kRd, kSplit, NP>();
  EQA_TICK(1);
  // region m = 1: row r: im, re + im
  __builtin_amdgcn_sched_barrier(0);
  read_frags<NP, 1, 1>(F1, rd);
  if constexpr (NP == 3) read_frags<NP, 1, 2>(F2, rd);
  read_frags<NP, 2, 0>(F0, rd);
  split_part<NP, 0, 1>(raw, wr, scale); split_part<NP, 0, 2>(raw, wr, scale);
  mma_rows<TERMS, 1, 0, FIRST>(F3, Bc, acc); mma_rows<TERMS, 1, 1, FIRST>(F1, Bc, acc);
  if constexpr (NP == 3) mma_rows<TERMS, 1, 2, FIRST>(F2, Bc, acc);
  pin_region<kN, kRd, 0, 2 * kSplit + 4, 2 * NP>();
  EQA_TICK(2);
  // region m = 2: row r + 64: re, im
  __builtin_amdgcn_sched_barrier(0);
  read_frags<NP, 2, 1>(F1, rd);
  if constexpr (NP == 3) read_frags<NP, 2, 2>(F2, rd);
  read_frags<NP, 3, 0>(F3, rd);
  split_part<NP, 1, 0>(raw, wr, scale); split_part<NP, 1, 1>(raw, wr, scale);
  mma_rows<TERMS, 2, 0, FIRST>(F0, Bc, acc); mma_rows<TERMS, 2, 1, FIRST>(F1, Bc, acc);
  if constexpr (NP == 3) mma_rows<TERMS, 2, 2, FIRST>(F2, Bc, acc);
  pin_region<kN, kRd, 0, 2 * kSplit, 2 * NP>();
  EQA_TICK(3);
  // region m = 3: row r + 64: re + im; the raw values of stage g + 4
  __builtin_amdgcn_sched_barrier(0);
  read_frags<NP, 3, 1>(F1, rd);
  if constexpr (NP == 3) read_frags<NP, 3, 2>(F2, rd);
  read_frags<NP, 0, 0>(F0, rd_next);
  split_part<NP, 1, 2>(raw, wr, scale);
  load_raw<0>(raw, ta, a_voff, row64, sa_off);
  mma_rows<TERMS, 3, 0, FIRST>(F3, Bc, acc); mma_rows<TERMS, 3, 1, FIRST>(F1, Bc, acc);
  if constexpr (NP == 3) mma_rows<TERMS, 3, 2, FIRST>(F2, Bc, acc);
  pin_region<kN, kRd, 2, kSplit + 4, NP>();
  __builtin_amdgcn_sched_barrier(0);
  load_raw<1>(raw, ta, a_voff, row64, sa_off);
  EQA_TICK(4);
  __builtin_amdgcn_sched_barrier(0);
  // the stage's pieces are written (mine: lgkmcnt), every wave is done with the buffer that stage g + 3 will be built in
#ifdef EQA_BLK_NOBARRIER
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#if defined(EQA_BLK_CLOCK) && EQA_BLK_CLOCK >= 2
  EQA_TICK(5);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 5; ++i) tot[i] += tk[i + 1] - tk[i];
#endif
}

#ifdef EQA_BLK_CLOCK
__device__ long long g_blk_clock[16];
#endif
template <int TERMS>
__global__ __launch_bounds__(256, 1) void fft_cgemm3m_bf16_block_kernel(const float* __restrict__ V, const uint16_t* __restrict__ Bp,
                                                                        float* __restrict__ Mo, int M, int pitch, int Cin, int Cout, int F,
                                                                        int n_rt, int n_cg, int blocks_per_xcd,
                                                                        const float* __restrict__ vbound, int nbound, float b_scale) {
  constexpr int NP = pieces_of(TERMS);
  constexpr int kAStageBytes = a_stage_bytes(TERMS);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & (kXcd - 1);
  const int q = blockIdx.x >> 3;
  const int S = Cin / kStageK;
  const int tpf = n_rt * n_cg;                                    // block tiles per frequency
  const int nf_x = (F - xcd + kXcd - 1) / kXcd;
  const int total = nf_x * tpf;
  if (q >= total) return;
#ifdef EQA_BLK_CLOCK
  const long long c0 = clock64(), w0 = wall_clock64();
  unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_epi = 0;
#define EQA_TOT , tot
#else
#define EQA_TOT
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // kABufs * kAStageBytes
  const size_t rowf = (size_t)2 * Cin, mo_row = (size_t)2 * Cout;
  const unsigned b_stage_bytes = (unsigned)(Cout / 32) * 3 * NP * kBFrag;
  const unsigned a_stage = 32u * 4u;
  const unsigned b_voff = lane * 16;
  const unsigned row64 = (unsigned)(64 * rowf * 4);
  // splitter: rows 16 w + (l >> 1 & 15) and + 64, k = 8 (l & 1) + 4 (l >> 5) + t
  const int sp_b = lane & 1, sp_r = (lane >> 1) & 15, sp_h = lane >> 5;
  const int sp_row = 16 * wave + sp_r;
  const unsigned sp_lds = (unsigned)((sp_row >> 5) * 3 * NP * kFragBytes + ((sp_row & 31) + 32 * sp_h) * 16 + 8 * sp_b);
  const unsigned sp_k = (unsigned)(8 * sp_b + 4 * sp_h) * 4u;

  auto locate = [&](int u, TileAt& at, unsigned& aoff, int& f, int& row0, int& ct) {
    const bool live = u < total;
    const int uu = live ? u : 0;
    const int fi = uu / tpf, r = uu - fi * tpf;
    f = xcd + kXcd * fi;
    const int rt = r / n_cg, cg = r - rt * n_cg;
    row0 = rt * kBlkM;
    ct = 4 * cg + wave;                                            // in units of 32 columns
    // past the block's last tile: empty descriptors (the look-ahead reads zeros, no traffic); rows past the pitch likewise
    at.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V) + (size_t)f * pitch * rowf, 0, live ? (unsigned)((size_t)pitch * rowf * 4) : 0u, 0x00020000);
    aoff = (unsigned)((size_t)(row0 + sp_row) * rowf * 4) + sp_k;
#ifdef EQA_BLK_SAME_B
    at.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp), 0, live ? (unsigned)S * b_stage_bytes : 0u, 0x00020000);
    at.sb = 0;
#else
    at.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp) + (size_t)f * S * (b_stage_bytes / 2), 0, live ? (unsigned)S * b_stage_bytes : 0u, 0x00020000);
    at.sb = (unsigned)ct * (3 * NP * kBFrag);
#endif
#ifdef EQA_BLK_SAME_A
    at.a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V), 0, live ? (unsigned)((size_t)pitch * rowf * 4) : 0u, 0x00020000);
    aoff = (unsigned)((size_t)(sp_row) * rowf * 4) + sp_k;
#endif
  };

  // three positions in the block's sequence of K-stages: A's loads (4 stages ahead), B's loads (1 ahead), the matrix instructions
  TileAt ta, tb;
  unsigned a_voff, b_unused;
  int ua = q, sa = 0, ub = q, sb = 0;
  int fa, rowa, cta, f, row0, ct;
  locate(q, ta, a_voff, f, row0, ct);
  tb = ta;
  auto next_a = [&]() {
    if (++sa == S) { sa = 0; ua += blocks_per_xcd; locate(ua, ta, a_voff, fa, rowa, cta); }
  };
  auto next_b = [&]() {
    if (++sb == S) { sb = 0; ub += blocks_per_xcd; locate(ub, tb, b_unused, fa, rowa, cta); }
  };

  // TERMS = 3: A is scaled by the power of two that takes the caller's bound of |V| (vbound[0 .. nbound): the largest counts) to
  // 2^14 (re + im then stays below 2^15.5 < 65504); the result is scaled back with B's factor in the epilogue.  Powers of two: exact.
  float a_scale = 1.0f, out_scale = 1.0f;
  if constexpr (TERMS == 3) {
    float mx = 0.0f;
    for (int t = lane; t < nbound; t += 64) mx = fmaxf(mx, vbound[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    int ex = 0;
    if (mx > 0.0f && mx < 3.0e38f) (void)frexpf(mx, &ex);      // mx <= 2^ex
    ex = min(max(14 - ex, -100), 100);
    ex = __builtin_amdgcn_readfirstlane(ex);
    a_scale = ldexpf(1.0f, ex);
    out_scale = ldexpf(1.0f, -ex) / b_scale;
  }
  RawA raw0, raw1;
  BSet<NP> B0, B1;
  AFrags F0, F1, F2, F3;
  // prologue: stages 0 and 1 of A built in buffers 0 and 1, stages 2 and 3 requested, stage 0 of B requested
  load_raw<0>(raw0, ta, a_voff, row64, sa * a_stage); load_raw<1>(raw0, ta, a_voff, row64, sa * a_stage); next_a();
  load_raw<0>(raw1, ta, a_voff, row64, sa * a_stage); load_raw<1>(raw1, ta, a_voff, row64, sa * a_stage); next_a();
  load_bset<NP>(B0, tb, b_voff, sb * b_stage_bytes); next_b();
  {
    unsigned char* w0p = lds + sp_lds;
    unsigned char* w1p = lds + kAStageBytes + sp_lds;
    split_part<NP, 0, 0>(raw0, w0p, a_scale); split_part<NP, 0, 1>(raw0, w0p, a_scale); split_part<NP, 0, 2>(raw0, w0p, a_scale);
    split_part<NP, 1, 0>(raw0, w0p, a_scale); split_part<NP, 1, 1>(raw0, w0p, a_scale); split_part<NP, 1, 2>(raw0, w0p, a_scale);
    split_part<NP, 0, 0>(raw1, w1p, a_scale); split_part<NP, 0, 1>(raw1, w1p, a_scale); split_part<NP, 0, 2>(raw1, w1p, a_scale);
    split_part<NP, 1, 0>(raw1, w1p, a_scale); split_part<NP, 1, 1>(raw1, w1p, a_scale); split_part<NP, 1, 2>(raw1, w1p, a_scale);
  }
  load_raw<0>(raw0, ta, a_voff, row64, sa * a_stage); load_raw<1>(raw0, ta, a_voff, row64, sa * a_stage); next_a();
  load_raw<0>(raw1, ta, a_voff, row64, sa * a_stage); load_raw<1>(raw1, ta, a_voff, row64, sa * a_stage); next_a();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  unsigned rbuf = 0;                                               // byte offset of the LDS buffer of the stage being multiplied
  read_frags<NP, 0, 0>(F0, lds + lane * 16);
  auto step = [&](unsigned o) { return o + kAStageBytes >= kABufs * kAStageBytes ? o + kAStageBytes - kABufs * kAStageBytes : o + kAStageBytes; };

  for (int u = q; u < total; u += blocks_per_xcd) {
    f32x16 acc[3][4];
#define EQA_STAGE(FIRST, BC, BN, RAW)                                                                                                 \
  {                                                                                                                                  \
    const unsigned r1 = step(rbuf), r2 = step(r1);                                                                                   \
    run_stage_block<TERMS, FIRST>(BC, BN, RAW, F0, F1, F2, F3, acc, lds + rbuf + lane * 16, lds + r1 + lane * 16, lds + r2 + sp_lds, \
                                  tb, sb * b_stage_bytes, b_voff, ta, sa * a_stage, a_voff, row64, a_scale EQA_TOT);                          \
    next_a(); next_b();                                                                                                              \
    rbuf = r1;                                                                                                                       \
  }
    EQA_STAGE(true, B0, B1, raw0)
    EQA_STAGE(false, B1, B0, raw1)
    for (int s = 2; s < S; s += 2) {
      EQA_STAGE(false, B0, B1, raw0)
      EQA_STAGE(false, B1, B0, raw1)
    }
#undef EQA_STAGE
    // Cr = T1 - T2, Ci = T3 - T1 - T2, straight from the accumulators: lane (i, h), slot e = row (e & 3) + 8 (e >> 2) + 4 h, column i
#ifdef EQA_BLK_CLOCK
    const long long e0 = clock64();
#endif
    {
      const int rows = __builtin_amdgcn_readfirstlane(min(kBlkM, M - row0));
#ifdef EQA_BLK_NOSTORE
      const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(Mo, 0, (unsigned)(rows * 0), 0x00020000);
#else
      const __amdgpu_buffer_rsrc_t dr =
          __builtin_amdgcn_make_buffer_rsrc(Mo + ((size_t)f * pitch + row0) * mo_row, 0, (unsigned)(rows * mo_row * 4), 0x00020000);
#endif
      const int i = lane & 31, h = lane >> 5;
      const unsigned voff = (unsigned)((4 * h * mo_row + (size_t)(32 * ct + i) * 2) * 4);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned roff = voff + (unsigned)((32 * m + (e & 3) + 8 * (e >> 2)) * mo_row * 4);     // in the range-checked offset
          const float t1 = acc[0][m][e], t2 = acc[1][m][e], t3 = acc[2][m][e];
          u32x2 c;
          c[0] = __builtin_bit_cast(unsigned, TERMS == 3 ? (t1 - t2) * out_scale : t1 - t2);
          c[1] = __builtin_bit_cast(unsigned, TERMS == 3 ? (t3 - t1 - t2) * out_scale : t3 - t1 - t2);
          __builtin_amdgcn_raw_buffer_store_b64(c, dr, roff, 0, 0);
        }
    }
    int nf, nrow0, nct;
    TileAt unused_t;
    unsigned unused_o;
    locate(u + blocks_per_xcd, unused_t, unused_o, nf, nrow0, nct);
    f = nf; row0 = nrow0; ct = nct;
#ifdef EQA_BLK_CLOCK
    t_epi += clock64() - e0;
#endif
  }
#ifdef EQA_BLK_CLOCK
  if (blockIdx.x == 100 && threadIdx.x == 0) {
    g_blk_clock[0] = clock64() - c0; g_blk_clock[1] = wall_clock64() - w0; g_blk_clock[2] = t_epi;
    for (int i = 0; i < 7; ++i) g_blk_clock[3 + i] = (long long)tot[i];
  }
#endif
}

// B3 (F, S, Cout/32, 3, 2, 64, 4) fp32 -> Bp (F, S, Cout/32, 3, 3, 64, 8) bf16: one thread per (f, s, column tile, part, lane)
__global__ __launch_bounds__(256) void spectra3m_split_kernel(const float* __restrict__ B3, uint16_t* __restrict__ Bp, size_t groups) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= groups * 64) return;
  const size_t g = t >> 6;            // (f, s, nt, part)
  const int lane = (int)(t & 63);
  const float* src = B3 + g * (2 * 64 * 4) + lane * 4;
  const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 64 * 4);
  const Pieces p = split8(lo, hi);
  u32x4* dst = reinterpret_cast<u32x4*>(Bp + g * (3 * 64 * 8)) + lane;
#pragma unroll
  for (int q = 0; q < 3; ++q) dst[q * 64] = p.p[q];
}


// ... -> Bh (F, S, Cout/32, 3, 2, 64, 8) fp16: the two pieces of b_scale * B3 (b_scale: a power of two that takes max |B3| to 2^14)
__global__ __launch_bounds__(256) void spectra3m_split_f16_kernel(const float* __restrict__ B3, uint16_t* __restrict__ Bh, size_t groups,
                                                                  float b_scale) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= groups * 64) return;
  const size_t g = t >> 6;            // (f, s, nt, part)
  const int lane = (int)(t & 63);
  const float* src = B3 + g * (2 * 64 * 4) + lane * 4;
  const f32x4 lo = *reinterpret_cast<const f32x4*>(src) * b_scale, hi = *reinterpret_cast<const f32x4*>(src + 64 * 4) * b_scale;
  u32x2 a[2], b[2];
  split4h(lo, a);
  split4h(hi, b);
  u32x4* dst = reinterpret_cast<u32x4*>(Bh + g * (2 * 64 * 8)) + lane;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const u32x4 v = {a[q][0], a[q][1], b[q][0], b[q][1]};
    dst[q * 64] = v;
  }
}

}  // namespace

namespace eqa {
int g_cgemm_bf16_form = 0;     // eqa_set_option key 2: 0 = the block form where it applies (Cout % 128 == 0), 1 = always the wave form
}

extern "C" {

int64_t eqa_fft48k5_spectra3m_bf16_bytes(int Cin, int Cout) {
  if (!eqa_fft48k5_cgemm3m_supported(Cin, Cout)) return 0;
  return (int64_t)eqa_fft48k5_frequencies() * Cin * Cout * 3 * 3 * 2;
}

#ifdef EQA_BLK_CLOCK
int eqa_debug_blk_clock(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_clock), 80); }
#endif

int eqa_fft48k5_spectra3m_split(const float* B3, void* Bp, int Cin, int Cout, void* stream) {
  if (!B3 || !Bp) return EQA_ERR_INVALID_ARG;
  if (!eqa_fft48k5_cgemm3m_supported(Cin, Cout) || (((uintptr_t)B3 | (uintptr_t)Bp) & 15)) return EQA_ERR_UNSUPPORTED;
  const size_t groups = (size_t)eqa_fft48k5_frequencies() * (Cin / kStageK) * (Cout / 32) * 3;
  const size_t threads = groups * 64;
  hipLaunchKernelGGL(spectra3m_split_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B3,
                     static_cast<uint16_t*>(Bp), groups);
  return launch_status();
}

int eqa_fft48k5_cgemm3m_bf16x3(const float* V, const void* Bp, float* Mo, int64_t M, int Cin, int Cout, int terms, void* stream) {
  if (!V || !Bp || !Mo || M < 0 || Cin <= 0 || Cout <= 0 || (terms != 9 && terms != 6)) return EQA_ERR_INVALID_ARG;
  if (M == 0) return EQA_OK;
  const int F = eqa_fft48k5_frequencies();
  const int64_t fm_bytes = ((M | 1) + 64) * 2 * (int64_t)std::max(Cin, Cout) * 4;
  if (!eqa_fft48k5_cgemm3m_supported(Cin, Cout) || M > 0x3fffff || fm_bytes > 0x7fffffffLL || (int64_t)Cin * Cout * 9 * 2 > 0x7fffffffLL ||
      (((uintptr_t)V | (uintptr_t)Bp | (uintptr_t)Mo) & 15))
    return EQA_ERR_UNSUPPORTED;
  const int n_rt = (int)((M + kTileM - 1) / kTileM), n_ct = Cout / kTileN;
  const int blocks = 256;
  const int S = Cin / kStageK;
  if (Cout % kBlkN == 0 && eqa::g_cgemm_bf16_form != 1) {      // the block form: A split once per block, shared through LDS
    const int n_rb = (int)((M + kBlkM - 1) / kBlkM), n_cg = Cout / kBlkN;
    constexpr int kLds = kABufs * a_stage_bytes(9);
    const void* kern = terms == 9 ? (const void*)fft_cgemm3m_bf16_block_kernel<9> : (const void*)fft_cgemm3m_bf16_block_kernel<6>;
    if (!allow_dynamic_lds(kern, kLds)) return EQA_ERR_LAUNCH;
    if (terms == 9)
      hipLaunchKernelGGL((fft_cgemm3m_bf16_block_kernel<9>), dim3(blocks), dim3(256), kLds, (hipStream_t)stream, V,
                         static_cast<const uint16_t*>(Bp), Mo, (int)M, (int)eqa_fft48k5_tile_pitch(M), Cin, Cout, F, n_rb, n_cg, blocks / kXcd,
                         (const float*)nullptr, 0, 1.0f);
    else
      hipLaunchKernelGGL((fft_cgemm3m_bf16_block_kernel<6>), dim3(blocks), dim3(256), kLds, (hipStream_t)stream, V,
                         static_cast<const uint16_t*>(Bp), Mo, (int)M, (int)eqa_fft48k5_tile_pitch(M), Cin, Cout, F, n_rb, n_cg, blocks / kXcd,
                         (const float*)nullptr, 0, 1.0f);
    return launch_status();
  }
#define EQA_CG_LAUNCH(NP, T)                                                                                                       \
  hipLaunchKernelGGL((fft_cgemm3m_bf16_kernel<NP, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, V, static_cast<const uint16_t*>(Bp), \
                     Mo, (int)M, (int)eqa_fft48k5_tile_pitch(M), Cin, Cout, F, n_rt, n_ct, (blocks / kXcd) * 4)
#define EQA_CG_TERMS(NP) do { if (terms == 9) EQA_CG_LAUNCH(NP, 9); else EQA_CG_LAUNCH(NP, 6); } while (0)
  switch (32 % S == 0 ? 32 / S : 0) {
    case 4: EQA_CG_TERMS(4); break;
    case 2: EQA_CG_TERMS(2); break;
    case 1: EQA_CG_TERMS(1); break;
    default: EQA_CG_TERMS(0); break;
  }
#undef EQA_CG_TERMS
#undef EQA_CG_LAUNCH
  return launch_status();
}


int64_t eqa_fft48k5_spectra3m_f16_bytes(int Cin, int Cout) {
  if (!eqa_fft48k5_cgemm3m_supported(Cin, Cout) || Cout % kBlkN != 0) return 0;
  return (int64_t)eqa_fft48k5_frequencies() * Cin * Cout * 3 * 2 * 2;
}

int eqa_fft48k5_spectra3m_split_f16(const float* B3, void* Bh, int Cin, int Cout, float b_scale, void* stream) {
  if (!B3 || !Bh || !(b_scale > 0.0f)) return EQA_ERR_INVALID_ARG;
  if (eqa_fft48k5_spectra3m_f16_bytes(Cin, Cout) == 0 || (((uintptr_t)B3 | (uintptr_t)Bh) & 15)) return EQA_ERR_UNSUPPORTED;
  int ex = 0;
  if (frexpf(b_scale, &ex) != 0.5f) return EQA_ERR_INVALID_ARG;            // a power of two: the scaling must be exact
  const size_t groups = (size_t)eqa_fft48k5_frequencies() * (Cin / kStageK) * (Cout / 32) * 3;
  const size_t threads = groups * 64;
  hipLaunchKernelGGL(spectra3m_split_f16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B3,
                     static_cast<uint16_t*>(Bh), groups, b_scale);
  return launch_status();
}

int eqa_fft48k5_cgemm3m_f16x2(const float* V, const void* Bh, float* Mo, int64_t M, int Cin, int Cout, const float* vbound, int nbound,
                              float b_scale, void* stream) {
  if (!V || !Bh || !Mo || M < 0 || Cin <= 0 || Cout <= 0 || !vbound || nbound <= 0 || !(b_scale > 0.0f)) return EQA_ERR_INVALID_ARG;
  if (M == 0) return EQA_OK;
  const int F = eqa_fft48k5_frequencies();
  const int64_t fm_bytes = ((M | 1) + 64) * 2 * (int64_t)std::max(Cin, Cout) * 4;
  if (eqa_fft48k5_spectra3m_f16_bytes(Cin, Cout) == 0 || M > 0x3fffff || fm_bytes > 0x7fffffffLL || (int64_t)Cin * Cout * 6 * 2 > 0x7fffffffLL ||
      (((uintptr_t)V | (uintptr_t)Bh | (uintptr_t)Mo) & 15) || (Cin / kStageK) % 2 != 0)
    return EQA_ERR_UNSUPPORTED;
  const int blocks = 256;
  const int n_rb = (int)((M + kBlkM - 1) / kBlkM), n_cg = Cout / kBlkN;
  constexpr int kLds = kABufs * a_stage_bytes(3);
  if (!allow_dynamic_lds((const void*)fft_cgemm3m_bf16_block_kernel<3>, kLds)) return EQA_ERR_LAUNCH;
  hipLaunchKernelGGL((fft_cgemm3m_bf16_block_kernel<3>), dim3(blocks), dim3(256), kLds, (hipStream_t)stream, V,
                     static_cast<const uint16_t*>(Bh), Mo, (int)M, (int)eqa_fft48k5_tile_pitch(M), Cin, Cout, F, n_rb, n_cg, blocks / kXcd,
                     vbound, nbound, b_scale);
  return launch_status();
}

}  // extern "C"
