// libeqa_hip.so, part 17 -- the lifting convolution FUSED into the forward FFT-48 transform of the layer behind it (I2a, inference).
// C ABI: include/eqa_hip.h (eqa_lift5_fft48k5_input).  Reference layers: escnn_networks.py:60-85 (R2Conv trivial -> regular, InnerBatchNorm
// folded, ReLU, then the first regular -> regular R2Conv, whose input transform this is).
//
// Unfused (rounds 2-5): eqa_lift_conv_grouped writes the lifted map (256 x 92 x 92 x 256 floats = 2.2 GB at the headline shape) and
// eqa_fft48k5_input_grouped reads it straight back -- 4.4 GB of HBM traffic for an activation whose every 48 x 48 x 16-channel tile
// is a function of a 52 x 52 x 3 input patch (32 KB).  Here a block owns (tile, 16-channel group) items like the fused forward
// transform and PRODUCES the tile it transforms:
//
//   conv      the tile's 2304 pixels x 16 channels x K = 76 on v_mfma_f32_16x16x4_f32 (weights = A: channels, pixels = B), bias as
//             the C operand of the first instruction, ReLU, zero outside the map; four sub-phases of 12 tile rows each, whose 16 input
//             rows (10 KB) are staged in LDS (prefetched into registers one sub-phase ahead); wave w computes tile row 12 sub + w
//             as three 16-pixel tiles and writes it into the tile buffer with one 16-byte LDS store per pixel quad of channels
//   rows      thread (row, channel): the 48-point real transform of its row, IN PLACE: 48 reals in, 23 complex + 2 real bins out
//             (the DC and Nyquist bins of a real row are real), so the tile buffer (48 x 48 x 16 floats = 147 KB of the CU's 160)
//             also is the spectrum buffer
//   columns   thread (kx, channel), 24 x 16 of them: the two real columns kx = 0 / 24 ride in ONE complex transform (z = c0 + i c24,
//             separated afterwards -- as fft48_fwd_pipe_kernel does), 48-point complex transform, spectra out in the layout of
//             eqa_fft48k5_input ([Re x 16 | Im x 16] per channel group, frequency-major, odd tile pitch)
//
// Blocks are persistent (one per CU, 12 waves) and software-pipelined over their items: the column transform + the 96 stores of
// item i run on waves 0-5 while waves 6-11 compute the first 12 rows of item i + 1 on the matrix cores, and the stores drain
// under the rest of that item's convolution.
//
// The order of the 76 products of a pixel is the order of lift_conv_dense_kernel (csrc/lift_conv.hip: filter-row elements paired
// j | j + 8, the five j = 7 elements last), four per matrix instruction instead of two.
#include <cstdlib>

#include "eqa_common.hpp"

#ifndef EQA_LF_AUX
#define EQA_LF_AUX 2   // cache policy bits of the spectrum stores (2 = non-temporal)
#endif

namespace {

#include "fft_common.inc"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef EQA_LF_CLOCK
// Debug build: shader cycles per phase, summed over the items of block 0, by thread 0 of a column wave ([0..15]) and of a
// convolution-only wave ([16..31]): tools/probe_lf_clock.py.
__device__ unsigned long long g_lf_clock[32];
#define LF_CLOCK_BEGIN() const bool lfc_on = blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 1 || wave == 4 || wave == 6); \
  unsigned long long lfc_t = __builtin_readcyclecounter(); unsigned long long lfc_s[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define LF_CLOCK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); lfc_s[i] += n_ - lfc_t; lfc_t = n_; } while (0)
#define LF_CLOCK_END() do { if (lfc_on) for (int i_ = 0; i_ < 8; ++i_) g_lf_clock[8 * ROLE + i_] = lfc_s[i_]; } while (0)
#else
#define LF_CLOCK_BEGIN() do { } while (0)
#define LF_CLOCK(i) do { } while (0)
#define LF_CLOCK_END() do { } while (0)
#endif

constexpr int kLfThreads = 768;                 // 12 waves
constexpr int kLfCh = 16;                       // channels per item
constexpr int kLfSteps = 19;                    // 76 products = 19 x (16x16x4)
constexpr int kLfQuadPitch = 196;               // floats per (row, channel quad): 48 slots x 4 channels + 4 (bank skew between quads)
constexpr int kLfRowPitch = 784;                // floats per tile row: 4 quads x 196 = 16 (mod 32) -> consecutive rows land 16 banks apart
constexpr int kLfTileFloats = kFftN * kLfRowPitch;              // 37,632 floats = 150,528 bytes
constexpr int kLfPatchRows = 16, kLfPatchPitch = 156;           // 12 output rows need 16 input rows of 52 pixels x 3 channels
constexpr int kLfPatchFloats = kLfPatchRows * kLfPatchPitch;    // 2,496 floats
constexpr int kLfLdsFloats = kLfTileFloats + kLfPatchFloats;    // 160,512 bytes of the CU's 163,840
constexpr int kLfStagers = 6 * 64;               // the convolution and row waves stage the patch (the column waves' registers are full)
constexpr int kLfPre = (kLfPatchFloats + kLfStagers - 1) / kLfStagers;   // staged floats per staging thread and sub-phase (7)

// slot n of the product order -> (filter row ky, element e = kx * 3 + ci); n = 75: the spare (weight 0, operand = element (4, 7) again)
__device__ __forceinline__ void lf_slot(int n, int& ky, int& e, bool& spare) {
  // (selects, not branches: n depends on the lane, and a branch here put every weight load of `setup` under its own exec test)
  spare = n == 75;
  const bool lo = n < 70;
  const int r = n % 14;
  ky = lo ? n / 14 : min(n - 70, 4);
  e = lo ? (r >> 1) + 8 * (r & 1) : 7;
}

struct LfItem {      // wave-uniform
  unsigned grp, m;   // channel group, tile index (img * TY + ty) * TX + tx
  int gy0, gx0;      // first lifted-map row / column of the tile = first input row / column of its patch
  size_t img;
};

__device__ __forceinline__ LfItem lf_item(unsigned work, int ngrp, int TY, int TX) {
  LfItem it;
  it.grp = work % ngrp;
#ifdef EQA_LF_XCDGRP   // experiment: the two groups an XCD's blocks work on are NEIGHBOURS (256-byte runs of a spectrum row per L2)
  if (ngrp == 16) it.grp = 2 * (it.grp % 8) + it.grp / 8;
#endif
  it.m = work / ngrp;
  const unsigned tx = it.m % TX, ty = (it.m / TX) % TY;
  it.img = it.m / ((unsigned)TX * TY);
  it.gy0 = kFftO * (int)ty;
  it.gx0 = kFftO * (int)tx;
  return it;
}

// x: (nimg, H0, W0, 3) channels-last input; bank: (Cout, 5, 5, 3) = the memory order of a channels-last (Cout, 3, 5, 5) filter bank;
// bias: (Cout) or null; V: (F, M | 1, 2 Cout) spectra.  H1 = H0 - 4, W1 = W0 - 4: the lifted map the tiles cover.
//
// ROLE (a block's 12 waves, one instantiation each so that a role's registers are live in its own code only; all three pass the
// same barriers in the same order -- ten per item):
//   kConv  waves 0..3, one per SIMD: the item's 48 tile rows, three per sub-phase (rows 12 s + 3 k + {0, 1, 2}); a wave's 57 matrix
//          instructions per row are back-to-back accumulator chains, which keep the matrix pipe of its SIMD busy by themselves
//   kRows  waves 4..5: the row transforms of the 12 rows the PREVIOUS sub-phase finished (three passes of four consecutive rows x 16
//          channels) -- they run beside the convolution, not behind it
//   kCols  waves 6..11: the column transform of the PREVIOUS item in sub-phase 0 and its stores in four chunks, one per sub-phase
//          (the store path of a CU moves ~10 bytes per clock: an item's 147 KB take ~15 k cycles, which the convolution of the
//          next item hides); then the column read of this item
// What no role can hide is the tail of an item: the row transforms of its last 12 rows and the column read.
enum { kConv = 0, kRows = 1, kCols = 2 };

template <int ROLE>
__device__ __forceinline__ void lift5_fft48_body(const float* __restrict__ x, const float* __restrict__ bank, const float* __restrict__ bias,
                                                 int relu, float* __restrict__ V, int H0, int W0, int C, int TY, int TX, size_t Mp,
                                                 unsigned nwork, unsigned v_bytes, unsigned x_bytes, unsigned bank_bytes) {
  extern __shared__ float lds[];
  float* const tile = lds;
  float* const patch = lds + kLfTileFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int ngrp = C / kLfCh;
  const int H1 = H0 - 4, W1 = W0 - 4;
  const unsigned nblk = gridDim.x;
  const unsigned patch_b = (unsigned)(kLfTileFloats * 4);
  LF_CLOCK_BEGIN();   // per role: [0] stage + barrier 1, [1] prefetch issue, [2] the role's work of a sub-phase, [3] barrier 2,
                      // [4] tail row passes, [5] barrier 3, [6] column read, [7] barrier 4

  // ---- the input patch of a sub-phase: 16 input rows x 52 pixels x 3 channels, staged by the threads of waves 0..5
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  float pre[kLfPre];        // the next sub-phase's share (rows / columns outside the image: the range check's 0)
  auto prefetch = [&](const LfItem& it, int sub, bool live) {
    if constexpr (ROLE == kCols) return;
    int tq = tid;     // opaque: the index arithmetic below is redone per call (hoisted out of the item loop it went to scratch)
    asm volatile("" : "+v"(tq));
#pragma unroll
    for (int i = 0; i < kLfPre; ++i) {
      const int idx = tq + kLfStagers * i;
      const int pr = idx / kLfPatchPitch, pc = idx - pr * kLfPatchPitch;
      const int gy = it.gy0 + 12 * sub + pr, gxc = it.gx0 * 3 + pc;
      const bool ok = live && idx < kLfPatchFloats && gy < H0 && gxc < W0 * 3;
      const unsigned off = ok ? (unsigned)(((it.img * H0 + gy) * (size_t)(W0 * 3) + gxc) * 4) : 0xfffffff0u;
      pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
    }
  };
  auto stage = [&]() {
    if constexpr (ROLE == kCols) return;
#pragma unroll
    for (int i = 0; i < kLfPre; ++i) {
      const int idx = tid + kLfStagers * i;
      if (idx < kLfPatchFloats) patch[idx] = pre[i];
    }
  };

  // ---- kConv: the lane's LDS byte offset of step t's operand (patch row 3 k + ky, element e of pixel j) and its weight (channel
  // i = lane % 16, k-slot q of step t), the bias of the lane's four output channels
  const int ck = ROLE == kConv ? wave : 0;
  unsigned a_off[kLfSteps];
  float wreg[kLfSteps];
  f32x4 bias4;
  auto setup = [&](const LfItem& it) {
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bank), 0, bank_bytes, 0x00020000);
    const unsigned wb = (unsigned)((it.grp * kLfCh + j) * 75 * 4);
#pragma unroll
    for (int t = 0; t < kLfSteps; ++t) {
      int ky, e;
      bool spare;
      lf_slot(4 * t + q, ky, e, spare);
      a_off[t] = patch_b + (unsigned)(((3 * ck + ky) * kLfPatchPitch + j * 3 + e) * 4);
      // (the spare slot's weight: an offset beyond the bank, i.e. the range check's 0)
      wreg[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, spare ? 0xfffffff0u : wb + (unsigned)((ky * 15 + e) * 4), 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = bias ? bias[it.grp * kLfCh + 4 * q + r] : 0.0f;
  };
  // One tile row (3 tiles of 16 pixels): patch rows 3 k + drow .. + 4 -> tile row y.  All 57 operands are requested first (immediate
  // offsets off the per-lane addresses of `setup`: no address arithmetic in the stream), then the 57 matrix instructions.  (Left to
  // the compiler the stream was read -> wait -> multiply per step: 100-160 cycles per matrix instruction and wave.)
  auto conv_row = [&](const LfItem& it, int drow, int y) {
    const bool row_ok = it.gy0 + y < H1;
    float b[3][kLfSteps];
#pragma unroll
    for (int t = 0; t < kLfSteps; ++t)
#pragma unroll
      for (int tx3 = 0; tx3 < 3; ++tx3)
        b[tx3][t] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + a_off[t])[drow * kLfPatchPitch + tx3 * 48];
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[3] = {bias4, bias4, bias4};
#pragma unroll
    for (int t = 0; t < kLfSteps; ++t)
#pragma unroll
      for (int tx3 = 0; tx3 < 3; ++tx3) acc[tx3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t], b[tx3][t], acc[tx3], 0, 0, 0);
#pragma unroll
    for (int tx3 = 0; tx3 < 3; ++tx3) {
      const bool ok = row_ok && it.gx0 + tx3 * 16 + j < W1;
      f32x4 v = acc[tx3];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float u = relu ? fmaxf(v[r], 0.0f) : v[r];
        v[r] = ok ? u : 0.0f;
      }
      *reinterpret_cast<f32x4*>(tile + y * kLfRowPitch + q * kLfQuadPitch + (tx3 * 16 + j) * 4) = v;
    }
  };

  // ---- row pass of four consecutive tile rows y0 .. y0 + 3: thread (r = lane / 16, c = lane % 16), 48 reals -> 23 complex + 2 real
  // bins IN PLACE (slot kx: Re[kx], kx = 0..23; slot 24: Re[24]; slot 24 + kx: Im[kx]).  Row pitch 784 = 16 (mod 32): the rows of
  // a 32-lane group sit 16 banks apart.
  auto row_pass = [&](int y0) {
    float* const rowp = tile + (y0 + (lane >> 4)) * kLfRowPitch + ((lane & 15) >> 2) * kLfQuadPitch + (lane & 3);
    float re[kFftN], ore[kFftH], oim[kFftH];
#pragma unroll
    for (int xx = 0; xx < kFftN; ++xx) re[xx] = rowp[xx * 4];
    fft48_r2c(re, ore, oim);
#pragma unroll
    for (int k = 0; k < kFftH - 1; ++k) rowp[k * 4] = ore[k];
    rowp[24 * 4] = ore[24];
#pragma unroll
    for (int k = 1; k < kFftH - 1; ++k) rowp[(24 + k) * 4] = oim[k];
  };

  // ---- kCols (and the edge columns on wave 4).  Lane l = (channel-in-quad ci = l / 16, task kxi = (l % 16) / 4, channel quad
  // cq = l % 4): the four channels of a quad sit in the four 16-lane rows at the same position, so that the 4 x 4 (frequency,
  // channel) blocks in front of the stores transpose with v_permlane16_swap / v_permlane32_swap; the tasks kxi = 2 a and 2 a + 1 are
  // lanes l and l ^ 4, which trade their imaginary / real quads (ds_swizzle) so that EIGHT lanes store one whole 128-byte line
  // [Re x 16 | Im x 16] -- written as two 64-byte halves by two instructions the store path ran at 3.0 TB/s, as whole lines it runs
  // at 5.2 (tools/micro/store_pattern.hip).
  // Column waves: task kc of {b, b + 4, b + 1, b + 5}, b = 8 (w / 2) + 2 (w % 2), kc = kx = 1..23 (kc = 0: an idle slot, its
  // stores are masked).  The two REAL columns kx = 0 / 24 (the DC and Nyquist bins of real rows) are 32 real-input transforms on
  // wave 4 (which has no row pass to run in sub-phase 0): tasks kxi 0 / 1 = kx 0 / 24, 25 stored frequencies each.
  const int cw = ROLE == kCols ? wave - 6 : 0;
  const int col_ci = lane >> 4, col_kxi = (lane >> 2) & 3, col_cq = lane & 3;
  const int kc_of0 = 8 * (cw >> 1) + 2 * (cw & 1) + (col_kxi >> 1);       // the pair's even task (kxi & ~1)
  const int kc = kc_of0 + ((col_kxi & 1) ? 4 : 0);
  const bool odd = (col_kxi & 1) != 0;
  const float* const colp = tile + col_cq * kLfQuadPitch + col_ci + kc * 4;   // Re: slot kc; Im: slot 24 + kc (+ 96 floats)
  const float* const edgep = tile + col_cq * kLfQuadPitch + col_ci + (odd ? 24 * 4 : 0);   // wave 4: slot 0 (Re of kx 0) | slot 24 (kx 24)
  const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(V, 0, v_bytes, 0x00020000);
  float cre[kFftN], cim[kFftN];      // column data read from LDS, then (in place) the spectrum waiting to be stored (unused per role: dead)
  unsigned pend_m = 0, pend_grp = 0;
  bool pending = false;
  // 4 x 4 transpose of (register k, lane row i) across the wave's four 16-lane rows: afterwards register k of row i holds what
  // register i held in row k.  (Inline asm with the hazard's two wait states inside the string: chained through the builtins'
  // two-element results, hipcc 7.2 folded the second element into the first -- tools/micro/permlane_swap.hip.)
  auto transpose4 = [&](float& r0, float& r1, float& r2, float& r3) {
    auto sw16 = [](float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); };
    auto sw32 = [](float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); };
    sw16(r0, r1);
    sw16(r2, r3);
    sw32(r0, r2);
    sw32(r1, r3);
  };
  auto swz4 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x101f)); };   // lane l <- lane l ^ 4
  // One frequency group g (ky 4 g .. 4 g + 3; lane row i stores ky = 4 g + i) of a lane pair: store 1 writes the line of the pair's
  // EVEN task (even lanes: its real quad, odd lanes: its imaginary quad, received), store 2 the odd task's.  voff1 / voff2: the
  // lane's byte offsets of the two lines (0xfffffff0: masked by the range check).
  auto store_pair = [&](float* re, float* im, unsigned voff1, unsigned voff2) {
    transpose4(re[0], re[1], re[2], re[3]);
    transpose4(im[0], im[1], im[2], im[3]);
    f32x4 o1, o2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float got = swz4(odd ? re[r] : im[r]);
      o1[r] = odd ? got : re[r];
      o2[r] = odd ? im[r] : got;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), vr, voff1, 0, EQA_LF_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o2), vr, voff2, 0, EQA_LF_AUX);
  };
  // kCols: the pending item's share of sub-phase SUB: the transform in sub-phase 0, three of its twelve frequency groups in each.
  // Frequency of (kx, ky), interior kx: 23 ky + kx - 1.
  auto col_work = [&](auto SUB) {
    constexpr int sub = decltype(SUB)::value;
    if (!pending) return;
    if (sub == 0) {
      float ore[kFftN], oim[kFftN];
      fft48(cre, cim, ore, oim);
#pragma unroll
      for (int ky = 0; ky < kFftN; ++ky) {
        cre[ky] = ore[ky];
        cim[ky] = oim[ky];
      }
    }
#ifndef EQA_LF_NOSTORE
    const size_t rowb = Mp * 2 * (size_t)C * 4;                         // bytes per stored frequency
    const unsigned col = (unsigned)(((size_t)pend_m * 2 * C + pend_grp * 2 * kLfCh) * 4) + (unsigned)((odd ? 64 : 0) + 16 * col_cq);
    const unsigned vb1 = (unsigned)((size_t)(kFftInner * col_ci + kc_of0 - 1) * rowb) + col;     // (kc_of0 = 0: masked below)
    const unsigned vb2 = (unsigned)((size_t)(kFftInner * col_ci + kc_of0 + 3) * rowb) + col;
    const unsigned vs = (unsigned)(4 * kFftInner * rowb);
#pragma unroll
    for (int g = 3 * sub; g < 3 * sub + 3; ++g)
      store_pair(&cre[4 * g], &cim[4 * g], kc_of0 == 0 ? 0xfffffff0u : vb1 + (unsigned)g * vs, vb2 + (unsigned)g * vs);
#endif
  };
  // wave 4, sub-phase 0: the two real edge columns of the pending item.  cre = the column (48 reals); frequencies 1104 + 2 ky (kx = 0)
  // and 1105 + 2 ky (kx = 24), ky = 0..24.
  auto edge_work = [&]() {
    if (!pending) return;
    float ore[28], oim[28];
    {
      float o25r[kFftH], o25i[kFftH];
      fft48_r2c(cre, o25r, o25i);
#pragma unroll
      for (int k = 0; k < 28; ++k) {
        ore[k] = k < kFftH ? o25r[k] : 0.0f;
        oim[k] = k < kFftH ? o25i[k] : 0.0f;
      }
    }
#ifndef EQA_LF_NOSTORE
    const size_t rowb = Mp * 2 * (size_t)C * 4;
    const unsigned col = (unsigned)(((size_t)pend_m * 2 * C + pend_grp * 2 * kLfCh) * 4) + (unsigned)((odd ? 64 : 0) + 16 * col_cq);
    const unsigned vb1 = (unsigned)((size_t)(kFftN * kFftInner + 2 * col_ci) * rowb) + col;
    const unsigned vs = (unsigned)(8 * rowb);
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      const bool live = col_kxi < 2 && 4 * g + col_ci < kFftH;
      const unsigned v1 = live ? vb1 + (unsigned)g * vs : 0xfffffff0u;
      store_pair(&ore[4 * g], &oim[4 * g], v1, live ? v1 + (unsigned)rowb : 0xfffffff0u);
    }
#endif
  };

  unsigned cur_grp = 0xffffffffu;
  unsigned v = blockIdx.x;
  if (v < nwork) prefetch(lf_item(v, ngrp, TY, TX), 0, true);
  for (; v < nwork; v += nblk) {
    const LfItem it = lf_item(v, ngrp, TY, TX);
    const unsigned vn = v + nblk;
    const LfItem nx = lf_item(vn < nwork ? vn : v, ngrp, TY, TX);
    auto subphase = [&](auto SUB) {
      constexpr int sub = decltype(SUB)::value;
      stage();
      __syncthreads();
      LF_CLOCK(0);
      if (sub < 3) prefetch(it, sub + 1, true);
      else prefetch(nx, 0, vn < nwork);
      LF_CLOCK(1);
      if constexpr (ROLE == kConv) {
        if (sub == 0 && it.grp != cur_grp) {     // (one block per CU and a group count that divides the grid: a block stays on its group)
          setup(it);
          cur_grp = it.grp;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) conv_row(it, r, 12 * sub + 3 * ck + r);
      } else if constexpr (ROLE == kRows) {
        if (sub > 0) {               // the 12 rows of the previous sub-phase: passes 0, 1 on wave 4, pass 2 on wave 5
          const int y0 = 12 * (sub > 0 ? sub - 1 : 0);
          if (wave == 4) {
            row_pass(y0);
            row_pass(y0 + 4);
          } else {
            row_pass(y0 + 8);
          }
        } else if (wave == 4) {
          edge_work();
        }
      } else {
        col_work(SUB);
      }
      LF_CLOCK(2);
      __syncthreads();
      LF_CLOCK(3);
    };
    subphase(std::integral_constant<int, 0>());
    subphase(std::integral_constant<int, 1>());
    subphase(std::integral_constant<int, 2>());
    subphase(std::integral_constant<int, 3>());
    // ---- the tail nobody hides: the row transforms of rows 36..47 (waves 4, 5 and 0), then the column read
    if constexpr (ROLE == kRows) row_pass(wave == 4 ? 36 : 40);
    if constexpr (ROLE == kConv) {
      if (wave == 0) row_pass(44);
    }
    LF_CLOCK(4);
    __syncthreads();
    LF_CLOCK(5);
    if constexpr (ROLE == kCols) {
#pragma unroll
      for (int y = 0; y < kFftN; ++y) {
        cre[y] = colp[y * kLfRowPitch];
        cim[y] = colp[y * kLfRowPitch + 24 * 4];
      }
    } else if constexpr (ROLE == kRows) {
      if (wave == 4) {
#pragma unroll
        for (int y = 0; y < kFftN; ++y) cre[y] = edgep[y * kLfRowPitch];
      }
    }
    pend_m = it.m;
    pend_grp = it.grp;
    pending = true;
    LF_CLOCK(6);
    __syncthreads();
    LF_CLOCK(7);
  }
  if constexpr (ROLE == kCols) {
   if (pending) {
    col_work(std::integral_constant<int, 0>());
    col_work(std::integral_constant<int, 1>());
    col_work(std::integral_constant<int, 2>());
    col_work(std::integral_constant<int, 3>());
   }
  }
  if constexpr (ROLE == kRows) {
    if (wave == 4) edge_work();
  }
  LF_CLOCK_END();
}

__global__ __launch_bounds__(kLfThreads) void lift5_fft48_fused_kernel(const float* __restrict__ x, const float* __restrict__ bank,
                                                                        const float* __restrict__ bias, int relu, float* __restrict__ V,
                                                                        int H0, int W0, int C, int TY, int TX, size_t Mp, unsigned nwork,
                                                                        unsigned v_bytes, unsigned x_bytes, unsigned bank_bytes) {
  if (threadIdx.x < 4 * 64) lift5_fft48_body<kConv>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes);
  else if (threadIdx.x < 6 * 64) lift5_fft48_body<kRows>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes);
  else lift5_fft48_body<kCols>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes);
}

}  // namespace

extern "C" {

#ifdef EQA_LF_CLOCK
int eqa_debug_lf_clock(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lf_clock), sizeof(g_lf_clock)) == hipSuccess ? 0 : -1; }
#endif

int eqa_lift5_fft48k5_input_supported(int Cin, int KH, int KW, int Cout) {
  return (Cin == 3 && KH == 5 && KW == 5 && Cout > 0 && Cout % kLfCh == 0) ? 1 : 0;
}

int eqa_lift5_fft48k5_input(const float* x, const float* bank, const float* bias, int relu, float* V, int nimg, int H0, int W0, int Cout,
                            void* stream) {
  if (!x || !bank || !V || nimg < 0 || H0 < 5 || W0 < 5 || Cout <= 0) return EQA_ERR_INVALID_ARG;
  if (Cout % kLfCh != 0) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int H1 = H0 - 4, W1 = W0 - 4;
  const int TY = (int)eqa_fft48k5_tiles(H1), TX = (int)eqa_fft48k5_tiles(W1);
  if (TY <= 0 || TX <= 0) return EQA_ERR_UNSUPPORTED;
  const size_t M = (size_t)nimg * TY * TX;
  const size_t nwork = M * (Cout / kLfCh);
  const size_t vb = (size_t)kFftF * fft_pitch(M) * 2 * Cout * 4;
  const size_t xb = (size_t)nimg * H0 * W0 * 3 * 4, bb = (size_t)Cout * 75 * 4;
  if (nwork > 0x7fffffffULL || vb > 0xffffff00ULL || xb > 0xffffff00ULL || bb > 0xffffff00ULL) return EQA_ERR_UNSUPPORTED;
  if (!allow_dynamic_lds((const void*)lift5_fft48_fused_kernel, kLfLdsFloats * 4)) return EQA_ERR_UNSUPPORTED;
  // persistent: one block per CU; a multiple of the group count keeps a block on ONE channel group (its weights stay put in L1)
  const unsigned ngrp = (unsigned)(Cout / kLfCh);
  unsigned nblk = 256;
  if (ngrp <= 256) nblk = (256 / ngrp) * ngrp;
  if ((size_t)nblk > nwork) nblk = (unsigned)nwork;
  hipLaunchKernelGGL(lift5_fft48_fused_kernel, dim3(nblk), dim3(kLfThreads), kLfLdsFloats * sizeof(float), (hipStream_t)stream, x, bank, bias,
                     relu, V, H0, W0, Cout, TY, TX, fft_pitch(M), (unsigned)nwork, (unsigned)vb, (unsigned)xb, (unsigned)bb);
  return launch_status();
}

}  // extern "C"
