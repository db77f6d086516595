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
//             the C operand of the first instruction, ReLU, zero outside the map; three sub-phases of 16 tile rows each, whose 20 input
//             rows (12 KB) are staged in LDS (prefetched into registers one sub-phase ahead); a tile row = three 16-pixel tiles,
//             written into the tile buffer with one 16-byte LDS store per pixel and channel quad
//   rows      thread (row, channel): the 48-point real transform of its row, IN PLACE: 48 reals in, 23 complex + 2 real bins out
//             (the DC and Nyquist bins of a real row are real), so the tile buffer (48 x 48 x 16 floats = 147 KB of the CU's 160)
//             also is the spectrum buffer
//   columns   thread (kx, channel), 24 x 16 of them: the two real columns kx = 0 / 24 ride in ONE complex transform (z = c0 + i c24,
//             separated afterwards -- as fft48_fwd_pipe_kernel does), 48-point complex transform, spectra out in the layout of
//             eqa_fft48k5_input ([Re x 16 | Im x 16] per channel group, frequency-major, odd tile pitch) as WHOLE 128-byte lines
//
// Blocks are persistent (one per CU, 12 waves) and software-pipelined over their items; the waves have roles (below).
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
#define LF_CLOCK_BEGIN() const bool lfc_on = blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 1 || wave == 8); \
  unsigned long long lfc_t = __builtin_readcyclecounter(); unsigned long long lfc_s[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define LF_CLOCK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); lfc_s[i] += n_ - lfc_t; lfc_t = n_; } while (0)
#define LF_CLOCK_END() do { if (lfc_on) for (int i_ = 0; i_ < 8; ++i_) g_lf_clock[8 * (ROLE == kConv ? 0 : (wave == 1 ? 1 : 2)) + i_] = lfc_s[i_]; } while (0)
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
constexpr int kLfSubRows = 16;                                  // tile rows per sub-phase (three sub-phases per item)
constexpr int kLfPatchRows = kLfSubRows + 4, kLfPatchPitch = 156;   // 16 output rows need 20 input rows of 52 pixels x 3 channels
constexpr int kLfPatchFloats = kLfPatchRows * kLfPatchPitch;    // 3,120 floats
constexpr int kLfLdsFloats = kLfTileFloats + kLfPatchFloats;    // 163,008 bytes of the CU's 163,840
constexpr int kLfStagers = 6 * 64;               // the column waves stage the patch (beside their stores they have registers to spare)
constexpr int kLfPre = (kLfPatchFloats + kLfStagers - 1) / kLfStagers;   // staged floats per staging thread and sub-phase (9)

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
// ROLE (a block's 12 waves; one instantiation each, so that a role's registers are live in its own code only; all pass the same
// eight barriers per item in the same order):
//   kCols  waves 0..5: in the three sub-phases the column transform of the PREVIOUS item and its stores, a third per
//          sub-phase (the store path of a CU moves ~10 bytes per clock: an item's 147 KB take ~15 k cycles, hidden behind the
//          convolution of this item), and the staging of the input patches; wave 0 also carries the packed edge columns
//   kConv  waves 6..11: the convolution, 16 tile rows per sub-phase.  Waves w and w + 4 share a SIMD: 6 | 10 and 7 | 11 take 3 + 2 rows,
//          8 and 9 (alone on their SIMDs) three each
//   tail   all twelve waves: the row transforms (four consecutive rows x 16 channels per wave), then the column read (waves 0..5)
enum { kConv = 0, kCols = 1, kRowp = 2 };   // kRowp (FORM 2 only): waves 6, 7, 11 -- staging and row transforms, no convolution

template <class F, int... Is>
__device__ __forceinline__ void lf_for_const(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>()), ...);
}

// PIECES (round 6, second form): the convolution on the bf16 matrix cores.  The fp32 matrix instruction runs on the vector ALU's
// datapath (profiles/r06/mfma_valu_overlap.txt): the transforms' vector work queues behind it at a third of its rate.  Here every
// fp32 pixel and weight is split EXACTLY into three bf16 pieces (8 + 8 + 8 mantissa bits) and a product is six piece products on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation (the three of relative size <= 2^-24 left out: the contract of eqa_fft48k5_cgemm3m_bf16x3).
//   K layout: a pixel = 4 bf16 (3 channels + 0), a chunk = 8 K-elements = 2 pixels (kx = 2 p, 2 p + 1; kx = 5: weight 0), a filter row
//   = 3 chunks, 15 chunks + 1 spare = 16 = 4 matrix instructions of K = 32 per piece product: 24 instructions per 16-pixel tile
//   (the fp32 form: 19 of twice the duration).
//   LDS: the patch as three piece planes of 8 rows x 53 pixels x 8 bytes (a ring over the input rows: rows 4 s .. 4 s + 7 serve the
//   sub-phase of tile rows 4 s .. 4 s + 3; twelve sub-phases per item), 10 KB like the fp32 patch of 20 rows.
constexpr int kLpRowB = 53 * 8;                  // bytes per patch row and plane (52 pixels + 1 finite pad pixel: the kx = 5 slot of the last tile)
constexpr int kLpPlaneB = 8 * kLpRowB;           // 3,392 bytes
constexpr int kLpPatchB = 3 * kLpPlaneB;         // 10,176 bytes
constexpr int kLpLdsBytes = kLfTileFloats * 4 + kLpPatchB;   // 160,704 bytes
// FORM 2 (round 6, third form): the convolution on TWO fp16 pieces per fp32 value, three exact products (h1 k1 + h1 k2 + h2 k1 on
// v_mfma_f32_16x16x32_f16, fp32 accumulate: the contract of eqa_fft48k5_cgemm3m_f16x2 -- within one fp32 ulp per operand; pixels
// scaled by the power of two that takes a caller-supplied bound of |x| to 2^14, weights when they are split, the accumulators scaled
// back in the epilogue).  Two planes of 8 bytes per pixel are the fp32 patch's bytes + a third, so a sub-phase is TWELVE tile rows (a
// ring of 16 patch rows; four sub-phases per item, where the three bf16 planes allowed four rows and needed twelve), and twelve
// matrix instructions of 16 cycles per 16-pixel tile replace 19 of 32 on a datapath the transforms do not share: the convolution
// waves are done in a third of a sub-phase, so THEY stage the patches (the column waves' loads queued behind their own stores:
// 8.6 k cycles per item) and transform the previous sub-phase's rows; only twelve rows are left for the tail.
//   K layout: a chunk = the pixel pair (j - 1 + 2 p, j + 2 p) of output pixel j, i.e. kx = 2 p - 1 (p = 0: weight 0), 2 p; the
//   pair of p = 0, j = 0 starts 8 bytes in front of the row: the previous ring row's last pixel, and in front of the patch the tile
//   buffer's last (never written, zeroed once) pad floats -- which is how the patch fits the CU's LDS to the byte.
constexpr int kLhRowB = 52 * 8;                  // bytes per patch row and plane
constexpr int kLhRing = 16;                      // ring rows (12 tile rows + 4)
constexpr int kLhPlaneB = kLhRing * kLhRowB;     // 6,656 bytes
constexpr int kLhPatchB = 2 * kLhPlaneB;         // 13,312 bytes
constexpr int kLhLdsBytes = kLfTileFloats * 4 + kLhPatchB;   // 163,840 bytes: all of the CU's LDS
constexpr int kLhSubRows = 12;
constexpr int kLhPre = 3;                        // pixels per staging thread and unit (16 rows x 52 pixels over 384 threads)
typedef _Float16 lf_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lf_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 lf_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lf_bf16x4 __attribute__((ext_vector_type(4)));
// x = p1 + p2 + p3 exactly (round to nearest each time: the remainder of an 8-bit piece fits the next)
__device__ __forceinline__ void lf_split3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - (float)p1;
  p2 = (__bf16)r1;
  p3 = (__bf16)(r1 - (float)p2);
}

template <int ROLE, int FORM>
__device__ __forceinline__ void lift5_fft48_body(const float* __restrict__ x, const float* __restrict__ bank, const float* __restrict__ bias,
                                                 int relu, float* __restrict__ V, int H0, int W0, int C, int TY, int TX, size_t Mp,
                                                 unsigned nwork, unsigned v_bytes, unsigned x_bytes, unsigned bank_bytes,
                                                 float* __restrict__ dcmax, const float* __restrict__ xbound, int nxbound, float w_scale) {
  constexpr bool COLS = ROLE == kCols;
  constexpr bool PIECES = FORM == 3;                 // three bf16 pieces, six products
  constexpr bool H2 = FORM == 2;                     // two fp16 pieces, three products
  constexpr int NSUB = PIECES ? 12 : (H2 ? 4 : 3);   // sub-phases per item
  constexpr int SUBROWS = kFftN / NSUB;              // tile rows per sub-phase (4 | 12 | 16)
  constexpr int NGRP = 12 / NSUB;                    // frequency groups a column wave stores per sub-phase (1 | 3 | 4)
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
                      // [4] tail row pass, [5] barrier 3, [6] column read, [7] barrier 4

  // ---- the input patch of a sub-phase: 20 input rows x 52 pixels x 3 channels, staged by the threads of the column waves
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, x_bytes, 0x00020000);
  float pre[COLS ? kLfPre : 1];        // the next sub-phase's share (rows / columns outside the image: the range check's 0)
  auto prefetch = [&](const LfItem& it, int sub, bool live) {
    if constexpr (COLS) {
      int tq = tid;     // opaque: the index arithmetic below is redone per call (hoisted out of the item loop it went to scratch)
      asm volatile("" : "+v"(tq));
#pragma unroll
      for (int i = 0; i < kLfPre; ++i) {
        const int idx = tq + kLfStagers * i;
        const int pr = idx / kLfPatchPitch, pc = idx - pr * kLfPatchPitch;
        const int gy = it.gy0 + kLfSubRows * sub + pr, gxc = it.gx0 * 3 + pc;
        const bool ok = live && idx < kLfPatchFloats && gy < H0 && gxc < W0 * 3;
        const unsigned off = ok ? (unsigned)(((it.img * H0 + gy) * (size_t)(W0 * 3) + gxc) * 4) : 0xfffffff0u;
        pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, 0, 0));
      }
    }
  };
  auto stage = [&]() {
    if constexpr (COLS) {
#pragma unroll
      for (int i = 0; i < kLfPre; ++i) {
        const int idx = tid + kLfStagers * i;
        if (idx < kLfPatchFloats) patch[idx] = pre[i];
      }
    }
  };

  // PIECES: input rows in groups of four (group g = patch rows 4 g .. 4 g + 3, 13 groups per item, ring slot g % 2): thread t < 208 of
  // the column waves owns pixel (row t / 52, x t % 52) of a group -- three dword loads one sub-phase ahead, three 8-byte LDS stores
  // (one per piece plane: c0 c1 c2 0).  `which`: the register set (the last sub-phase of an item fetches two groups of the next).
  float preg[COLS && PIECES ? 2 : 1][3];
  auto prefetch_grp = [&](const LfItem& it, int g, int which, bool live) {
    if constexpr (COLS && PIECES) {
      int tq = tid;
      asm volatile("" : "+v"(tq));
      const int pr = tq / 52, px = tq - pr * 52;
      const int gy = it.gy0 + 4 * g + pr, gx = it.gx0 + px;
      const bool ok = live && tq < 208 && gy < H0 && gx < W0;
      const unsigned off = ok ? (unsigned)((((it.img * H0 + gy) * (size_t)W0 + gx) * 3) * 4) : 0xfffffff0u;
#pragma unroll
      for (int c = 0; c < 3; ++c) preg[which][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, c * 4, 0));
    }
  };
  auto stage_grp = [&](int g, int which) {
    if constexpr (COLS && PIECES) {
      if (tid < 208) {
        const int pr = tid / 52, px = tid - pr * 52;
        __bf16 pc[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) lf_split3(preg[which][c], pc[0][c], pc[1][c], pc[2][c]);
        char* dst = reinterpret_cast<char*>(lds) + patch_b + ((4 * g + pr) & 7) * kLpRowB + px * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const lf_bf16x4 v = {pc[pl][0], pc[pl][1], pc[pl][2], (__bf16)0.0f};
          *reinterpret_cast<lf_bf16x4*>(dst + pl * kLpPlaneB) = v;
        }
      }
    }
  };

  // FORM 2: a staging unit = patch rows [row0, row0 + nrows) of an item (16 rows in front of its first sub-phase, 12 afterwards), by
  // the 384 threads of the CONVOLUTION waves: thread t owns pixels t, t + 384, t + 768 of the unit -- three dword loads each a
  // sub-phase ahead; behind the sub-phase's second barrier the scaled values are split and written, 8 bytes per plane.
  float preh[(!COLS && H2) ? kLhPre : 1][3];
  float x_scale = 1.0f, out_scale = 1.0f;
  if constexpr (H2) {
    float mx = 0.0f;
    for (int t = lane; t < nxbound; t += 64) mx = fmaxf(mx, xbound[t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    int ex = 0;
    if (mx > 0.0f && mx < 3.0e38f) (void)frexpf(mx, &ex);      // mx <= 2^ex
    ex = __builtin_amdgcn_readfirstlane(min(max(14 - ex, -100), 100));
    x_scale = ldexpf(1.0f, ex);
    out_scale = ldexpf(1.0f, -ex) / w_scale;
  }
  int h_pr[(!COLS && H2) ? kLhPre : 1], h_px[(!COLS && H2) ? kLhPre : 1];     // the staging thread's three pixels of a unit
  if constexpr (!COLS && H2) {
#pragma unroll
    for (int i = 0; i < kLhPre; ++i) {
      const int idx = tid - 6 * 64 + kLfStagers * i;
      h_pr[i] = idx / 52;
      h_px[i] = idx - h_pr[i] * 52;
    }
  }
  auto prefetch_h = [&](const LfItem& it, int row0, int nrows, bool live) {
    if constexpr (!COLS && H2) {
#ifdef EQA_LF_H2_NOPREFETCH    // ablation: no patch loads (the ring keeps the first item's rows)
      if (row0 >= 0) return;
#endif
#pragma unroll
      for (int i = 0; i < kLhPre; ++i) {
        const int pr = h_pr[i], px = h_px[i];
        const int gy = it.gy0 + row0 + pr, gx = it.gx0 + px;
        const bool ok = live && pr < nrows && gy < H0 && gx < W0;
        const unsigned off = ok ? (unsigned)((((it.img * H0 + gy) * (size_t)W0 + gx) * 3) * 4) : 0xfffffff0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) preh[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, off, c * 4, 0));
      }
    }
  };
  auto stage_h = [&](int row0, int nrows) {
    if constexpr (!COLS && H2) {
#pragma unroll
      for (int i = 0; i < kLhPre; ++i) {
        // (opaque here: the scaling below is plain register arithmetic, and hoisted above the barrier it put the wait for the loads
        // -- 2-4 us under this kernel's traffic -- in front of the convolution: 32 k cycles per item)
        asm volatile("" : "+v"(preh[i][0]), "+v"(preh[i][1]), "+v"(preh[i][2]));
        if (h_pr[i] < nrows) {
          const int pr = h_pr[i], px = h_px[i];
          _Float16 hi[3], lo[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float v = preh[i][c] * x_scale;
            hi[c] = (_Float16)v;
            lo[c] = (_Float16)(v - (float)hi[c]);
          }
          char* dst = reinterpret_cast<char*>(lds) + patch_b + ((row0 + pr) & (kLhRing - 1)) * kLhRowB + px * 8;
          const lf_f16x4 vh = {hi[0], hi[1], hi[2], (_Float16)0.0f}, vl = {lo[0], lo[1], lo[2], (_Float16)0.0f};
          *reinterpret_cast<lf_f16x4*>(dst) = vh;
          *reinterpret_cast<lf_f16x4*>(dst + kLhPlaneB) = vl;
        }
      }
    }
  };

  // ---- kConv: the wave's rows of a sub-phase: first row r0, count nr
  // (measured, cycles per row: a wave alone on its SIMD 3.3 k -- its operand reads and its epilogue are exposed --, two waves
  // sharing a SIMD cover each other: ~2 k per row and SIMD.  Hence 3 + 3 rows for the lone waves 8, 9 and 3 + 2 per shared SIMD.)
  const int nr = (wave == 10 || wave == 11) ? 2 : 3;
  const int r0 = wave == 8 ? 0 : (wave == 9 ? 3 : (wave == 6 ? 6 : (wave == 10 ? 9 : (wave == 7 ? 11 : 14))));
  // the lane's LDS byte offset of step t's operand (patch row r0 + ky, element e of pixel j) and its weight (channel i = lane % 16,
  // k-slot q of step t), the bias of the lane's four output channels
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
      a_off[t] = patch_b + (unsigned)(((r0 + ky) * kLfPatchPitch + j * 3 + e) * 4);
      // (the spare slot's weight: an offset beyond the bank, i.e. the range check's 0)
      wreg[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, spare ? 0xfffffff0u : wb + (unsigned)((ky * 15 + e) * 4), 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = bias ? bias[it.grp * kLfCh + 4 * q + r] : 0.0f;
    asm volatile("" : "+v"(bias4));      // opaque: otherwise re-loaded from memory in front of every sub-phase's first matrix instruction
  };
  // The wave's tile rows of a sub-phase (a row = 3 tiles of 16 pixels: patch rows r0 + r .. + 4 -> tile row y0 + r).  The 57 matrix
  // instructions of a row run STEP by step (the three tiles' instructions of a step are independent; the scheduler barriers keep
  // that order) and the operand reads are interleaved with them by hand, five steps ahead: a wave can have 15 LDS reads in flight
  // (lgkmcnt), so all 57 requested up front stood in front of the first matrix instruction for ~500 cycles per row.  The first five
  // steps of the NEXT row are requested under the last five steps of this one.  All reads use immediate offsets off the per-lane
  // addresses of `setup`.
#ifndef EQA_LF_AHEAD
#define EQA_LF_AHEAD 5
#endif
  constexpr int kAhead = EQA_LF_AHEAD;
  auto conv_rows = [&](const LfItem& it, int y0) {
    float b[3][kLfSteps], bq[2][3][kAhead];
    auto lds_at = [&](int t, int drow, int tx3) {
#ifdef EQA_LF_NOLDSREAD   // ablation: the matrix stream without its operand reads
      return __builtin_bit_cast(float, a_off[t] + (unsigned)(drow + tx3));
#else
      return reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + a_off[t])[drow * kLfPatchPitch + tx3 * 48];
#endif
    };
#pragma unroll
    for (int t = 0; t < kAhead; ++t)
#pragma unroll
      for (int tx3 = 0; tx3 < 3; ++tx3) bq[0][tx3][t] = lds_at(t, 0, tx3);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      if (r < nr) {       // wave-uniform
        const int par = r & 1;
        const bool row_ok = it.gy0 + y0 + r < H1;
        f32x4 acc[3] = {bias4, bias4, bias4};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kLfSteps; ++t) {
#pragma unroll
          for (int tx3 = 0; tx3 < 3; ++tx3)
#ifdef EQA_LF_NOMFMA      // ablation: the operand reads and the epilogue without the matrix instructions
            acc[tx3][t & 3] += wreg[t] * (t < kAhead ? bq[par][tx3][t] : b[tx3][t]);
#else
            acc[tx3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[t], t < kAhead ? bq[par][tx3][t] : b[tx3][t], acc[tx3], 0, 0, 0);
#endif
          __builtin_amdgcn_sched_barrier(0);
          if (t + kAhead < kLfSteps) {
#pragma unroll
            for (int tx3 = 0; tx3 < 3; ++tx3) b[tx3][t + kAhead] = lds_at(t + kAhead, r, tx3);
          } else if (r + 1 < nr) {
#pragma unroll
            for (int tx3 = 0; tx3 < 3; ++tx3) bq[par ^ 1][tx3][t + kAhead - kLfSteps] = lds_at(t + kAhead - kLfSteps, r + 1, tx3);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tx3 = 0; tx3 < 3; ++tx3) {
          const bool ok = row_ok && it.gx0 + tx3 * 16 + j < W1;
          f32x4 v = acc[tx3];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float u = relu ? fmaxf(v[k], 0.0f) : v[k];
            v[k] = ok ? u : 0.0f;
          }
          *reinterpret_cast<f32x4*>(tile + (y0 + r) * kLfRowPitch + q * kLfQuadPitch + (tx3 * 16 + j) * 4) = v;
        }
      }
    }
  };

  // ---- PIECES convolution.  Lane (i = lane % 16, q = lane / 16): chunk c = 4 s + q of K-step s = (filter row ky = c / 3, pixel pair
  // c % 3); A = weights of channel i (three pieces x four K-steps x 16 bytes, resident), B = the pixel pair (j + 2 pair, + 1) of pixel j.
  lf_bf16x8 wq[PIECES ? 4 : 1][PIECES ? 3 : 1];
  unsigned c_off[PIECES ? 4 : 1];      // per K-step: byte offset of the lane's chunk in patch row 0, tile column 0, plane 0
  int c_ky[PIECES ? 4 : 1];
  auto setup_p = [&](const LfItem& it) {
    if constexpr (PIECES) {
      const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bank), 0, bank_bytes, 0x00020000);
#pragma unroll
      for (int sstep = 0; sstep < 4; ++sstep) {
        const int c = 4 * sstep + q;
        c_ky[sstep] = min(c / 3, 4);
        c_off[sstep] = (unsigned)((j + 2 * (c % 3)) * 8);
#pragma unroll
        for (int wp = 0; wp < 3; ++wp) {
          // wpieces: (Cout, 3 pieces, 16 chunks, 8) bf16
          const unsigned off = (unsigned)((((it.grp * kLfCh + j) * 3 + wp) * 16 + c) * 16);
          wq[sstep][wp] = __builtin_bit_cast(lf_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(br, off, 0, 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = bias ? bias[it.grp * kLfCh + 4 * q + r] : 0.0f;
      asm volatile("" : "+v"(bias4));
    }
  };
  // nt (1..3, wave-uniform) tiles tx0 .. tx0 + nt - 1 of tile row y (row of the ITEM: patch rows y .. y + 4 sit in ring slots (y + ky) % 8)
  auto conv_tiles_p = [&](const LfItem& it, int y, int tx0, auto NT) {
    constexpr int nt = decltype(NT)::value;     // compile-time: a wave-uniform `t < nt` in front of every matrix instruction became a branch each
    if constexpr (PIECES) {
      const bool row_ok = it.gy0 + y < H1;
      unsigned addr[4];
#pragma unroll
      for (int sstep = 0; sstep < 4; ++sstep) addr[sstep] = patch_b + (unsigned)(((y + c_ky[sstep]) & 7) * kLpRowB) + c_off[sstep] + (unsigned)(tx0 * 16 * 8);
      f32x4 acc[3] = {bias4, bias4, bias4};
      lf_bf16x8 bp[2][3][3];
      auto read_step = [&](int sstep, int buf) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (t < nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#ifdef EQA_LF_NOLDSREAD
            {
              const f32x4 cv = {1.0f + (float)(addr[sstep] & 7), 0.5f + t, 0.25f + pl, 2.0f};
              bp[buf][t][pl] = __builtin_bit_cast(lf_bf16x8, cv);
            }
#else
            {
              // two 8-byte reads (ds_read2_b64): the pixel pair sits at an 8-byte boundary, and a 16-byte LDS read off a 16-byte
              // boundary takes ~200 cycles on this chip (the first build: 122 k instead of 33 k cycles per item)
              typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
              const unsigned long long* pp = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(lds) + addr[sstep] + t * 128 + pl * kLpPlaneB);
              const u64x2_t two = {pp[0], pp[1]};
              bp[buf][t][pl] = __builtin_bit_cast(lf_bf16x8, two);
            }
#endif
      };
      read_step(0, 0);
#pragma unroll
      for (int sstep = 0; sstep < 4; ++sstep) {
        const int buf = sstep & 1;
        if (sstep + 1 < 4) read_step(sstep + 1, buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // six piece products, small terms first: (w3, p1) (w1, p3) (w2, p2) (w2, p1) (w1, p2) (w1, p1)
        constexpr int kW[6] = {2, 0, 1, 1, 0, 0}, kP[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
#pragma unroll
          for (int t = 0; t < 3; ++t)
            if (t < nt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[sstep][kW[pr]], bp[buf][t][kP[pr]], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t < nt) {
          const int tx3 = tx0 + t;
          const bool ok = row_ok && it.gx0 + tx3 * 16 + j < W1;
          f32x4 v = acc[t];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float u = relu ? fmaxf(v[k], 0.0f) : v[k];
            v[k] = ok ? u : 0.0f;
          }
          *reinterpret_cast<f32x4*>(tile + y * kLfRowPitch + q * kLfQuadPitch + (tx3 * 16 + j) * 4) = v;
        }
      }
    }
  };

  // ---- FORM 2 convolution.  K-step s = filter row ky = s (five steps of K = 32); lane (i = lane % 16, q = lane / 16) holds chunk q of a
  // step: the pixel pair p = q (pixels j - 1 + 2 p, j + 2 p: kx = 2 p - 1, 2 p) for q < 3, a chunk of zero weights for q = 3 (its
  // lanes read pair 2 again: the same addresses as their neighbours, a broadcast).  So the B fragment of (patch row R, tile column t)
  // does not depend on ky: a wave works DOWN one tile column, holds a window of five patch rows in registers and reads ONE new row
  // (two 16-byte fragments) per tile row -- 2 KB of LDS reads per tile where the row-major order with its 4 K-steps per tile read 8
  // (the first build of this form: 33 k cycles per item in the convolution waves, on LDS bandwidth).  15 matrix instructions per
  // tile in ONE accumulator chain (per filter row the small products first).
  lf_f16x8 wh[H2 ? 5 : 1][H2 ? 2 : 1];
  int h_off = 0;
  auto setup_h = [&](const LfItem& it) {
    if constexpr (H2) {
      const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bank), 0, bank_bytes, 0x00020000);
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int wp = 0; wp < 2; ++wp) {
          // wpieces: (Cout, 2 pieces, 5 filter rows, 4 chunks, 8) fp16
          const unsigned off = (unsigned)(((((it.grp * kLfCh + j) * 2 + wp) * 5 + ky) * 4 + q) * 16);
          wh[ky][wp] = __builtin_bit_cast(lf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(br, off, 0, 0));
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = bias ? bias[it.grp * kLfCh + 4 * q + r] : 0.0f;
      asm volatile("" : "+v"(bias4));
    }
  };
  // tile column t (16 pixels), tile rows y0 .. y0 + NROWS - 1 of the item (patch row R sits in ring slot R % 16)
  auto conv_col_h = [&](const LfItem& it, int y0, int t, auto NR) {
    if constexpr (H2) {
      constexpr int NROWS = decltype(NR)::value;
#ifdef EQA_LF_H2_NOCONV    // ablation: the convolution waves without their convolution (the tile keeps what it held)
      return;
#endif
      const int base = (int)patch_b + (j - 1 + 2 * min(q, 2)) * 8 + t * 128;
      lf_f16x8 win[7][2];                     // patch rows y0 + k in slot k % 7: five in use, two arriving (an LDS read comes back
                                              // after 400-600 cycles while the row transforms and column reads queue beside it)
      auto read_row = [&](int k) {
        const int a = base + ((y0 + k) & (kLhRing - 1)) * kLhRowB;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          // two 8-byte reads: the pixel pair sits at an 8-byte boundary (a 16-byte LDS read off a 16-byte boundary: ~200 cycles)
          typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
          const unsigned long long* pp = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(lds) + a + pl * kLhPlaneB);
          const u64x2_t two = {pp[0], pp[1]};
          win[k % 7][pl] = __builtin_bit_cast(lf_f16x8, two);
        }
      };
#pragma unroll
      for (int k = 0; k < 6; ++k) read_row(k);
      const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
      const bool col_ok = it.gx0 + t * 16 + j < W1;
      // accumulators of two rows: a row's epilogue (scale, bias, relu, LDS write) sits BEHIND the next row's matrix instructions in
      // program order, so that it issues under them -- directly behind its own row it waited out the pipe's latency first, and the
      // wave issues in order: 39 cycles per matrix instruction instead of 16
      // ONE accumulator chain per row (tools/micro/mfma16_chains.hip: dependent v_mfma_f32_16x16x32_f16 issue every 21.7 counter cycles,
      // two chains 22.8, THREE 28.7, four or six 22.3 -- the three chains "one per piece product" of the first builds were the worst
      // choice); two rows' accumulators, so that a row's epilogue runs a row later, in pieces between the next row's instructions.
      // Per filter row the small products come first: (w lo, p hi), (w hi, p lo), then (w hi, p hi).
      f32x4 acc[2];
      f32x4 ev = zero;
      // piece m of the epilogue of row rp: scale, bias, relu, mask, write
      auto epi = [&](int m, int rp) {
        const int y = y0 + rp;
        if (m == 0) ev = acc[rp & 1] * out_scale + bias4;
        else if (m == 1) { ev[0] = relu ? fmaxf(ev[0], 0.0f) : ev[0]; ev[1] = relu ? fmaxf(ev[1], 0.0f) : ev[1]; }
        else if (m == 2) { ev[2] = relu ? fmaxf(ev[2], 0.0f) : ev[2]; ev[3] = relu ? fmaxf(ev[3], 0.0f) : ev[3]; }
        else if (m == 3) {
          const bool ok = col_ok && it.gy0 + y < H1;
#pragma unroll
          for (int k = 0; k < 4; ++k) ev[k] = ok ? ev[k] : 0.0f;
        } else if (m == 4) {
          *reinterpret_cast<f32x4*>(tile + y * kLfRowPitch + q * kLfQuadPitch + (t * 16 + j) * 4) = ev;
        }
      };
      // The wave issues in order: what stands BETWEEN two matrix instructions is free, what stands behind a block of them is not.
#pragma unroll
      for (int r = 0; r < NROWS; ++r) {
#pragma unroll
        for (int m = 0; m < 15; ++m) {
          const int ky = m / 3, ch = m % 3;
          __builtin_amdgcn_sched_barrier(0);
          acc[r & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[ky][ch == 0 ? 1 : 0], win[(r + ky) % 7][ch == 1 ? 1 : 0], m == 0 ? zero : acc[r & 1], 0, 0, 0);
          if (r > 0 && m >= 2 && m < 7) epi(m - 2, r - 1);
          if (m == 9 && r + 2 < NROWS) read_row(r + 6);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 5; ++m) epi(m, NROWS - 1);
    }
  };

  // ---- row pass of four consecutive tile rows y0 .. y0 + 3: thread (r = lane / 16, c = lane % 16), 48 reals -> 23 complex + 2 real
  // bins IN PLACE (slot kx: Re[kx], kx = 0..23; slot 24: Re[24]; slot 24 + kx: Im[kx]).  Row pitch 784 = 16 (mod 32): the rows of
  // a 32-lane group sit 16 banks apart.
  auto row_pass = [&](int y0) {
#ifdef EQA_LF_H2_NOROWP     // ablation: no row transforms (wrong spectra)
    if (H2) return;
#endif
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

  // ---- columns.  Lane l = (channel-in-quad ci = l / 16, task kxi = (l % 16) / 4, channel quad cq = l % 4): the four channels of a quad
  // sit in the four 16-lane rows at the same position, so that the 4 x 4 (frequency, channel) blocks in front of the stores
  // transpose with v_permlane16_swap / v_permlane32_swap; the tasks kxi = 2 a and 2 a + 1 are lanes l and l ^ 4, which trade their
  // imaginary / real quads (ds_swizzle) so that EIGHT lanes store one whole 128-byte line [Re x 16 | Im x 16] -- written as two
  // 64-byte halves by two instructions the store path ran at 3.0 TB/s, as whole lines it runs at 5.2 (tools/micro/store_pattern.hip).
  // Task kc of {b, b + 4, b + 1, b + 5}, b = 8 (w / 2) + 2 (w % 2); task 0 (wave 0) is the packed pair (kx = 0 in the real part,
  // kx = 24 in the imaginary part), task kc > 0 is kx = kc.
  const int cw = COLS ? wave : 0;
  const int col_ci = lane >> 4, col_kxi = (lane >> 2) & 3, col_cq = lane & 3;
  const int kc_of0 = 8 * (cw >> 1) + 2 * (cw & 1) + (col_kxi >> 1);       // the pair's even task (kxi & ~1)
  const int kc = kc_of0 + ((col_kxi & 1) ? 4 : 0);
  const bool odd = (col_kxi & 1) != 0;
  const bool has_packed = COLS && __builtin_amdgcn_readfirstlane((int)(cw == 0)) != 0;   // wave 0 (wave-uniform)
  const bool pair_packed = COLS && kc_of0 == 0;                             // the pair whose even task is the packed one (wave 0, kxi 0 / 1)
  const bool packed = COLS && kc == 0;
  const float* const colp = tile + col_cq * kLfQuadPitch + col_ci + kc * 4;   // Re: slot kc; Im: slot 24 + kc (+ 96 floats)
  const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(V, 0, v_bytes, 0x00020000);
  float cre[kFftN], cim[kFftN];      // column data read from LDS, then (in place) the spectrum waiting to be stored
  unsigned pend_m = 0, pend_grp = 0;
  bool pending = false;
  float dc_seen = 0.0f;              // wave 0: the largest DC bin it stored (see the end of the kernel)
  // 4 x 4 transpose of (register k, lane row i) across the wave's four 16-lane rows: afterwards register k of row i holds what
  // register i held in row k.  (Inline asm with the hazard's two wait states inside the string: chained through the builtins'
  // two-element results, hipcc 7.2 folded the second element into the first -- tools/micro/permlane_swap.hip.)
  auto transpose4 = [&](float& r0_, float& r1_, float& r2_, float& r3_) {
    auto sw16 = [](float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); };
    auto sw32 = [](float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); };
    sw16(r0_, r1_);
    sw16(r2_, r3_);
    sw32(r0_, r2_);
    sw32(r1_, r3_);
  };
  auto swz4 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x101f)); };   // lane l <- lane l ^ 4
  // One frequency group g (ky 4 g .. 4 g + 3; lane row i stores ky = 4 g + i) of a lane pair: store 1 writes the line of the pair's
  // EVEN task (even lanes: its real quad, odd lanes: its imaginary quad, received), store 2 the odd task's (SECOND).  voff1 / voff2:
  // the lane's byte offsets of the two lines (0xfffffff0: masked by the range check).
  auto store_pair = [&](float* re, float* im, unsigned voff1, unsigned voff2, bool second) {
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
    if (second) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o2), vr, voff2, 0, EQA_LF_AUX);
  };
  // The pending item's share of sub-phase SUB: the transform in sub-phase 0, four of its twelve frequency groups in each.
  // Frequency of (kx, ky): interior kx: 23 ky + kx - 1; packed: 1104 + 2 ky (kx = 0), 1105 + 2 ky (kx = 24), ky <= 24 only.
  auto col_work = [&](auto SUB) {
    constexpr int sub = decltype(SUB)::value;
    if (!pending) return;
#ifdef EQA_LF_H2_NOCOLS     // ablation: no column transforms / stores
    if (H2) return;
#endif
    const size_t rowb = Mp * 2 * (size_t)C * 4;                         // bytes per stored frequency
    const unsigned col = (unsigned)(((size_t)pend_m * 2 * C + pend_grp * 2 * kLfCh) * 4) + (unsigned)((odd ? 64 : 0) + 16 * col_cq);
    if (sub == 0) {
      float ore[kFftN], oim[kFftN];
      fft48(cre, cim, ore, oim);
#pragma unroll
      for (int ky = 0; ky < kFftN; ++ky) {
        cre[ky] = ore[ky];
        cim[ky] = oim[ky];
      }
      if (has_packed) {
        // Wave 0.  The packed lanes separate their two real columns: C0 = (Z[k] + conj Z[-k]) / 2 replaces Z[k] in place (k = 0..24);
        // C24 = (Z[k] - conj Z[-k]) / 2i leaves at once (seven line stores: even lanes the real quad, odd lanes the imaginary quad,
        // received) -- kept until its turn in a later sub-phase it would cost every column wave 56 registers.
        float p24re[28], p24im[28];
#pragma unroll
        for (int ky = 0; ky < kFftH; ++ky) {
          const int kn = (kFftN - ky) % kFftN;
          const float a = ore[ky], b = oim[ky], c2 = ore[kn], d = oim[kn];
          p24re[ky] = 0.5f * (b + d);
          p24im[ky] = 0.5f * (c2 - a);
          cre[ky] = packed ? 0.5f * (a + c2) : a;
          cim[ky] = packed ? 0.5f * (b - d) : b;
        }
        dc_seen = fmaxf(dc_seen, packed ? cre[0] : 0.0f);     // the DC bin (kx = 0, ky = 0) of this lane's channel
#pragma unroll
        for (int ky = kFftH; ky < 28; ++ky) p24re[ky] = p24im[ky] = 0.0f;
#ifndef EQA_LF_NOSTORE
        const unsigned vb24 = (unsigned)((size_t)(kFftN * kFftInner + 1 + 2 * col_ci) * rowb) + col;
#pragma unroll
        for (int g = 0; g < 7; ++g) {
          const bool live = pair_packed && 4 * g + col_ci < kFftH;
          store_pair(&p24re[4 * g], &p24im[4 * g], live ? vb24 + (unsigned)g * (unsigned)(8 * rowb) : 0xfffffff0u, 0xfffffff0u, false);
        }
#endif
      }
    }
#ifndef EQA_LF_NOSTORE
    const unsigned f1 = pair_packed ? (unsigned)(kFftN * kFftInner + 2 * col_ci) : (unsigned)(kFftInner * col_ci + kc_of0 - 1);
    const unsigned vb1 = (unsigned)((size_t)f1 * rowb) + col, vs1 = (unsigned)((pair_packed ? 8 : 4 * kFftInner) * rowb);
    const unsigned vb2 = (unsigned)((size_t)(kFftInner * col_ci + kc_of0 + 3) * rowb) + col, vs2 = (unsigned)(4 * kFftInner * rowb);
#pragma unroll
    for (int g = NGRP * sub; g < NGRP * sub + NGRP; ++g) {
      const bool live1 = !pair_packed || 4 * g + col_ci < kFftH;
      store_pair(&cre[4 * g], &cim[4 * g], live1 ? vb1 + (unsigned)g * vs1 : 0xfffffff0u, vb2 + (unsigned)g * vs2, true);
    }
#endif
  };

  // The item loop runs ONE iteration past the block's last item: in it only the column waves work (the last item's transform and
  // stores) and everybody passes the barriers -- a second copy of that code behind the loop, and a copy of the sub-phase per
  // sub-phase in the convolution role, made the kernel 70 KB: more than the 64 KB instruction cache two CUs share, and the matrix
  // instruction streams waited for their own code.
  unsigned cur_grp = 0xffffffffu;
  unsigned v = blockIdx.x;
  if constexpr (PIECES) {
    // the pad pixel of every patch row and plane: finite (its weight is 0), written once
    if (COLS && tid < 24) *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(lds) + patch_b + tid * kLpRowB + 52 * 8) = 0ull;
    const LfItem f = lf_item(v < nwork ? v : 0, ngrp, TY, TX);
    prefetch_grp(f, 0, 0, v < nwork);
    prefetch_grp(f, 1, 1, v < nwork);
    stage_grp(0, 0);
    stage_grp(1, 1);
  } else if constexpr (H2) {
    // the tile buffer's last pad floats: the 8 bytes in front of the patch that the pair (pixel -1, pixel 0) of ring row 0 reads
    if (tid == 0) *reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(lds) + patch_b - 8) = 0ull;
    prefetch_h(lf_item(v < nwork ? v : 0, ngrp, TY, TX), 0, kLhRing, v < nwork);
    stage_h(0, kLhRing);
    __syncthreads();
  } else {
    // the first item's first patch: staged here; afterwards an item's first patch is staged in the TAIL of the item before it, so
    // that the convolution of sub-phase 0 starts as soon as the column waves have read tile rows 0..15 (see the tail)
    if (v < nwork) prefetch(lf_item(v, ngrp, TY, TX), 0, true);
    stage();
    __syncthreads();
  }
  for (;; v += nblk) {
    const bool live = v < nwork;          // block-uniform
    if (!live && !pending) break;
    const LfItem it = lf_item(live ? v : 0, ngrp, TY, TX);
    const unsigned vn = v + nblk;
    const bool next_live = live && vn < nwork;
    const LfItem nx = lf_item(next_live ? vn : 0, ngrp, TY, TX);
    if constexpr (!COLS) {
#pragma unroll 1
      for (int sub = 0; sub < NSUB; ++sub) {
        if (PIECES || sub > 0) __syncthreads();
        LF_CLOCK(0);
        if constexpr (H2) {
          // requested now, staged behind this sub-phase's second barrier: the next sub-phase's twelve new rows, or the next item's first 16
          if (sub < NSUB - 1) prefetch_h(it, kLhSubRows * sub + kLhRing, kLhSubRows, live);
          else prefetch_h(nx, 0, kLhRing, next_live);
          LF_CLOCK(1);
        }
        if (live) {
          if (sub == 0 && it.grp != cur_grp) {     // (one block per CU and a group count that divides the grid: a block stays on its group)
            if constexpr (PIECES) setup_p(it);
            else if constexpr (H2) { if constexpr (ROLE == kConv) setup_h(it); }
            else setup(it);
            cur_grp = it.grp;
          }
          if constexpr (H2) {
            // waves 8, 9, 10: a tile column each of the sub-phase's twelve rows; waves 6, 7, 11: the row transform of four of the PREVIOUS
            // sub-phase's rows each (the tail keeps rows 36..47)
            if constexpr (ROLE == kConv) {
              conv_col_h(it, kLhSubRows * sub, wave - 8, std::integral_constant<int, kLhSubRows>());
            } else {
              if (sub > 0) row_pass(kLhSubRows * (sub - 1) + 4 * (wave == 11 ? 2 : wave - 6));
            }
          } else if constexpr (PIECES) {
            // the sub-phase's 12 tiles, three per SIMD: waves 8, 9 (alone on theirs) a row each, 6 | 10 and 7 | 11 share a row 2 + 1
            const int y = SUBROWS * sub + (wave == 8 ? 0 : (wave == 9 ? 1 : ((wave == 6 || wave == 10) ? 2 : 3)));
            const int tx0 = (wave == 10 || wave == 11) ? 2 : 0;
            if (wave == 8 || wave == 9) conv_tiles_p(it, y, tx0, std::integral_constant<int, 3>());
            else if (wave == 10 || wave == 11) conv_tiles_p(it, y, tx0, std::integral_constant<int, 1>());
            else conv_tiles_p(it, y, tx0, std::integral_constant<int, 2>());
          } else {
            conv_rows(it, kLfSubRows * sub + r0);
            // The waves that are alone on their SIMD finish their three rows ~2.6 k cycles before the shared SIMDs finish their
            // five: in sub-phases 1 and 2 they spend that wait on a row pass of the PREVIOUS sub-phase's rows (four passes, rows
            // 0..15, leave the tail, which then has two passes per SIMD instead of three).
            if (sub > 0 && (wave == 8 || wave == 9)) row_pass(8 * (sub - 1) + 4 * (wave - 8));
          }
        }
        LF_CLOCK(2);
        __syncthreads();
        LF_CLOCK(3);
        if constexpr (H2) {
          if (sub < NSUB - 1) stage_h(kLhSubRows * sub + kLhRing, kLhSubRows);
          else stage_h(0, kLhRing);
        }
      }
    } else {
      auto subphase = [&](auto SUB) {
        constexpr int sub = decltype(SUB)::value;
        if constexpr (FORM == 0) {
          if (sub > 0) stage();
        }
        if (PIECES || sub > 0) __syncthreads();
        LF_CLOCK(0);
        if constexpr (H2) {
          // (the convolution / row waves stage: the column waves' loads queue behind their own stores)
        } else if constexpr (PIECES) {
          // what is staged behind this sub-phase's second barrier: group sub + 2 of this item, or the first two groups of the next
          if (sub < NSUB - 1) {
            prefetch_grp(it, sub + 2, 0, live);
          } else {
            prefetch_grp(nx, 0, 0, next_live);
            prefetch_grp(nx, 1, 1, next_live);
          }
        } else {
          if (sub < NSUB - 1) prefetch(it, sub + 1, live);
          else prefetch(nx, 0, next_live);
        }
        LF_CLOCK(1);
        col_work(SUB);
        LF_CLOCK(2);
        __syncthreads();
        LF_CLOCK(3);
        if constexpr (PIECES) {
          if (sub < NSUB - 1) {
            stage_grp(sub + 2, 0);
          } else {
            stage_grp(0, 0);
            stage_grp(1, 1);
          }
        }
      };
      lf_for_const(subphase, std::make_integer_sequence<int, NSUB>());
      if constexpr (FORM == 0) stage();    // the NEXT item's first patch (fetched during the last sub-phase); the patch buffer is free
    }
    // ---- the tail: every wave transforms four rows, then the column waves read their columns
    if constexpr (PIECES) {
      if (live) row_pass(4 * wave);
    } else if constexpr (H2) {
      if constexpr (ROLE == kRowp) {
        if (live) row_pass(36 + 4 * (wave == 11 ? 2 : wave - 6));      // rows 36..47
      }
    } else {
      if (live && wave < 8) row_pass(16 + 4 * wave);      // rows 16..47 (rows 0..15: waves 8, 9 during sub-phases 1, 2)
    }
    LF_CLOCK(4);
    __syncthreads();
    LF_CLOCK(5);
    // fp32 form: the barrier that releases the tile buffer sits behind the column waves' read of tile rows 0..15 -- all the next
    // item's first sub-phase overwrites --, and they read rows 16..47 while its convolution has already started
    constexpr int kEarly = PIECES ? kFftN : (H2 ? kLhSubRows : kLfSubRows);
    if constexpr (COLS) {
      if (live) {
#pragma unroll
        for (int y = 0; y < kEarly; ++y) {
          cre[y] = colp[y * kLfRowPitch];
          cim[y] = colp[y * kLfRowPitch + 24 * 4];
        }
      }
    }
    LF_CLOCK(6);
    __syncthreads();
    LF_CLOCK(7);
    if constexpr (COLS && !PIECES) {
      if (live) {
#pragma unroll
        for (int y = kEarly; y < kFftN; ++y) {
          cre[y] = colp[y * kLfRowPitch];
          cim[y] = colp[y * kLfRowPitch + 24 * 4];
        }
      }
    }
    pend_m = it.m;
    pend_grp = it.grp;
    pending = live;
  }
  // The block's largest DC bin -> dcmax[block]; block 0 clears the slots no block writes.  With relu the activations are >= 0 and
  // |X[k]| <= X[0] for every frequency: the maximum over the kEqaLiftDcSlots slots bounds every entry of V -- what the fp16 form of the
  // channel contraction scales its operands by (eqa_fft48k5_cgemm3m_f16x2).
  if constexpr (COLS) {
    if (dcmax != nullptr && has_packed) {
      float m = dc_seen;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      if (lane == 0) dcmax[blockIdx.x] = m;
      if (blockIdx.x == 0) {
        for (unsigned t = nblk + lane; t < (unsigned)EQA_LIFT5_DCMAX_SLOTS; t += 64) dcmax[t] = 0.0f;
      }
    }
  }
  LF_CLOCK_END();
}

__global__ __launch_bounds__(kLfThreads) void lift5_fft48_fused_kernel(const float* __restrict__ x, const float* __restrict__ bank,
                                                                        const float* __restrict__ bias, int relu, float* __restrict__ V,
                                                                        int H0, int W0, int C, int TY, int TX, size_t Mp, unsigned nwork,
                                                                        unsigned v_bytes, unsigned x_bytes, unsigned bank_bytes,
                                                                        float* __restrict__ dcmax) {
  if (threadIdx.x < 6 * 64) lift5_fft48_body<kCols, 0>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, nullptr, 0, 1.0f);
  else lift5_fft48_body<kConv, 0>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, nullptr, 0, 1.0f);
}

// the same with the convolution on the bf16 matrix cores; `bank` = the weights' pieces (Cout, 3, 16, 8) bf16
__global__ __launch_bounds__(kLfThreads) void lift5_fft48_fused_pieces_kernel(const float* __restrict__ x, const float* __restrict__ bank,
                                                                               const float* __restrict__ bias, int relu, float* __restrict__ V,
                                                                               int H0, int W0, int C, int TY, int TX, size_t Mp, unsigned nwork,
                                                                               unsigned v_bytes, unsigned x_bytes, unsigned bank_bytes,
                                                                               float* __restrict__ dcmax) {
  if (threadIdx.x < 6 * 64) lift5_fft48_body<kCols, 3>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, nullptr, 0, 1.0f);
  else lift5_fft48_body<kConv, 3>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, nullptr, 0, 1.0f);
}

// the same with the convolution on two fp16 pieces (FORM 2); `bank` = the weights' pieces (Cout, 2, 16, 8) fp16 of w_scale * w
__global__ __launch_bounds__(kLfThreads) void lift5_fft48_fused_h2_kernel(const float* __restrict__ x, const float* __restrict__ bank,
                                                                           const float* __restrict__ bias, int relu, float* __restrict__ V,
                                                                           int H0, int W0, int C, int TY, int TX, size_t Mp, unsigned nwork,
                                                                           unsigned v_bytes, unsigned x_bytes, unsigned bank_bytes,
                                                                           float* __restrict__ dcmax, const float* __restrict__ xbound,
                                                                           int nxbound, float w_scale) {
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (w < 6) lift5_fft48_body<kCols, 2>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, xbound, nxbound, w_scale);
  else if (w >= 8 && w <= 10) lift5_fft48_body<kConv, 2>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, xbound, nxbound, w_scale);
  else lift5_fft48_body<kRowp, 2>(x, bank, bias, relu, V, H0, W0, C, TY, TX, Mp, nwork, v_bytes, x_bytes, bank_bytes, dcmax, xbound, nxbound, w_scale);
}

// |x| maxima in EQA_LIFT5_DCMAX_SLOTS slots (one per block; the consumer takes the largest): the bound FORM 2 scales its pixels by
__global__ __launch_bounds__(256) void absmax_slots_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  float m = 0.0f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[n4 * 4 + threadIdx.x]));
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace

extern "C" {

#ifdef EQA_LF_CLOCK
int eqa_debug_lf_clock(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lf_clock), sizeof(g_lf_clock)) == hipSuccess ? 0 : -1; }
#endif

int eqa_lift5_fft48k5_input_supported(int Cin, int KH, int KW, int Cout) {
  return (Cin == 3 && KH == 5 && KW == 5 && Cout > 0 && Cout % kLfCh == 0) ? 1 : 0;
}

static int lift_fft_launch(const float* x, const void* bank, size_t bank_bytes, const float* bias, int relu, float* V, int nimg, int H0, int W0,
                           int Cout, void* stream, bool pieces, float* dcmax = nullptr, const float* xbound = nullptr, int nxbound = 0,
                           float w_scale = 0.0f) {
  if (!x || !bank || !V || nimg < 0 || H0 < 5 || W0 < 5 || Cout <= 0) return EQA_ERR_INVALID_ARG;
  if (Cout % kLfCh != 0) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int H1 = H0 - 4, W1 = W0 - 4;
  const int TY = (int)eqa_fft48k5_tiles(H1), TX = (int)eqa_fft48k5_tiles(W1);
  if (TY <= 0 || TX <= 0) return EQA_ERR_UNSUPPORTED;
  const size_t M = (size_t)nimg * TY * TX;
  const size_t nwork = M * (Cout / kLfCh);
  const size_t vb = (size_t)kFftF * fft_pitch(M) * 2 * Cout * 4;
  const size_t xb = (size_t)nimg * H0 * W0 * 3 * 4;
  if (nwork > 0x7fffffffULL || vb > 0xffffff00ULL || xb > 0xffffff00ULL || bank_bytes > 0xffffff00ULL) return EQA_ERR_UNSUPPORTED;
  const bool h2 = xbound != nullptr;
  const void* kern = h2 ? (const void*)lift5_fft48_fused_h2_kernel : pieces ? (const void*)lift5_fft48_fused_pieces_kernel : (const void*)lift5_fft48_fused_kernel;
  const int lds_bytes = h2 ? kLhLdsBytes : pieces ? kLpLdsBytes : kLfLdsFloats * 4;
  if (!allow_dynamic_lds(kern, lds_bytes)) return EQA_ERR_UNSUPPORTED;
  // persistent: one block per CU; a multiple of the group count keeps a block on ONE channel group (its weights stay in registers)
  const unsigned ngrp = (unsigned)(Cout / kLfCh);
  unsigned nblk = 256;
  if (ngrp <= 256) nblk = (256 / ngrp) * ngrp;
  if ((size_t)nblk > nwork) nblk = (unsigned)nwork;
  if (h2)
    hipLaunchKernelGGL(lift5_fft48_fused_h2_kernel, dim3(nblk), dim3(kLfThreads), lds_bytes, (hipStream_t)stream, x, (const float*)bank, bias,
                       relu, V, H0, W0, Cout, TY, TX, fft_pitch(M), (unsigned)nwork, (unsigned)vb, (unsigned)xb, (unsigned)bank_bytes, dcmax,
                       xbound, nxbound, w_scale);
  else if (pieces)
    hipLaunchKernelGGL(lift5_fft48_fused_pieces_kernel, dim3(nblk), dim3(kLfThreads), lds_bytes, (hipStream_t)stream, x, (const float*)bank, bias,
                       relu, V, H0, W0, Cout, TY, TX, fft_pitch(M), (unsigned)nwork, (unsigned)vb, (unsigned)xb, (unsigned)bank_bytes, dcmax);
  else
    hipLaunchKernelGGL(lift5_fft48_fused_kernel, dim3(nblk), dim3(kLfThreads), lds_bytes, (hipStream_t)stream, x, (const float*)bank, bias,
                       relu, V, H0, W0, Cout, TY, TX, fft_pitch(M), (unsigned)nwork, (unsigned)vb, (unsigned)xb, (unsigned)bank_bytes, dcmax);
  return launch_status();
}

int eqa_lift5_fft48k5_input(const float* x, const float* bank, const float* bias, int relu, float* V, int nimg, int H0, int W0, int Cout,
                            void* stream) {
  return lift_fft_launch(x, bank, (size_t)(Cout > 0 ? Cout : 0) * 75 * 4, bias, relu, V, nimg, H0, W0, Cout, stream, false);
}

int eqa_lift5_fft48k5_input_dcmax(const float* x, const float* bank, const float* bias, int relu, float* V, float* dcmax, int nimg, int H0,
                                  int W0, int Cout, void* stream) {
  if (!dcmax) return EQA_ERR_INVALID_ARG;
  if (!relu) return EQA_ERR_UNSUPPORTED;            // the DC bins bound the spectrum of NON-NEGATIVE activations only
  if (nimg == 0) return hipMemsetAsync(dcmax, 0, EQA_LIFT5_DCMAX_SLOTS * sizeof(float), (hipStream_t)stream) == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
  return lift_fft_launch(x, bank, (size_t)(Cout > 0 ? Cout : 0) * 75 * 4, bias, relu, V, nimg, H0, W0, Cout, stream, false, dcmax);
}

int64_t eqa_lift5_pieces_f16_bytes(int Cout) { return Cout > 0 ? (int64_t)Cout * 2 * 5 * 4 * 8 * 2 : 0; }

int eqa_absmax_slots(const float* x, int64_t n, float* slots, void* stream) {
  if (n < 0 || !slots || (n > 0 && !x)) return EQA_ERR_INVALID_ARG;
  if (((uintptr_t)x & 15) != 0) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(absmax_slots_kernel, dim3(EQA_LIFT5_DCMAX_SLOTS), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, slots);
  return launch_status();
}

int eqa_lift5_fft48k5_input_f16x2(const float* x, const void* wpieces, float w_scale, const float* xbound, int nxbound, const float* bias,
                                  int relu, float* V, float* dcmax, int nimg, int H0, int W0, int Cout, void* stream) {
  if (!xbound || nxbound <= 0 || !(w_scale > 0.0f)) return EQA_ERR_INVALID_ARG;
  int ex = 0;
  if (frexpf(w_scale, &ex) != 0.5f) return EQA_ERR_INVALID_ARG;            // a power of two: the scaling must be exact
  if (dcmax && !relu) return EQA_ERR_UNSUPPORTED;                           // the DC bins bound the spectrum of NON-NEGATIVE activations only
  if (nimg == 0 && dcmax) return hipMemsetAsync(dcmax, 0, EQA_LIFT5_DCMAX_SLOTS * sizeof(float), (hipStream_t)stream) == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
  return lift_fft_launch(x, wpieces, (size_t)eqa_lift5_pieces_f16_bytes(Cout), bias, relu, V, nimg, H0, W0, Cout, stream, false, dcmax, xbound,
                         nxbound, w_scale);
}

int64_t eqa_lift5_pieces_bytes(int Cout) { return Cout > 0 ? (int64_t)Cout * 3 * 16 * 8 * 2 : 0; }

int eqa_lift5_fft48k5_input_bf16x3(const float* x, const void* wpieces, const float* bias, int relu, float* V, int nimg, int H0, int W0,
                                   int Cout, void* stream) {
  return lift_fft_launch(x, wpieces, (size_t)eqa_lift5_pieces_bytes(Cout), bias, relu, V, nimg, H0, W0, Cout, stream, true);
}

}  // extern "C"
