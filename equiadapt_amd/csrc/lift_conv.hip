// libeqa_hip.so, part 4 of 5 -- the lifting convolution (RGB -> regular fields) as an implicit GEMM on the fp32 MFMA (I2a).
// C ABI: include/eqa_hip.h.  Design notes: DESIGN.md section 3.4.
#include "eqa_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// I2a lifting convolution on the fp32 MFMA: few input channels (RGB) -> Cout regular-field channels, KH x KW, stride 1,
// no padding, channels-last.  Implicit GEMM  y[pixel][co] = sum_kk A[pixel][kk] W[kk][co]  on v_mfma_f32_32x32x2_f32
// (f32 in, f32 accumulate: an exact fmaf chain; 64 FLOP/clk/SIMD = 157 TF peak).
//  * A wave owns a 64-channel slice and keeps ALL its weights in registers for the whole kernel (KH*8 steps x 2 N-tiles,
//    one VGPR each: lane l holds W[k = l>>5][co = l&31] of every step); blocks are persistent and loop over M-tiles of 32
//    consecutive pixels of one output row, so the only streamed operand is the 3-channel input.
//  * The KH input-row segments a tile reads ((31 + KW) * Cin contiguous floats each) are staged into LDS by the whole
//    block with coalesced dword loads (double-buffered, one barrier per tile) and shared by the four waves.  (First
//    version: every lane gathered its own rows with unaligned dwordx4 loads at a 12-byte lane stride -- 1040 us, of which
//    480 us were those loads; this version: see DESIGN.md.)
//  * One filter row = R = KW*Cin <= 16 contiguous floats.  The two k-halves of the MFMA read 8 floats each from LDS: half
//    0 -> elements 0..7, half 1 -> elements R-8..R-1 (lane stride Cin dwords: conflict-free for odd Cin).  For R < 16 the
//    halves overlap; the duplicated elements carry weight 0 in half 1 (packing below), so only elements of the pixel's
//    own receptive field enter its sum.
//  * The MFMA takes the weights as A and the pixels as B: the accumulators are C[channel][pixel], each lane holds groups of 4
//    consecutive channels of one pixel; epilogue = bias + ReLU + one 16-byte store per group (8 per tile and wave).
// Packed weights (host): wpk[(ky*8 + q)*2 + h][co] = w[co][ci][ky][kx],  j = (R-8)*h + q, kx = j / Cin, ci = j % Cin,
// and 0 where h == 1 and q < 16 - R.
// MFMA work at the headline shape (256 x 92 rows x 3 tiles, K = 80 incl. padding, 256 channels): 92.6 GFLOP -> 0.59 ms
// at the 157 TF peak, 0.67 ms at the 2.1 GHz the chip holds under this load; output 2.22 GB -> 0.37 ms at the write
// roofline.  Measured 0.96 ms (MIOpen / CK for the same layer: 1.33 ms): PMC SQ_VALU_MFMA_BUSY_CYCLES / active cycles =
// 70 % (78 % with the stores compiled out, 84 % with loads and stores compiled out) -- see DESIGN.md for what was tried.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef EQA_LIFT_BLOCKS
#define EQA_LIFT_BLOCKS 512
#endif
constexpr int kLiftRow = 192;  // LDS floats per staged input-row segment >= (31 + KW) * Cin = 31*Cin + R <= 31*5 + 16

template <int KH>
struct LiftStage {  // the block's share of one tile's input rows: element e = tid + 256*k -> (ky = e / kLiftRow, c = e % kLiftRow)
  static constexpr int kIters = (KH * kLiftRow + kThreads - 1) / kThreads;
  float v[kIters];
};

template <int KH>
__device__ __forceinline__ void lift_stage_load(const float* __restrict__ x, unsigned rowid, unsigned ox0, int H, int W,
                                                int Cin, int OH, int n_el, size_t x_last, LiftStage<KH>& g) {
  const unsigned img = rowid / (unsigned)OH, oy = rowid % (unsigned)OH;  // uniform
  const size_t base = (((size_t)img * H + oy) * W + ox0) * Cin;
#pragma unroll
  for (int k = 0; k < LiftStage<KH>::kIters; ++k) {
    const int e = threadIdx.x + kThreads * k;
    const int ky = e / kLiftRow, c = e % kLiftRow;
    if (ky < KH && c < n_el) {
      // a partial tile (OW < 32) reaches past the row end; those values land on pixels that are not stored, the clamp
      // only keeps the address inside the buffer
      const size_t idx = base + (size_t)ky * W * Cin + c;
#ifdef EQA_LABL_NOLOAD
      g.v[k] = (float)e;
#else
      g.v[k] = x[idx < x_last ? idx : x_last];
#endif
    }
  }
}

template <int KH>
__device__ __forceinline__ void lift_stage_store(float* __restrict__ lds, int n_el, const LiftStage<KH>& g) {
#pragma unroll
  for (int k = 0; k < LiftStage<KH>::kIters; ++k) {
    const int e = threadIdx.x + kThreads * k;
    if (e / kLiftRow < KH && e % kLiftRow < n_el) lds[e] = g.v[k];
  }
}

// Tile -> (output row id = img*OH + oy, first column).  With OW >= 32 the last tile of a row starts at OW - 32 and overlaps
// its neighbour (the shared pixels are computed twice from the same operands in the same order and stored twice with the
// same value), so every tile has 32 valid pixels and the stores need no predicate; MASKED (OW < 32): one partial tile.
template <bool MASKED>
__device__ __forceinline__ void lift_tile_pos(unsigned tile, unsigned tiles_per_row, int OW, unsigned& rowid, unsigned& ox0) {
  rowid = tile / tiles_per_row;
  const unsigned tx = tile % tiles_per_row;
  ox0 = MASKED ? 0u : min(tx * 32u, (unsigned)OW - 32u);
}

// Epilogue.  The MFMA is issued with the WEIGHTS as the A operand and the pixels as B, so the 32x32 result is
// C[channel][pixel]: lane l holds pixel l & 31 and, in register r, channel (r & 3) + 8*(r >> 2) + 4*(l >> 5) of the
// 32-channel tile -- four groups of four CONSECUTIVE channels.  One group = one 16-byte store per lane: 8 store instructions
// per tile and wave instead of the 32 dword stores of the pixel-major layout (with 8 waves per CU those 256 store
// instructions per tile period were suspected of holding the MFMAs of their waves up; measured, the time is the same).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool MASKED>
__device__ __forceinline__ void lift_store_group(int g, const f32x16& p, const f32x4& bias4, float lo, float* __restrict__ o,
                                                 int col, int cols_left) {
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = fmaxf(p[4 * g + k] + bias4[k], lo);
#ifdef EQA_LABL_NOSTORE
  if (v[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(o + 8 * g) = v;
#else
  if (!MASKED || col < cols_left) *reinterpret_cast<f32x4*>(o + 8 * g) = v;
#endif
}

template <int KH, bool MASKED>
__global__ __launch_bounds__(kThreads, 2) void lift_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias, int relu,
                                                                    float* __restrict__ y, int H, int W, int Cin, int R,
                                                                    int OH, int OW, int Cout, unsigned tiles_per_row,
                                                                    unsigned ntiles, size_t x_last) {
  __shared__ float lds[2][KH * kLiftRow];
  constexpr int NM = KH * 16;      // MFMAs per tile and wave (KH*8 k-steps x 2 N-tiles)
  constexpr int PER = NM / 8;      // MFMAs between two epilogue groups
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a wave past the last 64-channel slice redoes an earlier slice (same values, same addresses) instead of idling: it
  // must take part in the staging and the barriers anyway, and the loop stays branch-free
  const int slice = (blockIdx.y * 4 + wave) % (Cout / 64);
  const int h = lane >> 5, col = lane & 31;
  const int ch0 = slice * 64 + col;
  float b0[KH * 8], b1[KH * 8];
#pragma unroll
  for (int s = 0; s < KH * 8; ++s) {
    b0[s] = wpk[(size_t)(s * 2 + h) * Cout + ch0];
    b1[s] = wpk[(size_t)(s * 2 + h) * Cout + ch0 + 32];
  }
  f32x4 bias4[2][4];  // this lane's channels: slice*64 + 32*t + 8*g + 4*h + {0..3}
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
      for (int k = 0; k < 4; ++k) bias4[t][gq][k] = bias ? bias[slice * 64 + 32 * t + 8 * gq + 4 * h + k] : 0.0f;
  const float lo = relu ? 0.0f : -__builtin_huge_valf();
  const int n_el = 31 * Cin + R;
  const int a_off = Cin * col + (R - 8) * h;  // this lane's first element inside a staged row
  unsigned tile = blockIdx.x;
  if (tile >= ntiles) return;
  LiftStage<KH> g;
  unsigned rowid, ox0;
  lift_tile_pos<MASKED>(tile, tiles_per_row, OW, rowid, ox0);
  lift_stage_load<KH>(x, rowid, ox0, H, W, Cin, OH, n_el, x_last, g);
  lift_stage_store<KH>(lds[0], n_el, g);
  __syncthreads();
  int buf = 0;
  f32x16 p0, p1;          // accumulators of the previous tile, stored while the current tile is in the MFMA pipe
  float* po = y;
  int p_left = 0;
  bool have_prev = false;
  // weights and bias have landed: without this the first use of `bias` INSIDE the loop carries a vmcnt(0) that, from the
  // second iteration on, waits for the staging loads issued a moment earlier
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  for (;;) {
    const unsigned next = tile + gridDim.x;
    if (next < ntiles) {  // the global loads fly under this tile's MFMAs
      unsigned nr, nx;
      lift_tile_pos<MASKED>(next, tiles_per_row, OW, nr, nx);
      lift_stage_load<KH>(x, nr, nx, H, W, Cin, OH, n_el, x_last, g);
    }
    float a[KH][8];
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
      for (int q = 0; q < 8; ++q) a[ky][q] = lds[buf][ky * kLiftRow + a_off + q];
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    // The waves sharing a SIMD fall into step (they wait for the same MFMA pipe), so an epilogue that is a phase of
    // its own leaves the pipe idle: measured 70 % MFMA-busy with 2, 3 or 4 waves per SIMD alike.  Hence the previous
    // tile's bias / ReLU / stores are issued in 8 groups between this tile's MFMAs.
    if (have_prev) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int m = PER * i; m < PER * i + PER; ++m) {
          const int step = m >> 1;
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[step], a[step >> 3][step & 7], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[step], a[step >> 3][step & 7], acc0, 0, 0, 0);
        }
        if (i < 4) lift_store_group<MASKED>(i, p0, bias4[0][i], lo, po, col, p_left);
        else lift_store_group<MASKED>(i - 4, p1, bias4[1][i - 4], lo, po + 32, col, p_left);
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // PER MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    // the group's VALU (add, max)
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);    // its store
      }
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int step = m >> 1;
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[step], a[step >> 3][step & 7], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[step], a[step >> 3][step & 7], acc0, 0, 0, 0);
      }
    }
    lift_tile_pos<MASKED>(tile, tiles_per_row, OW, rowid, ox0);
    p0 = acc0;
    p1 = acc1;
    po = y + ((size_t)rowid * OW + ox0 + col) * Cout + slice * 64 + 4 * h;
    p_left = OW - (int)ox0;
    have_prev = true;
    if (next >= ntiles) break;
    lift_stage_store<KH>(lds[buf ^ 1], n_el, g);
    __syncthreads();  // everyone has read lds[buf ^ 1] two tiles ago (before the previous barrier) and lds[buf] above
    buf ^= 1;
    tile = next;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    lift_store_group<MASKED>(g, p0, bias4[0][g], lo, po, col, p_left);
    lift_store_group<MASKED>(g, p1, bias4[1][g], lo, po + 32, col, p_left);
  }
}

}  // namespace

extern "C" {

int eqa_lift_conv_nhwc(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W,
                       int Cin, int KH, int KW, int Cout, void* stream) {
  if (!x || !wpk || !y || nimg < 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout <= 0 || H < KH || W < KW) return EQA_ERR_INVALID_ARG;
  const int R = KW * Cin;
  if ((KH != 3 && KH != 5) || R < 9 || R > 16 || (Cout % 64) != 0 || 31 * Cin + R > kLiftRow) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int OH = H - KH + 1, OW = W - KW + 1;
  const unsigned tiles_per_row = (unsigned)(OW + 31) / 32;
  const size_t ntiles = (size_t)nimg * OH * tiles_per_row;
  if (ntiles > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  const size_t x_last = (size_t)nimg * H * W * Cin - 1;
  // persistent blocks (the weights live in registers), 2 per CU, each looping over M-tiles
  const dim3 grid((unsigned)std::min<size_t>(ntiles, EQA_LIFT_BLOCKS), (Cout + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define EQA_LIFT_LAUNCH(KH_, MASKED_)                                                                                      \
  hipLaunchKernelGGL((lift_conv_mfma_kernel<KH_, MASKED_>), grid, dim3(kThreads), 0, st, x, wpk, bias, relu, y, H, W, Cin, R, \
                     OH, OW, Cout, tiles_per_row, (unsigned)ntiles, x_last)
  if (KH == 5) {
    if (OW < 32) EQA_LIFT_LAUNCH(5, true); else EQA_LIFT_LAUNCH(5, false);
  } else {
    if (OW < 32) EQA_LIFT_LAUNCH(3, true); else EQA_LIFT_LAUNCH(3, false);
  }
#undef EQA_LIFT_LAUNCH
  return launch_status();
}

}  // extern "C"
