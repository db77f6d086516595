// libeqa_hip.so, part 4 of 5 -- the lifting convolution (RGB -> regular fields) as an implicit GEMM on the fp32 MFMA (I2a).
// C ABI: include/eqa_hip.h.  Design notes: HISTORY.md section 3.4.
#include "eqa_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// I2a lifting convolution on the fp32 MFMA: few input channels (RGB) -> Cout regular-field channels, KH x KW, stride 1,
// no padding, channels-last.  Implicit GEMM  y[pixel][co] = sum_kk A[pixel][kk] W[kk][co]  on v_mfma_f32_32x32x2_f32
// (f32 in, f32 accumulate: an exact fmaf chain; 64 FLOP/clk/SIMD = 157 TF peak).
//  * A wave owns a 64-channel slice and keeps ALL its weights in registers for the whole kernel (KH*8 steps x 2 N-tiles,
//    one VGPR each: lane l holds W[k = l>>5][co = l&31] of every step); waves are persistent and loop over M-tiles of 32
//    consecutive pixels of one output row, so the only streamed operand is the 3-channel input.
//  * Every wave is its own pipeline -- no barrier anywhere.  The KH input-row segments a tile reads ((31 + KW) * Cin
//    contiguous floats each) come in with <= 4 contiguous 16-byte buffer loads per lane (the descriptor is rebased on the
//    tile, so the per-lane offsets are loop constants and the hardware range check replaces all clamping), are written to
//    the wave's private LDS double buffer, and are read back in the MFMA operand layout.  Three tiles are in flight per
//    wave: tile t in the MFMA pipe, tile t+1 moving LDS -> registers row by row as the MFMAs release the rows of tile t,
//    tile t+2 moving HBM -> registers -> LDS; the previous tile's epilogue is spread over the same instruction stream.
//    (Version 1: per-lane gathers with unaligned dwordx4 loads at a 12-byte lane stride -- 1040 us.  Version 2: rows staged
//    by the block and shared by its four waves behind one barrier per tile -- 945 us; cycle stamps showed each wave
//    spending as long outside its MFMA stream (LDS round trip, staging, barrier) as inside, and with two waves per SIMD
//    that cannot overlap completely.)
//  * One filter row = R = KW*Cin <= 16 contiguous floats.  The two k-halves of the MFMA read 8 floats each from LDS: half
//    0 -> elements 0..7, half 1 -> elements R-8..R-1 (lane stride Cin dwords: conflict-free for odd Cin).  For R < 16 the
//    halves overlap; the duplicated elements carry weight 0 in half 1 (packing below), so only elements of the pixel's
//    own receptive field enter its sum.
//  * The MFMA takes the weights as A and the pixels as B: the accumulators are C[channel][pixel], each lane holds groups of 4
//    consecutive channels of one pixel.  The bias is the C operand of the first MFMA of a tile, so the epilogue is
//    ReLU + one 16-byte store per group (8 per tile and wave).
// Packed weights (host): wpk[(ky*8 + q)*2 + h][co] = w[co][ci][ky][kx],  j = (R-8)*h + q, kx = j / Cin, ci = j % Cin,
// and 0 where h == 1 and q < 16 - R.
// MFMA work at the headline shape (256 x 92 rows x 3 tiles, K = 80 incl. padding, 256 channels): 92.6 GFLOP -> 0.59 ms
// at the 157 TF peak; output 2.22 GB -> 0.37 ms at the write roofline.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef EQA_LIFT_CLOCK
__device__ unsigned long long g_lift_clock[4 * 2048];
__device__ unsigned long long g_lift_hist[64];
__device__ unsigned g_lift_launch;  // debug build: shader cycles and 100 MHz ticks of block 0, wave 0
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef EQA_LIFT_OCC
#define EQA_LIFT_OCC 1  // blocks per CU = waves per SIMD
#endif
#ifndef EQA_LIFT_WAVES
#define EQA_LIFT_WAVES (1024 * EQA_LIFT_OCC)  // persistent waves: 256 CUs x 4 SIMDs x occupancy
#endif
constexpr int kLiftChunks = 44;             // 16-byte chunks per staged input-row segment: 4*44 >= 31*Cin + R (<= 171)
constexpr int kLiftRow = 4 * kLiftChunks;   // ... in floats

template <int KH>
struct LiftStage {  // a wave's share of one tile's input rows: chunk id = lane + 64*k -> (ky = id / kLiftChunks, c = id % kLiftChunks)
  static constexpr int kIters = (KH * kLiftChunks + 63) / 64;
  static constexpr int kFloats = 64 * kIters * 4;  // LDS floats of one buffer: every lane owns a slot, used or not
  f32x4 v[kIters];
  unsigned voff[kIters];  // byte offset from the tile's first input element; tile-independent
};

template <int KH>
__device__ __forceinline__ void lift_stage_init(LiftStage<KH>& g, int lane, int W, int Cin, int n_el) {
  const int nchunk = (n_el + 3) / 4;
#pragma unroll
  for (int k = 0; k < LiftStage<KH>::kIters; ++k) {
    const int id = lane + 64 * k;
    const int ky = id / kLiftChunks, c = id % kLiftChunks;
    // slots past the segment re-read its first chunk (no predicate, no branch: the loop body stays one scheduling region)
    g.voff[k] = (ky < KH && c < nchunk) ? (unsigned)(ky * W * Cin + c * 4) * 4u : 0u;
  }
}

// Position of a tile of the wave's stream, advanced by `nstreams` tiles at a time without divisions; all wave-uniform.
struct LiftPos {
  unsigned tx, oy, img;          // tile column, output row, image
};
struct LiftStep {
  unsigned d_tx, d_oy, d_img;    // the stream's stride, decomposed the same way
};
__device__ __forceinline__ void lift_pos_init(unsigned tile, unsigned step, unsigned tiles_per_row, unsigned OH, LiftPos& p,
                                              LiftStep& d) {
  const unsigned r0 = tile / tiles_per_row, rs = step / tiles_per_row;
  p.tx = tile % tiles_per_row; p.oy = r0 % OH; p.img = r0 / OH;
  d.d_tx = step % tiles_per_row; d.d_oy = rs % OH; d.d_img = rs / OH;
}
__device__ __forceinline__ LiftPos lift_pos_next(const LiftPos& p, const LiftStep& d, unsigned tiles_per_row, unsigned OH, bool go) {
  LiftPos n;
  n.tx = p.tx + d.d_tx;
  unsigned carry = n.tx >= tiles_per_row ? 1u : 0u;
  n.tx -= carry ? tiles_per_row : 0u;
  n.oy = p.oy + d.d_oy + carry;
  carry = n.oy >= OH ? 1u : 0u;
  n.oy -= carry ? OH : 0u;
  const unsigned carry2 = n.oy >= OH ? 1u : 0u;  // d_oy + carry can reach OH
  n.oy -= carry2 ? OH : 0u;
  n.img = p.img + d.d_img + carry + carry2;
  // past the end of the stream the position stays on the last tile: its loads are redone and never used
  n.tx = go ? n.tx : p.tx; n.oy = go ? n.oy : p.oy; n.img = go ? n.img : p.img;
  return n;
}
template <bool MASKED>
__device__ __forceinline__ unsigned lift_ox0(const LiftPos& p, int OW) {
  // with OW >= 32 the last tile of a row starts at OW - 32 and overlaps its neighbour (the shared pixels are computed twice
  // from the same operands in the same order and stored twice with the same value), so every tile has 32 valid pixels and
  // the stores need no predicate; MASKED (OW < 32, or a channel count off the 64-multiples): tiles at multiples of 32, the
  // last one of a row partial, every store predicated
  return MASKED ? p.tx * 32u : min(p.tx * 32u, (unsigned)OW - 32u);
}

template <int KH>
__device__ __forceinline__ void lift_stage_load_at(const float* __restrict__ x, size_t x_numel, size_t base, LiftStage<KH>& g);

template <int KH>
__device__ __forceinline__ void lift_stage_load(const float* __restrict__ x, size_t x_numel, const LiftPos& p, unsigned ox0,
                                                int H, int W, int Cin, LiftStage<KH>& g) {
  // uniform: one 32 x 32 -> 64 bit product (input row x row length), not a chain of 64-bit multiplies
  lift_stage_load_at<KH>(x, x_numel, (size_t)(p.img * (unsigned)H + p.oy) * ((unsigned)W * (unsigned)Cin) + ox0 * (unsigned)Cin, g);
}

template <int KH>
__device__ __forceinline__ void lift_stage_load_at(const float* __restrict__ x, size_t x_numel, size_t base, LiftStage<KH>& g) {
  const size_t left = (x_numel - base) * sizeof(float);
  // a partial tile (OW < 32) and the last rows of the last image reach past the end of x: the range check returns 0 for
  // those dwords, and they only ever land on pixels that are not stored
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + base), 0,
                                                                         (int)((left >> 32) ? 0xffffffffu : (unsigned)left), 0x00020000);
#pragma unroll
  for (int k = 0; k < LiftStage<KH>::kIters; ++k) {
#ifdef EQA_LABL_NOLOAD
    const u32x4 r = {g.voff[k], 1u, 2u, 3u};
#else
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)g.voff[k], 0, 0);
#endif
    g.v[k] = __builtin_bit_cast(f32x4, r);
  }
}

// The staged values are first used one tile later (the LDS write of the next step), beyond the loop's exit test, and the
// compiler sinks loads to their first use: without this "use" at the end of the step that issued them the loads end up
// directly in front of the LDS write and their whole latency is exposed.
template <int KH>
__device__ __forceinline__ void lift_stage_pin(LiftStage<KH>& g) {
#pragma unroll
  for (int k = 0; k < LiftStage<KH>::kIters; ++k) asm volatile("" : "+v"(g.v[k]));
}

template <int KH>
__device__ __forceinline__ void lift_stage_store(float* __restrict__ lds, int lane, const LiftStage<KH>& g) {
#pragma unroll
  for (int k = 0; k < LiftStage<KH>::kIters; ++k) *reinterpret_cast<f32x4*>(lds + 4 * (lane + 64 * k)) = g.v[k];
}

// Row ky of a staged tile in the MFMA operand layout.  Element (ky 0, q 0) of k-half 1 is a spare (R <= 15: it duplicates an
// element of half 0 and its packed weight is 0): there the lanes read a constant 1.0 kept behind the stage buffer and the
// wave's weight registers hold the bias, so the bias is added by the matrix core and costs no vector instruction.
template <int KH>
__device__ __forceinline__ void lift_read_row(const float* __restrict__ lds, int ky, int a_off, int a_q0, float (&a)[KH][8]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) a[ky][q] = lds[ky == 0 && q == 0 ? a_q0 : ky * kLiftRow + a_off + q];
}

// Epilogue.  Lane l of a 32x32 accumulator C[channel][pixel] holds pixel l & 31 and, in register r, channel (r & 3) +
// 8*(r >> 2) + 4*(l >> 5) of the 32-channel tile: stored from there, one instruction writes 32-byte pieces of 32 different
// pixels (a 128-byte line is completed by four instructions, hundreds of cycles apart), and the stores cost 180 of the
// kernel's 950 us.  So the tile goes through the wave's LDS once (bias and ReLU on the way in; 68-float pixel pitch:
// conflict-free both ways) and leaves in pixel-major order: 16 lanes x 16 bytes = the 256 contiguous bytes of one pixel's
// 64-channel slice, 4 pixels per instruction, 8 instructions per tile.
constexpr int kLiftTrPitch = 68;

__device__ __forceinline__ void lift_epi_write(int g, const f32x16& p, float* __restrict__ tr) {
  f32x4 v;  // the raw sums: straight from the accumulator registers into LDS
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = p[4 * g + k];
  *reinterpret_cast<f32x4*>(tr + 8 * g) = v;  // tr = &lds_tr[col * pitch + 32 * t + 4 * h]
}

struct LiftEpi {  // loop constants of the epilogue
  float* tr_w;        // this lane's LDS slot as the owner of pixel `col` (k-half h)
  const float* tr_r;  // ... and as the writer of pixels (lane >> 4) + 4j, channels 4 * (lane & 15)
  int lo;             // ReLU as an integer max on the bit patterns (one VALU op; every VALU op costs ~6 cycles of MFMA
                      // time, tools/micro/mfma_shadow.hip): 0 clamps the negative floats, INT_MIN clamps nothing
  unsigned out_voff;  // byte offset of (pixel lane >> 4, channel slice*64 + 4 * (lane & 15)) from the tile's first output element
  unsigned row4;      // bytes of four output pixels
  int lane;
};

// The tile's output as a buffer: the descriptor is rebased on the tile (scalar work), the lane part is a loop constant and
// the pixel group goes into the scalar offset -- no 64-bit vector address arithmetic per store.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lift_out_rsrc(float* __restrict__ y, size_t y_numel, size_t base) {
  const size_t left = (y_numel - base) * sizeof(float);
  return __builtin_amdgcn_make_buffer_rsrc(y + base, 0, (int)((left >> 32) ? 0xffffffffu : (unsigned)left), 0x00020000);
}

// STATS (training: the batch-norm that follows wants per-channel sum and sum of squares of this very map): a lane sees 4 channels
// of 8 pixels per tile on their way out, always the same 4 channels -- so it keeps their running sums, packed two channels per
// register pair: 2 v_pk_add_f32 + 2 v_pk_fma_f32 per store IN PLACE of the integer max of the ReLU (no activation in front of a
// batch-norm), i.e. the same 32 vector instructions per tile as the plain kernel.  fp32 over the ~2,200 values a lane sees, fp64
// across lanes / waves (lift_stats_flush, then the caller's sum over the partial rows).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct LiftStats {
  f32x2 s01, s23, q01, q23;
};

template <bool MASKED, bool STATS = false>
__device__ __forceinline__ void lift_epi_store(int j, const f32x4& t, const LiftEpi& e, __amdgpu_buffer_rsrc_t out, int cols_left,
                                               LiftStats& st) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 lo4 = {e.lo, e.lo, e.lo, e.lo};
  u32x4 v;
  if (STATS) {
    const f32x2 a = {t[0], t[1]}, b = {t[2], t[3]};
    st.s01 += a; st.s23 += b;
    st.q01 = a * a + st.q01; st.q23 = b * b + st.q23;
    v = __builtin_bit_cast(u32x4, t);
  } else {
    v = __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(i32x4, t), lo4));
  }
#ifdef EQA_LABL_NOSTORE
  if (v[0] == 12345u)
#else
  if (!MASKED || 4 * j + (e.lane >> 4) < cols_left)
#endif
    __builtin_amdgcn_raw_buffer_store_b128(v, out, (int)e.out_voff, (int)(j * e.row4), 0);
}

template <bool MASKED, bool HALF = false, bool STATS = false>
__device__ __forceinline__ void lift_epi_all(const LiftEpi& e, const f32x16& p0, const f32x16& p1, __amdgpu_buffer_rsrc_t out, int left,
                                             LiftStats& st) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    lift_epi_write(g, p0, e.tr_w);
    if (!HALF) lift_epi_write(g, p1, e.tr_w + 32);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    lift_epi_store<MASKED, STATS>(j, *reinterpret_cast<const f32x4*>(e.tr_r + 4 * j * kLiftTrPitch), e, out, left, st);
}

// The wave's sums -> stats[(stream * Cout + channel) * 2 + {0, 1}] (fp64): the four 16-lane groups of a wave carry the same 64
// channels (different pixels), so two xor-shuffles fold them and lanes 0..15 write 4 channels each.
__device__ __forceinline__ void lift_stats_flush(const LiftStats& st, double* __restrict__ stats, unsigned stream, unsigned slice,
                                                 int Cout, int lane) {
  float v[8] = {st.s01[0], st.q01[0], st.s01[1], st.q01[1], st.s23[0], st.q23[0], st.s23[1], st.q23[1]};
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d[i] = (double)v[i];
    d[i] += __shfl_xor(d[i], 16);
    d[i] += __shfl_xor(d[i], 32);
  }
  if (lane < 16) {
    double* o = stats + ((size_t)stream * Cout + slice * 64 + 4 * lane) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = d[i];
  }
}

// One tile's MFMA stream.  Between the MFMAs: row ky of the NEXT tile moves LDS -> a[ky] as soon as the 16 MFMAs reading
// a[ky] have issued, and (EPI) the previous tile's accumulators p0 / p1 are clamped and stored in 8 groups.
// HALF: the wave's slice has at most 32 channels (Cout <= 32): the second N-tile's MFMAs, half of the stream, are not issued.
template <int KH, bool EPI, bool MASKED, bool HALF = false, bool STATS = false>
__device__ __forceinline__ void lift_tile(const float (&b0)[KH * 8], const float (&b1)[KH * 8], float (&a)[KH][8],
                                          const float* __restrict__ lds_next, int a_off, int a_q0, f32x16& acc0, f32x16& acc1,
                                          const f32x16& p0, const f32x16& p1, const LiftEpi& e,
                                          __amdgpu_buffer_rsrc_t out, int p_left, LiftStats& st) {
  f32x16 zero;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero[i] = 0.0f;
  constexpr int NG = 8 / (KH - 1);  // output pixel groups per row ky >= 1
  static_assert(NG * (KH - 1) == 8 && NG % 2 == 0, "epilogue schedule");
#pragma unroll
  for (int ky = 0; ky < KH; ++ky) {
    f32x4 t[NG];
    if (EPI && ky > 0) {  // the previous tile, pixel-major, from LDS ...
#pragma unroll
      for (int n = 0; n < NG; ++n) t[n] = *reinterpret_cast<const f32x4*>(e.tr_r + 4 * (NG * (ky - 1) + n) * kLiftTrPitch);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int s = ky * 8 + q;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a[ky][q], s == 0 ? zero : acc0, 0, 0, 0);
      if (!HALF) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a[ky][q], s == 0 ? zero : acc1, 0, 0, 0);
    }
    lift_read_row<KH>(lds_next, ky, a_off, a_q0, a);
    if (EPI) {
      if (ky == 0) {  // the previous tile: into LDS ...
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          lift_epi_write(g, p0, e.tr_w);
          if (!HALF) lift_epi_write(g, p1, e.tr_w + 32);
        }
      } else {        // ... and out: bias, ReLU, store
#pragma unroll
        for (int n = 0; n < NG; ++n) lift_epi_store<MASKED, STATS>(NG * (ky - 1) + n, t[n], e, out, p_left, st);
      }
    }
  }
  // One wave per SIMD: whatever is not issued in the shadow of an MFMA (64 cycles each) idles the matrix pipe, and the
  // default schedule bunches the MFMAs and leaves the other ~170 instructions of a tile in a few clusters (6400 cycles
  // per tile against 5120 of MFMA time).  So the whole stream is pinned: behind every MFMA at most one LDS instruction,
  // one VALU, two SALU and one VMEM instruction, in dependency order.  What that costs (tools/micro/mfma_shadow.hip, one
  // wave per SIMD, cycles per MFMA): SALU, s_nop, one ds_read or ds_write_b128: 64 (free); two ds_write_b128: 105; four
  // ds_read_b32: 128; every VALU: +6 wherever it sits (the matrix core and the vector ALU share the issue port) -- hence
  // no bias add, no accumulator copies and an integer max as the ReLU: 32 VALU instructions per tile.
#pragma unroll
  for (int m = 0; m < KH * (HALF ? 8 : 16); ++m) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);  // one LDS instruction (two writes behind one MFMA cost 40 cycles)
    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // one VALU (each costs ~6 cycles of matrix time wherever it sits)
    __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);  // SALU: free
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // VMEM
  }
}

template <int KH, bool MASKED, bool HALF = false, bool STATS = false>
__global__ __launch_bounds__(kThreads, EQA_LIFT_OCC) void lift_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                    const float* __restrict__ bias, int relu,
                                                                    float* __restrict__ y, int H, int W, int Cin, int R,
                                                                    int OH, int OW, int Cout, unsigned tiles_per_row,
                                                                    unsigned ntiles, size_t x_numel, size_t y_numel, unsigned nslices,
                                                                    unsigned nstreams, int grouped, double* __restrict__ stats) {
  using Stage = LiftStage<KH>;
  __shared__ float lds_all[kThreads / 64][2][Stage::kFloats + 4];
  __shared__ float lds_tr_all[kThreads / 64][32 * kLiftTrPitch];
  const int lane = threadIdx.x & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned gw = blockIdx.x * (kThreads / 64) + wave;  // slices of one tile stream sit in neighbouring waves (x stays in L2)
  if (gw >= nslices * nstreams) return;
  const unsigned slice = gw % nslices, stream = gw / nslices;
  float (&lds)[2][Stage::kFloats + 4] = lds_all[wave];
  if (lane < 2) lds[lane][Stage::kFloats] = 1.0f;
  const int h = lane >> 5, col = lane & 31;
  const int ch0 = slice * 64 + col;
  // Cout off the 64-multiples (MASKED launches only): the channels past Cout of the last slice get weight 0 and are never stored
  const bool in0 = ch0 < Cout, in1 = ch0 + 32 < Cout;
  float b0[KH * 8], b1[KH * 8];
#pragma unroll
  for (int s = 0; s < KH * 8; ++s) {
    b0[s] = in0 ? wpk[(size_t)(s * 2 + h) * Cout + ch0] : 0.0f;
    b1[s] = in1 ? wpk[(size_t)(s * 2 + h) * Cout + ch0 + 32] : 0.0f;
  }
  if (h == 1) {  // the spare slot (see lift_read_row): weight = bias, input = 1.0
    b0[0] = bias && in0 ? bias[ch0] : 0.0f;
    b1[0] = bias && in1 ? bias[ch0 + 32] : 0.0f;
  }
  LiftEpi epi;
  epi.tr_w = lds_tr_all[wave] + col * kLiftTrPitch + 4 * h;
  epi.tr_r = lds_tr_all[wave] + (lane >> 4) * kLiftTrPitch + 4 * (lane & 15);
  epi.lo = relu ? 0 : (int)0x80000000;
  // grouped: the output goes out as (img, channel group of 16, y, x, 16) -- what the FFT convolution's input transform reads in
  // whole cache lines.  Same store instruction: a lane's 16 bytes are 4 channels of one pixel; the four 16-lane pixel groups of
  // an instruction now write four consecutive pixels of each of the slice's four channel groups (4 runs of 256 bytes, as before).
  const unsigned plane16 = (unsigned)OH * (unsigned)OW * 16u;  // floats of one (image, channel group) plane
  epi.out_voff = grouped ? ((slice * 4 + ((lane & 15) >> 2)) * plane16 + (unsigned)(lane >> 4) * 16u + 4 * (lane & 3)) * 4u
                         : ((unsigned)(lane >> 4) * Cout + slice * 64 + 4 * (lane & 15)) * 4u;
  epi.row4 = grouped ? 4u * 16u * 4u : 4u * Cout * 4u;
  // a lane whose four channels lie past Cout fails every store predicate of the MASKED form (pixel index beyond any row)
  epi.lane = (int)(slice * 64 + 4 * (lane & 15)) < Cout ? lane : (1 << 20);
  const int n_el = 31 * Cin + R;
  const int a_off = Cin * col + (R - 8) * h;  // this lane's first element inside a staged row
  const int a_q0 = h ? Stage::kFloats : a_off;
  const unsigned count = (ntiles - stream + nstreams - 1) / nstreams;  // tiles of this stream, >= 1

#ifdef EQA_LIFT_CLOCK
  const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
  if (gw == 0 && lane == 0) g_lift_hist[atomicAdd(&g_lift_launch, 1u) & 63] = rt0;
#define LIFT_CLOCK_END() do { if (lane == 0 && gw < 2048) { g_lift_clock[4 * gw] = __builtin_readcyclecounter() - clk0; g_lift_clock[4 * gw + 1] = rt0; g_lift_clock[4 * gw + 2] = __builtin_amdgcn_s_memrealtime(); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_lift_clock[4 * gw + 3] = ((unsigned long long)xcc << 32) | hw; } } while (0)
#else
#define LIFT_CLOCK_END() do { } while (0)
#endif
  LiftStats lst;
  lst.s01 = lst.s23 = lst.q01 = lst.q23 = f32x2{0.0f, 0.0f};
  Stage g;
  lift_stage_init<KH>(g, lane, W, Cin, n_el);
  LiftPos pA, pB, pC;  // tiles i, i+1, i+2 of the stream
  LiftStep step;
  lift_pos_init(stream, nstreams, tiles_per_row, (unsigned)OH, pA, step);
  pB = lift_pos_next(pA, step, tiles_per_row, (unsigned)OH, 1 < count);
  pC = lift_pos_next(pB, step, tiles_per_row, (unsigned)OH, 2 < count);

  // both layouts as one expression: (image, pixel of the image) x (stride of an image, stride of a pixel) -- a branch on `grouped`
  // here ends the scheduling region of the tile's MFMAs, and the scalar work in front of it then runs with the matrix pipe drained
  const unsigned pix_stride = grouped ? 16u : (unsigned)Cout;
  const size_t img_stride = (size_t)OH * OW * Cout;
  auto out_of = [&](const LiftPos& p, int& cols_left) -> __amdgpu_buffer_rsrc_t {
    const unsigned ox0 = lift_ox0<MASKED>(p, OW);
    cols_left = OW - (int)ox0;
    return lift_out_rsrc(y, y_numel, (size_t)p.img * img_stride + (size_t)((p.oy * (unsigned)OW + ox0) * pix_stride));
  };

  // Tiles i+2 and i+3 of the stream are on their way from HBM while tile i is in the MFMA pipe (register sets G0 / G1;
  // a load is consumed two steps after it was issued, ~10000 cycles: nothing in the loop waits for memory).
  Stage G0, G1;
  G0 = g;
  G1 = g;
  float a[KH][8];
  lift_stage_load<KH>(x, x_numel, pA, lift_ox0<MASKED>(pA, OW), H, W, Cin, G0);
  lift_stage_store<KH>(lds[0], lane, G0);
#pragma unroll
  for (int ky = 0; ky < KH; ++ky) lift_read_row<KH>(lds[0], ky, a_off, a_q0, a);
  lift_stage_load<KH>(x, x_numel, pB, lift_ox0<MASKED>(pB, OW), H, W, Cin, G1);  // tile 1
  lift_stage_load<KH>(x, x_numel, pC, lift_ox0<MASKED>(pC, OW), H, W, Cin, G0);  // tile 2
  LiftPos pD = lift_pos_next(pC, step, tiles_per_row, (unsigned)OH, 3 < count);   // tile 3
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): the loop below starts from a clean slate
  __builtin_amdgcn_sched_barrier(0);

  // Step i = { tile i+1: registers -> lds[(i+1) & 1];  tile i+3: HBM -> the same registers;  MFMA stream of tile i with a[]
  // refilling from lds[(i+1) & 1] and the stores of tile i-1 }.  Two accumulator sets take turns (no copies, and no wait for
  // the last MFMA of a tile before the next one starts): even tiles accumulate in e0 / e1, odd tiles in o0 / o1.
  f32x16 e0, e1, o0, o1;
  int left = 0, left_next = 0;
  __amdgpu_buffer_rsrc_t po = out_of(pA, left);  // where tile 0 goes (stored during step 1)
  lift_stage_store<KH>(lds[1], lane, G1);
  lift_stage_load<KH>(x, x_numel, pD, lift_ox0<MASKED>(pD, OW), H, W, Cin, G1);
  lift_tile<KH, false, MASKED, HALF, STATS>(b0, b1, a, lds[1], a_off, a_q0, e0, e1, e0, e1, epi, po, 0, lst);
  // the first trip of the loop must not inherit a shorter queue than the later ones: the compiler takes the minimum over
  // both ways in when it counts how many loads and stores may still be in flight at the LDS writes
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  for (unsigned i = 1;; i += 2) {
    if (i >= count) {
      lift_epi_all<MASKED, HALF, STATS>(epi, e0, e1, po, left, lst);
      lift_stage_pin<KH>(G0);  // a use on the way out as well: otherwise the compiler sinks the loads below the exit test,
      lift_stage_pin<KH>(G1);  // in front of their first use one or two steps later
      if (STATS) lift_stats_flush(lst, stats, stream, slice, Cout, lane);
      LIFT_CLOCK_END();
      return;
    }
    pA = pB;
    pB = pC;
    pC = pD;
    pD = lift_pos_next(pD, step, tiles_per_row, (unsigned)OH, i + 3 < count);
    lift_stage_store<KH>(lds[0], lane, G0);
    lift_stage_load<KH>(x, x_numel, pD, lift_ox0<MASKED>(pD, OW), H, W, Cin, G0);
    __amdgpu_buffer_rsrc_t pn = out_of(pA, left_next);  // tile i: scalar work inside the stream, not in front of it
    lift_tile<KH, true, MASKED, HALF, STATS>(b0, b1, a, lds[0], a_off, a_q0, o0, o1, e0, e1, epi, po, left, lst);
    po = pn;
    left = left_next;

    if (i + 1 >= count) {
      lift_epi_all<MASKED, HALF, STATS>(epi, o0, o1, po, left, lst);
      lift_stage_pin<KH>(G0);
      lift_stage_pin<KH>(G1);
      if (STATS) lift_stats_flush(lst, stats, stream, slice, Cout, lane);
      LIFT_CLOCK_END();
      return;
    }
    pA = pB;
    pB = pC;
    pC = pD;
    pD = lift_pos_next(pD, step, tiles_per_row, (unsigned)OH, i + 4 < count);
    lift_stage_store<KH>(lds[1], lane, G1);
    lift_stage_load<KH>(x, x_numel, pD, lift_ox0<MASKED>(pD, OW), H, W, Cin, G1);
    pn = out_of(pA, left_next);  // tile i + 1
    lift_tile<KH, true, MASKED, HALF, STATS>(b0, b1, a, lds[1], a_off, a_q0, e0, e1, o0, o1, epi, po, left, lst);
    po = pn;
    left = left_next;
  }
}

// ------------------------------------------------------------------------------------------------
// DENSE form: 5 x 5 filters over 3 channels (R = 15), Cout % 64 == 0, output rows of >= 32 pixels -- the headline shape.  Two
// changes to the kernel above remove 9 % of its MFMAs:
//  * K = 76 instead of 80.  A filter row's 15 elements go to the two k-halves of the MFMA as j = q | q + 8 (q = 0..6): 7 steps per
//    row instead of 8 with an idle slot; the five left-over elements j = 7 pair up ACROSS rows: (row 0 | row 1), (row 2 | row 3),
//    (row 4 | the constant 1.0 whose weight is the bias).  38 steps.  The lanes of half 1 then read those three operands one staged
//    row further down -- a second per-lane LDS address, not a select in the stream.  The packed weights keep the layout of
//    include/eqa_hip.h; the wave only picks its registers from other slots of it.
//  * Tiles are 32 consecutive pixels of the image's FLATTENED output map, not of one output row: 92-pixel rows needed three
//    tiles (96 pixels computed, the seam stored twice); flattened, only the last tile of an image overlaps its neighbour.  A tile
//    that runs over the end of an output row finds the pixels of the next row (KW - 1) * Cin floats further into the same
//    contiguous piece of the input, so the staged segment is that much longer and the lanes behind the row end add the gap to
//    their LDS address: one compare + three selects per tile.  (OW >= 32: at most one row end per tile.)
// ------------------------------------------------------------------------------------------------
constexpr int kDenseSteps = 38;

struct DensePlace {   // wave-uniform
  size_t in_base;     // first input element of the tile
  unsigned p0;        // first output pixel (flattened, within the image)
  unsigned nb;        // pixels of the tile in front of the end of the output row (>= 32: the tile stays in its row)
};

__device__ __forceinline__ DensePlace dense_place(const LiftPos& p, unsigned P, unsigned OW, unsigned magic, int H, int W, int Cin) {
  DensePlace d;
  d.p0 = min(p.tx * 32u, P - 32u);
  const unsigned oy0 = __umulhi(d.p0, magic);   // = p0 / OW: magic = ceil(2^32 / OW), exact while OH * OW * OW < 2^32 (host-checked)
  const unsigned ox0 = d.p0 - oy0 * OW;
  d.nb = OW - ox0;
  d.in_base = (size_t)(p.img * (unsigned)H + oy0) * ((unsigned)W * (unsigned)Cin) + ox0 * (unsigned)Cin;
  return d;
}

struct DenseLane {   // loop constants: the lane's LDS addresses in stage buffer 0 for a tile that stays in its output row
  const float *a, *b, *last;   // rows at j = 0 | 8;  the j = 7 pairs (row 0 | 1), (row 2 | 3);  (row 4 | row 4 again, weight 0)
};
__device__ __forceinline__ void dense_fill(const DenseLane& l, int d, float (&a)[5][7], float (&a7)[3]) {
#pragma unroll
  for (int ky = 0; ky < 5; ++ky)
#pragma unroll
    for (int q = 0; q < 7; ++q) a[ky][q] = l.a[d + ky * kLiftRow + q];
  a7[0] = l.b[d];
  a7[1] = l.b[d + 2 * kLiftRow];
  a7[2] = l.last[d];
}

// d: the refilled tile's stage buffer offset + (for the lanes behind the end of its output row) the row gap, in floats
template <bool EPI, bool STATS>
__device__ __forceinline__ void dense_tile(const float (&b0)[kDenseSteps], const float (&b1)[kDenseSteps], float (&a)[5][7], float (&a7)[3],
                                           const DenseLane& l, int d, const f32x16& c0, const f32x16& c1, f32x16& acc0, f32x16& acc1,
                                           const f32x16& p0, const f32x16& p1, const LiftEpi& e, __amdgpu_buffer_rsrc_t out,
                                           LiftStats& st) {
#pragma unroll
  for (int ky = 0; ky < 5; ++ky) {
    f32x4 t[2];
    if (EPI && ky > 0) {
#pragma unroll
      for (int n = 0; n < 2; ++n) t[n] = *reinterpret_cast<const f32x4*>(e.tr_r + 4 * (2 * (ky - 1) + n) * kLiftTrPitch);
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int s = ky * 7 + q;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[s], a[ky][q], s == 0 ? c0 : acc0, 0, 0, 0);   // c0 / c1: the bias
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[s], a[ky][q], s == 0 ? c1 : acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) a[ky][q] = l.a[d + ky * kLiftRow + q];
    if (EPI) {
      if (ky == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          lift_epi_write(g, p0, e.tr_w);
          lift_epi_write(g, p1, e.tr_w + 32);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 2; ++n) lift_epi_store<false, STATS>(2 * (ky - 1) + n, t[n], e, out, 0, st);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[35 + k], a7[k], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[35 + k], a7[k], acc1, 0, 0, 0);
  }
  a7[0] = l.b[d];
  a7[1] = l.b[d + 2 * kLiftRow];
  a7[2] = l.last[d];
}

// the pinned order of one step's region: behind every MFMA at most one LDS instruction, one VALU, two SALU, one VMEM (lift_tile)
__device__ __forceinline__ void dense_pin() {
#pragma unroll
  for (int m = 0; m < 2 * kDenseSteps; ++m) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
  }
}

template <bool STATS>
__global__ __launch_bounds__(kThreads, 1) void lift_conv_dense_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                      const float* __restrict__ bias, int relu, float* __restrict__ y,
                                                                      int H, int W, int Cin, int OH, int OW, int Cout,
                                                                      unsigned tiles_per_img, unsigned ntiles, size_t x_numel,
                                                                      size_t y_numel, unsigned nslices, unsigned nstreams, int grouped,
                                                                      unsigned magic, double* __restrict__ stats) {
  constexpr int KH = 5;
  using Stage = LiftStage<KH>;
  constexpr int kBuf = Stage::kFloats + 4;
  __shared__ float lds_all[kThreads / 64][2][kBuf];
  __shared__ float lds_tr_all[kThreads / 64][32 * kLiftTrPitch];
  const int lane = threadIdx.x & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned gw = blockIdx.x * (kThreads / 64) + wave;
  if (gw >= nslices * nstreams) return;
  const unsigned slice = gw % nslices, stream = gw / nslices;
  float (&lds)[2][kBuf] = lds_all[wave];
  const int h = lane >> 5, col = lane & 31;
  const int ch0 = slice * 64 + col;
  // the wave's weights, picked from the packed layout (ops.pack_lift_weights / include/eqa_hip.h): element j of row ky sits at
  // [(ky * 8 + j) * 2 + 0] for j <= 7 and at [(ky * 8 + j - 7) * 2 + 1] for j >= 8
  float b0[kDenseSteps], b1[kDenseSteps];
#pragma unroll
  for (int s = 0; s < 35; ++s) {
    const int ky = s / 7, q = s % 7;
    const size_t idx = h ? (size_t)((ky * 8 + q + 1) * 2 + 1) : (size_t)((ky * 8 + q) * 2);
    b0[s] = wpk[idx * Cout + ch0];
    b1[s] = wpk[idx * Cout + ch0 + 32];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ky = 2 * k + h;   // 5: the spare k-slot -- weight 0 (its operand is row 4's element again: finite whenever the pixel's own
    const size_t idx = (size_t)((min(ky, 4) * 8 + 7) * 2);   // receptive field is)
    b0[35 + k] = ky < 5 ? wpk[idx * Cout + ch0] : 0.0f;
    b1[35 + k] = ky < 5 ? wpk[idx * Cout + ch0 + 32] : 0.0f;
  }
  // the bias is the C operand of a tile's first MFMA: register r of lane (h, pixel) is channel (r & 3) + 8 (r >> 2) + 4 h
  f32x16 c0, c1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = slice * 64 + (r & 3) + 8 * (r >> 2) + 4 * h;
    c0[r] = bias ? bias[ch] : 0.0f;
    c1[r] = bias ? bias[ch + 32] : 0.0f;
  }
  LiftEpi epi;
  epi.tr_w = lds_tr_all[wave] + col * kLiftTrPitch + 4 * h;
  epi.tr_r = lds_tr_all[wave] + (lane >> 4) * kLiftTrPitch + 4 * (lane & 15);
  epi.lo = relu ? 0 : (int)0x80000000;
  const unsigned P = (unsigned)OH * (unsigned)OW;
  epi.out_voff = grouped ? ((slice * 4 + ((lane & 15) >> 2)) * (P * 16u) + (unsigned)(lane >> 4) * 16u + 4 * (lane & 3)) * 4u
                         : ((unsigned)(lane >> 4) * Cout + slice * 64 + 4 * (lane & 15)) * 4u;
  epi.row4 = grouped ? 4u * 16u * 4u : 4u * Cout * 4u;
  epi.lane = lane;
  const int gap = (W - OW) * Cin;
  const int n_el = 31 * Cin + gap + 15;
  DenseLane dl;
  dl.a = &lds[0][0] + Cin * col + 8 * h;
  dl.b = &lds[0][0] + Cin * col + 7 + kLiftRow * h;
  dl.last = &lds[0][0] + Cin * col + 7 + 4 * kLiftRow;
  auto offset_of = [&](unsigned nb, int buf) { return ((unsigned)col >= nb ? gap : 0) + buf * kBuf; };
  const unsigned count = (ntiles - stream + nstreams - 1) / nstreams;

  LiftStats lst;
  lst.s01 = lst.s23 = lst.q01 = lst.q23 = f32x2{0.0f, 0.0f};
  Stage g;
  lift_stage_init<KH>(g, lane, W, Cin, n_el);
  LiftPos pA, pB, pC, pD, pE;   // tiles i .. i + 4 of the stream
  LiftStep step;
  lift_pos_init(stream, nstreams, tiles_per_img, 1u, pA, step);   // (tile of the image, 0, image)
  pB = lift_pos_next(pA, step, tiles_per_img, 1u, 1 < count);
  pC = lift_pos_next(pB, step, tiles_per_img, 1u, 2 < count);
  pD = lift_pos_next(pC, step, tiles_per_img, 1u, 3 < count);
  auto place = [&](const LiftPos& p) { return dense_place(p, P, (unsigned)OW, magic, H, W, Cin); };
  // (no branch on `grouped` below: a branch ends the scheduling region, and what sits outside the region of a tile's MFMAs runs
  // with the matrix pipe drained)
  const unsigned pix_stride = grouped ? 16u : (unsigned)Cout;
  const size_t img_stride = (size_t)P * Cout;
  auto out_of = [&](const LiftPos& p) -> __amdgpu_buffer_rsrc_t {
    const unsigned p0 = min(p.tx * 32u, P - 32u);
    return lift_out_rsrc(y, y_numel, (size_t)p.img * img_stride + (size_t)(p0 * pix_stride));
  };
  auto in_of = [&](const LiftPos& p) -> __amdgpu_buffer_rsrc_t {
    const size_t base = place(p).in_base;
    const size_t left = (x_numel - base) * sizeof(float);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + base), 0, (int)((left >> 32) ? 0xffffffffu : (unsigned)left), 0x00020000);
  };
  auto load = [&](Stage& G, __amdgpu_buffer_rsrc_t r) {
#pragma unroll
    for (int k = 0; k < Stage::kIters; ++k) G.v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)G.voff[k], 0, 0));
  };

  Stage G0, G1;
  G0 = g;
  G1 = g;
  float a[5][7], a7[3];
  load(G0, in_of(pA));
  lift_stage_store<KH>(lds[0], lane, G0);
  dense_fill(dl, offset_of(place(pA).nb, 0), a, a7);
  load(G1, in_of(pB));  // tile 1
  load(G0, in_of(pC));  // tile 2
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_sched_barrier(0);

  // Step i = { tile i+1: registers -> stage buffer (i+1) & 1;  tile i+3: HBM -> the same registers;  the MFMA stream of tile i,
  // a[] refilling from that buffer, the stores of tile i-1;  AND the scalar work that places step i+1's tiles (its load descriptor,
  // its store descriptor, the lanes' LDS offset of its refill) } -- one scheduling region: nothing a step's memory instructions
  // need is computed in that step, so the ~70 scalar instructions spread through the MFMA shadows instead of sitting in front.
  f32x16 e0, e1, o0, o1;
  __amdgpu_buffer_rsrc_t po = out_of(pA);                 // where the tile stored during the NEXT step goes
  __amdgpu_buffer_rsrc_t r_in = in_of(pD);                // this step's load
  int d = offset_of(place(pB).nb, 1);                     // this step's refill (tile 1, buffer 1)
  {
    lift_stage_store<KH>(lds[1], lane, G1);
    load(G1, r_in);
    dense_tile<false, STATS>(b0, b1, a, a7, dl, d, c0, c1, e0, e1, e0, e1, epi, po, lst);
    pE = lift_pos_next(pD, step, tiles_per_img, 1u, 4 < count);
    r_in = in_of(pE);
    d = offset_of(place(pC).nb, 0);
    dense_pin();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see lift_conv_mfma_kernel
  for (unsigned i = 1;; i += 2) {
    if (i >= count) {
      lift_epi_all<false, false, STATS>(epi, e0, e1, po, 0, lst);
      lift_stage_pin<KH>(G0);
      lift_stage_pin<KH>(G1);
      if (STATS) lift_stats_flush(lst, stats, stream, slice, Cout, lane);
      return;
    }
    {  // tile i (odd) -> o, stores tile i-1 (e) to po; positions: pB = tile i, pC = i+1, pD = i+2, pE = i+3
      const __amdgpu_buffer_rsrc_t pn = out_of(pB);
      lift_stage_store<KH>(lds[0], lane, G0);
      load(G0, r_in);
      dense_tile<true, STATS>(b0, b1, a, a7, dl, d, c0, c1, o0, o1, e0, e1, epi, po, lst);
      pA = pB; pB = pC; pC = pD; pD = pE;
      pE = lift_pos_next(pD, step, tiles_per_img, 1u, i + 4 < count);
      r_in = in_of(pE);
      d = offset_of(place(pC).nb, 1);
      po = pn;
      dense_pin();
    }
    if (i + 1 >= count) {
      lift_epi_all<false, false, STATS>(epi, o0, o1, po, 0, lst);
      lift_stage_pin<KH>(G0);
      lift_stage_pin<KH>(G1);
      if (STATS) lift_stats_flush(lst, stats, stream, slice, Cout, lane);
      return;
    }
    {
      const __amdgpu_buffer_rsrc_t pn = out_of(pB);
      lift_stage_store<KH>(lds[1], lane, G1);
      load(G1, r_in);
      dense_tile<true, STATS>(b0, b1, a, a7, dl, d, c0, c1, e0, e1, o0, o1, epi, po, lst);
      pA = pB; pB = pC; pC = pD; pD = pE;
      pE = lift_pos_next(pD, step, tiles_per_img, 1u, i + 5 < count);
      r_in = in_of(pE);
      d = offset_of(place(pC).nb, 0);
      po = pn;
      dense_pin();
    }
  }
}

// The unmasked kernel computes the pixels [x0, x1) of every output row twice (the last tile of a row starts at OW - 32 and
// overlaps its neighbour), so its running sums count them twice: this pass writes MINUS their sums into the partial rows behind
// the kernel's.  One block per group of output rows, thread = channel (coalesced over the channels-last map), fp64.
__global__ __launch_bounds__(kThreads) void lift_stats_dup_kernel(const float* __restrict__ y, double* __restrict__ rows, size_t nrows,
                                                                  int OW, int C, int x0, int x1) {
  // the block's (row, seam pixel) pairs as one index space, eight independent loads in flight per thread
  const unsigned npx = (unsigned)(x1 - x0);
  const size_t my_rows = (nrows - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const size_t n_items = my_rows * npx;
  for (int c = threadIdx.x; c < C; c += kThreads) {
    double s = 0.0, q = 0.0;
    for (size_t i0 = 0; i0 < n_items; i0 += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t i = i0 + k;
        const size_t r = blockIdx.x + (i / npx) * gridDim.x;
        const unsigned px = (unsigned)x0 + (unsigned)(i % npx);
        v[k] = i < n_items ? y[(r * OW + px) * C + c] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double d = (double)v[k];
        s += d;
        q += d * d;
      }
    }
    double* o = rows + ((size_t)blockIdx.x * C + c) * 2;
    o[0] = -s;
    o[1] = -q;
  }
}

constexpr unsigned kLiftDupBlocks = 1024;

}  // namespace

extern "C" {
#ifdef EQA_LIFT_CLOCK
int eqa_debug_lift_hist(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lift_hist), sizeof(g_lift_hist)) == hipSuccess ? 0 : -1; }
int eqa_debug_lift_clock(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lift_clock), sizeof(g_lift_clock)) == hipSuccess ? 0 : -1; }
#endif
// stats != nullptr: the STATS form (unmasked, >= 64-channel slices, no bias / activation); `rows_only` returns the number of
// partial rows that form writes instead of launching anything
static int lift_conv_launch(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W,
                            int Cin, int KH, int KW, int Cout, void* stream, int grouped, double* stats = nullptr,
                            bool rows_only = false) {
  if (rows_only) { x = wpk = reinterpret_cast<const float*>(16); y = reinterpret_cast<float*>(16); }
  if (!x || !wpk || !y || nimg < 0 || Cin <= 0 || KH <= 0 || KW <= 0 || Cout <= 0 || H < KH || W < KW) return EQA_ERR_INVALID_ARG;
  const int R = KW * Cin;
  if ((KH != 3 && KH != 5) || R < 9 || R > 15 || (Cout % 16) != 0 || 31 * Cin + R > kLiftRow) return EQA_ERR_UNSUPPORTED;  // R <= 15: the bias slot
  const bool narrow = (Cout % 64) != 0;   // 16 / 32 / 48 channels in the last slice: predicated stores (the MASKED form)
  if (narrow && grouped) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int OH = H - KH + 1, OW = W - KW + 1;
  const unsigned tiles_per_row = (unsigned)(OW + 31) / 32;
  const size_t ntiles = (size_t)nimg * OH * tiles_per_row;
  if (ntiles > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  const size_t x_numel = (size_t)nimg * H * W * Cin, y_numel = (size_t)nimg * OH * OW * Cout;
  // persistent waves (the weights live in registers), 2 per SIMD: `nstreams` tile streams x `nslices` 64-channel slices
  const unsigned nslices = ((unsigned)Cout + 63) / 64;
  const unsigned nstreams = (unsigned)std::min<size_t>(ntiles, std::max(1u, (unsigned)EQA_LIFT_WAVES / nslices));
  const unsigned waves = nslices * nstreams, per_block = kThreads / 64;
  const dim3 grid((waves + per_block - 1) / per_block);
  hipStream_t st = (hipStream_t)stream;
  // the DENSE form (K = 76, flattened tiles): 5 x 5 over 3 channels, whole 64-channel slices, rows of >= 32 pixels
  static const bool dense_off = [] { const char* e = getenv("EQA_LIFT_DENSE"); return e && e[0] == '0'; }();
  const uint64_t P = (uint64_t)OH * OW;
  const bool dense = !dense_off && KH == 5 && R == 15 && !narrow && OW >= 32 && (31 + KW - 1) * Cin + R <= kLiftRow &&
                     P * OW < 0x100000000ULL && P * (uint64_t)Cout < 0x100000000ULL && (size_t)nimg * ((P + 31) / 32) <= 0x7fffffffULL;
  if (dense) {
    if ((stats || rows_only) && (grouped || bias || relu)) return rows_only ? 0 : EQA_ERR_UNSUPPORTED;
    const unsigned tpi = (unsigned)((P + 31) / 32);
    const unsigned nt = (unsigned)nimg * tpi;
    const unsigned ns = (unsigned)std::min<size_t>(nt, std::max(1u, (unsigned)EQA_LIFT_WAVES / nslices));
    const dim3 dgrid((nslices * ns + per_block - 1) / per_block);
    const unsigned magic = 0xffffffffu / (unsigned)OW + 1u;   // ceil(2^32 / OW)
    const int x0 = (int)P - 32, x1 = 32 * ((int)tpi - 1);     // the pixels the last tile of an image shares with its neighbour
    const unsigned dup_blocks = (stats || rows_only) && x1 > x0 ? (unsigned)std::min<size_t>((size_t)nimg, kLiftDupBlocks) : 0u;
    if (rows_only) return (int)(ns + dup_blocks);
    if (stats)
      hipLaunchKernelGGL((lift_conv_dense_kernel<true>), dgrid, dim3(kThreads), 0, st, x, wpk, bias, relu, y, H, W, Cin, OH, OW, Cout, tpi, nt,
                         x_numel, y_numel, nslices, ns, grouped, magic, stats);
    else
      hipLaunchKernelGGL((lift_conv_dense_kernel<false>), dgrid, dim3(kThreads), 0, st, x, wpk, bias, relu, y, H, W, Cin, OH, OW, Cout, tpi, nt,
                         x_numel, y_numel, nslices, ns, grouped, magic, (double*)nullptr);
    if (dup_blocks)   // the flattened map as one "row" per image
      hipLaunchKernelGGL(lift_stats_dup_kernel, dim3(dup_blocks), dim3(kThreads), 0, st, y, stats + (size_t)ns * Cout * 2, (size_t)nimg, (int)P,
                         Cout, x0, x1);
    return launch_status();
  }
  if (stats || rows_only) {
    if (narrow || OW < 32 || grouped || bias || relu) return rows_only ? 0 : EQA_ERR_UNSUPPORTED;
    const int x0 = OW - 32, x1 = 32 * ((int)tiles_per_row - 1);  // the columns the last tile of a row shares with its neighbour
    const size_t nrows = (size_t)nimg * OH;
    const unsigned dup_blocks = x1 > x0 ? (unsigned)std::min<size_t>(nrows, kLiftDupBlocks) : 0u;
    if (rows_only) return (int)(nstreams + dup_blocks);
    if (KH == 5)
      hipLaunchKernelGGL((lift_conv_mfma_kernel<5, false, false, true>), grid, dim3(kThreads), 0, st, x, wpk, bias, relu, y, H, W, Cin, R,
                         OH, OW, Cout, tiles_per_row, (unsigned)ntiles, x_numel, y_numel, nslices, nstreams, grouped, stats);
    else
      hipLaunchKernelGGL((lift_conv_mfma_kernel<3, false, false, true>), grid, dim3(kThreads), 0, st, x, wpk, bias, relu, y, H, W, Cin, R,
                         OH, OW, Cout, tiles_per_row, (unsigned)ntiles, x_numel, y_numel, nslices, nstreams, grouped, stats);
    if (dup_blocks)
      hipLaunchKernelGGL(lift_stats_dup_kernel, dim3(dup_blocks), dim3(kThreads), 0, st, y, stats + (size_t)nstreams * Cout * 2, nrows, OW,
                         Cout, x0, x1);
    return launch_status();
  }
#define EQA_LIFT_LAUNCH(KH_, MASKED_, HALF_)                                                                                      \
  hipLaunchKernelGGL((lift_conv_mfma_kernel<KH_, MASKED_, HALF_>), grid, dim3(kThreads), 0, st, x, wpk, bias, relu, y, H, W, Cin, R, \
                     OH, OW, Cout, tiles_per_row, (unsigned)ntiles, x_numel, y_numel, nslices, nstreams, grouped, (double*)nullptr)
  if (KH == 5) {
    if (Cout <= 32) EQA_LIFT_LAUNCH(5, true, true); else if (OW < 32 || narrow) EQA_LIFT_LAUNCH(5, true, false); else EQA_LIFT_LAUNCH(5, false, false);
  } else {
    if (Cout <= 32) EQA_LIFT_LAUNCH(3, true, true); else if (OW < 32 || narrow) EQA_LIFT_LAUNCH(3, true, false); else EQA_LIFT_LAUNCH(3, false, false);
  }
#undef EQA_LIFT_LAUNCH
  return launch_status();
}

int eqa_lift_conv_nhwc(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W,
                       int Cin, int KH, int KW, int Cout, void* stream) {
  return lift_conv_launch(x, wpk, bias, relu, y, nimg, H, W, Cin, KH, KW, Cout, stream, 0);
}

int eqa_lift_conv_stats_rows(int nimg, int H, int W, int Cin, int KH, int KW, int Cout) {
  const int r = lift_conv_launch(nullptr, nullptr, nullptr, 0, nullptr, nimg, H, W, Cin, KH, KW, Cout, nullptr, 0, nullptr, true);
  return r < 0 ? 0 : r;
}

int eqa_lift_conv_nhwc_stats(const float* x, const float* wpk, float* y, double* partial, int nimg, int H, int W, int Cin, int KH, int KW,
                             int Cout, void* stream) {
  if (!partial) return EQA_ERR_INVALID_ARG;
  return lift_conv_launch(x, wpk, nullptr, 0, y, nimg, H, W, Cin, KH, KW, Cout, stream, 0, partial);
}

int eqa_lift_conv_grouped(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W,
                          int Cin, int KH, int KW, int Cout, void* stream) {
  if ((size_t)(H - KH + 1) * (W - KW + 1) * Cout * 4 > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;  // per-image plane offsets are 32-bit
  return lift_conv_launch(x, wpk, bias, relu, y, nimg, H, W, Cin, KH, KW, Cout, stream, 1);
}

}  // extern "C"
