// libeqa_hip.so, part 12 -- the lifting convolution for the filters lift_conv.hip does not take: wide filter rows (7 x 7 and
// 9 x 9 over RGB: the reference tutorial's ESCNN canonicalizer is k = 9) and single-channel input (any of 3 / 5 / 7 / 9).
// Until round 4 these went to the framework's convolution (MIOpen igemm, 0.61 ms per step of the tutorial's first loop).
// C ABI: include/eqa_hip.h (eqa_lift_conv_wide).  Reference arithmetic: escnn_networks.py:60-66 (first R2Conv),
// custom_group_equivariant_layers.py lifting layer.
//
//   y[n,oy,ox,co] = [relu]( sum_{ky,kx,ci} x[n,oy+ky,ox+kx,ci] * w[co,ci,ky,kx] + bias[co] )      channels-last, stride 1, no padding
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32 with the weights as the A operand (rows = channels) and the pixels as B, like
// lift_conv.hip, but the whole filter is ONE run of R = KH*KW*Cin taps, two per matrix instruction: tap r = 2 step + (lane >> 5)
// is input element (r / RW) rows down and r % RW floats along the row (RW = KW*Cin), so no filter-row padding is spent.
//  * A wave owns a 64-channel slice and keeps its (R + 1) / 2 x 2 weight registers for the whole kernel (122 x 2 at 9 x 9 x 3);
//    it walks tiles of 32 consecutive pixels of one output row.
//  * The KH input-row segments a tile reads ((31 + KW) * Cin floats each) go HBM -> registers -> the wave's private LDS double
//    buffer one tile ahead; the B operand of a step is one ds_read_b32 (lane stride Cin dwords: conflict-free for odd Cin).
//  * No barrier anywhere: a wave is its own pipeline (one wave per SIMD: the weights take most of the register file).
//  * Known cost: 0.62 ms at the tutorial's shape is 82 cycles per matrix instruction where the instruction takes 64.  At 9 x 9 x 3
//    the 244 weight registers live in the accumulator half and the compiler copies each into a vector register in front of its
//    instruction; copying them a step early into their own registers changed nothing (0.63 ms), inline-asm instructions reading the
//    accumulator half directly ran at 0.57 ms but produced wrong values for 9 x 9 x 3 (wait states the compiler does not pad inside
//    asm) and were dropped.  Staging costs 0.06 ms of it (-DEQA_WIDE_NOSTAGE); the rest is not accounted for yet.
#include "eqa_common.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWideWaves = 1024;            // persistent waves: 256 CUs x 4 SIMDs
constexpr int kWideSegMax = 128;            // floats of one staged row segment at most: (31 + KW) * Cin <= 120
constexpr int kWideRowsMax = 9;

template <int KS, int CIN>
struct WideShape {
  static constexpr int R = KS * KS * CIN, RW = KS * CIN, STEPS = (R + 1) / 2;
  static constexpr int SEG = (31 + KS) * CIN;                 // floats of a tile's row segment
  static constexpr int NLD = (KS * SEG + 63) / 64;            // staging loads per lane and tile
  static constexpr int BUF = NLD * 64;                        // floats of one LDS buffer (every lane owns NLD slots)
};

// tap r -> offset inside the staged block (rows of SEG floats); taps past R carry weight 0 and read element 0
template <int KS, int CIN>
__host__ __device__ constexpr int wide_off(int r) {
  return r < WideShape<KS, CIN>::R ? (r / WideShape<KS, CIN>::RW) * WideShape<KS, CIN>::SEG + r % WideShape<KS, CIN>::RW : 0;
}

template <int KS, int CIN, bool MASKED>
__global__ __launch_bounds__(256, 1) void lift_conv_wide_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                                const float* __restrict__ bias, int relu, float* __restrict__ y,
                                                                int nimg, int H, int W, int Cout, int slices, int tiles_per_row) {
  using S = WideShape<KS, CIN>;
  __shared__ float lds_all[4 * 2 * S::BUF];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = blockIdx.x * 4 + wave;                       // global wave
  const int slice = gw % slices;                              // its 64-channel slice, for the whole kernel
  const int stream = gw / slices, nstreams = (int)(gridDim.x * 4) / slices;
  const int OH = H - KS + 1, OW = W - KS + 1;
  const long long ntiles = (long long)nimg * OH * tiles_per_row;
  float* lds = lds_all + wave * 2 * S::BUF;
  const int pix = lane & 31, kh = lane >> 5;

  // weights: wpk (slices, STEPS, 2, 64)
  float wreg[S::STEPS][2];
  {
    const float* wp = wpk + (size_t)slice * S::STEPS * 128 + lane;
#pragma unroll
    for (int s = 0; s < S::STEPS; ++s) {
      wreg[s][0] = wp[s * 128];
      wreg[s][1] = wp[s * 128 + 64];
    }
  }
  // the lane's bias values: accumulator e of lane (pix, kh) is channel 32 nt + 8 (e >> 2) + 4 kh + (e & 3)
  float bv[2][16];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[nt][e] = bias ? bias[slice * 64 + 32 * nt + 8 * (e >> 2) + 4 * kh + (e & 3)] : 0.0f;

  // staging slots of this lane: element id = lane + 64 j of the KS x SEG block -> (row, column)
  int srow[S::NLD], scol[S::NLD];
#pragma unroll
  for (int j = 0; j < S::NLD; ++j) {
    const int id = min(lane + 64 * j, KS * S::SEG - 1);
    srow[j] = id / S::SEG;
    scol[j] = id - srow[j] * S::SEG;
  }
  // a tile's position (tile column, output row, image), advanced by the stream's stride without divisions; all wave-uniform
  struct Pos { unsigned tx, oy, n; };
  const unsigned tpr = (unsigned)tiles_per_row, uOH = (unsigned)OH;
  auto pos_of = [&](unsigned t) { const unsigned r = t / tpr; return Pos{t % tpr, r % uOH, r / uOH}; };
  const Pos dpos = pos_of((unsigned)nstreams);
  auto pos_next = [&](const Pos& p) {
    Pos q;
    q.tx = p.tx + dpos.tx;
    unsigned c = q.tx >= tpr ? 1u : 0u;
    q.tx -= c ? tpr : 0u;
    q.oy = p.oy + dpos.oy + c;
    c = q.oy >= uOH ? 1u : 0u;
    q.oy -= c ? uOH : 0u;
    const unsigned c2 = q.oy >= uOH ? 1u : 0u;      // d.oy + carry can reach OH
    q.oy -= c2 ? uOH : 0u;
    q.n = p.n + dpos.n + c + c2;
    return q;
  };
  auto ox0_of = [&](const Pos& p) { return MASKED ? (int)p.tx * 32 : min((int)p.tx * 32, OW - 32); };  // (the last tile of a row overlaps its neighbour)
  float stage[S::NLD];
  auto request = [&](const Pos& p) {
    const int ox0 = ox0_of(p);
    const float* base = x + ((size_t)p.n * H + p.oy) * (size_t)W * CIN + (size_t)ox0 * CIN;
    const int lim = (W - ox0) * CIN - 1;                      // last float of the row segment that exists
#pragma unroll
    for (int j = 0; j < S::NLD; ++j) stage[j] = base[srow[j] * W * CIN + min(scol[j], lim)];
  };
  auto deposit = [&](float* buf) {
#pragma unroll
    for (int j = 0; j < S::NLD; ++j) buf[lane + 64 * j] = stage[j];
  };

  long long t = stream;
  if (t >= ntiles) return;
  Pos pos = pos_of((unsigned)stream);
  request(pos);
  deposit(lds);
  int cur = 0;
  for (; t < ntiles; t += nstreams) {
    const Pos pn = t + nstreams < ntiles ? pos_next(pos) : pos;
#ifndef EQA_WIDE_NOSTAGE      // (experiment switch: no staging of the next tile)
    request(pn);                                              // next tile's rows: in flight during this tile's matrix instructions
#endif
    const float* buf = lds + cur * S::BUF + pix * CIN;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = bv[nt][e];
    // The B operand of a step is one ds_read_b32.  Left to itself the compiler reads it right in front of the two matrix instructions
    // that use it and waits (~100 cycles per 128 of matrix work: 0.72 ms at the tutorial's shape); here the reads run one chunk of
    // kChunk steps ahead, one read behind every second matrix instruction.
    constexpr int kChunk = 8, kNch = (S::STEPS + kChunk - 1) / kChunk;
    float bq[2][kChunk];
#pragma unroll
    for (int j = 0; j < kChunk; ++j)
      if (j < S::STEPS) bq[0][j] = buf[kh ? wide_off<KS, CIN>(2 * j + 1) : wide_off<KS, CIN>(2 * j)];
#pragma unroll
    for (int c = 0; c < kNch; ++c) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const int sn = (c + 1) * kChunk + j;
        if (sn < S::STEPS) bq[(c + 1) & 1][j] = buf[kh ? wide_off<KS, CIN>(2 * sn + 1) : wide_off<KS, CIN>(2 * sn)];
      }
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        const int sc = c * kChunk + j;
        if (sc < S::STEPS) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[sc][0], bq[c & 1][j], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[sc][1], bq[c & 1][j], acc[1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#ifndef EQA_WIDE_NOSTAGE
    deposit(lds + (cur ^ 1) * S::BUF);
    cur ^= 1;
#endif
    // epilogue: [relu], 16-byte stores of 4 consecutive channels of the lane's pixel
    const int ox = ox0_of(pos) + pix;
#ifdef EQA_WIDE_NOSTORE        // (experiment switch: one lane stores)
    if (ox == 0x7fffffff) {
#else
    if (!MASKED || ox < OW) {
#endif
      float* o = y + (((size_t)pos.n * OH + pos.oy) * OW + ox) * (size_t)Cout + slice * 64 + 4 * kh;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]};
          if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.0f);
          }
          *reinterpret_cast<f32x4*>(o + 32 * nt + 8 * g) = v;
        }
    }
    pos = pn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Filter gradient of the same layer (training; the reference gets it from autograd's convolution-weight-gradient, which the
// framework sent to MIOpen: igemm_wrw 1.4 ms per step of the tutorial's first loop, after a "find" phase that tries a naive solver
// at 0.56 s per call):    dW[co][tap r] = sum over (n, oy, ox) of dy[n][oy][ox][co] * x[n][oy + ky][ox + kx][ci]
// As a GEMM the contraction runs over the output pixels, two per v_mfma_f32_32x32x2_f32: A = the patch values (row = tap, read from the
// same staged input rows as the forward kernel), B = dy (column = channel).  A block's four waves take 64 taps each (2 row blocks) of
// one 64-channel slice (2 column blocks): four accumulators per wave.  Tiles are 32 pixels of one output row WITHOUT overlap (a
// pixel must count once): dy of the pixels past the row's end is read as zero.  Every wave stages its own copy of the rows (no
// barrier); dy for the next tile is requested before the current tile's matrix instructions.  Each block leaves its partial
// (256 taps x 64 channels) in `part`; lift_wgrad_wide_reduce_kernel adds the partials of a slice in a fixed order (deterministic)
// and writes the bank in the framework's (Cout, Cin, KH, KW) order.
template <int KS, int CIN>
__global__ __launch_bounds__(256, 1) void lift_wgrad_wide_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 float* __restrict__ part, int nimg, int H, int W, int Cout, int slices,
                                                                 int tiles_per_row) {
  using S = WideShape<KS, CIN>;
  __shared__ float lds_all[4 * 2 * S::BUF];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = (int)blockIdx.x % slices;
  const int stream = (int)blockIdx.x / slices, nstreams = (int)gridDim.x / slices;
  const int OH = H - KS + 1, OW = W - KS + 1;
  const long long ntiles = (long long)nimg * OH * tiles_per_row;
  float* lds = lds_all + wave * 2 * S::BUF;
  const int li = lane & 31, kk = lane >> 5;

  // A operand: tap = 64 wave + 32 tb + li -> offset of its element for pixel 0 (+ kk pixels); taps past R read element 0 (their rows of
  // the partial are never read back)
  int aoff[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const int tap = 64 * wave + 32 * tb + li;
    aoff[tb] = (tap < S::R ? (tap / S::RW) * S::SEG + tap % S::RW : 0) + kk * CIN;
  }
  int srow[S::NLD], scol[S::NLD];
#pragma unroll
  for (int j = 0; j < S::NLD; ++j) {
    const int id = min(lane + 64 * j, KS * S::SEG - 1);
    srow[j] = id / S::SEG;
    scol[j] = id - srow[j] * S::SEG;
  }
  struct Pos { unsigned tx, oy, n; };
  const unsigned tpr = (unsigned)tiles_per_row, uOH = (unsigned)OH;
  auto pos_of = [&](unsigned t) { const unsigned r = t / tpr; return Pos{t % tpr, r % uOH, r / uOH}; };
  const Pos dpos = pos_of((unsigned)nstreams);
  auto pos_next = [&](const Pos& p) {
    Pos q;
    q.tx = p.tx + dpos.tx;
    unsigned c = q.tx >= tpr ? 1u : 0u;
    q.tx -= c ? tpr : 0u;
    q.oy = p.oy + dpos.oy + c;
    c = q.oy >= uOH ? 1u : 0u;
    q.oy -= c ? uOH : 0u;
    const unsigned c2 = q.oy >= uOH ? 1u : 0u;
    q.oy -= c2 ? uOH : 0u;
    q.n = p.n + dpos.n + c + c2;
    return q;
  };
  float stage[S::NLD];
  float dyr[16][2];                                           // [step][column block]: dy of pixel 2 step + kk, channel 32 cb + li
  auto request = [&](const Pos& p) {
    const int ox0 = (int)p.tx * 32;
    const float* base = x + ((size_t)p.n * H + p.oy) * (size_t)W * CIN + (size_t)ox0 * CIN;
    const int lim = (W - ox0) * CIN - 1;
#pragma unroll
    for (int j = 0; j < S::NLD; ++j) stage[j] = base[srow[j] * W * CIN + min(scol[j], lim)];
    const float* dbase = dy + (((size_t)p.n * OH + p.oy) * OW + ox0) * (size_t)Cout + slice * 64 + li;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const int px = 2 * s2 + kk;
      const bool in = ox0 + px < OW;
      const float* d = dbase + (size_t)(in ? px : 0) * Cout;
      const float v0 = d[0], v1 = d[32];
      dyr[s2][0] = in ? v0 : 0.0f;
      dyr[s2][1] = in ? v1 : 0.0f;
    }
  };
  auto deposit = [&](float* buf) {
#pragma unroll
    for (int j = 0; j < S::NLD; ++j) buf[lane + 64 * j] = stage[j];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tb][cb][e] = 0.0f;

  long long t = stream;
  if (t < ntiles) {
    Pos pos = pos_of((unsigned)stream);
    request(pos);
    int cur = 0;
    for (; t < ntiles; t += nstreams) {
      deposit(lds + cur * S::BUF);
      float b[16][2];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) { b[s2][0] = dyr[s2][0]; b[s2][1] = dyr[s2][1]; }
      const Pos pn = t + nstreams < ntiles ? pos_next(pos) : pos;
      request(pn);                                            // next tile: in flight during this tile's matrix instructions
      const float* buf = lds + cur * S::BUF;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const float a0 = buf[aoff[0] + 2 * s2 * CIN], a1 = buf[aoff[1] + 2 * s2 * CIN];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[s2][0], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[s2][1], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[s2][0], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[s2][1], acc[1][1], 0, 0, 0);
      }
      cur ^= 1;
      pos = pn;
    }
  }
  // partial of this block: part[block][tap 0..255][channel 0..63]; accumulator e of lane (li, kk): row (e & 3) + 8 (e >> 2) + 4 kk, column li
  float* pb = part + (size_t)blockIdx.x * 256 * 64;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tap = 64 * wave + 32 * tb + (e & 3) + 8 * (e >> 2) + 4 * kk;
        pb[tap * 64 + 32 * cb + li] = acc[tb][cb][e];
      }
}

// dbank[co][ci][ky][kx] = sum over the blocks of co's slice, in block order; one thread per (co, tap)
__global__ __launch_bounds__(256) void lift_wgrad_wide_reduce_kernel(const float* __restrict__ part, float* __restrict__ dbank, int Cout,
                                                                     int Cin, int KH, int KW, int slices, int nblocks) {
  const int R = KH * KW * Cin;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= Cout * R) return;
  const int co = id / R, r = id - co * R;
  const int slice = co / 64, c = co - slice * 64;
  double sum = 0.0;
  for (int b = slice; b < nblocks; b += slices) sum += (double)part[((size_t)b * 256 + r) * 64 + c];
  const int ci = r % Cin, kx = (r / Cin) % KW, ky = r / (Cin * KW);
  dbank[((size_t)(co * Cin + ci) * KH + ky) * KW + kx] = (float)sum;
}

template <int KS, int CIN>
int wide_wgrad_launch(const float* x, const float* dy, float* part, float* dbank, int nimg, int H, int W, int Cout, int nblocks, hipStream_t st) {
  const int OW = W - KS + 1, slices = Cout / 64;
  hipLaunchKernelGGL((lift_wgrad_wide_kernel<KS, CIN>), dim3(nblocks), dim3(256), 0, st, x, dy, part, nimg, H, W, Cout, slices, (OW + 31) / 32);
  const int n = Cout * KS * KS * CIN;
  hipLaunchKernelGGL(lift_wgrad_wide_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, part, dbank, Cout, CIN, KS, KS, slices, nblocks);
  return launch_status();
}

template <int KS, int CIN>
int wide_launch(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W, int Cout, hipStream_t st) {
  const int OW = W - KS + 1, slices = Cout / 64;
  const bool masked = OW < 32;
  const int tiles_per_row = (OW + 31) / 32;
  const int blocks = kWideWaves / 4;
  if (masked)
    hipLaunchKernelGGL((lift_conv_wide_kernel<KS, CIN, true>), dim3(blocks), dim3(256), 0, st, x, wpk, bias, relu, y, nimg, H, W, Cout, slices,
                       tiles_per_row);
  else
    hipLaunchKernelGGL((lift_conv_wide_kernel<KS, CIN, false>), dim3(blocks), dim3(256), 0, st, x, wpk, bias, relu, y, nimg, H, W, Cout, slices,
                       tiles_per_row);
  return launch_status();
}

}  // namespace

extern "C" {

int eqa_lift_conv_wide_supported(int Cin, int KH, int KW, int Cout) {
  if (KH != KW || Cout <= 0 || Cout % 64 || kWideWaves % (Cout / 64)) return 0;
  if (Cin == 3) return KH == 7 || KH == 9;
  if (Cin == 1) return KH == 3 || KH == 5 || KH == 7 || KH == 9;
  return 0;
}

int64_t eqa_lift_conv_wide_weight_floats(int Cin, int KH, int KW, int Cout) {
  if (!eqa_lift_conv_wide_supported(Cin, KH, KW, Cout)) return 0;
  return (int64_t)(Cout / 64) * ((KH * KW * Cin + 1) / 2) * 128;
}

int eqa_lift_conv_wide(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W, int Cin, int KH,
                       int KW, int Cout, void* stream) {
  if (nimg < 0 || H <= 0 || W <= 0) return EQA_ERR_INVALID_ARG;
  if (!eqa_lift_conv_wide_supported(Cin, KH, KW, Cout) || H < KH || W < KW) return EQA_ERR_UNSUPPORTED;
  if ((int64_t)nimg * (H - KH + 1) * ((W - KW + 32) / 32) > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;   // tiles are counted in 32 bits
  if (nimg == 0) return EQA_OK;                     // (an empty batch has no storage: nothing to check, nothing launched)
  if (!x || !wpk || !y) return EQA_ERR_INVALID_ARG;
  if (((uintptr_t)y) & 15) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define EQA_WIDE(K, C) if (KH == K && Cin == C) return wide_launch<K, C>(x, wpk, bias, relu, y, nimg, H, W, Cout, st)
  EQA_WIDE(9, 3); EQA_WIDE(7, 3); EQA_WIDE(9, 1); EQA_WIDE(7, 1); EQA_WIDE(5, 1); EQA_WIDE(3, 1);
#undef EQA_WIDE
  return EQA_ERR_UNSUPPORTED;
}

// blocks of the filter-gradient kernel: one per CU, a multiple of the slices
static int wide_wgrad_blocks(int Cout) { const int slices = Cout / 64; return (256 / slices) * slices; }

int64_t eqa_lift_conv_wide_wgrad_workspace_bytes(int Cin, int KH, int KW, int Cout) {
  if (!eqa_lift_conv_wide_supported(Cin, KH, KW, Cout) || Cout / 64 > 256) return 0;
  return (int64_t)wide_wgrad_blocks(Cout) * 256 * 64 * (int64_t)sizeof(float);
}

int eqa_lift_conv_wide_wgrad(const float* x, const float* dy, void* workspace, float* dbank, int nimg, int H, int W, int Cin, int KH,
                             int KW, int Cout, void* stream) {
  if (nimg <= 0 || H <= 0 || W <= 0 || !x || !dy || !workspace || !dbank) return EQA_ERR_INVALID_ARG;
  if (!eqa_lift_conv_wide_supported(Cin, KH, KW, Cout) || Cout / 64 > 256 || H < KH || W < KW) return EQA_ERR_UNSUPPORTED;
  if ((int64_t)nimg * (H - KH + 1) * ((W - KW + 32) / 32) > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  float* part = static_cast<float*>(workspace);
  const int nb = wide_wgrad_blocks(Cout);
#define EQA_WIDE(K, C) if (KH == K && Cin == C) return wide_wgrad_launch<K, C>(x, dy, part, dbank, nimg, H, W, Cout, nb, st)
  EQA_WIDE(9, 3); EQA_WIDE(7, 3); EQA_WIDE(9, 1); EQA_WIDE(7, 1); EQA_WIDE(5, 1); EQA_WIDE(3, 1);
#undef EQA_WIDE
  return EQA_ERR_UNSUPPORTED;
}

}  // extern "C"
