// libeqa_hip.so, part 1 of 5 -- the group action on images: fused pad / rotate / flip / crop resampling (I5, I7, I8), its
// backward, the nearest-neighbour action on masks and images (I6, GroupInference) and the crop + antialiased resize (I1).
// HBM-bound gathers: coalesced global access, LDS-staged source tiles fed by global->LDS DMA, XCD-aware block->image
// mapping (each XCD's private L2 sees whole images).  C ABI: include/eqa_hip.h.  Design notes: HISTORY.md section 3.1.
#include "eqa_common.hpp"

namespace {


constexpr int kTile = 32;      // output tile edge (px): 256 threads x 4 px
constexpr int kBox = 47;       // staged source window edge: floor(31*sqrt(2)) + neighbour + floor/guard slack = 47
constexpr int kLdsStride = 47; // odd dword stride: the 8x4-lane gather pattern is bank-conflict-free at 0/90/180/270 deg
                               // 3 channels x 47 x 47 x 4 B = 26.5 KB -> 6 blocks per CU (160 KB LDS)
constexpr int kMaxMapG = 64;   // channel-map row cached in LDS
constexpr int kRowIters = (kBox + 3) / 4;  // window rows per wave (4 waves interleave rows)

int g_force_direct = 0;

struct ActionArgs {
  const float* src;
  float* dst;
  const int32_t* gidx;
  const float* theta;
  const int32_t* flags;
  const int32_t* chan_map;
  int E, G, n_out, B, C;
  int H, W, pad, Hp, Wp;
  int OH, OW, top, left;
  float half_w, half_h, step_x, step_y;
  int force_direct;
  int lds_rows;       // window rows the launch reserved LDS for (kBox unless the caller bounds the window: eqa_group_action_fwd_hint)
  // backward only
  const float* gout;  // dL/d(output), shape of dst
  float* gsrc;        // dL/d(source), shape of src, pre-zeroed (nullable)
  float* partial;     // per (output image, tile) partial of dL/d(angle [rad]) (nullable)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// torch.linspace(-1, 1, steps) as the CPU kernel evaluates it (symmetric halves), fp32.
// Written select-style (one integer select, one fma-shaped op, one select) so it stays branch-free.
// Every multiply-add of the coordinate arithmetic is spelled out (contraction off, explicit fma) so that two evaluations of the same
// pixel are the same bits wherever they are inlined: the source window of a tile is derived from the sample points of its four
// corner pixels (group_action_body).
__device__ __forceinline__ float lin_m1_p1(int idx, int steps, float step) {
#pragma clang fp contract(off)
  const bool lo = idx < (steps >> 1);
  const float k = (float)(lo ? idx : steps - 1 - idx);
  const float up = __builtin_fmaf(step, k, -1.0f), dn = __builtin_fmaf(-step, k, 1.0f);
  return lo ? up : dn;
}
// affine_grid: [xn, yn, 1] . theta^T ; grid_sample(align_corners=True): ((g + 1) / 2) * (size - 1).  Monotone in xn for a fixed yn
// and in yn for a fixed xn (every step is a correctly rounded monotone function of its varying operand).
// Which fp32 spelling of the three-term dot product is "the reference's" depends on the host: torch's CPU affine_grid is a batched
// matrix product, and the same torch build evaluates it as x * t0, fma(y, t1, .), + t2 on an 8-thread container and as
// (x * t0 + y * t1) + t2 with every operation rounded on the 128-thread host of the GPU box (tests/cpu_arith_probe.py, both outputs
// in profiles/r04/cpu_arith_probe.txt); a GPU run of the reference goes through yet another BLAS.  The kernel pins the second,
// plain IEEE form: together with the weights and the blend below (blend4, the same on both hosts) it reproduces the oracle of the
// GPU box bit for bit on most pixels of every element, and EVERY host's oracle bit for bit for the elements that are multiples
// of 90 degrees (tests/parity_scan.py, profiles/r04/parity_scan.txt).  Rounds 1-3 left the contraction to the compiler, which
// rounded t1 * y first and fused the final multiply into the fractional part: 1.5e-4 from the oracle on white noise.
__device__ __forceinline__ void sample_point(float t0, float t1, float t2, float t3, float t4, float t5, float xn, float yn,
                                             float half_w, float half_h, float& ix, float& iy) {
#pragma clang fp contract(off)
  ix = (((t0 * xn + t1 * yn) + t2) + 1.0f) * half_w;
  iy = (((t3 * xn + t4 * yn) + t5) + 1.0f) * half_h;
}
// F.grid_sample's CPU kernel blends the four neighbours as one multiply and three fused multiply-adds in the order nw, ne, sw, se
__device__ __forceinline__ float blend4(float nw, float ne, float sw, float se, float w_nw, float w_ne, float w_sw, float w_se) {
#pragma clang fp contract(off)
  return __builtin_fmaf(se, w_se, __builtin_fmaf(sw, w_sw, __builtin_fmaf(ne, w_ne, nw * w_nw)));
}


// ablation switches for tools/ablate.sh (never set in the product build)
#ifndef EQA_ABL_MASKGUARD
#define EQA_ABL_MASKGUARD 0     // -DEQA_ABL_MASKGUARD=1: the guard ring of mask_action_u8_kernel's staged box (rounds 1-3)
#endif
#ifndef EQA_ABL_BOXGUARD
#define EQA_ABL_BOXGUARD 0.0f   // -DEQA_ABL_BOXGUARD=1e-3f: the guarded window of rounds 1-3
#endif
#ifdef EQA_ABL_NOLOAD
#define EQA_ABL_YB(yb) (a.force_direct == 12345 ? (yb) : 0)
#else
#define EQA_ABL_YB(yb) (yb)
#endif
#ifdef EQA_ABL_NOSTORE
#define EQA_ABL_STORE_OK(v) ((v) == 123.456f)
#else
#define EQA_ABL_STORE_OK(v) true
#endif
#ifndef EQA_ACTION_WAVES
#define EQA_ACTION_WAVES 1
#endif
#ifndef EQA_FORCE_CH
#define EQA_FORCE_CH 0
#endif

// Which (image, tile) does this block work on?  Full groups of 8 images: image n = 8 * bz + xcd, tile (blockIdx.x >> 3, blockIdx.y)
// -- an image per XCD.  The last, ragged group (r = n_out - 8 * bz < 8 images; the whole job when n_out < 8: config 5 runs B = 4)
// would leave 8 - r XCDs without work that way, so its r * tiles work items are dealt to the 8 XCDs in contiguous runs instead
// (an XCD still sees neighbouring tiles of one image: the overlapping source windows keep hitting in its L2); the 8 * tiles blocks
// of the group's grid slice beyond those exit at once.  Same tile, same arithmetic: results are bit-identical either way.
// The two scalar divisions are paid by the blocks of a ragged group only.
__device__ __forceinline__ bool block_tile(const int n_out, const int bz, int& n, int& tx, int& ty) {
  const int xcd = (int)(blockIdx.x & (kXcd - 1));
  tx = (int)(blockIdx.x >> 3);
  ty = (int)blockIdx.y;
  const int first = bz * kXcd, r = n_out - first;
  if (r >= kXcd) {
    n = first + xcd;
    return true;
  }
  const int tiles_x = (int)(gridDim.x >> 3), tiles = tiles_x * (int)gridDim.y;
  const int per = (r * tiles + kXcd - 1) >> 3;
  const int slot = ty * tiles_x + tx;
  const int w = xcd * per + slot;
  if (slot >= per || w >= r * tiles) return false;
  const int img = w / tiles, t = w - img * tiles;
  ty = t / tiles_x;
  tx = t - ty * tiles_x;
  n = first + img;
  return true;
}

// One block = one 32x32 output tile of one output image, all channels, CH channels per LDS stage.
//   grid = (8 * tiles_x, tiles_y, ceil(n_out / 8)):  blockIdx.x & 7 is the XCD the dispatcher deals the block
//   to, so each XCD works on whole images (n = 8*z + xcd) and the overlapping source windows of neighbouring
//   tiles hit in that XCD's private L2 (block_tile above; a ragged last group is split by tiles instead).
// Thread t owns 4 consecutive pixels of tile row t/8 (float4 stores, 128 B per 8 lanes).
// Sampling arithmetic = torch affine_grid + grid_sample(bilinear, zeros, align_corners=True) on the
// (Hp, Wp) frame, the frame itself being the edge-replicated (pad) and optionally h-flipped source.
// `bz` = the block's image-group index (blockIdx.z of a single job; blockIdx.z minus the first job's groups in a pair launch).
// MODE 0: the action itself (dst).  MODE 1: its derivative with respect to the rotation angle contracted with an output gradient
// (a.gout), one partial sum per (output image, tile) in a.partial -- the training step's angle gradient, with the same staged window
// (the derivative samples the same four neighbours the forward blends): round 3's group_action_bwd_kernel<1, false> gathered them
// from global memory, 119-169 us per 256 images; it remains the fallback for forced-direct runs and the reference of the tests.
template <int CH, bool VEC, int MODE = 0>
__device__ __forceinline__ void group_action_body(const ActionArgs& a, const int bz) {
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: row math stays on the SALU
  int n, tile_x, tile_y;
  if (!block_tile(a.n_out, bz, n, tile_x, tile_y)) return;

  int e, b;
  if (a.gidx) {
    e = a.gidx[n];
    b = n;
  } else {  // orbit mode: element-major output, n = e * B + b
    e = n / a.B;
    b = n - e * a.B;
  }
  e = min(max(e, 0), a.E - 1);
  const int fl = a.flags ? a.flags[e] : 0;
  const float* th = a.theta + e * 6;
  const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
  const bool flip_dst = (fl & EQA_FLIP_DST) != 0, flip_src = (fl & EQA_FLIP_SRC) != 0;
#ifndef EQA_ABL_ROWWALK
  // Quarter turns read source ROWS along output COLUMNS: the blocks of an image then walk its tiles column by column, so that the
  // blocks in flight together read whole source rows (DRAM pages) -- rows of output tiles read 128-byte pieces of 32 rows 4 KB
  // apart on config 5's planes.  tools/micro/tile_shape.hip: 0.592 -> 0.629 of the HBM peak on 1024 x 1024 planes, no difference on
  // 224 x 224 ones; the kernel (profiles/r04/kbench_exact_window.txt): a batch of quarter turns 171 -> 157 us per 32 frames of
  // config 5, mixed batches and the metric's 224 x 224 frames unchanged.  (Any bijection of the image's tiles serves here.)
  if (fabsf(t1) > fabsf(t0) && gridDim.y > 1) {
    const int tiles_x = (int)(gridDim.x >> 3), tiles_y = (int)gridDim.y;
    const int lin = tile_y * tiles_x + tile_x;
    tile_x = lin / tiles_y;
    tile_y = lin - tile_x * tiles_y;
  }
#endif
  const int j0 = tile_x * kTile, i0 = tile_y * kTile;

  // frame column of output column j (post-flip: hflip of the rotated frame, then the crop)
  auto frame_x = [&](int j) { return flip_dst ? (a.Wp - 1 - (a.left + j)) : (a.left + j); };

  // ---- source window of the tile = the bounding box of the north-west neighbours its pixels have, + 1 for the south-east ones.
  // The sample point is monotone along a row and along a column of the tile (sample_point), so its extremes over the tile are
  // those of the four corner pixels -- evaluated here with the per-pixel arithmetic itself, which makes the box EXACT.  Rounds 1-3
  // bounded it with the corners of the real-valued map and a 1e-3 px guard for the rounding difference.  For elements whose
  // sample points are whole pixels (every multiple of 90 degrees: all of C4 / D4, half of C8) that guard always added the column
  // left of the tile, which lies in the PREVIOUS 128-byte line: three line requests per window row instead of two.  The copy model
  // (tools/micro/pc_tile.hip, window 35 vs 33) prices that at 7 % on 224 x 224 planes and 12 % on 1024 x 1024 ones; the exact box
  // has the extra column only where rounding really puts a sample point below its pixel (a fifth of the tiles).
  const int i1 = min(i0 + kTile - 1, a.OH - 1), j1 = min(j0 + kTile - 1, a.OW - 1);
  int x_lo, y_lo, x_hi, y_hi;
  {
    const float xa = lin_m1_p1(frame_x(j0), a.Wp, a.step_x), xb = lin_m1_p1(frame_x(j1), a.Wp, a.step_x);
    const float ya = lin_m1_p1(a.top + i0, a.Hp, a.step_y), yb = lin_m1_p1(a.top + i1, a.Hp, a.step_y);
    float cx[4], cy[4];
    sample_point(t0, t1, t2, t3, t4, t5, xa, ya, a.half_w, a.half_h, cx[0], cy[0]);
    sample_point(t0, t1, t2, t3, t4, t5, xb, ya, a.half_w, a.half_h, cx[1], cy[1]);
    sample_point(t0, t1, t2, t3, t4, t5, xa, yb, a.half_w, a.half_h, cx[2], cy[2]);
    sample_point(t0, t1, t2, t3, t4, t5, xb, yb, a.half_w, a.half_h, cx[3], cy[3]);
    const float minx_f = floorf(fminf(fminf(cx[0], cx[1]), fminf(cx[2], cx[3])) - EQA_ABL_BOXGUARD);
    const float maxx_f = floorf(fmaxf(fmaxf(cx[0], cx[1]), fmaxf(cx[2], cx[3])) + EQA_ABL_BOXGUARD);
    const float miny_f = floorf(fminf(fminf(cy[0], cy[1]), fminf(cy[2], cy[3])) - EQA_ABL_BOXGUARD);
    const float maxy_f = floorf(fmaxf(fmaxf(cy[0], cy[1]), fmaxf(cy[2], cy[3])) + EQA_ABL_BOXGUARD);
    // keep at most one ring of off-frame (zero) pixels; a tile entirely off the frame keeps a 2 x 2 window at the frame's edge
    x_lo = (int)fminf(fmaxf(minx_f, -1.0f), (float)(a.Wp - 1));
    y_lo = (int)fminf(fmaxf(miny_f, -1.0f), (float)(a.Hp - 1));
    // at least 2x2 so the clamped neighbour reads of fully off-frame pixels stay inside staged data
    x_hi = max((int)fminf(fmaxf(maxx_f, -1.0f), (float)(a.Wp - 1)) + 1, x_lo + 1);
    y_hi = max((int)fminf(fmaxf(maxy_f, -1.0f), (float)(a.Hp - 1)) + 1, y_lo + 1);
  }
  const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
  const bool use_lds = (bw <= kBox) && (bh <= a.lds_rows) && !a.force_direct;

  // ---- per-thread output pixels
  const int r = tid >> 3, q = tid & 7;
  const int i = i0 + r, jb = j0 + 4 * q;
  int lidx[4];         // LDS path: index of the north-west neighbour inside the staged window
  int gx0[4], gy0[4];  // direct path: frame coords of the north-west neighbour
  bool live[4];        // false: all four neighbours are off the frame -> exact zero
  float w00[4], w01[4], w10[4], w11[4];
  auto pixel_setup = [&](int pi, int pj) {
#ifdef EQA_ABL_CHEAPSETUP  // ablation (tools/ablate.sh): what the kernel costs without the per-pixel coordinate arithmetic (wrong pixels)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      lidx[k] = min(pi - i0, bh - 2) * (CH * kLdsStride) + min(pj - j0 + k, bw - 2);
      gx0[k] = 0; gy0[k] = 0; live[k] = true;
      w00[k] = 0.25f; w01[k] = 0.25f; w10[k] = 0.25f; w11[k] = t0;
    }
    return;
#endif
    const float yn = lin_m1_p1(a.top + pi, a.Hp, a.step_y);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // affine_grid: [xn, yn, 1] . theta^T ; grid_sample(align_corners=True): ((g + 1) / 2) * (size - 1)
      const float xn = lin_m1_p1(frame_x(pj + k), a.Wp, a.step_x);
      float ix, iy;
      sample_point(t0, t1, t2, t3, t4, t5, xn, yn, a.half_w, a.half_h, ix, iy);
      const float xf = floorf(ix), yf = floorf(iy);
      const float wx1 = ix - xf, wy1 = iy - yf;
      const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
      // neighbours entirely off the frame contribute zero (grid_sample padding_mode="zeros")
      const bool xin = (xf >= -1.0f) && (xf <= (float)(a.Wp - 1));
      const bool yin = (yf >= -1.0f) && (yf <= (float)(a.Hp - 1));
      live[k] = xin && yin;
      const int xi = xin ? (int)xf : -1, yi = yin ? (int)yf : -1;
      gx0[k] = xi;
      gy0[k] = yi;
      // (pixels of a partial tile beyond OW/OH are computed but never stored: keep their reads in the window)
      const int lx = min(max(xi - x_lo, 0), bw - 2), ly = min(max(yi - y_lo, 0), bh - 2);
#ifdef EQA_CHECK_WINDOW   // validation build (tools/fuzz_r03.py --lib): a stored pixel whose neighbours are not in the corner-derived window
      if (live[k] && pi < a.OH && pj + k < a.OW && (xi < x_lo || xi + 1 > x_hi || yi < y_lo || yi + 1 > y_hi)) __builtin_trap();
#endif
      lidx[k] = ly * (CH * kLdsStride) + lx;
      if (MODE == 0) {
        w00[k] = wy0 * wx0;  // nw
        w01[k] = wy0 * wx1;  // ne
        w10[k] = wy1 * wx0;  // sw
        w11[k] = wy1 * wx1;  // se
      } else {
        // the angle derivative needs the fractional parts themselves and the lever arm of the sample point about the frame
        // centre (ds/dphi = (-(s_y - c_y), s_x - c_x) per radian; see group_action_bwd_kernel)
        live[k] = live[k] && (pi < a.OH) && (pj + k < a.OW);
        w00[k] = wx1;
        w01[k] = wy1;
        w10[k] = -(iy - a.half_h);
        w11[k] = ix - a.half_w;
      }
    }
  };

  // the element's channel-map row (regular features) goes to LDS once
  int* s_cmap = reinterpret_cast<int*>(smem + CH * a.lds_rows * kLdsStride);
  const bool has_cmap = a.chan_map != nullptr;
  if (has_cmap) {
    if (tid < a.G) s_cmap[tid] = a.chan_map[e * a.G + tid];
    __syncthreads();
  }

  // staging role of this thread: window column `lane`; the 4 waves interleave window rows
  const bool col_ok = lane < bw;
  const int col_fx = x_lo + lane;
  const bool col_inside = (unsigned)col_fx < (unsigned)a.Wp;
  const unsigned col_off = (unsigned)min(max((flip_src ? (a.Wp - 1 - col_fx) : col_fx) - a.pad, 0), a.W - 1) * 4u;

  // frame pixel -> source offset (edge-replicated pad, optional pre-flip); `inside` = not zero padding
  auto src_offset = [&](int fy, int fx, bool& inside) -> int {
    inside = ((unsigned)fx < (unsigned)a.Wp) && ((unsigned)fy < (unsigned)a.Hp);
    int sx = flip_src ? (a.Wp - 1 - fx) : fx;
    sx = min(max(sx - a.pad, 0), a.W - 1);
    const int sy = min(max(fy - a.pad, 0), a.H - 1);
    return sy * a.W + sx;
  };

  const unsigned src_plane = (unsigned)(a.H * a.W);
  const unsigned dst_plane = (unsigned)(a.OH * a.OW);
  const bool row_ok = i < a.OH;
  float* const dst_img = a.dst + (size_t)n * ((size_t)a.C * dst_plane);

  // plane base pointers of one stage (wave-uniform; readfirstlane makes that provable)
  const float* const src_img = a.src + (size_t)b * ((size_t)a.C * src_plane);  // one 64-bit multiply per block
  auto stage_planes = [&](int c0, const float* (&planes)[CH]) {
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      const int c = min(c0 + cc, a.C - 1);
      const int cs = __builtin_amdgcn_readfirstlane(has_cmap ? (c / a.G) * a.G + s_cmap[c % a.G] : c);
      planes[cc] = src_img + (unsigned)cs * (unsigned)src_plane;  // C*H*W < 2^30 (checked on the host)
    }
  };
  // Stage one window with direct-to-LDS DMA (global_load_lds_dword): each instruction moves one window-row
  // segment L2/HBM -> LDS.  LDS address = M0 (row base, per channel) + lane*4; global address = plane (SGPR
  // pair, saddr form) + [clamped row offset (SALU) + clamped/flipped column offset] (one VGPR add per row, shared
  // by the CH channels).  No staging VGPRs, no ds_write, no select, no 64-bit address VALU.
  // Off-frame rows/columns (padding_mode="zeros") are zero-filled afterwards by the lanes/rows that own them;
  // those never issue a DMA, so there is no ordering problem.
  // Inline asm because hipcc will not pick the saddr form for the builtin.  It does not count these loads:
  // stage_wait() below is the s_waitcnt.  M0 (compiler-reserved) is saved once before the row loop and restored
  // after it; every statement that reads M0 writes it first (cdna guide 5.7).
  const bool lane_dma = col_ok && col_inside;
  // Which window pixels does the tile actually sample?  The window is the bounding BOX of the tile's pre-image; for an element
  // that is not a multiple of 90 degrees the pre-image is a rotated square and fills about half of it (45 degrees: 1250 of the
  // 47 x 47 = 2209 pixels).  Requesting the rest costs L2 -> LDS traffic and cache-line requests for nothing: the copy model
  // (tools/micro/pc_tile.hip, profiles/r04/pc_tile.txt) moves 47-wide windows at 4.81 TB/s and the same windows with the lanes
  // outside the 45-degree diamond switched off at 5.43.  A frame pixel p is a neighbour of some output pixel o of the tile iff o's
  // sample point lies within one pixel of p; the sampling map is affine, o = M (p - b), so p is wanted iff M (p - b) lies in the
  // tile's rectangle grown by the pre-image of that unit square (the row L1 norms of M) -- plus a quarter pixel of slack, three
  // orders of magnitude above the rounding difference between this evaluation and the per-pixel one below.  Per lane (window
  // column) the two coordinates are affine in the row: two adds and two compares per DMA row.
  float mask_uj = 0.0f, mask_ui = 0.0f, mask_dj = 0.0f, mask_di = 0.0f, mask_hj = __builtin_inff(), mask_hi = __builtin_inff();
#ifndef EQA_ABL_NOMASK
  {
    const float a00 = a.half_w * t0 * a.step_x, a01 = a.half_w * t1 * a.step_y, b0 = a.half_w * ((t2 - t0 - t1) + 1.0f);
    const float a10 = a.half_h * t3 * a.step_x, a11 = a.half_h * t4 * a.step_y, b1 = a.half_h * ((t5 - t3 - t4) + 1.0f);
    const float det = a00 * a11 - a01 * a10;
    if (fabsf(det) > 1e-12f) {  // (uniform) a singular map keeps every lane
      const float rdet = 1.0f / det;
      const float m00 = a11 * rdet, m01 = -a01 * rdet, m10 = -a10 * rdet, m11 = a00 * rdet;
      const float jfa = (float)frame_x(j0), jfb = (float)frame_x(j1);
      const float px = (float)col_fx - b0, py = (float)y_lo - b1;
      mask_uj = (m00 * px + m01 * py) - 0.5f * (jfa + jfb);
      mask_ui = (m10 * px + m11 * py) - ((float)a.top + 0.5f * (float)(i0 + i1));
      mask_dj = m01;
      mask_di = m11;
      mask_hj = 0.5f * fabsf(jfb - jfa) + fabsf(m00) + fabsf(m01) + 0.25f;
      mask_hi = 0.5f * (float)(i1 - i0) + fabsf(m10) + fabsf(m11) + 0.25f;
    }
  }
#endif
  const bool any_zero = (x_lo < 0) || (y_lo < 0) || (x_hi > a.Wp - 1) || (y_hi > a.Hp - 1);
  // LDS layout [window row][channel][column]: one M0 write per row serves all CH channels, each DMA adding its
  // channel's row offset through the instruction's immediate (which shifts the global address too, so the plane
  // base handed to the DMA is pre-biased by -cc*kRowB).
  constexpr int kRowB = kLdsStride * 4;  // bytes of one channel's row
  auto stage_issue = [&](const float* const (&planes)[CH]) {
    if (lane_dma) {
      // window rows inside the frame: [ya, yb); this wave takes ya + ((wave - ya) mod 4), +4, ...
      const int ya = max(-y_lo, 0), yb = min(bh, a.Hp - y_lo);
      const char* p0 = reinterpret_cast<const char*>(planes[0]);
      const char* p1 = reinterpret_cast<const char*>(planes[CH > 1 ? 1 : 0]) - kRowB;
      const char* p2 = reinterpret_cast<const char*>(planes[CH > 2 ? 2 : 0]) - 2 * kRowB;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
      const int y_first = ya + ((wave - ya) & 3);
      float vj = mask_uj + mask_dj * (float)y_first, vi = mask_ui + mask_di * (float)y_first;
      const float sj = 4.0f * mask_dj, si = 4.0f * mask_di;
#pragma unroll 1
      for (int y = y_first; y < EQA_ABL_YB(yb); y += 4, vj += sj, vi += si) {
        const int fy = y_lo + y;
        const unsigned voff = (unsigned)(min(max(fy - a.pad, 0), a.H - 1) * a.W) * 4u + col_off;
        const unsigned lrow = (unsigned)(uintptr_t)(lptr_t)(smem + y * (CH * kLdsStride));
        if (!(fabsf(vj) <= mask_hj && fabsf(vi) <= mask_hi)) continue;   // this lane's pixel of the row is outside the tile's pre-image
        if (CH == 1) {
          asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]"
                       :: [v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0) : "memory");
        } else if (CH == 2) {
          asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]\n\t"
                       "global_load_lds_dword %[v], %[p1] offset:%[o1]"
                       :: [v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0), [p1] "s"(p1), [o1] "i"(kRowB) : "memory");
        } else {
          asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]\n\t"
                       "global_load_lds_dword %[v], %[p1] offset:%[o1]\n\t"
                       "global_load_lds_dword %[v], %[p2] offset:%[o2]"
                       :: [v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [o1] "i"(kRowB),
                          [o2] "i"(2 * kRowB) : "memory");
        }
      }
      asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
    }
    if (any_zero && col_ok) {  // rare: tiles touching the zero ring of an unpadded frame
#pragma unroll 1
      for (int y = wave; y < bh; y += 4) {
        if (!(col_inside && ((unsigned)(y_lo + y) < (unsigned)a.Hp))) {
#pragma unroll
          for (int cc = 0; cc < CH; ++cc) smem[(y * CH + cc) * kLdsStride + lane] = 0.0f;
        }
      }
    }
  };
  auto stage_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed in LDS
    __syncthreads();                                   // ... and so has every other wave's
  };

  const float* planes[CH];
  stage_planes(0, planes);
  if (use_lds) stage_issue(planes);
  {
    // the per-pixel setup does not depend on the loads: it runs while the DMA is in flight.  The empty asm makes
    // its inputs opaque here so the compiler cannot hoist the arithmetic above the DMA issue.
    int pi = i, pj = jb;
    asm volatile("" : "+v"(pi), "+v"(pj));
    pixel_setup(pi, pj);
  }

  float angle_sum = 0.0f;   // MODE 1
  for (int c0 = 0; c0 < a.C; c0 += CH) {
    float acc[CH][4];
    if (MODE == 1) {
      const float* const gout_img = a.gout + (size_t)n * ((size_t)a.C * dst_plane);
      if (!use_lds) {
        // window too large for LDS (never for a rotation) or forced: everything from global memory, pixel by pixel in ROLLED loops
        // with the coordinates recomputed per pixel -- slow and rare, and it must not inflate the staged path's register budget
        if (c0 > 0) stage_planes(c0, planes);
#pragma unroll 1
        for (int cc = 0; cc < CH; ++cc) {
          if (c0 + cc >= a.C) break;
          const float* pl = cc == 0 ? planes[0] : (cc == 1 ? planes[CH > 1 ? 1 : 0] : planes[CH > 2 ? 2 : 0]);
#pragma unroll 1
          for (int k = 0; k < 4; ++k) {
            const float yn = lin_m1_p1(a.top + i, a.Hp, a.step_y), xn = lin_m1_p1(frame_x(jb + k), a.Wp, a.step_x);
            float ix, iy;
            sample_point(t0, t1, t2, t3, t4, t5, xn, yn, a.half_w, a.half_h, ix, iy);
            const float xf = floorf(ix), yf = floorf(iy);
            const bool xin = (xf >= -1.0f) && (xf <= (float)(a.Wp - 1)), yin = (yf >= -1.0f) && (yf <= (float)(a.Hp - 1));
            const bool lv = xin && yin && row_ok && (jb + k < a.OW);
            const int gx = xin ? (int)xf : -1, gy = yin ? (int)yf : -1;
            bool in00, in01, in10, in11;
            const int o00 = src_offset(gy, gx, in00), o01 = src_offset(gy, gx + 1, in01);
            const int o10 = src_offset(gy + 1, gx, in10), o11 = src_offset(gy + 1, gx + 1, in11);
            const float v00 = pl[o00], v01 = pl[o01], v10 = pl[o10], v11 = pl[o11];
            const float g = gout_img[(unsigned)(c0 + cc) * dst_plane + (lv ? (unsigned)(i * a.OW + jb + k) : 0u)];
            const float nw = in00 ? v00 : 0.0f, ne = in01 ? v01 : 0.0f, sw = in10 ? v10 : 0.0f, se = in11 ? v11 : 0.0f;
            const float wx1 = ix - xf, wy1 = iy - yf, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
            const float dix = wy0 * (ne - nw) + wy1 * (se - sw), diy = wx0 * (sw - nw) + wx1 * (se - ne);
            angle_sum += lv ? g * (dix * (-(iy - a.half_h)) + diy * (ix - a.half_w)) : 0.0f;
          }
        }
        continue;
      }
      // the output gradient of this stage's channels (a dead pixel reads the plane's first value), requested before the wait
      // for the window so that both are in flight together; same order of additions as group_action_bwd_kernel<1, false>
      float gv[CH][4];
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const unsigned cpl = (unsigned)min(c0 + cc, a.C - 1) * dst_plane;
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[cc][k] = gout_img[cpl + (live[k] ? (unsigned)(i * a.OW + jb + k) : 0u)];
      }
      if (c0 > 0) {
        stage_planes(c0, planes);
        __syncthreads();
        stage_issue(planes);
      }
      stage_wait();
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const float* s = smem + cc * kLdsStride;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float nw = s[lidx[k]], ne = s[lidx[k] + 1];
          const float sw = s[lidx[k] + CH * kLdsStride], se = s[lidx[k] + CH * kLdsStride + 1];
          const float wx1 = w00[k], wy1 = w01[k];
          const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
          const float dix = wy0 * (ne - nw) + wy1 * (se - sw);
          const float diy = wx0 * (sw - nw) + wx1 * (se - ne);
          // (a dead pixel contributes nothing -- a select, not g = 0: a non-finite neighbour must not turn 0 * inf into NaN)
          if (c0 + cc < a.C) angle_sum += live[k] ? gv[cc][k] * (dix * w10[k] + diy * w11[k]) : 0.0f;
        }
      }
      continue;
    }
    if (use_lds) {
      if (c0 > 0) {
        stage_planes(c0, planes);
        __syncthreads();  // previous stage's gathers are done with the window
        stage_issue(planes);
      }
      stage_wait();
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const float* s = smem + cc * kLdsStride;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float nw = s[lidx[k]], ne = s[lidx[k] + 1];
          const float sw = s[lidx[k] + CH * kLdsStride], se = s[lidx[k] + CH * kLdsStride + 1];
          const float v = blend4(nw, ne, sw, se, w00[k], w01[k], w10[k], w11[k]);
          acc[cc][k] = live[k] ? v : 0.0f;
        }
      }
    } else {
      // direct gather (window too large for LDS, or forced): rare fallback, same arithmetic.  Rolled loops on
      // purpose: it must not inflate the register budget of the LDS path it shares the kernel with.
      if (c0 > 0) stage_planes(c0, planes);
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[cc][k] = 0.0f;
      }
#pragma unroll 1
      for (int t = 0; t < 4 * CH; ++t) {
        const int cc = t >> 2, k = t & 3;
        // (dynamic k: read the per-pixel state through selects, not indexed registers)
        const int gx = k == 0 ? gx0[0] : k == 1 ? gx0[1] : k == 2 ? gx0[2] : gx0[3];
        const int gy = k == 0 ? gy0[0] : k == 1 ? gy0[1] : k == 2 ? gy0[2] : gy0[3];
        const float a00 = k == 0 ? w00[0] : k == 1 ? w00[1] : k == 2 ? w00[2] : w00[3];
        const float a01 = k == 0 ? w01[0] : k == 1 ? w01[1] : k == 2 ? w01[2] : w01[3];
        const float a10 = k == 0 ? w10[0] : k == 1 ? w10[1] : k == 2 ? w10[2] : w10[3];
        const float a11 = k == 0 ? w11[0] : k == 1 ? w11[1] : k == 2 ? w11[2] : w11[3];
        const bool lv = k == 0 ? live[0] : k == 1 ? live[1] : k == 2 ? live[2] : live[3];
        const float* pl = cc == 0 ? planes[0] : (cc == 1 ? planes[CH > 1 ? 1 : 0] : planes[CH > 2 ? 2 : 0]);
        bool in00, in01, in10, in11;
        const int o00 = src_offset(gy, gx, in00), o01 = src_offset(gy, gx + 1, in01);
        const int o10 = src_offset(gy + 1, gx, in10), o11 = src_offset(gy + 1, gx + 1, in11);
        const float v00 = pl[o00], v01 = pl[o01], v10 = pl[o10], v11 = pl[o11];
        float v = blend4(in00 ? v00 : 0.0f, in01 ? v01 : 0.0f, in10 ? v10 : 0.0f, in11 ? v11 : 0.0f, a00, a01, a10, a11);
        v = lv ? v : 0.0f;
#pragma unroll
        for (int c2 = 0; c2 < CH; ++c2) {
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2)
            if (c2 == cc && k2 == k) acc[c2][k2] = v;
        }
      }
    }
    if (row_ok) {
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        if (c0 + cc < a.C) {
          float* o = dst_img + (unsigned)(c0 + cc) * dst_plane + (unsigned)(i * a.OW + jb);
          if (VEC) {  // OW % 4 == 0 and jb % 4 == 0: a pixel quad is entirely inside or entirely outside the row
            if (jb < a.OW && EQA_ABL_STORE_OK(acc[cc][0]))
#ifdef EQA_ACTION_NT
            {
              // opt-in: non-temporal stores.  They win only while the SOURCE fits the 256 MB Infinity Cache (B = 256 launched
              // back to back: 47 vs 58 us); with the working set in HBM (B = 1024, or inside the real step) both forms run
              // at the copy ceiling (226 us per 1024 images = 5.44 TB/s) and the regular stores drain after the kernel
              typedef float f4v __attribute__((ext_vector_type(4)));
              f4v v4 = {acc[cc][0], acc[cc][1], acc[cc][2], acc[cc][3]};
              __builtin_nontemporal_store(v4, reinterpret_cast<f4v*>(o));
            }
#else
              *reinterpret_cast<float4*>(o) = make_float4(acc[cc][0], acc[cc][1], acc[cc][2], acc[cc][3]);
#endif
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (jb + k < a.OW) o[k] = acc[cc][k];
          }
        }
      }
    }
  }
  if (MODE == 1) {
    __shared__ float s_red[kThreads / 64];
    const float w = wave_sum_f(angle_sum);
    if (lane == 0) s_red[wave] = w;
    __syncthreads();
    if (tid == 0) {
      const int tiles_x = (int)(gridDim.x >> 3);
      a.partial[((size_t)n * gridDim.y + tile_y) * tiles_x + tile_x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
  }
}

template <int CH, bool VEC>
__global__ __launch_bounds__(kThreads, EQA_ACTION_WAVES) void group_action_kernel(const ActionArgs a) {
  group_action_body<CH, VEC>(a, (int)blockIdx.z);
}

template <int CH>
__global__ __launch_bounds__(kThreads) void group_action_angle_kernel(const ActionArgs a) {
  group_action_body<CH, false, 1>(a, (int)blockIdx.z);
}

// Two jobs with the same tile grid in ONE launch (eqa_group_action_pair: canonicalize x / invert f with the same group
// index): blocks [0, zsplit) of the z axis run job 0, the rest job 1.  The dispatcher walks z last, so job 1's first blocks
// fill the CUs that job 0's tail leaves idle -- the ~12 us drain + launch gap between two dependent-looking launches
// (they are independent) disappears.  Each block executes exactly one of the two bodies: no register or LDS cost.
template <int CH, bool VEC>
__global__ __launch_bounds__(kThreads, EQA_ACTION_WAVES) void group_action_pair_kernel(const ActionArgs a0, const ActionArgs a1,
                                                                                      const int zsplit) {
  if ((int)blockIdx.z < zsplit)
    group_action_body<CH, VEC>(a0, (int)blockIdx.z);
  else
    group_action_body<CH, VEC>(a1, (int)blockIdx.z - zsplit);
}

// ------------------------------------------------------------------------------------------------
// One-channel maps (configs[4]'s inverse on a (B, 1, 1024, 1024) output; discrete_group.py:204-238): NT tiles per block.  With one
// channel a block of the kernel above moves 4.6 KB in and 4 KB out over a life of three dependent memory round trips (element ->
// matrix -> window -> stores), and the eight blocks a CU holds keep 2.5 TB/s in flight (105 us per 32 maps of 1024 x 1024).  Here
// a block requests the windows of NT consecutive tiles of the walk TOGETHER (NT windows in LDS), then gathers and stores them one
// after the other.  The per-pixel arithmetic is that of group_action_body (the same functions in the same order): the output is
// bit-identical.  grid = (8 * ceil(tiles / NT), 1, ceil(n_out / 8)): block_tile deals (image, slot) as it deals (image, tile).
// 32 maps of 1024 x 1024, random D4 (profiles/r06/kbench_invert_c1.txt): 97 us in the general kernel, 70 / 66 us with 2 / 4 tiles
// per block (8: 95 us -- the per-tile scalars no longer fit the SGPR file), torch's copy of the same bytes 51 us.  What is left
// is vector arithmetic: the per-pixel coordinates of a tile cost the same for one channel as for three.
#ifndef EQA_ACTION_C1_TILES
#define EQA_ACTION_C1_TILES 4
#endif
int g_c1_tiles = EQA_ACTION_C1_TILES;   // eqa_set_option(3, 0 | 2 | 4): 0 = one-channel maps through group_action_kernel<1>

template <int NT, bool VEC>
__global__ __launch_bounds__(kThreads) void group_action_c1_kernel(const ActionArgs a, const int tiles_x, const int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int n, slot, ty_unused;
  if (!block_tile(a.n_out, (int)blockIdx.z, n, slot, ty_unused)) return;
  int e, b;
  if (a.gidx) {
    e = a.gidx[n];
    b = n;
  } else {
    e = n / a.B;
    b = n - e * a.B;
  }
  e = __builtin_amdgcn_readfirstlane(min(max(e, 0), a.E - 1));   // (block-uniform; the DMA below wants its plane pointer in SGPRs)
  b = __builtin_amdgcn_readfirstlane(b);
  n = __builtin_amdgcn_readfirstlane(n);
  slot = __builtin_amdgcn_readfirstlane(slot);
  const int fl = a.flags ? a.flags[e] : 0;
  const float* th = a.theta + e * 6;
  const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
  const bool flip_dst = (fl & EQA_FLIP_DST) != 0, flip_src = (fl & EQA_FLIP_SRC) != 0;
  const bool col_walk = fabsf(t1) > fabsf(t0) && tiles_y > 1;   // quarter turns walk the tiles column by column (group_action_body)
  const int tiles = tiles_x * tiles_y;
  const int win_floats = a.lds_rows * kLdsStride;
  auto frame_x = [&](int j) { return flip_dst ? (a.Wp - 1 - (a.left + j)) : (a.left + j); };
  auto src_offset = [&](int fy, int fx, bool& inside) -> int {
    inside = ((unsigned)fx < (unsigned)a.Wp) && ((unsigned)fy < (unsigned)a.Hp);
    int sx = flip_src ? (a.Wp - 1 - fx) : fx;
    sx = min(max(sx - a.pad, 0), a.W - 1);
    const int sy = min(max(fy - a.pad, 0), a.H - 1);
    return sy * a.W + sx;
  };
  const unsigned src_plane = (unsigned)(a.H * a.W), dst_plane = (unsigned)(a.OH * a.OW);
  const float* const plane = a.src + (size_t)b * src_plane;
  float* const dst_img = a.dst + (size_t)n * dst_plane;

  // the inverse of the sampling map, for the lanes of a window row that lie outside the tile's pre-image (group_action_body)
  float m00 = 0.0f, m01 = 0.0f, m10 = 0.0f, m11 = 0.0f, b0 = 0.0f, b1 = 0.0f;
  bool have_mask = false;
  {
    const float a00 = a.half_w * t0 * a.step_x, a01 = a.half_w * t1 * a.step_y;
    const float a10 = a.half_h * t3 * a.step_x, a11 = a.half_h * t4 * a.step_y;
    b0 = a.half_w * ((t2 - t0 - t1) + 1.0f);
    b1 = a.half_h * ((t5 - t3 - t4) + 1.0f);
    const float det = a00 * a11 - a01 * a10;
    if (fabsf(det) > 1e-12f) {
      const float rdet = 1.0f / det;
      m00 = a11 * rdet; m01 = -a01 * rdet; m10 = -a10 * rdet; m11 = a00 * rdet;
      have_mask = true;
    }
  }

  // ---- the windows of the slot's NT tiles: lane t of every wave evaluates tile t (one pass of the corner arithmetic per wave
  // instead of NT), v_readlane hands the results to the wave as scalars.  (With every lane evaluating every tile this part was
  // a third of the kernel's vector instructions: profiles/r06/kbench_invert_c1_v1_ablations.txt.)
  int v_i0, v_j0, v_xl, v_yl, v_bw, v_bh, v_in;
  {
    const int lin = slot * NT + min(lane, NT - 1);
    const int lc = min(lin, tiles - 1);
    int tile_x, tile_y;
    if (col_walk) { tile_x = lc / tiles_y; tile_y = lc - tile_x * tiles_y; }
    else { tile_y = lc / tiles_x; tile_x = lc - tile_y * tiles_x; }
    const int j0 = tile_x * kTile, i0 = tile_y * kTile;
    const int i1 = min(i0 + kTile - 1, a.OH - 1), j1 = min(j0 + kTile - 1, a.OW - 1);
    const float xa = lin_m1_p1(frame_x(j0), a.Wp, a.step_x), xb = lin_m1_p1(frame_x(j1), a.Wp, a.step_x);
    const float ya = lin_m1_p1(a.top + i0, a.Hp, a.step_y), yb = lin_m1_p1(a.top + i1, a.Hp, a.step_y);
    float cx[4], cy[4];
    sample_point(t0, t1, t2, t3, t4, t5, xa, ya, a.half_w, a.half_h, cx[0], cy[0]);
    sample_point(t0, t1, t2, t3, t4, t5, xb, ya, a.half_w, a.half_h, cx[1], cy[1]);
    sample_point(t0, t1, t2, t3, t4, t5, xa, yb, a.half_w, a.half_h, cx[2], cy[2]);
    sample_point(t0, t1, t2, t3, t4, t5, xb, yb, a.half_w, a.half_h, cx[3], cy[3]);
    const float minx_f = floorf(fminf(fminf(cx[0], cx[1]), fminf(cx[2], cx[3])));
    const float maxx_f = floorf(fmaxf(fmaxf(cx[0], cx[1]), fmaxf(cx[2], cx[3])));
    const float miny_f = floorf(fminf(fminf(cy[0], cy[1]), fminf(cy[2], cy[3])));
    const float maxy_f = floorf(fmaxf(fmaxf(cy[0], cy[1]), fmaxf(cy[2], cy[3])));
    const int xl = (int)fminf(fmaxf(minx_f, -1.0f), (float)(a.Wp - 1));
    const int yl = (int)fminf(fmaxf(miny_f, -1.0f), (float)(a.Hp - 1));
    const int xh = max((int)fminf(fmaxf(maxx_f, -1.0f), (float)(a.Wp - 1)) + 1, xl + 1);
    const int yh = max((int)fminf(fmaxf(maxy_f, -1.0f), (float)(a.Hp - 1)) + 1, yl + 1);
    v_i0 = i0; v_j0 = j0; v_xl = xl; v_yl = yl;
    v_bw = lin < tiles ? xh - xl + 1 : 0;                     // 0: no such tile
    v_bh = yh - yl + 1;
    // a whole tile whose window lies inside the frame: no pixel of it needs a range check or a clamp
    v_in = (xl >= 0 && yl >= 0 && xh <= a.Wp - 1 && yh <= a.Hp - 1 && i0 + kTile <= a.OH && j0 + kTile <= a.OW) ? 1 : 0;
  }
  int wi0[NT], wj0[NT], x_lo[NT], y_lo[NT], bw[NT], bh[NT];
  bool use_lds[NT], inner[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    wi0[t] = __builtin_amdgcn_readlane(v_i0, t); wj0[t] = __builtin_amdgcn_readlane(v_j0, t);
    x_lo[t] = __builtin_amdgcn_readlane(v_xl, t); y_lo[t] = __builtin_amdgcn_readlane(v_yl, t);
    bw[t] = __builtin_amdgcn_readlane(v_bw, t); bh[t] = __builtin_amdgcn_readlane(v_bh, t);
    inner[t] = __builtin_amdgcn_readlane(v_in, t) != 0;
    use_lds[t] = (bw[t] != 0) && (bw[t] <= kBox) && (bh[t] <= a.lds_rows) && !a.force_direct;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (bw[t] != 0) {               // block-uniform
      const int i0 = wi0[t], j0 = wj0[t];
      const int i1 = min(i0 + kTile - 1, a.OH - 1), j1 = min(j0 + kTile - 1, a.OW - 1);
      const int xl = x_lo[t], yl = y_lo[t];
      const int xh = xl + bw[t] - 1, yh = yl + bh[t] - 1;
      if (use_lds[t]) {
        // ---- request the window: lane = window column, the four waves interleave the window rows (global -> LDS DMA)
        float* const win = smem + t * win_floats;
        const bool col_ok = lane < bw[t];
        const int col_fx = xl + lane;
        const bool col_inside = (unsigned)col_fx < (unsigned)a.Wp;
        const unsigned col_off = (unsigned)min(max((flip_src ? (a.Wp - 1 - col_fx) : col_fx) - a.pad, 0), a.W - 1) * 4u;
        float mask_uj = 0.0f, mask_ui = 0.0f, mask_hj = __builtin_inff(), mask_hi = __builtin_inff();
        if (have_mask) {
          const float jfa = (float)frame_x(j0), jfb = (float)frame_x(j1);
          const float px = (float)col_fx - b0, py = (float)yl - b1;
          mask_uj = (m00 * px + m01 * py) - 0.5f * (jfa + jfb);
          mask_ui = (m10 * px + m11 * py) - ((float)a.top + 0.5f * (float)(i0 + i1));
          mask_hj = 0.5f * fabsf(jfb - jfa) + fabsf(m00) + fabsf(m01) + 0.25f;
          mask_hi = 0.5f * (float)(i1 - i0) + fabsf(m10) + fabsf(m11) + 0.25f;
        }
        if (col_ok && col_inside) {
          const int ya = max(-yl, 0), yb = min(bh[t], a.Hp - yl);
          const char* p0 = reinterpret_cast<const char*>(plane);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
          const int y_first = ya + ((wave - ya) & 3);
          float vj = mask_uj + m01 * (float)y_first, vi = mask_ui + m11 * (float)y_first;
          const float sj = 4.0f * m01, si = 4.0f * m11;
#pragma unroll 1
          for (int y = y_first; y < EQA_ABL_YB(yb); y += 4, vj += sj, vi += si) {
            const int fy = yl + y;
            const unsigned voff = (unsigned)(min(max(fy - a.pad, 0), a.H - 1) * a.W) * 4u + col_off;
            const unsigned lrow = (unsigned)(uintptr_t)(lptr_t)(win + y * kLdsStride);
            if (!(fabsf(vj) <= mask_hj && fabsf(vi) <= mask_hi)) continue;
            asm volatile("s_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[v], %[p0]"
                         :: [v] "v"(voff), [l] "s"(lrow), [p0] "s"(p0) : "memory");
          }
          asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
        }
        const bool any_zero = (xl < 0) || (yl < 0) || (xh > a.Wp - 1) || (yh > a.Hp - 1);
        if (any_zero && col_ok) {
#pragma unroll 1
          for (int y = wave; y < bh[t]; y += 4)
            if (!(col_inside && ((unsigned)(yl + y) < (unsigned)a.Hp))) win[y * kLdsStride + lane] = 0.0f;
        }
      }
    }
  }

  const int r = tid >> 3, q = tid & 7;
  // one tile: per-pixel setup (under the DMA for the first tile), gather, store.  INNER (block-uniform per tile): a whole tile
  // whose window lies inside the frame -- every neighbour is a frame pixel inside the window, the range checks and clamps of the
  // general form are identities and are not evaluated.
  auto do_tile = [&](const int t, auto inner_c) {
    constexpr bool INNER = decltype(inner_c)::value;
    const int i = wi0[t] + r, jb = wj0[t] + 4 * q;
    int pi = i, pj = jb;
    if (t == 0) asm volatile("" : "+v"(pi), "+v"(pj));   // the first tile's setup runs under the DMA
    int lidx[4], gx0[4], gy0[4];
    bool live[4];
    float w00[4], w01[4], w10[4], w11[4];
    const float yn = lin_m1_p1(a.top + pi, a.Hp, a.step_y);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#ifdef EQA_ABL_CHEAPSETUP  // ablation: the kernel without the per-pixel coordinate arithmetic (wrong pixels)
      lidx[k] = min(pi - wi0[t], bh[t] - 2) * kLdsStride + min(pj - wj0[t] + k, bw[t] - 2);
      gx0[k] = 0; gy0[k] = 0; live[k] = true;
      w00[k] = 0.25f; w01[k] = 0.25f; w10[k] = 0.25f; w11[k] = yn;
      continue;
#endif
      const float xn = lin_m1_p1(frame_x(pj + k), a.Wp, a.step_x);
      float ix, iy;
      sample_point(t0, t1, t2, t3, t4, t5, xn, yn, a.half_w, a.half_h, ix, iy);
      const float xf = floorf(ix), yf = floorf(iy);
      const float wx1 = ix - xf, wy1 = iy - yf;
      const float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
      int xi, yi, lx, ly;
      if (INNER) {
        live[k] = true;
        xi = (int)xf; yi = (int)yf;
        lx = xi - x_lo[t]; ly = yi - y_lo[t];
      } else {
        const bool xin = (xf >= -1.0f) && (xf <= (float)(a.Wp - 1));
        const bool yin = (yf >= -1.0f) && (yf <= (float)(a.Hp - 1));
        live[k] = xin && yin;
        xi = xin ? (int)xf : -1; yi = yin ? (int)yf : -1;
        lx = min(max(xi - x_lo[t], 0), bw[t] - 2); ly = min(max(yi - y_lo[t], 0), bh[t] - 2);
      }
      gx0[k] = xi;
      gy0[k] = yi;
#ifdef EQA_CHECK_WINDOW
      if (live[k] && pi < a.OH && pj + k < a.OW &&
          (xi < x_lo[t] || xi + 1 > x_lo[t] + bw[t] - 1 || yi < y_lo[t] || yi + 1 > y_lo[t] + bh[t] - 1)) __builtin_trap();
#endif
      lidx[k] = ly * kLdsStride + lx;
      w00[k] = wy0 * wx0;
      w01[k] = wy0 * wx1;
      w10[k] = wy1 * wx0;
      w11[k] = wy1 * wx1;
    }
    if (t == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA (every tile's) has landed in LDS
      __syncthreads();                                    // ... and so has every other wave's
    }
    float acc[4];
    if (use_lds[t]) {
      const float* s = smem + t * win_floats;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float nw = s[lidx[k]], ne = s[lidx[k] + 1];
        const float sw = s[lidx[k] + kLdsStride], se = s[lidx[k] + kLdsStride + 1];
        const float v = blend4(nw, ne, sw, se, w00[k], w01[k], w10[k], w11[k]);
        acc[k] = (INNER || live[k]) ? v : 0.0f;
      }
    } else {
      // window too large for LDS, or forced: rare, the same arithmetic from global memory
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = 0.0f;
#pragma unroll 1
      for (int k = 0; k < 4; ++k) {
        const int gx = k == 0 ? gx0[0] : k == 1 ? gx0[1] : k == 2 ? gx0[2] : gx0[3];
        const int gy = k == 0 ? gy0[0] : k == 1 ? gy0[1] : k == 2 ? gy0[2] : gy0[3];
        const float a00 = k == 0 ? w00[0] : k == 1 ? w00[1] : k == 2 ? w00[2] : w00[3];
        const float a01 = k == 0 ? w01[0] : k == 1 ? w01[1] : k == 2 ? w01[2] : w01[3];
        const float a10 = k == 0 ? w10[0] : k == 1 ? w10[1] : k == 2 ? w10[2] : w10[3];
        const float a11 = k == 0 ? w11[0] : k == 1 ? w11[1] : k == 2 ? w11[2] : w11[3];
        const bool lv = k == 0 ? live[0] : k == 1 ? live[1] : k == 2 ? live[2] : live[3];
        bool in00, in01, in10, in11;
        const int o00 = src_offset(gy, gx, in00), o01 = src_offset(gy, gx + 1, in01);
        const int o10 = src_offset(gy + 1, gx, in10), o11 = src_offset(gy + 1, gx + 1, in11);
        const float v00 = plane[o00], v01 = plane[o01], v10 = plane[o10], v11 = plane[o11];
        float v = blend4(in00 ? v00 : 0.0f, in01 ? v01 : 0.0f, in10 ? v10 : 0.0f, in11 ? v11 : 0.0f, a00, a01, a10, a11);
        v = lv ? v : 0.0f;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
          if (k2 == k) acc[k2] = v;
      }
    }
    if (i < a.OH) {
      float* o = dst_img + (unsigned)(i * a.OW + jb);
      if (VEC) {
        if (jb < a.OW && EQA_ABL_STORE_OK(acc[0])) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (jb + k < a.OW) o[k] = acc[k];
      }
    }
  };
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (bw[t] == 0) break;          // block-uniform: the tiles of a slot are consecutive
    if (inner[t]) do_tile(t, std::true_type());
    else do_tile(t, std::false_type());
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of the group action.  y[n,c,i,j] = sum_k w_k(phi) * frame[c, nbr_k(i,j; phi)]  (bilinear, 4 neighbours)
//   ANGLE: dL/dphi = sum gy * (dy/dix * dix/dphi + dy/diy * diy/dphi), with the source point rotating about the frame
//          centre c:  s = c + R(phi)^-1 (dst - c)  =>  ds/dphi = (-(s_y - c_y), s_x - c_x)  [per radian],
//          dy/dix = wy0 (ne - nw) + wy1 (se - sw),  dy/diy = wx0 (sw - nw) + wx1 (se - ne)
//          (what autograd derives through kornia's rotation-matrix -> affine_grid -> grid_sample chain,
//          discrete_group.py:213 / images/utils.py:57).  One partial per (output image, tile): deterministic.
//   INPUT: adjoint of the gather: scatter gy * w_k to the (clamped = replicate-pad adjoint, flipped, channel-mapped)
//          source pixels with hardware float atomics (same approach as torch's grid_sampler backward).
//   THETA (GRAD == 2): dL/dtheta[6] for a per-sample affine matrix (continuous groups: K.geometry.warp_affine in
//          images/canonicalization/continuous_group.py:203): ix = ((t0 xn + t1 yn + t2) + 1) half_w  =>
//          d ix / d(t0, t1, t2) = half_w (xn, yn, 1), likewise iy with half_h; six partials per (output image, tile).
// Same grid decomposition as the forward kernel; direct gathers (L1/L2), no LDS staging: correctness first.
template <int GRAD, bool INPUT>  // GRAD: 0 none, 1 rotation angle, 2 affine matrix
__global__ __launch_bounds__(kThreads) void group_action_bwd_kernel(const ActionArgs a) {
  constexpr bool ANGLE = GRAD != 0;  // needs the image gradient at the sample point
  constexpr int NS = GRAD == 2 ? 6 : 1;
  __shared__ float s_red[4][NS];
  const int tid = threadIdx.x;
  int n, tile_x, tile_y;
  if (!block_tile(a.n_out, (int)blockIdx.z, n, tile_x, tile_y)) return;
  const int j0 = tile_x * kTile, i0 = tile_y * kTile;
  int e, b;
  if (a.gidx) {
    e = a.gidx[n];
    b = n;
  } else {
    e = n / a.B;
    b = n - e * a.B;
  }
  e = min(max(e, 0), a.E - 1);
  const int fl = a.flags ? a.flags[e] : 0;
  const float* th = a.theta + e * 6;
  const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
  const bool flip_dst = (fl & EQA_FLIP_DST) != 0, flip_src = (fl & EQA_FLIP_SRC) != 0;
  const float cx = a.half_w, cy = a.half_h;  // frame centre ((Wp-1)/2, (Hp-1)/2)

  const int r = tid >> 3, q = tid & 7;
  const int i = i0 + r, jb = j0 + 4 * q;
  const bool row_ok = i < a.OH;
  int gx0[4], gy0[4];
  bool live[4];
  float wx1[4], wy1[4], armx[4], army[4], xns[4];
  const float yn = lin_m1_p1(a.top + i, a.Hp, a.step_y);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int fj = a.left + jb + k;
    const float xn = lin_m1_p1(flip_dst ? (a.Wp - 1 - fj) : fj, a.Wp, a.step_x);
    float ix, iy;
    sample_point(t0, t1, t2, t3, t4, t5, xn, yn, a.half_w, a.half_h, ix, iy);
    const float xf = floorf(ix), yf = floorf(iy);
    wx1[k] = ix - xf;
    wy1[k] = iy - yf;
    const bool xin = (xf >= -1.0f) && (xf <= (float)(a.Wp - 1));
    const bool yin = (yf >= -1.0f) && (yf <= (float)(a.Hp - 1));
    live[k] = xin && yin && row_ok && (jb + k < a.OW);
    gx0[k] = xin ? (int)xf : -1;
    gy0[k] = yin ? (int)yf : -1;
    armx[k] = -(iy - cy);
    army[k] = ix - cx;
    xns[k] = xn;
  }
  auto src_offset = [&](int fy, int fx, bool& inside) -> int {
    inside = ((unsigned)fx < (unsigned)a.Wp) && ((unsigned)fy < (unsigned)a.Hp);
    int sx = flip_src ? (a.Wp - 1 - fx) : fx;
    sx = min(max(sx - a.pad, 0), a.W - 1);
    const int sy = min(max(fy - a.pad, 0), a.H - 1);
    return sy * a.W + sx;
  };
  int off[4][4];
  bool in[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    off[k][0] = src_offset(gy0[k], gx0[k], in[k][0]);
    off[k][1] = src_offset(gy0[k], gx0[k] + 1, in[k][1]);
    off[k][2] = src_offset(gy0[k] + 1, gx0[k], in[k][2]);
    off[k][3] = src_offset(gy0[k] + 1, gx0[k] + 1, in[k][3]);
  }

  const unsigned src_plane = (unsigned)(a.H * a.W), dst_plane = (unsigned)(a.OH * a.OW);
  const size_t img_off = (size_t)b * ((size_t)a.C * src_plane);
  const float* const src_img = a.src + img_off;
  const float* const gout_img = a.gout + (size_t)n * ((size_t)a.C * dst_plane);
  float sum = 0.0f;
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sy[4] = {0.f, 0.f, 0.f, 0.f};  // GRAD == 2: per-pixel sums of g*dix, g*diy over channels
#pragma unroll 1
  for (int c = 0; c < a.C; ++c) {
    const int cs = a.chan_map ? (c / a.G) * a.G + a.chan_map[e * a.G + c % a.G] : c;
    const float* pl = src_img + (unsigned)cs * src_plane;
    const float* go = gout_img + (unsigned)c * dst_plane + (unsigned)(i * a.OW + jb);
    // (ANGLE without INPUT -- the training step's angle gradient -- requests a channel's 16 neighbours and 4 gradient values
    // together: the offsets are clamped into the plane, so the loads need no predicate, only the values a select; as conditional
    // loads behind `if (!live) continue` every one of them was waited for on the spot: 174 us per 256 images)
    float nbv[4][4], gv[4];
    if (ANGLE && !INPUT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gv[k] = gout_img[(unsigned)c * dst_plane + (live[k] ? (unsigned)(i * a.OW + jb + k) : 0u)];   // (a dead pixel reads the plane's first value)
#pragma unroll
        for (int t = 0; t < 4; ++t) nbv[k][t] = pl[off[k][t]];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!(ANGLE && !INPUT) && !live[k]) continue;
      const float g = (ANGLE && !INPUT) ? (live[k] ? gv[k] : 0.0f) : go[k];
      const float wx0 = 1.0f - wx1[k], wy0 = 1.0f - wy1[k];
      if (ANGLE) {
        float nw, ne, sw, se;
        if (!INPUT) {
          nw = in[k][0] ? nbv[k][0] : 0.0f; ne = in[k][1] ? nbv[k][1] : 0.0f;
          sw = in[k][2] ? nbv[k][2] : 0.0f; se = in[k][3] ? nbv[k][3] : 0.0f;
        } else {
          nw = in[k][0] ? pl[off[k][0]] : 0.0f; ne = in[k][1] ? pl[off[k][1]] : 0.0f;
          sw = in[k][2] ? pl[off[k][2]] : 0.0f; se = in[k][3] ? pl[off[k][3]] : 0.0f;
        }
        const float dix = wy0 * (ne - nw) + wy1[k] * (se - sw);
        const float diy = wx0 * (sw - nw) + wx1[k] * (se - ne);
        // (a dead pixel contributes nothing -- a select, not g = 0: a non-finite neighbour must not turn 0 * inf into NaN)
        if (GRAD == 1) sum += live[k] ? g * (dix * armx[k] + diy * army[k]) : 0.0f;
        else { sx[k] += live[k] ? g * dix : 0.0f; sy[k] += live[k] ? g * diy : 0.0f; }
      }
      if (INPUT) {
        float* gp = a.gsrc + img_off + (size_t)cs * src_plane;
        if (in[k][0]) unsafeAtomicAdd(gp + off[k][0], g * wy0 * wx0);
        if (in[k][1]) unsafeAtomicAdd(gp + off[k][1], g * wy0 * wx1[k]);
        if (in[k][2]) unsafeAtomicAdd(gp + off[k][2], g * wy1[k] * wx0);
        if (in[k][3]) unsafeAtomicAdd(gp + off[k][3], g * wy1[k] * wx1[k]);
      }
    }
  }
  if (ANGLE) {
    float v[NS];
    if (GRAD == 1) {
      v[0] = sum;
    } else {
#pragma unroll
      for (int m = 0; m < NS; ++m) v[m] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[0] += sx[k] * xns[k]; v[1] += sx[k] * yn; v[2] += sx[k];
        v[NS - 3] += sy[k] * xns[k]; v[NS - 2] += sy[k] * yn; v[NS - 1] += sy[k];
      }
#pragma unroll
      for (int m = 0; m < NS; ++m) v[m] *= (m < 3 ? a.half_w : a.half_h);
    }
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      const float w = wave_sum_f(v[m]);
      if ((tid & 63) == 0) s_red[tid >> 6][m] = w;
    }
    __syncthreads();
    if (tid < NS) {
      const int tiles_x = (int)(gridDim.x >> 3);
      const size_t t = ((size_t)n * gridDim.y + tile_y) * tiles_x + tile_x;
      a.partial[t * NS + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
    }
  }
}

// Input gradient as a GATHER (no atomics, deterministic, no pre-zeroed buffer) for the un-padded, one-output-per-image
// case -- the invert action I7 on image-shaped network outputs, which is what a segmentation loss differentiates through.
// A source pixel s receives g[o] * w(o, s) from the output pixels o whose sample point p(o) lies within one pixel of s;
// p is affine in o, so the candidates are the integer points of A^-1(s + (-1,1)^2): at most 3 x 3 for a rotation, 4 x 4
// slots here.  Each candidate's weight is recomputed exactly as the forward computes it (floor, fractional parts).
// Measured against the atomic scatter: 3.4 ms -> see HISTORY.md for 256 x 3 x 224 x 224.
template <bool MAPPED>
__global__ __launch_bounds__(kThreads) void group_action_bwd_gather_kernel(const ActionArgs a) {
  __shared__ int s_inv[kMaxMapG];
  const int b = blockIdx.z;
  const int e = min(max(a.gidx[b], 0), a.E - 1);
  if (MAPPED) {
    if (threadIdx.x < a.G) s_inv[a.chan_map[e * a.G + threadIdx.x]] = threadIdx.x;  // inverse of the channel permutation
    __syncthreads();
  }
  const int sx = blockIdx.x * 64 + (threadIdx.x & 63), sy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (sx >= a.W || sy >= a.H) return;
  const int fl = a.flags ? a.flags[e] : 0;
  const bool flip_dst = (fl & EQA_FLIP_DST) != 0, flip_src = (fl & EQA_FLIP_SRC) != 0;
  const float* th = a.theta + e * 6;
  const float t0 = th[0], t1 = th[1], t2 = th[2], t3 = th[3], t4 = th[4], t5 = th[5];
  const int fx = flip_src ? (a.Wp - 1 - sx) : sx, fy = sy;  // pad == 0: the frame is the source
  // (jf, i') -> sample point, as an affine map, and its inverse
  const float a00 = a.half_w * t0 * a.step_x, a01 = a.half_w * t1 * a.step_y, b0 = a.half_w * ((t2 - t0 - t1) + 1.0f);
  const float a10 = a.half_h * t3 * a.step_x, a11 = a.half_h * t4 * a.step_y, b1 = a.half_h * ((t5 - t3 - t4) + 1.0f);
  const float det = a00 * a11 - a01 * a10;
  const float m00 = a11 / det, m01 = -a01 / det, m10 = -a10 / det, m11 = a00 / det;
  const float px = (float)fx - b0, py = (float)fy - b1;
  const float jc = m00 * px + m01 * py, ic = m10 * px + m11 * py;
  const float dj = fabsf(m00) + fabsf(m01) + 0.05f, di = fabsf(m10) + fabsf(m11) + 0.05f;
  const int j_lo = max((int)ceilf(jc - dj), 0), j_hi = min((int)floorf(jc + dj), a.Wp - 1);
  const int i_lo = max((int)ceilf(ic - di), 0), i_hi = min((int)floorf(ic + di), a.Hp - 1);
  float w[16];
  int off[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ii = i_lo + r, jf = j_lo + c;
      float wt = 0.0f;
      int o = 0;
      const int i = ii - a.top, j = (flip_dst ? (a.Wp - 1 - jf) : jf) - a.left;
      if (ii <= i_hi && jf <= j_hi && i >= 0 && i < a.OH && j >= 0 && j < a.OW) {
        const float xn = lin_m1_p1(jf, a.Wp, a.step_x), yn = lin_m1_p1(ii, a.Hp, a.step_y);
        float ix, iy;
        sample_point(t0, t1, t2, t3, t4, t5, xn, yn, a.half_w, a.half_h, ix, iy);
        const float xf = floorf(ix), yf = floorf(iy);
        const float wx1 = ix - xf, wy1 = iy - yf;
        const float ffx = (float)fx, ffy = (float)fy;
        const float wx = (ffx == xf) ? 1.0f - wx1 : ((ffx == xf + 1.0f) ? wx1 : 0.0f);
        const float wy = (ffy == yf) ? 1.0f - wy1 : ((ffy == yf + 1.0f) ? wy1 : 0.0f);
        wt = wy * wx;
        o = i * a.OW + j;
      }
      w[r * 4 + c] = wt;
      off[r * 4 + c] = o;
    }
  }
  const size_t src_plane = (size_t)a.H * a.W, dst_plane = (size_t)a.OH * a.OW;
  const float* gimg = a.gout + (size_t)b * a.C * dst_plane;
  float* simg = a.gsrc + (size_t)b * a.C * src_plane + (size_t)sy * a.W + sx;
  for (int cs = 0; cs < a.C; ++cs) {
    const int c = MAPPED ? (cs / a.G) * a.G + s_inv[cs % a.G] : cs;
    const float* gp = gimg + (size_t)c * dst_plane;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (w[k] != 0.0f) acc += gp[off[k]] * w[k];
    simg[(size_t)cs * src_plane] = acc;
  }
}

// Adjoint of the replicate ("edge") padding: gsrc[sy][sx] = sum of gframe over the frame pixels that clamp to (sy, sx) --
// itself for interior pixels, a strip of pad+1 pixels on the borders, a (pad+1)^2 square in the corners.  Together with the
// gather above run on the padded frame as its source this is the deterministic input gradient of the canonicalizing
// transform I5 (pad -> [flip] -> rotate -> crop).  Separable: fold the columns of every frame row (the two border sums
// by a block reduction), then the rows of every column (one thread per column, coalesced over columns).
__global__ __launch_bounds__(kThreads) void fold_edge_pad_x_kernel(const float* __restrict__ gframe, float* __restrict__ tmp,
                                                                  int W, int pad) {
  __shared__ float s_red[2][kThreads / 64];
  const int Wp = W + 2 * pad;
  const size_t row = blockIdx.x;  // plane * Hp + fy
  const float* g = gframe + row * Wp;
  float* o = tmp + row * W;
  for (int sx = 1 + threadIdx.x; sx < W - 1; sx += kThreads) o[sx] = g[sx + pad];
  float l = 0.f, r = 0.f;  // left border: frame columns [0, pad]; right border: [pad + W - 1, Wp - 1]
  for (int k = threadIdx.x; k <= pad; k += kThreads) { l += g[k]; r += g[pad + W - 1 + k]; }
  l = wave_sum_f(l);
  r = wave_sum_f(r);
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = l; s_red[1][threadIdx.x >> 6] = r; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, c = 0.f;
    for (int w = 0; w < kThreads / 64; ++w) { a += s_red[0][w]; c += s_red[1][w]; }
    if (W == 1) o[0] = a + c - g[pad];  // both strips contain the single column once too often
    else { o[0] = a; o[W - 1] = c; }
  }
}

__global__ __launch_bounds__(kThreads) void fold_edge_pad_y_kernel(const float* __restrict__ tmp, float* __restrict__ gsrc,
                                                                  int H, int W, int pad) {
  const int sx = blockIdx.x * kThreads + threadIdx.x;
  if (sx >= W) return;
  const int Hp = H + 2 * pad;
  const size_t plane = blockIdx.y;
  const float* t = tmp + plane * (size_t)Hp * W + sx;
  float* o = gsrc + plane * (size_t)H * W + sx;
  float top = 0.f, bot = 0.f;
  for (int k = 0; k <= pad; ++k) { top += t[(size_t)k * W]; bot += t[(size_t)(pad + H - 1 + k) * W]; }
  if (H == 1) { o[0] = top + bot - t[(size_t)pad * W]; return; }
  o[0] = top;
  o[(size_t)(H - 1) * W] = bot;
  for (int sy = 1; sy < H - 1; ++sy) o[(size_t)sy * W] = t[(size_t)(sy + pad) * W];
}

template <int CH>
int launch_action_ch(const ActionArgs& a, bool vec, hipStream_t st) {
  const int tiles_x = (a.OW + kTile - 1) / kTile, tiles_y = (a.OH + kTile - 1) / kTile;
  const int groups = (a.n_out + kXcd - 1) / kXcd;
  if (tiles_y > 65535 || groups > 65535) return EQA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(kXcd * tiles_x), (unsigned)tiles_y, (unsigned)groups);
  const size_t lds = (size_t)CH * a.lds_rows * kLdsStride * sizeof(float) + (a.chan_map ? kMaxMapG * sizeof(int) : 0);
  if (vec)
    hipLaunchKernelGGL((group_action_kernel<CH, true>), grid, dim3(kThreads), lds, st, a);
  else
    hipLaunchKernelGGL((group_action_kernel<CH, false>), grid, dim3(kThreads), lds, st, a);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

template <int NT>
int launch_action_c1_nt(const ActionArgs& a, bool vec, hipStream_t st) {
  const int tiles_x = (a.OW + kTile - 1) / kTile, tiles_y = (a.OH + kTile - 1) / kTile;
  const long long slots = ((long long)tiles_x * tiles_y + NT - 1) / NT;
  const int groups = (a.n_out + kXcd - 1) / kXcd;
  if (slots * kXcd > 0x7fffffffLL || groups > 65535) return EQA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(kXcd * slots), 1u, (unsigned)groups);
  const size_t lds = (size_t)NT * a.lds_rows * kLdsStride * sizeof(float);
  if (vec)
    hipLaunchKernelGGL((group_action_c1_kernel<NT, true>), grid, dim3(kThreads), lds, st, a, tiles_x, tiles_y);
  else
    hipLaunchKernelGGL((group_action_c1_kernel<NT, false>), grid, dim3(kThreads), lds, st, a, tiles_x, tiles_y);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

// one-channel maps: NT tiles per block where the map has enough tiles to fill the chip that way
int launch_action_c1(const ActionArgs& a, bool vec, hipStream_t st) {
  const long long tiles = (long long)((a.OW + kTile - 1) / kTile) * ((a.OH + kTile - 1) / kTile);
  if (tiles * a.n_out < 4096) return EQA_ERR_UNSUPPORTED;      // small jobs: a tile per block fills the CUs better
  if (g_c1_tiles >= 4) return launch_action_c1_nt<4>(a, vec, st);
  return launch_action_c1_nt<2>(a, vec, st);
}

int fill_action_args(ActionArgs& a, const float* src, float* dst, const int32_t* gidx, const float* theta,
                     const int32_t* flags, const int32_t* chan_map, int E, int G, int n_out, int B, int C, int H, int W,
                     int pad, int OH, int OW, int top, int left) {
  if (n_out == 0 && B >= 0) {  // empty batch: nothing to validate against (empty tensors have null data pointers)
    a.n_out = 0;
    return EQA_OK;
  }
  if (!src || !theta || E <= 0 || n_out < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || pad < 0 || OH <= 0 || OW <= 0 ||
      top < 0 || left < 0)
    return EQA_ERR_INVALID_ARG;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  if (Hp < 2 || Wp < 2 || top + OH > Hp || left + OW > Wp) return EQA_ERR_INVALID_ARG;
  if (chan_map && (G <= 0 || C % G != 0)) return EQA_ERR_INVALID_ARG;
  if (chan_map && G > kMaxMapG) return EQA_ERR_UNSUPPORTED;
  if ((long long)C * H * W >= (1LL << 30) || (long long)C * OH * OW >= (1LL << 30)) return EQA_ERR_UNSUPPORTED;  // 32-bit offsets inside one image
  a.src = src; a.dst = dst; a.gidx = gidx; a.theta = theta; a.flags = flags; a.chan_map = chan_map;
  a.E = E; a.G = chan_map ? G : 1; a.n_out = n_out; a.B = B; a.C = C;
  a.H = H; a.W = W; a.pad = pad; a.Hp = Hp; a.Wp = Wp;
  a.OH = OH; a.OW = OW; a.top = top; a.left = left;
  a.half_w = (float)(Wp - 1) / 2.0f;
  a.half_h = (float)(Hp - 1) / 2.0f;
  a.step_x = 2.0f / (float)(Wp - 1);
  a.step_y = 2.0f / (float)(Hp - 1);
  a.force_direct = g_force_direct;
  a.lds_rows = kBox;
  a.gout = nullptr; a.gsrc = nullptr; a.partial = nullptr;
  return EQA_OK;
}

int launch_action(const float* src, float* dst, const int32_t* gidx, const float* theta, const int32_t* flags,
                  const int32_t* chan_map, int E, int G, int n_out, int B, int C, int H, int W, int pad, int OH,
                  int OW, int top, int left, void* stream, int max_window = 0) {
  if (!dst && n_out != 0) return EQA_ERR_INVALID_ARG;
  ActionArgs a;
  const int rc = fill_action_args(a, src, dst, gidx, theta, flags, chan_map, E, G, n_out, B, C, H, W, pad, OH, OW, top, left);
  if (rc != EQA_OK) return rc;
  if (n_out == 0) return EQA_OK;
  // the caller's bound on a tile's source window (right-angle elements: 35 rows instead of 47 -> 19.7 KB of LDS per block at three
  // channels, 8 blocks per CU instead of 6); a window that turns out larger takes the direct path: slow, never wrong
  if (max_window > 0) a.lds_rows = std::min(std::max(max_window, 2), kBox);
  const bool vec = (OW % 4 == 0) && (((uintptr_t)dst & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
#if EQA_FORCE_CH
  return launch_action_ch<EQA_FORCE_CH>(a, vec, st);
#else
  if (C == 1 && !chan_map && g_c1_tiles > 0) {
    const int rc1 = launch_action_c1(a, vec, st);
    if (rc1 != EQA_ERR_UNSUPPORTED) return rc1;
  }
  if (C % 3 == 0) return launch_action_ch<3>(a, vec, st);
  if (C % 2 == 0) return launch_action_ch<2>(a, vec, st);
  return launch_action_ch<1>(a, vec, st);
#endif
}

template <int CH>
int launch_pair_ch(const ActionArgs& a0, const ActionArgs& a1, bool vec, hipStream_t st) {
  const int tiles_x = (a0.OW + kTile - 1) / kTile, tiles_y = (a0.OH + kTile - 1) / kTile;
  const int g0 = (a0.n_out + kXcd - 1) / kXcd, g1 = (a1.n_out + kXcd - 1) / kXcd;
  if (tiles_y > 65535 || g0 + g1 > 65535) return EQA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(kXcd * tiles_x), (unsigned)tiles_y, (unsigned)(g0 + g1));
  const size_t lds = (size_t)CH * kBox * kLdsStride * sizeof(float) + ((a0.chan_map || a1.chan_map) ? kMaxMapG * sizeof(int) : 0);
  if (vec)
    hipLaunchKernelGGL((group_action_pair_kernel<CH, true>), grid, dim3(kThreads), lds, st, a0, a1, g0);
  else
    hipLaunchKernelGGL((group_action_pair_kernel<CH, false>), grid, dim3(kThreads), lds, st, a0, a1, g0);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

inline int action_ch(int C) { return C % 3 == 0 ? 3 : (C % 2 == 0 ? 2 : 1); }

// canonicalize x (edge-padded frame, crop back) and invert f (whole frame, optional regular-representation roll) for the same
// per-image group index.  One launch when the two jobs share the tile grid, the channel staging width and the store width
// (always the case for x, f of the same H x W with 3 | C or equal parity); two launches otherwise -- same results either way.
int launch_pair(const float* x, float* y, const float* theta_c, const int32_t* flags_c, int pad, int C, const float* f, float* out,
                const float* theta_i, const int32_t* flags_i, const int32_t* chan_map, int G, int Cf, const int32_t* gidx, int E,
                int B, int H, int W, void* stream) {
  if (B == 0) return EQA_OK;
  if (!gidx || !y || !out) return EQA_ERR_INVALID_ARG;
  ActionArgs a0, a1;
  int rc = fill_action_args(a0, x, y, gidx, theta_c, flags_c, nullptr, E, 1, B, B, C, H, W, pad, H, W, pad, pad);
  if (rc != EQA_OK) return rc;
  rc = fill_action_args(a1, f, out, gidx, theta_i, flags_i, chan_map, E, G, B, B, Cf, H, W, 0, H, W, 0, 0);
  if (rc != EQA_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const bool v0 = (W % 4 == 0) && (((uintptr_t)y & 15) == 0), v1 = (W % 4 == 0) && (((uintptr_t)out & 15) == 0);
  const int ch0 = EQA_FORCE_CH ? EQA_FORCE_CH : action_ch(C), ch1 = EQA_FORCE_CH ? EQA_FORCE_CH : action_ch(Cf);
  if (ch0 == ch1 && v0 == v1) {
    if (ch0 == 3) return launch_pair_ch<3>(a0, a1, v0, st);
    if (ch0 == 2) return launch_pair_ch<2>(a0, a1, v0, st);
    return launch_pair_ch<1>(a0, a1, v0, st);
  }
  rc = launch_action(x, y, gidx, theta_c, flags_c, nullptr, E, 1, B, B, C, H, W, pad, H, W, pad, pad, stream);
  if (rc != EQA_OK) return rc;
  return launch_action(f, out, gidx, theta_i, flags_i, chan_map, E, G, B, B, Cf, H, W, 0, H, W, 0, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// I1: centre crop + antialiased bilinear resize (torchvision CenterCrop + Resize on a tensor ==
// F.interpolate(bilinear, antialias=True, align_corners=False); discrete_group.py:174-188).
// Separable like torch's kernel and in the same order: horizontal pass (fp32 intermediates), then vertical pass.
// The per-output-index tap ranges and normalised triangle weights are built on the host with torch's own formula
// (UpSampleKernel.cpp _compute_indices_min_size_weights_aa) and passed as small tables; the crop is folded into the
// tap start indices.  One block = one (image, channel) plane x a band of kAaBand output rows; the band's horizontally
// resampled input rows live in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int kAaBand = 8;  // = the `band` the host tables are built for (geometry.aa_resize_tables); 16 / 32 measured slower
#ifndef EQA_AA_WIDE_MIN_K
#define EQA_AA_WIDE_MIN_K 8  // filters wider than this take the LDS row-staged kernel
#endif
constexpr int kAaMaxK = 20;  // taps kept in registers by the wide-filter kernel (K = 17 at 8x down-sampling)

__global__ __launch_bounds__(kThreads) void crop_resize_aa_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 const float* __restrict__ wx, const int32_t* __restrict__ x0,
                                                                 const float* __restrict__ wy, const int32_t* __restrict__ y0,
                                                                 int H, int W, int OH, int OW, int K, int max_rows) {
  extern __shared__ __attribute__((aligned(16))) float aa_tmp[];  // [max_rows][OW]
  const int plane = blockIdx.y;
  const int r0 = blockIdx.x * kAaBand, r1 = min(r0 + kAaBand, OH);
  const int ybeg = y0[r0];
  const int yend = min(y0[r1 - 1] + K, H);  // taps past a row's own range carry zero weight
  const int nrows = min(yend - ybeg, max_rows);
  const float* src = x + (size_t)plane * H * W;
  // K <= EQA_AA_WIDE_MIN_K here.  A thread keeps ONE output column (kThreads / OW rows are worked on at a time, the
  // threads beyond that idle): tap start and weights are loaded once per block instead of once per value, no division per
  // value, and the K loads of a value go out together (unrolled with a predicate).  Measured at 256 x 3 x 224^2 -> 96^2:
  // 123 us with one (row, column) pair per thread and trip, of which 100 us were this pass.
  const int rows_par = kThreads / OW;
  if (rows_par >= 1) {
    const int ox = threadIdx.x % OW, rsub = threadIdx.x / OW;
    if (rsub < rows_par) {
      const int xs = x0[ox];
      float wv[EQA_AA_WIDE_MIN_K];
      int xo[EQA_AA_WIDE_MIN_K];
#pragma unroll
      for (int j = 0; j < EQA_AA_WIDE_MIN_K; ++j) {
        wv[j] = j < K ? wx[ox * K + j] : 0.0f;
        xo[j] = min(xs + j, W - 1);
      }
      // (keeping four row trips' loads in flight at once was tried: 84 -> 92 us, the extra registers cost more occupancy than the
      // shorter dependency chain gains)
      for (int ry = rsub; ry < nrows; ry += rows_par) {
        const float* row = src + (size_t)(ybeg + ry) * W;
        float xv[EQA_AA_WIDE_MIN_K];
#pragma unroll
        for (int j = 0; j < EQA_AA_WIDE_MIN_K; ++j) xv[j] = j < K ? row[xo[j]] : 0.0f;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < EQA_AA_WIDE_MIN_K; ++j)
          if (j < K) acc += wv[j] * xv[j];
        aa_tmp[ry * OW + ox] = acc;
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < nrows * OW; idx += kThreads) {
      const int ry = idx / OW, ox = idx - ry * OW;
      const float* row = src + (size_t)(ybeg + ry) * W;
      const int xs = x0[ox];
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < EQA_AA_WIDE_MIN_K; ++j)
        if (j < K) acc += wx[ox * K + j] * row[min(xs + j, W - 1)];
      aa_tmp[ry * OW + ox] = acc;
    }
  }
  __syncthreads();
  float* dst = y + (size_t)plane * OH * OW;
  for (int idx = threadIdx.x; idx < (r1 - r0) * OW; idx += kThreads) {
    const int r = idx / OW, ox = idx - r * OW;
    const int oy = r0 + r;

    const int ys = y0[oy] - ybeg;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < EQA_AA_WIDE_MIN_K; ++j)
      if (j < K) acc += wy[oy * K + j] * aa_tmp[min(ys + j, nrows - 1) * OW + ox];
    dst[(size_t)oy * OW + ox] = acc;
  }
}

// Narrow filters over 16-byte aligned rows (the headline's 224 -> crop 180 -> 96, K = 5), round 3.  The kernel above gathers its K
// taps from global memory (two input rows per trip, one dependent round trip per trip) and chains table load -> address -> data
// load: ~11 us per block whatever its size, 1.65 TB/s.  Here a block keeps ONE band of kAaBand output rows and walks over planes
// (the tables of a band are the same for every plane: loaded once), stages the band's input rows in LDS with 16-byte loads of the
// aligned column window -- the NEXT plane's rows are requested (registers) before this plane's two passes run from LDS, so the
// HBM round trip hides behind the LDS work -- and writes the band.

template <int K, int BAND, int NL>   // K: taps per output index; BAND: output rows per block; NL: 16-byte loads per thread and plane (a template argument: no branch per tap, and the wait counts stay exact)
__global__ __launch_bounds__(kThreads, (NL > 5 && K > 5) ? 2 : 4) void crop_resize_aa_staged_kernel(   // (wide prefetch + many taps: 128 registers spill)
    const float* __restrict__ x, float* __restrict__ y,
                                                                        const float* __restrict__ wx, const int32_t* __restrict__ x0,
                                                                        const float* __restrict__ wy, const int32_t* __restrict__ y0,
                                                                        int planes, int H, int W, int OH, int OW, int cap_rows,
                                                                        int xb, int xl) {
  extern __shared__ __attribute__((aligned(16))) float aa_tmp[];  // rows [cap_rows][xl], the horizontal pass [cap_rows][OW], tables
  float* rows = aa_tmp;
  float* tmp = aa_tmp + (size_t)cap_rows * xl;
  float* tabw = tmp + (size_t)cap_rows * OW;                              // [BAND][K] vertical weights of the band
  int* taby = reinterpret_cast<int*>(tabw + BAND * K);  // [BAND] first input row of each output row
  // grid (8, bands, plane groups): blockIdx.x is the XCD the dispatcher deals the block to (x is the fastest grid axis and 8 wide),
  // so the bands of one plane -- whose input rows overlap by K - 1 and share the cache lines at the window's edges -- are worked on
  // by blocks of ONE XCD at about the same time and meet in its L2 (round 3: band b of every plane on XCD b % 8, the overlap rows
  // fetched from HBM twice: 1.38 x the algorithmic bytes)
  const int r0 = blockIdx.y * BAND, r1 = min(r0 + BAND, OH);
  const int nband = r1 - r0;
  const int plane0 = (int)(blockIdx.z * kXcd + blockIdx.x), plane_step = (int)(gridDim.z * kXcd);
  const int rows_par = kThreads / OW;
  const int ox = rows_par >= 1 ? threadIdx.x % OW : 0, rsub = rows_par >= 1 ? threadIdx.x / OW : 0;
  const int xs_g = x0[ox];
  float wv[K];
  int xo[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    wv[j] = wx[ox * K + j];
    xo[j] = min(xs_g - xb + j, xl - 1);
  }
  if ((int)threadIdx.x < nband * K) tabw[threadIdx.x] = wy[r0 * K + threadIdx.x];
  const int ybeg = y0[r0];
  if ((int)threadIdx.x < nband) taby[threadIdx.x] = y0[r0 + threadIdx.x] - ybeg;
  const int yend = min(y0[r1 - 1] + K, H);
  const int nrows = min(yend - ybeg, cap_rows);
  const int nq = xl >> 2;
  const int tq = threadIdx.x & 63, tr = threadIdx.x >> 6;
  const bool prefetch = nq <= 64 && nrows <= 4 * NL;   // uniform: one 16-byte load per (thread, row group member)
  const size_t plane_sz = (size_t)H * W;
  const float* src0 = x + (size_t)ybeg * W + xb;
  typedef float aa_f4 __attribute__((ext_vector_type(4)));
  aa_f4 v[NL];
  // (a macro, not a lambda: called from two places the lambda is not inlined and v[] goes to scratch)
#define EQA_AA_PF_LOAD(plane_)                                                                                          \
  do {                                                                                                                  \
    const float* src_ = src0 + (size_t)(plane_) * plane_sz;                                                             \
    _Pragma("unroll") for (int k = 0; k < NL; ++k)                                                           \
      v[k] = *reinterpret_cast<const aa_f4*>(src_ + (size_t)min(tr + 4 * k, nrows - 1) * W + 4 * min(tq, nq - 1));     \
  } while (0)
  // the tables have arrived before the first row is requested: from here on only row loads are ever outstanding, and the waits
  // the compiler places inside the passes are for those it names (a pending table load made them vmcnt(0): the prefetch drained)
  __builtin_amdgcn_s_waitcnt(0x0070);
  if (prefetch && plane0 < planes) EQA_AA_PF_LOAD(plane0);
  for (int plane = plane0; plane < planes; plane += plane_step) {
    if (prefetch) {
      if (tq < nq) {
#pragma unroll
        for (int k = 0; k < NL; ++k)
          if (tr + 4 * k < nrows) *reinterpret_cast<aa_f4*>(rows + (tr + 4 * k) * xl + 4 * tq) = v[k];
      }
      if (plane + plane_step < planes) EQA_AA_PF_LOAD(plane + plane_step);
    } else {
      const float* src = src0 + (size_t)plane * plane_sz;
      for (int q = tq; q < nq; q += 64)
        for (int rb = tr; rb < nrows; rb += 4) *reinterpret_cast<aa_f4*>(rows + rb * xl + 4 * q) = *reinterpret_cast<const aa_f4*>(src + (size_t)rb * W + 4 * q);
    }
    __syncthreads();
    // horizontal pass (fp32 intermediates, as torch's kernel): a thread keeps one output column -- tap starts and weights in registers
    if (rows_par >= 1) {
      if (rsub < rows_par) {
        for (int ry0 = rsub; ry0 < nrows; ry0 += 4 * rows_par) {   // four rows' taps in flight at once (LDS latency, not bandwidth)
          float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float* row = rows + min(ry0 + u * rows_par, nrows - 1) * xl;
#pragma unroll
            for (int j = 0; j < K; ++j)
              acc[u] += wv[j] * row[xo[j]];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (ry0 + u * rows_par < nrows) tmp[(ry0 + u * rows_par) * OW + ox] = acc[u];
        }
      }
    } else {
      for (int idx = threadIdx.x; idx < nrows * OW; idx += kThreads) {
        const int ry = idx / OW, oxx = idx - ry * OW;
        const float* row = rows + ry * xl;
        const int xs = x0[oxx] - xb;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j)
          acc += wx[oxx * K + j] * row[min(xs + j, xl - 1)];
        tmp[ry * OW + oxx] = acc;
      }
    }
    __syncthreads();
    float* dst = y + (size_t)plane * OH * OW;
    for (int idx = threadIdx.x; idx < nband * OW; idx += kThreads) {
      const int r = idx / OW, oxx = idx - r * OW;
      const int ys = taby[r];
      float acc = 0.0f;
#pragma unroll
      for (int j = 0; j < K; ++j)
        acc += tabw[r * K + j] * tmp[min(ys + j, nrows - 1) * OW + oxx];
      dst[(size_t)(r0 + r) * OW + oxx] = acc;
    }
    __syncthreads();   // the next plane's horizontal pass overwrites tmp
  }
#undef EQA_AA_PF_LOAD
}

// Wide filters (K > 8, i.e. down-sampling by more than ~3.5x: config 5 resizes 1024 -> 128 with 17 taps): the K strided
// global loads per intermediate value of the kernel above become the bottleneck (0.73 ms for 32 x 3 x 1024^2, 9x its HBM
// time).  Here every needed input row segment is first staged into LDS with coalesced loads, `rpi` rows per iteration,
// and the taps are taken from LDS.  Neighbouring lanes read addresses ~scale apart; for an even integer stride s = 2^a m
// (m odd) the row is stored with one pad float every 2^a elements, which makes the lane stride s + m odd (conflict-free).
__global__ __launch_bounds__(kThreads) void crop_resize_aa_wide_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                      const float* __restrict__ wx, const int32_t* __restrict__ x0,
                                                                      const float* __restrict__ wy, const int32_t* __restrict__ y0,
                                                                      int H, int W, int OH, int OW, int K, int max_rows, int xbeg,
                                                                      int xlen, int pad_shift, int row_stride, int rpi) {
  extern __shared__ __attribute__((aligned(16))) float aa_tmp[];  // [max_rows][OW] then [rpi][row_stride]
  float* rowbuf = aa_tmp + (size_t)max_rows * OW;
  const int plane = blockIdx.y;
  const int r0 = blockIdx.x * kAaBand, r1 = min(r0 + kAaBand, OH);
  const int ybeg = y0[r0];
  const int yend = min(y0[r1 - 1] + K, H);
  const int nrows = min(yend - ybeg, max_rows);
  const float* src = x + (size_t)plane * H * W;
  auto pos = [&](int e) { return pad_shift ? e + (e >> pad_shift) : e; };
  const bool fixed_col = (kThreads % OW) == 0 && K <= kAaMaxK;
  // whole float4s of 16-byte aligned rows (uniform): the staging below then loads 16 bytes per lane
  const bool vec_stage = rpi <= 8 && (xlen & 3) == 0 && (xbeg & 3) == 0 && (W & 3) == 0 && ((((uintptr_t)src) & 15) == 0);
  const int ox_fixed = threadIdx.x % OW;
  const int xs_fixed = x0[ox_fixed] - xbeg;
  float wreg[kAaMaxK];
#pragma unroll
  for (int j = 0; j < kAaMaxK; ++j) wreg[j] = (fixed_col && j < K) ? wx[ox_fixed * K + j] : 0.0f;
  // Rows no wider than one float4 per thread (1024 floats: config 5): the NEXT iteration's rows are requested before this
  // iteration's taps are taken, so the HBM round trip of an iteration hides behind the previous one's LDS work (round 3; a block
  // runs ~10 iterations and only two blocks fit a CU, so each exposed round trip was paid in full: 171 -> see DESIGN 3.7).
  const bool prefetch = vec_stage && xlen <= 4 * kThreads;
  const int e_pf = 4 * threadIdx.x;
  float4 pf[8];
  auto pf_load = [&](int ry0) {
    const int nr = min(rpi, nrows - ry0);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
      pf[rr] = *reinterpret_cast<const float4*>(src + (size_t)(ybeg + ry0 + min(rr, nr - 1)) * W + xbeg + min(e_pf, xlen - 4));
  };
  if (prefetch && nrows > 0) pf_load(0);
  for (int ry0 = 0; ry0 < nrows; ry0 += rpi) {
    const int nr = min(rpi, nrows - ry0);
    if (prefetch) {
      if (e_pf < xlen) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          if (rr < nr) {
            float* lrow = rowbuf + rr * row_stride;
            lrow[pos(e_pf)] = pf[rr].x; lrow[pos(e_pf + 1)] = pf[rr].y; lrow[pos(e_pf + 2)] = pf[rr].z; lrow[pos(e_pf + 3)] = pf[rr].w;
          }
        }
      }
      if (ry0 + rpi < nrows) pf_load(ry0 + rpi);
    } else if (vec_stage) {
      // 16-byte loads, one per (row, thread) and trip, ALL rows' loads in flight before the first LDS store: the rolled
      // load -> store loop paid one HBM round trip per row and 256 floats (32 trips per iteration of 8 rows of 1024)
      for (int e = 4 * threadIdx.x; e < xlen; e += 4 * kThreads) {
        float4 v[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr)
          v[rr] = *reinterpret_cast<const float4*>(src + (size_t)(ybeg + ry0 + min(rr, nr - 1)) * W + xbeg + e);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          if (rr < nr) {
            float* lrow = rowbuf + rr * row_stride;
            lrow[pos(e)] = v[rr].x; lrow[pos(e + 1)] = v[rr].y; lrow[pos(e + 2)] = v[rr].z; lrow[pos(e + 3)] = v[rr].w;
          }
        }
      }
    } else {
      for (int rr = 0; rr < nr; ++rr) {
        const float* grow = src + (size_t)(ybeg + ry0 + rr) * W + xbeg;
        float* lrow = rowbuf + rr * row_stride;
        for (int e = threadIdx.x; e < xlen; e += kThreads) lrow[pos(e)] = grow[e];  // xbeg + xlen <= W
      }
    }
    __syncthreads();
    if (fixed_col) {  // kThreads % OW == 0: the thread keeps its output column, weights and tap start live in registers
      for (int rr = threadIdx.x / OW; rr < nr; rr += kThreads / OW) {
        const float* row = rowbuf + rr * row_stride;
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < kAaMaxK; ++j)
          if (j < K) acc += wreg[j] * row[pos(min(xs_fixed + j, xlen - 1))];
        aa_tmp[(ry0 + rr) * OW + ox_fixed] = acc;
      }
    } else {
      for (int idx = threadIdx.x; idx < nr * OW; idx += kThreads) {
        const int rr = idx / OW, ox = idx - rr * OW;
        const float* row = rowbuf + rr * row_stride;
        const int xs = x0[ox] - xbeg;
        float acc = 0.0f;
        for (int j = 0; j < K; ++j) acc += wx[ox * K + j] * row[pos(min(xs + j, xlen - 1))];
        aa_tmp[(ry0 + rr) * OW + ox] = acc;
      }
    }
    __syncthreads();
  }
  float* dst = y + (size_t)plane * OH * OW;
  for (int idx = threadIdx.x; idx < (r1 - r0) * OW; idx += kThreads) {
    const int r = idx / OW, ox = idx - r * OW;
    const int oy = r0 + r;
    const int ys = y0[oy] - ybeg;
    float acc = 0.0f;
    for (int j = 0; j < K; ++j) acc += wy[oy * K + j] * aa_tmp[min(ys + j, nrows - 1) * OW + ox];
    dst[(size_t)oy * OW + ox] = acc;
  }
}

// The same for rows no wider than one float4 per thread (x_span <= 1024: config 5), VERTICAL pass first and without staging the
// input: the row-staged kernel above keeps one block per CU (74 KB intermediate band + 37 KB of staged rows) and 32 KB of loads in
// flight between two barriers per eight input rows -- 155 us for 32 x 3 x 1024^2 (2.6 TB/s).  Here a thread owns four columns:
// every input row of the band's span is loaded ONCE, 16 bytes per lane, eight rows ahead, and added to the band's eight output
// rows with its (uniform) vertical weight -- zero outside a row's K taps, so any overlap of the windows is handled -- and only the
// eight finished 1024-wide rows go through LDS (four at a time, 18 KB) for the horizontal taps.  Four blocks per CU, 128 KB of loads
// in flight per CU, four barriers per block.  (Summation order: vertical taps first; the staged kernels sum the horizontal taps first.)
constexpr int kAaStreamRows = 8;
__global__ __launch_bounds__(kThreads) void crop_resize_aa_stream_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                        const float* __restrict__ wx, const int32_t* __restrict__ x0,
                                                                        const float* __restrict__ wy, const int32_t* __restrict__ y0,
                                                                        int H, int W, int OH, int OW, int K, int xbeg, int xlen,
                                                                        int pad_shift, int row_stride) {
  extern __shared__ __attribute__((aligned(16))) float aa_rows[];   // [kAaBand / 2][row_stride]
  const int plane = blockIdx.y;
  const int r0 = blockIdx.x * kAaBand, nr_out = min(kAaBand, OH - r0);
  int ys[kAaBand];                                                   // first input row of each output row (uniform)
#pragma unroll
  for (int r = 0; r < kAaBand; ++r) ys[r] = y0[r0 + min(r, nr_out - 1)];
  const int ybeg = ys[0];
  const int span = ys[kAaBand - 1] + K - ybeg;                       // input rows the band touches (ys is non-decreasing)
  const float* src = x + (size_t)plane * H * W + xbeg;
  const int e = 4 * (int)threadIdx.x;
  const int e_ld = min(e, xlen - 4);
  auto pos = [&](int i) { return pad_shift ? i + (i >> pad_shift) : i; };
  float4 acc[kAaBand];
#pragma unroll
  for (int r = 0; r < kAaBand; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load8 = [&](float4 (&v)[kAaStreamRows], int yc) {           // rows ybeg + yc .. + 7 (clamped to the image: their weights are 0)
#pragma unroll
    for (int q = 0; q < kAaStreamRows; ++q)
      v[q] = *reinterpret_cast<const float4*>(src + (size_t)min(ybeg + yc + q, H - 1) * W + e_ld);
  };
  // the 8 x 8 vertical weights of a trip (input row q of the trip, output row r): lane 8 q + r loads its one weight, the others
  // get it by v_readlane (64 scalar loads in a chain, one per weight, cost 13 k cycles a trip: every one waited out its latency)
  const int lane = threadIdx.x & 63;
  const int wq = lane >> 3, wr = lane & 7;
  const int ys_w = y0[r0 + min(wr, nr_out - 1)];
  auto wload = [&](int yc) {
    const int t = ybeg + yc + wq - ys_w;
    const bool ok = wr < nr_out && t >= 0 && t < K && yc + wq < span;
    return ok ? wy[(size_t)(r0 + wr) * K + min(max(t, 0), K - 1)] : 0.0f;
  };
  auto add8 = [&](const float4 (&v)[kAaStreamRows], float wv) {
#pragma unroll
    for (int q = 0; q < kAaStreamRows; ++q) {
#pragma unroll
      for (int r = 0; r < kAaBand; ++r) {
        const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), 8 * q + r));
        acc[r].x += w * v[q].x; acc[r].y += w * v[q].y; acc[r].z += w * v[q].z; acc[r].w += w * v[q].w;
      }
    }
  };
  static_assert(kAaStreamRows == 8 && kAaBand == 8, "one weight per lane: 8 rows of a trip x 8 output rows = 64 lanes");
  float4 va[kAaStreamRows], vb[kAaStreamRows];
  float wa, wb = 0.0f;
  load8(va, 0);
  wa = wload(0);
  for (int yc = 0; yc < span; yc += 2 * kAaStreamRows) {
    if (yc + kAaStreamRows < span) { load8(vb, yc + kAaStreamRows); wb = wload(yc + kAaStreamRows); }
    add8(va, wa);
    if (yc + 2 * kAaStreamRows < span) { load8(va, yc + 2 * kAaStreamRows); wa = wload(yc + 2 * kAaStreamRows); }
    if (yc + kAaStreamRows < span) add8(vb, wb);
  }
  float* dst = y + (size_t)plane * OH * OW;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (e < xlen) {
#pragma unroll
      for (int rr = 0; rr < kAaBand / 2; ++rr) {
        float* lrow = aa_rows + rr * row_stride;
        const float4 v = acc[half * (kAaBand / 2) + rr];
        lrow[pos(e)] = v.x; lrow[pos(e + 1)] = v.y; lrow[pos(e + 2)] = v.z; lrow[pos(e + 3)] = v.w;
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < (kAaBand / 2) * OW; idx += kThreads) {
      const int rr = idx / OW, ox = idx - rr * OW;
      const int r = half * (kAaBand / 2) + rr;
      if (r < nr_out) {
        const float* row = aa_rows + rr * row_stride;
        const int xs = x0[ox] - xbeg;
        float a = 0.0f;
        for (int j = 0; j < K; ++j) a += wx[ox * K + j] * row[pos(min(xs + j, xlen - 1))];
        dst[(size_t)(r0 + r) * OW + ox] = a;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// I6: nearest-neighbour action on uint8 masks (torchvision.transforms.functional.rotate defaults on a uint8 tensor:
// half-pixel base grid, theta rescaled by (0.5 W, 0.5 H), grid_sample(nearest, zeros, align_corners=False), round;
// images/utils.py:125-136, optionally after flip_masks :112-122).  rtheta[e] = the RESCALED 3x2 matrix in the order
// (r00, r10, r20, r01, r11, r21): gx = xb*r00 + yb*r10 + r20, gy = xb*r01 + yb*r11 + r21.
// One thread = 4 consecutive output pixels (one 32-bit store).
// ------------------------------------------------------------------------------------------------
// Generic form (T = uint8 masks or fp32 images): output plane p of (n_planes) samples source plane p % src_mod with
// element eidx[p]; the sampling frame is the source plane edge-padded by `pad`, the output the (OH,OW) window at
// (top,left) of the frame -- GroupInference's pad(0.4 H) -> [hflip] -> rotate(+deg) -> CenterCrop on float images
// (examples/images/classification/inference_utils.py:100-123: torchvision rotate defaults to NEAREST) uses all of it.
template <typename T>
struct Pack4;
template <>
struct Pack4<uint8_t> {
  typedef uint32_t type;
  static __device__ __forceinline__ type make(const uint8_t (&v)[4]) {
    return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
  }
};
template <>
struct Pack4<float> {
  typedef float4 type;
  static __device__ __forceinline__ type make(const float (&v)[4]) { return make_float4(v[0], v[1], v[2], v[3]); }
};

template <typename T>
__global__ __launch_bounds__(kThreads) void nearest_action_kernel(const T* __restrict__ m, T* __restrict__ out,
                                                                 const int32_t* __restrict__ eidx,
                                                                 const float* __restrict__ rtheta,
                                                                 const int32_t* __restrict__ flags, int E, int H, int W,
                                                                 int pad, int OH, int OW, int top, int left, int src_mod) {
  const int p = blockIdx.z;
  const int i = blockIdx.y;
  const int jb = (blockIdx.x * kThreads + threadIdx.x) * 4;
  if (jb >= OW) return;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int e = min(max(eidx[p], 0), E - 1);
  const float* t = rtheta + e * 6;
  const bool flip = flags && (flags[e] & EQA_FLIP_SRC);
  const T* src = m + (size_t)(src_mod > 0 ? p % src_mod : p) * H * W;
  const float yb = ((float)(top + i) + 0.5f) - 0.5f * (float)Hp;
  T v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int j = left + jb + k;
    const float xb = ((float)j + 0.5f) - 0.5f * (float)Wp;
    const float gx = xb * t[0] + yb * t[1] + t[2];
    const float gy = xb * t[3] + yb * t[4] + t[5];
    const float ix = ((gx + 1.0f) * (float)Wp - 1.0f) / 2.0f;
    const float iy = ((gy + 1.0f) * (float)Hp - 1.0f) / 2.0f;
    const float xr = rintf(ix), yr = rintf(iy);  // std::nearbyint: round half to even
    T val = (T)0;
    if (xr >= 0.0f && xr <= (float)(Wp - 1) && yr >= 0.0f && yr <= (float)(Hp - 1)) {
      const int fx = flip ? (Wp - 1 - (int)xr) : (int)xr;
      const int sx = min(max(fx - pad, 0), W - 1), sy = min(max((int)yr - pad, 0), H - 1);
      val = src[(size_t)sy * W + sx];
    }
    v[k] = val;
  }
  T* o = out + (size_t)p * OH * OW + (size_t)i * OW + jb;
  typedef typename Pack4<T>::type P4;
  if (jb + 3 < OW && ((((uintptr_t)o) & (sizeof(P4) - 1)) == 0)) {
    *reinterpret_cast<P4*>(o) = Pack4<T>::make(v);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (jb + k < OW) o[k] = v[k];
  }
}

// Tiled form of the kernel above: one block = a 64 x 64 output tile whose source bounding box (the tile corners' images,
// one pixel of rounding slack) is first staged into LDS row by row, so that the 90-degree elements of C4 / D4 -- whose
// output rows are source COLUMNS -- no longer touch one cache line per pixel (config 5: 96 uint8 masks of 1024^2 took
// 0.50 ms, 12x their HBM time, in the row-per-block kernel).  Same arithmetic per pixel, bit-identical results; a pixel
// whose source falls outside the staged box (never for rotations) is read from global memory.
constexpr int kNearTile = 64, kNearBox = 96;

template <typename T>
__global__ __launch_bounds__(kThreads) void nearest_action_tile_kernel(const T* __restrict__ m, T* __restrict__ out,
                                                                      const int32_t* __restrict__ eidx,
                                                                      const float* __restrict__ rtheta,
                                                                      const int32_t* __restrict__ flags, int E, int H, int W,
                                                                      int pad, int OH, int OW, int top, int left, int src_mod) {
  __shared__ T s_src[kNearBox * kNearBox];
  const int p = blockIdx.z;
  const int i0 = blockIdx.y * kNearTile, j0 = blockIdx.x * kNearTile;
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int e = min(max(eidx[p], 0), E - 1);
  const float* t = rtheta + e * 6;
  const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
  const bool flip = flags && (flags[e] & EQA_FLIP_SRC);
  const T* src = m + (size_t)(src_mod > 0 ? p % src_mod : p) * H * W;
  auto frame_xy = [&](int i, int j, float& xr, float& yr) {
    const float yb = ((float)(top + i) + 0.5f) - 0.5f * (float)Hp;
    const float xb = ((float)(left + j) + 0.5f) - 0.5f * (float)Wp;
    const float gx = xb * t0 + yb * t1 + t2;
    const float gy = xb * t3 + yb * t4 + t5;
    xr = rintf(((gx + 1.0f) * (float)Wp - 1.0f) / 2.0f);  // std::nearbyint: round half to even
    yr = rintf(((gy + 1.0f) * (float)Hp - 1.0f) / 2.0f);
  };
  // source box of the tile: the map is affine before rounding, so its extremes are at the corners
  const int i1 = min(i0 + kNearTile, OH) - 1, j1 = min(j0 + kNearTile, OW) - 1;
  float xa, ya, xb_, yb_, xc, yc, xd, yd;
  frame_xy(i0, j0, xa, ya); frame_xy(i0, j1, xb_, yb_); frame_xy(i1, j0, xc, yc); frame_xy(i1, j1, xd, yd);
  // (no guard ring: the rounded coordinate is monotone along rows and columns, and a pixel outside the box is read from global memory)
  int fx0 = (int)fminf(fminf(xa, xb_), fminf(xc, xd)) - EQA_ABL_MASKGUARD, fx1 = (int)fmaxf(fmaxf(xa, xb_), fmaxf(xc, xd)) + EQA_ABL_MASKGUARD;
  int fy0 = (int)fminf(fminf(ya, yb_), fminf(yc, yd)) - EQA_ABL_MASKGUARD, fy1 = (int)fmaxf(fmaxf(ya, yb_), fmaxf(yc, yd)) + EQA_ABL_MASKGUARD;
  fx0 = max(fx0, 0); fx1 = min(fx1, Wp - 1); fy0 = max(fy0, 0); fy1 = min(fy1, Hp - 1);
  if (flip) { const int a = Wp - 1 - fx1, b = Wp - 1 - fx0; fx0 = a; fx1 = b; }
  const int sx0 = min(max(fx0 - pad, 0), W - 1), sx1 = min(max(fx1 - pad, 0), W - 1);
  const int sy0 = min(max(fy0 - pad, 0), H - 1), sy1 = min(max(fy1 - pad, 0), H - 1);
  const int bw = sx1 - sx0 + 1, bh = sy1 - sy0 + 1;
  const bool staged = bw > 0 && bh > 0 && bw <= kNearBox && bh <= kNearBox;  // block-uniform
  if (staged) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < bh; r += kThreads / 64) {
      const T* grow = src + (size_t)(sy0 + r) * W + sx0;
      for (int c = lane; c < bw; c += 64) s_src[r * kNearBox + c] = grow[c];
    }
  }
  __syncthreads();
  const int jb = j0 + (threadIdx.x & 15) * 4;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = i0 + (threadIdx.x >> 4) + 16 * g;
    if (i >= OH || jb >= OW) continue;
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float xr, yr;
      frame_xy(i, jb + k, xr, yr);
      T val = (T)0;
      if (xr >= 0.0f && xr <= (float)(Wp - 1) && yr >= 0.0f && yr <= (float)(Hp - 1)) {
        const int fx = flip ? (Wp - 1 - (int)xr) : (int)xr;
        const int sx = min(max(fx - pad, 0), W - 1), sy = min(max((int)yr - pad, 0), H - 1);
        const int lx = sx - sx0, ly = sy - sy0;
        val = (staged && (unsigned)lx < (unsigned)bw && (unsigned)ly < (unsigned)bh) ? s_src[ly * kNearBox + lx]
                                                                                     : src[(size_t)sy * W + sx];
      }
      v[k] = val;
    }
    T* o = out + (size_t)p * OH * OW + (size_t)i * OW + jb;
    typedef typename Pack4<T>::type P4;
    if (jb + 3 < OW && ((((uintptr_t)o) & (sizeof(P4) - 1)) == 0)) {
      *reinterpret_cast<P4*>(o) = Pack4<T>::make(v);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (jb + k < OW) o[k] = v[k];
    }
  }
}

// uint8 masks, the config-5 case (96 masks of 1024 x 1024 per step): the tiled kernel above staged its source box byte by
// byte (one 64-byte wave load per 64 pixels) and stored 4 bytes per lane -- 0.21 ms for 0.2 GB = 1 TB/s, bound by the number
// of memory instructions.  Same tile, same per-pixel arithmetic (bit-identical results), but the box is staged in whole dwords
// (rows start 4-byte aligned: W % 4 == 0) and a thread owns 16 consecutive pixels of one row = one 16-byte store.  The source
// planes come either from one contiguous tensor or from a table of per-plane pointers (the masks of a batch live in one
// tensor per sample: no concatenation pass in front of the kernel).
constexpr int kU8Pitch = 104;  // bytes per staged row: 96 + 3 (alignment slack), rounded to a dword multiple + 1 dword
__global__ __launch_bounds__(kThreads) void mask_action_u8_kernel(const uint8_t* __restrict__ m, const uint8_t* const* __restrict__ planes,
                                                                 uint8_t* __restrict__ out, const int32_t* __restrict__ eidx,
                                                                 const float* __restrict__ rtheta, const int32_t* __restrict__ flags, int E,
                                                                 int H, int W) {
  __shared__ __attribute__((aligned(16))) uint8_t s_src[kNearBox * kU8Pitch + 8];  // + 8: the fifth dword of a row run at the very end
  const int p = blockIdx.z;
  const int i0 = blockIdx.y * kNearTile, j0 = blockIdx.x * kNearTile;
  const int e = min(max(eidx[p], 0), E - 1);
  const float* t = rtheta + e * 6;
  const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = t[4], t5 = t[5];
  const bool flip = flags && (flags[e] & EQA_FLIP_SRC);
  const uint8_t* src = planes ? planes[p] : m + (size_t)p * H * W;
  auto frame_xy = [&](int i, int j, float& xr, float& yr) {
    const float yb = ((float)i + 0.5f) - 0.5f * (float)H;
    const float xb = ((float)j + 0.5f) - 0.5f * (float)W;
    const float gx = xb * t0 + yb * t1 + t2;
    const float gy = xb * t3 + yb * t4 + t5;
    xr = rintf(((gx + 1.0f) * (float)W - 1.0f) / 2.0f);  // std::nearbyint: round half to even
    yr = rintf(((gy + 1.0f) * (float)H - 1.0f) / 2.0f);
  };
  const int i1 = min(i0 + kNearTile, H) - 1, j1 = min(j0 + kNearTile, W) - 1;
  float xa, ya, xb_, yb_, xc, yc, xd, yd;
  frame_xy(i0, j0, xa, ya); frame_xy(i0, j1, xb_, yb_); frame_xy(i1, j0, xc, yc); frame_xy(i1, j1, xd, yd);
  // The rounded coordinate is monotone along rows and columns of the tile, so the four corners bound it exactly; a pixel that
  // landed outside the box all the same is read from global memory below, so the box needs no guard ring for correctness.  With
  // one (rounds 1-3) the staged rows of an axis-aligned element were 66 bytes starting one byte in front of the tile's 64:
  // two 128-byte lines per row instead of one.
  int fx0 = (int)fminf(fminf(xa, xb_), fminf(xc, xd)) - EQA_ABL_MASKGUARD, fx1 = (int)fmaxf(fmaxf(xa, xb_), fmaxf(xc, xd)) + EQA_ABL_MASKGUARD;
  int fy0 = (int)fminf(fminf(ya, yb_), fminf(yc, yd)) - EQA_ABL_MASKGUARD, fy1 = (int)fmaxf(fmaxf(ya, yb_), fmaxf(yc, yd)) + EQA_ABL_MASKGUARD;
  fx0 = max(fx0, 0); fx1 = min(fx1, W - 1); fy0 = max(fy0, 0); fy1 = min(fy1, H - 1);
  if (flip) { const int a = W - 1 - fx1, b = W - 1 - fx0; fx0 = a; fx1 = b; }
  const int sx0 = fx0 & ~3, sx1 = fx1;               // dword-aligned left edge
  const int sy0 = fy0, sy1 = fy1;
  const int bw = sx1 - sx0 + 1, bh = sy1 - sy0 + 1;
  const bool staged = bw > 0 && bh > 0 && bw <= kU8Pitch - 4 && bh <= kNearBox;  // block-uniform
  // Axis-aligned elements on aligned tiles (every element of C4 / D4 on the 1024 x 1024 masks of config 5): the box is 64 rows of 64
  // bytes starting on a 16-byte boundary -- ONE 16-byte load and two 8-byte LDS stores per thread instead of twelve predicated
  // dword passes (a third of whose lanes and passes carry data): the staging was half of the kernel's instructions.
  const bool staged16 = staged && bw <= 64 && bh <= 64 && (sx0 & 15) == 0 && (W & 15) == 0 && sx0 + 64 <= W &&
                        (reinterpret_cast<uintptr_t>(src) & 15) == 0;   // block-uniform
  if (staged16) {
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (r < bh) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(sy0 + r) * W + sx0 + 16 * q);
      uint2* d = reinterpret_cast<uint2*>(s_src + r * kU8Pitch + 16 * q);
      d[0] = make_uint2(v.x, v.y);
      d[1] = make_uint2(v.z, v.w);
    }
  } else if (staged) {
    const int nd = (bw + 3) >> 2;                     // dwords per row (the last one may reach past sx1: still inside the row, W % 4 == 0)
    // 32 dword slots per row (nd <= 25), 8 rows per pass, all 12 passes' loads in flight before the first LDS store (a rolled
    // load -> store loop pays one HBM round trip per pass: 0.14 instead of 0.21 ms was all the dword staging bought that way)
    constexpr int kPasses = kNearBox * 32 / kThreads;
    const int d = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    uint32_t w[kPasses];
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
      const int r = r0 + 8 * k;
      const bool on = d < nd && r < bh;
      const uint32_t v = *reinterpret_cast<const uint32_t*>(src + (size_t)(sy0 + (on ? r : 0)) * W + sx0 + 4 * (on ? d : 0));
      w[k] = v;
    }
#pragma unroll
    for (int k = 0; k < kPasses; ++k) {
      const int r = r0 + 8 * k;
      if (d < nd && r < bh) *reinterpret_cast<uint32_t*>(s_src + r * kU8Pitch + 4 * d) = w[k];
    }
  }
  __syncthreads();
  const int i = i0 + (threadIdx.x >> 2);
  const int jb = j0 + (threadIdx.x & 3) * 16;
  if (i >= H || jb >= W) return;
  uint32_t w4[4] = {0u, 0u, 0u, 0u};
  // Axis-aligned elements (every element of C4 / D4: the config-5 case) move a run of 16 output pixels onto 16 consecutive
  // source pixels of one row or one column.  The run's two END pixels go through the reference's arithmetic; if they land 15
  // apart along one axis, on the same line of the other, both inside the frame and the staged box, and their unrounded
  // coordinates are within 0.25 of the integers they round to, then the 14 pixels between them round to the integers between
  // (the coordinate is affine in the pixel index up to ~1e-4 of fp32 noise at |x| <= 2^15: an interior pixel could only round
  // elsewhere from within that noise of a .5 tie, and a quarter pixel is far from it) -- bit-identical to evaluating all 16,
  // at 2 coordinate evaluations instead of 16 (the kernel was bound by its ~28 vector instructions per pixel: 1.9 TB/s).
  bool fast = false;
  if (staged && jb + 15 < W) {
    auto raw_xy = [&](int ii, int jj, float& fx, float& fy) {
      const float yb = ((float)ii + 0.5f) - 0.5f * (float)H;
      const float xb = ((float)jj + 0.5f) - 0.5f * (float)W;
      const float gx = xb * t0 + yb * t1 + t2;
      const float gy = xb * t3 + yb * t4 + t5;
      fx = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
      fy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    };
    float fxa, fya, fxb, fyb;
    raw_xy(i, jb, fxa, fya);
    raw_xy(i, jb + 15, fxb, fyb);
    const float xra = rintf(fxa), yra = rintf(fya), xrb = rintf(fxb), yrb = rintf(fyb);
    const bool inside = fminf(xra, xrb) >= 0.0f && fmaxf(xra, xrb) <= (float)(W - 1) && fminf(yra, yrb) >= 0.0f && fmaxf(yra, yrb) <= (float)(H - 1);
    const bool snug = fabsf(fxa - xra) < 0.25f && fabsf(fya - yra) < 0.25f && fabsf(fxb - xrb) < 0.25f && fabsf(fyb - yrb) < 0.25f;
    const int sxa = flip ? (W - 1 - (int)xra) : (int)xra, sxb = flip ? (W - 1 - (int)xrb) : (int)xrb;
    const int sya = (int)yra, syb = (int)yrb;
    const int ddx = sxb - sxa, ddy = syb - sya;
    const bool line = (ddy == 0 && (ddx == 15 || ddx == -15)) || (ddx == 0 && (ddy == 15 || ddy == -15));
    const int lxa = sxa - sx0, lya = sya - sy0, lxb = sxb - sx0, lyb = syb - sy0;
    const bool boxed = (unsigned)lxa < (unsigned)bw && (unsigned)lya < (unsigned)bh && (unsigned)lxb < (unsigned)bw && (unsigned)lyb < (unsigned)bh;
    fast = inside && snug && line && boxed;
    if (fast) {
      if (ddy == 0) {
        // along a source row: the 16 bytes [lo, lo + 16) come out of 5 aligned dword reads and 4 funnel shifts; a run that walks
        // the row backwards (flips, 180 degrees) is the same bytes in reverse order
        const int lo = lya * kU8Pitch + min(lxa, lxb);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(s_src + (lo & ~3));
        const uint32_t sh = (uint32_t)(lo & 3);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
        const uint32_t f0 = __builtin_amdgcn_alignbyte(d1, d0, sh), f1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
        const uint32_t f2 = __builtin_amdgcn_alignbyte(d3, d2, sh), f3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
        const bool rev = ddx < 0;
        w4[0] = rev ? __builtin_bswap32(f3) : f0;
        w4[1] = rev ? __builtin_bswap32(f2) : f1;
        w4[2] = rev ? __builtin_bswap32(f1) : f2;
        w4[3] = rev ? __builtin_bswap32(f0) : f3;
      } else {
        const int stride = (ddy / 15) * kU8Pitch;
        const uint8_t* sp = s_src + lya * kU8Pitch + lxa;
#pragma unroll
        for (int k = 0; k < 16; ++k) w4[k >> 2] |= (uint32_t)sp[k * stride] << (8 * (k & 3));
      }
    }
  }
  if (!fast) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float xr, yr;
      frame_xy(i, jb + k, xr, yr);
      uint32_t val = 0u;
      if (xr >= 0.0f && xr <= (float)(W - 1) && yr >= 0.0f && yr <= (float)(H - 1)) {
        const int sx = flip ? (W - 1 - (int)xr) : (int)xr, sy = (int)yr;
        const int lx = sx - sx0, ly = sy - sy0;
        val = (staged && (unsigned)lx < (unsigned)bw && (unsigned)ly < (unsigned)bh) ? s_src[ly * kU8Pitch + lx] : src[(size_t)sy * W + sx];
      }
      w4[k >> 2] |= val << (8 * (k & 3));
    }
  }
  uint8_t* o = out + (size_t)p * H * W + (size_t)i * W + jb;
  if (jb + 15 < W) {
    *reinterpret_cast<uint4*>(o) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
  } else {
    for (int k = 0; k < 16; ++k)
      if (jb + k < W) o[k] = (uint8_t)(w4[k >> 2] >> (8 * (k & 3)));
  }
}

int launch_mask_u8(const uint8_t* m, const uint8_t* const* planes, uint8_t* out, const int32_t* eidx, const float* rtheta,
                   const int32_t* flags, int E, int n_planes, int H, int W, void* stream) {
  if ((!m && !planes) || !out || !eidx || !rtheta || E <= 0 || n_planes < 0 || H <= 0 || W <= 0) return EQA_ERR_INVALID_ARG;
  if (n_planes > 65535 || H > 65535 * kNearTile || (W & 15) || ((uintptr_t)out & 15) || ((uintptr_t)m & 3)) return EQA_ERR_UNSUPPORTED;
  if (n_planes == 0) return EQA_OK;
  hipLaunchKernelGGL(mask_action_u8_kernel, dim3((W + kNearTile - 1) / kNearTile, (H + kNearTile - 1) / kNearTile, n_planes), dim3(kThreads),
                     0, (hipStream_t)stream, m, planes, out, eidx, rtheta, flags, E, H, W);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

template <typename T>
int launch_nearest(const T* m, T* out, const int32_t* eidx, const float* rtheta, const int32_t* flags, int E,
                          int n_planes, int H, int W, int pad, int OH, int OW, int top, int left, int src_mod, void* stream) {
  if (!m || !out || !eidx || !rtheta || E <= 0 || n_planes < 0 || H <= 0 || W <= 0 || pad < 0 || OH <= 0 || OW <= 0 ||
      top < 0 || left < 0 || top + OH > H + 2 * pad || left + OW > W + 2 * pad || src_mod < 0)
    return EQA_ERR_INVALID_ARG;
  if (n_planes > 65535 || OH > 65535) return EQA_ERR_UNSUPPORTED;
  if (n_planes == 0) return EQA_OK;
  if (g_force_direct)  // eqa_set_option(0, 1): the row-per-block kernel without LDS staging (tests compare the two)
    hipLaunchKernelGGL((nearest_action_kernel<T>), dim3((OW / 4 + kThreads) / kThreads, OH, n_planes), dim3(kThreads), 0,
                       (hipStream_t)stream, m, out, eidx, rtheta, flags, E, H, W, pad, OH, OW, top, left, src_mod);
  else
    hipLaunchKernelGGL((nearest_action_tile_kernel<T>),
                       dim3((OW + kNearTile - 1) / kNearTile, (OH + kNearTile - 1) / kNearTile, n_planes), dim3(kThreads), 0,
                       (hipStream_t)stream, m, out, eidx, rtheta, flags, E, H, W, pad, OH, OW, top, left, src_mod);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

// dL/d(angle) partials alone: the LDS-staged form (the forward's window, gathered for the derivative), or with eqa_set_option(0, 1)
// round 3's direct-gather kernel
int launch_angle_grad(const ActionArgs& a, const dim3& grid, hipStream_t st) {
  if (a.force_direct) {
    hipLaunchKernelGGL((group_action_bwd_kernel<1, false>), grid, dim3(kThreads), 0, st, a);
    return launch_status();
  }
  const int ch = action_ch(a.C);
  const size_t lds = (size_t)ch * kBox * kLdsStride * sizeof(float) + (a.chan_map ? kMaxMapG * sizeof(int) : 0);
  if (ch == 3) hipLaunchKernelGGL(group_action_angle_kernel<3>, grid, dim3(kThreads), lds, st, a);
  else if (ch == 2) hipLaunchKernelGGL(group_action_angle_kernel<2>, grid, dim3(kThreads), lds, st, a);
  else hipLaunchKernelGGL(group_action_angle_kernel<1>, grid, dim3(kThreads), lds, st, a);
  return launch_status();
}

int launch_action_bwd(int grad_mode, const float* src, const float* grad_out, const int32_t* gidx, const float* theta,
                      const int32_t* flags, const int32_t* chan_map, float* grad_src, float* partial, int num_elements, int G,
                      int n_out, int B, int C, int H, int W, int pad, int OH, int OW, int top, int left, void* stream) {
  if (n_out == 0 && B >= 0) return EQA_OK;
  if (!grad_out || (!grad_src && !partial)) return EQA_ERR_INVALID_ARG;
  ActionArgs a;
  const int rc = fill_action_args(a, src, nullptr, gidx, theta, flags, chan_map, num_elements, G, n_out, B, C, H, W, pad,
                                  OH, OW, top, left);
  if (rc != EQA_OK) return rc;
  if (n_out == 0) return EQA_OK;
  a.gout = grad_out; a.gsrc = grad_src; a.partial = partial;
  const int tiles_x = (OW + kTile - 1) / kTile, tiles_y = (OH + kTile - 1) / kTile;
  const int groups = (n_out + kXcd - 1) / kXcd;
  if (tiles_y > 65535 || groups > 65535) return EQA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(kXcd * tiles_x), (unsigned)tiles_y, (unsigned)groups);
  hipStream_t st = (hipStream_t)stream;
  // un-padded, one output per image, whole-frame output: the input gradient is an exact gather (no atomics; grad_src need
  // not be zeroed).  The transform gradient, if wanted too, comes from its own launch of the scatter-free mode.
  const bool gather = grad_src && pad == 0 && gidx && n_out == B && !g_force_direct;
  if (gather) {
    const dim3 ggrid((W + 63) / 64, (H + 3) / 4, B);
    if (ggrid.y > 65535 || ggrid.z > 65535) return EQA_ERR_UNSUPPORTED;
    if (chan_map)
      hipLaunchKernelGGL((group_action_bwd_gather_kernel<true>), ggrid, dim3(kThreads), 0, st, a);
    else
      hipLaunchKernelGGL((group_action_bwd_gather_kernel<false>), ggrid, dim3(kThreads), 0, st, a);
    if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
    if (!partial) return EQA_OK;
    if (grad_mode == 1) return launch_angle_grad(a, grid, st);
    hipLaunchKernelGGL((group_action_bwd_kernel<2, false>), grid, dim3(kThreads), 0, st, a);
    return launch_status();
  }
  if (!partial)
    hipLaunchKernelGGL((group_action_bwd_kernel<0, true>), grid, dim3(kThreads), 0, st, a);
  else if (grad_mode == 1 && grad_src)
    hipLaunchKernelGGL((group_action_bwd_kernel<1, true>), grid, dim3(kThreads), 0, st, a);
  else if (grad_mode == 1)
    return launch_angle_grad(a, grid, st);
  else if (grad_src)
    hipLaunchKernelGGL((group_action_bwd_kernel<2, true>), grid, dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL((group_action_bwd_kernel<2, false>), grid, dim3(kThreads), 0, st, a);
  return launch_status();
}

// ------------------------------------------------------------------------------------------------
// I6, boxes: flip_boxes (images/utils.py:97-109) + rotate_boxes (:161-187, rotate_points :139-158) for every box of the
// batch in one launch.  The reference does this per sample with a dozen element-wise launches each; the batched torch form
// still was ~55 launches of 1-2 us spaced ~10 us apart -- 0.6 of config 5's 1.9 ms step.  Same fp32 arithmetic in the same
// order, no fused multiply-adds: rad = deg * (pi/180); x' = ox + cos*(x-ox) - sin*(y-oy); y' = oy + sin*(x-ox) + cos*(y-oy)
// about (W/2, W/2); the box is then re-sorted corner-wise.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void boxes_action_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ img_of_box,
                                                               const float* __restrict__ rotation_deg, float* __restrict__ flipped,
                                                               float* __restrict__ out, int n, float width, int flip_all) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  float4 b = reinterpret_cast<const float4*>(boxes)[i];
  if (flip_all) {  // boxes[:, [0, 2]] = width - boxes[:, [2, 0]]
    const float x0 = width - b.z, x1 = width - b.x;
    b.x = x0;
    b.z = x1;
    if (flipped) reinterpret_cast<float4*>(flipped)[i] = b;
  }
  const float rad = rotation_deg[img_of_box[i]] * 0.017453292519943295f;  // torch.deg2rad
  const float c = cosf(rad), sn = sinf(rad);
  const float o = width / 2;
  const float x0 = o + c * (b.x - o) - sn * (b.y - o), y0 = o + sn * (b.x - o) + c * (b.y - o);
  const float x1 = o + c * (b.z - o) - sn * (b.w - o), y1 = o + sn * (b.z - o) + c * (b.w - o);
  reinterpret_cast<float4*>(out)[i] = make_float4(fminf(x0, x1), fminf(y0, y1), fmaxf(x0, x1), fmaxf(y0, y1));
}

}  // namespace

extern "C" {

int eqa_abi_version(void) { return EQA_ABI_VERSION; }

int eqa_get_option(int key) {
  if (key == 100) return kMaxWinK;   // read-only: the largest window the window-sum kernels take (ops.MAX_WINDOW_K must equal it)
  return key == 0 ? g_force_direct : key == 1 ? eqa::g_vn_kernel_choice : key == 2 ? eqa::g_cgemm_bf16_form : key == 3 ? g_c1_tiles : EQA_ERR_INVALID_ARG;
}

int64_t eqa_fold_edge_pad_workspace_bytes(int planes, int H, int W, int pad) {
  if (planes <= 0 || H <= 0 || W <= 0 || pad < 0) return 0;
  return (int64_t)planes * (H + 2 * pad) * W * (int64_t)sizeof(float);
}

int eqa_fold_edge_pad(const float* gframe, float* gsrc, void* workspace, int planes, int H, int W, int pad, void* stream) {
  if (planes < 0 || H <= 0 || W <= 0 || pad < 0) return EQA_ERR_INVALID_ARG;
  if (planes == 0) return EQA_OK;
  if (!gframe || !gsrc || !workspace) return EQA_ERR_INVALID_ARG;
  const size_t rows = (size_t)planes * (H + 2 * pad);
  if (rows > 0x7fffffffULL || planes > 65535) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fold_edge_pad_x_kernel, dim3((unsigned)rows), dim3(kThreads), 0, st, gframe, (float*)workspace, W, pad);
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  hipLaunchKernelGGL(fold_edge_pad_y_kernel, dim3((W + kThreads - 1) / kThreads, planes), dim3(kThreads), 0, st,
                     (const float*)workspace, gsrc, H, W, pad);
  return launch_status();
}

int eqa_set_option(int key, int value) {
  if (key == 0) {
    g_force_direct = value ? 1 : 0;
    return EQA_OK;
  }
  if (key == 1 && value >= 0 && value <= 2) {
    eqa::g_vn_kernel_choice = value;
    return EQA_OK;
  }
  if (key == 2 && value >= 0 && value <= 1) {
    eqa::g_cgemm_bf16_form = value;
    return EQA_OK;
  }
  if (key == 3 && (value == 0 || value == 2 || value == 4)) {
    g_c1_tiles = value;
    return EQA_OK;
  }
  return EQA_ERR_INVALID_ARG;
}

int eqa_group_action_fwd(const float* src, float* dst, const int32_t* gidx, const float* theta, const int32_t* flags,
                         const int32_t* chan_map, int num_elements, int G, int n_out, int B, int C, int H, int W,
                         int pad, int OH, int OW, int top, int left, void* stream) {
  return launch_action(src, dst, gidx, theta, flags, chan_map, num_elements, G, n_out, B, C, H, W, pad, OH, OW, top,
                       left, stream);
}

int eqa_group_action_fwd_hint(const float* src, float* dst, const int32_t* gidx, const float* theta, const int32_t* flags,
                              const int32_t* chan_map, int num_elements, int G, int n_out, int B, int C, int H, int W,
                              int pad, int OH, int OW, int top, int left, int max_window, void* stream) {
  if (max_window < 0) return EQA_ERR_INVALID_ARG;
  return launch_action(src, dst, gidx, theta, flags, chan_map, num_elements, G, n_out, B, C, H, W, pad, OH, OW, top, left, stream,
                       max_window);
}

int eqa_canon_transform_fwd(const float* x, float* y, const int32_t* gidx, const float* theta, const int32_t* flags,
                            int num_elements, int B, int C, int H, int W, int pad, void* stream) {
  if (B == 0) return EQA_OK;
  if (!gidx) return EQA_ERR_INVALID_ARG;
  // CenterCrop offset of torchvision: int(round((Hp - H) / 2)) == pad exactly, since Hp - H = 2*pad
  return launch_action(x, y, gidx, theta, flags, nullptr, num_elements, 1, B, B, C, H, W, pad, H, W, pad, pad, stream);
}

int eqa_invert_action_fwd(const float* f, float* out, const int32_t* gidx, const float* theta, const int32_t* flags,
                          const int32_t* chan_map, int num_elements, int G, int B, int C, int H, int W, void* stream) {
  if (B == 0) return EQA_OK;
  if (!gidx) return EQA_ERR_INVALID_ARG;
  return launch_action(f, out, gidx, theta, flags, chan_map, num_elements, G, B, B, C, H, W, 0, H, W, 0, 0, stream);
}

int eqa_group_action_pair(const float* x, float* y, const float* theta_canon, const int32_t* flags_canon, int pad, int C,
                          const float* f, float* out, const float* theta_inv, const int32_t* flags_inv, const int32_t* chan_map,
                          int G, int Cf, const int32_t* gidx, int num_elements, int B, int H, int W, void* stream) {
  return launch_pair(x, y, theta_canon, flags_canon, pad, C, f, out, theta_inv, flags_inv, chan_map, G, Cf, gidx, num_elements, B, H,
                     W, stream);
}

int eqa_orbit_expand_fwd(const float* x, float* y, const float* theta, const int32_t* flags, int num_elements, int B,
                         int C, int S, int pad, void* stream) {
  if (B == 0 && num_elements > 0) return EQA_OK;
  if (num_elements <= 0 || B <= 0) return EQA_ERR_INVALID_ARG;
  if ((long long)num_elements * B > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;
  return launch_action(x, y, nullptr, theta, flags, nullptr, num_elements, 1, num_elements * B, B, C, S, S, pad, S, S,
                       pad, pad, stream);
}

int eqa_group_action_bwd_tiles(int OH, int OW) {
  if (OH <= 0 || OW <= 0) return 0;
  return ((OH + kTile - 1) / kTile) * ((OW + kTile - 1) / kTile);
}

int eqa_group_action_bwd(const float* src, const float* grad_out, const int32_t* gidx, const float* theta,
                         const int32_t* flags, const int32_t* chan_map, float* grad_src, float* grad_angle_partial,
                         int num_elements, int G, int n_out, int B, int C, int H, int W, int pad, int OH, int OW,
                         int top, int left, void* stream) {
  return launch_action_bwd(1, src, grad_out, gidx, theta, flags, chan_map, grad_src, grad_angle_partial, num_elements, G, n_out,
                           B, C, H, W, pad, OH, OW, top, left, stream);
}

int eqa_group_action_bwd_theta(const float* src, const float* grad_out, const int32_t* gidx, const float* theta,
                               const int32_t* flags, const int32_t* chan_map, float* grad_src, float* grad_theta_partial,
                               int num_elements, int G, int n_out, int B, int C, int H, int W, int pad, int OH, int OW,
                               int top, int left, void* stream) {
  return launch_action_bwd(2, src, grad_out, gidx, theta, flags, chan_map, grad_src, grad_theta_partial, num_elements, G, n_out,
                           B, C, H, W, pad, OH, OW, top, left, stream);
}

int eqa_crop_resize_aa(const float* x, float* y, const float* wx, const int32_t* x0, const float* wy, const int32_t* y0,
                       int planes, int H, int W, int OH, int OW, int K, int max_rows, int x_begin, int x_span, void* stream) {
  if (!x || !y || !wx || !x0 || !wy || !y0 || planes < 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || K <= 0 || max_rows <= 0)
    return EQA_ERR_INVALID_ARG;
  const size_t lds = (size_t)max_rows * OW * sizeof(float);
  if (lds > 96 * 1024 || planes > 65535) return EQA_ERR_UNSUPPORTED;
  if (planes == 0) return EQA_OK;
  const dim3 grid((OH + kAaBand - 1) / kAaBand, planes);
  if (K > EQA_AA_WIDE_MIN_K && x_span > 0 && x_begin >= 0 && x_begin + x_span <= W) {
    // wide filters: stage the input rows in LDS.  Lanes read ~x_span / OW floats apart; make that stride odd.
    const int stride = (x_span + OW / 2) / OW;
    const int pad_shift = (stride >= 2 && (stride & 1) == 0) ? __builtin_ctz((unsigned)stride) : 0;
    const int row_stride = x_span + (pad_shift ? (x_span >> pad_shift) : 0) + 1;
    // input rows staged per iteration: as many as fit next to the intermediate band (fewer barrier rounds), at most 8
    const size_t row_bytes = (size_t)row_stride * sizeof(float);
    const int rpi = (int)std::min<size_t>(8, lds + row_bytes <= 96 * 1024 ? (96 * 1024 - lds) / row_bytes : 0);
    const size_t lds2 = lds + (size_t)rpi * row_bytes;
    static const bool stream_off = [] { const char* e = getenv("EQA_AA_STREAM"); return e && e[0] == '0'; }();
    if (!stream_off && x_span <= 4 * kThreads && (x_span & 3) == 0 && (x_begin & 3) == 0 && (W & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
        (size_t)(kAaBand / 2) * row_bytes <= 64 * 1024) {
      hipLaunchKernelGGL(crop_resize_aa_stream_kernel, grid, dim3(kThreads), (kAaBand / 2) * row_bytes, (hipStream_t)stream, x, y, wx, x0, wy,
                         y0, H, W, OH, OW, K, x_begin, x_span, pad_shift, row_stride);
      return launch_status();
    }
    if (rpi >= 1) {
      hipLaunchKernelGGL(crop_resize_aa_wide_kernel, grid, dim3(kThreads), lds2, (hipStream_t)stream, x, y, wx, x0, wy, y0, H, W, OH,
                         OW, K, max_rows, x_begin, x_span, pad_shift, row_stride, rpi);
      return launch_status();
    }
  }
  // narrow filters over aligned rows: the LDS-staged form (whole band requested at once)
  static const bool staged_off = [] { const char* e = getenv("EQA_AA_STAGED"); return e && e[0] == '0'; }();
  if (!staged_off && K <= EQA_AA_WIDE_MIN_K && (W & 3) == 0 && (((uintptr_t)x) & 15) == 0 && x_span > 0 && x_begin >= 0 &&
      x_begin + x_span <= W) {
    const int xb = x_begin & ~3, xl = std::min(W, (x_begin + x_span + 3) & ~3) - xb;
    // 16 output rows per block where the map has at least four such bands: 4 of 34 staged rows are shared with the next band instead of
    // 4 of 19 (with the bands of a plane on one XCD -- round 4 -- 35.9 us per 256 x 3 planes of 224 -> 180 -> 96 against 39-40 for
    // bands of 8; before that mapping the larger band was the slower one).  EQA_AA_BAND=8 / 16 forces either.
    static const int band_env = [] { const char* e = getenv("EQA_AA_BAND"); return e ? atoi(e) : 0; }();
    int band = band_env == 16 ? 16 : (band_env == 8 ? 8 : (OH >= 64 ? 16 : 8));
    auto staged_lds = [&](int bnd) { return ((size_t)(bnd / kAaBand) * max_rows * (xl + OW) + bnd * (EQA_AA_WIDE_MIN_K + 1)) * sizeof(float); };
    if (band == 16 && band_env != 16 && (staged_lds(16) > 64 * 1024 || 16 * K > kThreads)) band = 8;   // the smaller band may still fit
    const int cap_rows = (band / kAaBand) * max_rows;   // a band of 16 rows = two of the 8-row bands `max_rows` was taken over
    const size_t lds3 = staged_lds(band);
    if (lds3 <= 64 * 1024 && band * K <= kThreads) {
      // persistent over planes: about 8 resident blocks per CU in all, each walking planes with a stride of gridDim.y
      static const int per_cu = [] { const char* e = getenv("EQA_AA_BLOCKS_PER_CU"); return e ? std::max(1, atoi(e)) : 12; }();
      const int nbands = (OH + band - 1) / band;
      const int groups = std::max(1, (std::min(planes, (256 * per_cu + nbands - 1) / nbands) + kXcd - 1) / kXcd);   // plane groups of 8 (one plane per XCD)
      const bool few = cap_rows <= 20;   // 5 loads per thread cover the band's rows (else 10: up to 40 rows)
#define EQA_AA_STAGED(K_)                                                                                                              \
  case K_:                                                                                                                             \
    if (band == 16)                                                                                                                    \
      hipLaunchKernelGGL((crop_resize_aa_staged_kernel<K_, 16, 10>), dim3(kXcd, nbands, groups), dim3(kThreads), lds3, (hipStream_t)stream, x, \
                         y, wx, x0, wy, y0, planes, H, W, OH, OW, cap_rows, xb, xl);                                                   \
    else if (few)                                                                                                                      \
      hipLaunchKernelGGL((crop_resize_aa_staged_kernel<K_, 8, 5>), dim3(kXcd, nbands, groups), dim3(kThreads), lds3, (hipStream_t)stream, x,  \
                         y, wx, x0, wy, y0, planes, H, W, OH, OW, cap_rows, xb, xl);                                                   \
    else                                                                                                                               \
      hipLaunchKernelGGL((crop_resize_aa_staged_kernel<K_, 8, 10>), dim3(kXcd, nbands, groups), dim3(kThreads), lds3, (hipStream_t)stream, x, \
                         y, wx, x0, wy, y0, planes, H, W, OH, OW, cap_rows, xb, xl);                                                   \
    break
      switch (K) {
        EQA_AA_STAGED(1); EQA_AA_STAGED(2); EQA_AA_STAGED(3); EQA_AA_STAGED(4); EQA_AA_STAGED(5); EQA_AA_STAGED(6); EQA_AA_STAGED(7);
        EQA_AA_STAGED(8);
        default: return EQA_ERR_UNSUPPORTED;
      }
#undef EQA_AA_STAGED
      return launch_status();
    }
  }
  hipLaunchKernelGGL(crop_resize_aa_kernel, grid, dim3(kThreads), lds, (hipStream_t)stream, x, y, wx, x0, wy, y0, H, W, OH, OW, K,
                     max_rows);
  return launch_status();
}

int eqa_mask_action_nearest(const uint8_t* m, uint8_t* out, const int32_t* eidx, const float* rtheta, const int32_t* flags,
                            int num_elements, int n_masks, int H, int W, void* stream) {
  if (!g_force_direct && m && (W & 15) == 0 && (((uintptr_t)out & 15) | ((uintptr_t)m & 3)) == 0 && n_masks <= 65535)
    return launch_mask_u8(m, nullptr, out, eidx, rtheta, flags, num_elements, n_masks, H, W, stream);
  return launch_nearest<uint8_t>(m, out, eidx, rtheta, flags, num_elements, n_masks, H, W, 0, H, W, 0, 0, 0, stream);
}

int eqa_mask_action_nearest_planes(const uint8_t* const* planes, uint8_t* out, const int32_t* eidx, const float* rtheta,
                                   const int32_t* flags, int num_elements, int n_masks, int H, int W, void* stream) {
  return launch_mask_u8(nullptr, planes, out, eidx, rtheta, flags, num_elements, n_masks, H, W, stream);
}

int eqa_boxes_action(const float* boxes, const int32_t* img_of_box, const float* rotation_deg, float* flipped, float* out,
                     int n, float width, int flip_all, void* stream) {
  if (n == 0) return EQA_OK;
  if (!boxes || !img_of_box || !rotation_deg || !out || n < 0) return EQA_ERR_INVALID_ARG;
  if ((((uintptr_t)boxes | (uintptr_t)out | (uintptr_t)flipped) & 15)) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(boxes_action_kernel, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, boxes,
                     img_of_box, rotation_deg, flipped, out, n, width, flip_all);
  return launch_status();
}

int eqa_image_action_nearest(const float* x, float* out, const int32_t* eidx, const float* rtheta, const int32_t* flags,
                             int num_elements, int n_planes, int src_mod, int H, int W, int pad, int OH, int OW, int top,
                             int left, void* stream) {
  return launch_nearest<float>(x, out, eidx, rtheta, flags, num_elements, n_planes, H, W, pad, OH, OW, top, left, src_mod, stream);
}

}  // extern "C"
