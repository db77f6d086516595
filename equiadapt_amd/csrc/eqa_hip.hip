// libeqa_hip.so -- hand-written CDNA4 (gfx950, wave64) kernels for equiadapt's canonicalization hot path.
// C ABI: include/eqa_hip.h.  Design notes: DESIGN.md.
//
// All kernels here are HBM-bound gathers / reductions / streams: the engineering is in coalesced
// global access, LDS-staged source tiles for the rotated resampling, XCD-aware block->image mapping
// (each XCD's private L2 sees whole images), and wave-level shuffles for the orientation reduction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "eqa_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 32;      // output tile edge (px): 256 threads x 4 px
constexpr int kBox = 47;       // staged source window edge: floor(31*sqrt(2)) + neighbour + floor/guard slack = 47
constexpr int kLdsStride = 47; // odd dword stride: the 8x4-lane gather pattern is bank-conflict-free at 0/90/180/270 deg
                               // 3 channels x 47 x 47 x 4 B = 26.5 KB -> 6 blocks per CU (160 KB LDS)
constexpr int kXcd = 8;
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
  // backward only
  const float* gout;  // dL/d(output), shape of dst
  float* gsrc;        // dL/d(source), shape of src, pre-zeroed (nullable)
  float* partial;     // per (output image, tile) partial of dL/d(angle [rad]) (nullable)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// torch.linspace(-1, 1, steps) as the CPU kernel evaluates it (symmetric halves), fp32.
// Written select-style (one integer select, one fma-shaped op, one select) so it stays branch-free.
__device__ __forceinline__ float lin_m1_p1(int idx, int steps, float step) {
  const bool lo = idx < (steps >> 1);
  const float k = (float)(lo ? idx : steps - 1 - idx);
  const float up = -1.0f + step * k, dn = 1.0f - step * k;
  return lo ? up : dn;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ablation switches for tools/ablate.sh (never set in the product build)
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

// One block = one 32x32 output tile of one output image, all channels, CH channels per LDS stage.
//   grid = (8 * tiles_x, tiles_y, ceil(n_out / 8)):  blockIdx.x & 7 is the XCD the dispatcher deals the block
//   to, so each XCD works on whole images (n = 8*z + xcd) and the overlapping source windows of neighbouring
//   tiles hit in that XCD's private L2.  No integer division anywhere in the kernel.
// Thread t owns 4 consecutive pixels of tile row t/8 (float4 stores, 128 B per 8 lanes).
// Sampling arithmetic = torch affine_grid + grid_sample(bilinear, zeros, align_corners=True) on the
// (Hp, Wp) frame, the frame itself being the edge-replicated (pad) and optionally h-flipped source.
template <int CH, bool VEC>
__global__ __launch_bounds__(kThreads, EQA_ACTION_WAVES) void group_action_kernel(const ActionArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kPlane = kBox * kLdsStride;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably uniform: row math stays on the SALU
  const int n = (int)blockIdx.z * kXcd + (int)(blockIdx.x & (kXcd - 1));
  if (n >= a.n_out) return;
  const int j0 = (int)(blockIdx.x >> 3) * kTile, i0 = (int)blockIdx.y * kTile;

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

  // frame column of output column j (post-flip: hflip of the rotated frame, then the crop)
  auto frame_x = [&](int j) { return flip_dst ? (a.Wp - 1 - (a.left + j)) : (a.left + j); };

  // ---- source window of the tile.  The map is affine in the normalised coords, so the extremes are sums of
  // per-axis extremes; a 1e-3 px guard covers the rounding difference to the per-pixel evaluation below.
  const int i1 = min(i0 + kTile - 1, a.OH - 1), j1 = min(j0 + kTile - 1, a.OW - 1);
  const float xa = lin_m1_p1(frame_x(j0), a.Wp, a.step_x), xb = lin_m1_p1(frame_x(j1), a.Wp, a.step_x);
  const float ya = lin_m1_p1(a.top + i0, a.Hp, a.step_y), yb = lin_m1_p1(a.top + i1, a.Hp, a.step_y);
  auto fmn = [](float p, float q) { return p < q ? p : q; };
  auto fmx = [](float p, float q) { return p > q ? p : q; };
  const float gx_lo = fmn(t0 * xa, t0 * xb) + fmn(t1 * ya, t1 * yb) + t2;
  const float gx_hi = fmx(t0 * xa, t0 * xb) + fmx(t1 * ya, t1 * yb) + t2;
  const float gy_lo = fmn(t3 * xa, t3 * xb) + fmn(t4 * ya, t4 * yb) + t5;
  const float gy_hi = fmx(t3 * xa, t3 * xb) + fmx(t4 * ya, t4 * yb) + t5;
  const float minx_f = (gx_lo + 1.0f) * a.half_w - 1e-3f, maxx_f = (gx_hi + 1.0f) * a.half_w + 1e-3f;
  const float miny_f = (gy_lo + 1.0f) * a.half_h - 1e-3f, maxy_f = (gy_hi + 1.0f) * a.half_h + 1e-3f;
  // keep at most one ring of off-frame (zero) pixels
  const int x_lo = (int)floorf(fmx(minx_f, -1.0f)), y_lo = (int)floorf(fmx(miny_f, -1.0f));
  // at least 2x2 so the clamped neighbour reads of fully off-frame pixels stay inside staged data
  const int x_hi = max((int)floorf(fmn(maxx_f, (float)(a.Wp - 1))) + 1, x_lo + 1);
  const int y_hi = max((int)floorf(fmn(maxy_f, (float)(a.Hp - 1))) + 1, y_lo + 1);
  const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
  const bool use_lds = (bw <= kBox) && (bh <= kBox) && !a.force_direct;

  // ---- per-thread output pixels
  const int r = tid >> 3, q = tid & 7;
  const int i = i0 + r, jb = j0 + 4 * q;
  int lidx[4];         // LDS path: index of the north-west neighbour inside the staged window
  int gx0[4], gy0[4];  // direct path: frame coords of the north-west neighbour
  bool live[4];        // false: all four neighbours are off the frame -> exact zero
  float w00[4], w01[4], w10[4], w11[4];
  auto pixel_setup = [&](int pi, int pj) {
    const float yn = lin_m1_p1(a.top + pi, a.Hp, a.step_y);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // affine_grid: [xn, yn, 1] . theta^T ; grid_sample(align_corners=True): ((g + 1) / 2) * (size - 1)
      const float xn = lin_m1_p1(frame_x(pj + k), a.Wp, a.step_x);
      const float ix = ((t0 * xn + t1 * yn + t2) + 1.0f) * a.half_w;
      const float iy = ((t3 * xn + t4 * yn + t5) + 1.0f) * a.half_h;
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
      lidx[k] = ly * (CH * kLdsStride) + lx;
      w00[k] = wy0 * wx0;  // nw
      w01[k] = wy0 * wx1;  // ne
      w10[k] = wy1 * wx0;  // sw
      w11[k] = wy1 * wx1;  // se
    }
  };

  // the element's channel-map row (regular features) goes to LDS once
  int* s_cmap = reinterpret_cast<int*>(smem + CH * kPlane);
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
#pragma unroll 1
      for (int y = ya + ((wave - ya) & 3); y < EQA_ABL_YB(yb); y += 4) {
        const int fy = y_lo + y;
        const unsigned voff = (unsigned)(min(max(fy - a.pad, 0), a.H - 1) * a.W) * 4u + col_off;
        const unsigned lrow = (unsigned)(uintptr_t)(lptr_t)(smem + y * (CH * kLdsStride));
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

  for (int c0 = 0; c0 < a.C; c0 += CH) {
    float acc[CH][4];
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
          const float v = nw * w00[k] + ne * w01[k] + sw * w10[k] + se * w11[k];
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
        float v = (in00 ? v00 : 0.0f) * a00 + (in01 ? v01 : 0.0f) * a01 + (in10 ? v10 : 0.0f) * a10 + (in11 ? v11 : 0.0f) * a11;
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
              *reinterpret_cast<float4*>(o) = make_float4(acc[cc][0], acc[cc][1], acc[cc][2], acc[cc][3]);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (jb + k < a.OW) o[k] = acc[cc][k];
          }
        }
      }
    }
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
  const int n = (int)blockIdx.z * kXcd + (int)(blockIdx.x & (kXcd - 1));
  if (n >= a.n_out) return;
  const int j0 = (int)(blockIdx.x >> 3) * kTile, i0 = (int)blockIdx.y * kTile;
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
    const float ix = ((t0 * xn + t1 * yn + t2) + 1.0f) * a.half_w;
    const float iy = ((t3 * xn + t4 * yn + t5) + 1.0f) * a.half_h;
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
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!live[k]) continue;
      const float g = go[k];
      const float wx0 = 1.0f - wx1[k], wy0 = 1.0f - wy1[k];
      if (ANGLE) {
        const float nw = in[k][0] ? pl[off[k][0]] : 0.0f, ne = in[k][1] ? pl[off[k][1]] : 0.0f;
        const float sw = in[k][2] ? pl[off[k][2]] : 0.0f, se = in[k][3] ? pl[off[k][3]] : 0.0f;
        const float dix = wy0 * (ne - nw) + wy1[k] * (se - sw);
        const float diy = wx0 * (sw - nw) + wx1[k] * (se - ne);
        if (GRAD == 1) sum += g * (dix * armx[k] + diy * army[k]);
        else { sx[k] += g * dix; sy[k] += g * diy; }
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
      const size_t t = ((size_t)n * gridDim.y + blockIdx.y) * tiles_x + (blockIdx.x >> 3);
      a.partial[t * NS + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
    }
  }
}

template <int CH>
int launch_action_ch(const ActionArgs& a, bool vec, hipStream_t st) {
  const int tiles_x = (a.OW + kTile - 1) / kTile, tiles_y = (a.OH + kTile - 1) / kTile;
  const int groups = (a.n_out + kXcd - 1) / kXcd;
  if (tiles_y > 65535 || groups > 65535) return EQA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(kXcd * tiles_x), (unsigned)tiles_y, (unsigned)groups);
  const size_t lds = (size_t)CH * kBox * kLdsStride * sizeof(float) + (a.chan_map ? kMaxMapG * sizeof(int) : 0);
  if (vec)
    hipLaunchKernelGGL((group_action_kernel<CH, true>), grid, dim3(kThreads), lds, st, a);
  else
    hipLaunchKernelGGL((group_action_kernel<CH, false>), grid, dim3(kThreads), lds, st, a);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
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
  a.gout = nullptr; a.gsrc = nullptr; a.partial = nullptr;
  return EQA_OK;
}

int launch_action(const float* src, float* dst, const int32_t* gidx, const float* theta, const int32_t* flags,
                  const int32_t* chan_map, int E, int G, int n_out, int B, int C, int H, int W, int pad, int OH,
                  int OW, int top, int left, void* stream) {
  if (!dst && n_out != 0) return EQA_ERR_INVALID_ARG;
  ActionArgs a;
  const int rc = fill_action_args(a, src, dst, gidx, theta, flags, chan_map, E, G, n_out, B, C, H, W, pad, OH, OW, top, left);
  if (rc != EQA_OK) return rc;
  if (n_out == 0) return EQA_OK;
  const bool vec = (OW % 4 == 0) && (((uintptr_t)dst & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
#if EQA_FORCE_CH
  return launch_action_ch<EQA_FORCE_CH>(a, vec, st);
#else
  if (C % 3 == 0) return launch_action_ch<3>(a, vec, st);
  if (C % 2 == 0) return launch_action_ch<2>(a, vec, st);
  return launch_action_ch<1>(a, vec, st);
#endif
}

// ------------------------------------------------------------------------------------------------
// I3 + I4: group pooling (mean over channels and space per group slot) and orientation argmax
// ------------------------------------------------------------------------------------------------

constexpr int kPoolSplitTarget = 2048;  // blocks wanted in flight for the streaming pass

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid (splits, B).  Block (b, s) streams channels [s*cps, (s+1)*cps) of image b: each wave takes whole
// (channel, group) planes of HW contiguous floats with float4 loads, reduces them with shuffles and
// adds the plane sum to its own per-group accumulator in LDS (fp64: the cross-plane sum is the long one).
template <bool VEC>
__global__ __launch_bounds__(kThreads) void group_pool_partial_kernel(const float* __restrict__ feat,
                                                                     double* __restrict__ partial, int Cf, int G,
                                                                     int HW, int cps, int splits) {
  extern __shared__ __attribute__((aligned(16))) double acc[];  // [4 waves][G]
  const int s = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < 4 * G; k += kThreads) acc[k] = 0.0;
  __syncthreads();
  const int c_lo = s * cps, c_hi = min(c_lo + cps, Cf);
  const int planes = (c_hi - c_lo) * G;
  const float* base = feat + ((size_t)b * Cf + c_lo) * G * HW;
  for (int p = wave; p < planes; p += 4) {
    const float* pl = base + (size_t)p * HW;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (VEC) {
      const float4* p4 = reinterpret_cast<const float4*>(pl);
      const int n4 = HW >> 2;
      for (int k = lane; k < n4; k += 64) {
        const float4 t = p4[k];
        v0 += t.x; v1 += t.y; v2 += t.z; v3 += t.w;
      }
    } else {
      for (int k = lane; k < HW; k += 64) v0 += pl[k];
    }
    const float tot = wave_sum((v0 + v1) + (v2 + v3));
    if (lane == 0) acc[wave * G + (p % G)] += (double)tot;
  }
  __syncthreads();
  if (tid < G) partial[((size_t)b * splits + s) * G + tid] = (acc[tid] + acc[G + tid]) + (acc[2 * G + tid] + acc[3 * G + tid]);
}

// one wave per image: lanes g < G own one orientation each; argmax by butterfly shuffles with
// (value, index) pairs, smaller index winning ties == torch.argmax's first-maximum rule.
__device__ __forceinline__ void wave_argmax_store(float v, int g, int G, int32_t* out) {
  int idx = (g < G) ? g : 0x7fffffff;
  float val = (g < G) ? v : -INFINITY;
  // NaN handling as torch: a NaN is "greater" than everything; first NaN wins.
  bool isn = (g < G) && (v != v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(val, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    const bool on = __shfl_xor((int)isn, o, 64) != 0;
    bool take;
    if (on != isn) take = on;
    else if (on) take = oi < idx;
    else take = (ov > val) || (ov == val && oi < idx);
    if (take) { val = ov; idx = oi; isn = on; }
  }
  if (g == 0) *out = idx;
}

__global__ __launch_bounds__(kThreads) void group_pool_finalize_kernel(const double* __restrict__ partial,
                                                                      float* __restrict__ act,
                                                                      int32_t* __restrict__ gidx, int B, int G,
                                                                      int splits, double inv_count) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  for (int g0 = 0; g0 < G; g0 += 64) {  // G <= 64 in every supported group; loop keeps it general for act
    const int g = g0 + lane;
    float a = 0.f;
    if (g < G) {
      double sum = 0.0;
      for (int s = 0; s < splits; ++s) sum += partial[((size_t)b * splits + s) * G + g];
      a = (float)(sum * inv_count);
      act[(size_t)b * G + g] = a;
    }
    if (G <= 64 && gidx) wave_argmax_store(a, g, G, gidx + b);
  }
}

__global__ __launch_bounds__(kThreads) void group_argmax_kernel(const float* __restrict__ act, int32_t* __restrict__ gidx,
                                                               int B, int G) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const float a = lane < G ? act[(size_t)b * G + lane] : 0.f;
  wave_argmax_store(a, lane, G, gidx + b);
}

int pool_splits(int B, int Cf) {
  int s = (kPoolSplitTarget + B - 1) / B;
  if (s > Cf) s = Cf;
  if (s < 1) s = 1;
  return s;
}

// ------------------------------------------------------------------------------------------------
// Window sums: the exact linear shortcut for "last convolution -> group mean".
//   mean_{o,y,x} conv(h, W)[o,g,y,x] = (1/count) * sum_{c,u,v} (sum_o W[(o,g),c,u,v]) * S[c,u,v] + mean(bias),
//   S[c,u,v] = sum_{y<OH, x<OW} act(h[c, y+u, x+v]),  OH = H-k+1, OW = W-k+1,  act = relu?(scale*h + shift).
// One block per (image, channel) plane: the plane is read from HBM exactly once (float4), transformed and parked in
// LDS; column totals and the k-1 leading / trailing rows give the k row-window sums per column, then the same trick
// across columns gives the k*k outputs.  fp64 accumulation (the consumer compares orientations by tiny margins).
// HBM-bound: H*W*4 bytes per plane in, k*k*8 bytes out.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxWinK = 8;

template <bool VEC>
__global__ __launch_bounds__(kThreads) void window_sums_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int relu,
                                                              double* __restrict__ out, int C, int H, int W, int k) {
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  const int HW = H * W;
  float* sp = ws_smem;                                                   // [H*W] transformed plane
  double* cs = reinterpret_cast<double*>(ws_smem + ((HW + 3) & ~3));      // [k][W] row-window sums per column
  const int plane = blockIdx.x;
  const int c = plane % C;
  const float sc = scale ? scale[c] : 1.0f, sh = shift ? shift[c] : 0.0f;
  const float* p = x + (size_t)plane * HW;
  const int tid = threadIdx.x;
  auto act = [&](float v) {
    v = v * sc + sh;
    return (relu && v < 0.0f) ? 0.0f : v;
  };
  if (VEC) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float4* s4 = reinterpret_cast<float4*>(sp);
    for (int i = tid; i < (HW >> 2); i += kThreads) {
      float4 t = p4[i];
      t.x = act(t.x); t.y = act(t.y); t.z = act(t.z); t.w = act(t.w);
      s4[i] = t;
    }
  } else {
    for (int i = tid; i < HW; i += kThreads) sp[i] = act(p[i]);
  }
  __syncthreads();
  const int OH = H - k + 1, OW = W - k + 1;
  // column x: total over rows, minus the u leading and the (k-1-u) trailing rows -> rows [u, u+OH)
  for (int xcol = tid; xcol < W; xcol += kThreads) {
    double tot = 0.0;
    for (int y = 0; y < H; ++y) tot += (double)sp[y * W + xcol];
    double pre = 0.0;
    for (int u = 0; u < k; ++u) {
      double suf = 0.0;
      for (int y = u + OH; y < H; ++y) suf += (double)sp[y * W + xcol];
      cs[u * W + xcol] = tot - pre - suf;
      pre += (double)sp[u * W + xcol];
    }
  }
  __syncthreads();
  // output (u, v): columns [v, v+OW) of row-window u.  One thread per output; k*k <= 64 of them.
  if (tid < k * k) {
    const int u = tid / k, v = tid - u * k;
    double acc = 0.0;
    for (int xcol = v; xcol < v + OW; ++xcol) acc += cs[u * W + xcol];
    out[(size_t)plane * (k * k) + tid] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Channels-last (NHWC) companions: MIOpen's fp32 implicit-GEMM convolutions run natively in NHWC, so the inference path
// of the canonicalization network stays in that layout end to end (no NCHW<->NHWC transposes).
//  * bias_relu_nhwc_kernel: x[p][c] = max(x[p][c] + bias[c], 0) in place, float4 over channels.
//  * window_sums_nhwc: same S[b,c,u,v] as window_sums_kernel, for a (B,H,W,C) buffer.  Rows are cut into segments:
//    each of the 2(k-1) border rows alone, the interior in bands.  A segment kernel streams its rows once (a wave reads
//    one pixel's channels = contiguous floats per instruction) and emits, per channel, the segment total and the sums of
//    the k-1 leading / trailing columns.  Every window sum is then total - excluded rows - excluded columns + their
//    intersections (inclusion-exclusion), assembled per (image, channel) in fp64 by a small finalize kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bias_relu_nhwc_kernel(float* __restrict__ x, const float* __restrict__ bias,
                                                                 size_t n_vec, int C4) {
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n_vec; i += stride) {
    const int c4 = (int)(i % (size_t)C4);
    float4 v = reinterpret_cast<float4*>(x)[i];
    const float4 b = reinterpret_cast<const float4*>(bias)[c4];
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

constexpr int kWsMaxBorder = kMaxWinK - 1;  // k - 1 <= 7

// segment s of image b: rows [seg_y0(s), seg_y1(s)).  Output per (b, s, c): 1 + 2(k-1) floats
//   [0] total, [1 + j] column j, [1 + (k-1) + j] column W-k+1+j      (j < k-1), all over the segment's rows.
__device__ __forceinline__ void ws_segment_rows(int s, int H, int k, int nbands, int& y0, int& y1) {
  const int nb = k - 1;
  if (s < nb) { y0 = s; y1 = s + 1; return; }                       // top border rows
  if (s < 2 * nb) { y0 = H - nb + (s - nb); y1 = y0 + 1; return; }  // bottom border rows
  const int lo = nb, hi = H - nb;                                   // interior rows, cut into nbands bands
  const int band = s - 2 * nb, rows = hi - lo;
  y0 = lo + (int)(((long long)rows * band) / nbands);
  y1 = lo + (int)(((long long)rows * (band + 1)) / nbands);
}

__global__ __launch_bounds__(kThreads) void window_sums_nhwc_segment_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                           const float* __restrict__ shift, int relu,
                                                                           float* __restrict__ part, int C, int H, int W, int k,
                                                                           int nbands) {
  __shared__ float4 s_tot[4][64];  // only the totals need a cross-wave sum; a border column has ONE owner wave
  const int s = blockIdx.x, b = blockIdx.y;
  const int nseg = gridDim.x;
  int y0, y1;
  ws_segment_rows(s, H, k, nbands, y0, y1);
  const int nb = k - 1, nval = 1 + 2 * nb;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Q = C >> 2;  // channel quads
  const float4* xb = reinterpret_cast<const float4*>(x + (size_t)b * H * W * C);
  for (int q0 = 0; q0 < Q; q0 += 64) {
    const int q = q0 + lane;
    const bool on = q < Q;
    const int qq = on ? q : Q - 1;
    const float4 sc = scale ? reinterpret_cast<const float4*>(scale)[qq] : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh = shift ? reinterpret_cast<const float4*>(shift)[qq] : make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](int y, int xc) {
      float4 v = xb[((size_t)y * W + xc) * Q + qq];
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      return v;
    };
    float4 acc[1 + 2 * kWsMaxBorder];
#pragma unroll
    for (int i = 0; i < 1 + 2 * kWsMaxBorder; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = y0; y < y1; ++y) {
      // the 4 waves take pixels x = wave, wave+4, ...: each load instruction reads one pixel's channels, contiguous
      float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
      int xc = wave;
      for (; xc + 4 < W; xc += 8) {
        const float4 a = ld(y, xc), c2 = ld(y, xc + 4);
        t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w;
        t1.x += c2.x; t1.y += c2.y; t1.z += c2.z; t1.w += c2.w;
      }
      if (xc < W) { const float4 a = ld(y, xc); t0.x += a.x; t0.y += a.y; t0.z += a.z; t0.w += a.w; }
      acc[0].x += t0.x + t1.x; acc[0].y += t0.y + t1.y; acc[0].z += t0.z + t1.z; acc[0].w += t0.w + t1.w;
      // border columns (static j, wave-uniform owner): re-read from L1
#pragma unroll
      for (int j = 0; j < kWsMaxBorder; ++j) {
        if (j < nb) {
          if ((j & 3) == wave) { const float4 a = ld(y, j); acc[1 + j].x += a.x; acc[1 + j].y += a.y; acc[1 + j].z += a.z; acc[1 + j].w += a.w; }
          const int xr = W - nb + j;
          if ((xr & 3) == wave) {
            const float4 a = ld(y, xr);
            acc[1 + kWsMaxBorder + j].x += a.x; acc[1 + kWsMaxBorder + j].y += a.y;
            acc[1 + kWsMaxBorder + j].z += a.z; acc[1 + kWsMaxBorder + j].w += a.w;
          }
        }
      }
    }
    auto put = [&](int i, const float4& v) {  // value index i of this lane's 4 channels
      float* o = part + ((((size_t)b * nseg + s) * Q + q) * 4) * nval + i;
      o[0] = v.x; o[nval] = v.y; o[2 * nval] = v.z; o[3 * nval] = v.w;
    };
    __syncthreads();
    s_tot[wave][lane] = acc[0];
#pragma unroll
    for (int j = 0; j < kWsMaxBorder; ++j) {
      if (j < nb && on) {
        if ((j & 3) == wave) put(1 + j, acc[1 + j]);
        if (((W - nb + j) & 3) == wave) put(1 + nb + j, acc[1 + kWsMaxBorder + j]);
      }
    }
    __syncthreads();
    if (wave == 0 && on) {
      const float4 a = s_tot[0][lane], b2 = s_tot[1][lane], c2 = s_tot[2][lane], d = s_tot[3][lane];
      put(0, make_float4((a.x + b2.x) + (c2.x + d.x), (a.y + b2.y) + (c2.y + d.y), (a.z + b2.z) + (c2.z + d.z), (a.w + b2.w) + (c2.w + d.w)));
    }
  }
}

// part: (B, nseg, C, nval) -> out (B, C, k, k) fp64.  Block = one image x 32 channels.  For a fixed (image, segment) the
// block's 32 channels x nval values are ONE contiguous run of part, so thread t owns element t of that run (channel t / nval,
// value t % nval) and walks the segment axis: every load instruction is a contiguous row (the first version gave each
// thread a whole channel, 36-byte lane stride: 0.26 ms for 207 MB; this one: see DESIGN.md).  Fixed order, deterministic.
// Then one thread per channel assembles the k*k window sums from the totals and the 2(k-1) border-row segments.
constexpr int kFinCh = 32;
constexpr int kFinVals = 1 + 2 * kWsMaxBorder;

__global__ __launch_bounds__(kThreads) void window_sums_nhwc_finalize_kernel(const float* __restrict__ part, double* __restrict__ out,
                                                                            int B, int C, int k, int nseg) {
  __shared__ double s_tot[kFinCh * kFinVals];
  __shared__ float s_brd[2 * kWsMaxBorder][kFinCh * kFinVals];
  const int b = blockIdx.y;
  const int c0 = blockIdx.x * kFinCh;
  const int nch = min(kFinCh, C - c0);
  const int nb = k - 1, nval = 1 + 2 * nb;
  const int run = nch * nval;  // contiguous elements of this block in one (image, segment) row
  const float* base = part + ((size_t)b * nseg * C + c0) * nval;
  const size_t seg_stride = (size_t)C * nval;
  for (int e = threadIdx.x; e < run; e += kThreads) {
    const float* q = base + e;
    double acc = 0.0;
    int s = 0;
    for (; s < 2 * nb; ++s) {  // border-row segments are needed individually as well
      const float v = q[(size_t)s * seg_stride];
      s_brd[s][e] = v;
      acc += (double)v;
    }
    for (; s + 8 <= nseg; s += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = q[(size_t)(s + j) * seg_stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (double)v[j];
    }
    for (; s < nseg; ++s) acc += (double)q[(size_t)s * seg_stride];
    s_tot[e] = acc;
  }
  __syncthreads();
  const int cl = threadIdx.x;
  if (cl >= nch) return;
  const int c = c0 + cl;
  auto P = [&](int s, int i) { return (double)s_brd[s][cl * nval + i]; };
  const double tot = s_tot[cl * nval];
  double col[2 * kWsMaxBorder];
#pragma unroll
  for (int j = 0; j < 2 * kWsMaxBorder; ++j) col[j] = j < 2 * nb ? s_tot[cl * nval + 1 + j] : 0.0;
  // window (u, v) keeps rows [u, H-nb+u) and columns [v, W-nb+v): it excludes top border rows r < u, bottom border rows
  // r >= u (of the nb bottom rows), left border columns j < v and right border columns j >= v
  for (int u = 0; u < k; ++u) {
    for (int v = 0; v < k; ++v) {
      double a = tot;
      for (int r = 0; r < nb; ++r) {
        if (r < u) a -= P(r, 0);
        if (r >= u) a -= P(nb + r, 0);
      }
      for (int j = 0; j < nb; ++j) {
        const bool l_ex = j < v, r_ex = j >= v;
        if (l_ex) a -= col[j];
        if (r_ex) a -= col[nb + j];
        for (int r = 0; r < nb; ++r) {  // excluded rows x excluded columns were subtracted twice
          if (r < u) a += (l_ex ? P(r, 1 + j) : 0.0) + (r_ex ? P(r, 1 + nb + j) : 0.0);
          if (r >= u) a += (l_ex ? P(nb + r, 1 + j) : 0.0) + (r_ex ? P(nb + r, 1 + nb + j) : 0.0);
        }
      }
      out[((size_t)b * C + c) * (k * k) + u * k + v] = a;
    }
  }
}

// The GEMV that follows the window sums (pooling.window_sums_to_activations; reference: the last convolution followed by
// torch.mean over (C, H', W') -- escnn_networks.py:115, custom_equivariant_networks.py:91):
//   act[b][e] = float( scale * sum_j S[b][j] * Wm[e][j] + shift ),  S fp64 (B, K), Wm fp64 (E, K), E <= 16.
// One block per image, fixed-order tree reduction (deterministic).  rocBLAS' dgemm takes 0.2 ms for this 256 x 6400 x 8 shape.
constexpr int kGemvMaxE = 16;

__global__ __launch_bounds__(kThreads) void sums_gemv_kernel(const double* __restrict__ S, const double* __restrict__ Wm,
                                                            float* __restrict__ act, int K, int E, double scale, double shift) {
  __shared__ double s_red[kThreads / 64][kGemvMaxE];
  const int b = blockIdx.x;
  const double* sb = S + (size_t)b * K;
  double acc[kGemvMaxE];
#pragma unroll
  for (int e = 0; e < kGemvMaxE; ++e) acc[e] = 0.0;
  for (int j = threadIdx.x; j < K; j += kThreads) {
    const double s = sb[j];
#pragma unroll
    for (int e = 0; e < kGemvMaxE; ++e)
      if (e < E) acc[e] += s * Wm[(size_t)e * K + j];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < kGemvMaxE; ++e) {
    double v = acc[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s_red[wave][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < E) {
    double v = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) v += s_red[w][threadIdx.x];
    act[(size_t)b * E + threadIdx.x] = (float)(v * scale + shift);
  }
}

// ------------------------------------------------------------------------------------------------
// Winograd F(m x m, 5x5) transforms, m = 2 or 4, for the 5x5 regular->regular layers of the canonicalization network
// (inference, channels-last).  Cook-Toom with N = m + 4 points per axis:
//   m = 2: {0, 1, -1, 2, -2, inf}             36 multiplies per 2x2 outputs = 9 per output   (direct: 25)
//   m = 4: {0, 1, -1, 2, -2, 1/2, -1/2, inf}  64 multiplies per 4x4 outputs = 4 per output
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A
// B^T and A^T have only small integers / dyadic fractions (exact in fp32); G is rational and is applied to the static
// filters offline in fp64 (images/canonicalization_networks/winograd.py).  The channel contraction of the N*N
// transformed planes is a plain strided-batched fp32 GEMM (library); these kernels are the HBM-bound ends:
//   input  : x (nimg, H, W, C) -> V (tiles, N*N, C),  tile (ty,tx) = rows m*ty..m*ty+N-1, cols m*tx..m*tx+N-1
//   output : M (tiles, N*N, C) -> y (nimg, OH, OW, C) = [relu](A^T M A + bias),  OH = H-4, OW = W-4 (multiples of m)
// fp32 error against an fp64 convolution, 256 channels: m = 2: 7e-6 of max|y|, m = 4: 9e-6 (direct fp32: 3e-7).
//
// Layout: V and M are tile-major -- every tile owns one contiguous N*N*C run, and plane `a` is the strided matrix
// V[:, a, :] (row stride N*N*C) the batched GEMM consumes.  (Measured for m = 2 against N*N dense planes: input
// transform 1236 vs 1383 us, GEMM and output equal.)
// One thread = one channel of a tile: all accesses are contiguous over channels.
// ------------------------------------------------------------------------------------------------
// t = B^T d
__device__ __forceinline__ void wino_bt(const float (&d)[6], float (&t)[6]) {
  t[0] = 4.0f * d[0] - 5.0f * d[2] + d[4];
  t[1] = 4.0f * (d[1] + d[2]) - (d[3] + d[4]);
  t[2] = 4.0f * (d[2] - d[1]) + (d[3] - d[4]);
  t[3] = 2.0f * (d[3] - d[1]) + (d[4] - d[2]);
  t[4] = 2.0f * (d[1] - d[3]) + (d[4] - d[2]);
  t[5] = 4.0f * d[1] - 5.0f * d[3] + d[5];
}
__device__ __forceinline__ void wino_bt(const float (&d)[8], float (&t)[8]) {
  const float e1 = d[2] + d[6] - 4.25f * d[4], o1 = d[1] + d[5] - 4.25f * d[3];
  const float e2 = 0.25f * d[2] - 1.25f * d[4] + d[6], o2 = 0.5f * d[1] - 2.5f * d[3] + 2.0f * d[5];
  const float e3 = 4.0f * d[2] - 5.0f * d[4] + d[6], o3 = 2.0f * d[1] - 2.5f * d[3] + 0.5f * d[5];
  t[0] = (d[6] - d[0]) + 5.25f * (d[2] - d[4]);
  t[1] = e1 + o1;
  t[2] = e1 - o1;
  t[3] = e2 + o2;
  t[4] = e2 - o2;
  t[5] = e3 + o3;
  t[6] = e3 - o3;
  t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
}
// y = A^T m
__device__ __forceinline__ void wino_at(const float (&m)[6], float (&y)[2]) {
  y[0] = (m[0] + m[1]) + (m[2] + m[3]) + m[4];
  y[1] = (m[1] - m[2]) + 2.0f * (m[3] - m[4]) + m[5];
}
__device__ __forceinline__ void wino_at(const float (&m)[8], float (&y)[4]) {
  const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4], s3 = m[5] + m[6], d3 = m[5] - m[6];
  y[0] = (m[0] + s1) + (s2 + s3);
  y[1] = d1 + 2.0f * d2 + 0.5f * d3;
  y[2] = s1 + 4.0f * s2 + 0.25f * s3;
  y[3] = (d1 + m[7]) + 8.0f * d2 + 0.125f * d3;
}

#ifdef EQA_WABL_NOLOAD
#define WINO_LD(ptr, k) ((float)(threadIdx.x + (k)))
#else
#define WINO_LD(ptr, k) (*(ptr))
#endif
#ifdef EQA_WABL_NOSTORE
#define WINO_ST(ptr, v) do { if ((v) == 1.2345e-30f) *(ptr) = (v); } while (0)
#else
#define WINO_ST(ptr, v) (*(ptr) = (v))
#endif
#ifndef EQA_WINO_BLOCKS
#define EQA_WINO_BLOCKS 4096
#endif
// Optional on-the-fly activation of the INPUT: d = relu(x + in_bias[c]) (the previous layer's folded bias / batch-norm
// and ReLU), which removes a separate pass over the previous feature map.
__device__ __forceinline__ float wino_act(float v, float ib, int in_relu) {
  v += ib;
  return (in_relu && v < 0.0f) ? 0.0f : v;
}

// One block = a strip of consecutive tiles in one tile row x 256 channels (thread = channel).  V = B^T d B is evaluated
// as (B^T d) B: the vertical transform u[:, col] = B^T d[:, col] depends only on the input column, so it is computed
// once per column and the N-column window slides by m per tile: m*N loads per tile instead of N*N.  The next tile's
// columns are requested before the current tile's N*N stores are issued (the memory pipeline is in-order per CU).
template <int N>
__global__ __launch_bounds__(kThreads) void winograd_k5_input_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                                    const float* __restrict__ in_bias, int in_relu,
                                                                    int H, int W, int C, int TY, int TX, int nstrip,
                                                                    int strip_len, size_t nwork) {
  constexpr int MT = N - 4;
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  // XCD-aware order: block b runs on XCD b % 8, so XCD k is given the contiguous work range [k*nwork/8, (k+1)*nwork/8)
  // (bijective form for nwork % 8 != 0).  Input pixels are shared by overlapping tiles; with neighbouring strips on one
  // XCD the overlap is served by that XCD's L2 instead of being re-fetched from HBM (measured, m = 2, tiles dealt
  // round-robin: 3.6 GB fetched per 0.55 GB of input; with this order 0.55 GB).
  const size_t bid = blockIdx.x;
  const size_t q8 = nwork / kXcd, r8 = nwork % kXcd, xcd = bid % kXcd;
  const size_t work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / kXcd;  // (img*TY + ty)*nstrip + s
  const int s = (int)(work % nstrip);
  const size_t r = work / nstrip;  // img*TY + ty
  const int ty = (int)(r % TY);
  const size_t img = r / TY;
  const int tx0 = s * strip_len;
  const int tx1 = min(TX, tx0 + strip_len);
  const float* p = x + ((img * H + MT * ty) * (size_t)W + MT * tx0) * C + c;
  const float ib = in_bias ? in_bias[c] : 0.0f;
  float u[N][N];  // u[i][k] = (B^T d)[i][window column k]
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float d[N], t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = wino_act(WINO_LD(p + ((size_t)i * W + k) * C, i * N + k), ib, in_relu);
    wino_bt(d, t);
#pragma unroll
    for (int i = 0; i < N; ++i) u[i][k] = t[i];
  }
  float* vout = V + (r * TX + tx0) * (size_t)(N * N) * C + c;
  for (int tx = tx0; tx < tx1; ++tx) {
    float nx[MT][N];
    const bool more = tx + 1 < tx1;  // uniform
    if (more) {
      const float* pn = p + (size_t)(MT * (tx - tx0) + N) * C;
#pragma unroll
      for (int k = 0; k < MT; ++k)
#pragma unroll
        for (int i = 0; i < N; ++i) nx[k][i] = WINO_LD(pn + ((size_t)i * W + k) * C, i * MT + k + tx);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float o[N];
      wino_bt(u[i], o);
#pragma unroll
      for (int j = 0; j < N; ++j) WINO_ST(vout + (size_t)(i * N + j) * C, o[j]);
    }
    vout += (size_t)(N * N) * C;
    if (more) {
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < N - MT; ++k) u[i][k] = u[i][k + MT];
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        float d[N], t[N];
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = wino_act(nx[k][i], ib, in_relu);
        wino_bt(d, t);
#pragma unroll
        for (int i = 0; i < N; ++i) u[i][N - MT + k] = t[i];
      }
    }
  }
}

// o = A^T M A (+ bias, ReLU) of one tile and channel; mp -> M[tile][0][c]
template <int N>
__device__ __forceinline__ void wino_tile_out(const float* __restrict__ mp, int C, float b, int relu, float (&o)[N - 4][N - 4]) {
  constexpr int MT = N - 4;
  float sc[MT][N];  // sc[r][j] = (A^T M)[r][j]
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float m[N], y[MT];
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = mp[(size_t)(i * N + j) * C];
    wino_at(m, y);
#pragma unroll
    for (int r = 0; r < MT; ++r) sc[r][j] = y[r];
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
    wino_at(sc[r], o[r]);
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      o[r][q] += b;
      if (relu) o[r][q] = fmaxf(o[r][q], 0.0f);
    }
  }
}

template <int N>
__global__ __launch_bounds__(kThreads) void winograd_k5_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                                     int relu, float* __restrict__ y, int OH, int OW, int C,
                                                                     int TY, int TX) {
  constexpr int MT = N - 4;
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t tile = blockIdx.x;
  const int tx = (int)(tile % TX);
  const size_t r = tile / TX;
  const int ty = (int)(r % TY);
  const size_t img = r / TY;
  float o[MT][MT];
  wino_tile_out<N>(M + tile * (size_t)(N * N) * C + c, C, bias ? bias[c] : 0.0f, relu, o);
  float* q = y + ((img * OH + MT * ty) * (size_t)OW + MT * tx) * C + c;
#pragma unroll
  for (int rr = 0; rr < MT; ++rr)
#pragma unroll
    for (int qq = 0; qq < MT; ++qq) q[((size_t)rr * OW + qq) * C] = o[rr][qq];
}

// Output transform fused with the window-sum segments of the NEXT (last, linearised) layer: instead of writing the
// (nimg, OH, OW, C) activation and re-reading it, every output row becomes one "segment" in the format of
// window_sums_nhwc_finalize_kernel -- per channel [row total, first NB columns, last NB columns] -- with the segment
// order that kernel expects: rows 0..NB-1, rows OH-NB..OH-1, then the interior rows.  One block = one tile row (m
// output rows) of one image x 256 channels, looping over the TX tiles.  NB = k_last - 1 must be a multiple of m.
template <int N, int NB>
__global__ __launch_bounds__(kThreads) void winograd_k5_output_sums_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                                          int relu, float* __restrict__ part, int OH, int OW,
                                                                          int C, int TY, int TX, int nseg) {
  constexpr int MT = N - 4;
  static_assert(NB % MT == 0, "border width must be whole tiles");
  const int c = blockIdx.y * kThreads + threadIdx.x;
  if (c >= C) return;
  const int ty = blockIdx.x % TY;
  const size_t img = blockIdx.x / TY;
  const float b = bias ? bias[c] : 0.0f;
  constexpr int NV = 1 + 2 * NB;
  float acc[MT][NV];
#pragma unroll
  for (int r = 0; r < MT; ++r)
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[r][i] = 0.0f;
  const float* mrow = M + (img * TY + ty) * (size_t)TX * (N * N) * C + c;
  constexpr int HB = NB / MT;  // border tiles per side
  // left border tiles (static column indices), interior, right border tiles
#pragma unroll
  for (int t = 0; t < HB; ++t) {
    float o[MT][MT];
    wino_tile_out<N>(mrow + (size_t)t * (N * N) * C, C, b, relu, o);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int q = 0; q < MT; ++q) { acc[r][0] += o[r][q]; acc[r][1 + MT * t + q] += o[r][q]; }
  }
  for (int tx = HB; tx < TX - HB; ++tx) {
    float o[MT][MT];
    wino_tile_out<N>(mrow + (size_t)tx * (N * N) * C, C, b, relu, o);
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      float a = o[r][0];
#pragma unroll
      for (int q = 1; q < MT; ++q) a += o[r][q];
      acc[r][0] += a;
    }
  }
#pragma unroll
  for (int t = 0; t < HB; ++t) {
    float o[MT][MT];
    wino_tile_out<N>(mrow + (size_t)(TX - HB + t) * (N * N) * C, C, b, relu, o);
#pragma unroll
    for (int r = 0; r < MT; ++r)
#pragma unroll
      for (int q = 0; q < MT; ++q) { acc[r][0] += o[r][q]; acc[r][1 + NB + MT * t + q] += o[r][q]; }
  }
#pragma unroll
  for (int r = 0; r < MT; ++r) {
    const int y = MT * ty + r;
    const int seg = y < NB ? y : (y >= OH - NB ? NB + (y - (OH - NB)) : 2 * NB + (y - NB));
    float* o = part + ((img * nseg + seg) * (size_t)C + c) * NV;
#pragma unroll
    for (int i = 0; i < NV; ++i) o[i] = acc[r][i];
  }
}

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
//  * Epilogue: bias + ReLU on the accumulators, 128 B (32 channels) per half-wave store.
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

// Epilogue of accumulator register i of a finished tile.  C/D map of the 32x32 MFMA: column = lane & 31,
// row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5): register i of acc0 holds channels 0..31 of pixel r (lanes 0-31) and of
// pixel r+4 (lanes 32-63), acc1 the same for channels 32..63.  Stored as they are, every instruction writes two 128-byte
// pieces 4 pixels apart (measured: 960 us, the store path being the limit whatever the schedule).  v_permlane32_swap
// exchanges acc0's upper half with acc1's lower half: one register then holds the 64 contiguous channels of pixel r, the
// other those of pixel r+4 -- one 256-byte run per store.
template <bool MASKED>
__device__ __forceinline__ void lift_store_reg(int i, const f32x16& p0, const f32x16& p1, float bias_l, float lo,
                                               float* __restrict__ o, int Cout, int rows_left) {
  const int row = (i & 3) + 8 * (i >> 2);
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0[i]), __float_as_uint(p1[i]), false, false);
  const float v0 = fmaxf(__uint_as_float(sw[0]) + bias_l, lo), v1 = fmaxf(__uint_as_float(sw[1]) + bias_l, lo);
#ifdef EQA_LABL_NOSTORE
  if (v0 == 1.2345e-30f) {
    o[(size_t)row * Cout] = v0;
    o[(size_t)(row + 4) * Cout] = v1;
  }
#else
  if (!MASKED || row < rows_left) o[(size_t)row * Cout] = v0;
  if (!MASKED || row + 4 < rows_left) o[(size_t)(row + 4) * Cout] = v1;
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
  constexpr int PER = NM / 16;     // MFMAs between two epilogue slices
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
  const float bias_l = bias ? bias[slice * 64 + lane] : 0.0f;  // after the half swap lane l holds channel slice*64 + l
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
    // tile's bias / ReLU / stores are issued in 16 slices between this tile's MFMAs.
    if (have_prev) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int m = PER * i; m < PER * i + PER; ++m) {
          const int step = m >> 1;
          if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step >> 3][step & 7], b1[step], acc1, 0, 0, 0);
          else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step >> 3][step & 7], b0[step], acc0, 0, 0, 0);
        }
        lift_store_reg<MASKED>(i, p0, p1, bias_l, lo, po, Cout, p_left);
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // PER MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    // the slice's VALU (swap, add, max, address)
        __builtin_amdgcn_sched_group_barrier(0x040, 2, 0);    // its two stores
      }
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int step = m >> 1;
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step >> 3][step & 7], b1[step], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[step >> 3][step & 7], b0[step], acc0, 0, 0, 0);
      }
    }
    lift_tile_pos<MASKED>(tile, tiles_per_row, OW, rowid, ox0);
    p0 = acc0;
    p1 = acc1;
    po = y + ((size_t)rowid * OW + ox0) * Cout + slice * 64 + lane;
    p_left = OW - (int)ox0;
    have_prev = true;
    if (next >= ntiles) break;
    lift_stage_store<KH>(lds[buf ^ 1], n_el, g);
    __syncthreads();  // everyone has read lds[buf ^ 1] two tiles ago (before the previous barrier) and lds[buf] above
    buf ^= 1;
    tile = next;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) lift_store_reg<MASKED>(i, p0, p1, bias_l, lo, po, Cout, p_left);
}

// ------------------------------------------------------------------------------------------------
// P4 / P3: SO(3) action on point clouds, batched Gram-Schmidt
// ------------------------------------------------------------------------------------------------

template <bool VEC>
__global__ __launch_bounds__(kThreads) void so3_rotate_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                             float* __restrict__ y, int N, int transpose) {
  const int b = blockIdx.y;
  const float* Rb = R + (size_t)b * 9;
  float m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = transpose ? Rb[(k % 3) * 3 + k / 3] : Rb[k];
  const float* xb = x + (size_t)b * 3 * N;
  float* yb = y + (size_t)b * 3 * N;
  if (VEC) {
    const int n4 = N >> 2;
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n4) return;
    const float4 p0 = reinterpret_cast<const float4*>(xb)[k];
    const float4 p1 = reinterpret_cast<const float4*>(xb + N)[k];
    const float4 p2 = reinterpret_cast<const float4*>(xb + 2 * (size_t)N)[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      // same accumulation order as a k-ordered dot product: (m0*x0 + m1*x1) + m2*x2
      float4 o;
      o.x = m[r * 3] * p0.x + m[r * 3 + 1] * p1.x + m[r * 3 + 2] * p2.x;
      o.y = m[r * 3] * p0.y + m[r * 3 + 1] * p1.y + m[r * 3 + 2] * p2.y;
      o.z = m[r * 3] * p0.z + m[r * 3 + 1] * p1.z + m[r * 3 + 2] * p2.z;
      o.w = m[r * 3] * p0.w + m[r * 3 + 1] * p1.w + m[r * 3 + 2] * p2.w;
      reinterpret_cast<float4*>(yb + (size_t)r * N)[k] = o;
    }
  } else {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= N) return;
    const float p0 = xb[k], p1 = xb[N + k], p2 = xb[2 * (size_t)N + k];
#pragma unroll
    for (int r = 0; r < 3; ++r) yb[(size_t)r * N + k] = m[r * 3] * p0 + m[r * 3 + 1] * p1 + m[r * 3 + 2] * p2;
  }
}

__global__ __launch_bounds__(kThreads) void gram_schmidt_kernel(const float* __restrict__ v, float* __restrict__ out, int B) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  const float* p = v + (size_t)b * 9;
  float a0 = p[0], a1 = p[1], a2 = p[2], b0 = p[3], b1 = p[4], b2 = p[5], c0 = p[6], c1 = p[7], c2 = p[8];
  // e1 = a / |a|   (torch.norm: sqrt of the sum of squares; division, not rsqrt, to stay on the reference's rounding)
  float n = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  a0 /= n; a1 /= n; a2 /= n;
  float d = b0 * a0 + b1 * a1 + b2 * a2;
  b0 -= d * a0; b1 -= d * a1; b2 -= d * a2;
  n = sqrtf(b0 * b0 + b1 * b1 + b2 * b2);
  b0 /= n; b1 /= n; b2 /= n;
  const float d1 = c0 * a0 + c1 * a1 + c2 * a2;
  const float d2 = c0 * b0 + c1 * b1 + c2 * b2;  // classical GS: both projections use the ORIGINAL c
  c0 = c0 - d1 * a0 - d2 * b0;
  c1 = c1 - d1 * a1 - d2 * b1;
  c2 = c2 - d1 * a2 - d2 * b2;
  n = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
  c0 /= n; c1 /= n; c2 /= n;
  float* o = out + (size_t)b * 9;
  o[0] = a0; o[1] = a1; o[2] = a2; o[3] = b0; o[4] = b1; o[5] = b2; o[6] = c0; o[7] = c1; o[8] = c2;
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
constexpr int kAaBand = 8;

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
  for (int idx = threadIdx.x; idx < nrows * OW; idx += kThreads) {
    const int ry = idx / OW, ox = idx - ry * OW;
    const float* row = src + (size_t)(ybeg + ry) * W;
    const int xs = x0[ox];
    float acc = 0.0f;
    for (int j = 0; j < K; ++j) acc += wx[ox * K + j] * row[min(xs + j, W - 1)];
    aa_tmp[ry * OW + ox] = acc;
  }
  __syncthreads();
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

// ------------------------------------------------------------------------------------------------
// P1 + P2: fused VNSmall forward (eval mode, mean pooling): kNN graph -> cross edge features -> VN linear / VN batch-norm
// / direction-gated ReLU (3->21) -> mean over neighbours -> (21->21) + VN batch-norm -> (21->4) -> mean over points.
// Reference: pointcloud/canonicalization_networks/equivariant_networks.py:15-76 (knn, get_graph_feature_cross),
// :128-150 (VNSmall.forward), vector_neuron_layers.py:251-273, :303-324.
// The reference materialises ten (B,21,3,N,k) tensors (5 MB per cloud each); here a cloud is 12 KB in LDS and every
// intermediate lives in registers: one thread = one point, its k nearest neighbours kept as a sorted register list
// while it streams over the cloud (LDS broadcast reads), then its k edges are pushed through the layers one by one.
// Packed parameter buffer (floats), eval-mode batch-norms pre-folded to scale/shift of the vector NORM:
//   [0,63) pos.Wf(21x3)  [63,126) pos.Wd  [126,147) pos.bn scale  [147,168) pos.bn shift
//   [168,609) c1.Wf(21x21)  [609,1050) c1.Wd  [1050,1071) c1.bn scale  [1071,1092) c1.bn shift
//   [1092,1113) bn1 scale  [1113,1134) bn1 shift
//   [1134,1218) c2.Wf(4x21)  [1218,1302) c2.Wd  [1302,1306) c2.bn scale  [1306,1310) c2.bn shift
// ------------------------------------------------------------------------------------------------
constexpr int kVnC = 21, kVnK = 20, kVnThreads = 128, kVnParams = 1310;
#ifndef EQA_VN_MIN_BLOCKS
#define EQA_VN_MIN_BLOCKS 3  // waves per SIMD the register allocation must allow (measured 2: 481 k, 3: 531 k, 4: 416 k clouds/s)
#endif
constexpr int kVnQueue = 12;  // pending kNN candidates per thread (LDS, 8 bytes each)
constexpr float kVnEps = 1e-6f;

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ float dot3(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// VN batch-norm (eval): q * BN(|q| + EPS) / (|q| + EPS);  then direction-gated ReLU with slope 0
__device__ __forceinline__ V3 vn_bn(V3 q, float scale, float shift) {
  const float n = sqrtf(dot3(q, q)) + kVnEps;
  const float r = (n * scale + shift) / n;
  return v3(q.x * r, q.y * r, q.z * r);
}
__device__ __forceinline__ V3 vn_relu(V3 q, const V3& d) {
  const float dp = dot3(q, d);
  if (dp < 0.0f) {
    const float t = dp / (dot3(d, d) + kVnEps);
    q.x -= t * d.x; q.y -= t * d.y; q.z -= t * d.z;
  }
  return q;
}

__global__ __launch_bounds__(kVnThreads, EQA_VN_MIN_BLOCKS) void vnsmall_fwd_kernel(const float* __restrict__ x, const float* __restrict__ prm,
                                                                 float* __restrict__ partial, int N, int nblk) {
  extern __shared__ __attribute__((aligned(16))) float vn_smem[];
  float4* pts = reinterpret_cast<float4*>(vn_smem);  // [Npad] (x, y, z, |p|^2): one ds_read_b128 per candidate
  __shared__ float s_part[kVnThreads / 64][12];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float* xb = x + (size_t)b * 3 * N;
  const int Npad = (N + 3) & ~3;
  for (int i = tid; i < Npad; i += kVnThreads) {
    if (i < N) {
      const float a = xb[i], c = xb[N + i], d = xb[2 * (size_t)N + i];
      pts[i] = make_float4(a, c, d, a * a + c * c + d * d);  // torch.sum(x**2, dim=1)
    } else {
      pts[i] = make_float4(0.f, 0.f, 0.f, INFINITY);  // padding: value -inf, never selected
    }
  }
  __syncthreads();
  const int n = blockIdx.x * kVnThreads + tid;
  const bool active = n < N;
  const int ni = active ? n : N - 1;
  const float4 c4 = pts[ni];
  const V3 ctr = v3(c4.x, c4.y, c4.z);
  const float cn = c4.w;

  // ---- kNN: k largest of  -|xj|^2 + 2 xi.xj - |xi|^2  (the reference's expansion, equivariant_networks.py:28-30),
  // kept sorted (descending) in registers; strict '>' so that the earlier index wins ties.
  // The sorted insertion is a ~100-instruction chain that the whole wave executes whenever ANY of its 64 points
  // accepts a candidate -- which is true for ~870 of the 1024 candidates although each point accepts only ~93 (first
  // version: 181 k VALU instructions per wave, ~145 k of them here, 11 % of the lanes doing useful work).  So a
  // candidate that beats the point's current 20th score is only APPENDED to a small per-thread queue in LDS (3
  // instructions), and the queues are drained -- in index order, each entry re-tested against the then-current
  // threshold, i.e. the same result as immediate insertion -- when any lane's queue could overflow on the next group:
  // ~185 chain executions per wave instead of ~870.
  float bv[kVnK];
  int bi[kVnK];
#pragma unroll
  for (int t = 0; t < kVnK; ++t) { bv[t] = -INFINITY; bi[t] = 0; }
  auto insert = [&](float val, int j) {
    if (val > bv[kVnK - 1]) {
      float cv = val;
      int ci = j;
#pragma unroll
      for (int t = 0; t < kVnK; ++t) {
        const bool sw = cv > bv[t];
        const float tv = bv[t];
        const int ti = bi[t];
        bv[t] = sw ? cv : tv;
        bi[t] = sw ? ci : ti;
        cv = sw ? tv : cv;
        ci = sw ? ti : ci;
      }
    }
  };
  auto score = [&](const float4& p) {
    const float inner = -2.0f * (ctr.x * p.x + ctr.y * p.y + ctr.z * p.z);
    return (-p.w - inner) - cn;
  };
  float2* queue = reinterpret_cast<float2*>(vn_smem + 4 * Npad) + tid;  // slot s of this thread: queue[s * kVnThreads]
  int cnt = 0;
  auto drain = [&]() {
#pragma unroll
    for (int s = 0; s < kVnQueue; ++s) {
      if (s < cnt) {
        const float2 e = queue[s * kVnThreads];
        insert(e.x, __float_as_int(e.y));
      }
    }
    cnt = 0;
  };
  auto offer = [&](float v, int j) {
    if (v > bv[kVnK - 1]) {
      queue[cnt * kVnThreads] = make_float2(v, __int_as_float(j));
      ++cnt;
    }
  };
  for (int j = 0; j < Npad; j += 4) {
    const float4 p0 = pts[j], p1 = pts[j + 1], p2 = pts[j + 2], p3 = pts[j + 3];
    offer(score(p0), j);
    offer(score(p1), j + 1);
    offer(score(p2), j + 2);
    offer(score(p3), j + 3);
    if (__any(cnt > kVnQueue - 4)) drain();  // wave-uniform
  }
  drain();

  // ---- conv_pos on the k edges + mean over neighbours
  V3 pooled[kVnC];
#pragma unroll
  for (int c = 0; c < kVnC; ++c) pooled[c] = v3(0.f, 0.f, 0.f);
  const float* Wf = prm;
  const float* Wd = prm + 63;
  const float* bsc = prm + 126;
  const float* bsh = prm + 147;
#pragma unroll 1
  for (int t = 0; t < kVnK; ++t) {
    // compiler barrier: otherwise the 168 loop-invariant scalar weights are hoisted out of the loop and, for want of
    // SGPRs, parked in VGPRs for its whole duration (the kernel then needs > 256 VGPRs: one wave per SIMD)
    asm volatile("" ::: "memory");
    // (dynamic t: pick the t-th neighbour index through a select chain, the list lives in registers)
    int j = bi[0];
#pragma unroll
    for (int u = 1; u < kVnK; ++u) j = (t == u) ? bi[u] : j;
    const float4 nb4 = pts[j];
    const V3 nb = v3(nb4.x, nb4.y, nb4.z);
    const V3 f0 = v3(nb.x - ctr.x, nb.y - ctr.y, nb.z - ctr.z);                                   // neighbour - centre
    const V3 f2 = v3(nb.y * ctr.z - nb.z * ctr.y, nb.z * ctr.x - nb.x * ctr.z, nb.x * ctr.y - nb.y * ctr.x);  // nbr x ctr
#pragma unroll
    for (int c = 0; c < kVnC; ++c) {
      const float a0 = Wf[c * 3], a1 = Wf[c * 3 + 1], a2 = Wf[c * 3 + 2];
      const float d0 = Wd[c * 3], d1 = Wd[c * 3 + 1], d2 = Wd[c * 3 + 2];
      V3 q = v3(a0 * f0.x + a1 * ctr.x + a2 * f2.x, a0 * f0.y + a1 * ctr.y + a2 * f2.y, a0 * f0.z + a1 * ctr.z + a2 * f2.z);
      const V3 d = v3(d0 * f0.x + d1 * ctr.x + d2 * f2.x, d0 * f0.y + d1 * ctr.y + d2 * f2.y, d0 * f0.z + d1 * ctr.z + d2 * f2.z);
      q = vn_relu(vn_bn(q, bsc[c], bsh[c]), d);
      pooled[c].x += q.x; pooled[c].y += q.y; pooled[c].z += q.z;
    }
  }
  const float inv_k = 1.0f / (float)kVnK;
#pragma unroll
  for (int c = 0; c < kVnC; ++c) { pooled[c].x *= inv_k; pooled[c].y *= inv_k; pooled[c].z *= inv_k; }

  // ---- conv1 (21->21) + its VN-BN + ReLU, then bn1; every output channel is folded into conv2's (21->4) two linear
  // maps as soon as it exists, so the 21 x 3 intermediate never has to be held in registers
  V3 q2[4], d2[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) { q2[o] = v3(0.f, 0.f, 0.f); d2[o] = v3(0.f, 0.f, 0.f); }
  {
    const float* W1f = prm + 168;
    const float* W1d = prm + 609;
    const float* s1 = prm + 1050;
    const float* t1 = prm + 1071;
    const float* s2 = prm + 1092;
    const float* t2 = prm + 1113;
    const float* W2f = prm + 1134;
    const float* W2d = prm + 1218;
#pragma unroll 1  // rolled: fully unrolled (882 scalar weights in flight) the kernel needs > 256 VGPRs
    for (int c = 0; c < kVnC; ++c) {
      asm volatile("" ::: "memory");
      V3 q = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 0.f);
#pragma unroll
      for (int a = 0; a < kVnC; ++a) {
        const float wf = W1f[c * kVnC + a], wd = W1d[c * kVnC + a];
        q.x += wf * pooled[a].x; q.y += wf * pooled[a].y; q.z += wf * pooled[a].z;
        d.x += wd * pooled[a].x; d.y += wd * pooled[a].y; d.z += wd * pooled[a].z;
      }
      q = vn_relu(vn_bn(q, s1[c], t1[c]), d);
      const V3 hc = vn_bn(q, s2[c], t2[c]);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const float wf = W2f[o * kVnC + c], wd = W2d[o * kVnC + c];
        q2[o].x += wf * hc.x; q2[o].y += wf * hc.y; q2[o].z += wf * hc.z;
        d2[o].x += wd * hc.x; d2[o].y += wd * hc.y; d2[o].z += wd * hc.z;
      }
    }
  }
  // ---- conv2's VN-BN + ReLU -> this point's contribution to the mean over points
  float outv[12];
  {
    const float* s3 = prm + 1302;
    const float* t3 = prm + 1306;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const V3 q = vn_relu(vn_bn(q2[c], s3[c], t3[c]), d2[c]);
      outv[c * 3] = active ? q.x : 0.f;
      outv[c * 3 + 1] = active ? q.y : 0.f;
      outv[c * 3 + 2] = active ? q.z : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const float sres = wave_sum_f(outv[c]);
    if ((tid & 63) == 0) s_part[tid >> 6][c] = sres;
  }
  __syncthreads();
  if (tid < 12) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < kVnThreads / 64; ++w) acc += s_part[w][tid];
    partial[((size_t)b * nblk + blockIdx.x) * 12 + tid] = acc;
  }
}

// (B, nblk, 12) partial sums -> (B,3,3): mean over the N points, first three of the four output channels
__global__ void vnsmall_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int B, int nblk, float inv_n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 9) return;
  const int b = i / 9, c = i - b * 9;
  float acc = 0.f;
  for (int k = 0; k < nblk; ++k) acc += partial[((size_t)b * nblk + k) * 12 + c];
  out[i] = acc * inv_n;
}

template <typename T>
int launch_nearest(const T* m, T* out, const int32_t* eidx, const float* rtheta, const int32_t* flags, int E,
                          int n_planes, int H, int W, int pad, int OH, int OW, int top, int left, int src_mod, void* stream) {
  if (!m || !out || !eidx || !rtheta || E <= 0 || n_planes < 0 || H <= 0 || W <= 0 || pad < 0 || OH <= 0 || OW <= 0 ||
      top < 0 || left < 0 || top + OH > H + 2 * pad || left + OW > W + 2 * pad || src_mod < 0)
    return EQA_ERR_INVALID_ARG;
  if (n_planes > 65535 || OH > 65535) return EQA_ERR_UNSUPPORTED;
  if (n_planes == 0) return EQA_OK;
  hipLaunchKernelGGL((nearest_action_kernel<T>), dim3((OW / 4 + kThreads) / kThreads, OH, n_planes), dim3(kThreads), 0,
                     (hipStream_t)stream, m, out, eidx, rtheta, flags, E, H, W, pad, OH, OW, top, left, src_mod);
  return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH;
}

// (f).4 -- n-body E(3) canonicalization: modified Gram-Schmidt and the per-row rigid action
// (nbody/canonicalization/euclidean_group.py:87-157).  Tiny tensors (nodes x 3): one thread per row.
__global__ __launch_bounds__(kThreads) void modified_gram_schmidt_kernel(const float* __restrict__ v, float* __restrict__ out, int B) {
  const int b = blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  const float* p = v + (size_t)b * 9;
  float a0 = p[0], a1 = p[1], a2 = p[2], b0 = p[3], b1 = p[4], b2 = p[5], c0 = p[6], c1 = p[7], c2 = p[8];
  float n = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  a0 /= n; a1 /= n; a2 /= n;
  float d = b0 * a0 + b1 * a1 + b2 * a2;
  b0 -= d * a0; b1 -= d * a1; b2 -= d * a2;
  n = sqrtf(b0 * b0 + b1 * b1 + b2 * b2);
  b0 /= n; b1 /= n; b2 /= n;
  d = c0 * a0 + c1 * a1 + c2 * a2;
  c0 -= d * a0; c1 -= d * a1; c2 -= d * a2;
  d = c0 * b0 + c1 * b1 + c2 * b2;  // modified GS: second projection uses the UPDATED third vector
  c0 -= d * b0; c1 -= d * b1; c2 -= d * b2;
  n = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
  c0 /= n; c1 /= n; c2 /= n;
  float* o = out + (size_t)b * 9;
  o[0] = a0; o[1] = a1; o[2] = a2; o[3] = b0; o[4] = b1; o[5] = b2; o[6] = c0; o[7] = c1; o[8] = c2;
}

// mode 0: out = x R + t            (invert_canonicalization :126-137; t may be NULL)
// mode 1: out = x R^T - t R^T      (canonicalize :108-124, the two products subtracted as the reference does)
__global__ __launch_bounds__(kThreads) void rigid_rows_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                             const float* __restrict__ t, float* __restrict__ out, int M, int mode) {
  const int m = blockIdx.x * kThreads + threadIdx.x;
  if (m >= M) return;
  const float* r = R + (size_t)m * 9;
  const float x0 = x[(size_t)m * 3], x1 = x[(size_t)m * 3 + 1], x2 = x[(size_t)m * 3 + 2];
  const float t0 = t ? t[(size_t)m * 3] : 0.f, t1 = t ? t[(size_t)m * 3 + 1] : 0.f, t2 = t ? t[(size_t)m * 3 + 2] : 0.f;
  float* o = out + (size_t)m * 3;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (mode == 0) {
      o[j] = (x0 * r[j] + x1 * r[3 + j] + x2 * r[6 + j]) + (j == 0 ? t0 : j == 1 ? t1 : t2);
    } else {
      const float a = x0 * r[3 * j] + x1 * r[3 * j + 1] + x2 * r[3 * j + 2];
      const float b = t0 * r[3 * j] + t1 * r[3 * j + 1] + t2 * r[3 * j + 2];
      o[j] = t ? a - b : a;
    }
  }
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? EQA_OK : EQA_ERR_LAUNCH; }

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
  if (!partial)
    hipLaunchKernelGGL((group_action_bwd_kernel<0, true>), grid, dim3(kThreads), 0, st, a);
  else if (grad_mode == 1 && grad_src)
    hipLaunchKernelGGL((group_action_bwd_kernel<1, true>), grid, dim3(kThreads), 0, st, a);
  else if (grad_mode == 1)
    hipLaunchKernelGGL((group_action_bwd_kernel<1, false>), grid, dim3(kThreads), 0, st, a);
  else if (grad_src)
    hipLaunchKernelGGL((group_action_bwd_kernel<2, true>), grid, dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL((group_action_bwd_kernel<2, false>), grid, dim3(kThreads), 0, st, a);
  return launch_status();
}

template <int N>
int launch_wino_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                      void* stream) {
  constexpr int MT = N - 4;
  if (!x || !V || nimg < 0 || H < N || W < N || C <= 0) return EQA_ERR_INVALID_ARG;
  if (((H - 4) % MT) || ((W - 4) % MT)) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int TY = (H - 4) / MT, TX = (W - 4) / MT;
  // strips: enough blocks to fill 256 CUs x their resident blocks, long enough that the N*N-load prologue is amortised
  // (m*N further loads per tile); measured for m = 2 at 64 x 92 x 92 x 256: whole rows 1112 us, 8-tile strips 1289 us
  const int cb = (C + kThreads - 1) / kThreads;
  const size_t rows = (size_t)nimg * TY;
  int nstrip = (int)((EQA_WINO_BLOCKS + rows * cb - 1) / (rows * cb));
  nstrip = std::max(1, std::min(nstrip, std::max(1, TX / 4)));
  const int strip_len = (TX + nstrip - 1) / nstrip;
  nstrip = (TX + strip_len - 1) / strip_len;
  const size_t nwork = rows * nstrip;
  if (nwork > 0x7fffffffULL || rows * TX > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((winograd_k5_input_kernel<N>), dim3((unsigned)nwork, cb), dim3(kThreads), 0, (hipStream_t)stream, x,
                     V, in_bias, in_relu, H, W, C, TY, TX, nstrip, strip_len, nwork);
  return launch_status();
}

template <int N>
int launch_wino_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                       void* stream) {
  constexpr int MT = N - 4;
  if (!M || !y || nimg < 0 || OH < MT || OW < MT || C <= 0) return EQA_ERR_INVALID_ARG;
  if ((OH % MT) || (OW % MT)) return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  const int TY = OH / MT, TX = OW / MT;
  const size_t tiles = (size_t)nimg * TY * TX;
  if (tiles > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((winograd_k5_output_kernel<N>), dim3((unsigned)tiles, (C + kThreads - 1) / kThreads), dim3(kThreads),
                     0, (hipStream_t)stream, M, bias, relu, y, OH, OW, C, TY, TX);
  return launch_status();
}

template <int N>
int launch_wino_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg, int OH,
                            int OW, int C, int k_next, void* stream) {
  constexpr int MT = N - 4;
  if (!M || !S || !workspace || nimg < 0 || OH < MT || OW < MT || C <= 0 || k_next <= 0) return EQA_ERR_INVALID_ARG;
  const int nb = k_next - 1;
  // border width in whole tiles, disjoint borders with an interior, same limits as eqa_window_sums_nhwc
  if ((OH % MT) || (OW % MT) || (nb != 4 && nb != 2) || (nb % MT) || OH < 2 * nb + MT || OW < 2 * nb + MT || k_next > kMaxWinK)
    return EQA_ERR_UNSUPPORTED;
  if (nimg == 0) return EQA_OK;
  if (nimg > 65535) return EQA_ERR_UNSUPPORTED;
  const int TY = OH / MT, TX = OW / MT;
  if ((size_t)nimg * TY > 0x7fffffffULL) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((size_t)nimg * TY), (C + kThreads - 1) / kThreads);
  const int nseg = OH;  // one segment per output row
  float* part = (float*)workspace;
  if (nb == 4) {
    hipLaunchKernelGGL((winograd_k5_output_sums_kernel<N, 4>), grid, dim3(kThreads), 0, st, M, bias, relu, part, OH, OW, C, TY, TX, nseg);
  } else {
    if constexpr (MT == 2)
      hipLaunchKernelGGL((winograd_k5_output_sums_kernel<N, 2>), grid, dim3(kThreads), 0, st, M, bias, relu, part, OH, OW, C, TY, TX, nseg);
  }
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  hipLaunchKernelGGL(window_sums_nhwc_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh, nimg), dim3(kThreads), 0, st,
                     (const float*)workspace, S, nimg, C, k_next, nseg);
  return launch_status();
}

}  // namespace

extern "C" {

int eqa_abi_version(void) { return 1; }

int eqa_set_option(int key, int value) {
  if (key == 0) {
    g_force_direct = value ? 1 : 0;
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

int64_t eqa_group_pool_workspace_bytes(int B, int Cf, int G, int HW) {
  (void)HW;
  if (B <= 0 || Cf <= 0 || G <= 0) return 0;
  return (int64_t)B * pool_splits(B, Cf) * G * (int64_t)sizeof(double);
}

int eqa_group_pool_argmax(const float* feat, float* act, int32_t* gidx, void* workspace, int B, int Cf, int G, int HW,
                          void* stream) {
  if (B == 0) return EQA_OK;
  if (!feat || !act || !workspace || B < 0 || Cf <= 0 || G <= 0 || HW <= 0) return EQA_ERR_INVALID_ARG;
  if (gidx && G > 64) return EQA_ERR_UNSUPPORTED;
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int splits = pool_splits(B, Cf);
  const int cps = (Cf + splits - 1) / splits;
  const int used = (Cf + cps - 1) / cps;  // splits that own at least one channel
  double* partial = (double*)workspace;
  const bool vec = (HW % 4 == 0) && (((uintptr_t)feat & 15) == 0);
  const size_t lds = (size_t)4 * G * sizeof(double);
  if (vec)
    hipLaunchKernelGGL((group_pool_partial_kernel<true>), dim3(used, B), dim3(kThreads), lds, st, feat, partial, Cf, G, HW, cps, used);
  else
    hipLaunchKernelGGL((group_pool_partial_kernel<false>), dim3(used, B), dim3(kThreads), lds, st, feat, partial, Cf, G, HW, cps, used);
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  const double inv_count = 1.0 / ((double)Cf * (double)HW);
  hipLaunchKernelGGL(group_pool_finalize_kernel, dim3((B + 3) / 4), dim3(kThreads), 0, st, partial, act, gidx, B, G, used, inv_count);
  return launch_status();
}

int eqa_group_argmax(const float* act, int32_t* gidx, int B, int G, void* stream) {
  if (B == 0 && G > 0 && G <= 64) return EQA_OK;
  if (!act || !gidx || B < 0 || G <= 0) return EQA_ERR_INVALID_ARG;
  if (G > 64) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(group_argmax_kernel, dim3((B + 3) / 4), dim3(kThreads), 0, (hipStream_t)stream, act, gidx, B, G);
  return launch_status();
}

int eqa_window_sums(const float* x, const float* scale, const float* shift, int relu, double* out, int B, int C, int H,
                    int W, int k, void* stream) {
  if (!x || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || k > H || k > W) return EQA_ERR_INVALID_ARG;
  if (k > kMaxWinK) return EQA_ERR_UNSUPPORTED;
  const size_t lds = (size_t)((H * W + 3) & ~3) * sizeof(float) + (size_t)k * W * sizeof(double);
  if (lds > 64 * 1024 || (long long)B * C > 0x7fffffffLL) return EQA_ERR_UNSUPPORTED;  // plane must fit one block's LDS
  if (B == 0) return EQA_OK;
  const bool vec = ((H * W) % 4 == 0) && (((uintptr_t)x & 15) == 0);
  hipStream_t st = (hipStream_t)stream;
  if (vec)
    hipLaunchKernelGGL((window_sums_kernel<true>), dim3((unsigned)(B * C)), dim3(kThreads), lds, st, x, scale, shift, relu, out, C, H, W, k);
  else
    hipLaunchKernelGGL((window_sums_kernel<false>), dim3((unsigned)(B * C)), dim3(kThreads), lds, st, x, scale, shift, relu, out, C, H, W, k);
  return launch_status();
}

int eqa_winograd_f2k5_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            void* stream) {
  return launch_wino_input<6>(x, V, in_bias, in_relu, nimg, H, W, C, stream);
}
int eqa_winograd_f4k5_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            void* stream) {
  return launch_wino_input<8>(x, V, in_bias, in_relu, nimg, H, W, C, stream);
}

int eqa_winograd_f2k5_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                             void* stream) {
  return launch_wino_output<6>(M, bias, relu, y, nimg, OH, OW, C, stream);
}
int eqa_winograd_f4k5_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                             void* stream) {
  return launch_wino_output<8>(M, bias, relu, y, nimg, OH, OW, C, stream);
}

int64_t eqa_winograd_f2k5_output_sums_workspace_bytes(int nimg, int OH, int C, int k_next) {
  if (nimg <= 0 || OH <= 0 || C <= 0 || k_next <= 0) return 0;
  return (int64_t)nimg * OH * C * (1 + 2 * (k_next - 1)) * (int64_t)sizeof(float);
}

int eqa_winograd_f2k5_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg,
                                  int OH, int OW, int C, int k_next, void* stream) {
  return launch_wino_output_sums<6>(M, bias, relu, S, workspace, nimg, OH, OW, C, k_next, stream);
}
int eqa_winograd_f4k5_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg,
                                  int OH, int OW, int C, int k_next, void* stream) {
  return launch_wino_output_sums<8>(M, bias, relu, S, workspace, nimg, OH, OW, C, k_next, stream);
}

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

int eqa_window_sums_gemv(const double* S, const double* Wm, float* act, int B, int K, int E, double scale, double shift,
                         void* stream) {
  if (B < 0 || K <= 0 || E <= 0) return EQA_ERR_INVALID_ARG;
  if (B == 0) return EQA_OK;
  if (!S || !Wm || !act) return EQA_ERR_INVALID_ARG;
  if (E > kGemvMaxE) return EQA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sums_gemv_kernel, dim3(B), dim3(kThreads), 0, (hipStream_t)stream, S, Wm, act, K, E, scale, shift);
  return launch_status();
}

int eqa_bias_relu_nhwc(float* x, const float* bias, int64_t n_pixels, int C, void* stream) {
  if (!x || !bias || n_pixels < 0 || C <= 0) return EQA_ERR_INVALID_ARG;
  if (C % 4 != 0 || (((uintptr_t)x | (uintptr_t)bias) & 15)) return EQA_ERR_UNSUPPORTED;
  if (n_pixels == 0) return EQA_OK;
  const size_t n_vec = (size_t)n_pixels * (C / 4);
  const unsigned blocks = (unsigned)((n_vec + kThreads - 1) / kThreads < 8192 ? (n_vec + kThreads - 1) / kThreads : 8192);
  hipLaunchKernelGGL(bias_relu_nhwc_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, x, bias, n_vec, C / 4);
  return launch_status();
}

static int ws_nhwc_bands(int H, int k) {
  const int interior = H - 2 * (k - 1);
  int nb = (interior + 9) / 10;  // ~10 rows per band
  if (nb < 1) nb = 1;
  return nb;
}

int64_t eqa_window_sums_nhwc_workspace_bytes(int B, int C, int H, int k) {
  if (B <= 0 || C <= 0 || H <= 0 || k <= 0) return 0;
  const int nseg = 2 * (k - 1) + ws_nhwc_bands(H, k);
  return (int64_t)B * nseg * C * (1 + 2 * (k - 1)) * (int64_t)sizeof(float);
}

int eqa_window_sums_nhwc(const float* x, const float* scale, const float* shift, int relu, double* out, void* workspace,
                         int B, int C, int H, int W, int k, void* stream) {
  if (!x || !out || !workspace || B < 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return EQA_ERR_INVALID_ARG;
  // needs disjoint top / bottom (left / right) border sets and at least one interior row
  if (k > kMaxWinK || C % 4 != 0 || H < 2 * (k - 1) + 1 || W < 2 * (k - 1) + 1 || B > 65535) return EQA_ERR_UNSUPPORTED;
  if ((((uintptr_t)x) & 15) || (scale && (((uintptr_t)scale) & 15)) || (shift && (((uintptr_t)shift) & 15))) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nbands = ws_nhwc_bands(H, k);
  const int nseg = 2 * (k - 1) + nbands;
  hipLaunchKernelGGL(window_sums_nhwc_segment_kernel, dim3(nseg, B), dim3(kThreads), 0, st, x, scale, shift, relu,
                     (float*)workspace, C, H, W, k, nbands);
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  hipLaunchKernelGGL(window_sums_nhwc_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh, B), dim3(kThreads), 0, st,
                     (const float*)workspace, out, B, C, k, nseg);
  return launch_status();
}

int eqa_so3_rotate(const float* x, const float* R, float* y, int B, int N, int transpose, void* stream) {
  if (B == 0 && N > 0) return EQA_OK;
  if (!x || !R || !y || B < 0 || N <= 0) return EQA_ERR_INVALID_ARG;
  if (B > 65535) return EQA_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (N % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL((so3_rotate_kernel<true>), dim3(((N >> 2) + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st, x, R, y, N, transpose);
  else
    hipLaunchKernelGGL((so3_rotate_kernel<false>), dim3((N + kThreads - 1) / kThreads, B), dim3(kThreads), 0, st, x, R, y, N, transpose);
  return launch_status();
}

int eqa_crop_resize_aa(const float* x, float* y, const float* wx, const int32_t* x0, const float* wy, const int32_t* y0,
                       int planes, int H, int W, int OH, int OW, int K, int max_rows, void* stream) {
  if (!x || !y || !wx || !x0 || !wy || !y0 || planes < 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || K <= 0 || max_rows <= 0)
    return EQA_ERR_INVALID_ARG;
  const size_t lds = (size_t)max_rows * OW * sizeof(float);
  if (lds > 96 * 1024 || planes > 65535) return EQA_ERR_UNSUPPORTED;
  if (planes == 0) return EQA_OK;
  hipLaunchKernelGGL(crop_resize_aa_kernel, dim3((OH + kAaBand - 1) / kAaBand, planes), dim3(kThreads), lds, (hipStream_t)stream,
                     x, y, wx, x0, wy, y0, H, W, OH, OW, K, max_rows);
  return launch_status();
}

int eqa_mask_action_nearest(const uint8_t* m, uint8_t* out, const int32_t* eidx, const float* rtheta, const int32_t* flags,
                            int num_elements, int n_masks, int H, int W, void* stream) {
  return launch_nearest<uint8_t>(m, out, eidx, rtheta, flags, num_elements, n_masks, H, W, 0, H, W, 0, 0, 0, stream);
}

int eqa_image_action_nearest(const float* x, float* out, const int32_t* eidx, const float* rtheta, const int32_t* flags,
                             int num_elements, int n_planes, int src_mod, int H, int W, int pad, int OH, int OW, int top,
                             int left, void* stream) {
  return launch_nearest<float>(x, out, eidx, rtheta, flags, num_elements, n_planes, H, W, pad, OH, OW, top, left, src_mod, stream);
}

int64_t eqa_vnsmall_workspace_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return 0;
  return (int64_t)B * ((N + kVnThreads - 1) / kVnThreads) * 12 * (int64_t)sizeof(float);
}

int eqa_vnsmall_fwd(const float* x, const float* params, float* out, void* workspace, int B, int N, int k, int pooling,
                    void* stream) {
  if (!x || !params || !out || !workspace || B < 0 || N <= 0) return EQA_ERR_INVALID_ARG;
  if (k != kVnK || pooling != 0 || N < kVnK) return EQA_ERR_UNSUPPORTED;  // fused path: k = 20, mean pooling
  const size_t lds = (size_t)4 * ((N + 3) & ~3) * sizeof(float) + (size_t)kVnThreads * kVnQueue * sizeof(float2);
  if (lds > 96 * 1024 || B > 65535) return EQA_ERR_UNSUPPORTED;
  if (B == 0) return EQA_OK;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (N + kVnThreads - 1) / kVnThreads;
  hipLaunchKernelGGL(vnsmall_fwd_kernel, dim3(nblk, B), dim3(kVnThreads), lds, st, x, params, (float*)workspace, N, nblk);
  if (hipGetLastError() != hipSuccess) return EQA_ERR_LAUNCH;
  hipLaunchKernelGGL(vnsmall_finalize_kernel, dim3((B * 9 + 255) / 256), dim3(256), 0, st, (const float*)workspace, out, B, nblk, 1.0f / (float)N);
  return launch_status();
}

int eqa_modified_gram_schmidt(const float* v, float* out, int B, void* stream) {
  if (B == 0) return EQA_OK;
  if (!v || !out || B < 0) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(modified_gram_schmidt_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, v, out, B);
  return launch_status();
}

int eqa_rigid_rows(const float* x, const float* R, const float* t, float* out, int M, int mode, void* stream) {
  if (M == 0) return EQA_OK;
  if (!x || !R || !out || M < 0 || (mode != 0 && mode != 1)) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rigid_rows_kernel, dim3((M + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, x, R, t, out, M, mode);
  return launch_status();
}

int eqa_gram_schmidt(const float* v, float* out, int B, void* stream) {
  if (B == 0) return EQA_OK;
  if (!v || !out || B < 0) return EQA_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gram_schmidt_kernel, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, v, out, B);
  return launch_status();
}

}  // extern "C"
