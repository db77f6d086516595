/*
 * eqa_hip.h -- C ABI of libeqa_hip.so, the MI355X (gfx950) implementation of equiadapt's
 * canonicalization hot path.
 *
 * The reference (arnab39/equiadapt) is pure Python and has no FFI; its boundary is the
 * torch.nn.Module contract of BaseCanonicalization.  This header defines the native boundary
 * underneath that contract: one entry point per fused op sequence of SURVEY.md section 8a, each citing the
 * reference lines (relative to the reference repo root) whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to caller-owned memory (e.g. torch tensor .data_ptr());
 *    the library allocates nothing and keeps no state between calls;
 *  - tensors are dense row-major ("contiguous NCHW") fp32 unless stated; indices are int32;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous and
 *    stream-ordered, so they can be captured in a hipGraph;
 *  - return 0 on success, EQA_ERR_* (< 0) otherwise; nothing throws across the boundary.
 *
 * "Group element tables".  A discrete group element e in [0, E) is described by three small device
 * tables that the host builds once per (group, frame size):
 *    theta[e*6 .. e*6+5]  the 2x3 matrix handed to torch's affine_grid by kornia.warp_affine
 *                         (normalised [-1,1] coords, align_corners=True), i.e. the map from an
 *                         OUTPUT pixel of the sampling frame to the SOURCE location;
 *    flags[e]             EQA_FLIP_SRC: sample the horizontally flipped frame (flip BEFORE rotation);
 *                         EQA_FLIP_DST: flip the result horizontally (flip AFTER rotation);
 *    chan_map[e*G + g]    for "regular" features: which input group slot feeds output slot g
 *                         (NULL = identity / "scalar" features).
 */
#ifndef EQA_HIP_H
#define EQA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQA_OK 0
#define EQA_ERR_INVALID_ARG (-1)
#define EQA_ERR_LAUNCH (-2)
#define EQA_ERR_UNSUPPORTED (-3)

#define EQA_FLIP_SRC 1
#define EQA_FLIP_DST 2

/* ABI version of this header.  History: 1 = round-1 entry points; 2 = eqa_crop_resize_aa gained x_begin / x_span;
 * 3 = the round-2 additions (cgemm3m / wgrad3m, conv_s2, vn_tail / vn_bn_finalize, gram_schmidt_bwd, mask planes, boxes,
 * cosine activations, fold_edge_pad, ...) and round 3's (group_action_pair, any-k VNSmall, narrow lifting convolution,
 * Winograd plane GEMM).  A caller compares eqa_abi_version() with the EQA_ABI_VERSION it was compiled against. */
#define EQA_ABI_VERSION 3
/* library / build identification: returns the EQA_ABI_VERSION the library was built from. */
int eqa_abi_version(void);

/* Debug/benchmark knobs (process-global, not part of the data path):
 *   key 0: 1 = force the direct-from-global gather path (no LDS staging) in the resampling kernels.
 *   key 1: VNSmall forward kernel: 0 = chosen by size (default), 1 = one thread per point (k = 20 only), 2 = four lanes per point.
 *   key 2: eqa_fft48k5_cgemm3m_bf16x3: 0 = the block form where it applies (Cout % 128 == 0; default), 1 = always the wave form.
 *   key 3: one-channel maps in eqa_group_action_fwd / _hint / eqa_invert_action_fwd: tiles per block of group_action_c1_kernel
 *          (4 = default, 2, or 0 = the general kernel with one tile per block); the output is bit-identical either way.
 *   key 100 (get only): the largest window size k the window-sum kernels take (eqa_window_sums*, the linearised last layer). */
int eqa_set_option(int key, int value);
int eqa_get_option(int key);

/*
 * I5 -- fused  pad(edge) -> [hflip blend] -> rotate(-theta_g) -> center-crop.
 * Replaces equiadapt/images/canonicalization/discrete_group.py:207-215
 *   (self.pad :62-66, K.geometry.hflip blend :209-211, K.geometry.rotate :213, self.crop :67-71).
 * x:(B,C,H,W) -> y:(B,C,H,W).  The sampling frame is (H+2*pad, W+2*pad); only the H x W pixels the
 * reference keeps are computed.  pad = 0 reproduces the grayscale branch (no pad/crop, zero corners).
 * gidx:(B) group element per image; theta:(E,6); flags:(E).
 */
int eqa_canon_transform_fwd(const float* x, float* y, const int32_t* gidx, const float* theta,
                            const int32_t* flags, int num_elements, int B, int C, int H, int W, int pad,
                            void* stream);

/*
 * I7 -- invert_canonicalization / get_action_on_image_features:
 *   rotate(+theta_g) with zero corners -> hflip blend (flipped when the reflection indicator is 0)
 *   -> for "regular" features a cyclic roll of the group axis.
 * Replaces equiadapt/images/utils.py:32-94 and roll_by_gather :8-29
 *   (called from images/canonicalization/discrete_group.py:240-259).
 * f:(B,C,H,W) -> out:(B,C,H,W); chan_map:(E,G) or NULL ("scalar"); C % G == 0 when chan_map != NULL.
 */
int eqa_invert_action_fwd(const float* f, float* out, const int32_t* gidx, const float* theta,
                          const int32_t* flags, const int32_t* chan_map, int num_elements, int G, int B, int C,
                          int H, int W, void* stream);

/*
 * I5 + I7 in ONE launch, for callers that hold both tensors at the same time (a pipelined loop canonicalizing batch t+1
 * while inverting the prediction of batch t; test-time evaluation; the transform-only benchmark): y = canonicalize(x)
 * exactly as eqa_canon_transform_fwd and out = invert(f) exactly as eqa_invert_action_fwd, same gidx:(B) for both
 * (discrete_group.py:204-215 and images/utils.py:54-89 of the reference, called back to back).
 * x:(B,C,H,W), f:(B,Cf,H,W); theta_canon / flags_canon and theta_inv / flags_inv: the (E,6) / (E) tables of the two
 * actions; chan_map:(E,G) or NULL.  Bit-identical to the two separate calls; when the two jobs cannot share a tile grid
 * (different channel staging widths) the library issues the two launches itself.
 */
int eqa_group_action_pair(const float* x, float* y, const float* theta_canon, const int32_t* flags_canon, int pad, int C,
                          const float* f, float* out, const float* theta_inv, const int32_t* flags_inv,
                          const int32_t* chan_map, int G, int Cf, const int32_t* gidx, int num_elements, int B, int H,
                          int W, void* stream);

/*
 * I8 -- orbit expansion (group_augment): for every group element e,
 *   pad(edge) -> rotate(-deg_e) -> [hflip AFTER the rotation] -> center-crop(S), concatenated
 *   element-major: y[(e*B + b)] = action_e(x[b]).   x:(B,C,S,S) -> y:(E*B,C,S,S).
 * Replaces equiadapt/images/canonicalization/discrete_group.py:387-427
 *   (rotate_and_maybe_reflect :387-409, group_augment :411-427).
 */
int eqa_orbit_expand_fwd(const float* x, float* y, const float* theta, const int32_t* flags, int num_elements,
                         int B, int C, int S, int pad, void* stream);

/*
 * Generic form of the three entry points above (also used for GroupInference-style orbits and the
 * artifact branch, discrete_group.py:448-473): out image n uses element
 *   e = gidx ? gidx[n] : n / B   and source image   b = gidx ? n : n % B.
 * Source planes are (H,W); the frame is (H+2*pad, W+2*pad); the output is the (OH,OW) window of the
 * frame whose top-left corner is (top,left).
 */
int eqa_group_action_fwd(const float* src, float* dst, const int32_t* gidx, const float* theta,
                         const int32_t* flags, const int32_t* chan_map, int num_elements, int G, int n_out,
                         int B, int C, int H, int W, int pad, int OH, int OW, int top, int left, void* stream);
/* eqa_group_action_fwd with the caller's bound on the source window of a 32 x 32 output tile, in pixels (0: unknown = 47, what an
 * arbitrary rotation needs).  The elements of C4 / D4 (multiples of 90 degrees -- the reference's num_rotations = 4 configurations,
 * discrete_group.py:110-112) need 35: the launch then reserves 35 instead of 47 window rows of LDS per block and two more blocks fit
 * a CU.  A tile whose window exceeds the bound is sampled straight from global memory (same arithmetic, slower): the hint can cost
 * time, never correctness. */
int eqa_group_action_fwd_hint(const float* src, float* dst, const int32_t* gidx, const float* theta, const int32_t* flags,
                              const int32_t* chan_map, int num_elements, int G, int n_out, int B, int C, int H, int W,
                              int pad, int OH, int OW, int top, int left, int max_window, void* stream);

/*
 * I1 -- centre crop + antialiased bilinear resize of the canonicalization network's input
 * (transforms.CenterCrop -> transforms.Resize on a tensor == F.interpolate(bilinear, antialias=True,
 * align_corners=False); equiadapt/images/canonicalization/discrete_group.py:174-188, built :73-92).
 * Separable (horizontal, then vertical, fp32 intermediates) like torch's kernel.  The caller supplies, per output
 * column / row, the first input tap (crop offset included) and K normalised weights (zero-padded), computed with
 * torch's formula: wx:(OW,K) x0:(OW) wy:(OH,K) y0:(OH).  x:(planes,H,W) -> y:(planes,OH,OW); max_rows >= the number of
 * input rows any band of 8 output rows touches; [x_begin, x_begin + x_span) = the input columns any output column reads
 * (min x0 .. max x0 + K, clipped to W; x_span <= 0 disables the LDS-staged path used for K > 8).
 */
int eqa_crop_resize_aa(const float* x, float* y, const float* wx, const int32_t* x0, const float* wy, const int32_t* y0,
                       int planes, int H, int W, int OH, int OW, int K, int max_rows, int x_begin, int x_span, void* stream);

/*
 * I6 -- nearest-neighbour group action on uint8 masks: torchvision.transforms.functional.rotate defaults
 * (equiadapt/images/utils.py:125-136 rotate_masks, after the optional flip_masks :112-122).
 * m,out:(n_masks,H,W) uint8; eidx:(n_masks) element per mask; rtheta:(E,6) the inverse affine matrix already rescaled by
 * (0.5 W, 0.5 H) in the order r00,r10,r20,r01,r11,r21; flags:(E) EQA_FLIP_SRC = flip the mask before rotating.
 */
int eqa_mask_action_nearest(const uint8_t* m, uint8_t* out, const int32_t* eidx, const float* rtheta, const int32_t* flags,
                            int num_elements, int n_masks, int H, int W, void* stream);
/* The same with the source planes given one by one: planes:(n_masks) device array of device pointers to (H,W) uint8 planes, each
 * 4-byte aligned -- the masks of a batch are one tensor per sample (discrete_group.py:217-236 loops over the samples), so no
 * concatenation pass is needed in front of the kernel.  W % 16 == 0 and out 16-byte aligned, else EQA_ERR_UNSUPPORTED. */
int eqa_mask_action_nearest_planes(const uint8_t* const* planes, uint8_t* out, const int32_t* eidx, const float* rtheta,
                                   const int32_t* flags, int num_elements, int n_masks, int H, int W, void* stream);

/*
 * I6 -- the boxes of a batch follow their images: flip_boxes (equiadapt/images/utils.py:97-109; applied to EVERY box when
 * the group has reflections, discrete_group.py:220-224) then rotate_boxes (:161-187: both corners rotated about
 * (width/2, width/2) by the image's angle in degrees, rotate_points :139-158, and re-sorted into x0<=x1, y0<=y1).
 * boxes,out:(n,4) xyxy fp32, 16-byte aligned; img_of_box:(n) image of each box; rotation_deg:(B) per image;
 * flipped:(n,4) or NULL receives the boxes after the flip (the reference flips the caller's tensors in place).
 */
int eqa_boxes_action(const float* boxes, const int32_t* img_of_box, const float* rotation_deg, float* flipped, float* out,
                     int n, float width, int flip_all, void* stream);

/*
 * (f).2 -- nearest-neighbour group action on fp32 image planes with edge padding and a crop window: the test-time orbit
 * of GroupInference (examples/images/classification/inference_utils.py:100-123: transforms.Pad(0.4 H, edge) ->
 * [hflip] -> transforms.functional.rotate(+deg) [NEAREST by default] -> CenterCrop).
 * x:(src planes,H,W); out plane p samples source plane (src_mod > 0 ? p % src_mod : p) with element eidx[p];
 * frame = (H+2 pad, W+2 pad); out:(n_planes,OH,OW) = the window at (top,left); rtheta rescaled by the FRAME size.
 */
int eqa_image_action_nearest(const float* x, float* out, const int32_t* eidx, const float* rtheta, const int32_t* flags,
                             int num_elements, int n_planes, int src_mod, int H, int W, int pad, int OH, int OW, int top,
                             int left, void* stream);

/*
 * Backward of eqa_group_action_fwd (and so of I5 / I7 / I8), what the reference obtains from autograd through
 * K.geometry.rotate (discrete_group.py:213, images/utils.py:57,82):
 *   grad_src            dL/d(src), shape of src, MUST be zero-filled by the caller; accumulated with float atomics
 *                       (NULL = not wanted);
 *   grad_angle_partial  (n_out, eqa_group_action_bwd_tiles(OH, OW)) partial sums of dL/d(phi) in RADIANS of the
 *                       rotation angle that theta encodes (d theta = rotation about the frame centre); the caller sums
 *                       over the tile axis (deterministic) and scales by pi/180 and the sign of its angle convention
 *                       (NULL = not wanted).
 * grad_out has the shape of dst.  The gradient w.r.t. a reflection indicator is a difference of two forward
 * transforms dotted with grad_out and needs no kernel of its own.
 */
/* Adjoint of the replicate ("edge") padding: gframe:(planes, H+2pad, W+2pad) -> gsrc:(planes, H, W), every source pixel
 * receives the sum over the frame pixels that clamp to it.  With eqa_group_action_bwd run on the padded frame as its (un-padded)
 * source this is the deterministic input gradient of the canonicalizing transform (discrete_group.py:204-215).
 * workspace: eqa_fold_edge_pad_workspace_bytes(planes, H, W, pad) bytes. */
int64_t eqa_fold_edge_pad_workspace_bytes(int planes, int H, int W, int pad);
int eqa_fold_edge_pad(const float* gframe, float* gsrc, void* workspace, int planes, int H, int W, int pad, void* stream);
int eqa_group_action_bwd_tiles(int OH, int OW);
int eqa_group_action_bwd(const float* src, const float* grad_out, const int32_t* gidx, const float* theta,
                         const int32_t* flags, const int32_t* chan_map, float* grad_src, float* grad_angle_partial,
                         int num_elements, int G, int n_out, int B, int C, int H, int W, int pad, int OH, int OW,
                         int top, int left, void* stream);
/* Same, for a per-sample AFFINE matrix (continuous groups: K.geometry.warp_affine, images/canonicalization/
 * continuous_group.py:203): grad_theta_partial (n_out, eqa_group_action_bwd_tiles(OH, OW), 6) partial sums of dL/d(theta
 * row of the output image's element); the caller sums over the tile axis. */
int eqa_group_action_bwd_theta(const float* src, const float* grad_out, const int32_t* gidx, const float* theta,
                               const int32_t* flags, const int32_t* chan_map, float* grad_src, float* grad_theta_partial,
                               int num_elements, int G, int n_out, int B, int C, int H, int W, int pad, int OH, int OW,
                               int top, int left, void* stream);

/*
 * I3 + I4 -- group pooling and orientation argmax.
 * feat:(B, Cf, G, HW) fp32 (the canonicalization network's last feature map, group axis second)
 *   act[b,g]  = mean over (Cf, HW)                              (escnn_networks.py:106-115,
 *                                                                custom_equivariant_networks.py:91)
 *   gidx[b]   = argmax_g act[b,g], first index on ties          (common/basecanonicalization.py:233-235)
 * workspace: at least eqa_group_pool_workspace_bytes(B, Cf, G, HW) bytes of device scratch.
 */
int64_t eqa_group_pool_workspace_bytes(int B, int Cf, int G, int HW);
int eqa_group_pool_argmax(const float* feat, float* act, int32_t* gidx, void* workspace, int B, int Cf, int G,
                          int HW, void* stream);

/*
 * I2 tail + I3 -- exact linear shortcut for "last convolution -> group mean" (escnn_networks.py:88-115,
 * custom_equivariant_networks.py:80-93): since the mean over (fields, space) of a convolution is linear in its input,
 *   act[b,g] = (1/count) * sum_{c,u,v} Weff[g,c,u,v] * S[b,c,u,v] + mean(bias),   Weff = sum over fields of the filter bank,
 * where S are the k*k shifted-window sums of each input plane.  This entry point computes S in one pass over the
 * feature map, applying the previous layer's per-channel affine (bias / eval-mode batch-norm) and ReLU on the fly:
 *   S[b,c,u,v] = sum_{y < H-k+1, x < W-k+1} act(x[b,c,y+u,x+v]),   act(t) = relu ? max(scale[c]*t + shift[c], 0) : ...
 * x:(B,C,H,W) fp32; scale, shift:(C) or NULL; out:(B,C,k,k) fp64.  k <= 8, one plane must fit 64 KB of LDS.
 */
int eqa_window_sums(const float* x, const float* scale, const float* shift, int relu, double* out, int B, int C, int H,
                    int W, int k, void* stream);

/*
 * Channels-last companions of the canonicalization-network inference path (MIOpen's fp32 convs run natively in NHWC):
 *   eqa_bias_relu_nhwc    x[p][c] = max(x[p][c] + bias[c], 0) in place; x:(n_pixels, C), C % 4 == 0
 *                         (bias / folded eval batch-norm + ReLU between two convolutions, escnn_networks.py:67-85);
 *   eqa_window_sums_nhwc  eqa_window_sums for a (B,H,W,C) buffer; out:(B,C,k,k) fp64 as above;
 *                         workspace: eqa_window_sums_nhwc_workspace_bytes(B, C, H, k) bytes.
 */
/* I10 -- the strided convolutions of ConvNetwork (custom_nonequivariant_networks.py:44-57: Conv2d(k, stride 2, padding 0 | 1) ->
 * BatchNorm2d -> GELU) in inference, as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32; csrc/smallconv.hip).
 * y:(B,OH,OW,Cout) channels-last = [gelu](conv(x) + bias), OH = (H + 2 pad - K) / 2 + 1; the caller folds the eval-mode batch-norm
 * into weights and bias.  planar = 1: x:(B,Cin,H,W) with Cin <= 4 (the first layer) and wp:(K*K, Cout/16, 4, 16) =
 * w[16 n + j][kq][u][v] (zero for kq >= Cin); planar = 0: x:(B,H,W,Cin) channels-last with Cin % 16 == 0 and
 * wp:(K*K, Cin/16, Cout/16, 4, 16, 4) = w[16 n + j][16 c + 4 kq + s][u][v].  K in {3,5,7}; channel pairs: eqa_conv_s2_supported. */
int eqa_conv_s2_supported(int Cin, int Cout, int K, int pad, int planar);
int eqa_conv_s2(const float* x, const float* wp, const float* bias, int gelu, float* y, int B, int Cin, int H, int W, int Cout, int K,
                int pad, int planar, void* stream);
/* I10 in TRAINING (csrc/convnet_train.hip) -- what the framework's autograd ran through MIOpen / ATen for
 * custom_nonequivariant_networks.py:44-80 (Conv2d stride 2 -> BatchNorm2d -> GELU per layer; head BatchNorm1d -> Dropout1d -> ReLU):
 *   eqa_conv_s2_wgrad   dw:(Cout,Cin,K,K) = d/dW of eqa_conv_s2 given dz:(B,OH,OW,Cout) channels-last and the layer's input x (planar = 1:
 *                       (B,Cin<=4,H,W); else (B,H,W,Cin) channels-last) -- torch.nn.grad.conv2d_weight.  Reduction over the output pixels on
 *                       the matrix cores, partial sums per pixel run in `workspace` (eqa_conv_s2_wgrad_workspace_bytes), summed in a
 *                       fixed order (fp64: chunks of 128 runs, then the chunks): deterministic, no atomics.
 *   eqa_conv_s2_dgrad   dx:(B,H,W,Cin) = d/dx (channels-last layers only: the image needs no gradient) given dz and
 *                       wd:(K*K, Cout/16, Cin/16, 4, 16, 4) = w[16 cc + 4 kq + s][16 n + j][u][v] -- torch.nn.grad.conv2d_input.  Every
 *                       element of dx is written (pixels no tap reaches get 0).
 *   eqa_bn_act_fwd      y[p][c] = rowscale[p] * act(scale[c] z[p][c] + shift[c]) on (npix, C) channels-last, C % 4 == 0; act 0 = exact GELU
 *                       (erf), 1 = ReLU; rowscale may be NULL (= 1).  scale / shift = the batch-norm folded with the batch statistics
 *                       (eqa_bn_stats_nhwc).  For the head, rowscale = Dropout1d's per-row factor (0 or 1 / (1 - p)).
 *   eqa_bn_act_stats    partial:(eqa_bn_act_partial_blocks(npix), C, 2) fp64 = per-block sum and sum of squares of z (eqa_bn_stats_nhwc with
 *                       the block's pixel count chosen per launch: the head's (rows, D) matrix has few "pixels" and many channels).
 *   eqa_bn_act_finalize one launch behind it: batch mean / biased variance (fp64) -> mean, rstd = 1 / sqrt(var + eps), scale = gamma * rstd,
 *                       shift = beta - mean * scale (fp32), and running_mean / running_var updated in place as nn.BatchNorm*d does
 *                       (momentum; unbiased variance; both NULL: no running statistics).
 *   eqa_bn_act_bwd_reduce  partial:(eqa_bn_act_partial_blocks(npix), C, 2) fp64 = per-block sums of g and g * zhat, g = gy * rowscale *
 *                       act'(scale z + shift), zhat = (z - mean) * rstd  (the two reductions of batch-norm's backward).
 *   eqa_bn_act_bwd_finalize  the partials summed in block order -> dgamma = sum(g zhat), dbeta = sum(g), gscale = gamma * rstd,
 *                       m1 = sum(g) / n, m2 = sum(g zhat) / n.
 *   eqa_bn_act_bwd_apply   dz = gscale[c] * (g - m1[c] - zhat * m2[c]). */
int eqa_conv_s2_wgrad_supported(int Cin, int Cout, int K, int pad, int planar);
int64_t eqa_conv_s2_wgrad_workspace_bytes(int B, int Cin, int H, int W, int Cout, int K, int pad, int planar);
int eqa_conv_s2_wgrad(const float* x, const float* dz, float* dw, void* workspace, int B, int Cin, int H, int W, int Cout, int K, int pad,
                      int planar, void* stream);
int eqa_conv_s2_dgrad_supported(int Cin, int Cout, int K, int pad);
int eqa_conv_s2_dgrad(const float* dz, const float* wd, float* dx, int B, int Cin, int H, int W, int Cout, int K, int pad, void* stream);
int eqa_bn_act_fwd(const float* z, const float* scale, const float* shift, const float* rowscale, float* y, int64_t npix, int C, int act,
                   void* stream);
int64_t eqa_bn_act_partial_blocks(int64_t npix);
int eqa_bn_act_stats(const float* z, double* partial, int64_t npix, int C, void* stream);
int eqa_bn_act_finalize(const double* partial, int64_t npix, int C, const float* gamma, const float* beta, double eps, double momentum,
                        float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* rstd, void* stream);
int eqa_bn_act_bwd_finalize(const double* partial, int64_t npix, int C, const float* gamma, const float* rstd, float* dgamma, float* dbeta,
                            float* gscale, float* m1, float* m2, void* stream);
int eqa_bn_act_bwd_reduce(const float* gy, const float* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                          const float* rowscale, double* partial, int64_t npix, int C, int act, void* stream);
int eqa_bn_act_bwd_apply(const float* gy, const float* z, const float* scale, const float* shift, const float* mean, const float* rstd,
                         const float* rowscale, const float* gscale, const float* m1, const float* m2, float* dz, int64_t npix, int C,
                         int act, void* stream);
/* Optimized canonicalizer, inference tail:
 *   eqa_affine_relu_rows           z:(rows,D) = relu(h * scale[d] + shift[d]): the eval-mode BatchNorm1d (folded to scale / shift) +
 *                                  ReLU in front of ConvNetwork's Linear head (custom_nonequivariant_networks.py:62-67); D % 4 == 0.
 *   eqa_cosine_group_activations   v:(G*B,V) element-major orbit embeddings, ref:(V) -> act:(B,G) =
 *                                  cosine_similarity(ref, v[g*B+b]) with torch's formula (each vector divided by max(norm, eps)):
 *                                  discrete_group.py:475-481 (cosine_similarity + reshape(G,-1).T) in one launch. */
int eqa_affine_relu_rows(const float* h, const float* scale, const float* shift, float* z, int64_t rows, int D, void* stream);
int eqa_cosine_group_activations(const float* v, const float* ref, float* act, int B, int G, int V, float eps, void* stream);
/* Training counterparts of the linearised last layer (escnn_networks.py:106-115 pooled, in training: pooling.py / escnn_networks.py):
 *   eqa_window_grad_table       dS:(B,C,k,k) fp64 -> table:(B,2k-1,2k-1,C) fp32, the backward of eqa_window_sums* as a class table (a pixel's
 *                               gradient depends on the class of its row / column only: k-1 border indices each side + the interior) --
 *                               rectangle sums of dS from 2-D prefix sums; replaces an fp64 mask einsum (two library batched GEMMs).
 *   eqa_window_sums_gemv_bwd    backward of eqa_window_sums_gemv (act = scale * S . Wm^T): dS:(B,K) = scale * dact . Wm and
 *                               dWm:(E,K) = scale * dact^T . S, both fp64; either may be NULL; workspace (for dWm):
 *                               eqa_window_sums_gemv_bwd_workspace_bytes(K, E); batch slices summed in order (deterministic). */
int eqa_window_grad_table(const double* dS, float* table, int B, int C, int H, int W, int k, void* stream);
int64_t eqa_window_sums_gemv_bwd_workspace_bytes(int K, int E);
int eqa_window_sums_gemv_bwd(const float* dact, const double* Wm, const double* S, double* dS, double* dWm, void* workspace, int B, int K,
                             int E, double scale, void* stream);
int eqa_bias_relu_nhwc(float* x, const float* bias, int64_t n_pixels, int C, void* stream);
int64_t eqa_window_sums_nhwc_workspace_bytes(int B, int C, int H, int k);
int eqa_window_sums_nhwc(const float* x, const float* scale, const float* shift, int relu, double* out, void* workspace,
                         int B, int C, int H, int W, int k, void* stream);
/*
 * Training: backward of eqa_window_sums_nhwc (the reference gets it from autograd through the last R2Conv + torch.mean,
 * escnn_networks.py:106-115).  A pixel receives the sum of dS over the windows containing it, which depends only on the class
 * of its row and column (border index i < k-1, interior = k-1, bottom / right border k + (i - (n-k+1))).
 * table:(B, 2k-1, 2k-1, C) fp32, one gradient per class pair (the caller builds it from dS); dx:(B,H,W,C).
 * H, W >= 2k-1, C % 4 == 0, 16-byte aligned.
 */
int eqa_window_sums_bwd_expand_nhwc(const float* table, float* dx, int B, int H, int W, int C, int k, void* stream);

/*
 * Training-mode hidden block of the canonicalization network, channels-last: InnerBatchNorm -> ReLU -> PointwiseDropout
 * (escnn_networks.py:67-85), forward and backward.  x, y, gy, dx: (n_pixels, C) fp32, C % 4 == 0, 16-byte aligned.
 *   eqa_bn_stats_nhwc         partial[(blk*C + c)*2 + {0,1}] = sum, sum of squares of channel c over block blk's pixels;
 *                             blk < eqa_bn_partial_blocks(n_pixels); the caller adds the blocks (fp64, deterministic) and folds
 *                             the G channels of a field into the per-field mean / variance;
 *   eqa_bn_relu_dropout_nhwc  y = dropout(relu(x*scale[c] + shift[c])), kept elements scaled by 1/(1-drop_p); the mask is a
 *                             counter-based hash of (seed, element index); drop_p = 0 disables it;
 *   eqa_bn_bwd_reduce_nhwc    with g = gy * (y > 0 ? 1/(1-drop_p) : 0) and xhat = (x - mean[c]) * rstd[c]:
 *                             partial[...] = sum g, sum g*xhat (same layout as above);
 *   eqa_bn_bwd_apply_nhwc     dx = a[c] * (g - b[c] - xhat * d[c])   (a = gamma*rstd, b = sum g / n, d = sum g xhat / n).
 *   In both backward passes y may be NULL: "kept and positive" is then recomputed from x, scale, shift and the seed exactly as
 *   the forward pass decided it, and the pass reads one map less (scale / shift / seed are ignored when y is given).
 */
int64_t eqa_bn_partial_blocks(int64_t n_pixels);
int eqa_bn_stats_nhwc(const float* x, double* partial, int64_t n_pixels, int C, void* stream);
int eqa_bn_relu_dropout_nhwc(const float* x, const float* scale, const float* shift, float* y, int64_t n_pixels, int C,
                             float drop_p, uint32_t seed, void* stream);
int eqa_bn_bwd_reduce_nhwc(const float* gy, const float* y, const float* x, const float* mean, const float* rstd, float drop_p,
                           double* partial, int64_t n_pixels, int C, const float* scale, const float* shift, uint32_t seed,
                           void* stream);
int eqa_bn_bwd_apply_nhwc(const float* gy, const float* y, const float* x, const float* mean, const float* rstd, const float* a,
                          const float* b, const float* d, float drop_p, float* dx, int64_t n_pixels, int C, const float* scale,
                          const float* shift, uint32_t seed, void* stream);
/*
 * The LAST hidden block (the one in front of the linearised final convolution, escnn_networks.py:67-91 + :106-115) without its
 * output ever being written, in training:
 *   eqa_window_sums_nhwc_act        eqa_window_sums_nhwc of dropout(relu(scale[c] * x + shift[c])) with the dropout mask of
 *                                   eqa_bn_relu_dropout_nhwc (same hash of (seed, element index)) applied on the fly;
 *   eqa_bn_bwd_reduce_nhwc_wsgrad,  the two backward passes of the block with the upstream gradient given as the window sums'
 *   eqa_bn_bwd_apply_nhwc_wsgrad    (B, 2k-1, 2k-1, C) class table (gy[b][y][x][c] = table[b][cls(y)][cls(x)][c]; classes: the
 *                                   k-1 top / left border indices, the interior, the k-1 bottom / right ones) instead of an
 *                                   expanded (B, H, W, C) map; "kept and positive" recomputed from x, scale, shift, seed.
 * x:(B,H,W,C) channels-last, C % 4 == 0; partial / a / b / d as in the plain forms.
 */
int eqa_window_sums_nhwc_act(const float* x, const float* scale, const float* shift, int relu, float drop_p, uint32_t seed, double* out,
                             void* workspace, int B, int C, int H, int W, int k, void* stream);
int eqa_bn_bwd_reduce_nhwc_wsgrad(const float* table, const float* x, const float* mean, const float* rstd, float drop_p,
                                  double* partial, int B, int H, int W, int C, int k, const float* scale, const float* shift,
                                  uint32_t seed, void* stream);
int eqa_bn_bwd_apply_nhwc_wsgrad(const float* table, const float* x, const float* mean, const float* rstd, const float* a,
                                 const float* b, const float* d, float drop_p, float* dx, int B, int H, int W, int C, int k,
                                 const float* scale, const float* shift, uint32_t seed, void* stream);

/*
 * The GEMV after the window sums (last convolution + mean over channels and positions, escnn_networks.py:115,
 * custom_equivariant_networks.py:91, collapsed to a linear map):
 *   act[b][e] = (float)(scale * sum_j S[b][j] * Wm[e][j] + shift);  S:(B,K) fp64, Wm:(E,K) fp64, act:(B,E) fp32, E <= 16.
 */
int eqa_window_sums_gemv(const double* S, const double* Wm, float* act, int B, int K, int E, double scale, double shift,
                         void* stream);

/*
 * I2a, lifting convolution in inference (escnn_networks.py:60-66 first R2Conv; custom_group_equivariant_layers.py lifting
 * layer): few input channels -> Cout channels, KH x KW, stride 1, no padding, channels-last, on the fp32 MFMA.
 *   y[n,oy,ox,co] = [relu]( sum_{ky,kx,ci} x[n,oy+ky,ox+kx,ci] * w[co,ci,ky,kx] + bias[co] )
 * x:(nimg,H,W,Cin); y:(nimg,H-KH+1,W-KW+1,Cout); bias:(Cout) or NULL.  Supported: KH in {3,5}, 9 <= R = KW*Cin <= 15,
 * Cout % 64 == 0 (else EQA_ERR_UNSUPPORTED: use the framework's convolution).
 * wpk: weights packed (KH*8, 2, Cout):  wpk[(ky*8+q)*2 + h][co] = w[co][ci][ky][kx],  j = (R-8)*h + q, kx = j / Cin,
 * ci = j % Cin, and 0 where h == 1 and q < 16-R  (the two 8-element halves of a filter row overlap; the kernel uses the
 * first of those zero slots to add the bias on the matrix core).
 * 5 x 5 filters over 3 channels with whole 64-channel slices and output rows of >= 32 pixels run a denser form of the kernel
 * (38 k-steps instead of 40, tiles over the flattened output map): same packed layout, same results to the rounding of the
 * summation order (HISTORY.md 3.7).
 */
int eqa_lift_conv_nhwc(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W,
                       int Cin, int KH, int KW, int Cout, void* stream);
/* The same convolution with the output in the channel-group-major layout (nimg, Cout/16, OH, OW, 16) that
 * eqa_fft48k5_input_grouped reads (inference: the lifting layer feeding an FFT-convolved layer).  Channels-last, the 64 bytes
 * a 16-channel block of the input transform needs from a pixel are half a cache line whose other half belongs to another
 * block; grouped, a tile row of a channel group is one contiguous run.  Same arguments and return codes. */
int eqa_lift_conv_grouped(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W, int Cin,
                          int KH, int KW, int Cout, void* stream);
/* The lifting convolution for the filters eqa_lift_conv_nhwc does not take (csrc/lift_conv_wide.hip): 7 x 7 and 9 x 9 over 3 input
 * channels (the reference tutorial's ESCNN canonicalizer: kernel_size = 9) and 3 x 3 ... 9 x 9 over one (grayscale); same
 * arithmetic and layouts (escnn_networks.py:60-66; custom_group_equivariant_layers.py lifting layer), Cout % 64 == 0 and
 * Cout / 64 a divisor of 1024.  wpk: (Cout/64, (R+1)/2, 2, 64) with R = KH*KW*Cin:
 *   wpk[slice][step][nt][lane] = w[co = 64 slice + 32 nt + (lane & 31)][ci][ky][kx],  r = 2 step + (lane >> 5) = (ky*KW + kx)*Cin + ci,
 *   0 for r >= R.  eqa_lift_conv_wide_weight_floats: its size. */
int eqa_lift_conv_wide_supported(int Cin, int KH, int KW, int Cout);
int64_t eqa_lift_conv_wide_weight_floats(int Cin, int KH, int KW, int Cout);
int eqa_lift_conv_wide(const float* x, const float* wpk, const float* bias, int relu, float* y, int nimg, int H, int W, int Cin,
                       int KH, int KW, int Cout, void* stream);
/* Its filter gradient (training; autograd's convolution-weight-gradient of the same layer, escnn_networks.py:60-66):
 * dbank[co][ci][ky][kx] = sum_{n,oy,ox} dy[n,oy,ox,co] * x[n,oy+ky,ox+kx,ci], deterministic (per-block partials reduced in a fixed
 * order).  workspace: eqa_lift_conv_wide_wgrad_workspace_bytes(...) bytes; x, dy channels-last. */
int64_t eqa_lift_conv_wide_wgrad_workspace_bytes(int Cin, int KH, int KW, int Cout);
int eqa_lift_conv_wide_wgrad(const float* x, const float* dy, void* workspace, float* dbank, int nimg, int H, int W, int Cin, int KH,
                             int KW, int Cout, void* stream);
/* Training: the lifting convolution of escnn_networks.py:60-66 feeds an InnerBatchNorm (escnn_networks.py:67-70) whose batch
 * statistics are per-channel sums over this very map.  This form (no bias, no activation) also leaves
 *   sum over rows r of partial[(r * Cout + c) * 2 + {0, 1}]  =  sum, sum of squares of y[.., c] over all pixels      (fp64)
 * -- the kernel's waves keep running sums of the values they store (in place of the ReLU's instructions), one partial row per
 * tile stream, followed by rows that take the twice-computed seam pixels of the tiling out again -- so that eqa_bn_stats_nhwc's
 * pass over the map is not needed.  rows = eqa_lift_conv_stats_rows(...) (0: this shape has no such form -- Cout % 64 != 0 or
 * output rows shorter than 32 pixels: use eqa_lift_conv_nhwc + eqa_bn_stats_nhwc); partial: rows * Cout * 2 doubles. */
int eqa_lift_conv_stats_rows(int nimg, int H, int W, int Cin, int KH, int KW, int Cout);
int eqa_lift_conv_nhwc_stats(const float* x, const float* wpk, float* y, double* partial, int nimg, int H, int W, int Cin, int KH,
                             int KW, int Cout, void* stream);

/*
 * Training: filter gradient of the same lifting convolution (the reference: autograd through e2cnn's R2Conv,
 * escnn_networks.py:48-66).  dbank:(Cout,Cin,KH,KW) = sum over (n,oy,ox) of dy[n][oy][ox][co] * x[n][oy+ky][ox+kx][ci], both
 * channels-last; fp32 MFMA (v_mfma_f32_16x16x4_f32), per-wave partials added in a fixed order (deterministic).
 * eqa_lift_conv_wgrad_supported: KH*KW*Cin <= 80, Cout % 256 == 0, (W-KW+1) / 4 in {7,11,15,23,31} exactly, KH*W*Cin <= 2048.
 * workspace: eqa_lift_conv_wgrad_workspace_bytes(...) bytes, 16-byte aligned; dy 16-byte aligned.
 */
int eqa_lift_conv_wgrad_supported(int nimg, int H, int W, int Cin, int Cout, int KH, int KW);
int64_t eqa_lift_conv_wgrad_workspace_bytes(int nimg, int H, int W, int Cin, int Cout, int KH, int KW);
int eqa_lift_conv_wgrad_nhwc(const float* x, const float* dy, void* workspace, float* dbank, int nimg, int H, int W, int Cin,
                             int Cout, int KH, int KW, void* stream);

/*
 * I2a, 5x5 stride-1 group convolutions in inference (escnn_networks.py:67-91), Winograd F(m x m, 5x5), channels-last.
 * f2k5: m = 2 (6x6 input tiles, 36 planes); f4k5: m = 4 (8x8 input tiles, 64 planes).  N = m + 4, P = N*N:
 *   eqa_winograd_f{m}k5_input   x:(nimg,H,W,C) -> V:(nimg*TY*TX, P, C), TY = (H-4)/m, TX = (W-4)/m   (B^T d B), with
 *                               d = in_relu ? max(x + in_bias[c], 0) : x + in_bias[c]  (in_bias NULL = 0): the previous
 *                               layer's bias / folded batch-norm / ReLU applied while loading
 *   eqa_plane_gemm              M[:,a] = V[:,a] (tiles x Cin, row stride P*Cin) . U[a] (Cin x Cout), a < P: the per-plane
 *                               channel contraction on the fp32 MFMA (declared below; channel counts it does not take: a
 *                               strided-batched GEMM by the caller)
 *   eqa_winograd_f{m}k5_output  M:(nimg*TY*TX, P, C) -> y:(nimg,OH,OW,C) = [relu](A^T M A + bias[c]),  m | OH, OW
 * Cook-Toom points {0, 1, -1, 2, -2, [1/2, -1/2,] inf}; matrices in csrc/winograd.hip and
 * images/canonicalization_networks/winograd.py (U = G g G^T in fp64).  EQA_ERR_UNSUPPORTED when m does not divide H-4, W-4.
 */
/* The per-plane channel contraction of the Winograd convolution (escnn_networks.py:67-91 at shapes the FFT tiles do not fit).
 * V:(T,P,Cin), M:(T,P,Cout) as above; Upk: U:(P,Cin,Cout) re-ordered into MFMA operand fragments,
 *   Upk[p][s][n][b][32 h + i][t] = U[p][16 s + 8 b + 4 h + t][32 n + i]     (P, Cin/16, Cout/32, 2, 64, 4)
 * eqa_plane_gemm_supported: Cin % 32 == 0 and Cout % 32 == 0.  Pointers 16-byte aligned. */
int eqa_plane_gemm_supported(int Cin, int Cout);
int eqa_plane_gemm(const float* V, const float* Upk, float* M, long long T, int P, int Cin, int Cout, void* stream);

int eqa_winograd_f2k5_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            void* stream);
int eqa_winograd_f2k5_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                             void* stream);
int eqa_winograd_f4k5_input(const float* x, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                            void* stream);
int eqa_winograd_f4k5_output(const float* M, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                             void* stream);
/* Training (autograd through the Winograd convolution; the reference gets it from autograd through R2Conv's conv2d):
 * adjoint of the output transform, dM = A dY A^T per tile, dY:(nimg,OH,OW,C) -> dM:(nimg*TY*TX, P, C).  The filter gradient
 * is then dU[a] = V[:,a]^T dM[:,a] (strided-batched GEMM by the caller, V from eqa_winograd_f{m}k5_input) and the input
 * gradient a forward Winograd convolution of the zero-padded dY with the flipped, transposed filters. */
/* Input transform of x zero-padded by `pad` pixels on every side without materialising the padded tensor (pad = 4 for the
 * input gradient): V:(nimg*TY*TX, P, C), TY = (H + 2 pad - 4)/m, TX = (W + 2 pad - 4)/m. */
int eqa_winograd_f2k5_input_padded(const float* x, float* V, int nimg, int H, int W, int C, int pad, void* stream);
int eqa_winograd_f4k5_input_padded(const float* x, float* V, int nimg, int H, int W, int C, int pad, void* stream);
int eqa_winograd_f2k5_output_adjoint(const float* dY, float* dM, int nimg, int OH, int OW, int C, void* stream);
int eqa_winograd_f4k5_output_adjoint(const float* dY, float* dM, int nimg, int OH, int OW, int C, void* stream);
/* Output transform fused with eqa_window_sums_nhwc of the NEXT layer (kernel size k_next; k_next - 1 a multiple of m:
 * {3, 5} for f2k5, {5} for f4k5): the activation is never written; S:(nimg, C, k_next, k_next) fp64 window sums of
 * [relu](A^T M A + bias).  workspace: eqa_winograd_f2k5_output_sums_workspace_bytes(nimg, OH, C, k_next) bytes (both m). */
int64_t eqa_winograd_f2k5_output_sums_workspace_bytes(int nimg, int OH, int C, int k_next);
int eqa_winograd_f2k5_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg,
                                  int OH, int OW, int C, int k_next, void* stream);
int eqa_winograd_f4k5_output_sums(const float* M, const float* bias, int relu, double* S, void* workspace, int nimg,
                                  int OH, int OW, int C, int k_next, void* stream);

/*
 * I2a, the same 5x5 stride-1 group convolutions (escnn_networks.py:67-91) as an overlap-save FFT convolution: 48x48 real
 * FFT tiles give 44x44 outputs each (tiles per axis: eqa_fft48k5_tiles(n) = ceil((n-4)/44)).  Of the 48 x 25 complex
 * frequencies (ky, kx <= 24) of a real tile, F = eqa_fft48k5_frequencies() = 1154 are stored: f = 23 ky + kx - 1 for 0 < kx < 24,
 * then 1104 + 2 ky + (kx == 24) for kx in {0, 24} and ky <= 24 (the rest of those two columns are conjugates).  2.4 real
 * multiplies per output where the tiles fit (Winograd F(4x4,5x5): 4).  Channels-last, fp32.
 *   eqa_fft48k5_input   x:(nimg,H,W,C) -> V:(F, M, 2C), M = nimg*TY*TX tiles, V[f][m] = the tile's spectrum over the C
 *                       channels, [Re x G | Im x G] per group of G = eqa_fft48k5_group(C, 0) channels (16, or 1 = interleaved); d = in_relu ? max(x + in_bias[c], 0) : x + in_bias[c] applied while loading.
 *                       T: workspace of eqa_fft48k5_workspace_bytes(nimg, H, W - 4, C) bytes (the two passes run on chunks
 *                       of images whose intermediate stays cache-resident).
 *   [ batched fp32 GEMM by the caller: Mo[f] = V[f] (M x 2Cin) . B[f] (2Cin x 2Cout), B[f] = [[Br, Bi], [-Bi, Br]] with
 *     Br + i Bi = conj(FFT48x48(filter[co][ci]))[ky][kx] / 2304, rows ordered like the rows of V, columns like the rows of
 *     Mo: complex numbers with re and im interleaved (group size eqa_fft48k5_group(Cout, 1) = 1) ]
 *   eqa_fft48k5_output  Mo:(F, M, 2C) -> y:(nimg,OH,OW,C) = [relu](ifft + bias); T2: workspace of
 *                       eqa_fft48k5_workspace_bytes(nimg, OH, OW, C) bytes.
 *   eqa_fft48k5_output_sums  ... -> S:(nimg,C,k_next,k_next) fp64, the window sums of eqa_window_sums_nhwc of that output
 *                       (k_next in {3,5}); workspace: nimg*OH*TX*C*(2*k_next-1) floats, TX = ceil(OW/44).
 */
int64_t eqa_fft48k5_tiles(int n);
/* Rows per stored frequency of every spectra buffer (V, Mo and the gradient-side G, Cg): tiles | 1.  "(F, M, 2C)" above
 * means M used rows at this pitch; the batched GEMMs take the pitch as their batch stride.  (An even tile count such as 1024
 * x 256 channels puts the frequencies of a tile exactly 2 MB apart, all in one HBM channel: 1.23 -> 1.01 ms for the inverse.) */
int64_t eqa_fft48k5_tile_pitch(int64_t tiles);
int eqa_fft48k5_frequencies(void);
/* bank:(Cout,Cin,5,5) -> B:(F, 2Cin, 2Cout) as described above (fp64 accumulation; cheap enough to run per training step);
 * correlate = 0: FFT(filter)/2304 instead of its conjugate (a convolution: the input gradient, with bank = the filters with
 * their channel axes swapped, (Cin,Cout,5,5)) */
int eqa_fft48k5_filter_spectra(const float* bank, float* B, int Cout, int Cin, int correlate, void* stream);
/* The channel contraction Mo[f] = V[f] . B[f] (complex, per stored frequency) as a hand-written batched complex GEMM on the fp32
 * matrix cores in the 3-multiplication form (csrc/cgemm3m.hip): T1 = Ar.Br, T2 = Ai.Bi, T3 = (Ar+Ai).(Br+Bi), Cr = T1 - T2,
 * Ci = T3 - T1 - T2 -- 25 % fewer MFMA flops than the real [M x 2Cin].[2Cin x 2Cout] product the GEMM library runs.  It is the
 * arithmetic of the reference layer's dense conv2d (escnn_networks.py:67-91 through R2Conv) carried out in the frequency domain.
 *   eqa_fft48k5_cgemm3m_supported  1 when Cin % 32 == 0 and Cout % 64 == 0 (other shapes: the caller's library GEMM on B).
 *   eqa_fft48k5_spectra3m_floats   floats of B3 = F * Cin * Cout * 3.
 *   eqa_fft48k5_filter_spectra3m   bank:(Cout,Cin,5,5) -> B3:(F, Cin/16, Cout/32, 3 [Br|Bi|Br+Bi], 2, 64, 4) in the operand
 *                                  fragment order of the kernel (same values as eqa_fft48k5_filter_spectra, `correlate` likewise).
 *   eqa_fft48k5_cgemm3m            V:(F, pitch(M), 2Cin) rows [Re x 16 | Im x 16] per 16 channels, B3 -> Mo:(F, pitch(M), 2Cout)
 *                                  interleaved complex; rows >= M of a frequency are neither read nor written. */
int eqa_fft48k5_cgemm3m_supported(int Cin, int Cout);
int64_t eqa_fft48k5_spectra3m_floats(int Cin, int Cout);
int eqa_fft48k5_filter_spectra3m(const float* bank, float* B3, int Cout, int Cin, int correlate, void* stream);
int eqa_fft48k5_cgemm3m(const float* V, const float* B3, float* Mo, int64_t M, int Cin, int Cout, void* stream);
/* The same contraction on the bf16 matrix cores with fp32 semantics (csrc/cgemm3m_bf16.hip): every fp32 operand is split without
 * error into three bf16 pieces (8 significant bits each), the product of two operands is the sum of the products of their pieces
 * -- each exact in the matrix core -- accumulated in fp32.  terms = 9: every piece product (exact products, fp32 accumulation: the
 * contract of the fp32 matrix instruction, in another summation order); terms = 6: without the three products of relative size
 * <= 2^-24 (error <= 2^-23 |a||b| per product).  Same reference arithmetic as eqa_fft48k5_cgemm3m (escnn_networks.py:67-91).
 *   eqa_fft48k5_spectra3m_bf16_bytes  bytes of Bp = F * Cin * Cout * 3 parts * 3 pieces * 2.
 *   eqa_fft48k5_spectra3m_split       B3 (the operand of eqa_fft48k5_cgemm3m) -> Bp:(F, Cin/16, Cout/32, 3, 3 pieces, 64, 8) bf16.
 *   eqa_fft48k5_cgemm3m_bf16x3        V, Bp -> Mo; shapes and layouts of V / Mo as for eqa_fft48k5_cgemm3m.  Two kernels: a wave
 *                                     per 64 x 64 tile (any supported shape), and -- Cout % 128 == 0 -- a block per 128 x 128 tile
 *                                     with A split once per block and shared through LDS (a third of the vector-L1 loads per
 *                                     matrix instruction; eqa_set_option key 2 = 1 forces the wave form). */
int64_t eqa_fft48k5_spectra3m_bf16_bytes(int Cin, int Cout);
int eqa_fft48k5_spectra3m_split(const float* B3, void* Bp, int Cin, int Cout, void* stream);
int eqa_fft48k5_cgemm3m_bf16x3(const float* V, const void* Bp, float* Mo, int64_t M, int Cin, int Cout, int terms, void* stream);
/* The same contraction on TWO fp16 pieces per fp32 operand (csrc/cgemm3m_bf16.hip, TERMS = 3): x = h1 + h2 + d with h1 = rn16(s x),
 * h2 = rn16(s x - h1), |d| <= 2^-23 |x| (s: a power of two that takes the operand's bound to 2^14); a product is h1 k1 + h1 k2 + h2 k1,
 * each exact in v_mfma_f32_32x32x16_f16, fp32 accumulate -- three matrix instructions per product instead of six, and closer to an
 * fp64 product than the six-product bf16 form or the fp32 instruction (profiles/r06/f16x2_gemm_check.txt, kbench_gemm_error.txt).
 * The caller bounds the operands: vbound[0 .. nbound) (device memory, read by the kernel: no host synchronisation) holds numbers
 * whose maximum is >= every |Re|, |Im| of V -- for the spectra of NON-NEGATIVE activations the DC bins (eqa_lift5_fft48k5_input_dcmax);
 * values more than 2^16 below the bound lose precision gradually (fp16 subnormals).  b_scale: the power of two the filter spectra
 * were multiplied by when split (max |B3| b_scale <= 2^14).  Same reference arithmetic as eqa_fft48k5_cgemm3m (escnn_networks.py:67-91).
 *   eqa_fft48k5_spectra3m_f16_bytes   bytes of Bh = F * Cin * Cout * 3 parts * 2 pieces * 2; 0: shape not taken (Cout % 128 != 0).
 *   eqa_fft48k5_spectra3m_split_f16   B3 -> Bh:(F, Cin/16, Cout/32, 3, 2 pieces, 64, 8) fp16 of b_scale * B3.
 *   eqa_fft48k5_cgemm3m_f16x2         V, Bh -> Mo; shapes and layouts of V / Mo as for eqa_fft48k5_cgemm3m. */
int64_t eqa_fft48k5_spectra3m_f16_bytes(int Cin, int Cout);
int eqa_fft48k5_spectra3m_split_f16(const float* B3, void* Bh, int Cin, int Cout, float b_scale, void* stream);
int eqa_fft48k5_cgemm3m_f16x2(const float* V, const void* Bh, float* Mo, int64_t M, int Cin, int Cout, const float* vbound, int nbound,
                              float b_scale, void* stream);
/* The filter gradient's contraction (training), replacing the library's real [2Cin x M].[M x 2Cout] product (autograd through the
 * same R2Conv): D[f] = V[f]^T . conj(G[f]) over the M tiles in the 3-multiplication form on the fp32 MFMA.  V:(F, M|1, 2Cin),
 * G:(F, M|1, 2Cout) as eqa_fft48k5_input / _grad_transform write them; D3:(F, Cin, 2, Cout) = Dr | Di per input channel, plain
 * channel order; Cin, Cout % 64 == 0.  eqa_fft48k5_filter_grad3m: eqa_fft48k5_filter_grad on that form. */
int eqa_fft48k5_wgrad3m_supported(int Cin, int Cout);
int eqa_fft48k5_wgrad3m(const float* V, const float* G, float* D, int64_t M, int Cin, int Cout, void* stream);
int eqa_fft48k5_filter_grad3m(const float* D, float* dbank, int Cout, int Cin, void* stream);
int eqa_fft48k5_group(int C, int side);
int64_t eqa_fft48k5_workspace_bytes(int nimg, int rows, int out_cols, int C);
int eqa_fft48k5_input(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                      void* stream);
/* eqa_fft48k5_input on a map stored (nimg, C/16, H, W, 16) (eqa_lift_conv_grouped); C % 16 == 0, fused kernel only
 * (eqa_fft48k5_input_grouped_supported(C) == 1), else EQA_ERR_UNSUPPORTED. */
int eqa_fft48k5_input_grouped_supported(int C);
int eqa_fft48k5_input_grouped(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C,
                              void* stream);
/* The lifting layer FUSED into the input transform of the layer behind it (round 6, csrc/lift_fft.hip; inference):
 *   V = eqa_fft48k5_input( [relu]( conv2d(x, bank) + bias ) )
 * without the lifted map ever existing in memory -- a persistent block owns (48 x 48 tile, 16-channel group) items, computes the
 * tile from its 52 x 52 x 3 input patch on the fp32 matrix cores (v_mfma_f32_16x16x4_f32; the 76 products of a pixel in the order
 * of eqa_lift_conv_nhwc's dense form), and transforms it in place.  Reference layers: escnn_networks.py:60-85 (R2Conv trivial ->
 * regular, InnerBatchNorm folded into bank / bias, ReLU) + the input side of the first regular -> regular R2Conv.
 * x:(nimg,H0,W0,3) channels-last; bank:(Cout,5,5,3) = the MEMORY ORDER of a channels-last (Cout,3,5,5) tensor; bias:(Cout) or NULL;
 * V:(F, M|1, 2 Cout), M = nimg * tiles(H0-4) * tiles(W0-4), as eqa_fft48k5_input writes it.  Cout % 16 == 0 (else
 * EQA_ERR_UNSUPPORTED); sizes beyond 32-bit byte offsets: EQA_ERR_UNSUPPORTED (use the two calls). */
int eqa_lift5_fft48k5_input_supported(int Cin, int KH, int KW, int Cout);
int eqa_lift5_fft48k5_input(const float* x, const float* bank, const float* bias, int relu, float* V, int nimg, int H0, int W0, int Cout,
                            void* stream);
/* The same launch with relu != 0 (EQA_ERR_UNSUPPORTED otherwise), which also writes dcmax[0 .. EQA_LIFT5_DCMAX_SLOTS): per block of
 * the persistent grid the largest DC bin (kx = 0, ky = 0: the sum of a tile's non-negative activations) it stored, 0 in the unused
 * slots.  |X[k]| <= X[0] for a non-negative signal, so the maximum of the slots bounds every |Re|, |Im| of V: the `vbound` of
 * eqa_fft48k5_cgemm3m_f16x2, produced without a pass over V and consumed without a host synchronisation. */
#define EQA_LIFT5_DCMAX_SLOTS 256
int eqa_lift5_fft48k5_input_dcmax(const float* x, const float* bank, const float* bias, int relu, float* V, float* dcmax, int nimg, int H0,
                                  int W0, int Cout, void* stream);
/* The same kernel with its convolution on TWO fp16 pieces per fp32 value, three exact products on v_mfma_f32_16x16x32_f16 (the contract of
 * eqa_fft48k5_cgemm3m_f16x2: every operand within one fp32 ulp, fp32 accumulation): fifteen matrix instructions of 16 cycles per
 * 16-pixel tile instead of 19 of 32 on the vector ALU's datapath, and a quarter of the LDS operand reads.
 *   wpieces  (Cout, 2 pieces, 5 filter rows, 4 chunks, 8) fp16 of w_scale * bank, eqa_lift5_pieces_f16_bytes(Cout) bytes: chunk p < 3 of
 *            filter row ky = [w(ci 0..2, kx = 2p - 1), 0, w(ci 0..2, kx = 2p), 0] (kx = -1: 0), chunk 3 = 0; w_scale a power of two with
 *            max |bank| w_scale <= 2^14.
 *   xbound   nxbound device floats whose maximum bounds |x| (eqa_absmax_slots writes EQA_LIFT5_DCMAX_SLOTS of them; read by the kernel,
 *            no host synchronisation); the pixels are scaled by the power of two that takes it to 2^14.
 *   dcmax    null, or as in eqa_lift5_fft48k5_input_dcmax (relu != 0 required then). */
int64_t eqa_lift5_pieces_f16_bytes(int Cout);
int eqa_absmax_slots(const float* x, int64_t n, float* slots, void* stream);
int eqa_lift5_fft48k5_input_f16x2(const float* x, const void* wpieces, float w_scale, const float* xbound, int nxbound, const float* bias,
                                  int relu, float* V, float* dcmax, int nimg, int H0, int W0, int Cout, void* stream);
/* The same with the convolution on the bf16 matrix cores: every fp32 pixel and weight split exactly into three bf16 pieces, six piece
 * products per product, fp32 accumulation (the contract of eqa_fft48k5_cgemm3m_bf16x3; the fp32 matrix instruction of the form above
 * runs on the vector ALU's datapath and holds the transforms' vector work back).  wpieces: the weights' pieces, (Cout, 3 pieces, 16
 * chunks, 8) bf16 = eqa_lift5_pieces_bytes(Cout) bytes; chunk c < 15 = (filter row c / 3, pixel pair p = c % 3): [w(ci 0..2, kx = 2 p), 0,
 * w(ci 0..2, kx = 2 p + 1), 0] with kx = 5 -> 0; chunk 15 = 0; piece 0 = bf16(w), piece 1 = bf16(w - p0), piece 2 = bf16(w - p0 - p1). */
int64_t eqa_lift5_pieces_bytes(int Cout);
int eqa_lift5_fft48k5_input_bf16x3(const float* x, const void* wpieces, const float* bias, int relu, float* V, int nimg, int H0, int W0,
                                   int Cout, void* stream);
int eqa_fft48k5_output(const float* Mo, float* T2, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C,
                       void* stream);
/* Training: eqa_fft48k5_output without bias / activation that also leaves the fp64 partial sums of the InnerBatchNorm behind the
 * layer (escnn_networks.py:67-91): sum over rows r of partial[(r * C + c) * 2 + {0, 1}] = sum, sum of squares of y[.., c], taken
 * from the values the inverse transform's consumer waves store.  rows = eqa_fft48k5_output_stats_rows(...) (0: the pipelined
 * inverse does not take this shape -- use eqa_fft48k5_output + eqa_bn_stats_nhwc); partial: rows * C * 2 doubles. */
int64_t eqa_fft48k5_output_stats_rows(int nimg, int OH, int OW, int C);
int eqa_fft48k5_output_stats(const float* Mo, float* T2, float* y, double* partial, int nimg, int OH, int OW, int C, void* stream);
/* Training, filter gradient in the frequency domain (the reference gets it from autograd through R2Conv's conv2d):
 *   eqa_fft48k5_grad_transform  dy:(nimg,OH,OW,C) -> G:(F, M, 2C): spectra of the DISJOINT 44 x 44 output-gradient tiles,
 *                               zero-padded to 48 x 48, same row layout as V; T: eqa_fft48k5_workspace_bytes(nimg, OH, OW, C).
 *   [ batched GEMM by the caller: D[f] = V[f]^T (2Cin x M) . G[f] (M x 2Cout), V from eqa_fft48k5_input of the layer's input ]
 *   eqa_fft48k5_filter_grad     D:(F, 2Cin, 2Cout) -> dbank:(Cout,Cin,5,5) = d loss / d filter (fp64 accumulation).
 */
int eqa_fft48k5_grad_transform(const float* dy, float* T, float* G, int nimg, int OH, int OW, int C, void* stream);
int eqa_fft48k5_filter_grad(const float* D, float* dbank, int Cout, int Cin, void* stream);
/* Training, input gradient: Cg:(F, M, 2C) = G[f] . B2[f] (B2 = eqa_fft48k5_filter_spectra of the channel-swapped bank with
 * correlate = 0) -> dx:(nimg,H,W,C), H = OH + 4: every 44 x 44 gradient tile yields a 48 x 48 block, blocks overlap by 4 and are
 * added in a fixed order.  T2: eqa_fft48k5_workspace_bytes(nimg, 48*TY, OW, C) bytes. */
int eqa_fft48k5_input_grad(const float* Cg, float* T2, float* dx, int nimg, int H, int W, int C, void* stream);
int eqa_fft48k5_output_sums(const float* Mo, float* T2, const float* bias, int relu, double* S, void* workspace, int nimg,
                            int OH, int OW, int C, int k_next, void* stream);

/*
 * I2a / I2b for the kernel sizes other than 5 (round 4).  The reference's kernel_size is a free constructor argument
 * (escnn_networks.py:19-44, custom_equivariant_networks.py:25-62): its tutorial trains ESCNNEquivariantNetwork with k = 9
 * (tutorials/images/understanding_discrete_canonicalization.ipynb cell 17), its own test builds k = 3
 * (tests/images/canonicalization/test_discrete_group.py:31-38).  Same overlap-save scheme as eqa_fft48k5_*, with
 * O = 49 - ksize outputs per 48 x 48 tile (46 / 44 / 42 / 40 for ksize 3 / 5 / 7 / 9; tiles per axis eqa_fft48_tiles(n, ksize) =
 * ceil((n - ksize + 1) / O)), the same stored frequencies, buffer pitch (eqa_fft48k5_tile_pitch) and row layouts, so the
 * k-independent contractions (eqa_fft48k5_cgemm3m, eqa_fft48k5_wgrad3m, or the caller's batched GEMM) apply unchanged.
 * The forward, gradient and output transforms run fused (row pass -> LDS -> column pass in one block) where C % 16 == 0, as two
 * passes otherwise; arguments as their eqa_fft48k5_* namesakes plus `ksize`:
 *   eqa_fft48_supported        1 for ksize in {3, 5, 7, 9}
 *   eqa_fft48_workspace_bytes  T / T2 sizes (out_cols = the OUTPUT width of the convolution, as for k = 5)
 *   eqa_fft48_filter_spectra / _filter_spectra3m   bank:(Cout,Cin,ksize,ksize) -> B / B3
 *   eqa_fft48_input            x:(nimg,H,W,C) -> V            eqa_fft48_output       Mo -> y:(nimg,OH,OW,C) = [relu](ifft + bias)
 *   eqa_fft48_grad_transform   dy -> G (disjoint O x O tiles)  eqa_fft48_input_grad   Cg -> dx:(nimg,OH+ksize-1,OW+ksize-1,C)
 *   eqa_fft48_filter_grad      D -> dbank:(Cout,Cin,ksize,ksize); packed = 1: D in the eqa_fft48k5_wgrad3m form
 */
int eqa_fft48_supported(int ksize);
int64_t eqa_fft48_tiles(int n, int ksize);
int64_t eqa_fft48_workspace_bytes(int nimg, int rows, int out_cols, int C, int ksize);
int eqa_fft48_filter_spectra(const float* bank, float* B, int Cout, int Cin, int ksize, int correlate, void* stream);
int eqa_fft48_filter_spectra3m(const float* bank, float* B3, int Cout, int Cin, int ksize, int correlate, void* stream);
int eqa_fft48_input(const float* x, float* T, float* V, const float* in_bias, int in_relu, int nimg, int H, int W, int C, int ksize,
                    void* stream);
int eqa_fft48_grad_transform(const float* dy, float* T, float* G, int nimg, int OH, int OW, int C, int ksize, void* stream);
int eqa_fft48_output(const float* Mo, float* T2, const float* bias, int relu, float* y, int nimg, int OH, int OW, int C, int ksize,
                     void* stream);
int eqa_fft48_input_grad(const float* Cg, float* T2, float* dx, int nimg, int H, int W, int C, int ksize, void* stream);
int eqa_fft48_filter_grad(const float* D, float* dbank, int Cout, int Cin, int ksize, int packed, void* stream);

/* I4 alone: gidx[b] = argmax_g act[b,g] (first index on ties); act:(B,G). */
int eqa_group_argmax(const float* act, int32_t* gidx, int B, int G, void* stream);

/*
 * P4 -- SO(3) action on point clouds:  y[b] = R[b] x[b]   (transpose = 0)  or  R[b]^T x[b] (transpose = 1).
 * Replaces torch.bmm(x^T, R^T)^T at equiadapt/pointcloud/canonicalization/continuous_group.py:74-79.
 * x,y:(B,3,N); R:(B,3,3).
 */
int eqa_so3_rotate(const float* x, const float* R, float* y, int B, int N, int transpose, void* stream);

/*
 * P1 + P2 -- fused VNSmall forward (eval mode): kNN graph, cross edge features, the vector-neuron layers and the
 * mean over points in one kernel; a cloud is staged once in LDS, every intermediate lives in registers.
 * Replaces equiadapt/pointcloud/canonicalization_networks/equivariant_networks.py:15-76 (knn,
 * get_graph_feature_cross) and :128-150 (VNSmall.forward) with vector_neuron_layers.py:251-273, :303-324.
 * x:(B,3,N); out:(B,3,3) = mean over points of the first 3 output vector channels; 1 <= k <= 32 (and k <= N); pooling 0 = "mean",
 * 1 = "max" (VNMaxPool, vector_neuron_layers.py:327-364: per point and channel the edge maximising <x, W_p x>, first index on
 * ties like torch.max).
 * params: EQA_VNSMALL_PARAMS floats, batch-norms folded to scale/shift of the vector norm (layout in csrc/pointcloud.hip);
 * pooling 1: + the 441 floats of the pooling layer's map_to_dir weight (EQA_VNSMALL_PARAMS_MAX in total).
 * workspace: eqa_vnsmall_workspace_bytes(B, N) bytes.  k > 32: EQA_ERR_UNSUPPORTED (the host keeps an op-by-op path).
 * Two kernels: four lanes per point (any k; the default) and one thread per point (k = 20; eqa_set_option key 1).
 */
#define EQA_VNSMALL_PARAMS 1310
#define EQA_VNSMALL_PARAMS_MAX 1751
int64_t eqa_vnsmall_workspace_bytes(int B, int N);
int eqa_vnsmall_fwd(const float* x, const float* params, float* out, void* workspace, int B, int N, int k, int pooling,
                    void* stream);
/* The eval-mode point-cloud canonicalizer in two launches: eqa_vnsmall_fwd's network kernel, then ONE kernel per batch that finishes
 * the network output (vectors:(B,3,3)), orthonormalises it (R:(B,3,3), eqa_gram_schmidt's arithmetic -- common/utils.py:22-51) and
 * writes the canonical cloud y:(B,3,N) = R x (eqa_so3_rotate's -- pointcloud/canonicalization/continuous_group.py:51-81, :107-134).
 * Same arguments and limits as eqa_vnsmall_fwd otherwise; replaces three launches behind the network kernel. */
int eqa_vnsmall_canonicalize(const float* x, const float* params, float* vectors, float* R, float* y, void* workspace, int B, int N,
                             int k, int pooling, void* stream);

/*
 * Training passes of VNSmall's first block (kNN graph -> cross edge features -> VNLinearLeakyReLU(3 -> 21, slope 0) with
 * training-mode VN batch-norm -> mean over the k <= 32 neighbours; equivariant_networks.py:15-76, :128-140,
 * vector_neuron_layers.py:251-273, :303-324), forward and the autograd backward w.r.t. the parameters.  Nothing of size
 * (B, 21, 3, N, k) is materialised: every pass re-derives the edge features from the cloud and the neighbour indices.
 * x:(B,3,N); idx:(B,N,k) int32; Wf, Wd:(21,3) = conv_pos.map_to_feat / map_to_dir weights; per-channel (21) vectors:
 * scale = gamma*rstd, shift = beta - mean*scale, mean, rstd of n = |W_f f| + 1e-6 over all B*N*k edges.
 * Partials are per block, blocks = B * eqa_vn_blocks(N), summed by the caller (fixed order: deterministic).
 *   eqa_vn_knn                 idx <- the k nearest neighbours of every point (self included), best first
 *   eqa_vn_convpos_stats       partial:(blocks, 21, 2) = sum n, sum n^2
 *   eqa_vn_convpos_fwd         pooled:(B, 21, 3, N)
 *   eqa_vn_convpos_bwd_reduce  partial:(blocks, 21, 2) = sum g, sum g*nhat   (g = dL/d BN output; = d beta, d gamma)
 *   eqa_vn_convpos_bwd_apply   partial:(blocks, 21, 6) = d W_f[c][0..2], d W_d[c][0..2];  m1 = sum g / M, m2 = sum g nhat / M
 *                              (M = B*N*k; zeros when the batch-norm uses running statistics)
 */
int eqa_vn_blocks(int N);
int eqa_vn_knn(const float* x, int32_t* idx, int B, int N, int k, void* stream);
int eqa_vn_convpos_stats(const float* x, const int32_t* idx, const float* Wf, float* partial, int B, int N, int k, void* stream);
int eqa_vn_convpos_fwd(const float* x, const int32_t* idx, const float* Wf, const float* Wd, const float* scale, const float* shift,
                       float* pooled, int B, int N, int k, void* stream);
int eqa_vn_convpos_bwd_reduce(const float* x, const int32_t* idx, const float* Wf, const float* Wd, const float* scale,
                              const float* shift, const float* mean, const float* rstd, const float* gpool, float* partial, int B,
                              int N, int k, void* stream);
int eqa_vn_convpos_bwd_apply(const float* x, const int32_t* idx, const float* Wf, const float* Wd, const float* scale,
                             const float* shift, const float* mean, const float* rstd, const float* m1, const float* m2,
                             const float* gpool, float* partial, int B, int N, int k, void* stream);

/*
 * Training passes of VNSmall's tail on the pooled features of the first block: conv1 = VNLinearLeakyReLU(21 -> 21, slope 0) ->
 * bn1 = VNBatchNorm(21) -> conv2 = VNLinearLeakyReLU(21 -> 4, slope 0) -> dropout -> mean over the points
 * (equivariant_networks.py:141-150; vector_neuron_layers.py:251-273, :303-324), the three batch-norms in training mode, forward
 * and the backward w.r.t. the parameters and the pooled features.  Every pass recomputes the chain from the 63 floats of a
 * point; nothing of size (B, 21, 3, N) is stored between the layers.  One launch per pass, per-block partials
 * (blocks = B * eqa_vn_tail_blocks(N), eqa_vn_tail_partial_floats(pass) floats each) reduced by the finalize kernels or the caller:
 *   pooled:(B,21,3,N)   weights:(1050) = conv1.map_to_feat (21x21) | conv1.map_to_dir (21x21) | conv2.map_to_feat (4x21) |
 *   conv2.map_to_dir (4x21)   stat:(3,128) per batch-norm (conv1, bn1, conv2): scale[32] | shift[32] | mean[32] | rstd[32]
 *   red:(3,64) per batch-norm: m1[32] | m2[32] (sum g / M, sum g nhat / M)   mask:(B,4,3,N) dropout factors or NULL
 *   gout:(B,4,3) gradient of the mean over the points (rows of unused channels zero)   g_pooled:(B,21,3,N)
 *   pass 0, 1, 2   partial = sum n, sum n^2 per channel of conv1 / bn1 / conv2 (each needs the stat of the layers before it)
 *   pass 3         partial:(blocks, 12) = sum over the block's points of the dropped-out conv2 output (the caller divides by N)
 *   pass 4, 5, 6   partial[0 .. 2C) = sum g, sum g nhat of conv2 / bn1 / conv1 (each needs red of the layers after it);
 *                  pass 5 also partial[42 .. 210) = d conv2.map_to_feat | d conv2.map_to_dir,
 *                  pass 6 also partial[42 .. 483) = d conv1.map_to_dir
 *   pass 7         partial:(blocks, 441) = d conv1.map_to_feat;  g_pooled written
 *   eqa_vn_bn_finalize      partial:(nblk, stride) sums -> stat of one batch-norm over M samples; running_mean / running_var
 *                           (may be NULL) updated with the unbiased variance and `momentum`, *num_batches_tracked += 1
 *   eqa_vn_bn_bwd_finalize  partial:(nblk, stride) sums -> grads = d beta[32] | d gamma[32], red = m1[32] | m2[32]
 */
int eqa_vn_tail_blocks(int N);
int eqa_vn_tail_partial_floats(int pass);
int eqa_vn_tail_pass(int pass, const float* pooled, const float* weights, const float* stat, const float* red, const float* mask,
                     const float* gout, float* partial, float* g_pooled, int B, int N, void* stream);
int eqa_vn_bn_finalize(const float* partial, int nblk, int stride, int C, long long M, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* stat,
                       void* stream);
int eqa_vn_bn_bwd_finalize(const float* partial, int nblk, int stride, int C, long long M, float* grads, float* red, void* stream);

/*
 * P3 -- batched 3x3 classical Gram-Schmidt on rows (no epsilon, no handedness fix).
 * Replaces equiadapt/common/utils.py:22-51.   v,out:(B,3,3).
 * fp32 in, fp32 out; the three steps are evaluated in fp64 and rounded once (the step has no epsilon and amplifies rounding by the
 * conditioning of the three vectors: an fp32 evaluation -- the reference's included -- sits 4e-5..6e-5 from the exact frame at
 * cond(V) = 540; this one is the exact Gram-Schmidt of its input to the last fp32 bit).  Same for eqa_modified_gram_schmidt.
 */
int eqa_gram_schmidt(const float* v, float* out, int B, void* stream);
/* its backward (training): grad_out:(B,3,3) = dL/d out -> grad_v:(B,3,3) = dL/d v, the analytic derivative of the three steps */
int eqa_gram_schmidt_bwd(const float* v, const float* grad_out, float* grad_v, int B, void* stream);

/*
 * (f).4 -- E(3) canonicalization of n-body systems (equiadapt/nbody/canonicalization/euclidean_group.py):
 *   eqa_modified_gram_schmidt  rows of (B,3,3), modified GS (:139-157);
 *   eqa_rigid_rows             per node m: mode 0  out = x R + t  (invert_canonicalization :126-137),
 *                                          mode 1  out = x R^T - t R^T  (canonicalize :108-124); t may be NULL.
 * x,t,out:(M,3) row vectors; R:(M,3,3).
 */
int eqa_modified_gram_schmidt(const float* v, float* out, int B, void* stream);
int eqa_rigid_rows(const float* x, const float* R, const float* t, float* out, int M, int mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EQA_HIP_H */
