/* mvsnerf_hip.h - C ABI of libmvsnerf_hip.so (gfx950 / MI355X): the STABLE tier.
 *
 * The reference (apchenstu/mvsnerf) has no FFI layer: its hot path is Python calling implicit
 * ATen/cuDNN kernels.  This header is the boundary a maintainer would bind instead (ctypes stub in
 * INTEGRATION.md).  Each entry point names the reference call site(s) whose arithmetic it replaces
 * (paths under /root/reference).
 *
 * Two tiers (ABI v12).  THIS header is the stage-level surface SURVEY.md 8(b) asks for - plane sweep forward / backward, the generic
 * CostRegNet / FeatureNet layer calls (any shape), volume / colour lookups, MLP pack / forward / backward in every arithmetic, compositing, the
 * one-call ray march (mvsnerf_raymarch_{fwd, fwd_batched, train_fwd, bwd}, mvsnerf_render_pixels_fwd), ray generation, importance sampling,
 * Adam: 70 entries, frozen (tests/test_abi_surface.py holds the list; a change here is an ABI bump).  Everything a scene encode or a training
 * step needs can be written against it.  include/mvsnerf_hip_internal.h declares the 66 further exports that mvsnerf_amd's own host layer
 * drives its TUNED layer loop with - per-shape matrix-core convolutions and their *_tiles / *_parts / *_packed_elems queries, blocked /
 * bf16 / two-piece-fp16 cost-volume layouts, multi-job pack and reduction helpers, the guarded conv sequences: plumbing that moves with the
 * kernels and carries no stability promise.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  Every pointer is a DEVICE pointer (fp32 unless noted)
 *     owned by the caller; outputs are caller-allocated and fully written.  The library allocates
 *     nothing and has no behavioural state: its only process-global data are per-device "dynamic-LDS cap already
 *     raised" bits (idempotent).  (Which kernel family a dispatcher picks is a compile-time constant, csrc/knobs.h; only the entry points declared in the two headers are exported, csrc/exports.map.)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work.
 *   - return 0 on success, a negative MVSNERF_E* for rejected arguments (nothing was launched),
 *     or a positive hipError_t from the launch.
 *   - volumes are CHANNEL-LAST in HBM: vol[d][y][x][c] (one corner of the 8-channel neural volume is
 *     one 32-byte sector).  The reference's NCDHW tensors are converted once by
 *     mvsnerf_ncdhw_to_ndhwc / produced channel-last directly by the encoder kernels.
 */
#ifndef MVSNERF_HIP_H
#define MVSNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVSNERF_OK 0
#define MVSNERF_EINVAL (-1)  /* bad size / null pointer */
#define MVSNERF_EUNSUPPORTED (-2) /* shape outside what the kernels are specialised for */
#define MVSNERF_EALIGN (-3)  /* pointer not 16-byte aligned where required */

/* Memory order of a neural volume handed to the ray-march entries (ABI v10).
 *   MVSNERF_VOL_DHWC  vol[d][y][x][c]  what the reference's NCDHW tensor becomes with one transpose (mvsnerf_ncdhw_to_ndhwc); learnable
 *                     volumes (RefVolume) keep it, and so do volume GRADIENTS in every case.
 *   MVSNERF_VOL_HWDC  vol[y][x][d][c]  depth fastest: what the scene encoder emits (mvsnerf_abn_apply_add_hwdc).  The samples of a ray
 *                     walk depth, so the consecutive samples of a ray read ONE contiguous run of voxels per (y, x) column instead of a
 *                     64-byte piece of a different 128-byte line per depth plane: the lookup's memory-side traffic halves.
 * Both describe the same logical (C, D, H, W) tensor; results are bit-identical. */
#define MVSNERF_VOL_DHWC 0
#define MVSNERF_VOL_HWDC 1

/* ABI version; bumped on any signature change. */
int mvsnerf_abi_version(void);

/* ---------------------------------------------------------------- layout helpers */

/* (C,D,H,W) -> (D,H,W,C) and back.  Used at the boundary when a caller hands over the reference's
 * NCDHW volume (models.py:930 `volume_feat.reshape(1,-1,D,h,w)`). */
int mvsnerf_ncdhw_to_ndhwc(const float* src, float* dst, int C, int D, int H, int W, void* stream);
int mvsnerf_ndhwc_to_ncdhw(const float* src, float* dst, int C, int D, int H, int W, void* stream);

/* ---------------------------------------------------------------- scene encode (L1a) */

/* [N][C][H][W] -> [N][H][W][Cpad] (channels >= C zero-filled): source feature maps / thumbnails are made
 * channel-last once so that a bilinear tap of the plane sweep is one contiguous vector. */
int mvsnerf_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cpad, void* stream);

/* F.interpolate(mode='bilinear', align_corners=False) of NC planes (models.py:859: images -> feature resolution). */
int mvsnerf_resize_bilinear(const float* src, float* dst, int NC, int Hi, int Wi, int Ho, int Wo, void* stream);

/* Plane-sweep cost volume in one pass: utils.py:580-630 (homo_warp) + models.py:839-893
 * (build_volume_costvar_img, with_img=1) or models.py:787-837 (build_volume_costvar, with_img=0).
 * feats_cl[V][H][W][32], imgs_cl[V][H][W][4] (thumbnails, with_img only), proj[V][3][4] (view 0 unused),
 * depth[D]; cost[D][H+2pad][W+2pad][CP] channel-last:
 *   with_img: [0:3 ref rgb (padded border = 0) | 3v:3v+3 warped src rgb | 3V:3V+32 variance | zeros to CP]
 *             masks[V][D][Hp][Wp] per-view in-frustum masks (view 0 = 1)
 *   else:     [0:32 variance | zeros], masks[D][Hp][Wp] = view count (models.py:814-821). */
int mvsnerf_planesweep_costvar_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                   int V, int C, int H, int W, int D, int pad, float* cost, int CP, float* masks,
                                   int with_img, void* stream);

/* Stand-alone homo_warp (utils.py:580-630) for one source view: src[C][H][W] (NCHW), proj[3][4], depth[D]
 * -> warped[C][D][Hp][Wp], grid_out[D*Hp*Wp][2] (either may reuse a given grid_in, as models.py:872 does). */
int mvsnerf_homo_warp_fwd(const float* src_nchw, const float* proj, const float* depth, const float* grid_in,
                          int C, int H, int W, int D, int pad, float* warped, float* grid_out, void* stream);

/* CostRegNet building blocks (models.py:674-685, 725-769), channel-last activations x[d][y][x][C].
 * A conv input is `leaky_relu(x*scale[c]+shift[c], 0.01)` applied on load (scale == NULL: raw input, no
 * activation), optionally plus a second such tensor (U-Net skip sums, models.py:762-766).
 *   conv3d_pack_weights: packed[tap][ci][co] = w[ci*s_ci + co*s_co + tap] (zero beyond ci_real/co_real; `flip` mirrors
 *            the taps): Conv3d (Cout,Cin,27): s_ci=27, s_co=Cin*27; ConvTranspose3d (Cin,Cout,27): s_ci=Cout*27, s_co=27;
 *            data-gradient kernels are the same convolutions with the roles of the two channel strides swapped
 *   conv3d_fwd: k3, padding 1, stride 1|2, no bias -> raw out[Do][Ho][Wo][Cout]
 *   conv_transpose3d_fwd: k3, stride 2, padding 1, output_padding 1 -> raw out[2D][2H][2W][Cout]
 *   abn_stats: train-mode InPlaceABN statistics of a raw tensor -> per-channel scale/shift
 *              (gamma=|w|+eps, biased variance) and the running_mean/var side effect (may be NULL)
 *   abn_apply_add: materialise leaky(x1*s1+t1) [+ leaky(x2*s2+t2)] */
int mvsnerf_conv3d_pack_weights(const float* w, int ci_real, int co_real, int cin_pad, int cout_pad,
                                int s_ci, int s_co, int flip, float* packed, void* stream);
int mvsnerf_conv3d_fwd(const float* x1, const float* scale1, const float* shift1,
                       const float* x2, const float* scale2, const float* shift2,
                       int Cin, int cin_ld, int D, int H, int W, const float* wpacked, int Cout, int stride,
                       float* out, void* stream);

/* The same convolution for the deep layers (Cout = 32 | 64; models.py:758-761: conv3..conv6, and the data gradients that have these
 * shapes) on v_mfma_f32_32x32x2_f32.  conv3d_mfma_supported: 1 when (Cin, Cout, stride) is built (and the "conv_mfma" switch is on);
 * conv3d_pack_weights_mfma: packed[tap][ci][co] (mvsnerf_conv3d_pack_weights) -> w32[tap][ci/8][co][8]; one lazily-activated source. */

/* The input is leaky(x1*scale1+shift1) [+ leaky(x2*scale2+shift2)] like mvsnerf_conv_transpose3d_fwd's (scale NULL: plain tensor): the
 * pending InPlaceABN of the producers and the U-Net skip sum are applied while staging, no materialised input.
 * Both 8-channel producers can leave the InPlaceABN statistics of their output as per-workgroup partial sums (stats_part: 2 * 8 floats per
 * workgroup, *_tiles(...) workgroups; NULL = not wanted) so that the full-resolution output is not read again; mvsnerf_abn_finalize is
 * stage 2 of mvsnerf_abn_stats on such partials (part[({sum, sum of squares} * C + c) * n_blocks + b]: channel-major, each channel's n_blocks partial sums contiguous). */

int mvsnerf_conv_transpose3d_fwd(const float* x1, const float* scale1, const float* shift1,
                                 const float* x2, const float* scale2, const float* shift2,
                                 int Cin, int D, int H, int W, const float* wpacked, int Cout, float* out, void* stream);
size_t mvsnerf_abn_workspace_floats(int C);
int mvsnerf_abn_stats(const float* x, int64_t n_vox, int C, const float* weight, const float* bias,
                      float* running_mean, float* running_var, float momentum, float eps,
                      float* scale, float* shift, float* mean_out, float* invstd_out, float* workspace, void* stream);
int mvsnerf_abn_apply_add(const float* x1, const float* scale1, const float* shift1,
                          const float* x2, const float* scale2, const float* shift2,
                          int64_t n_vox, int C, float* out, void* stream);
/* The same sum for the 8-channel neural volume (conv0 + conv11(x), models.py:766), WRITTEN depth-fastest: x1, x2 [D][H][W][8] ->
 * out[H][W][D][8] (MVSNERF_VOL_HWDC: what the ray march reads best).  Tiles of 32 depth planes x 16 columns go through LDS, so that
 * both the reads (along x) and the writes (along depth) are contiguous runs. */
int mvsnerf_abn_apply_add_hwdc(const float* x1, const float* scale1, const float* shift1,
                               const float* x2, const float* scale2, const float* shift2,
                               int D, int H, int W, float* out, void* stream);

/* ---- encoder backward (generalizable training, train_mvs_nerf_pl.py:104-168) ----
 * Data gradients of the convolutions reuse the forward kernels with re-packed weights (mvsnerf_conv3d_pack_weights:
 * `flip` mirrors the taps; a stride-2 conv's data gradient is the transposed conv and vice versa).
 *   abn_bwd: train-mode InPlaceABN backward of one layer: x raw, (scale,shift,mean,invstd) from abn_stats, upstream
 *            gradient g1 (+ g2) w.r.t. the ACTIVATED output -> gx (w.r.t. the raw conv output), g_weight, g_bias.
 *            workspace: mvsnerf_abn_workspace_floats(C).
 *   conv3d_wgrad: gW[a][b][tap] = sum_o G[o][a] * X[o*stride-1+tap][b]; G on the conv's output grid, X on its input
 *            grid, each optionally lazily activated / a sum of two tensors.  Conv3d: G = grad of raw output, X = input
 *            -> (Cout,Cin,27).  ConvTranspose3d: G = its (coarse) input, X = grad of its raw output -> (Cin,Cout,27).
 *   planesweep_costvar_bwd: d cost (variance channels) -> d feats_cl[V][H][W][32] (+=, float atomics; caller zeroes). */
int mvsnerf_abn_bwd(const float* x, int64_t n_vox, int C, const float* weight, const float* scale, const float* shift,
                    const float* mean, const float* invstd, const float* g1, const float* g2,
                    float* gx, float* g_weight, float* g_bias, float* workspace, void* stream);
size_t mvsnerf_conv3d_wgrad_workspace_floats(int A, int B);
int mvsnerf_conv3d_wgrad(const float* g1, const float* g1_scale, const float* g1_shift,
                         const float* g2, const float* g2_scale, const float* g2_shift, int A,
                         const float* x1, const float* x1_scale, const float* x1_shift,
                         const float* x2, const float* x2_scale, const float* x2_shift, int B, int ldx,
                         int Do, int Ho, int Wo, int Di, int Hi, int Wi, int stride,
                         float* gw, float* workspace, void* stream);
/* The same weight gradients with bf16 operands (use_amp; csrc/wgrad_bf16.hip): G and X are rounded to bf16 when they are staged, products
 * accumulate in fp32 on v_mfma_f32_16x16x32_bf16, deterministic.  kz x k x k taps, padding k / 2 (kz / 2): (3, 3) with stride 1 | 2 for the 3-D
 * layers, (1, 3) stride 1, (1, 5) stride 2 and (1, 1) for FeatureNet's 2-D layers (D = the images, not strided).  A, B multiples of 4, <= 64.
 * workspace: mvsnerf_conv_wgrad_bf16_workspace_floats(A, B, kz, k); gw NULL leaves mvsnerf_conv_wgrad_bf16_parts(...) partial results at its
 * start for mvsnerf_partial_sum_multi (0 parts: shape not built). */

int mvsnerf_planesweep_costvar_bwd(const float* feats_cl, const float* proj, const float* depth, int V, int C, int H, int W, int D, int pad,
                                   const float* g_cost, int CP, int with_img, float* g_feats_cl, void* stream);
/* The same gradient with an ORDER-INDEPENDENT reduction: the source views' sums are accumulated in 64-bit fixed point (integer atomics commute;
 * the scale is derived on the device from max |g_cost| * max |feats|, a contribution is rounded to 2^-36 of the largest possible one), then
 * added to g_feats_cl.  Two runs - and N ranks against one - give bit-identical results; ~0.1 ms slower at config 3 (two extra passes).
 * workspace_zeroed: mvsnerf_planesweep_costvar_bwd_det_workspace_words(V, C, H, W) int64 words, 8-byte aligned, zeroed before every call. */
size_t mvsnerf_planesweep_costvar_bwd_det_workspace_words(int V, int C, int H, int W);
int mvsnerf_planesweep_costvar_bwd_det(const float* feats_cl, const float* proj, const float* depth, int V, int C, int H, int W, int D, int pad,
                                       const float* g_cost, int CP, int with_img, float* g_feats_cl, void* workspace_zeroed, void* stream);

/* ---- FeatureNet (models.py:688-722; ConvBnReLU :661-672): 2-D convolutions over N images, channel-last
 * act[n][y][x][C], same lazy-InPlaceABN convention as the 3-D blocks (statistics via mvsnerf_abn_stats / mvsnerf_abn_bwd
 * with n_vox = N*H*W).
 *   conv2d_pack_weights: packed[tap][ci][co] = w[ci*s_ci + co*s_co + tap], ksize*ksize taps (`flip` mirrors them).
 *            Conv2d (Cout,Cin,k,k): s_ci = k*k, s_co = Cin*k*k; data gradient: strides swapped (+ flip when stride 1).
 *   conv2d_fwd: k in {1,3,5}, padding k/2, stride 1|2, optional bias[Cout] -> raw out[N][Ho][Wo][Cout]; the input is
 *            leaky(x*scale+shift) when scale != NULL.  Also computes the data gradient of every stride-1 layer.
 *   conv2d_dgrad_k5s2: data gradient of a k5 s2 p2 layer: g[N][Ho][Wo][Cin] -> out[N][Hi][Wi][Cout].
 *   conv2d_wgrad: gW[a][b][tap] = sum_o G[o][a] * X[o*stride - k/2 + tap][b]  -> (A,B,k,k) = Conv2d's weight layout.
 *   channel_sum: out[c] = sum_i g[i][c]  (bias gradient of `toplayer`). */
int mvsnerf_conv2d_pack_weights(const float* w, int ci_real, int co_real, int cin_pad, int cout_pad,
                                int s_ci, int s_co, int ksize, int flip, float* packed, void* stream);
int mvsnerf_conv2d_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld,
                       int N, int H, int W, const float* wpacked, const float* bias, int Cout,
                       int ksize, int stride, float* out, void* stream);
/* The 16- / 32-output-channel layers (conv1.x, conv2.x, and the stride-1 data gradients) run on the fp32 matrix cores inside
 * mvsnerf_conv2d_fwd.  conv2d_fwd_stats is a layer's launch + the InPlaceABN partial sums of the raw output (stats_part[2][Cout][tiles] for
 * mvsnerf_abn_finalize): the matrix-core layers and the two full-resolution 8-channel layers conv0.0 / conv0.1 (VALU kernel, 16 x 16 pixel
 * tiles).  conv2d_mfma_tiles(...) = the number of partial-sum rows that launch leaves; 0 = this layer cannot -> MVSNERF_EUNSUPPORTED. */

size_t mvsnerf_conv2d_wgrad_workspace_floats(int A, int B, int ksize);
int mvsnerf_conv2d_wgrad(const float* g, int A, const float* x, const float* x_scale, const float* x_shift, int B, int ldx,
                         int N, int Ho, int Wo, int Hi, int Wi, int ksize, int stride,
                         float* gw, float* workspace, void* stream);
size_t mvsnerf_channel_sum_workspace_floats(int C);
int mvsnerf_channel_sum(const float* g, int64_t n, int C, float* out, float* workspace, void* stream);

/* ---------------------------------------------------------------- ray march (L1b) */

/* Ray generation: the arithmetic of build_rays / build_rays_test (utils.py:86-108, 148-297) downstream of the RNG
 * draws.  xs/ys[N]: pixel ids as floats drawn by the caller, or NULL for the row-major ids first_pixel + n of
 * build_rays_test.  K_*[3][3], c2w_tgt / w2c_ref [4][4] row-major, near_far_*[2]: all DEVICE pointers (read with
 * wave-uniform loads, so no host synchronisation is needed to launch).  t_rand[N][S]: stratified jitter or NULL.
 * W_img x H_img: the target view (pixel ids); W_ref x H_ref: the view whose pixel grid the NDC coordinates are normalised by
 * (`inv_scale` of utils.py:112-146).  The reference uses the target size for both (utils.py:252); pass 0 for that, or the
 * source-view size when the target grid differs from the sources (BASELINE config 5: 1008x756 rays over 960x640 sources).
 * Outputs: rays_pts[N][S][3], rays_dir[N][3], rays_ndc[N][S][3], z_vals[N][S], pix[2][N] (= (ys,xs), may be NULL). */
int mvsnerf_raygen_fwd(const float* xs, const float* ys, int64_t first_pixel, int W_img, int H_img, int W_ref, int H_ref,
                       const float* K_tgt, const float* c2w_tgt, const float* K_ref, const float* w2c_ref,
                       const float* near_far_tgt, const float* near_far_ref, int pad, int lindisp,
                       const float* t_rand, int64_t N, int S,
                       float* rays_pts, float* rays_dir, float* rays_ndc, float* z_vals, float* pix, void* stream);

/* build_rays of one training step in ONE launch (utils.py:148-241 downstream of the RNG draws; train_mvs_nerf_pl.py:119): the ray
 * generation above plus the per-pixel gathers - target colours (tgt_img [3][H_img][W_img] -> colors [N][3], utils.py:190-192), ground-truth
 * depth (depth_map [H_img][W_img] -> rays_depth [N], :194-196; both NULL when the batch carries no depth) - and the per-pixel depth
 * ranges: depth_mode 0 = near/far of the view; 1 = importanceSampling (near, far = depth -+ 0.1, :202-204); 2 = with_depth (the single
 * candidate of a ray is z_map[y][x], :199-200; S must be 1). */
int mvsnerf_raygen_train_fwd(const float* xs, const float* ys, int W_img, int H_img, int W_ref, int H_ref,
                             const float* K_tgt, const float* c2w_tgt, const float* K_ref, const float* w2c_ref,
                             const float* near_far_tgt, const float* near_far_ref, int pad, int lindisp,
                             const float* t_rand, int64_t N, int S,
                             const float* tgt_img, const float* depth_map, const float* z_map, int depth_mode,
                             float* rays_pts, float* rays_dir, float* rays_ndc, float* z_vals, float* pix, float* colors, float* rays_depth,
                             void* stream);

/* Trilinear lookup of the channel-last volume: replaces F.grid_sample 5-D in
 * utils.py:357-383 (index_point_feature) and models.py:941-950 (RefVolume.forward).
 * vol: C channels per voxel in the memory order `vol_layout` (MVSNERF_VOL_DHWC / MVSNERF_VOL_HWDC); ndc[P][3] = (x->W, y->H, z->D) in
 * [0,1]; zeros padding, align_corners=True.  out[p*out_stride + c], c < C  (out_stride lets it write the first 8 columns of input_feat).
 * C == 8 with 16-byte aligned rows takes the four-lanes-per-sample kernels; the same bits in either layout. */
int mvsnerf_volume_sample_fwd(const float* vol, int D, int H, int W, int C,
                              const float* ndc, int64_t P,
                              float* out, int out_stride, int vol_layout, void* stream);

/* Per-view colour lookup: replaces utils.py:300-332 (build_color_volume, img_feat=None) including
 * get_ndc_coordinate (utils.py:112-146) for each source view.
 * imgs[V][3][H][W] un-normalised; w2c[V][4][4]; K[V][3][3] (device); pts[P][3] world.
 * out[p*out_stride + v*Cv + {0,1,2,(3)}] = (r,g,b,(mask)), Cv = 3 + with_mask.
 * bilinear, *border* padding, align_corners=True; mask = strict -1<g<1 on both axes. */
int mvsnerf_color_sample_fwd(const float* imgs, int V, int H, int W,
                             const float* w2c, const float* K,
                             const float* pts, int64_t P, int with_mask,
                             float* out, int out_stride, void* stream);

/* build_color_volume with per-view feature maps (utils.py:300-332, `img_feat is not None`; gen_pts_feats renderer.py:124-136 passes it
 * through): per sample and view [r, g, b (border padding) | Cf channels of img_feat[v] (zeros padding, same normalised grid, the map's
 * own Hf x Wf) | mask].  imgs [V][3][H][W], img_feat [V][Cf][Hf][Wf] (NCHW, the reference's layout); out rows of out_stride floats,
 * view v at columns v*(3+Cf+mask). */
int mvsnerf_color_feat_sample_fwd(const float* imgs, int V, int H, int W, const float* img_feat, int Cf, int Hf, int Wf,
                                  const float* w2c, const float* K, const float* pts, int64_t P, int with_mask,
                                  float* out, int out_stride, void* stream);

/* View-direction feature: renderer.py:142-147 + gen_dir_feature renderer.py:111-122.
 * dirs_out[n] = (rays_dir[n]/|rays_dir[n]|) @ w2c_ref[:3,:3]^T ; w2c_ref may be NULL (no rotation);
 * normalize == 0 skips the division (gen_dir_feature called on already-unit directions). */
int mvsnerf_dir_feature_fwd(const float* rays_dir, const float* w2c_ref, int64_t N, int normalize,
                            float* dirs_out, void* stream);

/* gen_pts_feats + gen_dir_feature (renderer.py:111-136) in one launch: the three lookups above for N rays x S samples.
 * imgs_nhwc4[V][IH][IW][4]: the un-normalised source images re-laid channel-last (rgb + one pad float; mvsnerf_nchw_to_nhwc
 * with Cpad = 4) so that a bilinear tap is one 16-byte load.  feat[p*feat_stride + {0..7 | 8+4v..8+4v+3}] = volume features |
 * (r,g,b,mask) of view v; dirs_out[N][3] (may be NULL) = normalised rays_dir rotated into view 0's frame (w2c[0]).
 * Results are bit-identical to volume_sample_fwd / color_sample_fwd(with_mask=1) / dir_feature_fwd(normalize=1). */
int mvsnerf_gather_fwd(const float* vol, int D, int H, int W, const float* imgs_nhwc4, int V, int IH, int IW,
                       const float* w2c, const float* K, const float* pts, const float* ndc, int64_t N, int S,
                       const float* rays_dir, float* feat, int feat_stride, float* dirs_out, int vol_layout, void* stream);

/* Stand-alone positional encoding, Embedder.embed (models.py:47-51): x[P][d] ->
 * out[P][d*(1+2L)] = [x | sin(x_c*2^f), f-major | cos(...)].  The MLP kernel below embeds internally;
 * this exists for callers that use `embed_fn` on its own. */
int mvsnerf_posenc_fwd(const float* x, int64_t P, int d, int L, float* out, void* stream);

/* Renderer_ours MLP (models.py:145-222) with the Embedder (models.py:17-51, multires=10) fused in.
 * Specialised for the shipped architecture: netdepth 6, netwidth 128, skips=[4], 63-dim embedding,
 * 3-dim raw view direction, feat_dim F = 8+4V (even, <= 40).
 *
 * mvsnerf_mlp_packed_floats(F) -> number of floats of the packed-weight buffer.
 * mvsnerf_mlp_pack: re-lays the 11 nn.Linear weight/bias tensors (row-major [out][in], the
 *   checkpoint's layout) into the MFMA-fragment order the kernel streams through LDS.
 *   Order of `w`/`b`: pts_linears.0..5, pts_bias, feature_linear, alpha_linear, views_linears.0, rgb_linear.
 * mvsnerf_mlp_fwd: raw[p] = (r,g,b,sigma).  ndc[p*ndc_stride + {0,1,2}] (embedded inside),
 *   feat[p*feat_stride + k], dirs[n*dirs_stride + {0,1,2}] per ray (P = N*S, point p belongs to ray
 *   p / S).  The strides let MVSNeRF.forward(x) (models.py:565) run on the reference's 86-wide rows
 *   in place: ndc = x, feat = x+63, dirs = x+63+F, all with stride 86, S = 1.
 *   alpha_only != 0 follows forward_alpha (models.py:176-191): dirs may be NULL, raw is [P][1].
 */
size_t mvsnerf_mlp_packed_floats(int F);
int mvsnerf_mlp_pack(const float* const w[11], const float* const b[11], int F,
                     float* packed, void* stream);
int mvsnerf_mlp_fwd(const float* packed, int F,
                    const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                    const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only,
                    float* raw, void* stream);

/* bf16-MFMA variant of the MLP forward (BASELINE configs 3/4: "bf16", "MFMA-bf16 MLP"): weights and layer inputs are
 * rounded to bf16, products accumulate in fp32; biases, modulation, ReLU, positional encoding and the heads stay fp32.
 * Opt-in (results differ from the fp32 path at the 1e-2 level); `packed_f32` is the buffer of mvsnerf_mlp_pack (its
 * bias/head vectors are reused), `packed_bf16` holds mvsnerf_mlp_packed_bf16_elems(F) 16-bit elements. */
size_t mvsnerf_mlp_packed_bf16_elems(int F);
int mvsnerf_mlp_pack_bf16(const float* const w[11], int F, void* packed_bf16, void* stream);
int mvsnerf_mlp_fwd_bf16(const void* packed_bf16, const float* packed_f32, int F, const float* ndc, int ndc_stride,
                         const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                         int64_t N, int S, int alpha_only, float* raw, void* stream);

/* bf16 TRAINING (the reference's AMP switch, train_mvs_nerf_pl.py:317-318 `precision=16 if args.use_amp`; BASELINE config 3):
 *   mlp_fwd_bf16_train  = mlp_fwd_bf16 + the activation store of mvsnerf_mlp_fwd_train as BF16 slots (same [tile][slot][lane] order, two
 *                       bytes per element: `saved` needs mvsnerf_mlp_saved_floats(N*S) / 2 floats) - what the bf16 backward reads
 *   mlp_pack_bwd_bf16   W^T fragments for v_mfma_f32_32x32x16_bf16 (mvsnerf_mlp_packed_bwd_bf16_elems() 16-bit elements)
 *   mlp_bwd_bf16        mvsnerf_mlp_bwd with every GEMM of the backward pass (data gradient W^T products, weight-gradient point
 *                       contractions) on the bf16 matrix cores: operands rounded to bf16, fp32 accumulate; activation derivatives,
 *                       bias gradients and reductions fp32; gradients are returned in fp32 (fp32 master weights, fp32 all-reduce).
 *                       `saved` must come from mvsnerf_mlp_fwd_bf16_train (bf16 slots); `gslots` holds bf16 slots as well
 *                       (mvsnerf_mlp_gradslot_floats(N*S) / 2 floats).  As with autocast, what is kept for the backward pass is 16-bit. */
int mvsnerf_mlp_fwd_bf16_train(const void* packed_bf16, const float* packed_f32, int F, const float* ndc, int ndc_stride,
                               const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                               int64_t N, int S, float* raw, float* saved, void* stream);
size_t mvsnerf_mlp_packed_bwd_bf16_elems(void);
int mvsnerf_mlp_pack_bwd_bf16(const float* const w[11], int F, void* packed_bwd_bf16, void* stream);
int mvsnerf_mlp_bwd_bf16(const float* packed_fwd, const void* packed_bwd_bf16, int F,
                         const float* raw, const float* d_raw, const float* saved, int64_t N, int S,
                         float* gslots, float* d_feat, int n_feat_out, float* const gw[11], float* const gb[11],
                         const int* maps, float* workspace, void* stream);

/* ---- training path of the MLP (autograd of models.py:194-222) ----
 * mvsnerf_mlp_fwd_train = mvsnerf_mlp_fwd + an activation store `saved` (mvsnerf_mlp_saved_floats(N*S) floats).
 * mvsnerf_mlp_pack_bwd re-lays W^T fragments for the gradient chain (mvsnerf_mlp_packed_bwd_floats() floats).
 * mvsnerf_mlp_bwd: given d_raw[P][4] (grad wrt (r,g,b,sigma)) writes
 *     d_feat[P][n_feat_out]  grad wrt the first n_feat_out feature columns: 8 = the trilinear volume features; F = every
 *                     input feature (the colour volume of --use_color_volume fine-tuning is a parameter too)
 *     gw[i], gb[i]    grads of the 11 nn.Linear weight/bias tensors (order of mvsnerf_mlp_pack), OVERWRITTEN
 *   gslots: scratch of mvsnerf_mlp_gradslot_floats(N*S) floats; workspace: mvsnerf_mlp_bwd_workspace_floats();
 *   maps: device int table built by the host side (fragment row -> nn.Linear row/column, mvsnerf_amd/ops.py).
 *   No gradient is produced for ndc / view directions / colour features (the reference's losses never need them:
 *   rays and source images carry no parameters). */
size_t mvsnerf_mlp_saved_floats(int64_t n_points);
size_t mvsnerf_mlp_gradslot_floats(int64_t n_points);
size_t mvsnerf_mlp_packed_bwd_floats(void);
size_t mvsnerf_mlp_bwd_workspace_floats(void);
int mvsnerf_mlp_pack_bwd(const float* const w[11], int F, float* packed_bwd, void* stream);
int mvsnerf_mlp_fwd_train(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                          const float* dirs, int dirs_stride, int64_t N, int S, float* raw, float* saved, void* stream);
int mvsnerf_mlp_bwd(const float* packed_fwd, const float* packed_bwd, int F,
                    const float* raw, const float* d_raw, const float* saved, int64_t N, int S,
                    float* gslots, float* d_feat, int n_feat_out, float* const gw[11], float* const gb[11],
                    const int* maps, float* workspace, void* stream);

/* Backward of the compositing w.r.t. raw (autograd of renderer.py:18-26,65-92).  Upstream grads g_rgb[N][3],
 * g_depth[N], g_acc[N], g_weights[N][S], g_alpha[N][S] (any may be NULL = zero) -> d_raw[N][S][4]. */
int mvsnerf_composite_bwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd,
                          const float* g_rgb, const float* g_depth, const float* g_acc,
                          const float* g_weights, const float* g_alpha, float* d_raw, void* stream);

/* Backward of the trilinear lookup w.r.t. the volume: gvol[D][H][W][C] += scatter of g[p*g_stride + c], C a multiple of 4
 * (8: the neural volume; 8+4V: the colour volume of --use_color_volume, train_mvs_nerf_finetuning_pl.py:72-82)
 * (float atomics; gvol must be zero-initialised by the caller).  Needed by RefVolume fine-tuning
 * (train_mvs_nerf_finetuning_pl.py:54) and, through the encoder, by generalizable training. */
int mvsnerf_volume_sample_bwd(int D, int H, int W, int C, const float* ndc, int64_t P,
                              const float* g, int g_stride, float* gvol, void* stream);
/* The same gradient with an ORDER-INDEPENDENT reduction (ABI v12; like mvsnerf_planesweep_costvar_bwd_det): contributions are accumulated in 64-bit fixed point
 * (integer atomics commute; scale from max |g| found on the device: a contribution resolves 2^-40 of the largest one), then added to gvol.  Two runs - and N ranks
 * against one - give bit-identical volume gradients; two extra passes (max, finish) and 8 bytes of zeroed workspace per volume element.
 * workspace_zeroed: mvsnerf_volume_sample_bwd_det_workspace_words(D, H, W, C) int64 words, 8-byte aligned, zeroed before every call. */
size_t mvsnerf_volume_sample_bwd_det_workspace_words(int D, int H, int W, int C);
int mvsnerf_volume_sample_bwd_det(int D, int H, int W, int C, const float* ndc, int64_t P, const float* g, int g_stride, float* gvol,
                                  void* workspace_zeroed, void* stream);

/* Alpha compositing: raw2alpha + raw2outputs (renderer.py:18-26, 65-92).
 * raw[N][S][4], z[N][S] -> rgb_map[N][3], disp[N], acc[N], weights[N][S], depth[N], alpha[N][S].
 * Any output pointer may be NULL (skipped). */
int mvsnerf_composite_fwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd,
                          float* rgb_map, float* disp, float* acc, float* weights,
                          float* depth, float* alpha, void* stream);

/* One-call ray march = rendering() (renderer.py:138-165): dir feature -> volume + colour lookup ->
 * MLP -> compositing, all enqueued on `stream`.  input_feat[N][S][F] and raw[N][S][4] are outputs too
 * (rendering returns input_feat; raw carries sigma). */
typedef struct {
    const float* vol; int D, H, W;          /* channel-last 8-channel volume */
    const float* imgs; int V, IH, IW;       /* [V][3][IH][IW] un-normalised source images */
    const float* w2c;                       /* [V][4][4], view 0 = reference view */
    const float* K;                         /* [V][3][3] */
    const float* packed_mlp;                /* from mvsnerf_mlp_pack, F = 8+4V */
    const float* rays_pts;                  /* [N][S][3] world */
    const float* rays_ndc;                  /* [N][S][3] reference-view NDC */
    const float* z_vals;                    /* [N][S] */
    const float* rays_dir;                  /* [N][3] un-normalised */
    int64_t N; int S; int white_bkgd;
    float* dirs_tmp;                        /* [N][3] workspace */
    float* input_feat;                      /* [N][S][F] out */
    float* raw;                             /* [N][S][4] out */
    float* rgb_map; float* disp; float* acc; float* weights; float* depth; float* alpha;  /* outs, may be NULL */
    const void* packed_mlp_bf16;            /* NULL: fp32-MFMA MLP (default); else the bf16-MFMA variant is used (ABI v2) */
    const float* imgs_nhwc4;                /* NULL: three gather launches from `imgs` (NCHW); else [V][IH][IW][4] copies of the
                                               same images and ONE fused gather launch (mvsnerf_gather_fwd) (ABI v3) */
    const void* packed_mlp_split; int n_split;   /* NULL/0, or mvsnerf_mlp_pack_split output: split-bf16 MLP (ABI v5) */
    int* guard;                             /* NULL, or a guard word pair with n_split = MVSNERF_SPLIT_FP16: the guarded sequence below (ABI v10) */
    int vol_layout;                         /* MVSNERF_VOL_DHWC (0) / MVSNERF_VOL_HWDC: memory order of `vol` (ABI v10) */
} mvsnerf_raymarch_args;
int mvsnerf_raymarch_fwd(const mvsnerf_raymarch_args* a, void* stream);
/* K independent ray batches (the K iterations of a render loop over renderer.py:138-165) enqueued by ONE host call: a[0..K) are complete argument
 * blocks (they may share everything but the ray tensors and the outputs).  The first failing block's code is returned; blocks before it have
 * been enqueued.  Exists because one batch is ~0.1 ms of GPU work: a caller that crosses the FFI once per batch is paced by its own host code (ABI v10). */
int mvsnerf_raymarch_fwd_batched(const mvsnerf_raymarch_args* a, int K, void* stream);

/* Split-bf16 MLP ("bf16x3" n_split = 2, "bf16x6" n_split = 3; n_split = 1 is plain bf16 on the same kernel): every fp32
 * operand is the sum of n_split bf16 pieces and a product is accumulated in fp32 from the piece products of combined order
 * < n_split, so n_split = 3 reproduces fp32 products to ~2^-24 on the bf16 matrix cores at 6/16 of the fp32-MFMA time
 * (the technique BLAS libraries ship as "BF16x9/x6 FP32 emulation").  Same arguments and results layout as mvsnerf_mlp_fwd;
 * `packed_f32` (mvsnerf_mlp_pack) supplies the fp32 bias / head vectors.  Opt-in: the default path is the fp32 MFMA kernel.
 *
 * n_split = MVSNERF_SPLIT_FP16 ("fp16x3", mlp_f16x3.hip): TWO FP16 pieces per operand, both rounded to nearest, and the three piece products
 * a0*w0 + a0*w1 + a1*w0 on v_mfma_f32_32x32x16_f16.  fp16 carries 11 significant bits, so two pieces hold 22 and what is dropped is
 * <= 2^-22 of a product - fp32-grade like n_split = 3, at half the matrix-core work.  fp16's five exponent bits are handled inside (ABI v12):
 * every K-block of the weights is stored with an exact power-of-two scale chosen at pack time, every point's activations are re-scaled by a power
 * of two whenever their largest magnitude leaves [2^-3, 2^12]; the scales are carried through the layers and divided out in the two heads.  Only a
 * non-finite weight or value is beyond the kernel (reported through the guard of the guarded sequences below). */
#define MVSNERF_SPLIT_FP16 18
size_t mvsnerf_mlp_packed_split_elems(int F, int n_split);
int mvsnerf_mlp_pack_split(const float* const w[11], int F, int n_split, void* packed_split, void* stream);
int mvsnerf_mlp_fwd_split(const void* packed_split, const float* packed_f32, int F, int n_split, const float* ndc, int ndc_stride,
                          const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                          int64_t N, int S, int alpha_only, float* raw, void* stream);

/* ---- the optimizer step (train_mvs_nerf_pl.py:84-88: torch.optim.Adam, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) ----
 * One Adam update of n fp32 tensors in ONE launch per 84 tensors (host arrays of device pointers; contiguous tensors; tensors whose size is a multiple of
 * four must be 16-byte aligned):  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g^2;  p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)  with
 * step_size = lr / (1 - beta1^step) and bc2_sqrt = sqrt(1 - beta2^step) computed by the caller.  torch's fused Adam needs three ~25 us launches for
 * the 78 tensors of the generalizable step; mvsnerf_amd.optim.Adam (a torch.optim.Adam subclass, same state_dict) calls this instead. */
int mvsnerf_adam_step_multi(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                            double step_size, double beta1, double beta2, double eps, double bc2_sqrt, void* stream);

/* ---- Guarded 16-bit sequences (ABI v10) ----
 * The two-piece fp16 kernels ("fp16x3": the MLP above, conv0 of CostRegNet below) give fp32-grade results at 2-2.5x the fp32-MFMA rate, but an
 * fp16 piece cannot hold more than 65504: a value outside that range saturates.  A GUARDED sequence makes them safe to use by default without a
 * host synchronisation: `guard` is a device buffer of 4 ints owned by the caller (zero it once); the 16-bit kernels set guard[0] when an operand,
 * a stored value or (at pack time, through a status word behind the packed weights) a weight left fp16's range; the fp32 kernels of the same
 * stage are enqueued right behind them PREDICATED on guard[0] (they leave at once when it is 0, and overwrite the results when it is set); the
 * last kernel of the sequence counts the event in guard[1] and re-arms guard[0] = 0.  The caller reads results that are the fp32 kernels'
 * whenever the 16-bit ones were out of range, and may read guard[1] (number of sequences that fell back) whenever it synchronises anyway.
 * What trips the guard: the conv kernels report RANGE - an operand, weight or layer output beyond what an fp16 piece holds after the kernel's scaling.  The
 * MLP kernel (ABI v12) manages exponents itself (see MVSNERF_SPLIT_FP16 above) and reports only a non-finite weight (status tail of the pack) or value:
 * a network with hidden activations 1e-4 or 1e+5 times the shipped one's stays on the fp16 kernel and within 2e-6 of the oracle
 * (through ABI v11 both were re-run in fp32: Kaiming-initialised networks paid for both kernels on every batch).
 * ONE GUARD BUFFER PER STREAM: guard[0] is armed, read and re-armed in stream order only (guard[1] += 1 is a plain store of the last kernel), so
 * sequences enqueued on different streams - or by different host threads - must be given different buffers; sharing one lets stream B re-arm the
 * word between stream A's 16-bit kernel setting it and A's predicated fp32 kernel reading it.  mvsnerf_amd.ops.guard_words() keeps one buffer per
 * (device, stream) for that reason.
 *   - mvsnerf_raymarch_fwd / mvsnerf_render_pixels_fwd: args.guard with n_split = MVSNERF_SPLIT_FP16 (gather -> fp16x3 MLP -> predicated fp32-MFMA
 *     MLP -> compositing, which also re-arms the guard);
 *   - mvsnerf_mlp_fwd_guarded: the stand-alone network query (run_network_mvs, renderer.py:42-63; alpha_only = forward_alpha);
 *   - mvsnerf_conv3d_f16x3_guarded_fwd (scene encode, conv1 / conv2; ABI v11);
 *   - mvsnerf_sweep_conv0_guarded_fwd (scene encode): two-piece plane sweep -> fp16x3 conv0 -> predicated {fp32 plane sweep in channel blocks,
 *     fp32-MFMA conv0, InPlaceABN partial sums} -> re-arm.  `cost32` (the fp32 hand-off, (CP/4) * D*Hp*Wp * 4 floats) is scratch that is only
 *     written when the guard trips.  stats_part: mvsnerf_conv0_bf16_tiles(D, Hp, Wp) slots x 16 floats in either case. */
int mvsnerf_mlp_fwd_guarded(const void* packed_fp16, const float* packed_f32, int F, const float* ndc, int ndc_stride,
                            const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                            int64_t N, int S, int alpha_only, float* raw, int* guard, void* stream);

/* ---- importance sampling of the fine-tuning option --use_density_volume ----
 * sample_pdf (data/ray_utils.py:96-139): bins[N][n_bins] ascending, weights[N][n_bins-1], u[N][n_importance] uniform draws
 *   supplied by the caller (the reference draws them with torch.rand, :109) -> samples[N][n_importance].
 * ray_marcher_fine (data/ray_utils.py:199-224): density[D][H][W]; ndc[N][S][3] reference-view NDC of the coarse samples;
 *   z_vals[N][S] ascending; u[N][n_importance] -> z_out[N][S+n_importance] = sort(cat(sample_pdf(mid points, weights[1:-1]),
 *   z_vals)).  The density lookup applies the reference's double [-1,1] mapping (:209-210) as is.  S, n_importance <= 512.
 * ray_points: pts = o + d*z (rays_o [N][3], or one shared origin when o_is_per_ray == 0) and, when rays_ndc != NULL, their
 *   NDC coordinates in the reference view (get_ndc_coordinate, utils.py:112-146; cameras are DEVICE pointers). */
int mvsnerf_sample_pdf_fwd(const float* bins, const float* weights, const float* u, int64_t N, int n_bins, int n_importance,
                           float* samples, void* stream);
int mvsnerf_ray_marcher_fine_fwd(const float* density, int D, int H, int W, const float* ndc, const float* z_vals, const float* u,
                                 int64_t N, int S, int n_importance, float* z_out, void* stream);
int mvsnerf_ray_points_fwd(const float* rays_o, int o_is_per_ray, const float* rays_d, const float* z_vals,
                           const float* w2c_ref, const float* K_ref, const float* near_far_ref, int W_ref, int H_ref, int pad, int lindisp,
                           int64_t N, int S, float* rays_pts, float* rays_ndc, void* stream);

/* Pixel-range render of one target view = the chunk loop of validation_step (train_mvs_nerf_pl.py:198-208:
 * build_rays_test utils.py:243-297 -> rendering renderer.py:138-165 per chunk) enqueued from one host call.
 * Renders row-major pixels [first_pixel, first_pixel + n_pixels) of a W_img x H_img target view in sub-batches of
 * `batch_rays` rays (free parameter: rays are independent, results do not depend on it) and writes rgb[n_pixels][3] and,
 * when non-NULL, depth/acc/disp[n_pixels].  Cameras and near/far pairs are DEVICE pointers (as for mvsnerf_raygen_fwd);
 * workspace: mvsnerf_render_workspace_floats(batch_rays, S, V) floats, 16-byte aligned. */
typedef struct {
    const float* vol; int D, H, W;          /* [D][H][W][8] */
    const float* imgs_nhwc4; int V, IH, IW; /* [V][IH][IW][4] un-normalised source images, channel-last */
    const float* w2c;                       /* [V][4][4] source views (view 0 = reference) */
    const float* K;                         /* [V][3][3] */
    const float* packed_mlp;                /* mvsnerf_mlp_pack output for F = 8+4V */
    const void* packed_mlp_bf16;            /* NULL or mvsnerf_mlp_pack_bf16 output */
    const float* K_tgt; const float* c2w_tgt;       /* target camera [3][3], [4][4] */
    const float* K_ref; const float* w2c_ref;       /* camera the NDC coordinates refer to */
    const float* near_far_tgt; const float* near_far_ref;   /* [2] each */
    int W_img, H_img, pad, lindisp;
    int W_ref, H_ref;                       /* 0: same as the target (the reference's assumption); see mvsnerf_raygen_fwd */
    int64_t first_pixel, n_pixels;
    int S, white_bkgd, batch_rays;
    float* workspace; size_t workspace_floats;
    float* rgb; float* depth; float* acc; float* disp;      /* rgb required, others may be NULL */
    const void* packed_mlp_split; int n_split;              /* NULL/0, or mvsnerf_mlp_pack_split output (ABI v5) */
    int* guard;                                             /* NULL, or guard words with n_split = MVSNERF_SPLIT_FP16: every sub-batch is a guarded sequence (ABI v10) */
    int vol_layout;                                         /* memory order of `vol` (ABI v10) */
} mvsnerf_render_args;
size_t mvsnerf_render_workspace_floats(int batch_rays, int S, int V);
int mvsnerf_render_pixels_fwd(const mvsnerf_render_args* a, void* stream);

/* ---- the differentiable ray march as TWO host calls (raymarch_fused_{fwd,bwd}) ----
 * rendering() (renderer.py:138-165) inside train_mvs_nerf_pl.py:123 / train_mvs_nerf_finetuning_pl.py:166:
 *   raymarch_train_fwd  gather (or, for an (8+4V)-channel colour volume, one lookup) -> MLP training forward with the activation
 *                       store -> compositing; everything the backward call needs stays in caller-owned buffers
 *   raymarch_bwd        compositing backward -> MLP backward (data + weight gradients) -> trilinear scatter into the volume gradient
 * bf16 != 0 selects the bf16-MFMA MLP kernels (packed_mlp_bf16 / packed_bwd = the bf16 packs), else fp32.
 * Gradient inputs g_* may be NULL (= zero).  gvol == NULL skips the volume gradient (frozen volume); it must be zero-initialised. */
typedef struct {
    const float* vol; int D, H, W, C;       /* C = 8 or 8+4V (--use_color_volume) channels per voxel, memory order vol_layout */
    const float* imgs_nhwc4; int V, IH, IW; /* [V][IH][IW][4]; unused when C == 8+4V */
    const float* w2c; const float* K;       /* [V][4][4], [V][3][3] */
    const float* packed_mlp; const void* packed_mlp_bf16; int bf16;
    const float* rays_pts; const float* rays_ndc; const float* z_vals; const float* rays_dir;
    int64_t N; int S; int white_bkgd;
    float* dirs_tmp;                        /* [N][3] */
    float* input_feat;                      /* [N][S][8+4V] (output of rendering() too) */
    float* raw;                             /* [N][S][4] */
    float* saved;                           /* mvsnerf_mlp_saved_floats(N*S) floats; half of that when bf16 (two-byte slots) */
    float* rgb_map; float* disp; float* acc; float* weights; float* depth; float* alpha;
    int vol_layout;                         /* memory order of `vol` (ABI v10); the gradient volume of mvsnerf_raymarch_bwd is DHWC in either case */
} mvsnerf_raymarch_train_args;
int mvsnerf_raymarch_train_fwd(const mvsnerf_raymarch_train_args* a, void* stream);

typedef struct {
    const float* packed_mlp; const void* packed_bwd; int bf16; int F;
    const float* raw; const float* saved; const float* z_vals; const float* rays_ndc;
    int64_t N; int S; int white_bkgd;
    const float* g_rgb; const float* g_depth; const float* g_weights; const float* g_alpha;
    float* d_raw;                           /* scratch [N][S][4] */
    float* gslots;                          /* scratch mvsnerf_mlp_gradslot_floats(N*S) floats; half of that when bf16 */
    float* d_feat; int n_feat_out;          /* scratch [N*S][n_feat_out]; 8, or F for a colour volume */
    float* const* gw; float* const* gb;     /* 11 weight / bias gradient tensors (nn.Linear layout) */
    const int* maps; float* workspace;      /* as mvsnerf_mlp_bwd */
    float* gvol; int D, H, W, C;            /* NULL or [D][H][W][C] zero-initialised, C == n_feat_out */
} mvsnerf_raymarch_bwd_args;
int mvsnerf_raymarch_bwd(const mvsnerf_raymarch_bwd_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
