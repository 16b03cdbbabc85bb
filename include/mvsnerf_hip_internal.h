/* mvsnerf_hip_internal.h - the INTERNAL tier of libmvsnerf_hip.so's exports (see include/mvsnerf_hip.h for the stable one and the conventions).
 *
 * What mvsnerf_amd/encoder.py, ops.py and train.py drive their tuned layer loop with: one entry per kernel family and layout (matrix-core
 * convolutions per shape, blocked / bf16 / two-piece-fp16 cost volumes, InPlaceABN partial-sum plumbing, multi-job weight packing and
 * reduction, the guarded conv sequences of the scene encode, a census probe).  Exported because the layer loop lives in Python; NOT part of the
 * interface a reference-side maintainer binds, and free to change with the kernels (no ABI bump).  The reference call sites are the same as
 * those of the stable entries they specialise (CostRegNet models.py:725-769, FeatureNet models.py:688-722, plane sweep models.py:839-893).
 */
#ifndef MVSNERF_HIP_INTERNAL_H
#define MVSNERF_HIP_INTERNAL_H

#include "mvsnerf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same sweep writing the cost volume in channel blocks of FOUR: cost_blocked[CP/4][D*Hp*Wp][4] (block b holds channels 4b..4b+3
 * of every voxel).  This is the layout the matrix-core conv0 (mvsnerf_conv3d_c8_blocked_fwd) stages from by LDS-DMA: it multiplies four
 * input channels at a time, and with the channel-last layout every such pass touched all of a voxel's 176-byte row again
 * (measured: 5.4x the algorithmic HBM reads); in blocks, 64 consecutive voxels of a pass are 1 KB of contiguous, fully used bytes. */
int mvsnerf_planesweep_costvar_blocked_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                           int V, int C, int H, int W, int D, int pad, float* cost_blocked, int CP, float* masks,
                                           int with_img, void* stream);
/* bf16 encoder (the reference's AMP switch, train_mvs_nerf_pl.py:317-318 `precision=16 if args.use_amp`; BASELINE config 3 "bf16"):
 * conv0 of CostRegNet (models.py:756, 74.5 % of the encoder's FLOPs) on v_mfma_f32_16x16x32_bf16 - operands rounded to bf16, fp32
 * accumulation, fp32 master weights / statistics / gradients.
 *   planesweep_costvar_bf16_fwd  the plane sweep above storing the cost volume as bf16 in channel blocks of sixteen:
 *                                cost16[ceil(CP/16)][D*Hp*Wp][16] (channels >= CP zero); the sweep's arithmetic stays fp32
 *   conv0_bf16_pack              nn.Conv3d weight w[8][Cin][3][3][3] -> the kernel's B fragments (conv0_bf16_packed_elems(Cin) bf16 values)
 *   conv0_bf16_fwd               out[D][H][W][8] fp32 (raw, before InPlaceABN); stats_part: NULL or conv0_bf16_tiles(D,H,W) * 16 floats of
 *                                per-tile sums / sums of squares for mvsnerf_abn_finalize */
int mvsnerf_planesweep_costvar_bf16_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                        int V, int C, int H, int W, int D, int pad, void* cost16, int CP, float* masks,
                                        int with_img, void* stream);
size_t mvsnerf_conv0_bf16_packed_elems(int Cin);
int mvsnerf_conv0_bf16_pack(const float* w, int Cin, void* packed, void* stream);
int mvsnerf_conv0_bf16_tiles(int D, int H, int W);
int mvsnerf_conv0_bf16_fwd(const void* x16, int Cin, int D, int H, int W, const void* packed, float* out, float* stats_part, void* stream);
/* conv0 with fp32-GRADE results from the fp16 matrix cores (csrc/conv_f16x3.hip; opt-in `encoder.encoder_precision("fp16x3")`, inference): every
 * operand as two fp16 pieces, x*w ~= x0*w0 + x0*w1 + x1*w0 on v_mfma_f32_16x16x32_f16 (dropped: <= 2^-22 of a product), fp32 accumulation.
 *   planesweep_costvar_f16x2_fwd  the plane sweep (fp32 arithmetic) storing fp16(x/16) and fp16(x/16 - hi): cost16x2[2][ceil(CP/16)][D*Hp*Wp][16],
 *                                 hi plane then lo plane (the reference operation it replaces: models.py:839-893, as mvsnerf_planesweep_costvar_fwd)
 *   conv0_f16x3_pack / _fwd       w[8][Cin][3][3][3] -> fp16 pieces of 16 w; out / stats_part / tiles as mvsnerf_conv0_bf16_fwd (models.py:756) */
int mvsnerf_planesweep_costvar_f16x2_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                         int V, int C, int H, int W, int D, int pad, void* cost16x2, int CP, float* masks,
                                         int with_img, void* stream);
size_t mvsnerf_conv0_f16x3_packed_elems(int Cin);
int mvsnerf_conv0_f16x3_pack(const float* w, int Cin, void* packed, void* stream);
int mvsnerf_conv0_f16x3_fwd(const void* x16x2, int Cin, int D, int H, int W, const void* packed, float* out, float* stats_part, void* stream);
/*   conv0_bf16_dgrad_pack / conv0_bf16_dgrad   data gradient w.r.t. the n_ci (16 or 32) input channels starting at c_first (the plane sweep's
 *                                backward needs the 32 variance channels only): g = gradient of conv0's raw output, fp32 [D][H][W][8], rounded
 *                                to bf16 on the way into the matrix cores; gx[D][H][W][n_ci] fp32 */
size_t mvsnerf_conv0_bf16_dgrad_packed_elems(int n_ci);
int mvsnerf_conv0_bf16_dgrad_pack(const float* w, int Cin, int c_first, int n_ci, void* packed, void* stream);
int mvsnerf_conv0_bf16_dgrad(const float* g, int D, int H, int W, const void* packed, int n_ci, float* gx, void* stream);
/*   conv0_bf16_wgrad             gw[8][Cin][3][3][3] from the bf16 cost volume and g (fp32, rounded to bf16 while staged): voxels are the
 *                                reduction dimension, both operands are transposed by the LDS read (ds_read_b64_tr_b16); deterministic.
 *                                workspace: mvsnerf_conv3d_wgrad_workspace_floats(8, Cin) floats; gw NULL leaves conv0_bf16_wgrad_parts(D,H,W)
 *                                partial results (rows of 8*Cin*27 floats) at its start for mvsnerf_partial_sum_multi */
int mvsnerf_conv0_bf16_wgrad_parts(int D, int H, int W);
int mvsnerf_conv0_bf16_wgrad(const void* x16, int Cin, int D, int H, int W, const float* g, float* gw, float* workspace, void* stream);
/* conv2 of CostRegNet (16 -> 16, stride 1; models.py:736 ConvBnReLU3D -> InPlaceABN) with the InPlaceABN partial sums of its raw output from the
 * same launch: stats_part[2][16][mvsnerf_conv3d_tiled_tiles(D, H, W)] floats for mvsnerf_abn_finalize.  One lazily-activated source; other
 * shapes return MVSNERF_EUNSUPPORTED. */
int mvsnerf_conv3d_tiled_tiles(int D, int H, int W);
int mvsnerf_conv3d_fwd_stats(const float* x1, const float* scale1, const float* shift1, int Cin, int cin_ld, int D, int H, int W,
                             const float* wpacked, int Cout, int stride, float* out, float* stats_part, void* stream);
/* conv0 of CostRegNet (models.py:756; k3, stride 1, Cout = 8, raw input) on v_mfma_f32_4x4x1_16B_f32, input in channel
 * blocks of four (see mvsnerf_planesweep_costvar_blocked_fwd), Cin = 4*ceil((32+3V)/4) channels of which the first Cin_real = 32+3V exist
 * (products with the zero padding are skipped).  Weights: wq[ci/4][tap][co][4] = mvsnerf_conv3d_pack_weights_c8 of the
 * packed[tap][ci][8] layout of mvsnerf_conv3d_pack_weights (27*Cin*8 floats either way). */
int mvsnerf_conv3d_pack_weights_c8(const float* wpacked, int Cin, float* wq, void* stream);
int mvsnerf_conv3d_c8_blocked_fwd(const float* x_blocked, int Cin, int Cin_real, int D, int H, int W, const float* wq, float* out, void* stream);
/* Weight gradient of that convolution from the same blocked input (the training path keeps the cost volume blocked; autograd of
 * models.py:756 through nn.Conv3d): gw[co][ci][tap] = sum_o g[o][co] * x[o + tap - 1][ci] on v_mfma_f32_4x4x1 with the voxels as the k
 * dimension.  g: gradient of the raw conv0 output, channel-last [D][H][W][8]; gw: (8, Cin_real, 3,3,3) floats; workspace:
 * mvsnerf_conv3d_wgrad_workspace_floats(8, Cin_real).  Deterministic. */
int mvsnerf_conv3d_c8_blocked_wgrad(const float* x_blocked, int Cin, int Cin_real, int D, int H, int W, const float* g, float* gw,
                                    float* workspace, void* stream);
/* All weight re-layouts of a step in one launch (<= 64 jobs, host arrays): job j gathers from the layer's own weight tensor w[j] with
 * params[9 j ..] = {kind, ntaps, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip}; kind 0 = conv3d/conv2d_pack_weights' [tap][ci][co],
 * 1 = conv3d_pack_weights_c8's [ci/4][tap][co][4], 2 = conv3d_pack_weights_mfma's [tap][ci/8][co][8] (both straight from w, not from a packed
 * copy), 3 / 4 = the bf16 fragments of mvsnerf_conv3d_bf16_fwd / mvsnerf_conv_transpose3d_bf16_fwd (dst[j] is then a bf16 buffer of
 * mvsnerf_conv3d_bf16_packed_elems(ci_pad, co_pad, kind == 4) elements; ntaps = 27).  The weights change with every optimizer step;
 * separately these are ~70 launches of a few microseconds each. */
int mvsnerf_pack_weights_multi(int n_jobs, const float* const* w, float* const* dst, const int* params, void* stream);
int mvsnerf_conv3d_mfma_supported(int Cin, int Cout, int stride);
int mvsnerf_conv3d_pack_weights_mfma(const float* wpacked, int Cin, int Cout, float* w32, void* stream);
int mvsnerf_conv3d_mfma_fwd(const float* x1, const float* scale1, const float* shift1, int Cin, int cin_ld, int D, int H, int W,
                            const float* w32, int Cout, int stride, float* out, float* stats_part, void* stream);
/* stats_part (NULL = not wanted): per-workgroup InPlaceABN partial sums of the output, 2 * Cout floats per workgroup,
 * mvsnerf_conv3d_mfma_tiles(D, H, W, stride) workgroups, finished by mvsnerf_abn_finalize. */
int mvsnerf_conv3d_mfma_tiles(int D, int H, int W, int stride);
/* The transposed convolutions (conv7/9/11) on v_mfma_f32_32x32x2_f32: plain (already activated) input x[D][H][W][Cin], weights from
 * mvsnerf_conv3d_pack_weights_mfma; raw out[2D][2H][2W][Cout]. */
int mvsnerf_conv_transpose3d_mfma_supported(int Cin, int Cout);
int mvsnerf_conv_transpose3d_mfma_fwd(const float* x, int Cin, int D, int H, int W, const float* w32, int Cout, float* out, void* stream);
/* ... and the InPlaceABN partial sums of the raw output from the same launch (conv7 / conv9: models.py:739-746 feed an InPlaceABN): stats_part
 * [2][Cout][mvsnerf_conv_transpose3d_mfma_tiles(Cin, Cout, D, H, W)] floats for mvsnerf_abn_finalize (0 tiles: layer not built). */
int mvsnerf_conv_transpose3d_mfma_tiles(int Cin, int Cout, int D, int H, int W);
int mvsnerf_conv_transpose3d_mfma_fwd_stats(const float* x, int Cin, int D, int H, int W, const float* w32, int Cout, float* out,
                                            float* stats_part, void* stream);
/* The 16 -> 8 layer (conv11, and conv1's data gradient) without padded products on v_mfma_f32_4x4x1 (the parity-class form above spends
 * 44 % of its products on zero weights at 8 output channels).  wq: [ci/4][tap][co][4] = mvsnerf_pack_weights_multi kind 1 of the layer. */
int mvsnerf_conv_transpose3d_c8_supported(int Cin, int Cout);
/* (described in mvsnerf_hip.h: "The input is leaky(x1*scale1+shift1) [+ leaky(x2*scale2+shift2)] like mvsnerf_conv_transpose3d_fwd's (scale ...") */
int mvsnerf_conv_transpose3d_c8_fwd(const float* x1, const float* scale1, const float* shift1, const float* x2, const float* scale2, const float* shift2,
                                    int Cin, int D, int H, int W, const float* wq, float* out, float* stats_part, void* stream);
int mvsnerf_conv_transpose3d_c8_tiles(int D, int H, int W);
int mvsnerf_conv3d_c8_blocked_tiles(int D, int H, int W);
int mvsnerf_conv3d_c8_blocked_fwd_stats(const float* x_blocked, int Cin, int Cin_real, int D, int H, int W, const float* wq, float* out,
                                        float* stats_part, void* stream);
int mvsnerf_abn_finalize(const float* part, int n_blocks, int C, int64_t n_vox, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float momentum, float eps,
                         float* scale, float* shift, float* mean_out, float* invstd_out, void* stream);
/* ---- conv1 ... conv11 of CostRegNet on the bf16 matrix cores (use_amp: train_mvs_nerf_pl.py:317-318 `precision=16`; models.py:725-769) ----
 * Forward and data gradients of the nine 3x3x3 layers behind conv0 (conv0 itself: the mvsnerf_conv0_bf16_* entries above): the arguments of
 * mvsnerf_conv3d_fwd / mvsnerf_conv_transpose3d_fwd - two lazily-activated fp32 sources, channel-last - with bf16 weight fragments
 * (mvsnerf_conv3d_bf16_pack from the generic [27][Cin][Cout] layout of mvsnerf_conv3d_pack_weights: the layer's own weights or the re-packed
 * ones of its data gradient) and, when stats_part != NULL, the InPlaceABN partial sums of the output (mvsnerf_conv3d_bf16_tiles /
 * mvsnerf_conv_transpose3d_bf16_tiles slots x 2 x Cout floats, for mvsnerf_abn_finalize).  Operands are rounded to bf16 on load (round to
 * nearest even), products accumulate in fp32, outputs are fp32.  Cin, Cout in {8, 16, 32, 64} (transposed: Cin >= 16); other shapes:
 * MVSNERF_EUNSUPPORTED (mvsnerf_conv3d_bf16_packed_elems returns 0). */
size_t mvsnerf_conv3d_bf16_packed_elems(int Cin, int Cout, int transposed);
int mvsnerf_conv3d_bf16_pack(const float* wpacked, int Cin, int Cout, int transposed, void* wq, void* stream);
int mvsnerf_conv3d_bf16_tiles(int D, int H, int W, int stride);
int mvsnerf_conv3d_bf16_fwd(const float* x1, const float* scale1, const float* shift1,
                            const float* x2, const float* scale2, const float* shift2,
                            int Cin, int cin_ld, int D, int H, int W, const void* wq, int Cout, int stride,
                            float* out, float* stats_part, void* stream);
int mvsnerf_conv_transpose3d_bf16_tiles(int D, int H, int W);
int mvsnerf_conv_transpose3d_bf16_fwd(const float* x1, const float* scale1, const float* shift1,
                                      const float* x2, const float* scale2, const float* shift2,
                                      int Cin, int D, int H, int W, const void* wq, int Cout, float* out, float* stats_part, void* stream);
/* FeatureNet's 2-D layers (models.py:688-722) on the same bf16 kernels under use_amp: [N][H][W][C] images, ksize 1 | 3 | 5, stride 1 | 2,
 * padding ksize / 2, one lazily-activated source, optional bias (the 1x1 toplayer); weights = mvsnerf_pack_weights_multi kind 3 with
 * ntaps = ksize^2.  Built for the eight ConvBnReLU layers, the toplayer and their stride-1 data gradients (the two 5x5 stride-2 data
 * gradients stay on mvsnerf_conv2d_dgrad_k5s2). */
size_t mvsnerf_conv2d_bf16_packed_elems(int Cin, int Cout, int ksize);
int mvsnerf_conv2d_bf16_tiles(int N, int H, int W, int ksize, int stride);
int mvsnerf_conv2d_bf16_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int N, int H, int W,
                            const void* wq, const float* bias, int Cout, int ksize, int stride, float* out, float* stats_part, void* stream);
/* (described in mvsnerf_hip.h: "The same weight gradients with bf16 operands (use_amp; csrc/wgrad_bf16.hip): G and X are rounded to bf16 wh ...") */
int mvsnerf_conv_wgrad_bf16_parts(int A, int B, int Do, int Ho, int Wo, int kz, int k, int stride);
size_t mvsnerf_conv_wgrad_bf16_workspace_floats(int A, int B, int kz, int k);
int mvsnerf_conv_wgrad_bf16(const float* g1, const float* g1_scale, const float* g1_shift,
                            const float* g2, const float* g2_scale, const float* g2_shift, int A,
                            const float* x1, const float* x1_scale, const float* x1_shift,
                            const float* x2, const float* x2_scale, const float* x2_shift, int B, int ldx,
                            int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kz, int k, int stride,
                            float* gw, float* workspace, void* stream);
/* Every weight-gradient entry (conv3d_wgrad, conv3d_c8_blocked_wgrad, conv2d_wgrad) leaves per-workgroup partial results at the start of
 * its workspace and then reduces them (two small launches).  With gw == NULL the reduction is skipped: the caller collects the
 * (workspace, *_wgrad_parts(...) rows, A*B*taps floats per row, gw) of all layers of a backward pass and finishes them together with
 * ONE call of partial_sum_multi (<= 32 jobs; host arrays; two launches in total; a training step has ~30 weight gradients).
 * scratch: mvsnerf_partial_sum_multi_scratch_floats(sum of the jobs' n_out) floats.  Same fixed summation order either way. */
int mvsnerf_conv3d_wgrad_parts(int A, int B, int Do, int Ho, int Wo, int stride, int two_x_sources);
int mvsnerf_conv3d_c8_blocked_wgrad_parts(int Cin, int Cin_real, int D, int H, int W);
int mvsnerf_conv2d_wgrad_parts(int A, int B, int N, int Ho, int Wo, int ksize, int stride);
size_t mvsnerf_partial_sum_multi_scratch_floats(int64_t total_n_out);
int mvsnerf_partial_sum_multi(int n_jobs, const float* const* partial, const int* n_part, const int64_t* n_out, float* const* dst,
                              float* scratch, void* stream);
/* (described in mvsnerf_hip.h: "The 16- / 32-output-channel layers (conv1.x, conv2.x, and the stride-1 data gradients) run on the fp32 matr ...") */
int mvsnerf_conv2d_mfma_tiles(int Cin, int Cout, int N, int H, int W, int ksize, int stride);
int mvsnerf_conv2d_fwd_stats(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int N, int H, int W,
                             const float* wpacked, int Cout, int ksize, int stride, float* out, float* stats_part, void* stream);
/* mvsnerf_resize_bilinear (models.py:859) of N three-channel images written as [N][Ho][Wo][4] (fourth channel 0) - the plane sweep's thumbnail input - in one
 * launch instead of resize + mvsnerf_nchw_to_nhwc; the same values. */
int mvsnerf_resize_bilinear_nhwc4(const float* src_n3hw, float* dst_nhw4, int N, int Hi, int Wi, int Ho, int Wo, void* stream);
/* The depth hypotheses of MVSNet.forward (models.py:903-906, linear in depth): out[i] = near_far[0] * (1 - t[i]) + near_far[1] * t[i] with the roundings of the
 * four elementwise kernels the reference runs; t = linspace(0, 1, D) and near_far on the device (one launch, no host synchronisation). */
int mvsnerf_depth_values(const float* t, const float* near_far, int D, float* out, void* stream);
/* FeatureNet's first layer (models.py:693) from the caller's (N, 3, H, W) images, without the channel-last copy: a no-grad encode; the bits of
 * mvsnerf_conv2d_fwd_stats(Cin = 4) on the zero-padded copy.  wpacked: that layer's packed weights (cin_pad 4, cout 8). */
int mvsnerf_conv2d_c3_nchw_fwd_stats(const float* imgs_nchw, int N, int H, int W, const float* wpacked, float* out, float* stats_part, void* stream);
int mvsnerf_conv2d_dgrad_k5s2(const float* g, int Cin, int N, int Ho, int Wo, const float* wpacked, int Cout,
                              int Hi, int Wi, float* out, void* stream);
/* The same launch, additionally filling a caller-owned timing record (stateless; bench.py derives the shader clock the chip sustains
 * under this kernel from it): census[(N*S + 127) / 128][16] int64 = {start, end (100 MHz wall clock), HW_ID, XCC_ID,
 * phase stamps [4..12], shader-clock ticks of the workgroup [13], -, -}. */
int mvsnerf_mlp_fwd_census(const float* packed, int F,
                           const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                           const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only,
                           float* raw, long long* census, void* stream);
/* (described in mvsnerf_hip.h: "---- Guarded 16-bit sequences (ABI v10) ---- ...") */
typedef struct {
    const float* feats_cl; const float* imgs_cl;   /* [V][H][W][32], [V][H][W][4] as for mvsnerf_planesweep_costvar_fwd (with_img = 1) */
    const float* proj; const float* depth;         /* [V][3][4], [D] */
    int V, H, W, D, pad, CP;                       /* CP = 3V + 32 rounded up to a multiple of 4 */
    float* masks;                                  /* [V][D][Hp][Wp] */
    void* cost16x2;                                /* two fp16 planes: 2 * ceil(CP/16) * D*Hp*Wp * 16 halfs */
    float* cost32;                                 /* fp32 blocks of four channels: (CP/4) * D*Hp*Wp * 4 floats (written only on fallback) */
    const void* w_f16x3;                           /* mvsnerf_conv0_f16x3_pack */
    const float* w_c8;                             /* mvsnerf_conv3d_pack_weights_c8 (the fp32-MFMA conv0's layout) */
    int Cin;                                       /* real input channels 3V + 32 */
    float* out;                                    /* raw conv0 output [D][Hp][Wp][8] */
    float* stats_part;                             /* InPlaceABN partial sums (see above) or NULL */
    int* guard;
} mvsnerf_sweep_conv0_args;
int mvsnerf_sweep_conv0_guarded_fwd(const mvsnerf_sweep_conv0_args* a, void* stream);
/* ---- conv1 / conv2 of CostRegNet on the fp16 matrix cores with fp32-grade results (ABI v11; models.py:743-746 `conv1 = ConvBnReLU3D(8, 16, stride=2)`,
 * `conv2 = ConvBnReLU3D(16, 16)`, called at models.py:757-758) - what a no-grad encode runs for these two layers from 256 K output voxels on.
 * Two fp16 pieces per operand (activations x 2^4, weights x 2^8, exact), x1 w0 + x0 w1 + x0 w0 on v_mfma_f32_16x16x32_f16, fp32 accumulation
 * (csrc/conv_f16x3_tiled.hip).  Shapes: (Cin 8, stride 2) and (Cin 16, stride 1), Cout <= 16 (mvsnerf_conv3d_f16x3_supported).
 *   x: [D][H][W][cin_ld] fp32 RAW output of the previous layer with its pending InPlaceABN (scale, shift; both NULL: x is taken as it is);
 *   out: [Do][Ho][Wo][Cout] fp32 raw; stats_part: NULL or 2 * Cout * mvsnerf_conv3d_f16x3_slots() floats for mvsnerf_abn_finalize (n_blocks = slots).
 *   mvsnerf_conv3d_f16x3_pack: nn.Conv3d weight w[Cout][Cin][3][3][3] -> mvsnerf_conv3d_f16x3_packed_elems(Cin) fp16 elements (16-byte aligned);
 *   a weight outside fp16's range is recorded in a status tail and reported through the guard by the kernel.
 *   mvsnerf_conv3d_f16x3_fwd: the UNGUARDED kernel (an operand beyond |x| < 4094 leaves NaNs in the outputs it touches).
 *   The other end of fp16 is NOT reported by these kernels: activated inputs that are ALL below ~2^-7, or weights below 2^-11, put the second
 *   pieces into fp16's subnormals and the result silently drops from fp32 grade towards 2^-11 + log2(largest |x|) bits.  The layers this serves (conv1 / conv2
 *   behind an InPlaceABN: unit-variance inputs, |w| ~ 1e-2 .. 1) are four to seven binary orders above that floor; a caller with other data uses the fp32 entry.
 *   (The MLP kernel manages both ends itself since ABI v12; the same per-tile power-of-two scale would work here and is not built.)
 *   mvsnerf_conv3d_f16x3_guarded_fwd: the guarded sequence (see "Guarded 16-bit sequences"): the kernel above sets guard[0] when an operand left the
 *   range, the layer's fp32 kernel (w_f32: mvsnerf_conv3d_pack_weights_mfma layout for Cin 8, mvsnerf_conv3d_pack_weights layout for Cin 16) and a
 *   statistics pass run behind it predicated on that word; consume != 0 counts the event in guard[1] and re-arms guard[0] at the end. */
int mvsnerf_conv3d_f16x3_supported(int Cin, int Cout, int stride);
int mvsnerf_conv3d_f16x3_slots(void);
size_t mvsnerf_conv3d_f16x3_packed_elems(int Cin);
int mvsnerf_conv3d_f16x3_pack(const float* w, int Cin, int Cout, void* packed, void* stream);
int mvsnerf_conv3d_f16x3_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int D, int H, int W, const void* packed, int Cout,
                             int stride, float* out, float* stats_part, void* stream);
int mvsnerf_conv3d_f16x3_guarded_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int D, int H, int W,
                                     const void* w_f16x3, const float* w_f32, int Cout, int stride, float* out, float* stats_part,
                                     int* guard, int consume, void* stream);

#ifdef __cplusplus
}
#endif
#endif
