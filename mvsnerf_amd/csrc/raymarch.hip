// One-call ray march = rendering() of the reference (renderer.py:138-165): enqueues the kernels of
// the batch on one stream from a single host call (one FFI crossing per 1024-ray batch instead of ~40
// ATen launches).
#include "common.h"

extern "C" int mvsnerf_abi_version(void) { return 12; }

// internal pieces of the guarded 16-bit sequences (include/mvsnerf_hip.h)
int mvs_mlp_f16x3_fwd(const void* packed_h, const float* packed_f32, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                      const float* dirs, int dirs_stride, int64_t P, int S, int alpha_only, float* raw, hipStream_t st, int* guard);                       // mlp_f16x3.hip
int mvs_mlp_fwd_if(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                   const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only, float* raw, const int* run_if, void* stream);                      // mlp.hip
int mvs_composite_fwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd, float* rgb_map, float* disp, float* acc, float* weights,
                      float* depth, float* alpha, int* guard, void* stream);                                                                               // composite.hip
int mvs_guard_consume(int* guard, hipStream_t st);                                                                                                         // encoder.hip

// fp16x3 kernel reporting through guard[0], then the fp32-MFMA kernel predicated on it (same inputs, same output buffer)
static int mlp_guarded_pair(const void* packed_h, const float* packed_f32, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                            const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only, float* raw, int* guard, void* stream)
{
    if (!packed_h || !packed_f32 || !ndc || !feat || !raw || !guard || N < 0 || S < 1 || feat_stride < F || ndc_stride < 3) return MVSNERF_EINVAL;
    if (!alpha_only && (!dirs || dirs_stride < 3)) return MVSNERF_EINVAL;
    if (F < 2 || F > 40 || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed_h) || !mvs_aligned16(packed_f32) || !mvs_aligned16(raw)) return MVSNERF_EALIGN;
    if (N * S == 0) return MVSNERF_OK;
    if (int rc = mvs_mlp_f16x3_fwd(packed_h, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, N * S, S, alpha_only, raw, (hipStream_t)stream, guard)) return rc;
    return mvs_mlp_fwd_if(packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, N, S, alpha_only, raw, guard, stream);
}

extern "C" int mvsnerf_mlp_fwd_guarded(const void* packed_fp16, const float* packed_f32, int F, const float* ndc, int ndc_stride,
                                       const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                                       int64_t N, int S, int alpha_only, float* raw, int* guard, void* stream)
{
    if (int rc = mlp_guarded_pair(packed_fp16, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, N, S, alpha_only, raw, guard, stream)) return rc;
    if (N * S == 0) return MVSNERF_OK;
    return mvs_guard_consume(guard, (hipStream_t)stream);
}

extern "C" int mvsnerf_raymarch_fwd(const mvsnerf_raymarch_args* a, void* stream)
{
    if (!a) return MVSNERF_EINVAL;
    if (!a->vol || !a->imgs || !a->w2c || !a->K || !a->packed_mlp || !a->rays_pts || !a->rays_ndc || !a->z_vals ||
        !a->rays_dir || !a->dirs_tmp || !a->input_feat || !a->raw)
        return MVSNERF_EINVAL;
    if (a->N < 0 || a->S < 1 || a->V < 1) return MVSNERF_EINVAL;
    const int F = 8 + 4 * a->V;
    const int64_t P = a->N * a->S;
    int rc;
    if (a->imgs_nhwc4) {
        // gen_dir_feature + gen_pts_feats in one launch (channel-last source images supplied by the caller)
        if ((rc = mvsnerf_gather_fwd(a->vol, a->D, a->H, a->W, a->imgs_nhwc4, a->V, a->IH, a->IW, a->w2c, a->K, a->rays_pts, a->rays_ndc,
                                     a->N, a->S, a->rays_dir, a->input_feat, F, a->dirs_tmp, a->vol_layout, stream))) return rc;
    } else {
        // view-direction feature in the reference camera frame (renderer.py:142-147)
        if ((rc = mvsnerf_dir_feature_fwd(a->rays_dir, a->w2c, a->N, 1, a->dirs_tmp, stream))) return rc;
        // gen_pts_feats (renderer.py:124-136): input_feat[..., :8] = volume lookup, [..., 8:] = colours + masks
        if ((rc = mvsnerf_volume_sample_fwd(a->vol, a->D, a->H, a->W, 8, a->rays_ndc, P, a->input_feat, F, a->vol_layout, stream))) return rc;
        if ((rc = mvsnerf_color_sample_fwd(a->imgs, a->V, a->IH, a->IW, a->w2c, a->K, a->rays_pts, P, 1, a->input_feat + 8, F, stream))) return rc;
    }
    // network_query_fn (renderer.py:156 -> run_network_mvs 42-63)
    const bool guarded = a->guard && a->packed_mlp_split && a->n_split == MVSNERF_SPLIT_FP16;
    if (a->guard && !guarded) return MVSNERF_EINVAL;
    if (guarded)
        rc = mlp_guarded_pair(a->packed_mlp_split, a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, 0, a->raw, a->guard, stream);
    else if (a->packed_mlp_split)
        rc = mvsnerf_mlp_fwd_split(a->packed_mlp_split, a->packed_mlp, F, a->n_split, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, 0, a->raw, stream);
    else if (a->packed_mlp_bf16)
        rc = mvsnerf_mlp_fwd_bf16(a->packed_mlp_bf16, a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, 0, a->raw, stream);
    else
        rc = mvsnerf_mlp_fwd(a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, 0, a->raw, stream);
    if (rc) return rc;
    // raw2outputs (renderer.py:162); in a guarded sequence the same launch counts a fallback and re-arms the guard
    return mvs_composite_fwd(a->raw, a->z_vals, a->N, a->S, a->white_bkgd, a->rgb_map, a->disp, a->acc, a->weights, a->depth, a->alpha,
                             guarded && P > 0 ? a->guard : nullptr, stream);
}

extern "C" int mvsnerf_raymarch_fwd_batched(const mvsnerf_raymarch_args* a, int K, void* stream)
{
    if (!a || K < 0) return MVSNERF_EINVAL;
    for (int k = 0; k < K; ++k)
        if (int rc = mvsnerf_raymarch_fwd(a + k, stream)) return rc;
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Full-frame / pixel-range render = the chunk loop of validation_step (train_mvs_nerf_pl.py:198-208):
//   for each chunk: build_rays_test (utils.py:243-297) -> rendering (renderer.py:138-165) -> keep rgb and depth.
// The whole loop is enqueued from ONE host call (4 launches per sub-batch: ray generation, fused gather, MLP,
// compositing), so a frame is not paced by ~0.1 ms of Python/ctypes work per 1024-ray chunk.  Rays are independent,
// so the sub-batch size is free (results do not depend on it); temporaries live in a caller-provided workspace that is
// reused by every sub-batch (stream order makes that safe).
// ---------------------------------------------------------------------------------------------
static size_t render_ws_floats(int64_t B, int S, int V)
{
    const int64_t P = B * S, F = 8 + 4 * V;
    // pts, ndc (3P each), z (P), feat (F*P), raw (4P), rays_dir + dirs (3B each); every block rounded up to 16 bytes
    auto r4 = [](int64_t n) { return (n + 3) & ~(int64_t)3; };
    return (size_t)(r4(3 * P) * 2 + r4(P) + r4(F * P) + r4(4 * P) + r4(3 * B) * 2);
}

extern "C" size_t mvsnerf_render_workspace_floats(int batch_rays, int S, int V)
{
    if (batch_rays < 1 || S < 1 || V < 1) return 0;
    return render_ws_floats(batch_rays, S, V);
}

extern "C" int mvsnerf_render_pixels_fwd(const mvsnerf_render_args* a, void* stream)
{
    if (!a) return MVSNERF_EINVAL;
    if (a->n_pixels == 0) return MVSNERF_OK;                       // empty pixel range (a rank with no chunks): nothing to do
    if (!a->vol || !a->imgs_nhwc4 || !a->w2c || !a->K || !a->packed_mlp || !a->K_tgt || !a->c2w_tgt || !a->K_ref || !a->w2c_ref ||
        !a->near_far_tgt || !a->near_far_ref || !a->workspace || !a->rgb)
        return MVSNERF_EINVAL;
    if (a->n_pixels < 0 || a->first_pixel < 0 || a->S < 1 || a->V < 1 || a->batch_rays < 1 || a->W_img < 2 || a->H_img < 2) return MVSNERF_EINVAL;
    if (a->first_pixel + a->n_pixels > (int64_t)a->W_img * a->H_img) return MVSNERF_EINVAL;
    if (a->workspace_floats < render_ws_floats(a->batch_rays, a->S, a->V)) return MVSNERF_EINVAL;
    const int F = 8 + 4 * a->V, S = a->S;
    const bool guarded = a->guard && a->packed_mlp_split && a->n_split == MVSNERF_SPLIT_FP16;
    if (a->guard && !guarded) return MVSNERF_EINVAL;
    const int64_t B = a->batch_rays, P = B * S;
    auto r4 = [](int64_t n) { return (n + 3) & ~(int64_t)3; };
    float* pts = a->workspace;
    float* ndc = pts + r4(3 * P);
    float* z = ndc + r4(3 * P);
    float* feat = z + r4(P);
    float* raw = feat + r4((int64_t)F * P);
    float* rdir = raw + r4(4 * P);
    float* dirs = rdir + r4(3 * B);
    int rc;
    for (int64_t off = 0; off < a->n_pixels; off += B) {
        const int64_t n = a->n_pixels - off < B ? a->n_pixels - off : B;
        if ((rc = mvsnerf_raygen_fwd(nullptr, nullptr, a->first_pixel + off, a->W_img, a->H_img, a->W_ref, a->H_ref, a->K_tgt, a->c2w_tgt, a->K_ref, a->w2c_ref,
                                     a->near_far_tgt, a->near_far_ref, a->pad, a->lindisp, nullptr, n, S, pts, rdir, ndc, z, nullptr, stream))) return rc;
        if ((rc = mvsnerf_gather_fwd(a->vol, a->D, a->H, a->W, a->imgs_nhwc4, a->V, a->IH, a->IW, a->w2c, a->K, pts, ndc, n, S, rdir,
                                     feat, F, dirs, a->vol_layout, stream))) return rc;
        if (guarded)
            rc = mlp_guarded_pair(a->packed_mlp_split, a->packed_mlp, F, ndc, 3, feat, F, dirs, 3, n, S, 0, raw, a->guard, stream);
        else if (a->packed_mlp_split)
            rc = mvsnerf_mlp_fwd_split(a->packed_mlp_split, a->packed_mlp, F, a->n_split, ndc, 3, feat, F, dirs, 3, n, S, 0, raw, stream);
        else if (a->packed_mlp_bf16)
            rc = mvsnerf_mlp_fwd_bf16(a->packed_mlp_bf16, a->packed_mlp, F, ndc, 3, feat, F, dirs, 3, n, S, 0, raw, stream);
        else
            rc = mvsnerf_mlp_fwd(a->packed_mlp, F, ndc, 3, feat, F, dirs, 3, n, S, 0, raw, stream);
        if (rc) return rc;
        if ((rc = mvs_composite_fwd(raw, z, n, S, a->white_bkgd, a->rgb + off * 3, a->disp ? a->disp + off : nullptr,
                                    a->acc ? a->acc + off : nullptr, nullptr, a->depth ? a->depth + off : nullptr, nullptr,
                                    guarded ? a->guard : nullptr, stream))) return rc;
    }
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// The differentiable ray march as two host calls: forward with activation store, backward to the MLP
// parameters and the volume.
// ---------------------------------------------------------------------------------------------
extern "C" int mvsnerf_raymarch_train_fwd(const mvsnerf_raymarch_train_args* a, void* stream)
{
    if (!a) return MVSNERF_EINVAL;
    if (!a->vol || !a->w2c || !a->packed_mlp || !a->rays_ndc || !a->z_vals || !a->rays_dir || !a->dirs_tmp || !a->input_feat || !a->raw || !a->saved)
        return MVSNERF_EINVAL;
    if (a->N < 0 || a->S < 1 || a->V < 1 || (a->bf16 && !a->packed_mlp_bf16)) return MVSNERF_EINVAL;
    const int F = 8 + 4 * a->V;
    const int64_t P = a->N * a->S;
    int rc;
    if (a->C == F && a->C != 8) {
        // --use_color_volume (renderer.py:134-135): the volume already holds the projected colours
        if ((rc = mvsnerf_dir_feature_fwd(a->rays_dir, a->w2c, a->N, 1, a->dirs_tmp, stream))) return rc;
        if ((rc = mvsnerf_volume_sample_fwd(a->vol, a->D, a->H, a->W, a->C, a->rays_ndc, P, a->input_feat, F, a->vol_layout, stream))) return rc;
    } else if (a->C == 8) {
        if (!a->imgs_nhwc4 || !a->K || !a->rays_pts) return MVSNERF_EINVAL;
        if ((rc = mvsnerf_gather_fwd(a->vol, a->D, a->H, a->W, a->imgs_nhwc4, a->V, a->IH, a->IW, a->w2c, a->K, a->rays_pts, a->rays_ndc,
                                     a->N, a->S, a->rays_dir, a->input_feat, F, a->dirs_tmp, a->vol_layout, stream))) return rc;
    } else {
        return MVSNERF_EUNSUPPORTED;
    }
    if (a->bf16) rc = mvsnerf_mlp_fwd_bf16_train(a->packed_mlp_bf16, a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, a->raw, a->saved, stream);
    else rc = mvsnerf_mlp_fwd_train(a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, a->raw, a->saved, stream);
    if (rc) return rc;
    return mvsnerf_composite_fwd(a->raw, a->z_vals, a->N, a->S, a->white_bkgd, a->rgb_map, a->disp, a->acc, a->weights, a->depth, a->alpha, stream);
}

extern "C" int mvsnerf_raymarch_bwd(const mvsnerf_raymarch_bwd_args* a, void* stream)
{
    if (!a) return MVSNERF_EINVAL;
    if (!a->packed_mlp || !a->packed_bwd || !a->raw || !a->saved || !a->z_vals || !a->rays_ndc || !a->d_raw || !a->gslots || !a->d_feat || !a->gw ||
        !a->gb || !a->maps || !a->workspace)
        return MVSNERF_EINVAL;
    if (a->N < 0 || a->S < 1) return MVSNERF_EINVAL;
    if (a->gvol && a->C != a->n_feat_out) return MVSNERF_EINVAL;
    int rc;
    if ((rc = mvsnerf_composite_bwd(a->raw, a->z_vals, a->N, a->S, a->white_bkgd, a->g_rgb, a->g_depth, nullptr, a->g_weights, a->g_alpha, a->d_raw, stream)))
        return rc;
    if (a->bf16) rc = mvsnerf_mlp_bwd_bf16(a->packed_mlp, a->packed_bwd, a->F, a->raw, a->d_raw, a->saved, a->N, a->S, a->gslots, a->d_feat, a->n_feat_out,
                                          a->gw, a->gb, a->maps, a->workspace, stream);
    else rc = mvsnerf_mlp_bwd(a->packed_mlp, reinterpret_cast<const float*>(a->packed_bwd), a->F, a->raw, a->d_raw, a->saved, a->N, a->S, a->gslots, a->d_feat,
                              a->n_feat_out, a->gw, a->gb, a->maps, a->workspace, stream);
    if (rc) return rc;
    if (a->gvol)
        return mvsnerf_volume_sample_bwd(a->D, a->H, a->W, a->C, a->rays_ndc, a->N * a->S, a->d_feat, a->n_feat_out, a->gvol, stream);
    return MVSNERF_OK;
}
