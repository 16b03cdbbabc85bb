// One-call ray march = rendering() of the reference (renderer.py:138-165): enqueues the kernels of
// the batch on one stream from a single host call (one FFI crossing per 1024-ray batch instead of ~40
// ATen launches).
#include "common.h"

extern "C" int mvsnerf_abi_version(void) { return 3; }

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
                                     a->N, a->S, a->rays_dir, a->input_feat, F, a->dirs_tmp, stream))) return rc;
    } else {
        // view-direction feature in the reference camera frame (renderer.py:142-147)
        if ((rc = mvsnerf_dir_feature_fwd(a->rays_dir, a->w2c, a->N, 1, a->dirs_tmp, stream))) return rc;
        // gen_pts_feats (renderer.py:124-136): input_feat[..., :8] = volume lookup, [..., 8:] = colours + masks
        if ((rc = mvsnerf_volume_sample_fwd(a->vol, a->D, a->H, a->W, 8, a->rays_ndc, P, a->input_feat, F, stream))) return rc;
        if ((rc = mvsnerf_color_sample_fwd(a->imgs, a->V, a->IH, a->IW, a->w2c, a->K, a->rays_pts, P, 1, a->input_feat + 8, F, stream))) return rc;
    }
    // network_query_fn (renderer.py:156 -> run_network_mvs 42-63)
    if (a->packed_mlp_bf16)
        rc = mvsnerf_mlp_fwd_bf16(a->packed_mlp_bf16, a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, 0, a->raw, stream);
    else
        rc = mvsnerf_mlp_fwd(a->packed_mlp, F, a->rays_ndc, 3, a->input_feat, F, a->dirs_tmp, 3, a->N, a->S, 0, a->raw, stream);
    if (rc) return rc;
    // raw2outputs (renderer.py:162)
    return mvsnerf_composite_fwd(a->raw, a->z_vals, a->N, a->S, a->white_bkgd, a->rgb_map, a->disp, a->acc, a->weights, a->depth, a->alpha, stream);
}
