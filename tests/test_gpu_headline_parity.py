"""Encoder + end-to-end parity AT THE HEADLINE SHAPE (BASELINE config 2): 3 x 512x640 source views, pad 24, 128 planes
-> cost volume 41 x 128x176x208 -> neural volume 8 x 128x176x208 -> 1024 rays x 128 samples -> RGB / sigma.

Every HIP stage against the CPU oracle (reference models.py:895-932, renderer.py:138-165):
  (1) stage-isolated: each HIP stage is fed the ORACLE's input for that stage, so its own error is visible;
  (2) chained: images -> RGB entirely on the HIP path vs. images -> RGB entirely on the oracle (north_star's 1e-4 bound
      is a statement about this number).
The measured numbers are printed (run with -s) and written to gpurun_out/headline_parity.json.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import load_weights, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"
H, W, PAD, D, N_RAYS, N_SAMPLES = 512, 640, 24, 128, 1024, 128

# Bounds asserted below (fp32; see DESIGN.md section "parity at the headline shape" for where each comes from).
# Every bound is <= 5x what was measured on MI355X (gpurun_out/headline_parity.json of round 3).
TOL_FEATS = 2e-5          # FeatureNet: 8 conv+ABN layers, |feats| <= 28; measured 7.6e-6
TOL_VOL_ISOLATED = 6e-5   # CostRegNet on the oracle's own cost volume, |vol| <= 12; measured 1.3e-5
TOL_VOL_CHAINED = 7e-5    # images -> volume, all HIP; measured 1.5e-5
TOL_RGB = 1e-5            # north_star asks 1e-4; measured 1.8e-6
TOL_SIGMA = 1e-4          # north_star, ABSOLUTE on sigma (<= 10.6 here): every one of the 131 072 samples; measured max 8.6e-6 (bound below 4e-5)
MAX_FLIPS = 0             # in-frustum mask decisions that differ, of 14 M: the projection is the reference's arithmetic, bit for bit


@pytest.fixture(scope="module")
def scene():
    """Oracle run of the whole path (CPU, a few seconds on the GPU box's host cores)."""
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from oracle import mvsnerf_oracle as O
    mlp_sd, mvs_sd = load_weights()
    rig = make_rig(H, W, seed=1234)
    pose = pose_ref_of(rig)
    imgs_n, proj, nf = rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0]
    with torch.no_grad():
        vol_ref, feats_ref, dv, cost_ref, masks_ref = O.mvsnet_forward(imgs_n, proj, nf, mvs_sd, pad=PAD, D=D)
        g = torch.Generator().manual_seed(0)
        pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], N_RAYS, N_SAMPLES, pad=PAD,
                                                   t_rand=torch.rand((N_RAYS, N_SAMPLES), generator=g), generator=g)
        ref = O.rendering(pose, pts, ndc, z, dirs, vol_ref, rig["images_raw"][:, :3], mlp_sd)
        # the yardstick: the same restatement evaluated in float64 on the same fp32 inputs, weights and rays.  |fp32 oracle - fp64|
        # is the rounding noise of the reference's own fp32 path; a HIP result as close to the fp64 values as that has nothing to fix.
        to64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
        with O.precision(torch.float64):
            vol64, _, _, cost64, _ = O.mvsnet_forward(imgs_n.double(), proj.double(), nf.double(), to64(mvs_sd), pad=PAD, D=D)
            ref64 = O.rendering(to64(pose), pts.double(), ndc.double(), z.double(), dirs.double(), vol64,
                                rig["images_raw"][:, :3].double(), to64(mlp_sd))
        noise = {"cost_variance": float((cost_ref[:, 9:].double() - cost64[:, 9:]).abs().max()),
                 "volume": float((vol_ref.double() - vol64).abs().max()),
                 "volume_rms": float(((vol_ref.double() - vol64) ** 2).mean().sqrt()),
                 "rgb": float((ref[0].double() - ref64[0]).abs().max()),
                 "sigma": float((ref[6][..., 3].double() - ref64[6][..., 3]).abs().max())}
        cvar64 = cost64[:, 9:].float()
        del cost64
    return dict(vol64=vol64, rgb64=ref64[0], sig64=ref64[6][..., 3], noise=noise, cvar64=cvar64,
                rig=rig, pose=pose, imgs_n=imgs_n, proj=proj, nf=nf, vol_ref=vol_ref, feats_ref=feats_ref, dv=dv, cost_ref=cost_ref,
                masks_ref=masks_ref, rays=(pts, dirs, ndc, z, ro), ref=ref, mlp_sd=mlp_sd, mvs_sd=mvs_sd)


@pytest.fixture(scope="module")
def mvs(scene):
    from mvsnerf_amd import models
    net = models.MVSNet()
    net.load_state_dict(scene["mvs_sd"])
    return net.to(DEV).train()


def _record(key, val):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "headline_parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        d = json.load(open(path))
    except Exception:
        d = {}
    d[key] = val
    json.dump(d, open(path, "w"), indent=1)
    print(f"[headline parity] {key} = {val}")


def test_stage_isolated(scene, mvs):
    """Each HIP stage on the oracle's input of that stage."""
    s = scene
    with torch.no_grad():
        # FeatureNet (models.py:688-722)
        feats = mvs.feature(s["imgs_n"][0].to(DEV))
        e_feats = maxabs(feats.cpu(), s["feats_ref"][0])
        _record("feature_net_max_abs_err", e_feats)
        _record("feature_abs_max", float(s["feats_ref"].abs().max()))
        # plane sweep on the oracle's features (models.py:839-893)
        cost, masks = mvs.build_volume_costvar_img(s["imgs_n"].to(DEV), s["feats_ref"].to(DEV), s["proj"].to(DEV), s["dv"].to(DEV), pad=PAD)
        assert cost.shape == s["cost_ref"].shape and masks.shape == s["masks_ref"].shape
        flips = masks.cpu() != s["masks_ref"]
        n_flips = int(flips.sum())
        _record("planesweep_mask_flips_of_%d" % flips.numel(), n_flips)
        bad = flips.any(1, keepdim=True)
        err = (cost.cpu() - s["cost_ref"]).abs()
        err.masked_fill_(bad.expand_as(err), 0)
        e_rgbch, e_var = float(err[:, :9].max()), float(err[:, 9:].max())
        _record("planesweep_rgb_channels_max_abs_err", e_rgbch)
        _record("planesweep_variance_max_abs_err", e_var)
        # the sweep follows the CPU reference operation for operation (planesweep.hip planesweep_kernel): count the values whose BITS differ
        n_bits = int((cost.cpu().view(torch.int32) != s["cost_ref"].view(torch.int32)).masked_fill_(bad.expand_as(err), False).sum())
        _record("planesweep_values_with_different_bits_of_%d" % cost.numel(), n_bits)
        e_var64 = float((cost.cpu()[:, 9:] - s["cvar64"]).abs().masked_fill_(bad.expand(-1, 32, -1, -1, -1), 0).max())
        _record("planesweep_variance_max_abs_err_vs_f64", e_var64)
        _record("oracle_f32_variance_max_abs_err_vs_f64", s["noise"]["cost_variance"])
        del err, cost
        # CostRegNet on the oracle's cost volume (models.py:725-769)
        vol = mvs.cost_reg_2(s["cost_ref"].to(DEV))
        e_vol = maxabs(vol.cpu(), s["vol_ref"])
        _record("costreg_isolated_max_abs_err", e_vol)
        _record("volume_abs_max", float(s["vol_ref"].abs().max()))
    assert e_feats < TOL_FEATS
    assert n_flips <= MAX_FLIPS
    assert e_rgbch == 0.0 and e_var == 0.0      # the plane sweep IS the reference's fp32 arithmetic (what differs in bits is the sign of zeros)
    assert e_var64 <= 1.01 * s["noise"]["cost_variance"] + 1e-5      # hence as far from the float64 variance as the reference is (e_var64 is measured against the fp32-rounded float64 values)
    assert e_vol < TOL_VOL_ISOLATED


@pytest.mark.parametrize("mode", ["fp32", "auto"])
def test_chained_images_to_rgb(scene, mvs, mode):
    """images -> FeatureNet -> plane sweep -> CostRegNet -> 1024x128 ray march, all on the HIP path, vs. all on the oracle.
    mode "fp32": fp32-MFMA conv0 + mlp_fwd_pipe_kernel (the arithmetic of bench.py's headline line); "auto": the library default of
    no-grad work (guarded fp16x3 conv0 and MLP).  Both own the same bounds; records are keyed "<mode>:<name>"."""
    from mvsnerf_amd import models, renderer as R, ops, encoder
    import types
    with ops.mlp_precision(mode), encoder.encoder_precision(mode):
        _chained(scene, mvs, mode)


def _chained(scene, mvs, mode):
    from mvsnerf_amd import models, renderer as R
    import types
    _record = lambda k, v: globals()["_record"](f"{mode}:{k}", v)
    s = scene
    pts, dirs, ndc, z, ro = s["rays"]
    ref = s["ref"]
    net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
    net.load_state_dict(s["mlp_sd"])
    net = net.to(DEV)
    emb, _ = models.get_embedder(10, 0, 3)
    qfn = lambda p, vd, f, fn: R.run_network_mvs(p, vd, f, fn, emb, None)
    qfn._mvsnerf_fused = True
    args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0, pts_dim=3,
                                 multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024, ckpt=None, perturb=1.0,
                                 N_samples=N_SAMPLES, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0)
    pose_d = {k: v.to(DEV) for k, v in s["pose"].items()}
    with torch.no_grad():
        vol, _, dv_g = mvs(s["imgs_n"].to(DEV), s["proj"].to(DEV), s["nf"].to(DEV), pad=PAD)
        assert vol.shape == (1, 8, D, H // 4 + 2 * PAD, W // 4 + 2 * PAD)
        assert maxabs(dv_g.cpu(), s["dv"]) < 1e-6
        verr = (vol.cpu() - s["vol_ref"]).abs()
        e_vol = float(verr.max())
        n_over = int((verr > TOL_VOL_CHAINED).sum())
        _record("volume_chained_max_abs_err", e_vol)
        _record("volume_chained_voxels_over_tol", n_over)
        _record("volume_chained_rms_err", float((verr.double() ** 2).mean().sqrt()))
        rgb, feat, wts, depth, alpha, _ = R.rendering(args, pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV), vol,
                                                      s["rig"]["images_raw"][:, :3].to(DEV), network_fn=net, network_query_fn=qfn)
        sig = R.rendering.last_raw.view(N_RAYS, N_SAMPLES, 4)[..., 3]
    e_rgb = maxabs(rgb.cpu(), ref[0])
    sig_ref = ref[6][..., 3]
    serr = sig.cpu().double() - sig_ref.double()
    e_sig_abs = float(serr.abs().max())
    n_sig_over = int((serr.abs() > TOL_SIGMA).sum())
    sig_p999 = float(serr.abs().flatten().kthvalue(int(0.999 * serr.numel()))[0])
    _record("sigma_end_to_end_samples_over_1e-4_of_%d" % serr.numel(), n_sig_over)
    _record("sigma_end_to_end_abs_err_p99.9", sig_p999)
    # the ray march alone: HIP ray march on the ORACLE's volume against the oracle (what is left of the sigma error when the encoder's is taken out)
    with torch.no_grad():
        vol_o = s["vol_ref"].to(DEV).contiguous(memory_format=torch.channels_last_3d)
        rgb_sv, feat_sv, *_ = R.rendering(args, pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV), vol_o,
                                 s["rig"]["images_raw"][:, :3].to(DEV), network_fn=net, network_query_fn=qfn)
        sig_sv = R.rendering.last_raw.view(N_RAYS, N_SAMPLES, 4)[..., 3].cpu()
        # ... and the network alone: the HIP MLP on the ORACLE's per-sample features (no lookup of ours in front of it)
        from mvsnerf_amd import ops
        angle = ops.dir_feature(dirs.to(DEV).contiguous(), pose_d["w2cs"][0].contiguous(), normalize=True)
        raw_iso = qfn(ndc.to(DEV).contiguous(), angle, ref[1].to(DEV).contiguous(), net).cpu()
    _record("sigma_mlp_isolated_max_abs_err", maxabs(raw_iso[..., 3], sig_ref))
    _record("sigma_mlp_isolated_samples_over_1e-4", int(((raw_iso[..., 3].double() - sig_ref.double()).abs() > TOL_SIGMA).sum()))
    _record("rgb_raw_mlp_isolated_max_abs_err", maxabs(raw_iso[..., :3], ref[6][..., :3]))
    _record("input_feat_same_volume_max_abs_err", maxabs(feat_sv.cpu(), ref[1]))
    e_sig_sv = maxabs(sig_sv, sig_ref)
    _record("sigma_same_volume_max_abs_err", e_sig_sv)
    _record("sigma_same_volume_samples_over_1e-4", int(((sig_sv.double() - sig_ref.double()).abs() > TOL_SIGMA).sum()))
    _record("rgb_same_volume_max_abs_err", maxabs(rgb_sv.cpu(), ref[0]))
    mse = float(((rgb.cpu().double() - ref[0].double()) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    _record("rgb_end_to_end_max_abs_err", e_rgb)
    _record("sigma_end_to_end_max_abs_err", e_sig_abs)
    _record("sigma_abs_max", float(sig_ref.abs().max()))
    _record("alpha_end_to_end_max_abs_err", maxabs(alpha.cpu(), ref[4]))
    _record("depth_end_to_end_max_abs_err", maxabs(depth.cpu(), ref[3]))
    _record("weights_end_to_end_max_abs_err", maxabs(wts.cpu(), ref[2]))
    _record("input_feat_end_to_end_max_abs_err", maxabs(feat.cpu(), ref[1]))
    _record("psnr_end_to_end_db", round(float(psnr), 1))
    # distances to the float64 evaluation: HIP path vs. the fp32 oracle's own rounding noise
    hip64 = {"volume": float((vol.cpu().double() - s["vol64"]).abs().max()),
             "volume_rms": float(((vol.cpu().double() - s["vol64"]) ** 2).mean().sqrt()),
             "rgb": float((rgb.cpu().double() - s["rgb64"]).abs().max()),
             "sigma": float((sig.cpu().double() - s["sig64"]).abs().max())}
    for k, v in hip64.items():
        _record(f"hip_vs_f64_{k}", v)
        _record(f"oracle_f32_vs_f64_{k}", s["noise"][k])
    assert e_rgb < TOL_RGB, e_rgb                                   # north_star: RGB within 1e-4 (bound here 1e-5)
    assert e_sig_abs < 4e-5 and n_sig_over == 0, (e_sig_abs, n_sig_over)   # north_star: sigma within 1e-4, absolute, all samples (measured 8.6e-6)
    assert e_sig_sv < 2.5e-5 and maxabs(feat_sv.cpu(), ref[1]) == 0.0  # same volume: the lookups are exact, what is left is the MLP (4.8e-6)
    assert maxabs(alpha.cpu(), ref[4]) < 1e-5                       # measured 1.7e-6
    assert maxabs(depth.cpu(), ref[3]) < 1.5e-5 and maxabs(wts.cpu(), ref[2]) < 6e-6        # measured 2.9e-6 / 1.3e-6
    # distances to the float64 evaluation: the HIP path is where the fp32 reference path is
    assert hip64["sigma"] <= 1.01 * s["noise"]["sigma"] + 2e-5, (hip64["sigma"], s["noise"]["sigma"])
    assert hip64["volume"] <= 1.05 * s["noise"]["volume"] + 2e-5, (hip64["volume"], s["noise"]["volume"])
    assert hip64["volume_rms"] <= 1.05 * s["noise"]["volume_rms"] + 1e-6
    assert e_vol < TOL_VOL_CHAINED and n_over == 0, (e_vol, n_over)
