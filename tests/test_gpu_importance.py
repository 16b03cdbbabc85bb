"""Importance sampling of the fine-tuning option: HIP kernels vs. the real reference's outputs
(tests/golden/caseC_importance.npz) and vs. the CPU oracle on larger / ragged seeded inputs."""
import numpy as np
import pytest
import torch

from tests.util import load_case, load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(c, k):
    return torch.from_numpy(np.asarray(c[k]))


def test_golden_reference_outputs():
    from mvsnerf_amd import ops, train, utils, models, renderer as R
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    c = load_case("caseC_importance")
    H, W, pad, D, S, NI = (int(c[k]) for k in ("H", "W", "pad", "D", "N_samples", "N_importance"))
    rig = make_rig(H, W, seed=int(c["rig_seed"]), rot_deg=float(c["rot_deg"]))
    pose = {k: v.to(DEV) for k, v in pose_ref_of(rig).items()}
    near_far = rig["near_fars"][0, 0].to(DEV)
    rays, z = _t(c, "rays").to(DEV), _t(c, "ref_z").to(DEV)
    with torch.no_grad():
        # o + d*z and get_ndc_coordinate
        pts, ndc = ops.ray_points(rays[:, :3], rays[:, 3:6], z, pose["w2cs"][0], pose["intrinsics"][0], near_far, ref_hw=(H, W), pad=pad)
        assert float((pts.cpu() - _t(c, "ref_pts")).abs().max()) < 1e-5
        assert float((ndc.cpu() - _t(c, "ref_ndc")).abs().max()) < 1e-5
        # ray_marcher_fine with the reference's uniform draws
        xyz, _, _, z_f = train.ray_marcher_fine(rays, _t(c, "density_volume").to(DEV), z, _t(c, "ref_ndc").to(DEV), N_importance=NI, u=_t(c, "u").to(DEV))
        assert z_f.shape == (rays.shape[0], S + NI) and bool((z_f[:, 1:] >= z_f[:, :-1]).all())
        assert float((z_f.cpu() - _t(c, "ref_z_fine")).abs().max()) < 1e-5
        assert float((xyz.cpu() - _t(c, "ref_pts_fine")).abs().max()) < 1e-4
        # sample_pdf(det=True) incl. all-zero weight rows
        s = train.sample_pdf(_t(c, "bins").to(DEV), _t(c, "wts").to(DEV), NI, det=True)
        assert float((s.cpu() - _t(c, "ref_sample_det")).abs().max()) < 1e-5
        # voxel centres and the sigma-only density queries
        Kq = pose["intrinsics"][0].clone(); Kq[:2] /= 4
        vox = utils.get_ptsvolume(H // 4, W // 4, D, pad, near_far, Kq, pose["c2ws"][0])
        assert float((vox.cpu() - _t(c, "ref_vox")).abs().max()) < 1e-5
        mlp_sd, _ = load_weights()
        net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
        net.load_state_dict(mlp_sd)
        net = net.to(DEV)
        emb, _ = models.get_embedder(10, 0, 3)
        qfn = lambda p, vd, f, fn: R.run_network_mvs(p, vd, f, fn, emb, None)
        dens = R.render_density(net, _t(c, "ref_vox").to(DEV), _t(c, "vox_feat").to(DEV), qfn, chunk=50)
        ref = _t(c, "ref_density")
        assert float((dens.cpu().reshape(ref.shape) - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("N,S,NI,scale", [(1000, 128, 64, 0.05), (7, 5, 3, 0.05), (33, 64, 300, 0.2), (64, 3, 1, 0.05), (5, 512, 512, 0.02),
                                          (1000, 128, 64, 3.0)])
def test_ray_marcher_fine_vs_oracle(N, S, NI, scale):
    from mvsnerf_amd import ops
    from oracle import mvsnerf_oracle as O
    g = torch.Generator().manual_seed(N + S)
    dens = torch.relu(torch.randn((9, 14, 11), generator=g) * scale)
    ndc = torch.rand((N, S, 3), generator=g) * 1.1 - 0.05
    z = torch.sort(torch.rand((N, S), generator=g) * 4 + 2, -1)[0]
    u = torch.rand((N, NI), generator=g)
    u[0] = 0.0
    u[-1, -1] = 1.0 - 1e-7
    rays = torch.cat([torch.randn((N, 6), generator=g), torch.zeros(N, 2)], 1)
    ref_z = O.ray_marcher_fine(rays, dens, z, ndc, u)[3]
    with torch.no_grad():
        out = ops.ray_marcher_fine_z(dens.to(DEV), ndc.to(DEV), z.to(DEV), u.to(DEV)).cpu()
    assert bool((out[:, 1:] >= out[:, :-1]).all())
    err = (out - ref_z).abs().max(-1)[0]
    widest = (z[:, 1:] - z[:, :-1]).max(-1)[0]
    # Conditioning: inside an (almost) empty bin the pdf is ~1e-5/sum, so t = (u - cdf)/pdf amplifies 1e-7-level differences
    # of the weights (expf / fma ulps) by ~1e4: the sample moves by up to ~1 % of the bin width.  Every sample must stay
    # inside its bin; beyond the 1 % band only knot flips (u within rounding of a cdf knot) are tolerated, and counted.
    assert bool((err <= widest + 1e-5).all())
    outliers = err > 1e-5 + 0.01 * widest
    if scale < 1.0:
        assert int(outliers.sum()) <= max(1, N // 100), (int(outliers.sum()), float(err.max()))
    else:
        # saturated rays: sum(weights) ~ 1 puts the pdf of every EMPTY bin at 1e-5/(1+eps), i.e. within fp32 rounding of the
        # reference's own `denom < 1e-5` switch (data/ray_utils.py:135); which side a bin falls on depends on rounding.
        assert float(outliers.float().mean()) < 0.1


def test_sample_pdf_properties():
    """Size-independent properties: samples lie within [bins[0], bins[-1]], are monotone in u, and u = cdf knots reproduce the bins."""
    from mvsnerf_amd import ops
    g = torch.Generator().manual_seed(1)
    N, nb, NI = 257, 63, 128
    bins = torch.sort(torch.rand((N, nb), generator=g), -1)[0].to(DEV)
    w = torch.rand((N, nb - 1), generator=g).to(DEV)
    u = torch.sort(torch.rand((N, NI), generator=g), -1)[0].to(DEV)
    with torch.no_grad():
        s = ops.sample_pdf(bins, w, u)
    assert bool((s >= bins[:, :1] - 1e-6).all()) and bool((s <= bins[:, -1:] + 1e-6).all())
    assert bool((s[:, 1:] >= s[:, :-1] - 1e-6).all())


def test_finetune_with_density_volume_and_importance_sampling():
    """train_mvs_nerf_finetuning_pl.py with --use_density_volume --N_importance: density volume from the MLP, importance
    samples merged into the coarse depths, training still reduces the loss and reaches volume + MLP."""
    from mvsnerf_amd import train
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from oracle import mvsnerf_oracle as O
    args = train.default_args(pad=4, batch_size=256, N_samples=32, N_importance=16, use_density_volume=True)
    rig = make_rig(64, 96, seed=8, smooth=True)
    pose = pose_ref_of(rig)
    src = (rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0], {k: v[:3] for k, v in pose.items()})
    ft = train.MVSSystemFinetune(args, src, n_depth_planes=16).to(DEV)
    mlp_sd, _ = load_weights()
    ft.network_fn.load_state_dict(mlp_sd)
    g = torch.Generator().manual_seed(0)
    ro, rd, pix = O.get_rays_mvs(64, 96, pose["intrinsics"][3], pose["c2ws"][3], 256, generator=g)
    rays = torch.cat([ro.expand(256, 3), rd, torch.full((256, 1), 2.125), torch.full((256, 1), 4.525)], 1)
    tgt = rig["images_raw"][0, 3][:, pix[0].long(), pix[1].long()].permute(1, 0)
    torch.manual_seed(0)
    losses = ft.fit_steps([{"rays": rays[None], "rgbs": tgt[None]}] * 8)
    assert ft.density_volume is not None and ft.density_volume.shape == (16, 24, 32) and bool(torch.isfinite(ft.density_volume).all())
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    # the density volume equals the oracle's render_density on the same features
    with torch.no_grad():
        vol_cl = ft.volume.feat_volume.detach()[0].permute(1, 2, 3, 0).reshape(16 * 24, 32, 8)
        feats = torch.cat((vol_cl, ft.color_feature), -1).cpu()
        ft.update_density_volume()
    sd = {k: v.detach().cpu() for k, v in ft.network_fn.state_dict().items()}
    ref = O.render_density(ft.vox_pts.cpu(), feats, sd).reshape(16, 24, 32)
    assert float((ft.density_volume.cpu() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max()))


def test_lindisp_paths_vs_oracle():
    """`--use_disp` (inverse-depth parametrisation) of the fine-tuning script: ray_marcher sampling, the NDC z of
    get_ndc_coordinate (utils.py:130-133) in the ray_points kernel, and MVSNet.forward's inverse-depth planes (models.py:917-920)."""
    from mvsnerf_amd import ops, train, models
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from oracle import mvsnerf_oracle as O
    H, W, pad, D, N, S = 64, 96, 4, 16, 77, 20
    rig = make_rig(H, W, seed=13, rot_deg=2.0, smooth=True)
    pose = pose_ref_of(rig)
    nf = rig["near_fars"][0, 0]
    g = torch.Generator().manual_seed(1)
    K, c2w = pose["intrinsics"][-1], pose["c2ws"][-1]
    xs, ys = torch.rand(N, generator=g) * (W - 1), torch.rand(N, generator=g) * (H - 1)
    d = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones(N)], -1) @ c2w[:3, :3].t()
    rays = torch.cat([c2w[:3, 3].expand(N, 3), d, nf[0].expand(N, 1), nf[1].expand(N, 1)], -1)
    pts_ref, _, _, z_ref = O.ray_marcher(rays, S, lindisp=True)
    ndc_ref = O.get_ndc_coordinate(pose["w2cs"][0], pose["intrinsics"][0], pts_ref, torch.tensor([W - 1.0, H - 1.0]), near=nf[0], far=nf[1], pad=pad, lindisp=True)
    pts_t, _, _, z_t = train.ray_marcher(rays.to(DEV), S, lindisp=True)
    assert float((z_t.cpu() - z_ref).abs().max()) < 1e-6
    with torch.no_grad():
        pts, ndc = ops.ray_points(rays[:, :3].to(DEV), rays[:, 3:6].to(DEV), z_t, pose["w2cs"][0].to(DEV), pose["intrinsics"][0].to(DEV), nf.to(DEV),
                                  ref_hw=(H, W), pad=pad, lindisp=True)
    assert float((pts.cpu() - pts_ref).abs().max()) < 1e-5
    assert float((ndc.cpu() - ndc_ref).abs().max()) < 1e-5
    # the ray-generation kernel's lindisp flag (same sampling + NDC arithmetic from pixel ids)
    pd = {k: v.to(DEV) for k, v in pose.items()}
    with torch.no_grad():
        rp, rd, rn, rz, _ = ops.raygen(H, W, pd["intrinsics"][-1], pd["c2ws"][-1], pd["intrinsics"][0], pd["w2cs"][0], nf.to(DEV), nf.to(DEV), S,
                                       pad=pad, lindisp=True, xs=xs.to(DEV), ys=ys.to(DEV))
    assert float((rz.cpu() - z_ref).abs().max()) < 1e-6 and float((rd.cpu() - d).abs().max()) < 1e-6
    assert float((rp.cpu() - pts_ref).abs().max()) < 1e-5 and float((rn.cpu() - ndc_ref).abs().max()) < 1e-5
    # inverse-depth plane sweep through the whole encoder
    _, mvs_sd = load_weights()
    vol_ref, _, dv_ref, _, _ = O.mvsnet_forward(rig["images"][:, :3], rig["proj_mats"][:, :3], nf, mvs_sd, pad=pad, D=D, lindisp=True)
    mvs = models.MVSNet()
    mvs.load_state_dict(mvs_sd)
    mvs = mvs.to(DEV).train()
    mvs.D = D
    with torch.no_grad():
        vol, _, dv = mvs(rig["images"][:, :3].to(DEV), rig["proj_mats"][:, :3].to(DEV), nf.to(DEV), pad=pad, lindisp=True)
    assert float((dv.cpu() - dv_ref).abs().max()) < 1e-6
    assert float(((vol.cpu() - vol_ref).abs() > 5e-3).float().mean()) < 1e-3
