"""Source-view counts other than the checkpoint's 3 (BASELINE config 4: 5 source views => CostRegNet(47), feat_dim 28).
No shipped weights fit these shapes, so both sides use the same seeded random initialisation; the
oracle is V-generic and is driven with the state_dict of the HIP-side modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _args(**kw):
    import types
    d = dict(feat_dim=28, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0, pts_dim=3,
             multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024, ckpt=None, perturb=1.0,
             N_samples=32, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _rig(V, H=64, W=96, seed=41):
    from mvsnerf_amd.synth import make_rig
    base = (0.0, 0.25, -0.25, 0.12, -0.12, 0.2, -0.2, 0.05, 0.1)
    return make_rig(H, W, n_views=V + 1, seed=seed, baselines=base[:V] + (0.1,), rot_deg=2.0, smooth=True)


def _nets(V, seed):
    from mvsnerf_amd import models
    torch.manual_seed(seed)
    mvs = models.MVSNet(n_views=V)
    mlp = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=8 + 4 * V, skips=[4], net_type="v0")
    with torch.no_grad():                   # non-trivial ABN affine parameters
        for m in mvs.modules():
            if isinstance(m, models.InPlaceABN):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    return mvs, mlp


@pytest.mark.parametrize("V", [5, 2, 4])
def test_forward_other_view_counts_vs_oracle(V):
    from mvsnerf_amd import models, renderer as R
    from mvsnerf_amd.synth import pose_ref_of
    from oracle import mvsnerf_oracle as O
    pad, D, n_rays, n_samples = 4, 16, 150, 32
    rig = _rig(V)
    pose = pose_ref_of(rig)
    mvs, mlp = _nets(V, 100 + V)
    mvs_sd = {k: v.clone() for k, v in mvs.state_dict().items()}
    mlp_sd = {k: v.clone() for k, v in mlp.state_dict().items()}
    imgs_n, proj, nf = rig["images"][:, :V], rig["proj_mats"][:, :V], rig["near_fars"][0, 0]
    vol_ref, feats_ref, dv, cost_ref, masks_ref = O.mvsnet_forward(imgs_n, proj, nf, mvs_sd, pad=pad, D=D)
    g = torch.Generator().manual_seed(3)
    pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], n_rays, n_samples, pad=pad,
                                               t_rand=torch.rand((n_rays, n_samples), generator=g), generator=g)
    ref = O.rendering(pose, pts, ndc, z, dirs, vol_ref, rig["images_raw"][:, :V], mlp_sd)

    mvs = mvs.to(DEV).train(); mvs.D = D
    mlp = mlp.to(DEV)
    emb, _ = models.get_embedder(10, 0, 3)
    qfn = lambda p, vd, f, fn: R.run_network_mvs(p, vd, f, fn, emb, None)
    qfn._mvsnerf_fused = True
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    with torch.no_grad():
        # stage 1: cost volume from the oracle's features (isolates the sweep from MIOpen-vs-oneDNN FeatureNet noise)
        cost, masks = mvs.build_volume_costvar_img(imgs_n.to(DEV), feats_ref.to(DEV), proj.to(DEV), dv.to(DEV), pad=pad)
        assert cost.shape == cost_ref.shape == (1, 32 + 3 * V, D, 16 + 2 * pad, 24 + 2 * pad)
        flips = masks.cpu() != masks_ref
        assert int(flips.sum()) <= 4
        bad = flips.any(1, keepdim=True).expand_as(cost_ref)
        assert float(((cost.cpu() - cost_ref).abs() * (~bad)).max()) < 1e-4 + 3e-6 * float(feats_ref.abs().max()) ** 2
        # stage 2: CostRegNet(32+3V) on identical input
        vol = mvs.cost_reg_2(cost_ref.to(DEV))
        assert float((vol.cpu() - vol_ref).abs().max()) < 2e-4 + 1e-3 * float(vol_ref.abs().max())
        # stage 3: ray march with feat_dim 8+4V on the oracle's volume
        rgb, feat, w, depth, alpha, _ = R.rendering(_args(feat_dim=8 + 4 * V, N_samples=n_samples), pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV),
                                                    ro.to(DEV), dirs.to(DEV), vol_ref.to(DEV), rig["images_raw"][:, :V].to(DEV),
                                                    network_fn=mlp, network_query_fn=qfn)
        raw = R.rendering.last_raw.cpu()
    assert feat.shape == (n_rays, n_samples, 8 + 4 * V)
    assert float((feat.cpu() - ref[1]).abs().max()) < 1e-4
    assert float((raw - ref[6]).abs().max()) < 1e-4, "raw rgb/sigma"
    assert float((rgb.cpu() - ref[0]).abs().max()) < 1e-4
    assert float((w.cpu() - ref[2]).abs().max()) < 1e-4 and float((depth.cpu() - ref[3]).abs().max()) < 1e-4


def test_backward_five_views_vs_autograd():
    """conv0 (47->8) data/weight gradients, plane-sweep scatter with 4 warped views, MLP with F=28: against autograd
    through the oracle."""
    from mvsnerf_amd import models, renderer as R
    from mvsnerf_amd.synth import pose_ref_of
    from oracle import mvsnerf_oracle as O
    V, pad, D, n_rays, n_samples = 5, 4, 16, 96, 24
    rig = _rig(V, seed=43)
    pose = pose_ref_of(rig)
    mvs, mlp = _nets(V, 7)
    sd0 = {k: v.clone() for k, v in mvs.state_dict().items()}
    mlp_sd0 = {k: v.clone() for k, v in mlp.state_dict().items()}
    imgs_n, proj = rig["images"][:, :V], rig["proj_mats"][:, :V]
    feats0 = O.feature_net(imgs_n[0], sd0)[None].detach()
    dv = O.depth_planes(2.125, 4.525, D)
    g = torch.Generator().manual_seed(4)
    pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], n_rays, n_samples, pad=pad,
                                               t_rand=torch.rand((n_rays, n_samples), generator=g), generator=g)
    Rw = torch.randn((n_rays, 3), generator=g)

    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd0.items()}
    msd = {k: v.clone().requires_grad_(True) for k, v in mlp_sd0.items()}
    f_ref = feats0.clone().requires_grad_(True)
    cost_ref, _ = O.build_volume_costvar_img(imgs_n, f_ref, proj, dv, pad)
    vol_ref = O.cost_reg_net(cost_ref, sd)
    out_ref = O.rendering(pose, pts, ndc, z, dirs, vol_ref, rig["images_raw"][:, :V], msd)
    (out_ref[0] * Rw).sum().backward()

    mvs = mvs.to(DEV).train()
    mlp = mlp.to(DEV)
    emb, _ = models.get_embedder(10, 0, 3)
    qfn = lambda p, vd, f, fn: R.run_network_mvs(p, vd, f, fn, emb, None)
    qfn._mvsnerf_fused = True
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    f = feats0.clone().to(DEV).requires_grad_(True)
    cost, _ = mvs.build_volume_costvar_img(imgs_n.to(DEV), f, proj.to(DEV), dv.to(DEV), pad=pad)
    vol = mvs.cost_reg_2(cost)
    rgb, *_ = R.rendering(_args(feat_dim=28, N_samples=n_samples), pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV),
                          vol, rig["images_raw"][:, :V].to(DEV), network_fn=mlp, network_query_fn=qfn)
    assert float((rgb.detach().cpu() - out_ref[0].detach()).abs().max()) < 2e-3
    (rgb * Rw.to(DEV)).sum().backward()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    errs = {"feats": rel(f.grad, f_ref.grad)}
    for name, p in mvs.cost_reg_2.named_parameters():
        errs[name] = rel(p.grad, sd["cost_reg_2." + name].grad)
    for name, p in mlp.named_parameters():
        errs["mlp." + name] = rel(p.grad, msd[name].grad)
    bad = {k: v for k, v in errs.items() if not v < 5e-3}
    assert not bad, f"gradient mismatches: {bad}\nall: {errs}"
