"""The remaining branches of the SURVEY 8(a)/(b) signatures, each against the CPU oracle:
  * gen_pts_feats / build_color_volume with `img_feat` (renderer.py:124-136, utils.py:300-332)
  * build_rays with ground-truth depths, `importanceSampling` and `with_depth` (utils.py:194-221) on the HIP ray-generation kernel
  * MVSSystem.validation_step / validation_epoch_end with the reference's `val_*` / `val/*` keys (train_mvs_nerf_pl.py:172-275)
  * the depth loss and depth metrics of training_step (train_mvs_nerf_pl.py:127-141)
"""
import pytest
import torch

from tests.util import load_weights, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rig(H=64, W=96, seed=5, rot=2.0):
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    rig = make_rig(H, W, seed=seed, rot_deg=rot)
    return rig, pose_ref_of(rig)


@pytest.mark.parametrize("Cf,fh,fw", [(8, 64, 96), (5, 16, 24), (1, 33, 17)])
def test_gen_pts_feats_with_img_feat_vs_oracle(Cf, fh, fw):
    from mvsnerf_amd import renderer as R, utils as U
    from oracle import mvsnerf_oracle as O
    rig, pose = _rig()
    g = torch.Generator().manual_seed(Cf)
    imgs = rig["images_raw"][:, :3]
    img_feat = torch.randn((1, 3, Cf, fh, fw), generator=g)
    vol = torch.randn((1, 8, 8, 20, 28), generator=g)
    pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], 70, 9, pad=2, t_rand=torch.rand((70, 9), generator=g), generator=g)
    pts = pts + torch.randn(pts.shape, generator=g) * 0.3                # push some samples out of the source frusta (zeros vs border padding)
    ref_c = O.build_color_volume(pts, pose, imgs, with_mask=True, img_feat=img_feat)
    ref = torch.cat([O.index_point_feature(vol, ndc), ref_c], -1)
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    with torch.no_grad():
        out_c = U.build_color_volume(pts.to(DEV), pose_d, imgs.to(DEV), img_feat=img_feat.to(DEV), with_mask=True)
        out = R.gen_pts_feats(imgs.to(DEV), vol.to(DEV), pts.to(DEV), pose_d, ndc.to(DEV), 20, img_feat=img_feat.to(DEV))
    assert out_c.shape == ref_c.shape == (70, 9, 3 * (3 + Cf + 1))
    assert out.shape == ref.shape == (70, 9, 20 + 3 * Cf)
    m = slice(3 + Cf, None, 3 + Cf + 1)
    assert torch.equal(out_c.cpu()[..., m], ref_c[..., m])               # in-frustum masks
    assert 0 < float(ref_c[..., m].mean()) < 1                            # both inside and outside samples are exercised
    assert maxabs(out_c.cpu(), ref_c) < 2e-5
    assert maxabs(out.cpu(), ref) < 5e-5


@pytest.mark.parametrize("mode", ["depths", "importanceSampling", "with_depth"])
def test_build_rays_depth_variants_vs_oracle(mode):
    """Ray indices bit-exact (same CPU-RNG draws), everything downstream against the oracle's restatement of utils.py:194-221."""
    from mvsnerf_amd import utils as U
    from oracle import mvsnerf_oracle as O
    H, W, N, S, pad = 64, 96, 130, 12, 2
    rig, pose = _rig(H, W)
    g = torch.Generator().manual_seed(3)
    depths = torch.rand((1, 4, H, W), generator=g) * 2 + 2.5
    depths[0, :, ::7, ::5] = 0.0                                          # background pixels (mask = depth > 0)
    zmap = torch.rand((H, W), generator=g) * 2 + 2.3
    imp, wd = mode == "importanceSampling", mode == "with_depth"
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    torch.manual_seed(11)
    with torch.no_grad():
        out = U.build_rays(rig["images_raw"].to(DEV), depths.to(DEV), pose_d, pose_d["w2cs"], pose_d["c2ws"], pose_d["intrinsics"],
                           zmap.to(DEV) if wd else rig["near_fars"].to(DEV), N, S, pad=pad, importanceSampling=imp, with_depth=wd)
    pts, rdir, colors, ndc, z, ro, rays_depth, _ = out
    # replay the draws: xs, ys on the CPU generator (utils.py:93), then the device jitter (:220; not drawn for with_depth)
    torch.manual_seed(11)
    xs, ys = torch.randint(0, W, (N,)), torch.randint(0, H, (N,))
    t_rand = None if wd else torch.rand((N, S), device=DEV).cpu()
    torch.manual_seed(11)                                                 # the oracle draws xs, ys from the global CPU generator again
    r = O.build_rays(rig["images_raw"], pose, zmap if wd else rig["near_fars"], N, S, pad=pad, t_rand=t_rand, generator=None,
                     depths=depths, importanceSampling=imp, with_depth=wd)
    rpts, rdir_r, rtgt, rndc, rz, rro, rpix, rdepth = r
    assert torch.equal(rpix[1], xs) and torch.equal(rpix[0], ys)          # the oracle replayed the same ids
    assert torch.equal(colors.cpu(), rtgt) and torch.equal(rays_depth.cpu(), rdepth)    # gathers are exact: ray indices bit-exact
    assert z.shape == rz.shape == ((N, 1) if wd else (N, S))
    assert maxabs(rdir.cpu(), rdir_r) < 1e-6 and maxabs(z.cpu(), rz) < 2e-6
    assert maxabs(pts.cpu(), rpts) < 5e-6 and maxabs(ndc.cpu(), rndc) < 5e-6
    assert torch.equal(ro.cpu(), rro)


def _system(dev, **over):
    import numpy as np
    from mvsnerf_amd import train
    args = train.default_args(pad=4, batch_size=96, N_samples=16, chunk=512, **over)
    system = train.MVSSystem(args, n_depth_planes=16).to(dev)
    mlp_sd, mvs_sd = load_weights()
    system.render_kwargs_train["network_fn"].load_state_dict(mlp_sd)
    system.MVSNet.load_state_dict(mvs_sd)
    return system


@pytest.mark.parametrize("with_depth", [False, True])
def test_validation_step_and_epoch_end_keys(with_depth):
    from mvsnerf_amd import train
    from mvsnerf_amd.utils import mse2psnr
    H, W = 64, 96
    system = _system(DEV, with_depth=with_depth)
    outs = []
    for seed in (1, 2):
        batch = train.synthetic_batch(H, W, seed=seed)
        if with_depth:
            g = torch.Generator().manual_seed(seed)
            batch["depths_h"] = torch.rand((1, 4, H, W), generator=g) * 2 + 2.5
            batch["depths_h"][0, :, :5] = 0.0
        log = system.validation_step(batch, 0)
        assert set(log) == {"val_psnr", "val_depth_loss_r", "val_abs_err", "mask_sum", "val_acc_0.01mm", "val_acc_0.05mm", "val_acc_0.1mm"}
        # the frame behind the numbers is render_view's; PSNR recomputed here from that frame
        rgb, depth = system.render_view(batch)
        tgt = system.unpreprocess(batch["images"])[0, -1]
        err = (torch.clamp(rgb.permute(2, 0, 1), 0, 1).cpu() - tgt).abs()
        if with_depth:
            gt = batch["depths_h"][0, -1]
            mask = gt > 0
            assert abs(float(log["val_psnr"]) - float(mse2psnr(torch.mean(err[:, mask] ** 2)))) < 1e-3
            assert float(log["mask_sum"]) == float(mask.sum())
            d = depth.cpu()
            assert abs(float(log["val_abs_err"]) - float((d - gt)[mask].abs().sum())) < 1e-2 * float(mask.sum())
            assert 0 <= float(log["val_acc_0.1mm"]) <= float(mask.sum())
        else:
            assert abs(float(log["val_psnr"]) - float(mse2psnr(torch.mean(err ** 2)))) < 1e-3
            assert float(log["mask_sum"]) == 0.0
        outs.append(log)
    if with_depth:
        system.validation_epoch_end(outs)
        vals = system.logged_values()
        assert {"val/d_loss_r", "val/PSNR", "val/abs_err", "val/acc_0.01mm", "val/acc_0.05mm", "val/acc_0.1mm"} <= set(vals)
        assert abs(vals["val/PSNR"] - sum(float(o["val_psnr"]) for o in outs) / 2) < 1e-4
        assert 0.0 <= vals["val/acc_0.1mm"] <= 1.0


def test_training_step_with_depth_loss_logs_and_backward():
    """--with_depth --with_depth_loss (train_mvs_nerf_pl.py:127-141): rays_depth gathered by the ray-generation kernel, the smooth-L1 term
    joins the loss, the reference's metric keys are logged, and the step still back-propagates to every parameter."""
    from mvsnerf_amd import train
    H, W = 64, 96
    system = _system(DEV, with_depth=True, with_depth_loss=True)
    batch = train.synthetic_batch(H, W, seed=4)
    g = torch.Generator().manual_seed(1)
    batch["depths_h"] = torch.rand((1, 4, H, W), generator=g) * 2 + 2.5
    batch["depths_h"][0, :, ::3, ::4] = 0.0
    torch.manual_seed(0)
    out = system.training_step(batch, 0)
    out["loss"].backward()
    vals = system.logged_values()
    assert {"train/loss", "train/img_mse_loss", "train/PSNR", "train/abs_err", "train/acc_l_0.01mm", "train/acc_l_0.05mm", "train/acc_l_0.1mm",
            "train/PSNR_out"} <= set(vals)
    assert vals["train/loss"] > vals["train/img_mse_loss"] > 0          # the depth term is in
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in system.grad_vars)
