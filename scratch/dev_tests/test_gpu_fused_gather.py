"""The opt-in gather fusion (mvsnerf_tune "mlp_gather"): gen_dir_feature + gen_pts_feats in the MLP kernel's prologue (one launch for lookups + network) against the
two-launch path: input_feat must be the SAME BITS (both run the device functions of csrc/sample_dev.h), raw / composited outputs
within fp32 rounding of each other (the MLP arithmetic is unchanged; only where its operands come from differs)."""
import pytest
import torch

from tests.util import load_weights, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n_rays,n_samples", [(1024, 128), (37, 16), (5, 200), (1, 1)])
def test_fused_gather_equals_two_launch_path(n_rays, n_samples):
    import types
    from mvsnerf_amd import _lib, models, renderer as R
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from oracle import mvsnerf_oracle as O
    mlp_sd, _ = load_weights()
    rig = make_rig(64, 96, seed=13, rot_deg=2.0, smooth=True)
    pose = pose_ref_of(rig)
    g = torch.Generator().manual_seed(n_rays)
    vol = torch.randn((1, 8, 16, 24, 32), generator=g)
    pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], n_rays, n_samples, pad=4,
                                               t_rand=torch.rand((n_rays, n_samples), generator=g), generator=g)
    ndc = ndc * 1.3 - 0.15                                   # some samples outside the volume (zeros padding) on purpose
    args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0, pts_dim=3,
                                 multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024, ckpt=None, perturb=1.0,
                                 N_samples=n_samples, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0)
    kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(mlp_sd)
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    outs = []
    for knob in (1, 0):
        assert _lib.lib().mvsnerf_tune(b"mlp_gather", knob) == 0
        try:
            with torch.no_grad():
                o = R.rendering(args, pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV), vol.to(DEV),
                                rig["images_raw"][:, :3].to(DEV), network_fn=net, network_query_fn=kw["network_query_fn"])
                outs.append([t.clone() for t in o[:5]] + [R.rendering.last_raw.clone()])
        finally:
            _lib.lib().mvsnerf_tune(b"mlp_gather", 0)          # the default (the fused launch measured 1 % slower, csrc/mlp.hip)
    fused, plain = outs
    assert torch.equal(fused[1], plain[1]), "input_feat differs: max %g" % maxabs(fused[1], plain[1])
    assert torch.equal(fused[5], plain[5]), "raw differs: max %g" % maxabs(fused[5], plain[5])      # same operands, same kernel arithmetic
    for a, b in zip(fused[:5], plain[:5]):
        assert torch.equal(a, b)
