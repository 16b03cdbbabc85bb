"""Pins the CPU oracle (oracle/mvsnerf_oracle.py) against golden vectors produced by the REAL
reference code (oracle/gen_golden.py, run in the authoring container).  CPU only."""
import pytest
import torch

from oracle import mvsnerf_oracle as O
from tests.util import load_case, load_weights, pose_of, maxabs

CASES = ["caseA", "caseB"]
TOL = 2e-6   # same torch CPU kernels underneath -> (near) bit-identical; leave slack for op-order differences


@pytest.fixture(scope="module")
def weights():
    return load_weights()


@pytest.mark.parametrize("name", CASES)
def test_encoder_pieces(name, weights):
    c = load_case(name)
    _, mvs = weights
    pad = c["pad"]
    feats = O.feature_net(c["images"][0, :3], mvs)[None]
    assert maxabs(feats, c["ref_feats"]) < 1e-5
    feats = c["ref_feats"]
    warped, grid = O.homo_warp(feats[:, 1], c["proj_mats"][:, 1], c["depth_values"], pad=pad)
    assert grid.shape == c["ref_grid_v1"].shape
    assert torch.equal(grid, c["ref_grid_v1"])                  # the sampling grid of the reference, bit for bit (oracle._sgemm_k3)
    assert maxabs(warped, c["ref_warped_v1"]) < 1e-5
    cost, masks = O.build_volume_costvar_img(c["images"][:, :3], feats, c["proj_mats"][:, :3], c["depth_values"], pad)
    assert torch.equal(masks, c["ref_in_masks"])
    assert maxabs(cost, c["ref_cost_img"]) < 1e-5
    var, cnt = O.build_volume_costvar(feats, c["proj_mats"][:, :3], c["depth_values"], pad)
    assert torch.equal(cnt, c["ref_cost_cnt"])
    assert maxabs(var, c["ref_cost_var"]) < 1e-5
    vol = O.cost_reg_net(c["ref_cost_img"], mvs)
    assert maxabs(vol, c["ref_vol_small"]) < 2e-5


@pytest.mark.parametrize("name", CASES)
def test_mvsnet_forward_d128(name, weights):
    c = load_case(name)
    _, mvs = weights
    vol, feats, dv, _, _ = O.mvsnet_forward(c["images"][:, :3], c["proj_mats"][:, :3], c["near_fars"][0, 0], mvs, pad=c["pad"])
    assert maxabs(dv, c["ref_dv128"]) == 0
    assert maxabs(vol[:, :, ::8], c["ref_vol128_sub"]) < 5e-5
    assert abs(float(vol.double().sum()) - c["ref_vol128_sum"]) < 1e-3 * max(1.0, abs(c["ref_vol128_abssum"]) * 1e-3)


@pytest.mark.parametrize("name", CASES)
def test_rays_bit_exact(name):
    c = load_case(name)
    pose = pose_of(c)
    torch.manual_seed(7)   # same global-RNG draws as the reference run (ids utils.py:93, jitter :220)
    pts, dirs, target, ndc, z, ro, pix = O.build_rays(c["images_raw"], pose, c["near_fars"], c["N_rays"], c["N_samples"],
                                                     pad=c["pad"], t_rand=None, generator=None)
    assert torch.equal(pix[1], c["pix_xs"]) and torch.equal(pix[0], c["pix_ys"])      # ray indices bit-exact
    assert torch.equal(dirs, c["ref_rays_dir"])
    assert torch.equal(target, c["ref_target"])
    # with the fixture's jitter
    z = O.stratified_depths(c["near_fars"][0, -1, 0], c["near_fars"][0, -1, 1], c["N_rays"], c["N_samples"], c["t_rand"])
    assert maxabs(z, c["ref_depth_cand"]) < TOL
    g = torch.Generator().manual_seed(7)
    pts, dirs, target, ndc, z, ro, pix = O.build_rays(c["images_raw"], pose, c["near_fars"], c["N_rays"], c["N_samples"],
                                                     pad=c["pad"], t_rand=c["t_rand"], generator=g)
    assert torch.equal(pix[1], c["pix_xs"])
    assert maxabs(pts, c["ref_rays_pts"]) < TOL
    assert torch.equal(ndc, c["ref_rays_ndc"]) if maxabs(pts, c["ref_rays_pts"]) == 0 else maxabs(ndc, c["ref_rays_ndc"]) < TOL
    # the NDC coordinates of the REFERENCE's own points, bit for bit (oracle._rows_times_mat3_t pins the authoring host's sgemm arithmetic)
    inv = torch.tensor([c["W"] - 1, c["H"] - 1], dtype=torch.float32)
    ndc2 = O.get_ndc_coordinate(pose["w2cs"][0], pose["intrinsics"][0], c["ref_rays_pts"], inv, near=pose["near_fars"][0, 0],
                                far=pose["near_fars"][0, 1], pad=c["pad"])
    assert torch.equal(ndc2, c["ref_rays_ndc"])
    assert maxabs(ro, c["ref_rays_o"]) == 0
    t = O.build_rays_test(c["H"], c["W"], pose["c2ws"][-1], pose["w2cs"][0], pose["intrinsics"][-1], pose["near_fars"],
                          pose["near_fars"][-1], c["N_samples"], pad=c["pad"], chunk=c["N_rays"], idx=1)
    for a, k in zip(t, ["ref_test_pts", "ref_test_dir", "ref_test_ndc", "ref_test_z", "ref_test_o"]):
        assert maxabs(a, c[k]) < TOL, k


@pytest.mark.parametrize("name", CASES)
def test_raymarch_pieces(name, weights):
    c = load_case(name)
    mlp, _ = weights
    pose = pose_of(c)
    vol, ndc, pts = c["ref_vol_small"], c["ref_rays_ndc"], c["ref_rays_pts"]
    assert torch.equal(O.index_point_feature(vol, ndc), c["ref_vfeat"])                       # same ATen kernel underneath: bit-identical
    assert torch.equal(O.build_color_volume(pts, pose, c["images_raw"][:, :3]), c["ref_colors"])   # incl. the pinned projection arithmetic
    d = c["ref_rays_dir"]
    assert maxabs(O.gen_dir_feature(pose["w2cs"][0], d / d.norm(dim=-1, keepdim=True)), c["ref_dirs"]) < TOL
    assert maxabs(O.embed(ndc), c["ref_embed"]) < TOL
    raw = O.run_network_mvs(ndc, c["ref_dirs"], c["ref_input_feat"], mlp)
    assert maxabs(raw, c["ref_raw"]) < 1e-5
    sig = O.run_network_mvs(ndc, None, c["ref_input_feat"], mlp)
    assert maxabs(sig, c["ref_sigma_only"]) < 1e-5
    rgb, disp, acc, w, depth, alpha = O.raw2outputs(c["ref_raw"], c["ref_depth_cand"])
    for a, k in [(rgb, "ref_rgb"), (disp, "ref_disp"), (acc, "ref_acc"), (w, "ref_weights"), (depth, "ref_depth_map"), (alpha, "ref_alpha")]:
        assert maxabs(a, c[k]) < 1e-5 * max(1.0, float(c[k].abs().max())), k


@pytest.mark.parametrize("name", CASES)
def test_rendering_end_to_end(name, weights):
    c = load_case(name)
    mlp, _ = weights
    pose = pose_of(c)
    out = O.rendering(pose, c["ref_rays_pts"], c["ref_rays_ndc"], c["ref_depth_cand"], c["ref_rays_dir"],
                      c["ref_vol_small"], c["images_raw"][:, :3], mlp)
    for a, k in zip(out[:5], ["ref_rgb", "ref_input_feat", "ref_weights", "ref_depth_map", "ref_alpha"]):
        assert maxabs(a, c[k]) < 1e-5, k
    assert maxabs(out[6], c["ref_raw"]) < 1e-5
    outw = O.rendering(pose, c["ref_test_pts"], c["ref_test_ndc"], c["ref_test_z"], c["ref_test_dir"],
                       c["ref_vol_small"], c["images_raw"][:, :3], mlp, white_bkgd=True)
    assert maxabs(outw[0], c["ref_rgb_white"]) < 1e-5


def test_fixtures_reproduce_from_the_reference():
    """`python -m oracle.gen_golden --check`: the committed fixtures are what the imported reference produces NOW, bit for bit
    (authoring container only: /root/reference does not exist on the GPU box)."""
    import os
    from oracle import ref_shim
    if not os.path.isdir(ref_shim.REF_ROOT):
        pytest.skip("reference not present (GPU box)")
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "oracle.gen_golden", "--check"], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "gen_golden --check: OK" in r.stdout
