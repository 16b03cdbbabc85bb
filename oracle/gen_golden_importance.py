"""TEST INFRASTRUCTURE ONLY.  tests/golden/caseC_importance.npz: outputs of the REAL reference's importance-sampling helpers
(data/ray_utils.py ray_marcher / sample_pdf / ray_marcher_fine, utils.get_ptsvolume, renderer.render_density) on seeded
inputs.  Run once in the authoring container:  python -m oracle.gen_golden_importance"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from oracle.gen_golden import save  # noqa: E402
from mvsnerf_amd.synth import make_rig, pose_ref_of  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref_models, ref_renderer, ref_utils = ref_shim.load_reference()
    ru = ref_shim.load_reference_ray_utils()
    args, kw = ref_shim.load_reference_networks()
    mlp, qfn = kw["network_fn"], kw["network_query_fn"]
    g = torch.Generator().manual_seed(11)
    H, W, pad, D, N, S, NI = 32, 48, 2, 12, 40, 16, 24
    rig = make_rig(H, W, seed=5, rot_deg=2.0)
    pose = pose_ref_of(rig)
    near_far = rig["near_fars"][0, 0]
    # rays through random pixels of the last view
    xs, ys = torch.rand(N, generator=g) * (W - 1), torch.rand(N, generator=g) * (H - 1)
    K, c2w = pose["intrinsics"][-1], pose["c2ws"][-1]
    d = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones(N)], -1) @ c2w[:3, :3].t()
    rays = torch.cat([c2w[:3, 3].expand(N, 3), d, near_far[0].expand(N, 1), near_far[1].expand(N, 1)], -1)
    with torch.no_grad():
        torch.manual_seed(3)
        pts, ro, rd, z = ru.ray_marcher(rays, N_samples=S, perturb=1.0)
        torch.manual_seed(3)
        perturb_rand = torch.rand(N, S)
        inv_scale = torch.tensor([W - 1, H - 1])
        ndc = ref_utils.get_ndc_coordinate(pose["w2cs"][0], pose["intrinsics"][0], pts, inv_scale, near=near_far[0], far=near_far[1], pad=pad)
        dens = torch.relu(torch.randn((D, H // 4 + 2 * pad, W // 4 + 2 * pad), generator=g) * 2.0)
        torch.manual_seed(4)
        pts_f, _, _, z_f = ru.ray_marcher_fine(rays, dens, z, ndc, N_importance=NI)
        torch.manual_seed(4)
        u = torch.rand(N, NI)
        # sample_pdf alone (det=True: linspace u)
        bins = torch.sort(torch.rand((N, S - 1), generator=g) * 3 + 2, -1)[0]
        wts = torch.rand((N, S - 2), generator=g)
        wts[::5] = 0.0                                                  # all-zero rows exercise the +1e-5 / denom<1e-5 branches
        s_det = ru.sample_pdf(bins, wts, NI, det=True)
        # voxel positions + density volume from the MLP
        Kq = pose["intrinsics"][0].clone(); Kq[:2] /= 4
        vox = ref_utils.get_ptsvolume(H // 4, W // 4, D, pad, near_far, Kq, pose["c2ws"][0])
        feat = torch.randn((vox.shape[0], vox.shape[1], 20), generator=g)
        density = ref_renderer.render_density(mlp, vox, feat, qfn, chunk=50)
    save("caseC_importance.npz", H=H, W=W, pad=pad, D=D, N_samples=S, N_importance=NI, rig_seed=5, rot_deg=2.0,
         rays=rays, perturb_rand=perturb_rand, density_volume=dens, u=u, bins=bins, wts=wts, vox_feat=feat,
         ref_pts=pts, ref_z=z, ref_ndc=ndc, ref_pts_fine=pts_f, ref_z_fine=z_f, ref_sample_det=s_det, ref_vox=vox, ref_density=density)


if __name__ == "__main__":
    main()
