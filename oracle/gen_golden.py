"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REAL reference code
(/root/reference, imported behind oracle/ref_shim.py) on seeded synthetic inputs.

Run once in the authoring container:   python -m oracle.gen_golden
The fixtures are committed; /root/reference does not exist on the GPU box.

Every array named `ref_*` is an output of a reference function; everything else is an input.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402
from mvsnerf_amd.synth import make_rig, pose_ref_of  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
np_ = lambda t: t.detach().cpu().numpy()


CHECK = [False]      # --check: compare what the reference produces NOW with the committed fixtures instead of writing them
MISMATCH = []


def save(name, **arrs):
    path = os.path.join(OUT, name)
    arrs = {k: (np_(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    if CHECK[0]:
        z = np.load(path)
        missing = sorted(set(arrs) ^ set(z.files))
        bad = [k for k in arrs if k in z.files and not (arrs[k].shape == z[k].shape and arrs[k].dtype == z[k].dtype
                                                        and np.array_equal(arrs[k], z[k], equal_nan=arrs[k].dtype.kind == "f"))]
        print(f"{name}: {len(arrs) - len(bad)} of {len(arrs)} arrays reproduce bit for bit" + (f"; DIFFER: {bad}" if bad else "")
              + (f"; key sets differ: {missing}" if missing else ""))
        MISMATCH.extend(f"{name}:{k}" for k in bad + missing)
        return
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrs)} arrays")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_models, ref_renderer, ref_utils = ref_shim.load_reference()
    args, kw = ref_shim.load_reference_networks()
    mlp, mvs = kw["network_fn"], kw["network_mvs"]
    mvs.train()  # the reference keeps MVSNet in train mode at inference (train_mvs_nerf_pl.py:182)
    qfn = kw["network_query_fn"]

    # ---- weights of the shipped checkpoint (ckpts/mvsnerf-v0.tar), key names unchanged
    w = {"mlp/" + k: v for k, v in mlp.state_dict().items()}
    w.update({"mvs/" + k: v for k, v in mvs.state_dict().items()})
    save("mvsnerf_v0_weights.npz", **w)

    for case, (H, W, pad, D_direct, rot, N_rays, N_samples) in {
        "caseA": (32, 64, 4, 8, 0.0, 48, 16),     # pad>0, identity rotations
        "caseB": (64, 96, 0, 16, 3.0, 64, 24),    # pad=0, rotated cameras, some samples leave the frustum
    }.items():
        rig = make_rig(H, W, seed=1234 + len(case) + int(pad), rot_deg=rot)
        pose_ref = pose_ref_of(rig)
        imgs_n, imgs_raw, proj = rig["images"], rig["images_raw"], rig["proj_mats"]
        near_far = rig["near_fars"][0, 0]
        with torch.no_grad():
            # --- encoder pieces, direct calls with a small D
            feats = mvs.feature(imgs_n[0, :3])[None]                                    # models.py:904
            dv = torch.linspace(float(near_far[0]), float(near_far[1]), D_direct)[None]
            warped, grid = ref_utils.homo_warp(feats[:, 1], proj[:, 1], dv, pad=pad)      # utils.py:580
            with ref_shim.zero_filled_empty():
                cost_img, in_masks = mvs.build_volume_costvar_img(imgs_n[:, :3], feats, proj[:, :3], dv, pad=pad)
            if pad > 0:  # convention: uninitialised border of channels 0:3 := 0
                m = torch.zeros_like(cost_img[:, :3])
                m[..., pad:-pad, pad:-pad] = 1
                cost_img[:, :3] = torch.where(m.bool(), cost_img[:, :3], torch.zeros(()))
            cost_var, cnt = mvs.build_volume_costvar(feats, proj[:, :3], dv, pad=pad)
            rm = {k: v.clone() for k, v in mvs.state_dict().items()}
            vol_small = mvs.cost_reg_2(cost_img)                                        # models.py:756
            mvs.load_state_dict(rm)  # undo running-stat update
            # --- full forward (D hard-coded 128, models.py:914)
            # models.py:858 allocates the cost volume with torch.empty and never writes the border of channels 0:3 (pad > 0): inside
            # mvs(...) that garbage reaches CostRegNet.  Convention of the fixtures (SURVEY 7/8b "outputs fully written"): border := 0,
            # made explicit here so that the fixture does not depend on what the allocator happens to return.
            with ref_shim.zero_filled_empty():
                vol128, _, dv128 = mvs(imgs_n[:, :3], proj[:, :3], near_far, pad=pad)
            mvs.load_state_dict(rm)

            # --- rays: reference build_rays with the global CPU RNG seeded (ids: utils.py:93, jitter: :220)
            torch.manual_seed(7)
            depths = torch.zeros(1, 4, 1, 1)
            rays_pts, rays_dir, target_s, rays_ndc, depth_cand, rays_o, _, _ = ref_utils.build_rays(
                imgs_raw, depths, pose_ref, pose_ref["w2cs"], pose_ref["c2ws"], pose_ref["intrinsics"],
                rig["near_fars"], N_rays, N_samples, pad=pad)
            torch.manual_seed(7)   # replay the same draws so that the fixture also holds the raw random numbers
            xs = torch.randint(0, W, (N_rays,)); ys = torch.randint(0, H, (N_rays,)); t_rand = torch.rand(N_rays, N_samples)
            # deterministic test rays (chunk 1 of the row-major order)
            t_pts, t_dir, t_ndc, t_z, t_o, _ = ref_utils.build_rays_test(
                H, W, pose_ref["c2ws"][-1], pose_ref["w2cs"][0], pose_ref["intrinsics"][-1], pose_ref["near_fars"],
                pose_ref["near_fars"][-1], N_samples, pad=pad, chunk=N_rays, idx=1)

            # --- ray-march pieces on the small-D volume (stored exactly in the fixture)
            vfeat = ref_utils.index_point_feature(vol_small, rays_ndc)
            colors = ref_utils.build_color_volume(rays_pts, pose_ref, imgs_raw[:, :3], with_mask=True)
            dirs = ref_renderer.gen_dir_feature(pose_ref["w2cs"][0], rays_dir / rays_dir.norm(dim=-1, keepdim=True))
            emb = ref_models.get_embedder(10, 0, 3)[0](rays_ndc)
            rgb, input_feat, weights, depth_map, alpha, _ = ref_renderer.rendering(
                args, pose_ref, rays_pts, rays_ndc, depth_cand, rays_o, rays_dir, vol_small, imgs_raw[:, :3],
                network_fn=mlp, network_query_fn=qfn)
            raw = qfn(rays_ndc, dirs, input_feat, mlp)
            sigma_only = qfn(rays_ndc, None, input_feat, mlp)
            rgb_w, *_ = ref_renderer.rendering(
                args, pose_ref, t_pts, t_ndc, t_z, t_o, t_dir, vol_small, imgs_raw[:, :3],
                network_fn=mlp, network_query_fn=qfn, white_bkgd=True)
            ro = ref_renderer.raw2outputs(raw, depth_cand, None, False, "v0")
        save(f"{case}.npz",
             H=H, W=W, pad=pad, rot_deg=rot, rig_seed=1234 + len(case) + int(pad), N_rays=N_rays, N_samples=N_samples,
             images=imgs_n, images_raw=imgs_raw, proj_mats=proj, w2cs=rig["w2cs"], c2ws=rig["c2ws"],
             intrinsics=rig["intrinsics"], near_fars=rig["near_fars"], depth_values=dv,
             ref_feats=feats, ref_warped_v1=warped, ref_grid_v1=grid, ref_cost_img=cost_img, ref_in_masks=in_masks,
             ref_cost_var=cost_var, ref_cost_cnt=cnt, ref_vol_small=vol_small,
             ref_vol128_sub=vol128[:, :, ::8], ref_vol128_sum=vol128.double().sum(), ref_vol128_abssum=vol128.double().abs().sum(),
             ref_dv128=dv128,
             pix_xs=xs, pix_ys=ys, t_rand=t_rand,
             ref_rays_pts=rays_pts, ref_rays_dir=rays_dir, ref_target=target_s, ref_rays_ndc=rays_ndc,
             ref_depth_cand=depth_cand, ref_rays_o=rays_o,
             ref_test_pts=t_pts, ref_test_dir=t_dir, ref_test_ndc=t_ndc, ref_test_z=t_z, ref_test_o=t_o,
             ref_vfeat=vfeat, ref_colors=colors, ref_dirs=dirs, ref_embed=emb,
             ref_rgb=rgb, ref_input_feat=input_feat, ref_weights=weights, ref_depth_map=depth_map, ref_alpha=alpha,
             ref_raw=raw, ref_sigma_only=sigma_only, ref_rgb_white=rgb_w,
             ref_disp=ro[1], ref_acc=ro[2])


def check():
    """Re-run the reference and compare with the committed fixtures bit for bit -> list of arrays that differ."""
    CHECK[0] = True
    del MISMATCH[:]
    try:
        main()
        from oracle import gen_golden_importance
        gen_golden_importance.main()
    finally:
        CHECK[0] = False
    return list(MISMATCH)


if __name__ == "__main__":
    if "--check" in sys.argv[1:]:
        from oracle import gen_golden as _self     # the module instance gen_golden_importance imports `save` from
        bad = _self.check()
        print("gen_golden --check:", "OK" if not bad else f"{len(bad)} arrays differ")
        sys.exit(1 if bad else 0)
    main()
