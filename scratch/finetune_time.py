"""Per-scene fine-tuning step time (config-4 code path at config-2 scene size, fp32)."""
import sys, time, torch
sys.path.insert(0, '.')
from mvsnerf_amd import train
from mvsnerf_amd.synth import make_rig, pose_ref_of
from tests.util import load_weights
DEV = 'cuda'
for use_dv, ni in ((False, 0), (True, 64)):
    args = train.default_args(pad=24, batch_size=1024, N_samples=128, N_importance=ni, use_density_volume=use_dv)
    rig = make_rig(512, 640, seed=8, smooth=True)
    pose = pose_ref_of(rig)
    src = (rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0], {k: v[:3] for k, v in pose.items()})
    ft = train.MVSSystemFinetune(args, src).to(DEV)
    mlp_sd, _ = load_weights()
    ft.network_fn.load_state_dict(mlp_sd)
    g = torch.Generator().manual_seed(0)
    K, c2w = pose["intrinsics"][3], pose["c2ws"][3]
    xs, ys = torch.rand(1024, generator=g) * 639, torch.rand(1024, generator=g) * 511
    d = torch.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], torch.ones(1024)], -1) @ c2w[:3, :3].t()
    rays = torch.cat([c2w[:3, 3].expand(1024, 3), d, torch.full((1024, 1), 2.125), torch.full((1024, 1), 4.525)], 1).to(DEV)
    tgt = torch.rand(1024, 3, generator=g).to(DEV)
    batch = {"rays": rays[None], "rgbs": tgt[None]}
    opt = ft.configure_optimizers()[0][0]
    ft.fit_steps([batch] * 3, opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ft.fit_steps([batch] * 10, opt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"finetune step (density volume {use_dv}, N_importance {ni}): {dt*1e3:.2f} ms  ({1024/dt/1e3:.0f} k rays/s)")
    del ft; torch.cuda.empty_cache()
