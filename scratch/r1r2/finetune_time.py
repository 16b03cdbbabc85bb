"""Per-scene fine-tuning step (BASELINE config 4 shape: 5 source views 800x800, 192 planes, bf16 MLP, 1024 rays x 128 samples) and the same
at the config-2 shape: MVSSystemFinetune.fit_steps = ray march forward/backward (MLP + trilinear scatter into the learnable volume) + Adam."""
import sys, time, torch
sys.path.insert(0, '.')
from mvsnerf_amd import train
from mvsnerf_amd.synth import make_rig, pose_ref_of
dev = 'cuda'

def run(name, H, W, V, D, pad, amp):
    base = (0.0, 0.25, -0.25, 0.12, -0.12, 0.1)
    rig = make_rig(H, W, n_views=V + 1, seed=404, baselines=base[:V + 1], smooth=True)
    pose = pose_ref_of(rig)
    src = (rig["images"][:, :V], rig["proj_mats"][:, :V], rig["near_fars"][0, 0], {k: v[:V] for k, v in pose.items()})
    args = train.default_args(pad=pad, batch_size=1024, N_samples=128, n_views=V, use_amp=amp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ft = train.MVSSystemFinetune(args, src, n_depth_planes=D).to(dev)
    torch.cuda.synchronize(); t_init = time.perf_counter() - t0
    g = torch.Generator().manual_seed(0)
    rays = torch.cat([torch.zeros(1024, 3), torch.nn.functional.normalize(torch.randn(1024, 3, generator=g) * 0.05 + torch.tensor([0., 0., 1.]), dim=1),
                      torch.full((1024, 1), float(rig["near_fars"][0, 0, 0])), torch.full((1024, 1), float(rig["near_fars"][0, 0, 1]))], 1)
    batch = {"rays": rays[None].to(dev), "rgbs": torch.rand(1, 1024, 3).to(dev)}
    opt = ft.configure_optimizers()[0][0]
    ft.fit_steps([batch] * 3, opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses = ft.fit_steps([batch] * 10, opt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: volume {tuple(ft.volume.feat_volume.shape)}, init (encode) {t_init*1e3:.0f} ms, fine-tune step {dt*1e3:.2f} ms "
          f"({1024/dt/1e3:.0f} k rays/s), loss {losses[0]:.4f} -> {losses[-1]:.4f}")
    del ft, opt
    torch.cuda.empty_cache()

run("config 2 shape (3 views 512x640, 128 planes, pad 24), fp32 MLP", 512, 640, 3, 128, 24, False)
run("config 2 shape, bf16 MLP", 512, 640, 3, 128, 24, True)
run("config 4 shape (5 views 800x800, 192 planes, pad 0), bf16 MLP", 800, 800, 5, 192, 0, True)
