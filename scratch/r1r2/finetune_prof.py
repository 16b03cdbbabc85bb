import sys, time, torch
sys.path.insert(0, '.')
from mvsnerf_amd import train
from mvsnerf_amd.synth import make_rig, pose_ref_of
dev = 'cuda'
H, W, V, D, pad = 512, 640, 3, 128, 24
rig = make_rig(H, W, n_views=V + 1, seed=404, smooth=True)
pose = pose_ref_of(rig)
src = (rig["images"][:, :V], rig["proj_mats"][:, :V], rig["near_fars"][0, 0], {k: v[:V] for k, v in pose.items()})
args = train.default_args(pad=pad, batch_size=1024, N_samples=128, n_views=V, use_amp=False)
ft = train.MVSSystemFinetune(args, src, n_depth_planes=D).to(dev)
g = torch.Generator().manual_seed(0)
rays = torch.cat([torch.zeros(1024, 3), torch.nn.functional.normalize(torch.randn(1024, 3, generator=g) * 0.05 + torch.tensor([0., 0., 1.]), dim=1),
                  torch.full((1024, 1), float(rig["near_fars"][0, 0, 0])), torch.full((1024, 1), float(rig["near_fars"][0, 0, 1]))], 1)
batch = {"rays": rays[None].to(dev), "rgbs": torch.rand(1, 1024, 3).to(dev)}
opt = ft.configure_optimizers()[0][0]
ft.fit_steps([batch] * 3, opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
ft.fit_steps([batch] * 10, opt)
torch.cuda.synchronize(); print("finetune step ms", (time.perf_counter() - t0) / 10 * 1e3)
