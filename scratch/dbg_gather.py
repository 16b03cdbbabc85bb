import sys, torch
sys.path.insert(0, '.')
from mvsnerf_amd import ops
from mvsnerf_amd.synth import make_rig
DEV='cuda'
V, n_rays, n_samples = 3, 1024, 128
g = torch.Generator().manual_seed(V * 100 + n_rays)
base = (0.0, 0.25, -0.25, 0.12, -0.12, 0.2, -0.2)
rig = make_rig(96, 128, n_views=V + 1, seed=7, baselines=base[:V] + (0.1,), rot_deg=2.0)
imgs = rig["images_raw"][0, :V].to(DEV)
w2cs, Ks = rig["w2cs"][0, :V].contiguous().to(DEV), rig["intrinsics"][0, :V].contiguous().to(DEV)
vol = torch.randn((12, 20, 28, 8), generator=g).to(DEV)
ndc = (torch.rand((n_rays, n_samples, 3), generator=g) * 1.3 - 0.15).to(DEV)
pts = (torch.randn((n_rays, n_samples, 3), generator=g) * torch.tensor([0.8, 0.6, 0.5]) + torch.tensor([0.0, 0.0, 3.0])).to(DEV)
rays_dir = torch.randn((n_rays, 3), generator=g).to(DEV)
with torch.no_grad():
    feat, dirs = ops.gather(vol, imgs, w2cs, Ks, pts, ndc, rays_dir)
    ref = torch.empty_like(feat)
    ops.volume_sample(vol, ndc, out=ref, out_stride=8 + 4 * V)
    ops.color_sample(imgs, w2cs, Ks, pts, out=ref, out_ptr=ref.data_ptr() + 32, out_stride=8 + 4 * V)
    dref = ops.dir_feature(rays_dir, w2cs[0])
d = (feat != ref)
print("n diff", int(d.sum()), "per column", d.view(-1, 8 + 4 * V).sum(0).tolist())
print("max abs", float((feat - ref).abs().max()))
idx = d.nonzero()[:5]
for i in idx:
    print(i.tolist(), float(feat[tuple(i)]), float(ref[tuple(i)]))
print("dirs equal", torch.equal(dirs, dref))
