"""A/B of the fused gather (mvsnerf_tune "mlp_gather"): steady-state ms per rendering() call at config 2."""
import sys, time, types, torch, numpy as np
sys.path.insert(0, '.')
from mvsnerf_amd import _lib, models, renderer
from mvsnerf_amd.synth import make_rig, pose_ref_of
from mvsnerf_amd.utils import build_rays
dev = torch.device('cuda')
rig = make_rig(512, 640, seed=1234)
pose = {k: v.to(dev) for k, v in pose_ref_of(rig).items()}
imgs_raw = rig["images_raw"].to(dev)
args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0, pts_dim=3, multires_views=4,
                             dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024, ckpt=None, perturb=1.0, N_samples=128, use_viewdirs=True,
                             white_bkgd=False, raw_noise_std=0.0, pad=24)
kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
net = kw["network_fn"]
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")})
vol = torch.randn((1, 8, 128, 176, 208), generator=torch.Generator().manual_seed(5)).to(dev).contiguous(memory_format=torch.channels_last_3d)
torch.manual_seed(1000)
batches = []
with torch.no_grad():
    for _ in range(8):
        pts, rdir, _t, ndc, zz, ro, _, _ = build_rays(imgs_raw, torch.zeros(1, 4, 1, 1, device=dev), pose, pose["w2cs"], pose["c2ws"], pose["intrinsics"],
                                                      rig["near_fars"].to(dev), 1024, 128, pad=24)
        batches.append(tuple(t.contiguous() for t in (pts, ndc, zz, ro, rdir)))
src = imgs_raw[:, :3]
def step(i):
    pts, ndc, zz, ro, rdir = batches[i % 8]
    return renderer.rendering(args, pose, pts, ndc, zz, ro, rdir, vol, src, network_fn=net, network_query_fn=kw["network_query_fn"])
with torch.no_grad():
    for rep in range(2):
        for knob in (1, 0):
            _lib.lib().mvsnerf_tune(b"mlp_gather", knob)
            for i in range(300): step(i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(400): step(i)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
            print(f"mlp_gather {knob}: {dt*1e3:.4f} ms/step -> {1024/dt/1e6:.3f} M rays/s")
