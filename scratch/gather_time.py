"""Launches the stand-alone volume lookup and the fused gather on config-2 batches (run under rocprofv3 --kernel-trace --stats)."""
import sys, torch
sys.path.insert(0, '.')
from mvsnerf_amd import ops, _lib
from mvsnerf_amd.synth import make_rig, pose_ref_of
from mvsnerf_amd.utils import build_rays
dev = torch.device('cuda')
rig = make_rig(512, 640, seed=1234)
pose = {k: v.to(dev) for k, v in pose_ref_of(rig).items()}
imgs = rig["images_raw"].to(dev)
vol = torch.randn((1, 8, 128, 176, 208), generator=torch.Generator().manual_seed(5)).to(dev).contiguous(memory_format=torch.channels_last_3d)
vol_cl = ops.channels_last_volume(vol)
torch.manual_seed(1000)
batches = []
for _ in range(8):
    pts, rdir, _t, ndc, z, ro, _, _ = build_rays(imgs, torch.zeros(1, 4, 1, 1, device=dev), pose, pose["w2cs"], pose["c2ws"], pose["intrinsics"],
                                                 rig["near_fars"].to(dev), 1024, 128, pad=24)
    batches.append((pts.contiguous(), ndc.contiguous(), rdir.contiguous()))
L = _lib.lib()
N, S, F = 1024, 128, 20
feat = torch.empty((N, S, F), device=dev); dirs = torch.empty((N, 3), device=dev)
icl = ops.channels_last_images(imgs[0, :3])
w2c3, k3 = pose["w2cs"][:3].contiguous(), pose["intrinsics"][:3].contiguous()
st = torch.cuda.current_stream().cuda_stream
for i in range(1500):
    pts, ndc, rdir = batches[i % 8]
    L.mvsnerf_volume_sample_fwd(vol_cl.data_ptr(), 128, 176, 208, 8, ndc.data_ptr(), N * S, feat.data_ptr(), F, st)
    L.mvsnerf_gather_fwd(vol_cl.data_ptr(), 128, 176, 208, icl.data_ptr(), 3, 512, 640, w2c3.data_ptr(), k3.data_ptr(), pts.data_ptr(), ndc.data_ptr(),
                         N, S, rdir.data_ptr(), feat.data_ptr(), F, dirs.data_ptr(), st)
torch.cuda.synchronize()
