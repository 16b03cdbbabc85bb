"""Which aten ops / memcpys / kernels does ONE MVSNet.forward (the product call, library defaults) issue?  torch profiler, config 2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mvsnerf_amd import encoder
from mvsnerf_amd.synth import make_rig
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
rig = make_rig(512, 640, seed=1234)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z = np.load(os.path.join(ROOT, "tests", "golden", "mvsnerf_v0_weights.npz"))
net = encoder.MVSNet().to(dev); net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")}); net.train()
imgs = rig["images"][:, :3].to(dev); proj = rig["proj_mats"][:, :3].to(dev); nf = rig["near_fars"][0, 0].to(dev)
with torch.no_grad():
    for _ in range(3): net(imgs, proj, nf, pad=24)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        net(imgs, proj, nf, pad=24)
        torch.cuda.synchronize()
ka = prof.key_averages()
print("---- by count")
for e in sorted(ka, key=lambda e: -e.count)[:45]:
    print(f"{e.count:5d} x  cpu {e.cpu_time_total:9.1f} us  dev {getattr(e, 'device_time_total', 0):9.1f} us  {e.key[:100]}")
