"""Which launches of a no-grad scene encode are NOT the library's kernels (ATen elementwise / copies / fills), and which Python line issues them.
torch.profiler over three MVSNet.forward calls at config 2; prints per kernel name: launches per encode, mean us, and for the ATen / runtime ones the
innermost mvsnerf_amd frame of the call stack."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from mvsnerf_amd import encoder
from mvsnerf_amd.synth import make_rig
import numpy as np
dev = torch.device("cuda")
rig = make_rig(512, 640, seed=1234)
z = np.load("tests/golden/mvsnerf_v0_weights.npz")
net = encoder.MVSNet().to(dev)
net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")})
net.train()
imgs, proj, nf = rig["images"][:, :3].to(dev), rig["proj_mats"][:, :3].to(dev), rig["near_fars"][0, 0].to(dev)
with torch.no_grad():
    for _ in range(3):
        net(imgs, proj, nf, pad=24)
    torch.cuda.synchronize()
    N = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(N):
            net(imgs, proj, nf, pad=24)
        torch.cuda.synchronize()
ev = prof.events()
# device kernels, and the CPU op that launched each (by correlation: kernel events carry the launching op as cpu_parent in recent torch)
by = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        name = e.name[:90]
        by[name][0] += 1
        by[name][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
for name, (n, t, _) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{n / N:7.1f} per encode  {t / max(n, 1):8.1f} us  {name}")
print("\nATen ops with stacks (CPU side), per encode:")
ops = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.stack:
        fr = [s for s in e.stack if "mvsnerf_amd" in s]
        if fr and e.name not in ("aten::empty", "aten::as_strided", "aten::view", "aten::select", "aten::detach", "aten::reshape", "aten::permute", "aten::unsqueeze", "aten::empty_strided", "aten::slice", "aten::_unsafe_view", "aten::alias", "aten::squeeze", "aten::contiguous", "aten::expand", "aten::t", "aten::transpose"):
            ops[(e.name, fr[0].split("/")[-1][:80])] += 1
for (name, where), n in sorted(ops.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:6.1f}  {name:28s} {where}")
