"""Free-running encode time (MVSNet.forward, library defaults, config 2) + the volume's checksum; MVS_LIB selects another build."""
import os, sys, time, torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import _lib
if os.environ.get("MVS_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["MVS_LIB"]); _lib._lib = None
from mvsnerf_amd import encoder
from mvsnerf_amd.synth import make_rig
dev = torch.device('cuda')
rig = make_rig(512, 640, seed=1234)
z = np.load(os.path.join(ROOT, "tests", "golden", "mvsnerf_v0_weights.npz"))
net = encoder.MVSNet().to(dev); net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")}); net.train()
imgs = rig["images"][:, :3].to(dev); proj = rig["proj_mats"][:, :3].to(dev); nf = rig["near_fars"][0, 0].to(dev)
with torch.no_grad():
    for _ in range(5): v = net(imgs, proj, nf, pad=24)[0]
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): v = net(imgs, proj, nf, pad=24)[0]
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
print(f"{os.environ.get('MVS_LIB', 'product'):44s} encode free-running {best * 1e3:7.3f} ms   volume checksum {float(v.double().sum()):.6f}  abs max {float(v.abs().max()):.5f}")
