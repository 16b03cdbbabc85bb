"""Scene-encode stage times (config 2), four iterations, twice."""
import sys, time, torch
sys.path.insert(0,'.')
from mvsnerf_amd import _lib, encoder
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
dev = torch.device('cuda')
for rep in range(3):
    vol, t = encoder.bench_encode(rig, dev, 24, iters=4)
    print(t, float(vol.abs().mean()))
