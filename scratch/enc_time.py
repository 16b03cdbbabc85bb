import sys, time, torch
sys.path.insert(0,'.')
from mvsnerf_amd import _lib, encoder
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
dev = torch.device('cuda')
for tiled in (0, 1):
    _lib.lib().mvsnerf_tune(b"conv_tiled", tiled)
    vol, t = encoder.bench_encode(rig, dev, 24, iters=3)
    print("conv_tiled", tiled, t, float(vol.abs().mean()))
    if tiled == 0: v0 = vol.clone()
print("tiled vs generic max abs diff", float((vol - v0).abs().max()))
