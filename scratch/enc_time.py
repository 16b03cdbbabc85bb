"""Scene-encode stage times with and without the XCD-contiguous tile numbering of the tiled convolutions (mvsnerf_tune "conv_xcd")."""
import sys, time, torch
sys.path.insert(0,'.')
from mvsnerf_amd import _lib, encoder
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
dev = torch.device('cuda')
vols = {}
L = _lib.lib()
for rep in range(2):
    for xcd in (0, 1):
        L.mvsnerf_tune(b"conv_xcd", xcd)
        vol, t = encoder.bench_encode(rig, dev, 24, iters=4)
        print("conv_xcd", xcd, t)
        vols[xcd] = vol.clone()
print("bit-identical", bool(torch.equal(vols[0], vols[1])))
L.mvsnerf_tune(b"conv_xcd", 1)
