"""Scene-encode stage times: transposed convolutions fed with a materialised (activated + skip-summed) input or with lazy operands."""
import sys, time, torch
sys.path.insert(0,'.')
from mvsnerf_amd import _lib, encoder
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
dev = torch.device('cuda')
vols = {}
for rep in range(2):
    for mat in (False, True):
        encoder.MATERIALIZE_UP_INPUT = mat
        vol, t = encoder.bench_encode(rig, dev, 24, iters=4)
        print("materialize_up_input", mat, t)
        vols[mat] = vol.clone()
print("bit-identical", bool(torch.equal(vols[False], vols[True])), "max abs diff", float((vols[False] - vols[True]).abs().max()))
