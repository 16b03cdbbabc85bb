import sys, torch
sys.path.insert(0,'.')
from mvsnerf_amd import ops
from oracle import mvsnerf_oracle as O
g = torch.Generator().manual_seed(1)
for S in (128, 256, 257, 300, 500):
    raw = torch.rand((3,S,4), generator=g)*3
    z = torch.sort(torch.rand((3,S), generator=g)*2+2, -1)[0]
    ref = O.raw2outputs(raw, z)
    out = ops.composite(raw.cuda(), z.cuda())
    print(S, [float((a.cpu()-b).abs().max()) for a,b in zip(out, ref)])
