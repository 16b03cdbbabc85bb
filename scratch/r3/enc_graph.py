import sys, torch
sys.path.insert(0, '.')
from mvsnerf_amd import encoder as E
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
for prec in ("fp32", "bf16"):
    with E.encoder_precision(prec):
        vol, t = E.bench_encode(rig, 'cuda', 24, iters=8)
    print(prec, t)
