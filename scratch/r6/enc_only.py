"""Three scene encodes at config 2 (for rocprofv3 passes over the encoder kernels)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsnerf_amd import _lib, encoder
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
vol, t = encoder.bench_encode(rig, torch.device('cuda'), 24, iters=3)
print(t)
