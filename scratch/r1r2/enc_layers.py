"""Scene-encode stage times (config 2) with the A/B switches of the encoder."""
import sys, torch
sys.path.insert(0, '.')
from mvsnerf_amd import _lib, encoder
from mvsnerf_amd.synth import make_rig
rig = make_rig(512, 640, seed=1234)
dev = torch.device('cuda')
ref = None
L = _lib.lib()
for blocked, mfma, fused in ((True, 1, True), (True, 1, False), (False, 1, True), (False, 0, True), (True, 1, False), (True, 1, True)):
    encoder.BLOCKED_COST = blocked
    encoder.FUSED_ABN_STATS = fused
    L.mvsnerf_tune(b"conv_mfma", mfma)
    vol, t = encoder.bench_encode(rig, dev, 24, iters=6)
    if ref is None:
        ref = vol.clone()
    print("blocked", blocked, "conv_mfma", mfma, "fused_abn_stats", fused, t, "max |vol - first|", float((vol - ref).abs().max()))
L.mvsnerf_tune(b"conv_mfma", 1)
