"""Accuracy and speed of the MLP precision modes (fp32 MFMA / bf16 / bf16x3 / bf16x6) on the config-2 ray batch."""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from mvsnerf_amd import ops, models, _lib
from tests.util import load_weights
from tests.test_gpu_raymarch import _config2_inputs
from oracle import mvsnerf_oracle as O
DEV = 'cuda'
rig, pose, vol, pts, dirs, ndc, z, ro = _config2_inputs(1024, 128, D=32, h=48, w=64, H=128, W=160, seed=5)
mlp_sd, _ = load_weights()
ref = O.rendering(pose, pts, ndc, z, dirs, vol, rig["images_raw"][:, :3], mlp_sd)
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
net.load_state_dict(mlp_sd); net = net.to(DEV)
vol_cl = ops.channels_last_volume(vol.to(DEV))
imgs = rig["images_raw"][0, :3].to(DEV)
w2cs, Ks = pose["w2cs"][:3].contiguous().to(DEV), pose["intrinsics"][:3].contiguous().to(DEV)
a = [t.to(DEV) for t in (pts, ndc, z, dirs)]
lib = _lib.lib()
for mode in ("fp32", "bf16", "bf16x3", "bf16x6"):
    ops.set_mlp_precision(mode)
    with torch.no_grad():
        f = lambda: ops.raymarch(vol_cl, imgs, w2cs, Ks, net.packed(20), a[0], a[1], a[2], a[3], **net.packed_alt(20))
        out = f()
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    e_rgb = float((out["rgb_map"].cpu() - ref[0]).abs().max())
    raw = out["raw"].cpu()
    e_raw = (raw - ref[6]).abs()
    viol = int((e_raw > 1e-4 + 1e-4 * ref[6].abs()).sum())
    print(f"{mode:7s} step {dt*1e3:.4f} ms ({1024/dt/1e6:.2f} M rays/s)  max|rgb_map err| {e_rgb:.2e}  max|raw rgb err| {float(e_raw[...,:3].max()):.2e} "
          f"max|sigma err| {float(e_raw[...,3].max()):.2e} (sigma max {float(ref[6][...,3].max()):.1f})  violations of 1e-4+1e-4|ref|: {viol}")
ops.set_mlp_precision("fp32")
