import sys, time, torch
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
ops.set_mlp_precision("bf16x6")
for sched in (0, 1, 0, 1):
    assert _lib.lib().mvsnerf_tune(b"split_sched", sched) == 0
    with torch.no_grad():
        f = lambda: ops.raymarch(vol_cl, imgs, w2cs, Ks, net.packed(20), a[0], a[1], a[2], a[3], **net.packed_alt(20))
        out = f()
        for _ in range(20): f()
        ts = []
        for k in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): f()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100 * 1e3)
    e = (out["raw"].cpu() - ref[6]).abs()
    print("sched", sched, " ".join(f"{t:.4f}" for t in ts), "ms/step; max raw err rgb %.2e sigma %.2e" % (float(e[..., :3].max()), float(e[..., 3].max())))
