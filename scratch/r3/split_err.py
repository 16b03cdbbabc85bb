"""Measured errors of the opt-in split-bf16 MLP modes against the CPU oracle (the inputs of tests/test_gpu_raymarch.py's split test)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_raymarch as T
from util import load_weights
from mvsnerf_amd import ops, renderer as R, models as M
from oracle import mvsnerf_oracle as O
DEV = 'cuda'
mlp_sd, _ = load_weights()
net = M.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
net.load_state_dict(mlp_sd)
net = net.to(DEV)
for n_rays, n_samples in ((1024, 128), (256, 128)):
    rig, pose, vol, pts, dirs, ndc, z, ro = T._config2_inputs(n_rays, n_samples, D=32, h=48, w=64, H=128, W=160, seed=n_rays)
    mlp_sd, _ = load_weights()
    ref = O.rendering(pose, pts, ndc, z, dirs, vol, rig["images_raw"][:, :3], mlp_sd)
    emb, _ = M.get_embedder(10, 0, 3)
    qfn = lambda p, vd, f, fn: R.run_network_mvs(p, vd, f, fn, emb, None)
    qfn._mvsnerf_fused = True
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    for mode in ("fp32", "bf16x6", "bf16x3", "bf16"):
        ops.set_mlp_precision(mode)
        try:
            with torch.no_grad():
                rgb, feat, w, depth, alpha, _ = R.rendering(T._args(N_samples=n_samples), pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV),
                                                            vol.to(DEV), rig["images_raw"][:, :3].to(DEV), network_fn=net, network_query_fn=qfn)
                raw = R.rendering.last_raw.cpu()
        finally:
            ops.set_mlp_precision("fp32")
        d = (raw - ref[6]).abs()
        print(f"{n_rays}x{n_samples} {mode:7s}: max |raw rgb err| {float(d[..., :3].max()):.2e}  max |sigma err| {float(d[..., 3].max()):.2e} (sigma max {float(ref[6][..., 3].abs().max()):.1f})  "
              f"rendered rgb {float((rgb.cpu() - ref[0]).abs().max()):.2e}  weights {float((w.cpu() - ref[2]).abs().max()):.2e}  depth {float((depth.cpu() - ref[3]).abs().max()):.2e}")
