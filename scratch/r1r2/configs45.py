"""Throughput of the BASELINE config-4 / config-5 shapes on one GPU (informational; bench.py measures config 2)."""
import sys, time, torch
sys.path.insert(0, '.')
from mvsnerf_amd import train, ops
dev = 'cuda'

def run(name, H, W, n_src, D, pad, S, precision, target=None, chunk=4096):
    base = (0.0, 0.25, -0.25, 0.12, -0.12, 0.2)
    args = train.default_args(pad=pad, batch_size=1024, N_samples=S, chunk=chunk, n_views=n_src)
    system = train.MVSSystem(args, n_depth_planes=D).to(dev)
    batch = train.synthetic_batch(H, W, seed=1234, n_views=n_src + 1, baselines=base[:n_src] + (0.1,))
    ops.set_mlp_precision(precision)
    try:
        with torch.no_grad():
            system.render_view(batch, target=target)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            system.MVSNet.train()
            d, pose = system.decode_batch(dict(batch))
            vol, _, _ = system.MVSNet(d["images"][:, :n_src], d["proj_mats"][:, :n_src], pose["near_fars"][0], pad=pad)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            rgb, depth = system.render_view(batch, target=target)
            torch.cuda.synchronize(); t2 = time.perf_counter()
    finally:
        ops.set_mlp_precision("fp32")
    n_rays = rgb.shape[0] * rgb.shape[1]
    print(f"{name}: volume {tuple(vol.shape)}, encode {1e3*(t1-t0):.1f} ms, frame (encode + {n_rays} rays x {S}) {1e3*(t2-t1):.1f} ms "
          f"=> {n_rays/(t2-t1)/1e6:.2f} M rays/s end to end, finite={bool(torch.isfinite(rgb).all())}")
    del system
    torch.cuda.empty_cache()

run("config 2 (3 views 512x640, 128 planes, pad 24, fp32)", 512, 640, 3, 128, 24, 128, "fp32")
run("config 4 shape (5 views 800x800, 192 planes, pad 0, bf16 MLP)", 800, 800, 5, 192, 0, 128, "bf16")
import torch
b = train.synthetic_batch(640, 960, seed=1234)
K = b["intrinsics"][0, -1].clone(); K[0] *= 1008 / 960.0; K[1] *= 756 / 640.0
tgt = {"hw": (756, 1008), "intrinsic": K, "c2w": b["c2ws"][0, -1]}
run("config 5 shape (1008x756 target over 3 sources 960x640, 128 planes, pad 24, fp32)", 640, 960, 3, 128, 24, 128, "fp32", target=tgt)
