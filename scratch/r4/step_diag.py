"""Where does a ray-march step in a split-MLP mode spend its time on the HOST?  (profiles/r03_bench.json reported 1.2 ms per step for
"bf16x6" while its kernels take 0.17 ms: gather 0.011 + MLP 0.150 + compositing 0.005.)

Per step: host time of the enqueue (perf_counter around step(), no synchronisation) for 300 steps per mode; the slowest steps, every cyclic-GC
run that happened inside the loop (gc.callbacks), then the same loop with the GC off and a cProfile of 100 steps.
Run on the GPU box:  python scratch/r3/split_step_diag.py"""
import cProfile
import gc
import io
import os
import pstats
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import models, ops, renderer            # noqa: E402
from mvsnerf_amd.synth import make_rig, pose_ref_of      # noqa: E402
from mvsnerf_amd.utils import build_rays                 # noqa: E402
import bench                                              # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    rig = make_rig(512, 640, seed=1234)
    pose = {k: v.to(dev) for k, v in pose_ref_of(rig).items()}
    imgs_raw = rig["images_raw"].to(dev)
    args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0, pts_dim=3,
                                 multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024, ckpt=None, perturb=1.0,
                                 N_samples=128, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0, pad=24)
    kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(bench.load_mlp_weights())
    qfn = kw["network_query_fn"]
    vol = torch.randn((1, 8, 128, 176, 208), generator=torch.Generator().manual_seed(5)).to(dev).contiguous(memory_format=torch.channels_last_3d)
    depths = torch.zeros(1, 4, 1, 1, device=dev)
    batches = []
    with torch.no_grad():
        for _ in range(8):
            pts, rdir, _t, ndc, z, ro, _, _ = build_rays(imgs_raw, depths, pose, pose["w2cs"], pose["c2ws"], pose["intrinsics"], rig["near_fars"].to(dev), 1024, 128, pad=24)
            batches.append(tuple(t.contiguous() for t in (pts, ndc, z, ro, rdir)))
    src = imgs_raw[:, :3]

    def step(i):
        pts, ndc, z, ro, rdir = batches[i % 8]
        return renderer.rendering(args, pose, pts, ndc, z, ro, rdir, vol, src, network_fn=net, network_query_fn=qfn)

    gc_log = []
    gc.callbacks.append(lambda phase, info: gc_log.append((phase, info.get("generation"), time.perf_counter())))
    for mode in ("fp32", "auto", "fp16x3", "bf16"):
        ops.set_mlp_precision(mode)
        with torch.no_grad():
            for i in range(20):
                step(i)
            torch.cuda.synchronize()
            for gc_off in (False, True):
                if gc_off:
                    gc.collect(); gc.disable()
                del gc_log[:]
                host = []
                t0 = time.perf_counter()
                for i in range(300):
                    h0 = time.perf_counter()
                    step(i)
                    host.append(time.perf_counter() - h0)
                t_enq = time.perf_counter() - t0
                torch.cuda.synchronize()
                t_all = time.perf_counter() - t0
                gc.enable()
                hs = sorted(host)
                gcs = [(g, round((b[2] - a[2]) * 1e3, 2)) for a, b, g in zip(gc_log[0::2], gc_log[1::2], [x[1] for x in gc_log[0::2]])]
                print(f"{mode:7s} gc_off={gc_off}: {t_all / 300 * 1e3:.4f} ms/step wall, enqueue {t_enq / 300 * 1e3:.4f} ms/step; host per step median {hs[150] * 1e3:.4f} "
                      f"p99 {hs[296] * 1e3:.4f} max {hs[-1] * 1e3:.3f} ms; slowest at steps {sorted(range(300), key=lambda k: -host[k])[:5]}; gc runs (gen, ms): {gcs}")
            pr = cProfile.Profile()
            pr.enable()
            for i in range(100):
                step(i)
            pr.disable()
            torch.cuda.synchronize()
            if mode in ("auto",):
                sio = io.StringIO()
                pstats.Stats(pr, stream=sio).sort_stats("cumulative").print_stats(22)
                print(sio.getvalue()[:3500])
    ops.set_mlp_precision("fp32")


if __name__ == "__main__":
    main()
