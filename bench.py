#!/usr/bin/env python
"""bench.py - rendered rays/s of the MVSNeRF hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the ray march (`rendering()`: view-dir feature -> trilinear volume lookup + per-view
colour lookup -> Renderer_ours MLP -> alpha compositing) over one batch of 1024 rays x 128 samples, BASELINE
config 2 ("DTU scan1, 3 views, 128 planes, 1024 rays x 128 samples, fp32, 1 MI355X"): synthetic 3-view
512x640 inputs, pad 24 => neural volume 128x176x208x8, the shipped checkpoint's weights (tests/golden) or
seeded random weights.  Inputs are resident in HBM before the timed region.  The scene encode (plane sweep +
CostRegNet, once per scene) is timed separately and reported in `encode_ms`.

  python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 the driver launches one rank per GPU via torch.distributed.run; rays shard across ranks with no
data-path collective (weak scaling: every rank renders its own 1024-ray batches of the same scene).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (fused MLP, MFMA-bound): achieved = 251 392 FLOP/sample x samples per launch
                / average launch duration measured with HIP events on the launch stream
  rooflines     same for the HBM-bound gathers (volume lookup 300 B/sample - the kernel north_star puts the
                60 % bar on -, colour lookup 204 B/sample)
  cpu_baseline  the CPU oracle (torch CPU kernels = what the reference runs on a CPU) on the same workload
"""
import argparse
import json
import os
import sys
import time

import torch

from bench_common import *        # noqa: F401,F403,E402  (constants, event_time, settle, load_mlp_weights, PMC helpers: shared with bench_extras.py)
from bench_common import _pmc_summary        # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-batches", type=int, default=20, help="CPU-oracle batches timed for cpu_baseline (0 = skip)")
    ap.add_argument("--mlp-precision", default="fp32", choices=["fp32", "bf16", "bf16x3", "bf16x6", "fp16x3"],
                    help="matrix-core arithmetic of the timed MLP (default fp32 = the headline; the others are the opt-in modes)")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed steps of the same workload run before the W warmup steps until the GPU clocks have left the idle state")
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end frame / training-step timings")
    ap.add_argument("--shared-gpu-dry-run", action="store_true",
                    help="plumbing check for a 1-GPU box: --gpus N ranks all on cuda:0, collectives over gloo (RCCL refuses several ranks per device); runs the "
                         "torch.distributed.run re-exec and the collective-carrying legs end to end and prints NO headline (no metric / value keys)")
    ap.add_argument("--multi-gpu-legs", action="store_true", help="run the collective-carrying legs (tile-parallel frame, DP training step) "
                                                                  "also at N = 1 (they always run at N > 1)")
    return ap.parse_args()

def self_launch(a):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): re-execute under torch.distributed.run with one rank per
    GPU - the same command line the driver uses.  Never degrades to fewer ranks: too few visible GPUs is an error."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()
    if n_vis < a.gpus and not a.shared_gpu_dry_run:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {n_vis} GPU(s) visible; refusing to report a {a.gpus}-GPU number from fewer ranks")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))

def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher must start exactly one rank per requested GPU")
    if a.shared_gpu_dry_run:
        local = 0                                  # every rank on the one visible GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if a.shared_gpu_dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {a.gpus}")
    if a.shared_gpu_dry_run:
        # NOT a measurement: N ranks time-share one GPU and the collectives cross host memory.  It proves that the launcher path, the sharding,
        # seeding and gather code and the HIP kernels run together under world size N, and that N-rank results equal 1-rank results.
        from bench_extras import multi_gpu_legs
        multi = multi_gpu_legs(dev, rank, world, train_steps=2, shared_gpu=True)
        if rank == 0:
            multi["measured_on_hardware"] = False
            multi["note"] = ("shared-GPU dry run: %d ranks on ONE GPU, gloo collectives staged through host memory; timings are meaningless and are "
                             "reported only to show that every leg ran" % world)
            print(json.dumps({"shared_gpu_dry_run": True, "n_ranks": world, "gpus_visible": torch.cuda.device_count(), "multi_gpu": multi}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    from mvsnerf_amd import _lib, models, ops, renderer
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from mvsnerf_amd.utils import build_rays
    import types
    ops.set_mlp_precision(a.mlp_precision)        # fp32 unless asked otherwise; restored to fp32 for the per-kernel section

    # ---------------- scene + network (resident in HBM before timing)
    rig = make_rig(H_IMG, W_IMG, seed=1234)
    pose = {k: v.to(dev) for k, v in pose_ref_of(rig).items()}
    imgs_raw = rig["images_raw"].to(dev)
    args = types.SimpleNamespace(feat_dim=8 + 4 * N_SRC, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10,
                                 i_embed=0, pts_dim=3, multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0,
                                 netchunk=1024, ckpt=None, perturb=1.0, N_samples=N_SAMPLES, use_viewdirs=True, white_bkgd=False,
                                 raw_noise_std=0.0, pad=PAD)
    kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(load_mlp_weights())
    qfn = kw["network_query_fn"]
    h, w = H_IMG // 4 + 2 * PAD, W_IMG // 4 + 2 * PAD
    encode_ms = None
    try:
        from mvsnerf_amd import encoder  # plane sweep + CostRegNet (HIP)
        enc_ready = True
    except ImportError:
        enc_ready = False
    encode_ms_bf16 = encode_ms_h3 = None
    if enc_ready:
        vol, encode_ms = encoder.bench_encode(rig, dev, PAD)
        encode_ms["conv0_arithmetic"] = ("guarded fp16x3 (the default of a no-grad encode): two fp16 pieces per operand, x0*w0 + x0*w1 + x1*w0 on v_mfma_f32_16x16x32_f16, fp32 "
                                         "accumulation; a cost value or weight outside fp16's range trips a device-side guard and the fp32 sweep + fp32-MFMA conv0 enqueued behind "
                                         "the pair (predicated on it) recompute the layer; encode_ms_fp32_conv0 = the same encode with conv0 on v_mfma_f32_4x4x1_16B_f32")
        encode_ms["guard_fallbacks_so_far"] = ops.guard_fallbacks()
        volume_src = "mvsnet-encode"
        if not a.no_extras:
            # opt-in (NOT the headline volume, which stays fp32): the encoder with conv0 on the bf16 matrix cores from a bf16 cost volume
            with encoder.encoder_precision("bf16"):
                vol_b, encode_ms_bf16 = encoder.bench_encode(rig, dev, PAD)
            encode_ms_bf16["max_abs_volume_diff_vs_default_encode"] = float((vol_b - vol).abs().max())
            encode_ms_bf16["volume_abs_max"] = float(vol.abs().max())
            del vol_b
            # the same encode with conv0 on the fp32-MFMA kernel (encoder_precision "fp32"): the default above ("auto") runs conv0 of a no-grad encode as
            # two fp16 pieces per operand, three fp16 matrix-core products per product, fp32 accumulation (csrc/conv_f16x3.hip; fp32-grade: DESIGN.md 0a)
            with encoder.encoder_precision("fp32"):
                vol_h, encode_ms_h3 = encoder.bench_encode(rig, dev, PAD)
            encode_ms_h3["max_abs_volume_diff_vs_default_encode"] = float((vol_h - vol).abs().max())
            encode_ms_h3["volume_abs_max"] = float(vol.abs().max())
            del vol_h
    else:
        vol = torch.randn((1, 8, D_PLANES, h, w), generator=torch.Generator().manual_seed(5)).to(dev)
        vol = vol.contiguous(memory_format=torch.channels_last_3d)
        volume_src = "random (encoder kernels not built yet)"

    # ---------------- per-step ray batches: pre-drawn (build_rays is host-side torch in the reference too)
    torch.manual_seed(1000 + rank)
    n_batches = 8
    batches = []
    depths = torch.zeros(1, 4, 1, 1, device=dev)
    with torch.no_grad():
        for _ in range(n_batches):
            pts, rdir, _tgt, ndc, z, ro, _, _ = build_rays(imgs_raw, depths, pose, pose["w2cs"], pose["c2ws"], pose["intrinsics"],
                                                           rig["near_fars"].to(dev), N_RAYS, N_SAMPLES, pad=PAD)
            batches.append(tuple(t.contiguous() for t in (pts, ndc, z, ro, rdir)))
    src = imgs_raw[:, :N_SRC]

    def step(i):
        pts, ndc, z, ro, rdir = batches[i % n_batches]
        return renderer.rendering(args, pose, pts, ndc, z, ro, rdir, vol, src, network_fn=net, network_query_fn=qfn)

    import gc
    gc.collect(); gc.disable()                # no cyclic-GC pause of the Python host inside the timed region (nothing is skipped: the steps allocate no cycles).
                                              # Collected HERE, before the clock-settle phase: a ~0.1 s host pause between warmup and timing lets the GPU clocks drop
    with torch.no_grad():
        settle(step, a.settle_ms, indexed=True)
        for i in range(a.warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        gc.enable()
    dt_rank = dt
    rccl_ranks_seen, per_rank = 1, [round(a.steps * N_RAYS / dt_rank, 1)]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # self-check of the scaling curve: the number of ranks an actual RCCL all-reduce summed over, and every rank's own rate (its K steps over ITS
        # barrier-to-barrier time): at N ranks each entry should equal the N = 1 value (weak scaling, no data-path collective)
        one = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        rccl_ranks_seen = int(round(float(one.item())))
        mine = torch.tensor([a.steps * N_RAYS / dt_rank], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [round(float(x.item()), 1) for x in allr]
        if rccl_ranks_seen != world or dist.get_world_size() != world:
            raise SystemExit(f"all-reduce summed over {rccl_ranks_seen} ranks, world size {dist.get_world_size()}, --gpus {world}")
    rays_per_s = world * a.steps * N_RAYS / dt

    ops.set_mlp_precision("fp32")
    # ---------------- N > 1: the collective-carrying paths (every rank takes part; rank 0 reports)
    multi = None
    if (world > 1 and not a.no_extras) or a.multi_gpu_legs:
        from bench_extras import multi_gpu_legs
        multi = multi_gpu_legs(dev, rank, world)
    # ---------------- per-kernel launch durations (HIP events on the launch stream), rank 0
    roof, roofs, cpu, o = None, [], None, None
    if rank == 0:
        with torch.no_grad():
            pts, ndc, z, ro, rdir = batches[0]
            vol_cl = ops.channels_last_volume(vol)
            P = N_RAYS * N_SAMPLES
            F = args.feat_dim
            w2c3, k3 = pose["w2cs"][:N_SRC].contiguous(), pose["intrinsics"][:N_SRC].contiguous()
            feat, dirs = ops.gather(vol_cl, src[0], w2c3, k3, pts, ndc, rdir)      # the MLP's inputs of batch 0 (the lookups' own timings: bench_extras.lookup_rooflines)
            packed = net.packed(F)
            lib = _lib.lib()
            st = torch.cuda.current_stream
            raw = torch.empty((N_RAYS, N_SAMPLES, 4), device=dev)
            # raw C-ABI call with pre-allocated outputs
            k_mlp = lambda: lib.mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N_RAYS, N_SAMPLES, 0,
                                                raw.data_ptr(), st().cuda_stream)
            t_mlp = event_time(k_mlp, 60)
            k_mlp_cen = lambda cen: lib.mvsnerf_mlp_fwd_census(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N_RAYS, N_SAMPLES, 0,
                                                               raw.data_ptr(), cen.data_ptr(), st().cuda_stream)
            for _ in range(50):
                k_mlp()                                    # the census launch runs in the steady state of the same kernel
            clock = sustained_clock_ghz(k_mlp_cen, (P + 127) // 128, dev)
        tf = FLOP_PER_SAMPLE * P / (t_mlp * 1e-3) / 1e12
        roof = {"kernel": "mlp_fwd_pipe_kernel", "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic("mlp_fwd_pipe_kernel"),
                "traffic_source": _pmc_summary()[1],
                "avg_launch_ms": round(t_mlp, 4),
                "mfma_pipe_busy_frac_pmc": pmc_mfma_busy_frac("mlp_fwd_pipe_kernel"),
                "s_memtime_ghz": round(clock, 3),
                # the datasheet peak assumes 2.4 GHz; under this kernel the chip sustains s_memtime_ghz (power-limited DVFS): the fp32-MFMA
                # rate at THAT clock is 64 FLOP/clk/SIMD x 1024 SIMDs x clock - what the matrix pipes could deliver in this launch at most
                "peak_at_sustained_clock": round(64 * 1024 * clock / 1e3, 1),
                "frac_of_sustained_clock_peak": round(tf / (64 * 1024 * clock / 1e3), 4)}
        import bench_extras
        if a.mlp_precision != "fp32":         # an opt-in arithmetic was timed: `roofline` describes ITS kernel, the fp32 kernel's object moves to `rooflines`
            roofs.append(roof)
            roof = bench_extras.mode_roofline(dict(locals()))
        roofs += bench_extras.lookup_rooflines(dict(locals()))
        # ---------------- CPU baseline: the oracle (torch CPU kernels) on a bounded sample of the same workload, + parity of the timed workload (bench_extras.py)
        if a.cpu_batches > 0 and world == 1:          # reported at N=1 only (bench contract)
            cpu, o = bench_extras.cpu_baseline_leg(dict(locals()))

        extras = {}
        if not a.no_extras and world == 1:
            extras = bench_extras.single_gpu_extras(dict(locals()))
        flat = bench_extras.flat_scalars(encode_ms, extras, multi)     # the numbers beyond the headline as top-level scalars (a driver that keeps only flat keys keeps these)
        print(json.dumps({
            "metric": "rendered rays/sec (1024-ray batch, 128 samples)", "value": round(rays_per_s, 1), "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "rccl_ranks_seen": rccl_ranks_seen, "backend": (dist.get_backend() if world > 1 else None), "per_rank_rays_per_s": per_rank,
            "dtype": {"fp32": "f32", "bf16": "bf16", "bf16x3": "bf16x3 (split-bf16 products, fp32 accumulate)",
                      "bf16x6": "bf16x6 (fp32 emulated by split-bf16 products, fp32 accumulate)",
                      "fp16x3": "fp16x3 (fp32 emulated by split-fp16 products, fp32 accumulate)"}[a.mlp_precision], "data": "synthetic",
            "config": {"workload": "config 2: 3 source views 512x640, 128 depth planes, pad 24 (volume 128x176x208x8), "
                                   f"1024 rays x 128 samples per step, MLP arithmetic {a.mlp_precision}, render-only (volume pre-built)",
                       "weights": "mvsnerf-v0 checkpoint", "volume": volume_src, "rays_per_step_per_gpu": N_RAYS,
                       "parallelism": f"ray-sharded x{world}, no data-path collective",
                       "clock_settle_ms": a.settle_ms},
            "encode_ms": encode_ms, "encode_ms_bf16_conv0": encode_ms_bf16, "encode_ms_fp32_conv0": encode_ms_h3,
            "roofline": roof, "rooflines": roofs, "cpu_baseline": cpu, **flat, "multi_gpu": multi, "extras": extras,
        }))
    if world > 1:
        dist.barrier()                    # rank 0 was alone in the per-kernel section above: nobody tears the group down before it is back
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

