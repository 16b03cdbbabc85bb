"""World size 2 on a ONE-GPU box: two ranks share cuda:0, the collectives run over gloo (staged through host memory by
mvsnerf_amd.distributed - RCCL refuses two ranks on one device), the compute is the HIP kernels.  Every line of the N-rank code paths
(reference wiring: train_mvs_nerf_pl.py:306,313 `gpus=args.num_gpus, accelerator='ddp'`) executes here before a multi-GPU node ever
sees it:
  * tile-parallel frame: the 2-rank frame equals the 1-rank frame bit for bit;
  * ray-sharded DP: the all-reduced gradients of a step equal the 1-rank gradients of the whole batch (to the float atomics' tolerance);
  * scene-sharded DP (the default): the all-reduced gradients equal the mean of the two scenes' 1-rank gradients; `fit_steps` keeps the
    ranks' parameters identical in both modes;
  * fine-tuning: the volume gradient rebuilt from exchanged sample gradients equals the 1-rank gradient;
  * `bench.py --gpus 2 --shared-gpu-dry-run` runs the torch.distributed.run re-exec and `multi_gpu_legs` end to end and prints no headline.
Documented tolerance of "N-rank == 1-rank" for gradients: 1e-4 of the tensor's largest entry (the scatter kernels of the backward use
float atomics: the summation order varies from run to run - DESIGN.md)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, PAD, S, NR = 64, 96, 4, 16, 256
GRAD_TOL = 1e-4
AMP_PER_TENSOR_TOL = 0.05  # use_amp: the same distance for the worst single tensor of the 78 (measured 1.0e-2, tensor 23 - a FeatureNet weight with a small gradient norm; <= 5x)
AMP_L2_TOL = 5e-2          # use_amp: relative L2 distance of the whole gradient vector (see _rank_body); a wrong shard or a missing all-reduce is O(1)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _system(dev, mode):
    from mvsnerf_amd import train
    from tests.util import load_weights
    mlp_sd, mvs_sd = load_weights()
    args = train.default_args(pad=PAD, batch_size=NR, N_samples=S, chunk=512, dp_mode=mode)
    sysm = train.MVSSystem(args, n_depth_planes=16).to(dev)
    sysm.network_fn.load_state_dict(mlp_sd)
    sysm.MVSNet.load_state_dict(mvs_sd)
    return sysm


def _grads_of(sysm, batch, seed, allreduce):
    from mvsnerf_amd import distributed as D
    if sysm._allreduce is None:
        sysm._allreduce = D.FlatGradAllReduce(sysm.grad_vars)
    sysm.zero_grad(set_to_none=True)
    torch.manual_seed(seed)
    out = sysm.training_step(batch, 0)
    out["loss"].backward()
    if allreduce:
        sysm._allreduce()
    return [None if p.grad is None else p.grad.detach().clone() for p in sysm.grad_vars]


def _worst(got, ref):
    w = 0.0
    for g, r in zip(got, ref):
        if g is None or r is None:
            assert g is None and r is None
            continue
        w = max(w, float((g - r).abs().max()) / max(float(r.abs().max()), 1e-20))
    return w


def _rank_body(rank, world, port, q):
    import torch.distributed as dist
    from mvsnerf_amd import distributed as D, encoder, train
    encoder.PSW_BWD_DETERMINISTIC = True          # the order-independent plane-sweep reduction: one source of run-to-run noise less in the comparisons
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {}
    try:
        assert D.world_rank() == (world, rank)
        # ---- tile-parallel frame
        batch = train.batch_to_device(train.synthetic_batch(H, W, seed=7, rot_deg=2.0), dev)
        sysm = _system(dev, "ray")
        from mvsnerf_amd import ops
        # Up to three attempts, every mismatch recorded.  History: until round 4 about one scene encode in sixty differed when processes shared the GPU - the
        # plane sweep's packed fp32 arithmetic went wrong next to the other rank's 16-bit MFMA waves (csrc/planesweep.hip, profiles/r04_pk_mfma_hazard.txt,
        # tests/test_gpu_costream.py); the sweep is compiled without packed fp32 instructions since, and 1152 shared-GPU encodes reproduced bit for bit.  The
        # repetition stays as a REPORTER: should another kernel turn out to have the same weakness, the event is printed instead of hidden.
        res["frame_attempts"] = []
        for attempt in range(3):
            fb0 = ops.guard_fallbacks()
            rgb, depth = sysm.render_view(batch, batch_rays=256)
            fb1 = ops.guard_fallbacks()
            with D.single_rank():
                rgb1, depth1 = sysm.render_view(batch, batch_rays=256)
            fb2 = ops.guard_fallbacks()
            eq = bool(torch.equal(rgb, rgb1) and torch.equal(depth, depth1))
            res["frame_attempts"].append({"equal": eq, "max_diff": (float((rgb - rgb1).abs().max()), float((depth - depth1).abs().max())),
                                          "diff_pixels": int(((rgb - rgb1).abs().amax(-1) > 0).sum()), "guard_fallbacks": (fb1 - fb0, fb2 - fb1)})
            eq_all = torch.tensor([1.0 if eq else 0.0], device=dev)
            D.all_reduce(eq_all)                               # both ranks repeat together (render_view carries a collective)
            if float(eq_all) == world:
                break
        res["frame_equal"] = res["frame_attempts"][-1]["equal"]
        # ---- ray-sharded DP: same draw on both ranks, each renders its half; all-reduced gradients == 1-rank gradients of the whole batch
        g2 = _grads_of(sysm, batch, 11, True)
        with D.single_rank():
            g1 = _grads_of(sysm, batch, 11, False)
        res["ray_grad_err"] = _worst(g2, g1)
        # ---- the same under args.use_amp = BASELINE config 3 as worded ("bf16, ray-sharded DP"): MLP, conv0 .. conv11 and FeatureNet on the bf16 matrix
        # cores; every rank rounds the same operands the same way, so N-rank == 1-rank holds to the fp32 summation-order tolerance here too
        sysm.args.use_amp = True
        try:
            g2b = _grads_of(sysm, batch, 11, True)
            with D.single_rank():
                g1b = _grads_of(sysm, batch, 11, False)
        finally:
            sysm.args.use_amp = False
        # bf16 arithmetic is chaotic in the last bits of its inputs: the two ranks' partial volume gradients are summed in another order than one rank's
        # atomics, a last-bit difference there occasionally flips a rounding to bf16 (2^-8 of that operand) and the flip grows through the 14 layers of the
        # encoder backward (profiles/r05_costream_ab.txt: 2.5e-3 on FeatureNet's gradients from ONE process re-running the same step) - so under use_amp
        # "N ranks == 1 rank" is asked of the gradient VECTOR (relative L2 distance over all parameters), not of every tensor's maximum
        num = sum(float(((a - b).double() ** 2).sum()) for a, b in zip(g2b, g1b) if a is not None)
        den = sum(float((b.double() ** 2).sum()) for b in g1b if b is not None)
        res["ray_amp_grad_err"] = (num / den) ** 0.5
        res["ray_amp_grad_err_max_norm"] = _worst(g2b, g1b)
        # ... and, so that a regression is LOCALISED, the same distance per tensor: the worst of the 78 relative L2 distances and which tensor
        # it is (tensors whose 1-rank gradient is numerically zero are compared in absolute terms against the largest gradient norm)
        gmax = max(float(b.double().norm()) for b in g1b if b is not None)
        per = []
        for i, (a, b) in enumerate(zip(g2b, g1b)):
            if a is None:
                continue
            nb = float(b.double().norm())
            per.append((float((a - b).double().norm()) / max(nb, 1e-6 * gmax), i, nb))
        worst = max(per)
        res["ray_amp_grad_err_per_tensor_worst"] = worst[0]
        res["ray_amp_grad_err_per_tensor_worst_index_norm"] = (worst[1], worst[2])
        res["ray_amp_differs_from_fp32"] = _worst(g1b, g1) > 1e-4         # the bf16 kernels really ran
        # ---- scene-sharded DP: rank r renders scene r with its own draw; all-reduced gradients == mean of the two 1-rank gradients
        scenes = [train.batch_to_device(train.synthetic_batch(H, W, seed=20 + j, rot_deg=2.0), dev) for j in range(world)]
        sm = _system(dev, "scene")
        gs = _grads_of(sm, scenes[rank], 100 + rank, True)
        with D.single_rank():
            per = [_grads_of(sm, scenes[j], 100 + j, False) for j in range(world)]
        mean = [None if a is None else sum(x[i] for x in per) / world for i, a in enumerate(per[0])]
        res["scene_grad_err"] = _worst(gs, mean)
        # ---- fit_steps in both modes: ranks stay in sync, losses finite
        for mode in ("ray", "scene"):
            sf = _system(dev, mode)
            torch.manual_seed(3)
            bl = [train.batch_to_device(train.synthetic_batch(H, W, seed=40 + j, rot_deg=2.0), dev) for j in range(2 * world)]
            losses = sf.fit_steps(bl if mode == "scene" else bl[:2])
            chk = torch.stack([p.detach().double().sum() for p in sf.grad_vars]).sum().reshape(1).cpu()
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            res[f"{mode}_in_sync"] = bool((lo == hi).item())
            res[f"{mode}_finite"] = all(l == l and abs(l) < 1e6 for l in losses)
        # ---- fine-tuning: volume gradient by sample exchange (ops.volume_grad_from_all_ranks)
        from mvsnerf_amd.synth import make_rig, pose_ref_of
        from oracle import mvsnerf_oracle as O
        from tests.util import load_weights
        rig = make_rig(64, 96, seed=8, smooth=True)
        pose = pose_ref_of(rig)
        src = (rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0], {k: v[:3] for k, v in pose.items()})
        mlp_sd, mvs_sd = load_weights()
        g = torch.Generator().manual_seed(0)
        ro, rd, pix = O.get_rays_mvs(64, 96, pose["intrinsics"][3], pose["c2ws"][3], 256, generator=g)
        rays = torch.cat([ro.expand(256, 3), rd, torch.full((256, 1), 2.125), torch.full((256, 1), 4.525)], 1)
        tgt = rig["images_raw"][0, 3][:, pix[0].long(), pix[1].long()].permute(1, 0)
        fb = {"rays": rays[None], "rgbs": tgt[None]}

        def vol_grad(mode, reduce):
            args = train.default_args(pad=4, batch_size=256, N_samples=32, dp_volume_grad=mode, dp_volume_resync=2)
            ft = train.MVSSystemFinetune(args, src, n_depth_planes=16).to(dev)
            ft.network_fn.load_state_dict(mlp_sd)
            ft.MVSNet.load_state_dict(mvs_sd)
            with torch.no_grad():
                ft.volume.feat_volume.copy_(torch.randn(ft.volume.feat_volume.shape, generator=torch.Generator().manual_seed(1)).to(dev))
            torch.manual_seed(4)
            out = ft.training_step(fb, 0)
            out["loss"].backward()
            if reduce:
                if ft._allreduce is None:
                    ft._allreduce = D.FlatGradAllReduce(ft.grad_vars)
                ft._allreduce()
            gv = ft.volume.feat_volume.grad.detach().clone()
            losses = None
            if reduce:
                ft.zero_grad(set_to_none=True)
                torch.manual_seed(4)
                losses = ft.fit_steps([fb] * 4)            # includes two re-broadcasts of the volume (resync = 2)
            return gv, losses

        got, losses = vol_grad("samples", True)
        with D.single_rank():
            ref, _ = vol_grad("allreduce", False)
        res["finetune_grad_err"] = float((got - ref).abs().max()) / float(ref.abs().max())
        res["finetune_ok"] = bool(losses is not None and all(l == l for l in losses) and losses[-1] < losses[0])
        q.put((rank, res))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_body, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=900) for _ in procs]
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    from tests.util import record_err
    for rank, r in res:
        print(f"rank {rank}: {r}")
        assert r["frame_equal"], f"rank {rank}: tile-parallel frame != single-rank frame in three attempts: {r['frame_attempts']}"
        if len(r["frame_attempts"]) > 1:
            print(f"rank {rank}: NOTE - {len(r['frame_attempts']) - 1} frame comparison(s) had to be repeated: {r['frame_attempts'][:-1]}")
            record_err(f"shared_gpu:frame_retries:rank{rank}", float(len(r["frame_attempts"]) - 1), tol=2.0)
        assert r["ray_amp_differs_from_fp32"], "use_amp gradients equal the fp32 ones: the bf16 kernels did not run"
        record_err(f"shared_gpu:ray_amp_grad_err:rank{rank}", r["ray_amp_grad_err"], tol=AMP_L2_TOL)
        record_err(f"shared_gpu:ray_amp_grad_err_max_norm:rank{rank}", r["ray_amp_grad_err_max_norm"], tol=1.0)
        assert r["ray_amp_grad_err"] < AMP_L2_TOL, f"rank {rank}: use_amp ray-DP gradients, relative L2 distance to the 1-rank gradients {r['ray_amp_grad_err']}"
        record_err(f"shared_gpu:ray_amp_grad_err_per_tensor_worst:rank{rank}", r["ray_amp_grad_err_per_tensor_worst"], tol=AMP_PER_TENSOR_TOL)
        assert r["ray_amp_grad_err_per_tensor_worst"] < AMP_PER_TENSOR_TOL, (f"rank {rank}: use_amp ray-DP gradients, worst per-tensor relative L2 distance "
                                                                             f"{r['ray_amp_grad_err_per_tensor_worst']} at (tensor index, its norm) {r['ray_amp_grad_err_per_tensor_worst_index_norm']}")
        for k in ("ray_grad_err", "scene_grad_err", "finetune_grad_err"):
            record_err(f"shared_gpu:{k}:rank{rank}", r[k], tol=GRAD_TOL)
            assert r[k] < GRAD_TOL, f"rank {rank}: {k} = {r[k]}"
        assert r["ray_in_sync"] and r["scene_in_sync"], f"rank {rank}: parameters diverged"
        assert r["ray_finite"] and r["scene_finite"] and r["finetune_ok"]


def test_bench_shared_gpu_dry_run():
    """`bench.py --gpus 2 --shared-gpu-dry-run`: the self-launch through torch.distributed.run (the driver's command line), process-group
    setup, multi_gpu_legs (frame + four DP training variants) - and no headline: the line carries neither `metric` nor `value`."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu-dry-run"],
                       capture_output=True, text=True, env=env, timeout=900)
    if p.returncode != 0:                                   # the assertion message is truncated by pytest: keep the whole text where a gpurun call merges it back
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "shared_dry_run_failure.txt"), "w") as f:
            f.write(p.stdout + "\n==== stderr ====\n" + p.stderr)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["shared_gpu_dry_run"] is True and d["n_ranks"] == 2 and "metric" not in d and "value" not in d
    m = d["multi_gpu"]
    assert m["world_size"] == 2 and m["measured_on_hardware"] is False
    assert m["frame_tile_parallel"]["equals_single_rank_frame"] is True
    assert m["frame_tile_parallel_config5"]["equals_single_rank_frame"] is True and m["frame_tile_parallel_config5"]["n_ranks"] == 2
    for k in ("train_step_dp_scene", "train_step_dp_scene_bf16", "train_step_dp_ray", "train_step_dp_ray_bf16"):
        assert m[k]["params_in_sync"] is True and m[k]["n_ranks"] == 2
