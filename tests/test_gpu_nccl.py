"""RCCL (backend "nccl") legs of the multi-GPU path on real hardware.

* world size 1 on any box: the collectives are issued anyway (distributed.force_collectives), so the all_gather of the tile-parallel
  frame and the flat-buffer gradient all-reduce go through RCCL on this GPU; results must equal the no-process-group results.
* world size 2 when two GPUs are visible (skipped otherwise): N-rank frame == 1-rank frame bit for bit; parameters stay in sync
  through ray-sharded and scene-sharded training steps.
* bench.py --gpus N must refuse to report an N-GPU number from fewer ranks.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, PAD, S, NR = 64, 96, 4, 16, 256


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _system(dev, mode="ray"):
    from mvsnerf_amd import train
    from tests.util import load_weights
    mlp_sd, mvs_sd = load_weights()
    args = train.default_args(pad=PAD, batch_size=NR, N_samples=S, chunk=512, dp_mode=mode)
    sysm = train.MVSSystem(args, n_depth_planes=16).to(dev)
    sysm.network_fn.load_state_dict(mlp_sd)
    sysm.MVSNet.load_state_dict(mvs_sd)
    return sysm


def _run_rank(rank, world, port, q):
    """Body of one rank (also used in-process for world 1)."""
    import torch.distributed as dist
    from mvsnerf_amd import distributed as D, train
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    try:
        assert dist.get_world_size() == world
        res = {}
        with D.force_collectives():
            batch = train.synthetic_batch(H, W, seed=7, rot_deg=2.0)
            sysm = _system(dev)
            rgb, depth = sysm.render_view(batch, batch_rays=256)
            with D.single_rank():
                rgb1, depth1 = sysm.render_view(batch, batch_rays=256)
            res["frame_equal"] = bool(torch.equal(rgb, rgb1) and torch.equal(depth, depth1))
            res["frame_finite"] = bool(torch.isfinite(rgb).all())
            for mode in ("ray", "scene"):
                sm = _system(dev, mode)
                torch.manual_seed(3)
                batches = [train.synthetic_batch(H, W, seed=20 + j, rot_deg=2.0) for j in range(2 * world)]
                losses = sm.fit_steps(batches if mode == "scene" else batches[:2])
                chk = torch.stack([p.detach().double().sum() for p in sm.grad_vars]).sum().reshape(1)
                lo, hi = chk.clone(), chk.clone()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                res[f"{mode}_in_sync"] = bool((lo == hi).item())
                res[f"{mode}_finite"] = all(l == l and abs(l) < 1e6 for l in losses)
                res[f"{mode}_chk"] = float(chk.item())
        if q is not None:
            q.put((rank, res))
        return res
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_world1_nccl_collectives_match_no_group():
    from mvsnerf_amd import train
    dev = torch.device("cuda", 0)
    # reference without any process group
    ref = {}
    for mode in ("ray", "scene"):
        sm = _system(dev, mode)
        torch.manual_seed(3)
        batches = [train.synthetic_batch(H, W, seed=20 + j, rot_deg=2.0) for j in range(2)]
        sm.fit_steps(batches)
        ref[mode] = float(torch.stack([p.detach().double().sum() for p in sm.grad_vars]).sum())
    res = _run_rank(0, 1, _free_port(), None)
    assert res["frame_equal"] and res["frame_finite"]
    for mode in ("ray", "scene"):
        assert res[f"{mode}_in_sync"] and res[f"{mode}_finite"]
    # scene mode at world 1 re-seeds with +0 and takes the same batches: same parameters up to the backward's float atomics
    assert abs(res["scene_chk"] - ref["scene"]) < 1e-3 * max(1.0, abs(ref["scene"]))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_world2_nccl():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_rank, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=600) for _ in procs]
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, r in res:
        assert r["frame_equal"], f"rank {rank}: tile-parallel frame != single-rank frame"
        assert r["ray_in_sync"] and r["scene_in_sync"], f"rank {rank}: parameters diverged"
        assert r["ray_finite"] and r["scene_finite"]


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "refusing" in (p.stderr + p.stdout)
    assert '"n_gpus"' not in p.stdout          # no JSON line at all


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_finetune_volume_grad_by_sample_exchange_world1():
    """Fine-tune DP (args.dp_volume_grad = "samples"): the volume gradient is rebuilt on every rank from the all_gathered per-sample
    feature gradients instead of all-reducing the 150-246 MB tensor.  World size 1 with the collectives forced through RCCL: the
    gradients must equal the plain (no process group) ones up to the atomics' summation order, and the periodic re-broadcast of the
    volume must run."""
    import torch.distributed as dist
    from mvsnerf_amd import distributed as D, train
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from oracle import mvsnerf_oracle as O
    from tests.util import load_weights
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rig = make_rig(64, 96, seed=8, smooth=True)
    pose = pose_ref_of(rig)
    src = (rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0], {k: v[:3] for k, v in pose.items()})
    mlp_sd, mvs_sd = load_weights()
    g = torch.Generator().manual_seed(0)
    ro, rd, pix = O.get_rays_mvs(64, 96, pose["intrinsics"][3], pose["c2ws"][3], 256, generator=g)
    rays = torch.cat([ro.expand(256, 3), rd, torch.full((256, 1), 2.125), torch.full((256, 1), 4.525)], 1)
    tgt = rig["images_raw"][0, 3][:, pix[0].long(), pix[1].long()].permute(1, 0)
    batch = {"rays": rays[None], "rgbs": tgt[None]}

    def grads(mode, forced):
        args = train.default_args(pad=4, batch_size=256, N_samples=32, dp_volume_grad=mode, dp_volume_resync=2)
        ft = train.MVSSystemFinetune(args, src, n_depth_planes=16).to(dev)
        ft.network_fn.load_state_dict(mlp_sd)
        ft.MVSNet.load_state_dict(mvs_sd)
        with torch.no_grad():                                   # same starting volume in every run
            ft.volume.feat_volume.copy_(torch.randn(ft.volume.feat_volume.shape, generator=torch.Generator().manual_seed(1)).to(dev))
        torch.manual_seed(4)
        if forced:
            with D.force_collectives():
                out = ft.training_step(batch, 0)
                out["loss"].backward()
                ft._allreduce()
                gv = ft.volume.feat_volume.grad.clone()
                ft.zero_grad(set_to_none=True)
                torch.manual_seed(4)
                losses = ft.fit_steps([batch] * 4)               # includes two re-broadcasts of the volume (resync = 2)
        else:
            out = ft.training_step(batch, 0)
            out["loss"].backward()
            gv = ft.volume.feat_volume.grad.clone()
            losses = None
        return gv, losses

    ref, _ = grads("allreduce", False)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        got, losses = grads("samples", True)
    finally:
        dist.barrier()
        dist.destroy_process_group()
    assert float(ref.abs().max()) > 0
    assert float((got - ref).abs().max()) < 1e-5 * float(ref.abs().max())
    assert losses is not None and all(l == l for l in losses) and losses[-1] < losses[0]
