"""World-size-2 tests of the multi-GPU plumbing on CPU (gloo): shard maps, tile-parallel frame assembly, and the
flat-buffer gradient all-reduce.  The kernels themselves are exercised by the -m gpu tests; here the render
function is a deterministic stand-in so that N-rank results can be compared with the 1-rank result exactly."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvsnerf_amd import distributed as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _fake_render(H, W, chunk):
    def render_chunk(idx):
        ids = torch.arange(idx * chunk, min((idx + 1) * chunk, H * W), dtype=torch.float32)
        return torch.stack([ids, ids * 2, ids * 3], -1), ids * 0.5          # rgb, depth as functions of the pixel id
    return render_chunk


def _worker(rank, world, port, H, W, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = D.init_from_env(device="cpu")
    assert (r, w) == (rank, world)
    # 1. tile-parallel frame: every rank ends with the full frame, identical to the single-process render
    rgb, depth = D.render_frame(_fake_render(H, W, chunk), H, W, chunk)
    ids = torch.arange(H * W, dtype=torch.float32)
    ok_frame = torch.equal(rgb, torch.stack([ids, ids * 2, ids * 3], -1)) and torch.equal(depth, ids * 0.5)
    # 1b. the same frame with one pixel-range call per rank (what MVSSystem.render_view issues: mvsnerf_render_pixels_fwd)
    def render_range(first, n):
        p = torch.arange(first, first + n, dtype=torch.float32)
        return torch.stack([p, p * 2, p * 3], -1), p * 0.5
    rgb2, depth2 = D.render_frame_pixels(render_range, H, W, chunk, device=torch.device("cpu"))
    ok_frame = ok_frame and torch.equal(rgb2, rgb) and torch.equal(depth2, depth)
    # 2. row gather of a ray-sharded batch
    n = 1001
    sl = D.shard_rays(n, world, rank)
    full = D.all_gather_rows(torch.arange(n, dtype=torch.float32)[sl, None].repeat(1, 3), n)
    ok_rows = torch.equal(full[:, 0], torch.arange(n, dtype=torch.float32))
    # 3. flat gradient all-reduce == mean over ranks; parameters without grad are handled
    torch.manual_seed(0)
    lin = torch.nn.Linear(7, 5)
    extra = torch.nn.Parameter(torch.zeros(3))
    x = torch.full((4, 7), float(rank + 1))
    lin(x).sum().backward()
    D.FlatGradAllReduce(list(lin.parameters()) + [extra])()
    expect_w = torch.full((5, 7), 4.0 * (1 + 2) / 2) if world == 2 else None
    ok_grad = torch.allclose(lin.weight.grad, expect_w) and torch.allclose(lin.bias.grad, torch.full((5,), 4.0)) \
        and torch.equal(extra.grad, torch.zeros(3))
    q.put((rank, ok_frame, ok_rows, ok_grad))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("H,W,chunk", [(6, 10, 7), (4, 4, 64), (9, 13, 5)])
def test_world2_gloo(H, W, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, H, W, chunk, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok_frame, ok_rows, ok_grad in res:
        assert ok_frame, f"rank {rank}: frame mismatch"
        assert ok_rows, f"rank {rank}: row gather mismatch"
        assert ok_grad, f"rank {rank}: gradient all-reduce mismatch"


def test_shard_range_partitions():
    for n in (0, 1, 7, 320, 1001):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_paths():
    rgb, depth = D.render_frame(_fake_render(5, 6, 4), 5, 6, 4)
    assert rgb.shape == (30, 3) and depth.shape == (30,)
    rgb2, depth2 = D.render_frame_pixels(lambda f, n: (torch.zeros(n, 3) + f, torch.zeros(n)), 5, 6, 4)
    assert rgb2.shape == (30, 3) and depth2.shape == (30,)
    t = torch.ones(4, 2)
    assert D.all_gather_rows(t, 4) is t
