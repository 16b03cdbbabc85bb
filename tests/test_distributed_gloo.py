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


# ------------------------------------------------------------------ data-parallel training modes (train.MVSSystem.fit_steps)
class _ToySystem:
    """A stand-in with the attributes MVSSystem.fit_steps uses; the 'renderer' is a Linear so the step runs on CPU.
    The sharding / seeding / all-reduce code under test is the real one (mvsnerf_amd.train, mvsnerf_amd.distributed)."""

    def __new__(cls, mode):
        import types
        from mvsnerf_amd import train as T

        class Toy(T._ModuleShim):
            fit_steps = T.MVSSystem.fit_steps
            dp_mode = T.MVSSystem.dp_mode
            sync_buffers = T.MVSSystem.sync_buffers

            def __init__(self):
                super().__init__()
                torch.manual_seed(5)
                self.lin = torch.nn.Linear(3, 2)
                self.MVSNet = torch.nn.BatchNorm1d(2)                          # stands for the encoder's InPlaceABN running statistics
                self.args = types.SimpleNamespace(dp_mode=mode)
                self.grad_vars = list(self.lin.parameters())
                self._allreduce = None
                self.draws = []

            def training_step(self, batch, nb):
                x, y = batch["x"], batch["y"]                                  # (N,3), (N,2): N "rays"
                n = x.shape[0]
                self.draws.append(torch.rand(4))                               # stands for pixel ids / jitter
                with torch.no_grad():
                    self.MVSNet.running_mean += y.mean(0)                      # per-rank statistics of the scene this rank saw
                scale = 1.0
                if self.dp_mode() == "ray":
                    (x, y), scale = D.shard_ray_batch((x, y), n)
                return {"loss": ((self.lin(x) - y) ** 2).mean() * scale}
        return Toy()


def _toy_batches(k):
    g = torch.Generator().manual_seed(11)
    return [{"x": torch.randn(7, 3, generator=g), "y": torch.randn(7, 2, generator=g)} for _ in range(k)]


def _dp_worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env(device="cpu")
    torch.manual_seed(100 + rank)                  # deliberately DIFFERENT RNG states: ray mode must still draw alike
    sys_ = _ToySystem(mode)
    opt = torch.optim.SGD(sys_.grad_vars, lr=1.0)
    w0 = sys_.lin.weight.detach().clone()
    sys_.fit_steps(_toy_batches(2), opt)
    q.put((rank, (sys_.lin.weight.detach() - w0).tolist(), sys_.lin.bias.detach().tolist(), torch.stack(sys_.draws).tolist(),
           sys_.MVSNet.running_mean.tolist()))   # plain lists: no fd passing
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ray", "scene"])
def test_dp_modes_world2(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, dw0, b0, draws0, rm0), (_, dw1, b1, draws1, rm1) = [(r, *(torch.tensor(t) for t in rest)) for r, *rest in res]
    assert torch.equal(dw0, dw1) and torch.equal(b0, b1)            # ranks stay in lock step
    assert torch.equal(rm0, rm1)                                    # ... and so do the running statistics (ray: same data; scene: rank 0's, broadcast)
    batches = _toy_batches(2)
    if mode == "ray":
        # 7 rays over 2 ranks (4 + 3): the all-reduced step must equal the single-process step on the full batches
        assert torch.equal(draws0, draws1)                           # same "pixel ids" on both ranks (broadcast seed)
        ref = _ToySystem("ray")
        opt = torch.optim.SGD(ref.grad_vars, lr=1.0)
        w0 = ref.lin.weight.detach().clone()
        ref.fit_steps(batches, opt)
        assert torch.allclose(ref.lin.weight.detach() - w0, dw0, atol=1e-6)
        assert torch.allclose(ref.lin.bias.detach(), b0, atol=1e-6)
    else:
        # one step, two scenes: the N-rank step is the average of the two single-rank steps (SGD lr 1: delta = -mean grad)
        assert not torch.equal(draws0, draws1)                       # independent draws per rank
        deltas = []
        for r in range(2):
            ref = _ToySystem("scene")
            w0 = ref.lin.weight.detach().clone()
            ref.fit_steps([batches[r]], torch.optim.SGD(ref.grad_vars, lr=1.0))
            deltas.append(ref.lin.weight.detach() - w0)
        assert torch.allclose(0.5 * (deltas[0] + deltas[1]), dw0, atol=1e-6)


def test_scene_shard_and_loss_scale_single_process():
    assert D.scene_shard([1, 2, 3]) == [1, 2, 3]
    (a,), scale = D.shard_ray_batch((torch.arange(5),), 5)
    assert scale == 1.0 and a.numel() == 5


# ------------------------------------------------------------------ fine-tune DP: volume gradient from exchanged sample gradients
def _cpu_scatter(gvol_cl, ndc, g):
    """Nearest-voxel stand-in for the trilinear scatter kernel (the exchange logic under test does not care which scatter runs)."""
    D, H, W, C = gvol_cl.shape
    idx = ((ndc[:, 2] * (D - 1)).round().long().clamp(0, D - 1) * H + (ndc[:, 1] * (H - 1)).round().long().clamp(0, H - 1)) * W \
        + (ndc[:, 0] * (W - 1)).round().long().clamp(0, W - 1)
    gvol_cl.view(-1, C).index_add_(0, idx, g)


def _volgrad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init_from_env(device="cpu")
    from mvsnerf_amd import ops
    g = torch.Generator().manual_seed(50 + rank)
    n = 37 if rank == 0 else 29                               # uneven shards
    d_feat, ndc = torch.randn(n, 8, generator=g), torch.rand(n, 3, generator=g)
    gvol = torch.zeros(4, 5, 6, 8)
    ops.volume_grad_from_all_ranks(d_feat, ndc, gvol, scatter=_cpu_scatter)
    q.put((rank, gvol.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_volume_grad_from_all_ranks_world2():
    """Every rank ends with (1/world) * scatter of ALL ranks' sample gradients - without a volume-sized collective."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_volgrad_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    expect = torch.zeros(4, 5, 6, 8)
    for rank, n in ((0, 37), (1, 29)):
        g = torch.Generator().manual_seed(50 + rank)
        d_feat, ndc = torch.randn(n, 8, generator=g), torch.rand(n, 3, generator=g)
        _cpu_scatter(expect, ndc, d_feat)
    expect /= 2
    for rank, gv in res:
        assert torch.allclose(torch.tensor(gv), expect, atol=1e-6), f"rank {rank}"
