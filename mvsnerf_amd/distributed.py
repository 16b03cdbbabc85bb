"""Multi-GPU plumbing for the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

The path shards by rays: given the 8-channel volume, the source images and the MLP weights
(163 MB, replicated) every ray is independent, so
  * inference is tile-parallel: the row-major chunk index range of `build_rays_test`
    (reference utils.py:95-98, train_mvs_nerf_pl.py:198) is split into contiguous per-rank ranges and the
    per-rank RGB/depth strips are concatenated with ONE all_gather - no collective in the data path;
  * training is ray-sharded DP: all ranks draw the SAME pixel ids (same CPU-RNG seed), each renders its
    slice, and the gradients are averaged with ONE all-reduce over a single flat fp32 buffer (1.9 MB for
    MLP + MVSNet: latency-bound, so one message instead of ~100 per-parameter ones).
"""
import contextlib
import os

import torch
import torch.distributed as dist

_SINGLE = False          # inside `single_rank()`: behave as world size 1 (local full-frame reference renders in bench/tests)
_FORCE_COLLECTIVES = False   # tests on a 1-GPU box: issue the collectives even at world size 1 (exercises RCCL itself)


@contextlib.contextmanager
def single_rank():
    """Everything in this module acts as if no process group existed (each rank does the whole job locally)."""
    global _SINGLE
    prev, _SINGLE = _SINGLE, True
    try:
        yield
    finally:
        _SINGLE = prev


@contextlib.contextmanager
def force_collectives():
    """World-size-1 groups normally skip their collectives; inside this context they are issued (an all-reduce / all-gather
    over one rank is the identity, but it goes through RCCL - what a 1-GPU test box can check)."""
    global _FORCE_COLLECTIVES
    prev, _FORCE_COLLECTIVES = _FORCE_COLLECTIVES, True
    try:
        yield
    finally:
        _FORCE_COLLECTIVES = prev


def world_rank(group=None):
    """(world, rank) of the default (or given) group; (1, 0) without a group or inside single_rank()."""
    if _SINGLE or not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _collective_needed(group=None):
    if _SINGLE or not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _FORCE_COLLECTIVES


# ------------------------------------------------------------------ collectives
# The three collectives of the path.  backend "nccl" (= RCCL over xGMI): straight through, device buffers.  backend "gloo" with DEVICE
# tensors - the shared-GPU dry run (tests/test_gpu_shared.py, `bench.py --shared-gpu-dry-run`: several ranks on ONE GPU, which RCCL
# refuses) - stages through host memory, so that every line of the N-rank code paths (sharding, seeding, padding, gather order) runs
# against the HIP kernels on a 1-GPU box.  Host tensors (the CPU tests) go straight to gloo.
def _stage(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce(t, op=None, group=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _stage(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def all_gather_into_tensor(out, inp, group=None):
    if _stage(inp, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, inp.cpu(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)
    return out


def broadcast(t, src=0, group=None):
    if _stage(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def init_from_env(device=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).
    Returns (rank, world).  No-op for world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n_items, world, rank):
    """Contiguous, balanced split of range(n_items): the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rays(n_rays, world, rank):
    """Slice of a ray batch owned by `rank` (all ranks hold the same seeded batch; indices stay bit-exact)."""
    lo, hi = shard_range(n_rays, world, rank)
    return slice(lo, hi)


def all_gather_rows(local, n_total, group=None):
    """Concatenate per-rank row blocks (rank r owns rows shard_range(n_total, world, r)) on every rank.
    One collective; blocks are padded to the largest shard so that a single all_gather_into_tensor suffices."""
    if not _collective_needed(group):
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, world, r) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * width, *local.shape[1:]), dtype=local.dtype, device=local.device)
    all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * width:r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


class FlatGradAllReduce:
    """Averages the gradients of `params` across ranks with ONE all-reduce on a flat fp32 buffer
    (sum over xGMI, then divide by the world size).  Parameters without a gradient contribute zeros, so every
    rank sends the same message size."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.views = None

    def __call__(self):
        if not _collective_needed(self.group) or not self.params:
            return
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            self.views, off = [], 0
            for p in self.params:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        # gather / scatter with one multi-tensor copy each (78 parameters: 156 single-tensor copies of a few microseconds otherwise)
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(dist.get_world_size(self.group))
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                p.grad = v.clone()
        if have:
            torch._foreach_copy_([g for _, g in have], [v for v, _ in have])


def render_frame(render_chunk, H, W, chunk, group=None):
    """Tile-parallel full-frame render (the chunk loop of validation_step, train_mvs_nerf_pl.py:198-208).
    `render_chunk(idx)` renders chunk `idx` of the row-major pixel order and returns (rgb (n,3), depth (n,)).
    Each rank renders a contiguous range of chunks; one all_gather assembles (H*W,3) and (H*W,) on every rank."""
    n_chunks = (H * W + chunk - 1) // chunk
    world, rank = world_rank(group)
    lo, hi = shard_range(n_chunks, world, rank)
    rgbs, depths = [], []
    for idx in range(lo, hi):
        rgb, depth = render_chunk(idx)
        rgbs.append(rgb)
        depths.append(depth)
    if rgbs:
        packed = torch.cat([torch.cat(rgbs, 0), torch.cat(depths, 0)[:, None]], 1)       # (n_local, 4): one message
    else:
        packed = None
    return _assemble_frame(packed, H, W, chunk, n_chunks, world, group)


def render_frame_pixels(render_range, H, W, chunk, group=None, device=None):
    """Same sharding as render_frame, but the rank's whole contiguous pixel range is rendered by ONE call
    `render_range(first_pixel, n_pixels) -> (rgb (n,3), depth (n,))` (ops.render_pixels: the chunk loop runs inside the
    library, one FFI crossing per rank and frame)."""
    n_chunks = (H * W + chunk - 1) // chunk
    world, rank = world_rank(group)
    lo, hi = shard_range(n_chunks, world, rank)
    first, last = min(lo * chunk, H * W), min(hi * chunk, H * W)
    packed = None
    if last > first:
        rgb, depth = render_range(first, last - first)
        packed = torch.cat([rgb, depth[:, None]], 1)
    return _assemble_frame(packed, H, W, chunk, n_chunks, world, group, device)


def _assemble_frame(packed, H, W, chunk, n_chunks, world, group, device=None):
    if world == 1 and not _collective_needed(group):
        return packed[:, :3], packed[:, 3]
    # rows are pixels; the per-rank pixel ranges follow from the chunk ranges (the last chunk may be short)
    px = [(min(l * chunk, H * W), min(h * chunk, H * W)) for l, h in (shard_range(n_chunks, world, r) for r in range(world))]
    width = max(b - a for a, b in px)
    if packed is not None:
        dev = packed.device
    else:
        dev = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    pad = torch.zeros((width, 4), dtype=torch.float32, device=dev)
    if packed is not None:
        pad[:packed.shape[0]] = packed
    out = torch.empty((world * width, 4), dtype=torch.float32, device=dev)
    all_gather_into_tensor(out, pad, group=group)
    full = torch.cat([out[r * width:r * width + (b - a)] for r, (a, b) in enumerate(px)], 0)
    return full[:, :3], full[:, 3]


# ------------------------------------------------------------------ data-parallel training
def shard_ray_batch(tensors, n_rays, group=None, dim=0):
    """Ray-sharded DP: every rank holds the SAME batch of n_rays rays (same seeded draw); returns this rank's slice of each tensor
    in `tensors` (sliced along `dim`; None entries pass through) and the factor its mean-over-local-rays loss must be multiplied
    with so that the rank-AVERAGED gradient (FlatGradAllReduce) is the gradient of the mean over all n_rays rays:
    sum_r (n_r/N) g_r = (1/world) sum_r (n_r world / N) g_r - which differs from 1 only when n_rays % world != 0."""
    world, rank = world_rank(group)
    if world == 1:
        return list(tensors), 1.0
    sl = shard_rays(n_rays, world, rank)
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
        else:
            idx = [slice(None)] * t.dim()
            idx[dim if t.shape[dim] == n_rays else 1] = sl          # rays_o-style (3, N) tensors carry the rays in dim 1
            out.append(t[tuple(idx)])
    n_local = sl.stop - sl.start
    return out, n_local * world / float(n_rays)


def common_seed(device=None, group=None):
    """One int64 drawn on rank 0 and broadcast: ray-sharded DP needs the same pixel ids (CPU RNG, reference utils.py:93) and the
    same stratified jitter (device RNG, utils.py:220) on every rank.  Seeding both generators with a broadcast value per step makes
    that true by construction instead of by an unchecked 'all ranks were seeded alike' assumption."""
    seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    if _collective_needed(group):
        t = seed.to(device) if device is not None else seed
        broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        seed = t.cpu()
    return int(seed.item())


def scene_shard(items, group=None):
    """Scene-sharded DP (what Lightning DDP's DistributedSampler gives the reference, train_mvs_nerf_pl.py:306,313): rank r takes
    items r, r+world, r+2 world, ... of the sample list; all ranks take the same NUMBER of steps (the tail is dropped)."""
    world, rank = world_rank(group)
    n = len(items) // world * world
    return [items[i] for i in range(rank, n, world)]
