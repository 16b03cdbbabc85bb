"""torch.optim.Adam (the reference's optimizer, train_mvs_nerf_pl.py:84-88 / train_mvs_nerf_finetuning_pl.py:84-87) with its update on ONE HIP
launch per 84 tensors (csrc/adam.hip) instead of torch's three ~25 us multi-tensor launches for the 78 small tensors of the generalizable step.
A subclass: same constructor, same `param_groups` (LR schedulers act on it), same `state_dict()` keys ('step', 'exp_avg', 'exp_avg_sq'), same
pre / post step hooks.  Supports what the reference uses - fp32 CUDA parameters, no weight decay, no amsgrad, no maximize; anything else raises.
Host side: the pointer tables of a group's parameters and moments are built once and reused while the same parameters receive gradients (only the
gradients' addresses change from step to step); the parameters of a group that have taken the same number of steps share ONE host `step` tensor
(0.6 ms -> 0.07 ms of Python per step: the use_amp step is 4.8 ms of GPU work behind ~4 ms of host work).  Parameters whose step counts differ
(a parameter that receives gradients only now and then under zero_grad(set_to_none=True); a loaded torch.optim.Adam state) are bucketed by step
count - one launch per distinct count, each with its own bias corrections - exactly as torch.optim.Adam's per-parameter `step` would give."""
import ctypes
import math

import torch

from . import _lib
from ._lib import check, stream_ptr


class _GroupTables:
    __slots__ = ("ps", "n", "p", "g", "m", "v", "numel", "step")


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}                                  # the moments are new tensors now

    def _build(self, gi, ps):
        """Tables of the parameters `ps` of group gi (those with a gradient this step), one per distinct step count."""
        buckets = {}
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("mvsnerf_amd.optim.Adam: contiguous fp32 CUDA parameters only")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)                   # as torch.optim.Adam keeps it (host tensor)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            buckets.setdefault(float(st["step"]), []).append(p)      # a loaded state may hold it on the device: one read, here only
        tables = []
        for step, bp in buckets.items():
            t = _GroupTables()
            n = t.n = len(bp)
            t.ps = bp
            t.step = torch.tensor(step, dtype=torch.float32)   # ONE host tensor per bucket, shared by the state of every parameter in it; a
            for p in bp:                                       # parameter that leaves the bucket later gets a new tensor with its own count
                self.state[p]["step"] = t.step
            arr = ctypes.c_void_p * n
            t.p = arr(*[p.data_ptr() for p in bp])
            t.m = arr(*[self.state[p]["exp_avg"].data_ptr() for p in bp])
            t.v = arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in bp])
            t.g = arr()
            t.numel = (ctypes.c_int64 * n)(*[p.numel() for p in bp])
            tables.append(t)
        self._tables[gi] = (tuple(id(p) for p in ps), tables)
        return tables

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        f32 = torch.float32
        for gi, group in enumerate(self.param_groups):
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise RuntimeError("mvsnerf_amd.optim.Adam: weight_decay / amsgrad / maximize are not implemented (the reference does not use them)")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            cached = self._tables.get(gi)
            if cached is None or len(cached[0]) != len(ps) or cached[0] != tuple(id(p) for p in ps):
                # the set of parameters with a gradient changed: the ones that sat in a shared bucket keep their count in a tensor of their own
                if cached is not None:
                    for t in cached[1]:
                        for p in t.ps:
                            self.state[p]["step"] = t.step.clone()
                tables = self._build(gi, ps)
            else:
                tables = cached[1]
            beta1, beta2 = group["betas"]
            for t in tables:
                keep = None
                garr = t.g
                for i, p in enumerate(t.ps):
                    g = p.grad
                    if g.dtype is not f32 or g.is_sparse:
                        raise RuntimeError("mvsnerf_amd.optim.Adam: dense fp32 gradients only")
                    if not g.is_contiguous():
                        g = g.contiguous()
                        keep = (keep or []) + [g]              # must outlive the launch
                    garr[i] = g.data_ptr()
                t.step += 1
                step = int(t.step)
                check(lib.mvsnerf_adam_step_multi(t.n, t.p, garr, t.m, t.v, t.numel, float(group["lr"]) / (1.0 - beta1 ** step), float(beta1), float(beta2),
                                                  float(group["eps"]), math.sqrt(1.0 - beta2 ** step), stream_ptr()), "adam_step_multi")
                del keep
        # the launch writes parameter memory without touching tensor._version: tell the packed-weight caches (models.MVSNeRF.packed*, encoder
        # _PackedConv*) directly, so their correctness does not rest on torch's global optimizer post-hook alone
        _lib._bump_weights_epoch()
        return loss
