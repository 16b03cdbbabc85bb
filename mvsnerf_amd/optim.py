"""torch.optim.Adam (the reference's optimizer, train_mvs_nerf_pl.py:84-88 / train_mvs_nerf_finetuning_pl.py:84-87) with its update on ONE HIP
launch per 84 tensors (csrc/adam.hip) instead of torch's three ~25 us multi-tensor launches for the 78 small tensors of the generalizable step.
A subclass: same constructor, same `param_groups` (LR schedulers act on it), same `state_dict()` keys ('step', 'exp_avg', 'exp_avg_sq'), same
pre / post step hooks.  Supports what the reference uses - fp32 CUDA parameters, no weight decay, no amsgrad, no maximize; anything else raises."""
import ctypes
import math

import torch

from . import _lib
from ._lib import check, stream_ptr


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.lib()
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise RuntimeError("mvsnerf_amd.optim.Adam: weight_decay / amsgrad / maximize are not implemented (the reference does not use them)")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            keep = []                                  # tensors that must outlive the launch
            ptr = {"p": [], "g": [], "m": [], "v": []}
            numel = []
            step = None
            for p in ps:
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse):
                    raise RuntimeError("mvsnerf_amd.optim.Adam: fp32 CUDA parameters with dense fp32 gradients only")
                if not p.is_contiguous():
                    raise RuntimeError("mvsnerf_amd.optim.Adam: parameters must be contiguous")
                if not g.is_contiguous():
                    g = g.contiguous(); keep.append(g)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)               # as torch.optim.Adam keeps it (host tensor)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                s = int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"])     # a HOST tensor: no device synchronisation
                if step is None:
                    step = s
                elif s != step:                        # parameters that joined later: their own launch
                    raise RuntimeError("mvsnerf_amd.optim.Adam: parameters of one group with different step counts are not supported")
                ptr["p"].append(p.data_ptr()); ptr["g"].append(g.data_ptr())
                ptr["m"].append(st["exp_avg"].data_ptr()); ptr["v"].append(st["exp_avg_sq"].data_ptr())
                numel.append(p.numel())
            n = len(ps)
            bc1 = 1.0 - beta1 ** step
            bc2 = 1.0 - beta2 ** step
            arr = lambda xs: (ctypes.c_void_p * n)(*xs)
            check(lib.mvsnerf_adam_step_multi(n, arr(ptr["p"]), arr(ptr["g"]), arr(ptr["m"]), arr(ptr["v"]), (ctypes.c_int64 * n)(*numel),
                                              float(group["lr"]) / bc1, float(beta1), float(beta2), float(group["eps"]), math.sqrt(bc2), stream_ptr()),
                  "adam_step_multi")
            del keep
        return loss
