"""Shared helpers for the tests: golden fixtures + checkpoint weights."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def load_weights():
    """-> (mlp_sd, mvs_sd) with the checkpoint's key names (ckpts/mvsnerf-v0.tar of the reference)."""
    z = np.load(os.path.join(GOLDEN, "mvsnerf_v0_weights.npz"))
    mlp = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")}
    mvs = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")}
    return mlp, mvs


def pose_of(c):
    return {k: c[k][0] for k in ("w2cs", "c2ws", "intrinsics", "near_fars")}


def maxabs(a, b):
    return float((a.double() - b.double()).abs().max())
