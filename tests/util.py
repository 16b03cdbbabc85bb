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


def record_err(tag, err, scale=None, tol=None):
    """Append a measured error to gpurun_out/measured_errs.jsonl (the asserts' bounds are kept within 5x of these)."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "measured_errs.jsonl")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"tag": tag, "err": err, "scale": scale, "tol": tol}) + "\n")
    except OSError:
        pass


def caller_tag(depth=2):
    import inspect
    st = inspect.stack()
    fr = st[depth]
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::")[-1]
    return f"{test}:{fr.lineno}"
