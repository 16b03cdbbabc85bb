"""Seeded synthetic multi-view rig (no dataset exists in the build image - SURVEY.md 8d).

Produces the same dict schema `MVSDatasetDTU.__getitem__` hands to `training_step`
(reference data/dtu.py:199-211, collated with B=1): images, proj_mats, w2cs, c2ws,
intrinsics, near_fars.  Used by bench.py, __graft_entry__.smoke() and the tests.
"""
import math

import numpy as np
import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _rot_y(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_x(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def make_rig(H=512, W=640, n_views=4, seed=1234, baselines=(0.0, 0.25, -0.25, 0.1), rot_deg=0.0,
             near_far=(2.125, 4.525), smooth=False):
    """Returns a dict of float32 CPU tensors:
        images      (1,V,3,H,W) ImageNet-normalised (what MVSNet eats)
        images_raw  (1,V,3,H,W) in [0,1)              (what the colour lookup eats)
        proj_mats   (1,V,3,4)  K/4 [R|t]_v (K/4 [R|t]_0)^-1, view 0 = identity   (data/dtu.py:90-92,172-176)
        w2cs, c2ws  (1,V,4,4);  intrinsics (1,V,3,3);  near_fars (1,V,2)
    The last view is the render target (reference convention, utils.py:177).
    rot_deg != 0 adds small per-view rotations (tests use it so that R != I is exercised)."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.rand((1, n_views, 3, H, W), generator=g, dtype=torch.float32)
    if smooth:
        k = torch.ones(3, 1, 5, 5) / 25.0
        raw = torch.nn.functional.conv2d(raw.view(n_views, 3, H, W), k, padding=2, groups=3).view(1, n_views, 3, H, W).clamp(0, 1)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 1, 3, 1, 1)
    K = np.array([[0.9 * W, 0, W / 2.0], [0, 0.9 * W, H / 2.0], [0, 0, 1.0]], dtype=np.float64)
    w2cs, c2ws, projs = [], [], []
    for v in range(n_views):
        R = np.eye(3)
        if rot_deg:
            R = _rot_y(rot_deg * ((v % 3) - 1)) @ _rot_x(0.5 * rot_deg * ((v + 1) % 2))
        t = np.array([baselines[v % len(baselines)], 0.02 * rot_deg * v, 0.0])
        w2c = np.eye(4)
        w2c[:3, :3], w2c[:3, 3] = R, t
        w2cs.append(w2c)
        c2ws.append(np.linalg.inv(w2c))
        Kq = K.copy()
        Kq[:2] /= 4.0
        P = np.eye(4)
        P[:3, :4] = Kq @ w2c[:3, :4]
        projs.append(P)
    ref_inv = np.linalg.inv(projs[0])
    proj_mats = np.stack([np.eye(4) if v == 0 else projs[v] @ ref_inv for v in range(n_views)])[:, :3]
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return {
        "images": ((raw - mean) / std).contiguous(),
        "images_raw": raw,
        "proj_mats": f32(proj_mats)[None],
        "w2cs": f32(np.stack(w2cs))[None],
        "c2ws": f32(np.stack(c2ws))[None],
        "intrinsics": f32(np.stack([K] * n_views))[None],
        "near_fars": torch.tensor(near_far, dtype=torch.float32).view(1, 1, 2).repeat(1, n_views, 1),
    }


def pose_ref_of(rig, device=None):
    """The `pose_ref` dict of train_mvs_nerf_pl.py:59-60 (batch dim squeezed)."""
    d = {k: rig[k][0] for k in ("w2cs", "c2ws", "intrinsics", "near_fars")}
    if device is not None:
        d = {k: v.to(device) for k, v in d.items()}
    return d
