import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mvsnerf_amd.synth import make_rig
from oracle import mvsnerf_oracle as O
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
for (H, W, seed, rot, pad, D) in ((128, 160, 77, 2.0, 4, 32), (512, 640, 1234, 0.0, 24, 128), (64, 96, 1240, 3.0, 0, 16), (200, 200, 3, 5.0, 0, 24)):
    rig = make_rig(H, W, seed=seed, rot_deg=rot)
    proj = rig["proj_mats"][:, :3]
    dv = O.depth_planes(2.125, 4.525, D)
    h, w = H // 4, W // 4
    feat = torch.zeros(1, 1, h, w)
    for vv in (1, 2):
        _, grid = O.homo_warp(feat, proj[:, vv], dv, pad=pad)
        ref_grid = grid.numpy().reshape(-1, 2)
        P = proj[0, vv].numpy().astype(f32)
        Hp, Wp = h + 2 * pad, w + 2 * pad
        ys, xs = np.meshgrid(np.arange(Hp, dtype=f32) - f32(pad), np.arange(Wp, dtype=f32) - f32(pad), indexing="ij")
        u = np.broadcast_to(xs[None], (D, Hp, Wp)).reshape(-1).astype(f32); v = np.broadcast_to(ys[None], (D, Hp, Wp)).reshape(-1).astype(f32)
        dep = np.broadcast_to(dv[0].numpy()[:, None, None], (D, Hp, Wp)).reshape(-1).astype(f32); one = np.ones_like(u)
        for kind in ("fma_k", "plain", "fma_rev", "fma_mid"):
            p = []
            for r in range(3):
                a, b, cc, t = P[r]
                A, Bv, Cv = np.full_like(u, a), np.full_like(u, b), np.full_like(u, cc)
                if kind == "fma_k": m = fma(Cv, one, fma(Bv, v, (a * u).astype(f32)))
                elif kind == "plain": m = ((a * u).astype(f32) + (b * v).astype(f32)).astype(f32) + cc
                elif kind == "fma_rev": m = fma(A, u, fma(Bv, v, Cv))
                elif kind == "fma_mid": m = fma(A, u, (b * v).astype(f32)) + cc
                p.append((m + (t / dep).astype(f32)).astype(f32))
            gx = ((p[0] / p[2]).astype(f32) / f32((w - 1) / 2)).astype(f32) - f32(1); gy = ((p[1] / p[2]).astype(f32) / f32((h - 1) / 2)).astype(f32) - f32(1)
            nb = int((gx.view(np.int32) != ref_grid[:, 0].view(np.int32)).sum() + (gy.view(np.int32) != ref_grid[:, 1].view(np.int32)).sum())
            print(f"{H}x{W} rot {rot} view {vv} N={u.size} [{kind}]: {nb} of {2*gx.size} differ")
