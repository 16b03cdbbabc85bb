"""Variance step of build_volume_costvar (models.py:879-890) on the CPU oracle: separate roundings or fused?  Bit comparison with the
reference-generated fixture, warped values from the bit-verified formulas of cpu_arith_probe.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.util import load_case
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
for name in ("caseA", "caseB"):
    c = load_case(name)
    pad = int(c["pad"]); F = c["ref_feats"][0].numpy(); dv = c["depth_values"][0].numpy().astype(f32)
    V, C, H, W = F.shape[0], F.shape[1], F.shape[2], F.shape[3]; V = 3
    Hp, Wp, D = H + 2 * pad, W + 2 * pad, dv.shape[0]
    ys, xs = np.meshgrid(np.arange(Hp, dtype=f32) - f32(pad), np.arange(Wp, dtype=f32) - f32(pad), indexing="ij")
    u = np.broadcast_to(xs[None], (D, Hp, Wp)).reshape(-1).astype(f32); v = np.broadcast_to(ys[None], (D, Hp, Wp)).reshape(-1).astype(f32)
    dep = np.broadcast_to(dv[:, None, None], (D, Hp, Wp)).reshape(-1).astype(f32); one = np.ones_like(u)
    ref = np.zeros((C, Hp, Wp), f32); ref[:, pad:pad + H, pad:pad + W] = F[0]
    ref = np.broadcast_to(ref[:, None], (C, D, Hp, Wp)).reshape(C, -1)
    warped, cnt = [], np.ones_like(u)
    for vv in (1, 2):
        P = c["proj_mats"][0, vv].numpy().astype(f32); p = []
        for r in range(3):
            m = fma(np.full_like(u, P[r, 2]), one, fma(np.full_like(u, P[r, 1]), v, (P[r, 0] * u).astype(f32)))
            p.append((m + (P[r, 3] / dep).astype(f32)).astype(f32))
        gx = ((p[0] / p[2]).astype(f32) / f32((W - 1) / 2)).astype(f32) - f32(1); gy = ((p[1] / p[2]).astype(f32) / f32((H - 1) / 2)).astype(f32) - f32(1)
        cnt = cnt + ((gx > -1) & (gx < 1) & (gy > -1) & (gy < 1)).astype(f32)
        ix = (((gx + f32(1)) / f32(2)).astype(f32) * f32(W - 1)).astype(f32); iy = (((gy + f32(1)) / f32(2)).astype(f32) * f32(H - 1)).astype(f32)
        fx, fy = np.floor(ix), np.floor(iy)
        wx1 = (ix - fx).astype(f32); wx0 = ((fx + f32(1)) - ix).astype(f32); wy1 = (iy - fy).astype(f32); wy0 = ((fy + f32(1)) - iy).astype(f32)
        x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
        def tap(xx, yy):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok[None], F[vv][:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], f32(0))
        B = lambda w: np.broadcast_to(w[None], (C, w.size))
        warped.append(fma(tap(x0 + 1, y0 + 1), B((wx1 * wy1).astype(f32)), fma(tap(x0, y0 + 1), B((wx0 * wy1).astype(f32)),
                      fma(tap(x0 + 1, y0), B((wx1 * wy0).astype(f32)), (tap(x0, y0) * B((wx0 * wy0).astype(f32))).astype(f32)))))
    inv = (f32(1) / cnt).astype(f32)
    ref_var = c["ref_cost_var"].numpy()[0].reshape(C, -1)
    # separate: s = (ref + w1) + w2 ; s2 = (ref^2 + w1^2) + w2^2 ; var = s2*inv - (s*inv)^2, one rounding per op
    s = ((ref + warped[0]).astype(f32) + warped[1]).astype(f32)
    s2 = (((ref * ref).astype(f32) + (warped[0] * warped[0]).astype(f32)).astype(f32) + (warped[1] * warped[1]).astype(f32)).astype(f32)
    mean = (s * inv).astype(f32)
    sep = ((s2 * inv).astype(f32) - (mean * mean).astype(f32)).astype(f32)
    s2f = fma(warped[1], warped[1], fma(warped[0], warped[0], (ref * ref).astype(f32)))
    fused_acc = ((s2f * inv).astype(f32) - (mean * mean).astype(f32)).astype(f32)
    fused_all = fma(-mean, mean, (s2f * inv).astype(f32))
    fused_var_only = fma(-mean, mean, (s2 * inv).astype(f32))
    for k, val in {"separate roundings": sep, "fma in the sum of squares": fused_acc, "fma in the final subtraction only": fused_var_only, "both fused (default contraction)": fused_all}.items():
        nb = int((val.view(np.int32) != ref_var.view(np.int32)).sum())
        print(f"{name} variance[{k}]: {nb} of {val.size} differ in bits; max abs {np.abs(val - ref_var).max():.2e}  (|var| max {np.abs(ref_var).max():.2f})")
