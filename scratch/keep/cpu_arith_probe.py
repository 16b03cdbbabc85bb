"""Which fp32 arithmetic does the CPU oracle (torch 2.10 CPU kernels under the reference's homo_warp / grid_sample) use?
Candidate formulas are evaluated with numpy float32 (one rounding per op; fma through float64) and compared BIT FOR BIT
with the reference-generated fixture (tests/golden/caseB.npz: ref_grid_v1, ref_warped_v1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.util import load_case
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
for name in ("caseA", "caseB"):
    c = load_case(name)
    pad = int(c["pad"]); feats = c["ref_feats"][0, 1].numpy(); P = c["proj_mats"][0, 1].numpy().astype(f32)
    dv = c["depth_values"][0].numpy().astype(f32)
    C, H, W = feats.shape; Hp, Wp = H + 2 * pad, W + 2 * pad; D = dv.shape[0]
    ys, xs = np.meshgrid(np.arange(Hp, dtype=f32) - f32(pad), np.arange(Wp, dtype=f32) - f32(pad), indexing="ij")
    u = np.broadcast_to(xs[None], (D, Hp, Wp)).reshape(-1).astype(f32); v = np.broadcast_to(ys[None], (D, Hp, Wp)).reshape(-1).astype(f32)
    dep = np.broadcast_to(dv[:, None, None], (D, Hp, Wp)).reshape(-1).astype(f32)
    one = np.ones_like(u)
    ref_grid = c["ref_grid_v1"].numpy().reshape(-1, 2)
    def rows(kind):
        out = []
        for r in range(3):
            a, b, cc, t = P[r, 0], P[r, 1], P[r, 2], P[r, 3]
            if kind == "fma_k":   m = fma(np.full_like(u, cc), one, fma(np.full_like(u, b), v, (a * u).astype(f32)))
            elif kind == "plain": m = ((a * u).astype(f32) + (b * v).astype(f32)).astype(f32) + f32(cc)
            elif kind == "fma_rev": m = fma(np.full_like(u, a), u, fma(np.full_like(u, b), v, np.full_like(u, cc)))
            out.append((m + (t / dep).astype(f32)).astype(f32))
        return out
    for kind in ("fma_k", "plain", "fma_rev"):
        p0, p1, p2 = rows(kind)
        gx = ((p0 / p2).astype(f32) / f32((W - 1) / 2)).astype(f32) - f32(1)
        gy = ((p1 / p2).astype(f32) / f32((H - 1) / 2)).astype(f32) - f32(1)
        nb = int((gx.view(np.int32) != ref_grid[:, 0].view(np.int32)).sum() + (gy.view(np.int32) != ref_grid[:, 1].view(np.int32)).sum())
        print(f"{name} grid[{kind}]: {nb} of {2 * gx.size} values differ in bits; max abs {max(np.abs(gx - ref_grid[:,0]).max(), np.abs(gy - ref_grid[:,1]).max()):.2e}")
    # ---- bilinear on the REFERENCE grid
    gx, gy = ref_grid[:, 0].astype(f32), ref_grid[:, 1].astype(f32)
    ref_w = c["ref_warped_v1"].numpy()[0].reshape(C, -1)
    def unnorm(g, size, kind):
        if kind == "a": return (((g + f32(1)).astype(f32) / f32(2)).astype(f32) * f32(size - 1)).astype(f32)
        if kind == "b": return ((g + f32(1)).astype(f32) * f32((size - 1) / 2)).astype(f32)      # vectorised CPU kernel: (x+1) * ((size-1)/2)
    for un in ("a", "b"):
        ix, iy = unnorm(gx, W, un), unnorm(gy, H, un)
        fx, fy = np.floor(ix), np.floor(iy)
        for wk in ("cuda", "vec"):
            if wk == "cuda":
                wx1 = (ix - fx).astype(f32); wx0 = ((fx + f32(1)).astype(f32) - ix).astype(f32); wy1 = (iy - fy).astype(f32); wy0 = ((fy + f32(1)).astype(f32) - iy).astype(f32)
            else:
                wx1 = (ix - fx).astype(f32); wx0 = (f32(1) - wx1).astype(f32); wy1 = (iy - fy).astype(f32); wy0 = (f32(1) - wy1).astype(f32)
            x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
            def tap(xx, yy):
                ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                return np.where(ok[None], feats[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], f32(0)), ok
            (nw, m0), (ne, m1), (sw, m2), (se, m3) = tap(x0, y0), tap(x0 + 1, y0), tap(x0, y0 + 1), tap(x0 + 1, y0 + 1)
            wnw, wne, wsw, wse = (wx0 * wy0).astype(f32), (wx1 * wy0).astype(f32), (wx0 * wy1).astype(f32), (wx1 * wy1).astype(f32)
            B = lambda w: np.broadcast_to(w[None], nw.shape)
            cands = {
                "sep": (((nw * wnw).astype(f32) + (ne * wne).astype(f32)).astype(f32) + (sw * wsw).astype(f32)).astype(f32) + (se * wse).astype(f32),
                "fma_chain": fma(se, B(wse), fma(sw, B(wsw), fma(ne, B(wne), (nw * wnw).astype(f32)))),
                "fma_pair": (fma(ne, B(wne), (nw * wnw).astype(f32)) + fma(se, B(wse), (sw * wsw).astype(f32))).astype(f32),
            }
            for k, val in cands.items():
                nb = int((val.astype(f32).view(np.int32) != ref_w.view(np.int32)).sum())
                print(f"{name} bilinear[unnorm {un}, weights {wk}, {k}]: {nb} of {val.size} differ; max abs {np.abs(val - ref_w).max():.2e}")
