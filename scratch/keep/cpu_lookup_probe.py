"""Which fp32 arithmetic do get_ndc_coordinate, build_color_volume (2-D grid_sample, border) and index_point_feature (5-D grid_sample)
have on the authoring host?  Candidates in numpy float32 vs the reference-generated fixtures, bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.util import load_case
f32 = np.float32
def fma(a, b, c): return (np.asarray(a, f32).astype(np.float64) * np.asarray(b, f32).astype(np.float64) + np.asarray(c, f32).astype(np.float64)).astype(f32)
def nbits(a, b): return int((np.ascontiguousarray(a, f32).view(np.int32) != np.ascontiguousarray(b, f32).view(np.int32)).sum())
def chain(M, x, y, z, kind):     # row-vector times M^T: out_j = x*M[j,0] + y*M[j,1] + z*M[j,2]
    out = []
    for j in range(3):
        a, b, c = f32(M[j, 0]), f32(M[j, 1]), f32(M[j, 2])
        if kind == "fma_k": v = fma(z, c, fma(y, b, (x * a).astype(f32)))
        elif kind == "plain": v = (((x * a).astype(f32) + (y * b).astype(f32)).astype(f32) + (z * c).astype(f32)).astype(f32)
        elif kind == "fma_rev": v = fma(x, a, fma(y, b, (z * c).astype(f32)))
        out.append(v)
    return out
for name in ("caseA", "caseB"):
    c = load_case(name)
    pad = int(c["pad"]); H, W = int(c["H"]), int(c["W"])
    pts = c["ref_rays_pts"].numpy().reshape(-1, 3).astype(f32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    w2c, K = c["w2cs"][0, 0].numpy().astype(f32), c["intrinsics"][0, 0].numpy().astype(f32)
    nf = c["near_fars"][0, 0].numpy().astype(f32)
    ref_ndc = c["ref_rays_ndc"].numpy().reshape(-1, 3)
    for k1 in ("fma_k", "plain", "fma_rev"):
        for k2 in ("fma_k", "plain", "fma_rev"):
            p = chain(w2c[:3, :3], x, y, z, k1)
            p = [(p[j] + w2c[j, 3]).astype(f32) for j in range(3)]
            q = chain(K, p[0], p[1], p[2], k2)
            nx = ((q[0] / q[2]).astype(f32) + f32(0)) / f32(W - 1); ny = ((q[1] / q[2]).astype(f32) + f32(0)) / f32(H - 1)
            nz = ((q[2] - nf[0]).astype(f32) / (nf[1] - nf[0]).astype(f32)).astype(f32)
            nx, ny = nx.astype(f32), ny.astype(f32)
            if pad > 0:
                Wf, Hf = f32(W) / f32(4), f32(H) / f32(4)
                ny = ((ny * Hf).astype(f32) / (Hf + f32(pad * 2))).astype(f32) + (f32(pad) / (Hf + f32(pad * 2))).astype(f32)
                nx = ((nx * Wf).astype(f32) / (Wf + f32(pad * 2))).astype(f32) + (f32(pad) / (Wf + f32(pad * 2))).astype(f32)
            print(f"{name} ndc [R:{k1} K:{k2}]: x {nbits(nx, ref_ndc[:,0])} y {nbits(ny, ref_ndc[:,1])} z {nbits(nz, ref_ndc[:,2])} of {nx.size} differ")

# ---------------- colour lookup (2-D grid_sample, border padding, align_corners) and trilinear lookup (5-D, zeros)
for name in ("caseA", "caseB"):
    c = load_case(name)
    H, W = int(c["H"]), int(c["W"])
    pts = c["ref_rays_pts"].numpy().reshape(-1, 3).astype(f32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    imgs = c["images_raw"][0].numpy().astype(f32)
    ref_col = c["ref_colors"].numpy().reshape(-1, 12)
    for v in range(3):
        w2c, K = c["w2cs"][0, v].numpy().astype(f32), c["intrinsics"][0, v].numpy().astype(f32)
        p = chain(w2c[:3, :3], x, y, z, "fma_k"); p = [(p[j] + w2c[j, 3]).astype(f32) for j in range(3)]
        q = chain(K, p[0], p[1], p[2], "fma_k")
        nx = (((q[0] / q[2]).astype(f32) + f32(0)) / f32(W - 1)).astype(f32); ny = (((q[1] / q[2]).astype(f32) + f32(0)) / f32(H - 1)).astype(f32)
        gx, gy = (nx * f32(2) - f32(1)).astype(f32), (ny * f32(2) - f32(1)).astype(f32)
        mask = ((gx > -1) & (gx < 1) & (gy > -1) & (gy < 1)).astype(f32)
        print(f"{name} view {v}: mask differs at {int((mask != ref_col[:, 4*v+3]).sum())}")
        ix = (((gx + f32(1)) / f32(2)).astype(f32) * f32(W - 1)).astype(f32); iy = (((gy + f32(1)) / f32(2)).astype(f32) * f32(H - 1)).astype(f32)
        ix = np.minimum(np.maximum(ix, f32(0)), f32(W - 1)); iy = np.minimum(np.maximum(iy, f32(0)), f32(H - 1))
        fx, fy = np.floor(ix), np.floor(iy)
        wx1 = (ix - fx).astype(f32); wx0 = ((fx + f32(1)) - ix).astype(f32); wy1 = (iy - fy).astype(f32); wy0 = ((fy + f32(1)) - iy).astype(f32)
        x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
        def tap(xx, yy):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok[None], imgs[v][:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], f32(0))
        nw, ne, sw, se = tap(x0, y0), tap(x0 + 1, y0), tap(x0, y0 + 1), tap(x0 + 1, y0 + 1)
        B = lambda w: np.broadcast_to(w[None], nw.shape)
        wnw, wne, wsw, wse = (wx0 * wy0).astype(f32), (wx1 * wy0).astype(f32), (wx0 * wy1).astype(f32), (wx1 * wy1).astype(f32)
        cands = {"fma chain nw,ne,sw,se": fma(se, B(wse), fma(sw, B(wsw), fma(ne, B(wne), (nw * B(wnw)).astype(f32)))),
                 "separate": ((((nw * B(wnw)).astype(f32) + (ne * B(wne)).astype(f32)).astype(f32) + (sw * B(wsw)).astype(f32)).astype(f32) + (se * B(wse)).astype(f32)).astype(f32)}
        for k, val in cands.items():
            print(f"{name} view {v} colours [{k}]: {nbits(val.T, ref_col[:, 4*v:4*v+3])} of {val.size} differ (max {np.abs(val.T - ref_col[:, 4*v:4*v+3]).max():.1e})")
    # ---- trilinear
    vol = c["ref_vol_small"][0].numpy().astype(f32)     # (8,D,h,w)
    C, D, h, w = vol.shape
    ndc = c["ref_rays_ndc"].numpy().reshape(-1, 3).astype(f32)
    ref_vf = c["ref_vfeat"].numpy().reshape(-1, 8)
    g = [(ndc[:, k] * f32(2) - f32(1)).astype(f32) for k in range(3)]
    ix = (((g[0] + f32(1)) / f32(2)).astype(f32) * f32(w - 1)).astype(f32); iy = (((g[1] + f32(1)) / f32(2)).astype(f32) * f32(h - 1)).astype(f32); iz = (((g[2] + f32(1)) / f32(2)).astype(f32) * f32(D - 1)).astype(f32)
    fx, fy, fz = np.floor(ix), np.floor(iy), np.floor(iz)
    wx = [((fx + f32(1)) - ix).astype(f32), (ix - fx).astype(f32)]; wy = [((fy + f32(1)) - iy).astype(f32), (iy - fy).astype(f32)]; wz = [((fz + f32(1)) - iz).astype(f32), (iz - fz).astype(f32)]
    def vtap(zc, yc, xc):
        zz, yy, xx = fz.astype(np.int64) + zc, fy.astype(np.int64) + yc, fx.astype(np.int64) + xc
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h) & (zz >= 0) & (zz < D)
        return np.where(ok[None], vol[:, np.clip(zz, 0, D - 1), np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], f32(0)), ok
    order = [(0,0,0),(0,0,1),(0,1,0),(0,1,1),(1,0,0),(1,0,1),(1,1,0),(1,1,1)]     # tnw,tne,tsw,tse,bnw,bne,bsw,bse  (z,y,x)
    for wk in ("(x*y)*z", "x*(y*z)"):
        for acc_kind in ("fma", "separate", "fma_skip_oob"):
            acc = np.zeros((C, ndc.shape[0]), f32)
            for (zc, yc, xc) in order:
                val, ok = vtap(zc, yc, xc)
                wgt = ((wx[xc] * wy[yc]).astype(f32) * wz[zc]).astype(f32) if wk == "(x*y)*z" else (wx[xc] * (wy[yc] * wz[zc]).astype(f32)).astype(f32)
                Bw = np.broadcast_to(wgt[None], val.shape)
                if acc_kind == "fma": acc = fma(val, Bw, acc)
                elif acc_kind == "separate": acc = (acc + (val * Bw).astype(f32)).astype(f32)
                else: acc = np.where(ok[None], fma(val, Bw, acc), acc)
            print(f"{name} trilinear [weights {wk}, {acc_kind}]: {nbits(acc.T, ref_vf)} of {acc.size} differ (max {np.abs(acc.T - ref_vf).max():.1e})")
