"""CPU probe (no GPU): what does a two-piece fp16 split of both GEMM operands cost in accuracy on the shipped MLP?

    a = a0 + a1 (+ r),  a0 = fp16(a), a1 = fp16(a - a0)            (round to nearest: |r| <= 2^-22 |a| while a1 is normal)
    a*w ~= a0*w0 + a0*w1 + a1*w0                                   (dropped: a1*w1 <= 2^-22 |a*w|)

= three v_mfma_f32_32x32x16_f16 per product ("fp16x3"), against six bf16 ones for the three-piece bf16 split ("bf16x6") and three
for the two-piece bf16 split ("bf16x3", 2^-16).  Products of fp16 pieces are exact in fp32; the sums here are taken in float64, so what
is printed is the error of the SPLIT alone (the matrix core's fp32 accumulation adds what it adds to every fp32 kernel).

Also printed: the largest |operand| per layer (fp16 overflows at 65504) and how many lo pieces are fp16 subnormals.
Run:  python scratch/keep/f16x3_numerics.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mvsnerf_oracle as O          # noqa: E402
from util import load_weights                    # noqa: E402
from test_gpu_raymarch import _config2_inputs   # noqa: E402


def split(x, kind):
    """pieces of x (fp32 tensor) as float64 tensors"""
    if kind == "f16x2":
        a0 = x.half()
        a1 = (x - a0.float()).half()
        return [a0.double(), a1.double()]
    if kind == "bf16x2":
        a0 = x.bfloat16()
        a1 = (x - a0.float()).bfloat16()
        return [a0.double(), a1.double()]
    if kind == "bf16x3":
        a0 = x.bfloat16()
        r = x - a0.float()
        a1 = r.bfloat16()
        a2 = (r - a1.float()).bfloat16()
        return [a0.double(), a1.double(), a2.double()]
    raise ValueError(kind)


STATS = {}


def lin_split(h, W, b, kind, name):
    hp, wp = split(h, kind), split(W, kind)
    n = len(hp)
    y = torch.zeros(h.shape[0], W.shape[0], dtype=torch.float64)
    for i in range(n):
        for j in range(n):
            if i + j < n:
                y += hp[i] @ wp[j].T
    if kind == "f16x2":
        lo = hp[1]
        STATS[name] = (float(h.abs().max()), float(W.abs().max()), float(((lo != 0) & (lo.abs() < 2.0 ** -14)).double().mean()))
    return (y + b.double()).float()


def mlp(x, sd, kind):
    """oracle.renderer_ours with the matrix products replaced (kind None: plain fp32, 'f64': float64)"""
    p = "nerf."
    if kind == "f64":
        lin = lambda name, h: (h.double() @ sd[p + name + ".weight"].double().T + sd[p + name + ".bias"].double())
        x = x.double()
    elif kind is None:
        lin = lambda name, h: torch.nn.functional.linear(h, sd[p + name + ".weight"], sd[p + name + ".bias"])
    else:
        lin = lambda name, h: lin_split(h, sd[p + name + ".weight"], sd[p + name + ".bias"], kind, name)
    pts, feat, dirs = x[..., :63], x[..., 63:-3], x[..., -3:]
    bias = lin("pts_bias", feat)
    h = pts
    for i in range(6):
        h = torch.relu(lin(f"pts_linears.{i}", h) * bias)
        if i == 4:
            h = torch.cat([pts, h], -1)
    alpha = torch.relu(lin("alpha_linear", h))
    h = torch.cat([lin("feature_linear", h), dirs], -1)
    h = torch.relu(lin("views_linears.0", h))
    rgb = torch.sigmoid(lin("rgb_linear", h))
    return torch.cat([rgb, alpha], -1)


def main():
    torch.manual_seed(0)
    n_rays = 256
    rig, pose, vol, pts, dirs, ndc, z, ro = _config2_inputs(n_rays, 128, D=32, h=48, w=64, H=128, W=160, seed=3)
    sd, _ = load_weights()
    with torch.no_grad():
        ref = O.rendering(pose, pts, ndc, z, dirs, vol, rig["images_raw"][:, :3], sd)
        feat = ref[1]
        ang = O.gen_dir_feature(pose["w2cs"][0], dirs / torch.norm(dirs, dim=-1, keepdim=True))
        x = torch.cat((O.embed(ndc), feat, ang[:, None].expand(-1, 128, -1)), -1).reshape(-1, 63 + feat.shape[-1] + 3)
        y32 = mlp(x, sd, None)
        y64 = mlp(x, sd, "f64").float()
        print("samples", x.shape[0], " sigma max", float(y32[:, 3].max()))
        print("fp32 (torch CPU) vs float64        : sigma %.3g  rgb %.3g" % (float((y32 - y64)[:, 3].abs().max()), float((y32 - y64)[:, :3].abs().max())))
        for kind in ("f16x2", "bf16x3", "bf16x2"):
            ys = mlp(x, sd, kind)
            print("%-7s split vs float64            : sigma %.3g  rgb %.3g   | vs fp32: sigma %.3g rgb %.3g" % (
                kind, float((ys - y64)[:, 3].abs().max()), float((ys - y64)[:, :3].abs().max()),
                float((ys - y32)[:, 3].abs().max()), float((ys - y32)[:, :3].abs().max())))
        print("fp16x3 operand ranges  (max |activation|, max |weight|, share of non-zero lo pieces that are fp16 subnormals):")
        for k, v in STATS.items():
            print("   %-16s %10.4g %10.4g %8.4f" % (k, *v))
        # what a flush of subnormal INPUTS would cost (MI200 did that for fp16 MFMA; gfx942+ does not - probed on the GPU by the parity test)
        def flush(t):
            return torch.where(t.abs() < 2.0 ** -14, torch.zeros_like(t), t)
        global split
        _split = split
        split = lambda x_, kind_: [flush(p_) for p_ in _split(x_, kind_)]
        ysf = mlp(x, sd, "f16x2")
        print("fp16x3 with subnormal pieces flushed: sigma %.3g  rgb %.3g (vs float64)" % (float((ysf - y64)[:, 3].abs().max()), float((ysf - y64)[:, :3].abs().max())))


if __name__ == "__main__":
    main()


def scaled_probe():
    """fixed power-of-two operand scales (exact) against a flush of subnormal pieces: activations x SA, weights x SW"""
    global split
    rig, pose, vol, pts, dirs, ndc, z, ro = _config2_inputs(256, 128, D=32, h=48, w=64, H=128, W=160, seed=3)
    sd, _ = load_weights()
    with torch.no_grad():
        ref = O.rendering(pose, pts, ndc, z, dirs, vol, rig["images_raw"][:, :3], sd)
        ang = O.gen_dir_feature(pose["w2cs"][0], dirs / torch.norm(dirs, dim=-1, keepdim=True))
        x = torch.cat((O.embed(ndc), ref[1], ang[:, None].expand(-1, 128, -1)), -1).reshape(-1, 63 + ref[1].shape[-1] + 3)
        y64 = mlp(x, sd, "f64").float()
        base = split
        for SA, SW in ((1, 1), (16, 256), (64, 1024), (128, 4096)):
            def sp(t, kind, SA=SA, SW=SW):
                s = SW if t.dim() == 2 and t.shape[0] in (128, 64) and t.shape[1] in (20, 63, 128, 191, 131) else SA
                a0 = (t * s).half()
                a1 = (t * s - a0.float()).half()
                fl = lambda u: torch.where(u.abs() < 2.0 ** -14, torch.zeros_like(u), u)
                return [fl(a0).double() / s, fl(a1).double() / s]
            split = sp
            ys = mlp(x, sd, "f16x2")
            print("flush + scales SA=%d SW=%d: sigma %.3g rgb %.3g (vs float64)" % (SA, SW, float((ys - y64)[:, 3].abs().max()), float((ys - y64)[:, :3].abs().max())))
        split = base


if __name__ == "__main__" and "--scaled" in sys.argv:
    scaled_probe()
