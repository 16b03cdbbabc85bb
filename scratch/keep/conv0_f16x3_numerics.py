"""CPU probe (no GPU): how far would the neural volume and the rendered sigma move if conv0 of CostRegNet (74.5 % of the encoder's FLOPs) took its
operands as two fp16 pieces each (three fp16 matrix-core products per product, csrc/mlp_f16x3.hip's scheme) instead of fp32?
Config 2 (3 x 512x640 images, 128 planes, pad 24), shipped weights; the three piece convolutions are evaluated in float64, so what is measured is the
SPLIT alone (22 significant bits per operand, a1*w1 dropped) - the yardsticks: the fp32 oracle (torch CPU) and the same conv0 in float64.
Run:  python scratch/keep/conv0_f16x3_numerics.py      (about 10 minutes and 20 GB of host memory)"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mvsnerf_oracle as O          # noqa: E402
from tests.util import load_weights              # noqa: E402
from mvsnerf_amd.synth import make_rig, pose_ref_of   # noqa: E402


def crn_from_c0raw(c0raw, sd, prefix="cost_reg_2."):
    """oracle.cost_reg_net downstream of conv0's RAW output"""
    def cbr(x, name, stride=1):
        w = sd[prefix + name + ".conv.weight"]
        x = O._conv3d_s1(x, w) if stride == 1 else F.conv3d(x, w, None, stride=stride, padding=1)
        return O.abn(x, sd, prefix + name + ".bn")

    def up(x, name):
        x = F.conv_transpose3d(x, sd[prefix + name + ".0.weight"], None, stride=2, padding=1, output_padding=1)
        return O.abn(x, sd, prefix + name + ".1")
    c0 = O.abn(c0raw, sd, prefix + "conv0.bn")
    c2 = cbr(cbr(c0, "conv1", 2), "conv2")
    c4 = cbr(cbr(c2, "conv3", 2), "conv4")
    y = cbr(cbr(c4, "conv5", 2), "conv6")
    y = c4 + up(y, "conv7")
    y = c2 + up(y, "conv9")
    return c0 + up(y, "conv11")


def conv64(x64, w64):
    return O._conv3d_s1(x64, w64)


def main():
    torch.manual_seed(0)
    small = "--small" in sys.argv
    H, W, D, pad = (128, 160, 32, 8) if small else (512, 640, 128, 24)
    rig = make_rig(H, W, seed=1234)
    pose = pose_ref_of(rig)
    mlp_sd, sd = load_weights()
    with torch.no_grad():
        t0 = time.time()
        imgs, proj, nf = rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0]
        feats = O.feature_net(imgs.reshape(3, 3, H, W), sd).view(1, 3, 32, H // 4, W // 4)
        dv = O.depth_planes(float(nf[0]), float(nf[1]), D)
        cost, _ = O.build_volume_costvar_img(imgs, feats, proj, dv, pad)
        w = sd["cost_reg_2.conv0.conv.weight"]
        print("cost volume", tuple(cost.shape), "max |x|", float(cost.abs().max()), " max |w|", float(w.abs().max()), f"({time.time() - t0:.0f} s)")
        c0_32 = O._conv3d_s1(cost, w)                                   # the fp32 oracle's conv0 (oneDNN)
        x64, w64 = cost.double(), w.double()
        c0_64 = conv64(x64, w64)
        xh = cost.half(); xl = (cost - xh.float()).half()
        wh = w.half(); wl = (w - wh.float()).half()
        c0_s = conv64(xh.double(), wh.double()) + conv64(xh.double(), wl.double()) + conv64(xl.double(), wh.double())
        print(f"conv0 raw output, max |y| {float(c0_64.abs().max()):.4g}:  fp32 oracle vs float64 {float((c0_32.double() - c0_64).abs().max()):.3g}   "
              f"fp16x3 split vs float64 {float((c0_s - c0_64).abs().max()):.3g}   ({time.time() - t0:.0f} s)")
        sub = float(((xl != 0) & (xl.float().abs() < 2.0 ** -14)).double().mean())
        print(f"lo pieces of the cost volume that are fp16 subnormals: {sub:.3f}; cost values above 65504: {int((cost.abs() > 65504).sum())}")
        del x64, xh, xl
        vol_32 = crn_from_c0raw(c0_32, sd)
        vol_64in = crn_from_c0raw(c0_64.float(), sd)
        vol_s = crn_from_c0raw(c0_s.float(), sd)
        vmax = float(vol_32.abs().max())
        print(f"neural volume (|v| <= {vmax:.3g}):  fp16x3-conv0 vs fp32 oracle {float((vol_s - vol_32).abs().max()):.3g} (rms {float((vol_s - vol_32).pow(2).mean().sqrt()):.3g});  "
              f"exact-conv0 vs fp32 oracle {float((vol_64in - vol_32).abs().max()):.3g} (rms {float((vol_64in - vol_32).pow(2).mean().sqrt()):.3g});  "
              f"fp16x3-conv0 vs exact-conv0 {float((vol_s - vol_64in).abs().max()):.3g}   ({time.time() - t0:.0f} s)")
        g = torch.Generator().manual_seed(3)
        n_rays = 256 if small else 1024
        pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], n_rays, 128, pad=pad, t_rand=torch.rand((n_rays, 128), generator=g), generator=g)
        outs = {}
        for name, v in (("fp32", vol_32), ("exact", vol_64in), ("fp16x3", vol_s)):
            outs[name] = O.rendering(pose, pts, ndc, z, dirs, v.reshape(1, 8, *v.shape[2:]), rig["images_raw"][:, :3], mlp_sd)
        for a, b in (("fp16x3", "fp32"), ("exact", "fp32"), ("fp16x3", "exact")):
            ds = (outs[a][6][..., 3] - outs[b][6][..., 3]).abs()
            print(f"rendering on those volumes, {a:6s} vs {b:5s}: sigma max {float(ds.max()):.3g} (n > 1e-4: {int((ds > 1e-4).sum())}), rgb map {float((outs[a][0] - outs[b][0]).abs().max()):.3g}")


if __name__ == "__main__":
    main()
