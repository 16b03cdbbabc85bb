"""GPU box diagnostic: does the HIP homo_warp / plane sweep reproduce the CPU oracle's BITS on this host's CPU?  (The authoring
container's CPU agrees bit for bit with the k-ordered fma chain: scratch/keep/cpu_arith_probe.py; a different host CPU may run another
sgemm / grid_sample code path.)"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mvsnerf_amd import encoder as E, models
from mvsnerf_amd.synth import make_rig
from oracle import mvsnerf_oracle as O
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Flags' | cut -c1-300", shell=True, capture_output=True, text=True).stdout)
print(torch.__config__.show().split("\n")[3:8])
print("threads", torch.get_num_threads())
bits = lambda a, b: int((a.contiguous().view(torch.int32) != b.contiguous().view(torch.int32)).sum())
for (H, W, seed, rot, pad, D) in ((128, 160, 77, 2.0, 4, 32), (64, 96, 1240, 3.0, 0, 16), (200, 200, 3, 5.0, 0, 24), (512, 640, 1234, 0.0, 24, 128)):
    rig = make_rig(H, W, seed=seed, rot_deg=rot)
    proj = rig["proj_mats"][:, :3]
    dv = O.depth_planes(2.125, 4.525, D)
    h, w = H // 4, W // 4
    g = torch.Generator().manual_seed(1)
    feats = torch.randn((1, 3, 32, h, w), generator=g) * 3
    for nt in (torch.get_num_threads(), 8, 1):
        torch.set_num_threads(nt)
        for vv in (1, 2):
            warped_o, grid_o = O.homo_warp(feats[:, vv], proj[:, vv], dv, pad=pad)
            with torch.no_grad():
                warped_h, grid_h = E.homo_warp(feats[:, vv].cuda(), proj[:, vv].cuda(), dv.cuda(), pad=pad)
            gh = grid_h.cpu().reshape(-1, 2); go = grid_o.reshape(-1, 2)
            # HIP bilinear on the ORACLE's grid: isolates grid_sample's arithmetic
            with torch.no_grad():
                warped_h2, _ = E.homo_warp(feats[:, vv].cuda(), proj[:, vv].cuda(), dv.cuda(), src_grid=grid_o.cuda(), pad=pad)
            print(f"{H}x{W} rot {rot} view {vv} threads {nt}: grid values with different bits {bits(gh, go)} of {go.numel()} (max abs {float((gh-go).abs().max()):.2e}); "
                  f"warp on own grid {bits(warped_h.cpu(), warped_o)} of {warped_o.numel()} differ (max {float((warped_h.cpu()-warped_o).abs().max()):.2e}); "
                  f"warp on the oracle's grid {bits(warped_h2.cpu(), warped_o)} differ (max {float((warped_h2.cpu()-warped_o).abs().max()):.2e})")
