"""Phase timeline of the bf16 inference kernel (mlp_fwd_bf16_pair_kernel, -DBF_CENSUS), same phases as the fp16x3 census it is adapted from:
Phase timeline of the fp16x3 MLP kernel from per-wave shader-clock stamps (variant built with -DH3_CENSUS: 32 stamps per 32-point tile,
written behind the results).  MVS_LIB=scratch/lib/libmvsnerf_hip_h3_cen.so python scratch/r6/h3_census.py"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import _lib                       # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, os.environ["MVS_LIB"]); _lib._lib = None
from mvsnerf_amd import models, ops               # noqa: E402
import bench                                       # noqa: E402

dev = torch.device("cuda", 0)
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
net.load_state_dict(bench.load_mlp_weights())
net = net.to(dev)
N, S, F = 1024, 128, 20
g = torch.Generator().manual_seed(0)
ndc = (torch.rand((N, S, 3), generator=g) * 1.2 - 0.1).to(dev)
feat = torch.randn((N, S, F), generator=g).to(dev)
dirs = torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=-1).to(dev)
n_tiles = N * S // 32
raw = torch.zeros(N * S * 4 + n_tiles * 32, device=dev)
lib = _lib.lib()
packed = net.packed(F)
pb = net.packed_bf16(F)
st = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    lib.mvsnerf_mlp_fwd_bf16(pb.data_ptr(), packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
torch.cuda.synchronize()
raw[N * S * 4:].zero_()          # stamps of ONE launch in the steady state of back-to-back launches (32-bit stamps wrap within seconds)
for _ in range(3):
    lib.mvsnerf_mlp_fwd_bf16(pb.data_ptr(), packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
torch.cuda.synchronize()
c = raw[N * S * 4:].view(torch.int32).cpu().numpy().astype(np.int64).reshape(n_tiles, 32) & 0xffffffff
names = ["start", "sync0 (features + slab 0)", "G bias", "PE + G l0", "E l0 (+ barrier: young)", "barrier: old"] + \
        sum([[f"G l{i}", f"E l{i} (+ barrier: young)", f"barrier: old, l{i}"] for i in range(1, 5)], []) + \
        ["G l5 pe", "S l5b", "G l5 act", "E l5 + sigma (+ barrier: young)", "barrier: old, l5", "G feat", "E feat + prep (+ barrier: young)", "barrier: old, feat", "G views",
         "end (rgb head, store)"]
nst = len(names)
t = c[:, :nst]
d = (t[:, 1:] - t[:, :-1]) & 0xffffffff
t0 = t[:, 0]
print(f"tiles {n_tiles}; wave lifetime mean {((t[:, nst - 1] - t0) & 0xffffffff).mean():.0f} cycles")
# (the s_memtime counters of different CUs are not synchronised: only differences within a wave are used)
tot = 0
old = (np.arange(n_tiles) % 8) < 4          # waves 0..3 of a workgroup
for i in range(nst - 1):
    print(f"  {names[i + 1]:36s} mean {d[:, i].mean():8.0f}   older waves {d[old, i].mean():8.0f}   younger waves {d[~old, i].mean():8.0f}")
    tot += d[:, i].mean()
kinds = {"S": 0.0, "G": 0.0, "E": 0.0, "other": 0.0}
for i in range(nst - 1):
    k = "S" if names[i + 1].startswith("barrier") else names[i + 1][0] if names[i + 1][0] in "SGE" and names[i + 1][1] == " " else "other"
    kinds[k] += d[:, i].mean()
print("sum of phase means", round(tot), {k: round(v) for k, v in kinds.items()}, "(S = barrier + slab wait, G = GEMM loops, E = epilogues; MFMA issue alone: 250 x 32 = 8000 per wave, two waves per SIMD)")
