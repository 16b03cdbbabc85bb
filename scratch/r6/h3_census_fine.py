"""Fine census of the fp16x3 MLP kernel (variant built with -DH3_CENSUS_FINE): per wave, layers 2 and 3: arrival at the layer barrier, barrier passed, after each of
the 8 k-steps, GEMM end, epilogue end.  Prints the timeline split by which of the two waves of a SIMD pair finishes its GEMM first.
MVS_LIB=scratch/lib/libmvsnerf_hip_cenf.so python scratch/r6/h3_census_fine.py"""
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
ps, ns = net.packed_split(F, ops.N_SPLIT["fp16x3"])
st = torch.cuda.current_stream().cuda_stream
call = lambda: lib.mvsnerf_mlp_fwd_split(ps.data_ptr(), packed.data_ptr(), F, ns, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
for _ in range(200):
    call()
torch.cuda.synchronize()
raw[N * S * 4:].zero_()
for _ in range(3):
    call()
torch.cuda.synchronize()
c = raw[N * S * 4:].view(torch.int32).cpu().numpy().astype(np.int64).reshape(n_tiles, 32) & 0xffffffff
# per layer 12 stamps: arrive, passed, k0..k7, gemm end, epilogue end
names = ["barrier wait", "k0 (incl. acc init, first split)", "k1", "k2", "k3", "k4", "k5", "k6", "k7", "(gemm end)", "epilogue"]
t = c[:, :24].reshape(n_tiles, 2, 12)
d = (t[:, :, 1:] - t[:, :, :-1]) & 0xffffffff
# waves w and w + 4 of a workgroup share a SIMD: tiles 8b + w and 8b + w + 4
tw = t.reshape(n_tiles // 8, 8, 2, 12)
dw = d.reshape(n_tiles // 8, 8, 2, 11)
for layer in (0, 1):
    gemm_end = (tw[:, :, layer, 10] - tw[:, :, layer, 1]) & 0xffffffff           # from barrier passed
    first = gemm_end[:, :4] <= gemm_end[:, 4:]                                    # lower-numbered wave finishes first?
    fast = np.where(first[..., None], dw[:, :4, layer], dw[:, 4:, layer])
    slow = np.where(first[..., None], dw[:, 4:, layer], dw[:, :4, layer])
    print(f"layer {layer + 2}: lower-numbered wave of the pair finishes its GEMM first in {first.mean() * 100:.0f} % of the pairs")
    for i, nm in enumerate(names):
        print(f"  {nm:36s} fast wave mean {fast[..., i].mean():7.0f}   slow wave mean {slow[..., i].mean():7.0f}")
    skew = ((tw[:, 4:, layer, 1] - tw[:, :4, layer, 1] + 2**31) & 0xffffffff) - 2**31
    print(f"  barrier-release skew within a pair (wave+4 minus wave): mean {skew.mean():.0f}, |.| mean {np.abs(skew).mean():.0f}")
