"""Fine census of the bf16 pair kernel (variant of /tmp/mlp_bf16_fine.hip, -DBF_CENSUS): layers 2 and 3: before the GEMM, after each of its 8 k-steps, GEMM end, epilogue end (+ barrier for the
younger waves), barrier (older waves); older / younger waves apart."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, os.environ["MVS_LIB"]); _lib._lib = None
from mvsnerf_amd import models, ops
import bench_common as bench
dev = torch.device("cuda", 0)
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
net.load_state_dict(bench.load_mlp_weights()); net = net.to(dev)
N, S, F = 1024, 128, 20
g = torch.Generator().manual_seed(0)
ndc = (torch.rand((N, S, 3), generator=g) * 1.2 - 0.1).to(dev)
feat = torch.randn((N, S, F), generator=g).to(dev)
dirs = torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=-1).to(dev)
n_tiles = N * S // 32
raw = torch.zeros(N * S * 4 + n_tiles * 32, device=dev)
lib = _lib.lib(); packed = net.packed(F); pb = net.packed_bf16(F)
st = torch.cuda.current_stream().cuda_stream
call = lambda: lib.mvsnerf_mlp_fwd_bf16(pb.data_ptr(), packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
for _ in range(200): call()
torch.cuda.synchronize(); raw[N * S * 4:].zero_()
for _ in range(3): call()
torch.cuda.synchronize()
c = raw[N * S * 4:].view(torch.int32).cpu().numpy().astype(np.int64).reshape(n_tiles, 32) & 0xffffffff
names = ["k0 (incl. first fragments)", "k1", "k2", "k3", "k4", "k5", "k6", "k7", "(gemm end)", "epilogue (+ barrier: young)", "barrier: old"]
t = c[:, :24].reshape(n_tiles, 2, 12)
d = (t[:, :, 1:] - t[:, :, :-1]) & 0xffffffff
old = (np.arange(n_tiles) % 8) < 4
for layer in (0, 1):
    print(f"layer {layer + 2}")
    for i, nm in enumerate(names):
        print(f"  {nm:32s} older waves {d[old, layer, i].mean():7.0f}   younger waves {d[~old, layer, i].mean():7.0f}")
