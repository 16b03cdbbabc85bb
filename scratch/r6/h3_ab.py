"""Same-box timing of the fp16x3 MLP kernel (mvsnerf_mlp_fwd_split, n_split 18) at config 2's batch (1024 x 128 points), HIP events over 200
back-to-back launches after a settle phase; MVS_LIB=<variant .so> selects another build (scratch/r6/build_variant.sh).  Also prints the
kernel's error against the product fp32 kernel on the same inputs (any variant must reproduce the product results)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import _lib                       # noqa: E402
if os.environ.get("MVS_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["MVS_LIB"]); _lib._lib = None
from mvsnerf_amd import models, ops               # noqa: E402
import bench                                       # noqa: E402

dev = torch.device("cuda", 0)
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
net.load_state_dict(bench.load_mlp_weights())
net = net.to(dev)
N, S, F = int(os.environ.get("H3_N", "1024")), 128, 20
g = torch.Generator().manual_seed(0)
ndc = (torch.rand((N, S, 3), generator=g) * 1.2 - 0.1).to(dev)
feat = torch.randn((N, S, F), generator=g).to(dev)
dirs = torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=-1).to(dev)
raw = torch.empty((N, S, 4), device=dev)
raw32 = torch.empty((N, S, 4), device=dev)
lib = _lib.lib()
packed = net.packed(F)
st = torch.cuda.current_stream().cuda_stream
lib.mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw32.data_ptr(), st)
out = {}
for mode in sys.argv[1:] or ["fp16x3"]:
    ps, ns = net.packed_split(F, ops.N_SPLIT[mode])
    k = lambda: lib.mvsnerf_mlp_fwd_split(ps.data_ptr(), packed.data_ptr(), F, ns, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
    t = min(bench.event_time(k, 200) for _ in range(3))
    torch.cuda.synchronize()
    err = float((raw - raw32).abs().max())
    print(f"{os.environ.get('MVS_LIB', 'product'):48s} {mode:7s} kernel {t * 1e3:8.2f} us   max |raw - fp32 kernel| {err:.3g}   sigma max {float(raw32[..., 3].max()):.3g}")
