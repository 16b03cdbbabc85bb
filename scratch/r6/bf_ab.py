"""Same-box timing of the bf16 MLP kernel (mvsnerf_mlp_fwd_bf16) at config 2's batch and at config 4's feat_dim; MVS_LIB selects another build."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import _lib                       # noqa: E402
if os.environ.get("MVS_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["MVS_LIB"]); _lib._lib = None
from mvsnerf_amd import models, ops               # noqa: E402
import bench_common as bench                       # noqa: E402

dev = torch.device("cuda", 0)
for F in (20, 28):
    torch.manual_seed(F)
    net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=F, skips=[4], net_type="v0")
    if F == 20:
        net.load_state_dict(bench.load_mlp_weights())
    net = net.to(dev)
    N, S = 1024, 128
    g = torch.Generator().manual_seed(0)
    ndc = (torch.rand((N, S, 3), generator=g) * 1.2 - 0.1).to(dev)
    feat = torch.randn((N, S, F), generator=g).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn((N, 3), generator=g), dim=-1).to(dev)
    raw = torch.empty((N, S, 4), device=dev)
    raw32 = torch.empty((N, S, 4), device=dev)
    lib = _lib.lib()
    packed = net.packed(F)
    pb = net.packed_bf16(F)
    st = torch.cuda.current_stream().cuda_stream
    lib.mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw32.data_ptr(), st)
    k = lambda: lib.mvsnerf_mlp_fwd_bf16(pb.data_ptr(), packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
    t = min(bench.event_time(k, 200) for _ in range(3))
    torch.cuda.synchronize()
    flop = {20: 251392, 28: 253440}[F]
    print(f"{os.environ.get('MVS_LIB', 'product'):44s} bf16 F={F}  kernel {t * 1e3:7.2f} us  {flop * N * S / (t * 1e-3) / 1e12:7.1f} TF = {flop * N * S / (t * 1e-3) / 2.5e15:.3f} of 2.5 PF   max |rgb - fp32 kernel| {float((raw - raw32)[..., :3].abs().max()):.3g}")
