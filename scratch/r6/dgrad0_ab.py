"""conv0's data gradient (8 -> 32 channels, 3x3x3, 128x176x208 voxels): the LDS-tiled VALU kernel (mvsnerf_conv3d_fwd) against the direct-load fp32-MFMA kernel
(mvsnerf_conv3d_mfma_fwd with the <8, 32, 1> instantiation).  HIP events; results compared."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvsnerf_amd import _lib
import bench
dev = torch.device("cuda", 0)
lib = _lib.lib()
D, H, W = 128, 176, 208
g = torch.Generator().manual_seed(0)
x = torch.randn((D, H, W, 8), generator=g).to(dev)
w = (torch.randn((27, 8, 32), generator=g) * 0.1).to(dev)           # [tap][ci][co]: mvsnerf_conv3d_pack_weights layout
w32 = torch.empty(27 * 8 * 32, device=dev)
st = torch.cuda.current_stream().cuda_stream
rc = lib.mvsnerf_conv3d_pack_weights_mfma(w.data_ptr(), 8, 32, w32.data_ptr(), st); assert rc == 0, rc
o1 = torch.empty((D, H, W, 32), device=dev); o2 = torch.empty_like(o1)
k1 = lambda: lib.mvsnerf_conv3d_fwd(x.data_ptr(), 0, 0, 0, 0, 0, 8, 8, D, H, W, w.data_ptr(), 32, 1, o1.data_ptr(), st)
k2 = lambda: lib.mvsnerf_conv3d_mfma_fwd(x.data_ptr(), 0, 0, 8, 8, D, H, W, w32.data_ptr(), 32, 1, o2.data_ptr(), 0, st)
assert k1() == 0 and k2() == 0
torch.cuda.synchronize()
print("max |tiled - mfma|", float((o1 - o2).abs().max()), "max |out|", float(o1.abs().max()))
for name, k in (("tiled VALU", k1), ("fp32 MFMA direct-load", k2)):
    t = min(bench.event_time(k, 20) for _ in range(3))
    print(f"{name:24s} {t * 1e3:8.1f} us   {64.8e9 * D * H * W / 4685824 / (t * 1e-3) / 1e12:6.1f} TF")
