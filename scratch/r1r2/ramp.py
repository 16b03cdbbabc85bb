"""Does the MLP kernel run slower right after the GPU was idle?  Blocks of 25 launches timed back to back after a 1 s sleep."""
import sys, time, torch
sys.path.insert(0, '.')
import numpy as np
from mvsnerf_amd import _lib, models
dev = 'cuda'
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('mlp/')}
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type='v0'); net.load_state_dict(sd); net = net.to(dev)
L = _lib.lib()
N, S, F = 1024, 128, 20
g = torch.Generator().manual_seed(0)
ndc = torch.rand((N, S, 3), generator=g).to(dev); feat = torch.randn((N, S, F), generator=g).to(dev); dirs = torch.randn((N, 3), generator=g).to(dev)
raw = torch.empty((N * S, 4), device=dev); packed = net.packed(F); st = torch.cuda.current_stream().cuda_stream
f = lambda: L.mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
for _ in range(5): f()
torch.cuda.synchronize()
for trial in range(3):
    time.sleep(1.0)
    nb = 16
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
    ev[0].record()
    for b in range(nb):
        for _ in range(25): f()
        ev[b + 1].record()
    torch.cuda.synchronize()
    print("after 1 s idle, ms/launch per block of 25:", " ".join("%.4f" % (ev[b].elapsed_time(ev[b + 1]) / 25) for b in range(nb)))
