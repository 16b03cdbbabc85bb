import sys, time, torch
sys.path.insert(0, '.')
import numpy as np
from mvsnerf_amd import _lib, ops, models
dev = 'cuda'
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('mlp/')}
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type='v0'); net.load_state_dict(sd); net = net.to(dev)
L = _lib.lib()
for (N, S) in ((1024, 128), (37, 5)):
    F = 20
    g = torch.Generator().manual_seed(0)
    ndc = torch.rand((N, S, 3), generator=g).to(dev); feat = torch.randn((N, S, F), generator=g).to(dev); dirs = torch.randn((N, 3), generator=g).to(dev)
    packed = net.packed(F)
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for var in (3, 4):
        assert L.mvsnerf_tune(b"mlp_variant", var) == 0
        for ao in (0, 1):
            raw = torch.full((N * S, 1 if ao else 4), float('nan'), device=dev)
            rc = L.mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, ao, raw.data_ptr(), st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs[(var, ao)] = raw.clone()
    for ao in (0, 1):
        d = (outs[(3, ao)] - outs[(4, ao)]).abs()
        print(f"N={N} S={S} alpha_only={ao}: max |v3 - v4| = {float(d.max()):.3e}  (|v3| max {float(outs[(3, ao)].abs().max()):.2f}), nan: {bool(torch.isnan(outs[(4, ao)]).any())}")
N, S, F = 1024, 128, 20
g = torch.Generator().manual_seed(0)
ndc = torch.rand((N, S, 3), generator=g).to(dev); feat = torch.randn((N, S, F), generator=g).to(dev); dirs = torch.randn((N, 3), generator=g).to(dev)
raw = torch.empty((N * S, 4), device=dev); packed = net.packed(F); st = torch.cuda.current_stream().cuda_stream
f = lambda: L.mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc.data_ptr(), 3, feat.data_ptr(), F, dirs.data_ptr(), 3, N, S, 0, raw.data_ptr(), st)
for rep in range(2):
    for var in (3, 4):
        L.mvsnerf_tune(b"mlp_variant", var)
        for _ in range(10): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): f()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 100
        print("variant", var, "%.4f ms  %.1f TFLOP/s" % (t, 251392 * N * S / (t * 1e-3) / 1e12))
L.mvsnerf_tune(b"mlp_variant", 3)
