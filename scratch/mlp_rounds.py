import sys, torch
sys.path.insert(0,'.')
import numpy as np
from mvsnerf_amd import _lib, ops, models
dev='cuda'
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('mlp/')}
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type='v0'); net.load_state_dict(sd); net=net.to(dev)
S,F=128,20
packed=net.packed(F)
L=_lib.lib(); L.mvsnerf_tune(b"mlp_variant", 3)
g=torch.Generator().manual_seed(0)
for N in (128, 256, 384, 512, 768, 1024, 1536, 2048, 4096, 8192):
    ndc=torch.rand((N,S,3),generator=g).to(dev); feat=torch.randn((N,S,F),generator=g).to(dev); dirs=torch.randn((N,3),generator=g).to(dev)
    f=lambda: ops.mlp_forward(packed,F,ndc.data_ptr(),3,feat.data_ptr(),F,dirs.data_ptr(),3,N,S,False,dev)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/30*1e3
    print(f"N={N:5d} WGs={N:5d} rounds={N/512:.2f}: {us:7.1f} us -> {251392*N*S/us/1e6:6.1f} TF")
