import sys, torch
sys.path.insert(0,'.')
import numpy as np
from mvsnerf_amd import _lib, ops, models
dev='cuda'
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('mlp/')}
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type='v0'); net.load_state_dict(sd); net=net.to(dev)
N,S,F=1024,128,20
g=torch.Generator().manual_seed(0)
ndc=torch.rand((N,S,3),generator=g).to(dev); feat=torch.randn((N,S,F),generator=g).to(dev); dirs=torch.randn((N,3),generator=g).to(dev)
packed=net.packed(F)
def t(iters=30):
    f=lambda: ops.mlp_forward(packed,F,ndc.data_ptr(),3,feat.data_ptr(),F,dirs.data_ptr(),3,N,S,False,dev)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
L=_lib.lib()
for variant in (3,):
    L.mvsnerf_tune(b"mlp_variant", variant)
    for dbg,name in ((0,"full"),(1,"no sincos"),(2,"no barriers/waits (wrong results)"),(3,"no sincos, no barriers")):
        L.mvsnerf_tune(b"mlp_dbg", dbg)
        us=t()
        print(f"variant {variant} dbg {dbg} ({name}): {us:.1f} us -> {251392*N*S/us/1e6:.1f} TF")
L.mvsnerf_tune(b"mlp_dbg", 0)
