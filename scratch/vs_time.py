import sys, torch
sys.path.insert(0,'.')
from mvsnerf_amd import _lib, ops
dev='cuda'
g = torch.Generator().manual_seed(0)
D,h,w = 128,176,208
vol = torch.randn((D,h,w,8), generator=g).to(dev)
# realistic rays: random pixel, samples marching in z with slight xy drift
N,S = 1024,128
xy0 = torch.rand((N,1,2), generator=g)*0.8+0.1
drift = (torch.rand((N,1,2), generator=g)-0.5)*0.1
t = torch.linspace(0,1,S).view(1,S,1)
ndc = torch.cat([xy0 + drift*t, t.expand(N,S,1)*0.98+0.01], -1).contiguous().to(dev)
feat = torch.empty((N,S,20), device=dev)
def timeit(fn, iters=300):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
ref=None
for spq in (1,2,3,4):
    _lib.lib().mvsnerf_tune(b"vs_spq", spq)
    o = ops.volume_sample(vol, ndc)
    if ref is None: ref=o
    us = timeit(lambda: ops.volume_sample(vol, ndc, out=feat, out_stride=20))
    us8 = timeit(lambda: ops.volume_sample(vol, ndc))
    print(f"spq={spq}: {us:.2f} us (stride20)  {us8:.2f} us (dense out) -> {300*N*S/us/1e6:.2f} TB/s eff; maxdiff={float((o-ref).abs().max()):.1e}")
