import sys, torch
sys.path.insert(0,'.')
import numpy as np
from collections import defaultdict
from mvsnerf_amd import _lib, ops, models
dev='cuda'
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('mlp/')}
net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type='v0'); net.load_state_dict(sd); net=net.to(dev)
N,S,F=1024,128,20
g=torch.Generator().manual_seed(0)
ndc=torch.rand((N,S,3),generator=g).to(dev); feat=torch.randn((N,S,F),generator=g).to(dev); dirs=torch.randn((N,3),generator=g).to(dev)
packed=net.packed(F)
L=_lib.lib(); L.mvsnerf_tune(b"mlp_variant", 3)
f=lambda: ops.mlp_forward(packed,F,ndc.data_ptr(),3,feat.data_ptr(),F,dirs.data_ptr(),3,N,S,False,dev)
for _ in range(3): f()
cen = torch.zeros((N,16), dtype=torch.int64, device=dev)
L.mvsnerf_debug_set_census(cen.data_ptr())
f(); torch.cuda.synchronize()
L.mvsnerf_debug_set_census(0)
c = cen.cpu().numpy()
t0 = c[:,0].min()
start = (c[:,0]-t0)/100.0; end = (c[:,1]-t0)/100.0   # us
hw = c[:,2]; xcc = c[:,3] & 0xf
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
print("kernel span %.1f us; WG duration mean %.1f min %.1f max %.1f us" % (end.max(), (end-start).mean(), (end-start).min(), (end-start).max()))
print("start times: first-wave (<5us) count", int((start<5).sum()), " second wave starts: min %.1f mean %.1f max %.1f" % (start[start>=5].min(), start[start>=5].mean(), start[start>=5].max()))
slots = defaultdict(list)
for i in range(N): slots[(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]))].append((start[i], end[i], i))
print("distinct CU ids seen:", len(slots), " WGs per CU: min %d max %d" % (min(len(v) for v in slots.values()), max(len(v) for v in slots.values())))
# concurrency on one CU
k0 = sorted(slots)[0]
print("CU", k0, sorted([(round(a,1), round(b,1), i) for a,b,i in slots[k0]]))
k1 = sorted(slots)[len(slots)//2]
print("CU", k1, sorted([(round(a,1), round(b,1), i) for a,b,i in slots[k1]]))
# max concurrency per CU
mc = []
for v in slots.values():
    ev = sorted([(a,1) for a,b,i in v]+[(b,-1) for a,b,i in v]); cur=0; m=0
    for _,d in ev: cur+=d; m=max(m,cur)
    mc.append(m)
print("max concurrent WGs per CU: histogram", {k: mc.count(k) for k in set(mc)})

names = ["startup","bias gemm","layer0","layer1","layer2","layer3","layer4","layer5+sigma","feature","views+rgb"]
def phases(rows):
    pts = np.concatenate([rows[:, 0:1], rows[:, 4:13], rows[:, 1:2]], 1).astype(np.float64) / 100.0
    return np.diff(pts, axis=1).mean(0)
ph2 = phases(c)
# a run with one WG per CU only (256 WGs): same kernel alone
N1=256
ndc1, feat1, dirs1 = ndc[:N1].contiguous(), feat[:N1].contiguous(), dirs[:N1].contiguous()
cen1 = torch.zeros((N1,16), dtype=torch.int64, device=dev)
f1=lambda: ops.mlp_forward(packed,F,ndc1.data_ptr(),3,feat1.data_ptr(),F,dirs1.data_ptr(),3,N1,S,False,dev)
for _ in range(3): f1()
L.mvsnerf_debug_set_census(cen1.data_ptr()); f1(); torch.cuda.synchronize(); L.mvsnerf_debug_set_census(0)
ph1 = phases(cen1.cpu().numpy())
mf = [40*64, 128*64, 256*64, 256*64, 256*64, 256*64, 384*64, 256*64, 132*64]
print("%-14s %10s %10s %12s" % ("phase", "alone us", "paired us", "MFMA-only us@2.3GHz"))
for i,n in enumerate(names):
    m = (mf[i-1]/2300.0) if i>0 else 0
    print("%-14s %10.2f %10.2f %12.2f" % (n, ph1[i], ph2[i], m))
print("total alone %.1f paired %.1f" % (ph1.sum(), ph2.sum()))

for nm, cc in (("paired (1024 WGs)", c), ("alone (256 WGs)", cen1.cpu().numpy())):
    ticks = cc[:,13].astype(np.float64); dur = (cc[:,1]-cc[:,0]).astype(np.float64)/100.0
    print(nm, "shader clock during the kernel: mean %.3f GHz (min %.3f max %.3f)" % ((ticks/dur/1e3).mean(), (ticks/dur/1e3).min(), (ticks/dur/1e3).max()))
