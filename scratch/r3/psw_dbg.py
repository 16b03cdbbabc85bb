"""Ablation of the plane-sweep forward kernel ("an ablation said why").  NOT runnable against the committed sources: it
drove a temporary `psw_dbg` bit mask that was patched into planesweep_kernel (the 64-column x 4-plane, 256-thread form of that moment) for
one measurement and removed again:
    bit 0: the flush keeps its LDS reads but skips the global stores        bit 1: no mask stores
    bit 2: phase 2 skips the source-view loop (no gathers, no blends)       bit 3: phase 1 skips the source-view loop (no divisions)
Measured (config 2, ms): 0 -> 0.360, 1 -> 0.326, 2 -> 0.348, 4 -> 0.221, 8 -> 0.314, 12 -> 0.192, 7 -> 0.202, 15 -> 0.173."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from mvsnerf_amd import _lib
_lib.LIB_PATH = os.path.join('scratch', 'lib', 'libmvsnerf_hip_dev.so'); _lib._lib = None
_lib.SIGNATURES["mvsnerf_tune"] = (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int])
from mvsnerf_amd import encoder as E
from mvsnerf_amd.synth import make_rig
DEV='cuda'; L=_lib.lib()
V,H,W,pad,D=3,128,160,24,128
rig = make_rig(H*4, W*4, n_views=V+1, seed=77, baselines=(0.0,0.25,-0.25,0.12), rot_deg=2.0, smooth=True)
proj = rig["proj_mats"][:, :V].contiguous().to(DEV)
nf = rig["near_fars"][0,0]
depth = torch.linspace(float(nf[0]), float(nf[1]), D).to(DEV).unsqueeze(0)
g = torch.Generator(DEV).manual_seed(1)
feats = torch.randn((1,V,32,H,W), device=DEV, generator=g)
imgs = torch.rand((1,V,3,H*4,W*4), device=DEV, generator=g)
from mvsnerf_amd.ops import stream_ptr
feats_cl = feats[0].permute(0,2,3,1).contiguous()
imgs_cl = torch.rand((V,H,W,4), device=DEV)
Hp,Wp=H+2*pad,W+2*pad
CP=44
cost=torch.empty((D,Hp,Wp,CP),device=DEV); masks=torch.empty((V,D,Hp,Wp),device=DEV)
pj=proj[0].contiguous(); dp=depth[0].contiguous()
for dbg in (0,1,2,3,4,8,12,15,7):
    L.mvsnerf_tune(b"psw_dbg", dbg)
    for blocked,fn in ((0,L.mvsnerf_planesweep_costvar_fwd),(1,L.mvsnerf_planesweep_costvar_blocked_fwd)):
        ts=[]
        for rep in range(5):
            torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); rc=fn(feats_cl.data_ptr(), imgs_cl.data_ptr(), pj.data_ptr(), dp.data_ptr(), V,32,H,W,D,pad,cost.data_ptr(),CP,masks.data_ptr(),1,stream_ptr()); e1.record(); torch.cuda.synchronize()
            assert rc==0; ts.append(e0.elapsed_time(e1))
        print(f"dbg={dbg:2d} blocked={blocked}: {min(ts):.3f} ms")
