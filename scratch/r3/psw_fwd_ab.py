"""A/B of the two plane-sweep forward kernels (dev build): planesweep_kernel (psw_fwd_reuse 0) vs planesweep_columns_kernel (1).
They do the same fp32 operations in the same order, so the outputs must be BIT-identical in every layout; prints times."""
import ctypes, os, sys
sys.path.insert(0, '.')
import torch
from mvsnerf_amd import _lib
_lib.LIB_PATH = os.path.join('scratch', 'lib', 'libmvsnerf_hip_dev.so'); _lib._lib = None
_lib.SIGNATURES["mvsnerf_tune"] = (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int])
from mvsnerf_amd import encoder as E
from mvsnerf_amd.synth import make_rig
DEV = 'cuda'
L = _lib.lib()
ok = True
for (V, H, W, pad, D, with_img) in [(3, 30, 41, 3, 10, True), (5, 16, 24, 4, 19, True), (2, 32, 32, 0, 8, False), (4, 40, 33, 5, 37, True),
                                    (8, 20, 28, 2, 9, True), (3, 128, 160, 24, 128, True), (3, 128, 160, 24, 128, False)]:
    base = (0.0, 0.25, -0.25, 0.12, -0.12, 0.1, -0.3, 0.3, 0.2)
    rig = make_rig(H * 4, W * 4, n_views=V + 1, seed=77, baselines=base[:V + 1], rot_deg=2.0, smooth=True)
    proj = rig["proj_mats"][:, :V].contiguous().to(DEV)
    nf = rig["near_fars"][0, 0]
    depth = torch.linspace(float(nf[0]), float(nf[1]), D).to(DEV).unsqueeze(0)
    g = torch.Generator(DEV).manual_seed(V * 100 + D)
    feats = torch.randn((1, V, 32, H, W), device=DEV, generator=g)
    imgs = torch.rand((1, V, 3, H * 4, W * 4), device=DEV, generator=g)
    for blocked in ((False, True, "bf16") if with_img else (False,)):
        outs, ms = {}, {}
        for mode in (0, 1):
            assert L.mvsnerf_tune(b"psw_fwd_reuse", mode) == 0
            for rep in range(3):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                cost, masks, _ = E._plane_sweep(imgs if with_img else None, feats, proj, depth, pad, with_img, blocked=blocked)
                e1.record(); torch.cuda.synchronize()
            buf = cost.buf if hasattr(cost, "buf") else cost
            outs[mode] = (buf.clone().view(torch.int16 if buf.dtype == torch.bfloat16 else torch.int32), masks.clone())
            ms[mode] = e0.elapsed_time(e1)
        L.mvsnerf_tune(b"psw_fwd_reuse", 1)
        same = [bool((outs[0][0] == outs[m][0]).all()) and bool((outs[0][1] == outs[m][1]).all()) for m in (1,)]
        ok &= all(same)
        print(f"V={V} {D}x{H+2*pad}x{W+2*pad} img={with_img} blocked={blocked}: blocks {ms[0]:.3f} ms (whole _plane_sweep), waves {ms[1]:.3f} ms, bit-identical {same}")
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
