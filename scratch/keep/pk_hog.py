"""scratch/keep/pk_probe.hip's victims (v_pk_fma_f32 op_sel:[0,1,0] / v_pk_mul+add op_sel:[0,1] against their scalar equivalents) next to aggressors that have NO matrix, LDS or memory
instruction: waves that only hold registers / wave slots of a SIMD (scratch/keep/pk_hog.hip).  Is the neighbour's ACTIVITY the condition, or where the victim's registers land?
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared scratch/keep/pk_hog.hip -o scratch/keep/libpk_hog.so   (and pk_probe.hip likewise)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
DEV = "cuda"
P = ctypes.CDLL(os.path.join(ROOT, "scratch", "r5", "libpk_probe.so"))
P.pk_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
Hg = ctypes.CDLL(os.path.join(ROOT, "scratch", "r5", "libpk_hog.so"))
Hg.pk_hog_launch.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, device=DEV)
def sp(): return torch.cuda.current_stream().cuda_stream
def hog(nv, mode, n_wg, waves=1, lds=0, iters=None):
    it = iters or (150 if mode == 0 else 1000)
    def f():
        rc = Hg.pk_hog_launch(nv, mode, n_wg, waves, lds, it, sink.data_ptr(), sp())
        assert rc == 0, rc
    return f
AGG_OCC = [("quiet", None),
       ("128 VGPRs, asleep, 1 wave / SIMD", hog(128, 0, 1024)),
       ("128 VGPRs, asleep, 2 waves / SIMD", hog(128, 0, 2048)),
       ("128 VGPRs, asleep, 3 waves / SIMD", hog(128, 0, 3072)),
       ("256 VGPRs, asleep, 1 wave / SIMD", hog(256, 0, 1024)),
       ("64 VGPRs, asleep, 4 waves / SIMD", hog(64, 0, 4096)),
       ("64 VGPRs, asleep, 6 waves / SIMD", hog(64, 0, 6144)),
       ("few VGPRs, asleep, 4 waves / SIMD", hog(32, 0, 4096)),
       ("128 VGPRs, v_add_f32 spin, 3 waves / SIMD", hog(128, 1, 3072)),
       ("128 VGPRs, MFMA 16x16x32 f16 spin, 3 / SIMD", hog(128, 2, 3072)),
       ("few VGPRs, MFMA 16x16x32 f16 spin, 4 / SIMD", hog(32, 2, 4096)),
       ("128 VGPRs, asleep, workgroups of 8 waves, 3 / SIMD", hog(128, 0, 384, waves=8)),
       ("few VGPRs, asleep, 74 KB LDS / workgroup (2 per CU)", hog(32, 0, 512, lds=74752))]
AGG_MFMA = [("quiet", None)] + [(f"{nm} spin, 4 waves / SIMD", hog(32, md, 4096)) for md, nm in
            ((2, "v_mfma_f32_16x16x32_f16"), (6, "v_mfma_f32_16x16x32_bf16"), (8, "v_mfma_f32_16x16x16_f16"), (3, "v_mfma_f32_32x32x16_f16"), (4, "v_mfma_f32_32x32x2_f32"),
             (5, "v_mfma_f32_16x16x4_f32"), (7, "v_mfma_f32_4x4x4_16B_f16"))] + [("v_mfma_f32_16x16x32_f16 spin, 1 wave / SIMD", hog(32, 2, 1024)), ("v_mfma_f32_16x16x32_f16 spin, ONE wave per CU", hog(32, 2, 256))]
AGG = AGG_MFMA if "mfma" in sys.argv[1:] else AGG_OCC
for _, f in AGG:
    if f: f()
torch.cuda.synchronize()
side = torch.cuda.Stream()
N_WG, ITERS = 73216, 40
for mode, name in ((4, "v_pk_fma op_sel:[0,1,0]"), (8, "v_pk_mul / add op_sel:[0,1]"), (0, "v_pk_fma_f32 (control)"), (6, "v_pk_fma op_sel:[1,0,0]"), (3, "v_pk_fma op_sel_hi:[1,0,1]")):
    for aname, agg in AGG:
        lanes = torch.zeros(64, device=DEV, dtype=torch.int32)
        total = torch.zeros(1, device=DEV, dtype=torch.int64)
        torch.cuda.synchronize()
        for _ in range(12):
            if agg:
                with torch.cuda.stream(side):
                    agg()
            assert P.pk_probe_launch(mode, N_WG, ITERS, lanes.data_ptr(), total.data_ptr(), sp()) == 0
        torch.cuda.synchronize()
        per = lanes.cpu().tolist()
        print(f"{name:28s} | {aname:52s} | {int(total.item()):9d} mismatches, lanes 0-15 / 16-31 / 32-47 / 48-63: {[sum(per[16 * q:16 * q + 16]) for q in range(4)]}")
