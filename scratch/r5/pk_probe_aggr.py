"""Which property of the fp16x3 conv0 makes it the neighbour next to which `v_pk_*_f32 ... op_sel:[x,1]` returns wrong lanes?  The victims of pk_probe.py (modes 0 = plain
v_pk_fma_f32 as control, 4 = v_pk_fma_f32 op_sel:[0,1,0], 8 = v_pk_mul/add op_sel:[0,1]) next to variants of the aggressor built as link-swapped libraries
(scratch/lib/libmvsnerf_hip_<name>.so): no DPP epilogue, one workgroup per CU, no tile DMA, (almost) no MFMA; and next to the bf16 conv0 of the shipped library."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
from mvsnerf_amd import _lib
from mvsnerf_amd import encoder as E
from mvsnerf_amd.ops import stream_ptr
from tests.test_gpu_bf16_encoder import _sweep_inputs
DEV = "cuda"
P = ctypes.CDLL(os.path.join(ROOT, "scratch", "r5", "libpk_probe.so"))
P.pk_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
L = _lib.lib()
V, H, W, D, pad = 3, 128, 160, 128, 24
imgs, feats, proj, dv = _sweep_inputs(V, H, W, D, pad, seed=5)
cin = 3 * V + 32
with torch.no_grad():
    c16 = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked="fp16x2")[0]
    cb = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked="bf16")[0]
    Dp, Hp, Wp = c16.dims
    w = torch.randn((8, cin, 3, 3, 3), device=DEV, generator=torch.Generator(DEV).manual_seed(1)) * 0.1
    pk = torch.empty(L.mvsnerf_conv0_f16x3_packed_elems(cin), device=DEV, dtype=torch.float16)
    assert L.mvsnerf_conv0_f16x3_pack(w.data_ptr(), cin, pk.data_ptr(), stream_ptr()) == 0
    pkb = torch.empty(L.mvsnerf_conv0_bf16_packed_elems(cin), device=DEV, dtype=torch.bfloat16)
    assert L.mvsnerf_conv0_bf16_pack(w.data_ptr(), cin, pkb.data_ptr(), stream_ptr()) == 0
    raw = torch.empty((Dp, Hp, Wp, 8), device=DEV)
def conv0_of(lib):
    def f():
        for _ in range(2):
            assert lib.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), cin, Dp, Hp, Wp, pk.data_ptr(), raw.data_ptr(), 0, stream_ptr()) == 0
    return f
def conv0_bf16():
    for _ in range(3):
        assert L.mvsnerf_conv0_bf16_fwd(cb.buf.data_ptr(), cin, Dp, Hp, Wp, pkb.data_ptr(), raw.data_ptr(), 0, stream_ptr()) == 0
AGG = [("quiet", None), ("fp16x3 conv0 as shipped", conv0_of(L)), ("bf16 conv0 (16x16x32_bf16, 4 waves)", conv0_bf16)]
for name, what in (("nodpp", "conv0, shuffle instead of DPP"), ("onewg", "conv0, ONE workgroup per CU"), ("nodma", "conv0, no tile DMA"), ("nomfma", "conv0, 1 MFMA per group")):
    path = os.path.join(ROOT, "scratch", "lib", f"libmvsnerf_hip_{name}.so")
    if os.path.exists(path):
        v = ctypes.CDLL(path)
        v.mvsnerf_conv0_f16x3_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        v.mvsnerf_conv0_f16x3_fwd.restype = ctypes.c_int
        AGG.append((what, conv0_of(v)))
for _, f in AGG:
    if f: f()
torch.cuda.synchronize()
side = torch.cuda.Stream()
N_WG, ITERS = 73216, 40
for mode, name in ((0, "v_pk_fma_f32"), (4, "v_pk_fma op_sel:[0,1,0]"), (8, "v_pk_mul / add op_sel:[0,1]")):
    for aname, agg in AGG:
        lanes = torch.zeros(64, device=DEV, dtype=torch.int32)
        total = torch.zeros(1, device=DEV, dtype=torch.int64)
        torch.cuda.synchronize()
        for _ in range(24):
            if agg:
                with torch.cuda.stream(side):
                    agg()
            assert P.pk_probe_launch(mode, N_WG, ITERS, lanes.data_ptr(), total.data_ptr(), stream_ptr()) == 0
        torch.cuda.synchronize()
        per = lanes.cpu().tolist()
        print(f"{name:30s} | {aname:40s} | {int(total.item()):8d} mismatches, lanes 0-15 / 16-31 / 32-47 / 48-63: {[sum(per[16 * q:16 * q + 16]) for q in range(4)]}")
