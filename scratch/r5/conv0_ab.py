"""Same-box A/B of the fp16x3 conv0 kernel: the shipped library against variant libraries (scratch/lib/libmvsnerf_hip_<name>.so, names on the command line),
config-2 volume, alternating launches, HIP-event time per launch (median of 5 rounds x 20 launches) and bit equality of the outputs."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
from mvsnerf_amd import _lib
from mvsnerf_amd import encoder as E
from mvsnerf_amd.ops import stream_ptr
from tests.test_gpu_bf16_encoder import _sweep_inputs
DEV = "cuda"
L = _lib.lib()
V, H, W, D, pad = 3, 128, 160, 128, 24
imgs, feats, proj, dv = _sweep_inputs(V, H, W, D, pad, seed=5)
cin = 3 * V + 32
with torch.no_grad():
    c16 = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked="fp16x2")[0]
    Dp, Hp, Wp = c16.dims
    w = torch.randn((8, cin, 3, 3, 3), device=DEV, generator=torch.Generator(DEV).manual_seed(1)) * 0.1
    pk = torch.empty(L.mvsnerf_conv0_f16x3_packed_elems(cin), device=DEV, dtype=torch.float16)
    assert L.mvsnerf_conv0_f16x3_pack(w.data_ptr(), cin, pk.data_ptr(), stream_ptr()) == 0
libs = [("shipped", L)]
for name in sys.argv[1:]:
    v = ctypes.CDLL(os.path.join(ROOT, "scratch", "lib", f"libmvsnerf_hip_{name}.so"))
    v.mvsnerf_conv0_f16x3_fwd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    v.mvsnerf_conv0_f16x3_fwd.restype = ctypes.c_int
    libs.append((name, v))
outs = {}
def run(lib, out):
    assert lib.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), cin, Dp, Hp, Wp, pk.data_ptr(), out.data_ptr(), 0, stream_ptr()) == 0
for name, lib in libs:
    outs[name] = torch.empty((Dp, Hp, Wp, 8), device=DEV)
    for _ in range(3): run(lib, outs[name])
torch.cuda.synchronize()
times = {name: [] for name, _ in libs}
for rnd in range(5):
    for name, lib in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(lib, outs[name])
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name, _ in libs:
    t = sorted(times[name])
    print(f"{name:12s} {t[2]:7.1f} us per launch (min {t[0]:.1f}, max {t[-1]:.1f})   equals shipped: {torch.equal(outs[name], outs['shipped'])}")
