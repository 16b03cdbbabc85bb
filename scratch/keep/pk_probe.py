"""scratch/keep/pk_probe.hip on a GPU box: the packed-vs-scalar recurrence of every packed fp32 form, quiet and next to one aggressor kernel on a second stream
(fp16x3 conv0, fp32 / bf16 / fp16x3 MLP).  Result: profiles/r05_pk_fma_opsel_reproducer.txt.
Build (here, cross-compiled):  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared scratch/keep/pk_probe.hip -o scratch/keep/libpk_probe.so
Run:  gpurun -- 'python scratch/keep/pk_probe.py'  (about 15 s of GPU time)"""
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
    Dp, Hp, Wp = c16.dims
    w = torch.randn((8, cin, 3, 3, 3), device=DEV, generator=torch.Generator(DEV).manual_seed(1)) * 0.1
    pk = torch.empty(L.mvsnerf_conv0_f16x3_packed_elems(cin), device=DEV, dtype=torch.float16)
    assert L.mvsnerf_conv0_f16x3_pack(w.data_ptr(), cin, pk.data_ptr(), stream_ptr()) == 0
    raw = torch.empty((Dp, Hp, Wp, 8), device=DEV)
side = torch.cuda.Stream()
N_WG, ITERS = 73216, 40          # the plane sweep's grid at config 2; 40 x 16 packed instructions per lane ~ 0.3 ms per launch
# aggressors on the second stream: the fp16x3 conv0 (v_mfma_f32_16x16x32_f16), the fp32 MLP (v_mfma_f32_32x32x2_f32), the bf16 MLP (v_mfma_f32_32x32x16_bf16), the fp16x3 MLP (32x32x16_f16)
from tests.util import load_weights
from mvsnerf_amd import models as M, ops
mlp_sd, _ = load_weights()
net = M.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
net.load_state_dict(mlp_sd); net = net.to(DEV)
g = torch.Generator().manual_seed(0)
ndc = torch.rand((1024, 128, 3), generator=g).to(DEV); featm = (torch.randn((1024, 128, 20), generator=g) * 0.3).to(DEV)
dirs = torch.nn.functional.normalize(torch.randn((1024, 3), generator=g), dim=-1).to(DEV)
def agg_conv0():
    for _ in range(2):
        assert L.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), cin, Dp, Hp, Wp, pk.data_ptr(), raw.data_ptr(), 0, stream_ptr()) == 0
def agg_mlp(mode):
    def f():
        with ops.mlp_precision(mode), torch.no_grad():
            for _ in range(3):
                net.nerf.query(ndc, featm, dirs, 1024, 128)
    return f
AGG = [("quiet", None), ("fp16x3 conv0 (16x16x32_f16)", agg_conv0), ("fp32 MLP (32x32x2_f32)", agg_mlp("fp32")), ("bf16 MLP (32x32x16_bf16)", agg_mlp("bf16")), ("fp16x3 MLP (32x32x16_f16)", agg_mlp("fp16x3"))]
for f in AGG:
    if f[1]: f[1]()
torch.cuda.synchronize()
MODES = ((0, "v_pk_fma_f32"), (1, "v_pk_mul_f32 + v_pk_add_f32"), (2, "v_pk_mul op_sel_hi:[1,0] + add"), (3, "v_pk_fma op_sel_hi:[1,0,1]"), (4, "v_pk_fma op_sel:[0,1,0]"),
         (5, "v_pk_mul / add neg_lo neg_hi"), (6, "v_pk_fma op_sel:[1,0,0]"), (7, "v_pk_fma op_sel:[0,0,1]"), (8, "v_pk_mul / add op_sel:[0,1]"))
for mode, name in MODES:
    row = []
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
        row.append(f"{aname}: {int(total.item())} {[sum(per[16 * q:16 * q + 16]) for q in range(4)]}")
    print(f"{name:32s} mismatching checks of {24 * N_WG * 64 * ITERS} [lanes 0-15, 16-31, 32-47, 48-63] | " + " | ".join(row))
