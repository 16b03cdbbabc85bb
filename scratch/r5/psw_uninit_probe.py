"""One library variant (MVS_LIB): the forward plane sweep (fp32 blocks / two fp16 planes / channel-last) quiet vs next to the fp16x3 conv0 on a second stream, and its quiet
output saved / compared with the shipped library's (argv[1] = 'save' | 'cmp', argv[2] = file)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsnerf_amd import _lib
if os.environ.get("MVS_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MVS_LIB"]); _lib._lib = None
from mvsnerf_amd import encoder as E
from mvsnerf_amd.ops import stream_ptr
from tests.test_gpu_bf16_encoder import _sweep_inputs
DEV = "cuda"
V, H, W, D, pad = 3, 128, 160, 128, 24
imgs, feats, proj, dv = _sweep_inputs(V, H, W, D, pad, seed=5)
L = _lib.lib()
cin = 3 * V + 32
bits = lambda c: (lambda t: t.view(torch.int16) if t.dtype == torch.float16 else t.contiguous().view(torch.int32))(c.buf if hasattr(c, "buf") else c)
out = {}
with torch.no_grad():
    c16 = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked="fp16x2")[0]
    Dp, Hp, Wp = c16.dims
    w = torch.randn((8, cin, 3, 3, 3), device=DEV, generator=torch.Generator(DEV).manual_seed(1)) * 0.1
    pk = torch.empty(L.mvsnerf_conv0_f16x3_packed_elems(cin), device=DEV, dtype=torch.float16)
    assert L.mvsnerf_conv0_f16x3_pack(w.data_ptr(), cin, pk.data_ptr(), stream_ptr()) == 0
    raw = torch.empty((Dp, Hp, Wp, 8), device=DEV)
    side = torch.cuda.Stream()
    for blocked in (True, "fp16x2", False):
        quiet = bits(E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked=blocked)[0]).clone()
        again = bits(E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked=blocked)[0])
        torch.cuda.synchronize()
        bad = 0
        for _ in range(24):
            with torch.cuda.stream(side):
                for _ in range(2):
                    assert L.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), cin, Dp, Hp, Wp, pk.data_ptr(), raw.data_ptr(), 0, stream_ptr()) == 0
            cur = bits(E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked=blocked)[0])
            bad += int((cur != quiet).any())
        torch.cuda.synchronize()
        q = quiet.float() if quiet.dtype != torch.int32 else quiet.view(torch.float32)
        out[str(blocked)] = quiet.cpu()
        print(f"  blocked={blocked}: quiet run reproduces itself {bool(torch.equal(quiet, again))}; {bad} of 24 aggressed sweeps differ; non-finite values in the quiet output {int((~torch.isfinite(q)).sum())}")
if sys.argv[1] == "save":
    torch.save(out, sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    for k in out:
        print(f"  blocked={k}: quiet output equals the shipped library's bit for bit: {bool(torch.equal(out[k], ref[k]))} ({int((out[k] != ref[k]).sum())} words differ)")
