"""Which gradients of a use_amp training step change when 16-bit MFMA waves of a second stream share the GPU?  Per parameter, in module order
(MLP, FeatureNet, CostRegNet): deviation of 3 aggressed steps from a quiet step (relative to the tensor's maximum) next to the quiet run-to-run spread.
Also the intermediate gradients the encoder backward hands from stage to stage (hooks on the volume and on FeatureNet's output)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsnerf_amd import _lib, train, ops
from mvsnerf_amd import encoder as E
from mvsnerf_amd.ops import stream_ptr
from tests.test_gpu_bf16_encoder import _sweep_inputs
from tests.test_gpu_train import _system
DEV = "cuda"
amp = (sys.argv[1] != "fp32") if len(sys.argv) > 1 else True
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
sys_, args, _, _ = _system(8, 512, 64, 32)
args.use_amp = amp
batch = train.synthetic_batch(128, 160, seed=3, rot_deg=2.0, smooth=True)
params = [(n, p) for n, p in list(sys_.render_kwargs_train["network_fn"].named_parameters()) + list(sys_.MVSNet.named_parameters())]
E.PSW_BWD_DETERMINISTIC = True
inter = {}
# intermediate gradients: the volume the encoder returns (d loss / d volume) and FeatureNet's output
orig_forward = sys_.MVSNet.forward
def fwd(*a, **k):
    out = orig_forward(*a, **k)
    if out[0].requires_grad:
        out[0].register_hook(lambda g: inter.__setitem__("d_volume", g.detach().clone()))
    return out
sys_.MVSNet.forward = fwd
orig_feat = sys_.MVSNet.feature.forward
def ffwd(*a, **k):
    o = orig_feat(*a, **k)
    if torch.is_tensor(o) and o.requires_grad:
        o.register_hook(lambda g: inter.__setitem__("d_feats", g.detach().clone()))
    return o
sys_.MVSNet.feature.forward = ffwd

def grads(aggress):
    for _, p in params:
        p.grad = None
    inter.clear()
    torch.manual_seed(11)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    if aggress:
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(aggress):
                assert L.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), cin, Dp, Hp, Wp, pk.data_ptr(), raw.data_ptr(), 0, stream_ptr()) == 0
    out = sys_.training_step(batch, 0)
    out["loss"].backward()
    torch.cuda.synchronize()
    d = {n: p.grad.detach().clone() for n, p in params}
    d.update({"[" + k + "]": v for k, v in inter.items()})
    return float(out["loss"].detach()), d

rel = lambda a, b: float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30)
l0, g0 = grads(0)
l1, g1 = grads(0)
runs = [grads(12) for _ in range(3)]
print(f"use_amp={amp}  loss quiet {l0:.8f} {l1:.8f}  aggressed {[round(r[0], 8) for r in runs]}")
order = [k for k in g0 if k.startswith("[")] + [n for n, _ in params]
for n in order:
    dev = [rel(g0[n], r[1][n]) for r in runs]
    flag = "  <<<" if max(dev) > 10 * rel(g0[n], g1[n]) + 1e-6 else ""
    print(f"{n:45s} shape {str(tuple(g0[n].shape)):22s} quiet {rel(g0[n], g1[n]):.1e}  aggressed " + " ".join(f"{d:.1e}" for d in dev) + flag)
# where in the tensor: for the worst parameter, how many elements moved and where
worst = max((n for n, _ in params), key=lambda n: max(rel(g0[n], r[1][n]) for r in runs))
dd = (g0[worst] - runs[0][1][worst]).abs() / g0[worst].abs().max()
print("worst:", worst, "elements over 1e-5:", int((dd > 1e-5).sum()), "of", dd.numel(), "max", float(dd.max()))
