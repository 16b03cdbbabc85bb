"""Same-box A/B of the forward plane sweep: the shipped library against variant libraries (scratch/lib/libmvsnerf_hip_<name>.so, names on the command line) - the variant's
entry points are swapped into the loaded library object, so both go through encoder._plane_sweep.  Config-2 size, every cost-volume layout: HIP-event time per launch
(median of 5 rounds x 20) and bit equality of the cost volume and the masks."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
from mvsnerf_amd import _lib
from mvsnerf_amd import encoder as E
from tests.test_gpu_bf16_encoder import _sweep_inputs
L = _lib.lib()
V, H, W, D, pad = 3, 128, 160, 128, 24
imgs, feats, proj, dv = _sweep_inputs(V, H, W, D, pad, seed=5)
ENTRIES = ["mvsnerf_planesweep_costvar_fwd", "mvsnerf_planesweep_costvar_blocked_fwd", "mvsnerf_planesweep_costvar_bf16_fwd", "mvsnerf_planesweep_costvar_f16x2_fwd"]
orig = {e: getattr(L, e) for e in ENTRIES}
variants = {"shipped": orig}
for name in sys.argv[1:]:
    v = ctypes.CDLL(os.path.join(ROOT, "scratch", "lib", f"libmvsnerf_hip_{name}.so"))
    fns = {}
    for e in ENTRIES:
        f = getattr(v, e); f.argtypes = orig[e].argtypes; f.restype = orig[e].restype; fns[e] = f
    variants[name] = fns
def use(fns):
    for e in ENTRIES: setattr(L, e, fns[e])
def bits(c):
    t = c.buf if hasattr(c, "buf") else c
    return t.view(torch.int16) if t.dtype in (torch.float16, torch.bfloat16) else t.contiguous().view(torch.int32)
for blocked in ("fp16x2", True, "bf16", False):
    ref = None
    for name, fns in variants.items():
        use(fns)
        with torch.no_grad():
            for _ in range(3): out = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked=blocked)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): out = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked=blocked)
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        cur = (bits(out[0]).clone(), out[1].clone())
        if ref is None: ref = cur
        ts.sort()
        print(f"layout {str(blocked):7s} {name:10s} {ts[2]:7.1f} us per sweep (incl. allocation; min {ts[0]:.1f})   bit-identical to shipped: {torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])}")
use(orig)
