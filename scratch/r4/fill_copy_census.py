"""Which Python lines of the use_amp training step launch the fill / copy kernels (torch.profiler, shapes + stacks of the raw events)."""
import sys, os, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
from mvsnerf_amd import train
from torch.profiler import profile, ProfilerActivity
dev = 'cuda'
args = train.default_args(pad=24, batch_size=1024, N_samples=128, chunk=1024, use_amp=True)
system = train.MVSSystem(args).to(dev)
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
system.render_kwargs_train["network_fn"].load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")})
system.MVSNet.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")})
batch = train.batch_to_device(train.synthetic_batch(512, 640, seed=1234), dev)
opt = system.configure_optimizers()[0][0]
torch.manual_seed(0)
system.fit_steps([batch] * 3, opt)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    system.fit_steps([batch] * N, opt)
    torch.cuda.synchronize()
names = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy",
         "aten::empty_strided", "aten::cat", "aten::stack", "aten::sum", "aten::add", "aten::mul", "aten::div", "aten::sub", "aten::mean", "aten::index", "aten::select",
         "aten::item", "aten::_local_scalar_dense", "aten::linspace", "aten::pow", "aten::log", "aten::clamp_min", "aten::where", "aten::lt", "aten::gt", "aten::ne", "aten::eq")
agg = collections.Counter(); dt = collections.Counter()
for e in prof.events():
    if e.name in names and e.device_time_total > 0 and e.self_device_time_total > 0:
        st = [s for s in (e.stack or []) if "mvsnerf_amd" in s or "/optim/" in s or "bench.py" in s][:2]
        key = (e.name, str(e.input_shapes)[:60], " <- ".join(s.split("/")[-1][:80] for s in st))
        agg[key] += 1; dt[key] += e.self_device_time_total
tot = 0
for key, t in sorted(dt.items(), key=lambda kv: -kv[1])[:70]:
    tot += t
    print(f"{t / N:8.1f} us/step {agg[key] / N:5.1f} x  {key[0]:18s} {key[1]:60s} {key[2]}")
print("sum %.1f us/step" % (tot / N))
# memcpy / memset events from the device side
km = collections.Counter(); kt = collections.Counter()
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA") and ("Memcpy" in e.name or "Memset" in e.name or "copyBuffer" in e.name or "fillBuffer" in e.name):
        km[e.name] += 1; kt[e.name] += e.device_time_total
for k in km: print(f"device {k}: {km[k] / N} per step, {kt[k] / N:.1f} us/step")
