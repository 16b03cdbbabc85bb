"""torch.profiler view of the use_amp training step: which Python lines launch the small copy / fill / elementwise kernels."""
import sys, torch
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
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
system.fit_steps([batch] * 2, opt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    system.fit_steps([batch] * 2, opt)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=8):
    if any(k in e.key for k in ("copy_", "fill_", "zero_", "aten::add", "aten::mul", "aten::sub", "aten::div", "aten::cat", "aten::stack", "aten::index", "aten::to",
                                "aten::sum", "aten::mean", "aten::pow", "aten::clone", "aten::contiguous", "aten::linspace", "aten::rand", "aten::log", "aten::clamp")):
        dev_t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if dev_t > 0:
            st = [s for s in e.stack if "mvsnerf_amd" in s or "torch/optim" in s][:3]
            rows.append((dev_t, e.count, e.key, st))
rows.sort(reverse=True)
tot = 0
for dev_t, cnt, key, st in rows[:60]:
    tot += dev_t
    print(f"{dev_t/2:9.1f} us/step  {cnt/2:5.1f} x  {key:28s} {' <- '.join(s.split('/')[-1][:70] for s in st)}")
print("sum of listed: %.1f us/step" % (tot / 2))
