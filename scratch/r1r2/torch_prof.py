"""torch.profiler view of one training step: which aten ops launch the small copy / fill kernels."""
import sys, torch
sys.path.insert(0, '.')
import numpy as np
from mvsnerf_amd import train
from torch.profiler import profile, ProfilerActivity
dev = 'cuda'
args = train.default_args(pad=24, batch_size=1024, N_samples=128, chunk=1024)
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
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=50, max_src_column_width=110))
