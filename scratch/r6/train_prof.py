"""One generalizable-training run at config-3 shapes for rocprofv3 (2 warm-up + N timed steps).  usage: train_prof.py [amp] [steps]
MVS_LIB=<path of another build of the library> selects it (same-box A/B of a kernel change)."""
import sys, time, torch, os
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
from mvsnerf_amd import train, _lib, encoder
if os.environ.get('MVS_BF16_WGRAD') == '0':
    encoder.BF16_WGRAD = False
if os.environ.get('MVS_BF16_LAYERS') == '0':
    encoder.BF16_LAYERS = False
if os.environ.get('MVS_LIB'):
    _lib.LIB_PATH = os.environ['MVS_LIB']; _lib._lib = None
amp = len(sys.argv) > 1 and sys.argv[1] == "amp"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = 'cuda'
args = train.default_args(pad=24, batch_size=1024, N_samples=128, chunk=1024, use_amp=amp)
system = train.MVSSystem(args).to(dev)
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
system.render_kwargs_train["network_fn"].load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")})
system.MVSNet.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")})
batch = train.batch_to_device(train.synthetic_batch(512, 640, seed=1234), dev)
opt = system.configure_optimizers()[0][0]
torch.manual_seed(0)
system.fit_steps([batch] * 2, opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
system.fit_steps([batch] * steps, opt)
torch.cuda.synchronize(); print("train step ms (%s, bf16 layers %s, bf16 wgrad %s)" % ("use_amp" if amp else "fp32", encoder.BF16_LAYERS, encoder.BF16_WGRAD), (time.perf_counter() - t0) / steps * 1e3)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    system.fit_steps([batch] * steps, opt)
    torch.cuda.synchronize(); print("   again:", (time.perf_counter() - t0) / steps * 1e3)
