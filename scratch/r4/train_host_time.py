"""Host (enqueue) time vs wall time of the use_amp training step: is the step GPU-bound or host-bound?  usage: train_host_time.py [amp|fp32]"""
import sys, time, gc, torch, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import numpy as np
from mvsnerf_amd import train
amp = len(sys.argv) < 2 or sys.argv[1] == "amp"
dev = 'cuda'
args = train.default_args(pad=24, batch_size=1024, N_samples=128, chunk=1024, use_amp=amp)
system = train.MVSSystem(args).to(dev)
z = np.load('tests/golden/mvsnerf_v0_weights.npz')
system.render_kwargs_train["network_fn"].load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")})
system.MVSNet.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")})
batch = train.batch_to_device(train.synthetic_batch(512, 640, seed=1234), dev)
opt = system.configure_optimizers()[0][0]
torch.manual_seed(0)
system.fit_steps([batch] * 3, opt)
for gc_off in (False, True):
    if gc_off: gc.collect(); gc.disable()
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        system.fit_steps([batch] * 10, opt)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{'amp' if amp else 'fp32'} gc_off={gc_off}: host enqueue {(t1 - t0) / 10 * 1e3:.3f} ms/step, wall {(t2 - t0) / 10 * 1e3:.3f} ms/step")
    gc.enable()
# where the host time goes
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); system.fit_steps([batch] * 10, opt); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
