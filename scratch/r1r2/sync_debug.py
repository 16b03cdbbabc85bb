"""Lists every host synchronisation inside a generalizable-training step (torch.cuda.set_sync_debug_mode) and the host-side time of
a step when nothing waits for the device."""
import sys, time, warnings, torch
sys.path.insert(0, '.')
import numpy as np
from mvsnerf_amd import train
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
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    opt.zero_grad(set_to_none=True)
    out = system.training_step(batch, 0)
    out["loss"].backward()
    opt.step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(len(w), "synchronising calls in one step")
for x in w:
    print("  ", x.filename.split("/repo/")[-1], x.lineno, str(x.message)[:90])
# host time per step: enqueue 6 steps without reading anything back
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(6):
    opt.zero_grad(set_to_none=True)
    out = system.training_step(batch, i)
    out["loss"].backward()
    opt.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue time {(t1 - t0) / 6 * 1e3:.2f} ms/step, device-complete {(t2 - t0) / 6 * 1e3:.2f} ms/step")
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for i in range(6):
    opt.zero_grad(set_to_none=True)
    out = system.training_step(batch, i)
    out["loss"].backward()
    opt.step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28)
print(st.getvalue()[:6000])
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(40)
print(st.getvalue()[:8000])
