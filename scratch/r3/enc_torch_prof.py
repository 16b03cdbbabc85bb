"""Which aten ops / memcpys does one scene encode issue (looking for the ~67 __amd_rocclr_copyBuffer launches per encode)?"""
import sys, torch
sys.path.insert(0, '.')
from mvsnerf_amd import encoder
from mvsnerf_amd.synth import make_rig
from torch.profiler import profile, ProfilerActivity
rig = make_rig(512, 640, seed=1234)
encoder.bench_encode(rig, torch.device('cuda'), 24, iters=2)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    encoder.bench_encode(rig, torch.device('cuda'), 24, iters=1)
    torch.cuda.synchronize()
ka = prof.key_averages()
for e in sorted(ka, key=lambda e: -e.count)[:40]:
    print(f"{e.count:5d} x  cpu {e.cpu_time_total:9.1f} us  dev {getattr(e, 'device_time_total', 0):9.1f} us  {e.key[:90]}")
