"""Frame time of render_view (512x640, encode + ray march) against the sub-batch size inside mvsnerf_render_pixels_fwd."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from mvsnerf_amd import train
dev = 'cuda'
system = bench.load_system(dev)
batch = train.batch_to_device(train.synthetic_batch(512, 640, seed=1234), dev)
ref = None
for br in (1024, 4096, 16384, 65536, 327680):
    system.render_view(batch, batch_rays=br)
    torch.cuda.synchronize(); ts = []
    for rep in range(3):
        t0 = time.perf_counter(); rgb, depth = system.render_view(batch, batch_rays=br); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    if ref is None: ref = rgb
    print(f"batch_rays {br:6d}: {min(ts)*1e3:.2f} ms per frame, {512*640/min(ts)/1e6:.3f} M rays/s incl. encode, same pixels {torch.equal(rgb, ref)}")
