"""Times the launch-shape / cache-hint variants of the stand-alone trilinear lookup (scratch/r3/vs_variants.hip) the way bench.py times the
product kernel: hipGraph replay of 40 launches, config-2 volume, 1024 x 128 random samples."""
import ctypes, sys, torch
sys.path.insert(0, '.')
import bench
dev = 'cuda'
L = ctypes.CDLL('scratch/lib/libvs_variants.so')
L.vs_variant.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator(dev).manual_seed(1)
D, H, W = 128, 176, 208
vol = torch.randn((D, H, W, 8), device=dev, generator=g)
P = 1024 * 128
ndc = torch.rand((P, 3), device=dev, generator=g)
out = torch.empty((P, 20), device=dev)
ref = None
for rnd in range(2):
    for v, name in ((0, "256 threads (product)"), (1, "128 threads"), (2, "64 threads"), (4, "512 threads"), (3, "256 + nontemporal"), (5, "128 + nontemporal")):
        fn = lambda: L.vs_variant(v, vol.data_ptr(), D, H, W, ndc.data_ptr(), P, out.data_ptr(), 20, torch.cuda.current_stream().cuda_stream)
        assert fn() == 0
        torch.cuda.synchronize()
        if ref is None: ref = out[:, :8].clone()
        same = torch.equal(out[:, :8], ref)
        t = bench.event_time(fn, 800, graph_batch=40)
        print(f"variant {v} {name:24s}: {t*1e3:.2f} us  frac {300*P/(t*1e-3)/1e9/8000:.4f}  same bits {same}")
