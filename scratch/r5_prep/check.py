"""Round-5 prototype check (GPU): scratch/r5_prep/libr5.so's fp32-grade fp16 LDS-tiled convolution (conv_f16x3_tiled.hip) at the shapes of conv1 / conv2 of the
config-2 scene encode against the float64 convolution of the same operands, next to torch's fp32 convolution as the yardstick for "fp32 grade"; InPlaceABN
partial sums; the guard; kernel time (HIP events, 30 launches).  usage: python scratch/r5_prep/check.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import torch.nn.functional as F

L = ctypes.CDLL(os.path.join(ROOT, "scratch", "r5_prep", os.environ.get("R5_LIB", "libr5.so")))
vp, i32 = ctypes.c_void_p, ctypes.c_int
L.r5_conv_f16x3_tiled_packed_elems.restype = ctypes.c_size_t
L.r5_conv_f16x3_tiled_packed_elems.argtypes = [i32, i32]
L.r5_conv_f16x3_tiled_pack.argtypes = [vp, i32, i32, i32, vp, vp]
L.r5_conv_f16x3_tiled_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i32, vp, vp, i32, vp, i32, vp]
dev = torch.device("cuda")
st = lambda: torch.cuda.current_stream().cuda_stream


def run(Cin, Cout, stride, dims, seed, lazy=True, grid=512, scale_x=1.0):
    D, H, W = dims
    g = torch.Generator(dev).manual_seed(seed)
    x = torch.randn((D, H, W, Cin), device=dev, generator=g) * scale_x
    sc = (torch.rand(Cin, device=dev, generator=g) + 0.5) if lazy else None
    sh = (torch.randn(Cin, device=dev, generator=g) * 0.3) if lazy else None
    w = torch.randn((Cout, Cin, 3, 3, 3), device=dev, generator=g) * 0.1
    pk = torch.zeros(L.r5_conv_f16x3_tiled_packed_elems(Cin, 27), device=dev, dtype=torch.float16)
    assert L.r5_conv_f16x3_tiled_pack(w.data_ptr(), Cin, Cout, 27, pk.data_ptr(), st()) == 0
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.full((Do, Ho, Wo, Cout), float("nan"), device=dev)
    nslots = abs(grid) + 3
    part = torch.full((2 * Cout * nslots,), float("nan"), device=dev)
    guard = torch.zeros(4, device=dev, dtype=torch.int32)
    call = lambda: L.r5_conv_f16x3_tiled_fwd(x.data_ptr(), sc.data_ptr() if lazy else None, sh.data_ptr() if lazy else None, Cin, Cin, D, H, W, pk.data_ptr(), Cout,
                                             stride, out.data_ptr(), part.data_ptr(), nslots, guard.data_ptr(), grid, st())
    assert call() == 0
    torch.cuda.synchronize()
    # float64 reference of the same operands (activation in float64 from the fp32 raw values), and torch's fp32 convolution as the yardstick
    xa = x.double()
    if lazy:
        xa = xa * sc.double() + sh.double()
        xa = torch.where(xa > 0, xa, 0.01 * xa)
    # (on a slab of the first planes when the volume is large: the float64 convolution is an im2col fallback)
    zo = min(Do, 16 if stride == 1 else 8)
    zs = min(D, zo * stride + 1)
    xin = xa[:zs].permute(3, 0, 1, 2)[None]
    ref = F.conv3d(xin, w.double(), stride=stride, padding=1)[0].permute(1, 2, 3, 0)[:zo]
    y32 = F.conv3d(xin.float(), w, stride=stride, padding=1)[0].permute(1, 2, 3, 0)[:zo]
    top = float(ref.abs().max())
    e16 = float((out[:zo].double() - ref).abs().max()) / top
    e32 = float((y32.double() - ref).abs().max()) / top
    # partial sums: slots of this grid hold the per-workgroup sums, the others zeros
    ps = part.view(2, Cout, nslots)
    s_err = float((ps[0].double().sum(1) - out.double().sum((0, 1, 2))).abs().max() / out.double().abs().sum((0, 1, 2)).max())
    q_err = float((ps[1].double().sum(1) - (out.double() ** 2).sum((0, 1, 2))).abs().max() / (out.double() ** 2).sum((0, 1, 2)).max())
    # time
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        call()
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    flop = 2.0 * Do * Ho * Wo * Cout * Cin * 27
    print(f"Cin {Cin} Cout {Cout} stride {stride} dims {dims} lazy {lazy}: max err / max|ref| fp16x3 {e16:.2e}  torch fp32 {e32:.2e}  (max|ref| {top:.1f}); "
          f"NaNs left {int(torch.isnan(out).sum())}; partial sums rel err {s_err:.1e} / {q_err:.1e}; guard {guard.tolist()}; {us:.1f} us = {flop / us / 1e6:.1f} TFLOP/s fp32-equivalent")
    return e16, e32


if os.environ.get("R5_GRIDS"):                   # timing only: the two encode shapes at several grid sizes
    for gsz in [int(t) for t in os.environ["R5_GRIDS"].split(",")]:          # a negative size selects the 16-wide tiles
        print("grid", gsz)
        run(16, 16, 1, (64, 88, 104), 3, grid=gsz)
        run(8, 16, 2, (128, 176, 208), 4, grid=gsz)
    sys.exit(0)
run(16, 16, 1, (8, 12, 40), 1)                    # small, ragged against the 2 x 4 x 32 tile
run(8, 16, 2, (10, 14, 70), 2)
run(16, 16, 1, (64, 88, 104), 3)                  # conv2 of the config-2 encode
run(8, 16, 2, (128, 176, 208), 4)                 # conv1
run(16, 8, 1, (64, 88, 104), 5, lazy=False)       # 8 output channels (half of the columns idle), no pending activation
run(16, 16, 1, (16, 24, 64), 6, scale_x=5000.0)   # |x 2^4| beyond fp16: the guard must trip
# the 16-wide tiles (negative grid): correctness at ragged and full shapes, then their time
run(16, 16, 1, (8, 12, 40), 1, grid=-64)
run(8, 16, 2, (10, 14, 70), 2, grid=-64)
run(16, 16, 1, (64, 88, 104), 3, grid=-512)
run(8, 16, 2, (128, 176, 208), 4, grid=-512)
