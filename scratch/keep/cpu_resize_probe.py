"""F.interpolate(bilinear, align_corners=False) on the CPU oracle: which fp32 formula?  (models.py:859)"""
import numpy as np, torch, torch.nn.functional as F
f32 = np.float32
def fma(a, b, c): return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)
torch.manual_seed(0)
for (Hi, Wi, Ho, Wo) in ((512, 640, 128, 160), (64, 96, 16, 24), (800, 800, 200, 200), (37, 53, 9, 13)):
    x = torch.rand(2, 3, Hi, Wi)
    ref = F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=False).numpy()
    src = x.numpy()
    sh, sw = f32(Hi) / f32(Ho), f32(Wi) / f32(Wo)
    ys = np.maximum((sh * (np.arange(Ho, dtype=f32) + f32(0.5))).astype(f32) - f32(0.5), f32(0)).astype(f32)
    xs = np.maximum((sw * (np.arange(Wo, dtype=f32) + f32(0.5))).astype(f32) - f32(0.5), f32(0)).astype(f32)
    y0 = ys.astype(np.int64); x0 = xs.astype(np.int64); y1 = np.minimum(y0 + 1, Hi - 1); x1 = np.minimum(x0 + 1, Wi - 1)
    ly = (ys - y0).astype(f32)[:, None]; lx = (xs - x0).astype(f32)[None, :]; hy = (f32(1) - ly).astype(f32); hx = (f32(1) - lx).astype(f32)
    a = src[:, :, y0][:, :, :, x0]; b = src[:, :, y0][:, :, :, x1]; c = src[:, :, y1][:, :, :, x0]; d = src[:, :, y1][:, :, :, x1]
    B = lambda w: np.broadcast_to(w, a.shape).astype(f32)
    cands = {
        "hy*(hx*a+lx*b)+ly*(hx*c+lx*d) separate": ((B(hy) * ((B(hx) * a).astype(f32) + (B(lx) * b).astype(f32)).astype(f32)).astype(f32) + (B(ly) * ((B(hx) * c).astype(f32) + (B(lx) * d).astype(f32)).astype(f32)).astype(f32)).astype(f32),
        "fma inner+outer": fma(B(ly), fma(B(lx), d, (B(hx) * c).astype(f32)), (B(hy) * fma(B(lx), b, (B(hx) * a).astype(f32))).astype(f32)),
        "4 weights sep: a*(hy*hx)+b*(hy*lx)+c*(ly*hx)+d*(ly*lx)": ((((a * B(hy * hx)).astype(f32) + (b * B(hy * lx)).astype(f32)).astype(f32) + (c * B(ly * hx)).astype(f32)).astype(f32) + (d * B(ly * lx)).astype(f32)).astype(f32),
        "4 weights fma chain": fma(d, B((ly * lx).astype(f32)), fma(c, B((ly * hx).astype(f32)), fma(b, B((hy * lx).astype(f32)), (a * B((hy * hx).astype(f32))).astype(f32)))),
        "4 weights fma chain from 0, order a,b,c,d": fma(d, B((ly * lx).astype(f32)), fma(c, B((ly * hx).astype(f32)), fma(b, B((hy * lx).astype(f32)), (a * B((hy * hx).astype(f32))).astype(f32)))),
    }
    for k, v in cands.items():
        print(f"{Hi}x{Wi}->{Ho}x{Wo} [{k}]: {(v.view(np.int32) != ref.view(np.int32)).sum()} of {v.size} differ, max {np.abs(v - ref).max():.1e}")
