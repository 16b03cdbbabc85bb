"""scratch/r5/pk_where.hip on a GPU box: ONE hog launch of 64 single-wave workgroups (so most CUs and SIMDs have none), long enough to cover 12 victim launches; then per
(XCC, SE, CU, SIMD): victim runs, mismatches, and whether a hog wave sat there.
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared scratch/r5/pk_where.hip -o scratch/r5/libpk_where.so"""
import ctypes, os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
Wl = ctypes.CDLL(os.path.join(ROOT, "scratch", "r5", "libpk_where.so"))
Wl.pk_where_victim.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
Wl.pk_where_hog.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
DEV = "cuda"
N = 1 << 16
for n_hog in (64, 256, 1024):
    runs, bad, hog = (torch.zeros(N, device=DEV, dtype=torch.int32) for _ in range(3))
    sink = torch.zeros(4, device=DEV)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        assert Wl.pk_where_hog(n_hog, 12000, hog.data_ptr(), sink.data_ptr(), side.cuda_stream) == 0     # 12000 x 64 MFMAs x 16 cycles ~ 5 ms alone on its SIMD
    for _ in range(12):
        assert Wl.pk_where_victim(73216, 40, runs.data_ptr(), bad.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    def fold(t):          # drop HW_ID's pipe_id (key bits 3:2): it is the compute pipe of the QUEUE, and the two streams use different ones
        t = t.cpu().view(16, 256, 4, 4).sum(2)          # [xcc][se, sh, cu][pipe][simd] -> [xcc][se, sh, cu][simd]
        return t.reshape(-1)
    runs, bad, hog = fold(runs), fold(bad), fold(hog)
    keys = torch.nonzero(runs + hog).flatten().tolist()
    cu_of = lambda k: k >> 2                      # drop simd_id (HW_ID bits 5:4 -> key bits 1:0)
    hog_simd = {k for k in keys if hog[k] > 0}
    hog_cu = {cu_of(k) for k in hog_simd}
    tot = collections.Counter(); badc = collections.Counter()
    for k in keys:
        cls = "same SIMD as a hog wave" if k in hog_simd else ("same CU, other SIMD" if cu_of(k) in hog_cu else "CU without a hog wave")
        tot[cls] += int(runs[k]); badc[cls] += int(bad[k])
    print(f"hog waves {n_hog}: on {len(hog_simd)} SIMDs of {len(hog_cu)} CUs; victim (SIMD slots seen: {sum(1 for k in keys if runs[k] > 0)})")
    for cls in ("same SIMD as a hog wave", "same CU, other SIMD", "CU without a hog wave"):
        print(f"    {cls:26s}: {tot[cls]:9d} victim waves, {badc[cls]:9d} mismatching iterations")
