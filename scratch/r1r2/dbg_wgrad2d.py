import sys, torch
sys.path.insert(0, '.')
from mvsnerf_amd import _lib
from mvsnerf_amd.ops import stream_ptr
L = _lib.lib()
DEV = 'cuda'
A, B, k, stride, ldx, N = 16, 16, 3, 1, 16, 3
for (Ho, Wo) in [(37, 45), (36, 44), (37, 44), (36, 45), (40, 48)]:
    G, X = torch.randn(N, Ho, Wo, A, device=DEV), torch.randn(N, Ho, Wo, ldx, device=DEV)
    ws = torch.full((L.mvsnerf_conv2d_wgrad_workspace_floats(A, B, k),), 0.0, device=DEV)
    for mode in (0, 1):
        L.mvsnerf_tune(b"conv_mfma", mode)
        gw = torch.full((A, B, k, k), float("nan"), device=DEV)
        rc = L.mvsnerf_conv2d_wgrad(G.data_ptr(), A, X.data_ptr(), 0, 0, B, ldx, N, Ho, Wo, Ho, Wo, k, stride, gw.data_ptr(), ws.data_ptr(), stream_ptr())
        torch.cuda.synchronize()
        nan = torch.isnan(gw)
        print(Ho, Wo, "mode", mode, "rc", rc, "nan count", int(nan.sum()), "parts", L.mvsnerf_conv2d_wgrad_parts(A, B, N, Ho, Wo, k, stride),
              "first nan idx", nan.nonzero()[:3].tolist())
L.mvsnerf_tune(b"conv_mfma", 1)
