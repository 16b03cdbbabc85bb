"""FeatureNet (reference models.py:688-722) on the HIP 2-D convolution kernels: forward against the CPU oracle with the
shipped checkpoint, backward (all 26 parameter tensors) against PyTorch autograd through the oracle."""
import pytest
import torch

from tests.util import load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net():
    from mvsnerf_amd import models
    _, mvs_sd = load_weights()
    net = models.MVSNet()
    net.load_state_dict(mvs_sd)
    return net.feature.to(DEV).train(), mvs_sd


@pytest.mark.parametrize("N,H,W", [(3, 64, 96), (2, 50, 70), (1, 33, 17), (4, 128, 160)])
def test_featurenet_forward_vs_oracle(N, H, W):
    from oracle import mvsnerf_oracle as O
    fn, sd = _net()
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = torch.randn((N, 3, H, W), generator=g)
    ref = O.feature_net(x, sd)
    rm0 = fn.conv0[0].bn.running_mean.clone()
    nbt0 = [int(l.bn.num_batches_tracked) for l in fn._layers()]
    with torch.no_grad():
        out = fn(x.to(DEV))
    assert out.shape == ref.shape
    err = float((out.cpu() - ref).abs().max())
    assert err < 1e-4 * max(1.0, float(ref.abs().max())), err
    assert not torch.equal(rm0, fn.conv0[0].bn.running_mean)            # train mode updates the running statistics
    assert [int(l.bn.num_batches_tracked) for l in fn._layers()] == [n + 1 for n in nbt0]   # ... and counts the batch, once per layer
    # eval mode: running statistics, against torch's own batch_norm in eval mode
    fn.eval()
    with torch.no_grad():
        out_e = fn(x.to(DEV))
        import torch.nn.functional as F
        h = x.to(DEV)
        for lay in fn._layers():
            h = F.conv2d(h, lay.conv.weight, None, stride=lay.stride, padding=lay.k // 2)
            h = lay.bn(h)
        ref_e = F.conv2d(h, fn.toplayer.weight, fn.toplayer.bias)
    assert float((out_e - ref_e).abs().max()) < 1e-4 * max(1.0, float(ref_e.abs().max()))


@pytest.mark.parametrize("N,H,W", [(3, 64, 96), (2, 50, 70)])
def test_featurenet_backward_vs_autograd(N, H, W):
    from oracle import mvsnerf_oracle as O
    fn, sd0 = _net()
    g = torch.Generator().manual_seed(H)
    x = torch.randn((N, 3, H, W), generator=g)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd0.items()}
    ref = O.feature_net(x, sd)
    Rw = torch.randn(ref.shape, generator=g)
    (ref * Rw).sum().backward()
    out = fn(x.to(DEV))
    assert out.requires_grad
    (out * Rw.to(DEV)).sum().backward()
    errs = {}
    for name, p in fn.named_parameters():
        r = sd["feature." + name].grad
        errs[name] = float((p.grad.cpu() - r).abs().max() / (r.abs().max() + 1e-12))
    bad = {k: v for k, v in errs.items() if not v < 2e-3}
    assert not bad, f"FeatureNet gradient mismatches: {bad}\nall: {errs}"


def test_convbnrelu_standalone():
    import torch.nn.functional as F
    from mvsnerf_amd import models
    torch.manual_seed(0)
    lay = models.ConvBnReLU(8, 16, 5, 2, 2).to(DEV).train()
    x = torch.randn((2, 8, 37, 41), device=DEV)
    with torch.no_grad():
        out = lay(x)
        y = F.conv2d(x, lay.conv.weight, None, stride=2, padding=2)
        ref = F.leaky_relu(F.batch_norm(y, None, None, lay.bn.weight.abs() + lay.bn.eps, lay.bn.bias, True, 0.1, lay.bn.eps), 0.01)
    assert float((out - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("A,B,k,stride,hw,ldx", [
    (8, 3, 3, 1, (40, 56), 4),       # conv0.0: image input, 3 of 4 channels real
    (8, 8, 3, 1, (512, 640), 8),     # conv0.1 at the training size
    (16, 8, 5, 2, (128, 160), 8),    # conv1.0 (output grid given)
    (16, 16, 3, 1, (37, 45), 16),    # conv1.1 (odd sizes)
    (32, 16, 5, 2, (128, 160), 16),  # conv2.0 at the training size
    (32, 32, 3, 1, (128, 160), 32),  # conv2.1 at the training size
])
def test_conv2d_wgrad_vs_float64(A, B, k, stride, hw, ldx):
    """FeatureNet's weight gradients on the matrix cores (wgrad_mfma.hip, the images as the z axis) with the lazily applied InPlaceABN
    of the input layer, against the float64 definition."""
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    N, (Ho, Wo) = 3, hw
    Hi, Wi = (Ho, Wo) if stride == 1 else (2 * Ho, 2 * Wo)
    gen = torch.Generator(DEV).manual_seed(A * 100 + B + k)
    r = lambda *s: torch.randn(s, device=DEV, generator=gen)
    G, X = r(N, Ho, Wo, A), r(N, Hi, Wi, ldx)
    if B < ldx:
        X[..., B:] = 0
    sc, sh = (r(B).abs() + 0.5, r(B)) if B > 3 else (None, None)
    L = _lib.lib()
    ws = torch.empty(L.mvsnerf_conv2d_wgrad_workspace_floats(A, B, k), device=DEV)
    gw = torch.full((A, B, k, k), float("nan"), device=DEV)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.mvsnerf_conv2d_wgrad(G.data_ptr(), A, X.data_ptr(), 0 if sc is None else sc.data_ptr(), 0 if sh is None else sh.data_ptr(), B, ldx,
                                    N, Ho, Wo, Hi, Wi, k, stride, gw.data_ptr(), ws.data_ptr(), stream_ptr())
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
    ms = e0.elapsed_time(e1)
    # float64 definition: gw = d/dW sum(conv2d(act(X), W) * G)
    xa = X[..., :B].double()
    if sc is not None:
        xa = torch.nn.functional.leaky_relu(xa * sc.double() + sh.double(), 0.01)
    ref = torch.nn.grad.conv2d_weight(xa.permute(0, 3, 1, 2), (A, B, k, k), G.double().permute(0, 3, 1, 2), stride=stride, padding=k // 2)
    scale = float(ref.abs().max())
    e_mfma = float((gw.double() - ref).abs().max())
    print(f"[conv2d wgrad A={A} B={B} k{k} s{stride} {N}x{Ho}x{Wo}] {ms:.3f} ms, max err vs float64 {e_mfma:.2e}; |gw| max {scale:.1f}")
    assert torch.isfinite(gw).all() and e_mfma < 1e-5 * scale
