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
