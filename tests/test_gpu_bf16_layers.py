"""conv1 ... conv11 of CostRegNet on the bf16 matrix cores (csrc/conv3d_bf16.hip; the reference's `precision=16 if args.use_amp`,
train_mvs_nerf_pl.py:317-318, applied to models.py:725-769): every layer shape, forward and data gradient, against float64 convolutions of
exactly the kernel's operands - activations and weights rounded to bf16 (round to nearest even), products and sums exact - plus the
InPlaceABN partial sums that leave with the same launch, lazily-activated / two-source inputs, ragged sizes and the k-split form of the
small layers; then the whole plane sweep -> CostRegNet training node under use_amp against the fp32 node."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.util import record_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


# (Cin, Cout, stride, transposed, dims): the nine layers at small sizes (both M-tile forms: > and < 8192 tiles), ragged edges
LAYERS = [
    (8, 16, 2, False, (12, 20, 36)),      # conv1
    (16, 16, 1, False, (9, 13, 21)),      # conv2
    (16, 16, 1, False, (40, 60, 64)),     # conv2, enough voxels for one M-tile per wave
    (16, 32, 2, False, (10, 14, 18)),     # conv3
    (32, 32, 1, False, (6, 10, 14)),      # conv4
    (32, 64, 2, False, (8, 12, 12)),      # conv5
    (64, 64, 1, False, (4, 6, 10)),       # conv6
    (64, 32, 2, True, (4, 5, 7)),         # conv7
    (32, 16, 2, True, (6, 7, 9)),         # conv9
    (16, 8, 2, True, (8, 12, 20)),        # conv11
    (16, 8, 2, True, (24, 40, 48)),       # conv11, many tiles
    # >= 256 K output voxels: the LDS-tiled kernel (conv_bf16_tiled_kernel) - forward of conv1 / conv2, data gradient of conv2 / conv11; ragged tiles
    (8, 16, 2, False, (96, 130, 172)),    # conv1 (tiled forward)
    (16, 16, 1, False, (50, 70, 78)),     # conv2 (tiled forward and data gradient)
    (16, 8, 2, True, (48, 65, 86)),       # conv11 (tiled data gradient)
]


def _layer(cin, cout, stride, transposed, seed):
    torch.manual_seed(seed)
    conv = (nn.ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False) if transposed
            else nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False))
    return conv.to(DEV)


@pytest.mark.parametrize("cin,cout,stride,transposed,dims", LAYERS)
def test_bf16_layer_forward_and_dgrad_vs_float64(cin, cout, stride, transposed, dims):
    from mvsnerf_amd import encoder as E
    D, H, W = dims
    conv = _layer(cin, cout, stride, transposed, cin * 100 + cout)
    pk = E._PackedConv(conv, transposed)
    g = torch.Generator(DEV).manual_seed(D * 31 + W)
    x = torch.randn((D, H, W, cin), device=DEV, generator=g)
    w64 = _bf(conv.weight.detach()).double()
    xr = _bf(x).double().permute(3, 0, 1, 2)[None]
    with torch.no_grad(), E._layer_precision(True):
        assert pk.get_bf16("fwd") is not None and pk.get_bf16("dgrad") is not None
        if transposed:
            out, partials = E._conv_t(x, None, (D, H, W, cin), pk.get, cin, cout, packed=pk, want_stats=True)
            ref = F.conv_transpose3d(xr, w64, stride=2, padding=1, output_padding=1)[0].permute(1, 2, 3, 0)
        else:
            out, partials = E._conv(x, None, (D, H, W, cin), cin, pk.get, cin, cout, stride, packed=pk, want_stats=True)
            ref = F.conv3d(xr, w64, stride=stride, padding=1)[0].permute(1, 2, 3, 0)
        assert out.shape == ref.shape
        scale, err = float(ref.abs().max()), float((out.double() - ref).abs().max())
        record_err(f"bf16_layer_fwd:{cin}->{cout}s{stride}{'T' if transposed else ''}:{dims}", err, scale=scale)
        assert torch.isfinite(out).all() and err < 3e-6 * scale, (err, scale)          # fp32 accumulation of exact products
        # InPlaceABN partial sums of the same launch
        part, nblk = partials
        s = part.view(2, cout, nblk).double().sum(2)
        o64 = out.double()
        assert float((s[0] - o64.sum((0, 1, 2))).abs().max()) < 1e-6 * float(o64.abs().sum((0, 1, 2)).max())
        assert float((s[1] - (o64 ** 2).sum((0, 1, 2))).abs().max()) < 1e-6 * float((o64 ** 2).sum((0, 1, 2)).max())
        # data gradient: the same kernels with re-packed weights (stride-2 conv <-> transposed conv)
        Do, Ho, Wo = out.shape[:3]
        go = torch.randn((Do, Ho, Wo, cout), device=DEV, generator=g)
        gr = _bf(go).double().permute(3, 0, 1, 2)[None]
        if transposed:
            gx = E._conv(go, None, (Do, Ho, Wo, cout), cout, lambda: pk.get("dgrad"), cout, cin, 2, packed=pk, mode="dgrad")
            refg = F.conv3d(gr, w64, stride=2, padding=1)[0].permute(1, 2, 3, 0)
        elif stride == 1:
            gx = E._conv(go, None, (Do, Ho, Wo, cout), cout, lambda: pk.get("dgrad"), cout, cin, 1, packed=pk, mode="dgrad")
            refg = F.conv_transpose3d(gr, w64, stride=1, padding=1)[0].permute(1, 2, 3, 0)
        else:
            gx = E._conv_t(go, None, (Do, Ho, Wo, cout), lambda: pk.get("dgrad"), cout, cin, packed=pk, mode="dgrad")
            refg = F.conv_transpose3d(gr, w64, stride=2, padding=1, output_padding=1)[0].permute(1, 2, 3, 0)[:D, :H, :W]
            gx = gx[:D, :H, :W]          # an odd input size has one row less than 2 x the output (the layer is only used on even sizes)
        scale, err = float(refg.abs().max()), float((gx.double() - refg).abs().max())
        record_err(f"bf16_layer_dgrad:{cin}->{cout}s{stride}{'T' if transposed else ''}:{dims}", err, scale=scale)
        assert tuple(gx.shape[:3]) == tuple(refg.shape[:3]) and err < 3e-6 * scale, (err, scale)
    # and the fp32 kernels are what runs outside the context
    with torch.no_grad():
        o32 = (E._conv_t(x, None, (D, H, W, cin), pk.get, cin, cout, packed=pk) if transposed
               else E._conv(x, None, (D, H, W, cin), cin, pk.get, cin, cout, stride, packed=pk))
    assert not torch.equal(o32, out)


def test_bf16_layer_with_lazy_and_two_source_input():
    """The U-Net hands a layer its input as raw conv output + pending InPlaceABN (scale, shift) and, for the up-sampling layers, as the SUM
    of two such tensors (skip connection): the activation is applied in fp32 on load, the sum is rounded to bf16 once."""
    from mvsnerf_amd import encoder as E
    D, H, W, cin, cout = 8, 12, 20, 16, 8
    conv = _layer(cin, cout, 2, True, 5)
    pk = E._PackedConv(conv, True)
    g = torch.Generator(DEV).manual_seed(3)
    x1, x2 = torch.randn((D, H, W, cin), device=DEV, generator=g), torch.randn((D, H, W, cin), device=DEV, generator=g)
    sc1, sh1 = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3
    sc2, sh2 = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3
    l1, l2 = E._Lazy(x1, sc1, sh1, (D, H, W, cin)), E._Lazy(x2, sc2, sh2, (D, H, W, cin))
    act = lambda x, sc, sh: F.leaky_relu(torch.addcmul(sh, x, sc), 0.01)
    with torch.no_grad(), E._layer_precision(True):
        out = E._conv_t(l1, l2, (D, H, W, cin), pk.get, cin, cout, packed=pk)
        xin = act(x1, sc1, sh1) + act(x2, sc2, sh2)
        ref = F.conv_transpose3d(_bf(xin).double().permute(3, 0, 1, 2)[None], _bf(conv.weight.detach()).double(), stride=2, padding=1,
                                 output_padding=1)[0].permute(1, 2, 3, 0)
        conv2 = _layer(cin, 16, 1, False, 6)
        pk2 = E._PackedConv(conv2, False)
        out2 = E._conv(l1, None, (D, H, W, cin), cin, pk2.get, cin, 16, 1, packed=pk2)
        ref2 = F.conv3d(_bf(act(x1, sc1, sh1)).double().permute(3, 0, 1, 2)[None], _bf(conv2.weight.detach()).double(), padding=1)[0].permute(1, 2, 3, 0)
    # the kernel forms x * scale + shift with one fma, torch's addcmul may round twice: a value on a bf16 rounding boundary can land on the
    # other side (one operand off by 2^-8 relative) - a handful of outputs move by ~1e-3 of the maximum, everything else agrees to 3e-6
    for o, r, tag in ((out, ref, "convT_two_lazy"), (out2, ref2, "conv_lazy")):
        d = (o.double() - r).abs()
        scale = float(r.abs().max())
        record_err(f"bf16_layer_lazy:{tag}", float(d.max()), scale=scale)
        assert float(d.max()) < 4e-3 * scale and float((d > 1e-5 * scale).double().mean()) < 0.02


def test_bf16_tiled_layers_with_lazy_input_match_the_direct_load_kernel():
    """conv_bf16_tiled_kernel (the layers with >= 256 K output voxels) applies the pending InPlaceABN once per staged voxel; the direct-load
    kernel applies it per tap.  Same operands, same fp32 accumulation order per k-step: the two must agree to fp32 summation noise.  The direct-load
    result comes from a second source that is all zeros with identity activation (two-source inputs never take the tiled kernel)."""
    from mvsnerf_amd import encoder as E
    g = torch.Generator(DEV).manual_seed(11)
    for cin, stride, dims in ((16, 1, (50, 70, 78)), (8, 2, (96, 130, 172))):
        D, H, W = dims
        conv = _layer(cin, 16, stride, False, 40 + cin)
        pk = E._PackedConv(conv, False)
        x = torch.randn((D, H, W, cin), device=DEV, generator=g)
        sc, sh = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3
        lz = E._Lazy(x, sc, sh, (D, H, W, cin))
        zero = torch.zeros_like(x)
        with torch.no_grad(), E._layer_precision(True):
            tiled, (part, nblk) = E._conv(lz, None, (D, H, W, cin), cin, pk.get, cin, 16, stride, packed=pk, want_stats=True)
            direct, (part_d, nblk_d) = E._conv(lz, zero, (D, H, W, cin), cin, pk.get, cin, 16, stride, packed=pk, want_stats=True)
        scale = float(direct.abs().max())
        err = float((tiled - direct).abs().max())
        record_err(f"bf16_tiled_vs_direct:{cin}->16s{stride}", err, scale=scale)
        assert err < 2e-6 * scale, (err, scale)
        assert nblk == nblk_d
        s_t, s_d = part.view(2, 16, nblk).double().sum(2), part_d.view(2, 16, nblk).double().sum(2)
        assert float((s_t - s_d).abs().max()) < 1e-6 * float(s_d.abs().max())
    # the tiled transposed layer (conv11) with its two lazily-activated sources (the skip sum), against float64 on the rounded sum
    D, H, W, cin, cout = 48, 65, 86, 16, 8
    conv = _layer(cin, cout, 2, True, 77)
    pk = E._PackedConv(conv, True)
    x1, x2 = torch.randn((D, H, W, cin), device=DEV, generator=g), torch.randn((D, H, W, cin), device=DEV, generator=g)
    sc1, sh1 = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3
    sc2, sh2 = torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g) * 0.3
    act = lambda x, sc, sh: F.leaky_relu(torch.addcmul(sh, x, sc), 0.01)
    with torch.no_grad(), E._layer_precision(True):
        out, (part, nblk) = E._conv_t(E._Lazy(x1, sc1, sh1, (D, H, W, cin)), E._Lazy(x2, sc2, sh2, (D, H, W, cin)), (D, H, W, cin), pk.get, cin, cout,
                                      packed=pk, want_stats=True)
        ref = F.conv_transpose3d(_bf(act(x1, sc1, sh1) + act(x2, sc2, sh2)).double().permute(3, 0, 1, 2)[None], _bf(conv.weight.detach()).double(),
                                 stride=2, padding=1, output_padding=1)[0].permute(1, 2, 3, 0)
    d = (out.double() - ref).abs()
    scale = float(ref.abs().max())
    record_err("bf16_tiled_convT_two_lazy", float(d.max()), scale=scale)
    assert float(d.max()) < 4e-3 * scale and float((d > 1e-5 * scale).double().mean()) < 0.02     # one fma vs two roundings: see the test above
    s = part.view(2, cout, nblk).double().sum(2)
    assert float((s[0] - out.double().sum((0, 1, 2))).abs().max()) < 1e-6 * float(out.double().abs().sum((0, 1, 2)).max())


def test_use_amp_training_node_runs_every_layer_in_bf16_and_stays_close_to_fp32():
    """The plane sweep -> CostRegNet autograd node with encoder_precision('bf16'): conv0 AND conv1 ... conv11, forward and data gradients, on
    bf16 operands (weight gradients of conv1 ... conv11 stay on the fp32 matrix-core kernels); against the fp32 node on the same inputs."""
    from mvsnerf_amd import encoder as E, models
    from tests.test_gpu_bf16_encoder import _sweep_inputs
    from tests.util import load_weights
    _, mvs_sd = load_weights()
    V, H, W, D, pad = 3, 32, 40, 32, 4
    imgs, feats, proj, dv = _sweep_inputs(V, H, W, D, pad, seed=9)
    res = {}
    for prec in ("fp32", "bf16"):
        net = models.MVSNet()
        net.load_state_dict(mvs_sd)
        net = net.to(DEV).train()
        f = feats.clone().requires_grad_(True)
        with E.encoder_precision(prec):
            vol = E._SweepRegFunction.apply(f, imgs, proj, dv, pad, net.cost_reg_2, *E._costreg_params(net.cost_reg_2))
        # the backward runs OUTSIDE the precision context, as in training_step: it must still take the forward's arithmetic
        gen = torch.Generator(DEV).manual_seed(1)
        if prec == "bf16":
            pk = net.cost_reg_2.conv2._packed
            assert "fwd_bf16" in pk.cache and "dgrad_bf16" not in pk.cache
        (vol * torch.randn(vol.shape, device=DEV, generator=gen)).sum().backward()
        if prec == "bf16":
            for lay in net.cost_reg_2._layers()[1:]:
                assert "fwd_bf16" in lay._packed.cache and "dgrad_bf16" in lay._packed.cache, "a layer fell back to fp32"
        else:
            assert all("fwd_bf16" not in lay._packed.cache for lay in net.cost_reg_2._layers())
        res[prec] = (vol.detach(), f.grad.detach(), {n: p.grad.detach() for n, p in net.cost_reg_2.named_parameters()})
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    e_vol, e_gf = rel(res["bf16"][0], res["fp32"][0]), rel(res["bf16"][1], res["fp32"][1])
    e_gw = {n: rel(res["bf16"][2][n], res["fp32"][2][n]) for n in res["fp32"][2]}
    worst = max(e_gw, key=e_gw.get)
    record_err("bf16_node:volume", e_vol); record_err("bf16_node:d_feats", e_gf); record_err("bf16_node:worst_weight_grad", e_gw[worst])
    print(f"bf16 training node (all layers) vs fp32: volume {e_vol:.2e}, d feats {e_gf:.2e}, worst weight gradient {e_gw[worst]:.2e} ({worst})")
    assert 0 < e_vol < 3e-2 and e_gf < 0.3 and e_gw[worst] < 0.6


# FeatureNet's 2-D layers (models.py:688-722) on the same kernels: (Cin, Cout, k, stride, N, H, W)
LAYERS_2D = [(3, 8, 3, 1, 2, 20, 36), (8, 8, 3, 1, 3, 17, 23), (8, 16, 5, 2, 3, 24, 40), (16, 16, 3, 1, 2, 13, 21), (16, 32, 5, 2, 2, 22, 30),
             (32, 32, 3, 1, 2, 9, 14), (16, 16, 3, 1, 3, 128, 160),
             # >= 128 K pixels: the LDS-tiled kernel (conv_bf16_tiled_kernel<.., KZ = 1>) forward and data gradient, ragged tiles
             (8, 8, 3, 1, 2, 301, 450), (16, 16, 3, 1, 3, 261, 340), (16, 16, 3, 1, 3, 256, 320)]


@pytest.mark.parametrize("cin,cout,k,stride,N,H,W", LAYERS_2D)
def test_bf16_featurenet_layer_forward_and_dgrad_vs_float64(cin, cout, k, stride, N, H, W):
    from mvsnerf_amd import encoder as E
    torch.manual_seed(cin * 10 + k)
    conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False).to(DEV)
    pk = E._PackedConv2d(conv)
    g = torch.Generator(DEV).manual_seed(H * 7 + W)
    x = torch.zeros((N, H, W, pk.cin_pad), device=DEV)
    x[..., :cin] = torch.randn((N, H, W, cin), device=DEV, generator=g)
    w64 = _bf(conv.weight.detach()).double()
    with torch.no_grad(), E._layer_precision(True):
        out, partials = E._conv2d(x, (N, H, W, pk.cin_pad), pk.cin_pad, pk.get, pk.cin_pad, cout, k, stride, want_stats=True, packed=pk)
        assert "fwd_bf16" in pk.cache
        ref = F.conv2d(_bf(x[..., :cin]).double().permute(0, 3, 1, 2), w64, stride=stride, padding=k // 2).permute(0, 2, 3, 1)
        assert out.shape == ref.shape
        scale, err = float(ref.abs().max()), float((out.double() - ref).abs().max())
        record_err(f"bf16_layer2d_fwd:{cin}->{cout}k{k}s{stride}", err, scale=scale)
        assert err < 3e-6 * scale, (err, scale)
        part, nblk = partials
        s = part.view(2, cout, nblk).double().sum(2)
        o64 = out.double()
        assert float((s[0] - o64.sum((0, 1, 2))).abs().max()) < 1e-6 * float(o64.abs().sum((0, 1, 2)).max())
        assert float((s[1] - (o64 ** 2).sum((0, 1, 2))).abs().max()) < 1e-6 * float((o64 ** 2).sum((0, 1, 2)).max())
        if stride == 1 and cin >= 8:
            go = torch.randn(tuple(out.shape), device=DEV, generator=g)
            gx = E._conv2d(go, tuple(out.shape), cout, lambda: pk.get("dgrad"), cout, cin, k, 1, packed=pk, mode="dgrad")
            assert "dgrad_bf16" in pk.cache
            refg = F.conv_transpose2d(_bf(go).double().permute(0, 3, 1, 2), w64, stride=1, padding=k // 2).permute(0, 2, 3, 1)
            scale, err = float(refg.abs().max()), float((gx.double() - refg).abs().max())
            record_err(f"bf16_layer2d_dgrad:{cin}->{cout}k{k}", err, scale=scale)
            assert err < 3e-6 * scale, (err, scale)


def test_bf16_toplayer_with_bias_and_whole_featurenet_vs_fp32():
    """The biased 1x1 toplayer on the bf16 kernel, then FeatureNet end to end under encoder_precision('bf16') against the fp32 kernels:
    forward within bf16 operand rounding, every layer (and every stride-1 data gradient) taken by the bf16 kernels, gradients close."""
    from mvsnerf_amd import encoder as E, models
    from tests.util import load_weights
    torch.manual_seed(1)
    top = nn.Conv2d(32, 32, 1).to(DEV)
    pk = E._PackedConv2d(top)
    g = torch.Generator(DEV).manual_seed(2)
    x = torch.randn((2, 11, 19, 32), device=DEV, generator=g)
    with torch.no_grad(), E._layer_precision(True):
        out = E._conv2d(x, (2, 11, 19, 32), 32, pk.get, 32, 32, 1, 1, bias=top.bias.detach(), packed=pk)
    ref = F.conv2d(_bf(x).double().permute(0, 3, 1, 2), _bf(top.weight.detach()).double(), top.bias.detach().double()).permute(0, 2, 3, 1)
    assert float((out.double() - ref).abs().max()) < 3e-6 * float(ref.abs().max())
    _, mvs_sd = load_weights()
    imgs = torch.rand((3, 3, 128, 160), generator=torch.Generator().manual_seed(4)).to(DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        net = models.MVSNet()
        net.load_state_dict(mvs_sd)
        fn = net.feature.to(DEV).train()
        with E.encoder_precision(prec):
            f = fn(imgs)
        gen = torch.Generator(DEV).manual_seed(5)
        (f * torch.randn(f.shape, device=DEV, generator=gen)).sum().backward()          # outside the context, as in training_step
        if prec == "bf16":
            for lay in fn._layers():
                assert "fwd_bf16" in lay._packed.cache, "a FeatureNet layer fell back to fp32"
            assert "fwd_bf16" in fn._top_packed.cache and "dgrad_bf16" in fn._top_packed.cache
            assert "dgrad_bf16" in fn.conv2[2]._packed.cache and "dgrad_bf16" in fn.conv0[1]._packed.cache
        res[prec] = (f.detach(), {n: p.grad.detach() for n, p in fn.named_parameters()})
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    e_f = rel(res["bf16"][0], res["fp32"][0])
    e_g = {n: rel(res["bf16"][1][n], res["fp32"][1][n]) for n in res["fp32"][1]}
    worst = max(e_g, key=e_g.get)
    record_err("bf16_featurenet:features", e_f); record_err("bf16_featurenet:worst_grad", e_g[worst])
    print(f"bf16 FeatureNet vs fp32: features {e_f:.2e}, worst gradient {e_g[worst]:.2e} ({worst})")
    # (the worst tensor is a bias gradient: a sum of random-sign terms that nearly cancels - measured 0.32 relative on conv0.0.bn.bias)
    assert 0 < e_f < 3e-2 and e_g[worst] < 0.8


def _wgrad_ref(G, X, kz, k, stride):
    """float64 definition on bf16-rounded operands: gw[a][b][tap] = sum_o G[o][a] X[o * s - pad + tap][b]; G (Do,Ho,Wo,A), X (Di,Hi,Wi,B)."""
    G, X = _bf(G).double(), _bf(X).double()
    Do, Ho, Wo, A = G.shape
    B = X.shape[3]
    pz, p, sz = kz // 2, k // 2, (1 if kz == 1 else stride)
    Xp = F.pad(X, (0, 0, p, p + stride, p, p + stride, pz, pz + sz))
    ref = torch.empty((A, B, kz, k, k), dtype=torch.float64, device=G.device)
    g2 = G.reshape(-1, A)
    for dz in range(kz):
        for dy in range(k):
            for dx in range(k):
                xs = Xp[dz:dz + sz * Do:sz, dy:dy + stride * Ho:stride, dx:dx + stride * Wo:stride][:Do, :Ho, :Wo]
                ref[:, :, dz, dy, dx] = g2.t() @ xs.reshape(-1, B)
    return ref


# (A = G channels, B = X channels, kz, k, stride, output dims): the nine 3-D layers' shapes and FeatureNet's
WGRADS = [(16, 8, 3, 3, 2, (6, 10, 40)), (16, 16, 3, 3, 1, (9, 13, 37)), (32, 16, 3, 3, 2, (5, 7, 33)), (32, 32, 3, 3, 1, (4, 6, 32)),
          (64, 32, 3, 3, 2, (4, 6, 13)), (64, 64, 3, 3, 1, (4, 5, 9)), (16, 8, 3, 3, 2, (12, 20, 36)),
          (8, 8, 1, 3, 1, (3, 17, 45)), (16, 8, 1, 5, 2, (3, 12, 20)), (16, 16, 1, 3, 1, (2, 9, 64)), (32, 16, 1, 5, 2, (3, 11, 15)), (32, 32, 1, 1, 1, (3, 10, 33))]


@pytest.mark.parametrize("A,B,kz,k,stride,odims", WGRADS)
def test_bf16_wgrad_vs_float64(A, B, kz, k, stride, odims):
    """csrc/wgrad_bf16.hip against the float64 definition on the bf16-rounded operands; deterministic (run twice); with lazily-activated
    two-source operands in the 3-D case (what the U-Net's backward hands it)."""
    from mvsnerf_amd import encoder as E, _lib
    from mvsnerf_amd.ops import stream_ptr
    L = _lib.lib()
    Do, Ho, Wo = odims
    sz = 1 if kz == 1 else stride
    Di, Hi, Wi = (Do if kz == 1 else (Do * stride if stride == 2 else Do)), Ho * stride if stride == 2 else Ho, Wo * stride if stride == 2 else Wo
    g = torch.Generator(DEV).manual_seed(A * 64 + B + k)
    G = torch.randn((Do, Ho, Wo, A), device=DEV, generator=g) * 0.1
    X = torch.randn((Di, Hi, Wi, B), device=DEV, generator=g)
    n_parts = L.mvsnerf_conv_wgrad_bf16_parts(A, B, Do, Ho, Wo, kz, k, stride)
    assert n_parts > 0
    ws = torch.empty(L.mvsnerf_conv_wgrad_bf16_workspace_floats(A, B, kz, k), device=DEV)
    outs = []
    for rep in range(2):
        gw = torch.full((A, B, kz, k, k), float("nan"), device=DEV)
        assert L.mvsnerf_conv_wgrad_bf16(G.data_ptr(), 0, 0, 0, 0, 0, A, X.data_ptr(), 0, 0, 0, 0, 0, B, B, Do, Ho, Wo, Di, Hi, Wi, kz, k, stride,
                                         gw.data_ptr(), ws.data_ptr(), stream_ptr()) == 0
        torch.cuda.synchronize()
        outs.append(gw)
    assert torch.equal(outs[0], outs[1])
    ref = _wgrad_ref(G, X, kz, k, stride)
    scale, err = float(ref.abs().max()), float((outs[0].double() - ref).abs().max())
    record_err(f"bf16_wgrad:{A}x{B}k{kz}{k}s{stride}:{odims}", err, scale=scale)
    assert torch.isfinite(outs[0]).all() and err < 1e-5 * scale, (err, scale)
    if kz == 3:      # lazy two-source G and X through the Python wrapper (sums = None: the entry reduces its own partials)
        sc, sh = torch.rand(A, device=DEV, generator=g) + 0.5, torch.randn(A, device=DEV, generator=g) * 0.1
        scx, shx = torch.rand(B, device=DEV, generator=g) + 0.5, torch.randn(B, device=DEV, generator=g) * 0.3
        G2, X2 = torch.randn(G.shape, device=DEV, generator=g) * 0.1, torch.randn(X.shape, device=DEV, generator=g)
        act = lambda x, s, h: F.leaky_relu(torch.addcmul(h, x, s), 0.01)
        with E._layer_precision(True):
            gw = E._wgrad(E._Lazy(G, sc, sh, G.shape), E._Lazy(G2, sc, sh, G.shape), A, E._Lazy(X, scx, shx, X.shape), E._Lazy(X2, scx, shx, X.shape), B, B,
                          (Do, Ho, Wo), (Di, Hi, Wi), stride, (A, B, 3, 3, 3))
        ref = _wgrad_ref(act(G, sc, sh) + act(G2, sc, sh), act(X, scx, shx) + act(X2, scx, shx), 3, 3, stride)
        d = (gw.double() - ref).abs()
        assert float(d.max()) < 2e-3 * float(ref.abs().max())          # fma vs addcmul at bf16 rounding boundaries (see the lazy-input test above)
