"""GPU parity tests of the scene-encode kernels (L1a): plane sweep / cost volume / CostRegNet - via the C ABI.
Golden fixtures come from the REAL reference; larger shapes are checked against the CPU oracle."""
import pytest
import torch

from tests.util import load_case, load_weights, maxabs

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, atol, rtol=1e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    from tests.util import record_err, caller_tag
    record_err(caller_tag(), float(err.max()), float(b.abs().max()), [atol, rtol])
    return bool((err <= atol + rtol * b.abs()).all()), float(err.max())


@pytest.fixture(scope="module")
def mvs():
    from mvsnerf_amd import models
    _, sd = load_weights()
    net = models.MVSNet()
    net.load_state_dict(sd)          # reference checkpoint keys, strict
    return net.to(DEV).train()


@pytest.mark.parametrize("name", ["caseA", "caseB"])
def test_golden_sweep(name, mvs):
    from mvsnerf_amd import utils as U
    c = load_case(name)
    g = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in c.items()}
    pad = c["pad"]
    with torch.no_grad():
        warped, grid = U.homo_warp(g["ref_feats"][:, 1], g["proj_mats"][:, 1], g["depth_values"], pad=pad)
        assert grid.shape == c["ref_grid_v1"].shape
        # homo_warp / the plane sweep follow the CPU reference's fp32 arithmetic operation for operation (planesweep.hip): the reference-generated
        # fixtures are reproduced exactly (measured 0.0; equality up to the sign of zero, hence `==` on values rather than on bits)
        assert bool((grid.cpu() == c["ref_grid_v1"]).all()), f"grid {maxabs(grid.cpu(), c['ref_grid_v1'])}"
        assert bool((warped.cpu() == c["ref_warped_v1"]).all()), f"warped {maxabs(warped.cpu(), c['ref_warped_v1'])}"
        w2, _ = U.homo_warp(g["ref_feats"][:, 1], g["proj_mats"][:, 1], g["depth_values"], src_grid=g["ref_grid_v1"], pad=pad)
        assert bool((w2.cpu() == c["ref_warped_v1"]).all()), f"warped(grid given) {maxabs(w2.cpu(), c['ref_warped_v1'])}"
        cost, masks = mvs.build_volume_costvar_img(g["images"][:, :3], g["ref_feats"], g["proj_mats"][:, :3], g["depth_values"], pad=pad)
        assert cost.shape == c["ref_cost_img"].shape and masks.shape == c["ref_in_masks"].shape
        assert torch.equal(masks.cpu(), c["ref_in_masks"])                       # no in-frustum decision differs
        # channels 0:9 are the 4x-resized images (ATen's resize rounds differently at some sizes: 1e-7 level) and their warps; the 32
        # variance channels - E[x^2]-E[x]^2, ill-conditioned, |x| up to 14 - are reproduced exactly
        assert maxabs(cost.cpu()[:, :9], c["ref_cost_img"][:, :9]) < 1e-6
        assert bool((cost.cpu()[:, 9:] == c["ref_cost_img"][:, 9:]).all()), f"variance channels {maxabs(cost.cpu()[:, 9:], c['ref_cost_img'][:, 9:])}"
        var, cnt = mvs.build_volume_costvar(g["ref_feats"], g["proj_mats"][:, :3], g["depth_values"], pad=pad)
        assert torch.equal(cnt.cpu(), c["ref_cost_cnt"])
        assert bool((var.cpu() == c["ref_cost_var"]).all()), f"variance {maxabs(var.cpu(), c['ref_cost_var'])}"


@pytest.mark.parametrize("name", ["caseA", "caseB"])
def test_golden_costreg(name, mvs):
    c = load_case(name)
    x = c["ref_cost_img"].to(DEV)
    rm_before = mvs.cost_reg_2.conv0.bn.running_mean.clone()
    with torch.no_grad():
        vol = mvs.cost_reg_2(x)                      # reference-layout NCDHW tensor in (boundary transpose inside)
    assert vol.shape == c["ref_vol_small"].shape
    ok, e = close(vol, c["ref_vol_small"], 2e-5, 2e-6)          # measured 7.4e-6 at |vol| <= 5.5
    assert ok, f"CostRegNet {e}"
    assert vol[0].permute(2, 3, 1, 0).is_contiguous()         # depth-fastest channel-last memory [y][x][d][c]: feeds the ray march with no transpose
    assert not torch.equal(rm_before, mvs.cost_reg_2.conv0.bn.running_mean)   # train-mode side effect reproduced


@pytest.mark.parametrize("name", ["caseA", "caseB"])
def test_golden_mvsnet_forward(name, mvs):
    """Full MVSNet.forward (D=128), all three stages on HIP, against the reference's own output (fixture)."""
    c = load_case(name)
    with torch.no_grad():
        vol, feats, dv = mvs(c["images"][:, :3].to(DEV), c["proj_mats"][:, :3].to(DEV), c["near_fars"][0, 0].to(DEV), pad=c["pad"])
    assert maxabs(dv.cpu(), c["ref_dv128"]) < 1e-6
    ok, e = close(feats, c["ref_feats"], 1e-5, 1e-6)            # measured 3.6e-6 at |f| <= 11.8
    assert ok, f"FeatureNet {e}"
    sub = vol[:, :, ::8].cpu()
    err = (sub - c["ref_vol128_sub"]).abs()
    from tests.util import record_err
    record_err(f"test_golden_mvsnet_forward[{name}]:vol128_sub", float(err.max()), float(c["ref_vol128_sub"].abs().max()), 1e-4)
    # FeatureNet's 1e-6-level differences move variance channels by 1e-4 (E[x^2]-E[x]^2 at |x| ~ 10) and CostRegNet's batch statistics with them
    assert float(err.max()) < 1e-4, f"max {float(err.max())}"
    assert abs(float(vol.double().sum()) - c["ref_vol128_sum"]) < 1e-5 * c["ref_vol128_abssum"]


def test_midsize_vs_oracle(mvs):
    """128x160 images -> features 32x40, pad 4, D=32: every HIP stage against the CPU oracle on the SAME inputs."""
    from mvsnerf_amd.synth import make_rig
    from oracle import mvsnerf_oracle as O
    _, sd = load_weights()
    rig = make_rig(128, 160, seed=77, rot_deg=2.0)
    pad, D = 4, 32
    imgs, proj = rig["images"][:, :3], rig["proj_mats"][:, :3]
    feats = O.feature_net(imgs[0], sd)[None]
    dv = O.depth_planes(2.125, 4.525, D)
    cost_ref, masks_ref = O.build_volume_costvar_img(imgs, feats, proj, dv, pad)
    vol_ref = O.cost_reg_net(cost_ref, sd)
    with torch.no_grad():
        cost, masks = mvs.build_volume_costvar_img(imgs.to(DEV), feats.to(DEV), proj.to(DEV), dv.to(DEV), pad=pad)
        assert torch.equal(masks.cpu(), masks_ref)
        assert maxabs(cost.cpu()[:, :9], cost_ref[:, :9]) < 1e-6
        assert bool((cost.cpu()[:, 9:] == cost_ref[:, 9:]).all()), maxabs(cost.cpu()[:, 9:], cost_ref[:, 9:])    # rotated cameras, pad 4: exact as well
        vol = mvs.cost_reg_2(cost_ref.to(DEV))
    ok, e = close(vol, vol_ref, 2e-5, 2e-6)                      # measured 6.4e-6 at |vol| <= 6.1
    assert ok, f"CostRegNet vs oracle {e}"


@pytest.mark.parametrize("V,bscale,with_img,pad,D", [(3, 6.0, True, 3, 21), (5, 1.0, True, 2, 10), (2, 3.0, False, 0, 13), (8, 1.0, True, 1, 6), (10, 1.0, True, 1, 5), (12, 1.0, False, 0, 4)])
def test_planesweep_tap_reuse_vs_oracle(mvs, V, bscale, with_img, pad, D):
    """planesweep_kernel walks a voxel column through 4 depth planes with the source taps in registers and gathers again only when a
    tap address changes.  Both paths against the CPU oracle, bit for bit: bscale = 1 rigs move the taps by a fraction of a pixel per
    plane (mostly reuse), bscale = 3 / 6 by more than a pixel (a gather on nearly every plane); 1, 2, 4, 7, 9 and 11 source views (more than
    eight views: the per-plane T / depth table in LDS is sized from V - ADVICE round 3); depths that
    are not a multiple of 4; widths that are not a multiple of the 16-column wave."""
    from mvsnerf_amd.synth import make_rig
    from oracle import mvsnerf_oracle as O
    H, W = 22, 29
    base = tuple(b * bscale for b in (0.0, 0.25, -0.25, 0.12, -0.12, 0.1, -0.3, 0.3, 0.2, -0.2, 0.15, -0.15, 0.05))
    rig = make_rig(H * 4, W * 4, n_views=V + 1, seed=31, baselines=base[:V + 1], rot_deg=2.0, smooth=True)
    imgs, proj = rig["images"][:, :V], rig["proj_mats"][:, :V]
    feats = torch.randn((1, V, 32, H, W), generator=torch.Generator().manual_seed(V))
    dv = O.depth_planes(2.125, 4.525, D)
    with torch.no_grad():
        if with_img:
            cost_ref, masks_ref = O.build_volume_costvar_img(imgs, feats, proj, dv, pad)
            cost, masks = mvs.build_volume_costvar_img(imgs.to(DEV), feats.to(DEV), proj.to(DEV), dv.to(DEV), pad=pad)
            assert maxabs(cost.cpu()[:, :3 * V], cost_ref[:, :3 * V]) < 1e-6      # thumbnails inherit the resize's 1e-7
        else:
            cost_ref, masks_ref = O.build_volume_costvar(feats, proj, dv, pad)
            cost, masks = mvs.build_volume_costvar(feats.to(DEV), proj.to(DEV), dv.to(DEV), pad=pad)
    assert torch.equal(masks.cpu().reshape(masks_ref.shape), masks_ref)
    assert bool((cost.cpu()[:, -32:] == cost_ref[:, -32:]).all()), maxabs(cost.cpu()[:, -32:], cost_ref[:, -32:])


def test_abn_stats_and_conv_vs_torch_full_size():
    """Config-2 size (128x176x208): the train-mode ABN statistics and conv0 against torch fp32 on the same GPU."""
    import torch.nn.functional as F
    from mvsnerf_amd import encoder as E
    g = torch.Generator().manual_seed(9)
    D, H, W = 128, 176, 208
    x = (torch.randn((D, H, W, 8), generator=g) * 2 + 0.7).to(DEV)
    bn = E.InPlaceABN(8).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(8, generator=g) + 0.5); bn.bias.copy_(torch.randn(8, generator=g))
        scale, shift, mean_k, invstd_k = E._abn_stats(x, D * H * W, bn, update_running=True)
        xf = x.view(-1, 8).double()
        mean, var = xf.mean(0), xf.var(0, unbiased=False)
        sc_ref = (bn.weight.abs().double() + 1e-5) / torch.sqrt(var + 1e-5)
        assert maxabs(scale.cpu(), sc_ref.cpu().float()) < 1e-6
        assert maxabs(shift.cpu(), (bn.bias.double() - mean * sc_ref).cpu().float()) < 1e-5
        assert maxabs(bn.running_mean.cpu(), (0.1 * mean).cpu().float()) < 1e-6
        assert maxabs(mean_k.cpu(), mean.cpu().float()) < 1e-6 and maxabs(invstd_k.cpu(), (1 / torch.sqrt(var + 1e-5)).cpu().float()) < 1e-6
        # conv0-shaped convolution on a smaller slab (torch's MIOpen conv as the fp32 reference)
        d = 16
        xin = torch.randn((1, 41, d, H, W), generator=g).to(DEV)
        conv = E.ConvBnReLU3D(41, 8).to(DEV)
        buf, ld = E._as_channel_last(xin, 44)
        raw = E._conv(buf, None, (d, H, W, ld), ld, conv._packed.get(), conv._packed.cin_pad, conv._packed.cout, 1)
        ref = F.conv3d(xin, conv.conv.weight, None, padding=1)[0].permute(1, 2, 3, 0)
        assert maxabs(raw.cpu(), ref.cpu()) < 2e-4


@pytest.mark.parametrize("layer", ["conv2", "conv7", "conv9", "feat0.0", "feat0.1"])
def test_fwd_stats_partials_equal_the_sums_of_the_output(layer):
    """The convolutions that leave their own InPlaceABN partial sums (conv2: LDS-tiled VALU kernel; conv7 / conv9: transposed matrix-core
    kernel, one slot per M-tile and parity class; FeatureNet conv0.0 / conv0.1: VALU kernel): the partials, summed in float64, are the
    per-channel sum and sum of squares of the raw output the same launch wrote - on shapes whose last tiles are ragged."""
    from mvsnerf_amd import encoder as E
    g = torch.Generator().manual_seed(11)
    if layer == "conv2":
        conv = E.ConvBnReLU3D(16, 16).to(DEV)
        d, h, w = 7, 19, 29
        x = torch.randn((d, h, w, 16), generator=g).to(DEV)
        raw, part = E._conv(x, None, (d, h, w, 16), 16, conv._packed.get, 16, 16, 1, packed=conv._packed, want_stats=True)
    elif layer in ("conv7", "conv9"):
        cin, cout = (64, 32) if layer == "conv7" else (32, 16)
        up = E._UpBlock(cin, cout, E.InPlaceABN).to(DEV)
        d, h, w = 3, 7, 11
        x = torch.randn((d, h, w, cin), generator=g).to(DEV)
        raw, part = E._conv_t(x, None, (d, h, w, cin), up._packed.get, cin, cout, packed=up._packed, want_stats=True)
    else:
        cin = 3 if layer == "feat0.0" else 8
        blk = E.ConvBnReLU(cin, 8, 3, 1, 1).to(DEV)
        n, h, w = 2, 37, 45
        ld = 4 if cin == 3 else 8
        x = torch.zeros((n, h, w, ld)); x[..., :cin] = torch.randn((n, h, w, cin), generator=g)
        x = x.to(DEV)
        pk = blk._packed
        raw, part = E._conv2d(x, (n, h, w, ld), ld, pk.get(), pk.cin_pad, pk.cout, 3, 1, want_stats=True)
    assert part is not None, "the layer did not leave its statistics"
    buf, nblk = part
    C = raw.shape[-1]
    sums = buf.view(2, C, nblk).double().sum(2).cpu()
    r = raw.reshape(-1, C).double().cpu()
    assert float((sums[0] - r.sum(0)).abs().max()) < 1e-3 * max(1.0, float(r.sum(0).abs().max()))
    assert float((sums[1] - (r * r).sum(0)).abs().max()) < 1e-5 * float((r * r).sum(0).max())


def _blocked(x_cl, cp):
    """(D,H,W,C) channel-last -> [cp/4][D*H*W][4] (zero padding channels)."""
    D, H, W, C = x_cl.shape
    xp = torch.zeros((D * H * W, cp), device=x_cl.device)
    xp[:, :C] = x_cl.reshape(-1, C)
    return xp.view(-1, cp // 4, 4).permute(1, 0, 2).contiguous()


@pytest.mark.parametrize("dims,cin", [((6, 20, 37), 41), ((3, 16, 16), 32), ((8, 33, 18), 47)])
def test_conv0_blocked_wgrad_vs_float64(dims, cin):
    """conv0's weight gradient on the matrix cores (blocked cost volume, voxels as the k dimension) against the float64 definition
    gw[co][ci][tap] = sum_o g[o][co] x[o+tap-1][ci] (torch.nn.grad.conv3d_weight on the CPU)."""
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    D, H, W = dims
    gen = torch.Generator().manual_seed(D * 1000 + cin)
    x = torch.randn((D, H, W, cin), generator=gen)
    g = torch.randn((D, H, W, 8), generator=gen)
    cp = (cin + 3) // 4 * 4
    ref = torch.nn.grad.conv3d_weight(x.permute(3, 0, 1, 2)[None].double(), (8, cin, 3, 3, 3), g.permute(3, 0, 1, 2)[None].double(), padding=1)
    L = _lib.lib()
    xb, gd = _blocked(x.to(DEV), cp), g.to(DEV)
    gw = torch.full((8, cin, 3, 3, 3), float("nan"), device=DEV)
    ws = torch.empty(L.mvsnerf_conv3d_wgrad_workspace_floats(8, cin), device=DEV)
    rc = L.mvsnerf_conv3d_c8_blocked_wgrad(xb.data_ptr(), cp, cin, D, H, W, gd.data_ptr(), gw.data_ptr(), ws.data_ptr(), stream_ptr())
    assert rc == 0
    err = float((gw.cpu().double() - ref).abs().max())
    assert err < 2e-6 * (D * H * W) ** 0.5 * 4, f"max err {err:.3e}"          # fp32 sums of D*H*W unit-variance products
    gw2 = torch.empty_like(gw)
    assert L.mvsnerf_conv3d_c8_blocked_wgrad(xb.data_ptr(), cp, cin, D, H, W, gd.data_ptr(), gw2.data_ptr(), ws.data_ptr(), stream_ptr()) == 0
    assert torch.equal(gw, gw2)                                                 # deterministic


def test_conv0_blocked_wgrad_full_size_vs_rows_kernel():
    """Config-2 size: the matrix-core weight gradient against the VALU kernel it replaces (different summation orders)."""
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    D, H, W, cin, cp = 128, 176, 208, 41, 44
    x = torch.randn((D, H, W, cp), device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    x[..., cin:] = 0
    g = torch.randn((D, H, W, 8), device=DEV, generator=torch.Generator(DEV).manual_seed(2)) * 0.1
    L = _lib.lib()
    ws = torch.empty(L.mvsnerf_conv3d_wgrad_workspace_floats(8, cin), device=DEV)
    old, new = torch.empty((8, cin, 3, 3, 3), device=DEV), torch.empty((8, cin, 3, 3, 3), device=DEV)
    xb = _blocked(x[..., :cin], cp)

    def run_old():
        assert L.mvsnerf_conv3d_wgrad(g.data_ptr(), 0, 0, 0, 0, 0, 8, x.data_ptr(), 0, 0, 0, 0, 0, cin, cp, D, H, W, D, H, W, 1,
                                      old.data_ptr(), ws.data_ptr(), stream_ptr()) == 0

    def run_new():
        assert L.mvsnerf_conv3d_c8_blocked_wgrad(xb.data_ptr(), cp, cin, D, H, W, g.data_ptr(), new.data_ptr(), ws.data_ptr(), stream_ptr()) == 0

    ms = {}
    for name, fn in (("rows", run_old), ("mfma", run_new)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms[name] = e0.elapsed_time(e1) / 5
    scale = float(old.abs().max())
    err = float((old - new).abs().max())
    print(f"[conv0 wgrad 128x176x208x{cin}] rows kernel {ms['rows']:.3f} ms, matrix cores {ms['mfma']:.3f} ms; max diff {err:.2e} (|gw| max {scale:.1f})")
    assert err < 1e-4 * scale


@pytest.mark.parametrize("A,B,stride,dims,g2,xact", [
    (16, 8, 2, (6, 10, 12), False, True),       # conv1 (odd tile counts)
    (16, 8, 2, (64, 88, 104), True, False),     # conv11^T at the training size: G = act(c2) + act(u9)
    (16, 16, 1, (64, 88, 104), False, True),    # conv2 at the training size
    (32, 16, 2, (5, 9, 7), True, False),        # conv9^T
    (32, 32, 1, (32, 44, 52), False, True),     # conv4 at the training size
    (64, 32, 2, (4, 6, 5), False, True),        # conv5
    (64, 64, 1, (16, 22, 26), False, True),     # conv6 at the training size
])
def test_conv3d_wgrad_vs_float64(A, B, stride, dims, g2, xact):
    """The matrix-core weight gradient of the 16/32/64-channel layers (wgrad_mfma.hip), with the lazily applied activations and the skip
    sum of the transposed layers, against its float64 definition  gw[a,b,tap] = sum_vox G[vox,a] * X[vox*stride + tap - 1, b]
    (27 float64 matrix products on the GPU)."""
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    import torch.nn.functional as F
    Do, Ho, Wo = dims
    Di, Hi, Wi = (Do, Ho, Wo) if stride == 1 else (2 * Do, 2 * Ho, 2 * Wo)
    gen = torch.Generator(DEV).manual_seed(A * 100 + B + stride)
    r = lambda *s: torch.randn(s, device=DEV, generator=gen)
    G1, X1 = r(Do, Ho, Wo, A), r(Di, Hi, Wi, B)
    G2 = r(Do, Ho, Wo, A) if g2 else None
    gs = [(r(A).abs() + 0.5, r(A)) for _ in range(2)] if g2 else None
    xs = (r(B).abs() + 0.5, r(B)) if xact else None
    L = _lib.lib()
    ws = torch.empty(L.mvsnerf_conv3d_wgrad_workspace_floats(A, B), device=DEV)
    p = lambda t: 0 if t is None else t.data_ptr()
    gw = torch.full((A, B, 3, 3, 3), float("nan"), device=DEV)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.mvsnerf_conv3d_wgrad(p(G1), p(gs[0][0]) if g2 else 0, p(gs[0][1]) if g2 else 0, p(G2), p(gs[1][0]) if g2 else 0, p(gs[1][1]) if g2 else 0, A,
                                    p(X1), p(xs[0]) if xact else 0, p(xs[1]) if xact else 0, 0, 0, 0, B, B, Do, Ho, Wo, Di, Hi, Wi, stride,
                                    gw.data_ptr(), ws.data_ptr(), stream_ptr())
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
    ms = e0.elapsed_time(e1)
    act = lambda t, sc, sh: F.leaky_relu(t.double() * sc.double() + sh.double(), 0.01)
    G = (act(G1, *gs[0]) + act(G2, *gs[1])) if g2 else G1.double()
    X = act(X1, *xs) if xact else X1.double()
    Xp = F.pad(X, (0, 0, 1, 1, 1, 1, 1, 1))                                  # zero padding 1 on z, y, x
    Gm = G.reshape(-1, A)
    ref = torch.empty((A, B, 3, 3, 3), dtype=torch.float64, device=DEV)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                xt = Xp[dz:dz + stride * Do:stride, dy:dy + stride * Ho:stride, dx:dx + stride * Wo:stride].reshape(-1, B)
                ref[:, :, dz, dy, dx] = Gm.t() @ xt
    scale, err = float(ref.abs().max()), float((gw.double() - ref).abs().max())
    gf = Do * Ho * Wo * A * B * 54 / 1e9
    print(f"[conv3d wgrad A={A} B={B} s{stride} {Do}x{Ho}x{Wo}] {ms:.3f} ms ({gf / ms:.1f} TFLOP/s); max err vs float64 {err:.2e} (|gw| max {scale:.1f})")
    assert torch.isfinite(gw).all() and err < 1e-5 * scale


def test_pack_weights_multi_equals_the_single_layout_entries():
    """mvsnerf_pack_weights_multi (all weight layouts of a step in one launch, straight from the layer's weight tensor) must give the
    bytes of the per-layout entries it replaces: conv3d_pack_weights, + _c8 and _mfma re-layouts of that result, conv2d_pack_weights."""
    import ctypes
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    L = _lib.lib()
    g = torch.Generator(DEV).manual_seed(5)
    w3 = torch.randn((16, 41, 3, 3, 3), device=DEV, generator=g)            # Conv3d(41 -> 16): fwd, cin padded to 44
    wt = torch.randn((32, 16, 3, 3, 3), device=DEV, generator=g)            # ConvTranspose3d(32 -> 16)
    w2 = torch.randn((16, 8, 5, 5), device=DEV, generator=g)                # Conv2d(8 -> 16, k5)
    jobs, refs = [], []

    def ref3(w, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip):
        buf = torch.empty(27 * ci_pad * co_pad, device=DEV)
        assert L.mvsnerf_conv3d_pack_weights(w.data_ptr(), ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip, buf.data_ptr(), stream_ptr()) == 0
        return buf

    # kind 0: forward of the Conv3d, data gradient (flip) of it, forward of the transposed layer
    for (w, q) in ((w3, (41, 16, 44, 16, 27, 41 * 27, 0)), (w3, (16, 41, 16, 44, 41 * 27, 27, 1)), (wt, (32, 16, 32, 16, 16 * 27, 27, 0))):
        refs.append(ref3(w, *q)); jobs.append((w, 0, 27, q))
    # kind 1 (c8): 8-output layer from a 44-channel input
    w8 = torch.randn((8, 41, 3, 3, 3), device=DEV, generator=g)
    base = ref3(w8, 41, 8, 44, 8, 27, 41 * 27, 0)
    c8 = torch.empty_like(base)
    assert L.mvsnerf_conv3d_pack_weights_c8(base.data_ptr(), 44, c8.data_ptr(), stream_ptr()) == 0
    refs.append(c8); jobs.append((w8, 1, 27, (41, 8, 44, 8, 27, 41 * 27, 0)))
    # kind 2 (w32) of the transposed layer
    m32 = torch.empty_like(refs[2])
    assert L.mvsnerf_conv3d_pack_weights_mfma(refs[2].data_ptr(), 32, 16, m32.data_ptr(), stream_ptr()) == 0
    refs.append(m32); jobs.append((wt, 2, 27, (32, 16, 32, 16, 16 * 27, 27, 0)))
    # 2-D, k5, data gradient without flip
    r2 = torch.empty(25 * 16 * 8, device=DEV)
    assert L.mvsnerf_conv2d_pack_weights(w2.data_ptr(), 16, 8, 16, 8, 8 * 25, 25, 5, 0, r2.data_ptr(), stream_ptr()) == 0
    refs.append(r2); jobs.append((w2, 0, 25, (16, 8, 16, 8, 8 * 25, 25, 0)))
    n = len(jobs)
    outs = [torch.full_like(r, float("nan")) for r in refs]
    params = []
    for (_, kind, ntaps, q) in jobs:
        params += [kind, ntaps, *q]
    rc = L.mvsnerf_pack_weights_multi(n, (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs]), (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]),
                                      (ctypes.c_int * (9 * n))(*params), stream_ptr())
    assert rc == 0
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert torch.equal(o, r), f"job {i}"


def test_partial_sum_multi_matches_float64_sums():
    import ctypes
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    L = _lib.lib()
    g = torch.Generator(DEV).manual_seed(6)
    shapes = [(1, 7), (31, 300), (32, 1000), (33, 64), (700, 5000), (2048, 513)]       # (partials, outputs): one slice / exactly 32 / several
    parts = [torch.randn(s, device=DEV, generator=g) for s in shapes]
    dsts = [torch.full((s[1],), float("nan"), device=DEV) for s in shapes]
    n = len(shapes)
    scratch = torch.empty(L.mvsnerf_partial_sum_multi_scratch_floats(sum(s[1] for s in shapes)), device=DEV)
    rc = L.mvsnerf_partial_sum_multi(n, (ctypes.c_void_p * n)(*[p.data_ptr() for p in parts]), (ctypes.c_int * n)(*[s[0] for s in shapes]),
                                     (ctypes.c_int64 * n)(*[s[1] for s in shapes]), (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts]),
                                     scratch.data_ptr(), stream_ptr())
    assert rc == 0
    for p, d in zip(parts, dsts):
        ref = p.double().sum(0)
        assert float((d.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    # deterministic
    d2 = [torch.empty_like(d) for d in dsts]
    assert L.mvsnerf_partial_sum_multi(n, (ctypes.c_void_p * n)(*[p.data_ptr() for p in parts]), (ctypes.c_int * n)(*[s[0] for s in shapes]),
                                       (ctypes.c_int64 * n)(*[s[1] for s in shapes]), (ctypes.c_void_p * n)(*[d.data_ptr() for d in d2]),
                                       scratch.data_ptr(), stream_ptr()) == 0
    assert all(torch.equal(a, b) for a, b in zip(dsts, d2))


@pytest.mark.parametrize("D,near,far", [(128, 2.125, 6.0), (16, 0.5, 1000.0), (192, 1.3333334, 4.7), (7, 3.0, 3.0)])
def test_depth_values_are_the_bits_of_the_aten_formula(D, near, far):
    """MVSNet.forward's depth hypotheses (models.py:903-906) come from ONE launch (mvsnerf_depth_values) instead of linspace / rsub / mul / mul / add:
    the same tensor, bit for bit; the inverse-depth form and host-side near_far keep the ATen path."""
    from mvsnerf_amd import encoder
    dev = torch.device("cuda", 0)
    net = encoder.MVSNet().to(dev)
    net.D = D
    imgs = torch.zeros((1, 3, 3, 8, 8), device=dev)
    nf = torch.tensor([near, far], device=dev)
    t = torch.linspace(0.0, 1.0, steps=D, device=dev)
    want = nf[0] * (1.0 - t) + nf[1] * t
    got = net._depth_values(nf, imgs, False)
    assert got.shape == want.shape and torch.equal(got, want)
    assert torch.equal(net._depth_values(nf, imgs, False), want)                       # second call: cached t_vals
    want_inv = 1.0 / (1.0 / nf[0] * (1.0 - t) + 1.0 / nf[1] * t)
    assert torch.equal(net._depth_values(nf, imgs, True), want_inv)
    assert torch.equal(net._depth_values((nf[0], nf[1]), imgs, False), want)            # a pair of 0-dim tensors: the ATen path
