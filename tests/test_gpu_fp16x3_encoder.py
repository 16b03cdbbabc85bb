"""fp32-GRADE conv0 on the fp16 matrix cores (csrc/conv_f16x3.hip; what a no-grad scene encode runs by default, `encoder.encoder_precision("fp32")` selects
the fp32-MFMA kernel instead): the plane sweep's
two-piece fp16 store, conv0 = x0*w0 + x0*w1 + x1*w0 against float64 convolutions of the UN-rounded fp32 operands (the claim is fp32 grade, not
"equal to an emulation of its own rounding"), and the scene encode end to end against the fp32 path and the CPU oracle
(reference: models.py:756 conv0, :839-893 plane sweep)."""
import pytest
import torch

from tests.util import load_weights, record_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("V,H,W,D,pad", [(3, 16, 24, 8, 4), (5, 12, 20, 16, 2), (3, 128, 160, 128, 24)])
def test_planesweep_two_piece_store_reproduces_the_fp32_volume(V, H, W, D, pad):
    """mvsnerf_planesweep_costvar_f16x2_fwd = the fp32 sweep, every value x stored as hi = fp16(x/16), lo = fp16(x/16 - hi) (round to nearest) in
    channel blocks of sixteen, hi plane then lo plane: bit-equal to that definition, and 16 (hi + lo) gives x back to 2^-22 |x| + 2^-21."""
    from mvsnerf_amd import encoder as E
    from tests.test_gpu_bf16_encoder import _sweep_inputs
    imgs, feats, proj, dv = _sweep_inputs(V, H, W, D, pad, seed=V * 10 + D)
    with torch.no_grad():
        cost32, masks32, _ = E._plane_sweep(imgs, feats, proj, dv, pad, True)
        costh, masksh, _ = E._plane_sweep(imgs, feats, proj, dv, pad, True, blocked="fp16x2")
    assert torch.equal(masks32, masksh)
    n_ch = 3 * V + 32
    nb = (n_ch + 15) // 16
    Dp, Hp, Wp = costh.dims
    assert costh.buf.dtype == torch.float16 and tuple(costh.buf.shape) == (2, nb, Dp * Hp * Wp, 16)
    ref = torch.zeros((nb * 16, Dp * Hp * Wp), device=DEV)
    ref[:n_ch] = cost32[0].reshape(n_ch, -1)
    ref = (ref * 0.0625).reshape(nb, 16, -1).permute(0, 2, 1)
    hi = ref.to(torch.float16)
    lo = (ref - hi.float()).to(torch.float16)
    assert torch.equal(costh.buf[0], hi) and torch.equal(costh.buf[1], lo)
    back = (costh.buf[0].double() + costh.buf[1].double()) * 16
    err = (back - ref.double() * 16).abs()
    assert bool((err <= 2.0 ** -22 * (ref.double() * 16).abs() + 2.0 ** -21).all())


@pytest.mark.parametrize("cin,dims", [(41, (8, 16, 32)), (47, (5, 9, 17)), (32, (4, 8, 16)), (35, (3, 7, 33)), (41, (128, 176, 208))])
def test_conv0_f16x3_forward_vs_float64(cin, dims):
    """conv0 on three v_mfma_f32_16x16x32_f16 per product against the float64 convolution of the fp32 operands themselves, next to the default
    fp32-MFMA kernel on the same operands: fp32 grade = no further from float64 than a few times the fp32 kernel.  The last case is config 2's
    volume (timed)."""
    import torch.nn.functional as F
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    L = _lib.lib()
    D, H, W = dims
    g = torch.Generator(DEV).manual_seed(cin + D)
    nb = (cin + 15) // 16
    x = torch.randn((nb * 16, D, H, W), device=DEV, generator=g) * 20            # cost-volume-like magnitudes (variance channels reach hundreds)
    x[:9] = torch.rand((9, D, H, W), device=DEV, generator=g)                    # thumbnails in [0, 1]
    x[cin:] = 0
    w = torch.randn((8, cin, 3, 3, 3), device=DEV, generator=g) * 0.1
    xs = (x * 0.0625).reshape(nb, 16, -1).permute(0, 2, 1).contiguous()
    hi = xs.to(torch.float16)
    x16 = torch.stack((hi, (xs - hi.float()).to(torch.float16))).contiguous()
    pk = torch.empty(L.mvsnerf_conv0_f16x3_packed_elems(cin), device=DEV, dtype=torch.float16)
    assert L.mvsnerf_conv0_f16x3_pack(w.data_ptr(), cin, pk.data_ptr(), stream_ptr()) == 0
    out = torch.full((D, H, W, 8), float("nan"), device=DEV)
    ntile = L.mvsnerf_conv0_bf16_tiles(D, H, W)
    part = torch.empty(ntile * 16, device=DEV)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.mvsnerf_conv0_f16x3_fwd(x16.data_ptr(), cin, D, H, W, pk.data_ptr(), out.data_ptr(), part.data_ptr(), stream_ptr()) == 0
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    big = D * H * W > 1_000_000
    if big:      # float64 on a depth slab only (the whole volume would take minutes on the GPU's fp64 path)
        sl = slice(40, 52)
        ref = F.conv3d(x[None, :cin, sl.start - 1:sl.stop + 1].double(), w.double(), padding=(0, 1, 1))[0].permute(1, 2, 3, 0)
        got = out[sl]
    else:
        ref = F.conv3d(x[None, :cin].double(), w.double(), padding=1)[0].permute(1, 2, 3, 0)
        got = out
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max())
    record_err(f"conv0_f16x3_fwd_{cin}_{D}x{H}x{W}", err, scale=scale)
    print(f"[conv0 fp16x3 fwd cin={cin} {D}x{H}x{W}] {ms:.3f} ms; max err vs float64 {err:.2e} (|out| max {scale:.1f})")
    assert torch.isfinite(out).all() and err < 2e-6 * scale
    s = part.view(2, 8, ntile).double().sum(2)          # channel-major InPlaceABN partials: [{sum, sum of squares}][channel][tile]
    assert float((s[0] - out.double().sum((0, 1, 2))).abs().max()) < 1e-6 * float(out.double().abs().sum((0, 1, 2)).max())
    assert float((s[1] - (out.double() ** 2).sum((0, 1, 2))).abs().max()) < 1e-6 * float((out.double() ** 2).sum((0, 1, 2)).max())


def test_encode_fp16x3_vs_fp32_path_and_oracle():
    """MVSNet.forward (3 x 256x320 images, 64 planes, shipped weights) with the fp16 conv0 (the default of a no-grad encode = "fp16x3") against
    the all-fp32 kernels (encoder_precision("fp32")) and against the CPU oracle: the volume moves by no more than the fp32 path's own distance
    from the oracle."""
    from mvsnerf_amd import encoder as E, models
    from mvsnerf_amd.synth import make_rig
    from oracle import mvsnerf_oracle as O
    _, mvs_sd = load_weights()
    H, W, D, pad = 256, 320, 64, 24
    rig = make_rig(H, W, seed=1234)
    imgs, proj, nf = rig["images"][:, :3], rig["proj_mats"][:, :3], rig["near_fars"][0, 0]
    with torch.no_grad():
        ovol = O.mvsnet_forward(imgs, proj, nf, mvs_sd, pad=pad, D=D)[0]
    net = models.MVSNet()
    net.load_state_dict(mvs_sd)
    net = net.to(DEV).train()
    net.D = D
    with torch.no_grad():
        with E.encoder_precision("fp32"):
            v32 = net(imgs.to(DEV), proj.to(DEV), nf.to(DEV), pad=pad)[0].float().cpu().reshape(ovol.shape)
        with E.encoder_precision("fp16x3"):
            v16 = net(imgs.to(DEV), proj.to(DEV), nf.to(DEV), pad=pad)[0].float().cpu().reshape(ovol.shape)
        assert E.ENCODER_PRECISION == "auto"
        vdef = net(imgs.to(DEV), proj.to(DEV), nf.to(DEV), pad=pad)[0].float().cpu().reshape(ovol.shape)
    assert torch.equal(vdef, v16) and not torch.equal(v32, v16)       # the default of a no-grad encode IS the fp16 conv0
    e32, e16, d = float((v32 - ovol).abs().max()), float((v16 - ovol).abs().max()), float((v16 - v32).abs().max())
    record_err("encode_fp16x3:vs_oracle", e16, scale=float(ovol.abs().max()))
    record_err("encode_fp16x3:fp32_path_vs_oracle", e32, scale=float(ovol.abs().max()))
    print(f"encode 256x320x64: fp32 kernels vs oracle {e32:.3g}, fp16x3 conv0 vs oracle {e16:.3g}, fp16x3 vs fp32 kernels {d:.3g} (|vol| <= {float(ovol.abs().max()):.3g})")
    assert e16 < 2 * e32 + 5e-6 and d < 2 * e32 + 5e-6
    with pytest.raises(ValueError):
        E.encoder_precision("fp8")


# ------------------------------------------------------------------ conv1 / conv2 on the two-piece fp16 LDS-tiled kernel (csrc/conv_f16x3_tiled.hip)
def _conv12_case(cin, stride, dims, seed, lazy=True, scale_x=1.0):
    D, H, W = dims
    g = torch.Generator(DEV).manual_seed(seed)
    x = torch.randn((D, H, W, cin), device=DEV, generator=g) * scale_x
    sc = (torch.rand(cin, device=DEV, generator=g) + 0.5) if lazy else None
    sh = (torch.randn(cin, device=DEV, generator=g) * 0.3) if lazy else None
    w = torch.randn((16, cin, 3, 3, 3), device=DEV, generator=g) * 0.1
    return x, sc, sh, w


@pytest.mark.parametrize("cin,stride,dims", [(16, 1, (8, 12, 40)), (8, 2, (10, 14, 70)), (16, 1, (8, 12, 64)), (8, 2, (10, 14, 128)), (16, 1, (5, 9, 33)),
                                             (16, 1, (64, 88, 104)), (8, 2, (128, 176, 208))])
def test_conv12_f16x3_tiled_vs_float64(cin, stride, dims):
    """conv2 (16 -> 16) / conv1 (8 -> 16, stride 2) of CostRegNet (models.py:743-746) on three v_mfma_f32_16x16x32_f16 per product against the float64
    convolution of the same fp32 operands (pending InPlaceABN applied in float64), with torch's own fp32 convolution as the yardstick for "fp32 grade";
    32-wide and 16-wide tiles (rows of 40 / 35 / 104 / 17 voxels take the narrow ones), ragged edges, the two config-2 shapes (timed), the InPlaceABN partial sums."""
    import torch.nn.functional as F
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    L = _lib.lib()
    D, H, W = dims
    x, sc, sh, w = _conv12_case(cin, stride, dims, cin * 7 + D)
    pk = torch.zeros(L.mvsnerf_conv3d_f16x3_packed_elems(cin), device=DEV, dtype=torch.float16)
    assert L.mvsnerf_conv3d_f16x3_pack(w.data_ptr(), cin, 16, pk.data_ptr(), stream_ptr()) == 0
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.full((Do, Ho, Wo, 16), float("nan"), device=DEV)
    nslots = L.mvsnerf_conv3d_f16x3_slots()
    part = torch.full((2 * 16 * nslots,), float("nan"), device=DEV)
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert L.mvsnerf_conv3d_f16x3_fwd(x.data_ptr(), sc.data_ptr(), sh.data_ptr(), cin, cin, D, H, W, pk.data_ptr(), 16, stride, out.data_ptr(), part.data_ptr(), stream_ptr()) == 0
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    xa = x.double() * sc.double() + sh.double()
    xa = torch.where(xa > 0, xa, 0.01 * xa)
    zo = min(Do, 16 if stride == 1 else 8)                    # float64 on a slab of the first planes (the fp64 convolution is an im2col fallback)
    zs = min(D, zo * stride + 1)
    xin = xa[:zs].permute(3, 0, 1, 2)[None]
    ref = F.conv3d(xin, w.double(), stride=stride, padding=1)[0].permute(1, 2, 3, 0)[:zo]
    y32 = F.conv3d(xin.float(), w, stride=stride, padding=1)[0].permute(1, 2, 3, 0)[:zo]
    top = float(ref.abs().max())
    e16, e32 = float((out[:zo].double() - ref).abs().max()) / top, float((y32.double() - ref).abs().max()) / top
    record_err(f"conv12_f16x3_{cin}s{stride}_{D}x{H}x{W}", e16 * top, scale=top)
    print(f"[conv f16x3 tiled cin={cin} s={stride} {D}x{H}x{W}] {us:.1f} us; err / max|ref|: fp16x3 {e16:.2e}, torch fp32 {e32:.2e}")
    assert bool(torch.isfinite(out).all())
    assert e16 < 2.5e-6 and e16 < 3 * e32 + 5e-7, (e16, e32)          # measured 2.3e-7 .. 5.8e-7 where torch's fp32 convolution is 3.5e-7 .. 9.1e-7
    ps = part.view(2, 16, nslots).double()
    o64 = out.double()
    assert float((ps[0].sum(1) - o64.sum((0, 1, 2))).abs().max()) < 1e-6 * float(o64.abs().sum((0, 1, 2)).max())
    assert float((ps[1].sum(1) - (o64 ** 2).sum((0, 1, 2))).abs().max()) < 1e-6 * float((o64 ** 2).sum((0, 1, 2)).max())


@pytest.mark.parametrize("cin,stride,dims", [(16, 1, (16, 24, 64)), (8, 2, (16, 24, 70))])
def test_conv12_guarded_sequence(cin, stride, dims, monkeypatch):
    """mvsnerf_conv3d_f16x3_guarded_fwd through encoder._conv, as CostRegNet._run issues it for a no-grad encode: in range the result IS the fp16 kernel's
    and no fallback is counted; with activations beyond fp16's range (|x 2^4| > 65504) the result is the layer's fp32 kernel's, bit for bit, the statistics
    are those of that result, the fallback is counted and the guard re-armed."""
    import torch.nn as nn
    from mvsnerf_amd import encoder as E, ops
    monkeypatch.setattr(E, "F16X3_MIN_VOXELS", 0)
    D, H, W = dims
    torch.manual_seed(cin)
    conv = nn.Conv3d(cin, 16, 3, stride=stride, padding=1, bias=False).to(DEV)
    pk = E._PackedConv(conv, False)
    for scale_x, trips in ((1.0, 0), (5000.0, 1)):
        x, sc, sh, _ = _conv12_case(cin, stride, dims, 3, scale_x=scale_x)
        lz = E._Lazy(x, sc, sh, (D, H, W, cin))
        with torch.no_grad():
            ref, ref_part = E._conv(lz, None, (D, H, W, cin), cin, pk.get, cin, 16, stride, packed=pk, want_stats=True)           # the fp32 kernel of the layer
            with E._layers_f16x3("plain"):
                plain, _ = E._conv(lz, None, (D, H, W, cin), cin, pk.get, cin, 16, stride, packed=pk, want_stats=True)
            before = ops.guard_fallbacks()
            with E._layers_f16x3("guarded"):
                got, (part, nblk) = E._conv(lz, None, (D, H, W, cin), cin, pk.get, cin, 16, stride, packed=pk, want_stats=True)
            n_fb = ops.guard_fallbacks() - before
        assert n_fb == trips and int(ops.guard_words()[0].item()) == 0
        if trips:
            assert torch.equal(got, ref) and not bool(torch.isfinite(plain).all())            # the fp32 kernel's bits; the unguarded kernel left NaNs
        else:
            assert torch.equal(got, plain)
            e = float((got - ref).abs().max()) / float(ref.abs().max())
            assert e < 5e-6, e                                                                 # two fp32-grade results of the same layer
        s = part.view(2, 16, nblk).double().sum(2)
        o64 = got.double()
        assert float((s[0] - o64.sum((0, 1, 2))).abs().max()) < 1e-6 * float(o64.abs().sum((0, 1, 2)).max())
        assert float((s[1] - (o64 ** 2).sum((0, 1, 2))).abs().max()) < 1e-6 * float((o64 ** 2).sum((0, 1, 2)).max())
