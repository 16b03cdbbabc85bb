"""Depth-fastest neural volumes (MVSNERF_VOL_HWDC, vol[y][x][d][8]: what the encoder emits since round 4) against the channel-last order
vol[d][y][x][8] (MVSNERF_VOL_DHWC) the reference's NCDHW converts to: the SAME logical tensor in the two memory orders must give the same
BITS from every entry that reads a volume - the stand-alone lookup (index_point_feature, utils.py:357-383), the fused gather
(gen_pts_feats, renderer.py:124-136), rendering() in the fp32 and the guarded default mode, the frame render, the generic-C kernel and the
differentiable ray march (forward values and all gradients) - including samples outside the volume, NaN coordinates and ragged sizes; and
the encoder's transposing epilogue (mvsnerf_abn_apply_add_hwdc) must equal its channel-last twin element for element."""
import pytest
import torch

from tests.util import load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(vol_ncdhw):
    """(DHWC view, HWDC view) of one logical (1,C,D,H,W) tensor, as ops.channels_last_volume returns them."""
    from mvsnerf_amd import ops
    v = vol_ncdhw.to(DEV)
    d = v.contiguous(memory_format=torch.channels_last_3d)                       # memory [d][y][x][c]
    h = v[0].permute(2, 3, 1, 0).contiguous().permute(3, 2, 0, 1).unsqueeze(0)    # memory [y][x][d][c], logical (1,C,D,H,W)
    assert torch.equal(d, h)
    cd, ch = ops.channels_last_volume(d), ops.channels_last_volume(h)
    assert ops.vol_ptr_layout(cd)[1] == ops.VOL_DHWC and ops.vol_ptr_layout(ch)[1] == ops.VOL_HWDC
    assert cd.shape == ch.shape and ch.data_ptr() == h.data_ptr()                 # zero-copy
    return d, h, cd, ch


@pytest.mark.parametrize("C,dims,P", [(8, (16, 24, 32), 4096), (8, (5, 3, 7), 333), (8, (128, 176, 208), 131072), (20, (6, 9, 11), 500), (3, (4, 4, 4), 64)])
def test_lookup_same_bits_in_both_layouts(C, dims, P):
    from mvsnerf_amd import ops, utils as U
    g = torch.Generator().manual_seed(C * 1000 + P)
    vol = torch.randn((1, C, *dims), generator=g)
    d, h, cd, ch = _both(vol)
    ndc = (torch.rand((P, 3), generator=g) * 1.3 - 0.15)                         # 23 % of the coordinates outside [0,1]: zeros padding on every face
    ndc[::97, 1] = float("nan")
    ndc[5::131, 2] = 1e30
    ndc = ndc.to(DEV)
    with torch.no_grad():
        a, b = ops.volume_sample(cd, ndc), ops.volume_sample(ch, ndc)
        assert torch.equal(a.nan_to_num(7.0), b.nan_to_num(7.0))
        if C == 8:                                                               # strided rows (what gen_pts_feats writes) and the reference-named entry
            oa, ob = torch.zeros((P, 20), device=DEV), torch.zeros((P, 20), device=DEV)
            ops.volume_sample(cd, ndc, out=oa, out_stride=20); ops.volume_sample(ch, ndc, out=ob, out_stride=20)
            assert torch.equal(oa.nan_to_num(7.0), ob.nan_to_num(7.0))
            ia = U.index_point_feature(d, ndc.view(-1, 1, 3)); ib = U.index_point_feature(h, ndc.view(-1, 1, 3))
            assert torch.equal(ia.nan_to_num(7.0), ib.nan_to_num(7.0))


def test_ray_aligned_samples_and_rendering_same_bits():
    """Rays as the ray march produces them (consecutive samples step in depth - the case the layout is made for): fused gather,
    rendering() on the fp32 kernels and in the guarded default, render_pixels; one sample, one ray, S > 64."""
    from mvsnerf_amd import ops, renderer as R
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from tests.test_gpu_fp16x3 import _load_net, _qfn
    from tests.test_gpu_raymarch import _args
    from oracle import mvsnerf_oracle as O
    net = _load_net()
    qfn, _ = _qfn()
    rig = make_rig(64, 96, seed=11, rot_deg=2.0)
    pose = pose_ref_of(rig)
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    g = torch.Generator().manual_seed(0)
    d, h, cd, ch = _both(torch.randn((1, 8, 16, 24, 32), generator=g))
    imgs = rig["images_raw"][:, :3].to(DEV)
    for n, s in ((96, 32), (1, 1), (7, 80)):
        pts, dirs, _, ndc, zv, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], n, s, pad=4, t_rand=torch.rand((n, s), generator=g), generator=g)
        b = tuple(t.to(DEV) for t in (pts, ndc, zv, ro, dirs))
        with torch.no_grad():
            fa, da = ops.gather(cd, imgs[0], pose_d["w2cs"][:3].contiguous(), pose_d["intrinsics"][:3].contiguous(), b[0], b[1], b[4])
            fb, db = ops.gather(ch, imgs[0], pose_d["w2cs"][:3].contiguous(), pose_d["intrinsics"][:3].contiguous(), b[0], b[1], b[4])
            assert torch.equal(fa, fb) and torch.equal(da, db)
            for mode in ("fp32", "auto"):
                with ops.mlp_precision(mode):
                    ra = R.rendering(_args(N_samples=s), pose_d, *b, d, imgs, network_fn=net, network_query_fn=qfn)
                    rb = R.rendering(_args(N_samples=s), pose_d, *b, h, imgs, network_fn=net, network_query_fn=qfn)
                for x, y in zip(ra[:5], rb[:5]):
                    assert torch.equal(x, y)
    H, W, S, pad = 64, 96, 24, 4
    pd = pose_d
    common = dict(first_pixel=77, n_pixels=3000, pad=pad, batch_rays=1024, want=("depth", "acc"))
    args = (imgs[0], pd["w2cs"][:3].contiguous(), pd["intrinsics"][:3].contiguous(), net.packed(20), H, W, pd["intrinsics"][-1], pd["c2ws"][-1],
            pd["intrinsics"][-1], pd["w2cs"][0], pd["near_fars"][-1], pd["near_fars"][0], S)
    with torch.no_grad():
        a, b = ops.render_pixels(cd, *args, **common), ops.render_pixels(ch, *args, **common)
    for k in ("rgb", "depth", "acc"):
        assert torch.equal(a[k], b[k]), k


def test_training_ray_march_same_values_and_gradients():
    """The differentiable ray march reads either layout; the volume gradient comes back in the logical shape whatever the forward's memory
    order (it is accumulated channel-last, [d][y][x][c], by float atomics: compared to the atomics' tolerance, forward values exactly)."""
    from mvsnerf_amd import ops, renderer as R
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from tests.test_gpu_fp16x3 import _load_net, _qfn
    from tests.test_gpu_raymarch import _args
    from oracle import mvsnerf_oracle as O
    qfn, _ = _qfn()
    rig = make_rig(64, 96, seed=3, rot_deg=2.0)
    pose = pose_ref_of(rig)
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    g = torch.Generator().manual_seed(1)
    d, h, _, _ = _both(torch.randn((1, 8, 16, 24, 32), generator=g))
    imgs = rig["images_raw"][:, :3].to(DEV)
    pts, dirs, _, ndc, zv, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], 64, 32, pad=4, t_rand=torch.rand((64, 32), generator=g), generator=g)
    b = tuple(t.to(DEV) for t in (pts, ndc, zv, ro, dirs))
    res = []
    for vol in (d, h):
        net = _load_net()
        v = vol.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        assert v.stride() == vol.stride()
        out = R.rendering(_args(N_samples=32), pose_d, *b, v, imgs, network_fn=net, network_query_fn=qfn)
        (out[0].sum() + out[3].sum()).backward()
        res.append((out, v.grad.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]))
    (oa, ga, pa), (ob, gb, pb) = res
    for x, y in zip(oa[:5], ob[:5]):
        assert torch.equal(x, y)
    assert ga.shape == gb.shape == d.shape
    assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max())
    for x, y in zip(pa, pb):
        assert float((x - y).abs().max()) <= 1e-5 * max(float(x.abs().max()), 1e-12)


def test_encoder_emits_depth_fastest_volume_with_the_same_values():
    """MVSNet.forward: the volume is handed out depth-fastest (zero-copy into the ray march) and equals, element for element, what the
    channel-last epilogue writes; the stand-alone CostRegNet module and a gradient-carrying forward do the same."""
    from mvsnerf_amd import encoder as E, models, ops
    from mvsnerf_amd.synth import make_rig
    _, mvs_sd = load_weights()
    Himg, Wimg, D, pad = 128, 160, 40, 8                                          # D = 40: a ragged last depth tile (32 + 8); widths not multiples of 16
    rig = make_rig(Himg, Wimg, seed=1234)
    imgs, proj, nf = rig["images"][:, :3].to(DEV), rig["proj_mats"][:, :3].to(DEV), rig["near_fars"][0, 0].to(DEV)
    net = models.MVSNet()
    net.load_state_dict(mvs_sd)
    net = net.to(DEV).train()
    net.D = D
    assert E.VOLUME_LAYOUT == "hwdc"
    with torch.no_grad(), E.encoder_precision("fp32"):
        vh = net(imgs, proj, nf, pad=pad)[0]
        E.VOLUME_LAYOUT = "dhwc"
        try:
            vd = net(imgs, proj, nf, pad=pad)[0]
        finally:
            E.VOLUME_LAYOUT = "hwdc"
    assert vh.shape == vd.shape == (1, 8, D, Himg // 4 + 2 * pad, Wimg // 4 + 2 * pad)
    assert ops.vol_ptr_layout(ops.channels_last_volume(vh))[1] == ops.VOL_HWDC and ops.vol_ptr_layout(ops.channels_last_volume(vd))[1] == ops.VOL_DHWC
    assert torch.equal(vh, vd)
    vg = net(imgs, proj, nf, pad=pad)[0]                                          # gradients enabled: the training node's forward
    assert vg.requires_grad and ops.vol_ptr_layout(ops.channels_last_volume(vg))[1] == ops.VOL_HWDC
    assert float((vg.detach() - vd).abs().max()) < 1e-4 * float(vd.abs().max())
