"""GPU parity of the ray-march BACKWARD kernels (compositing, MLP dgrad/wgrad, trilinear scatter) against PyTorch
autograd run through the CPU oracle on identical inputs (fp32 reference for a floating-point kernel)."""
import pytest
import torch

from tests.util import load_weights

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(n_rays, n_samples, seed, white=False):
    import types
    from mvsnerf_amd import models
    from mvsnerf_amd.synth import make_rig, pose_ref_of
    from oracle import mvsnerf_oracle as O
    rig = make_rig(64, 96, seed=21, rot_deg=2.0, smooth=True)
    pose = pose_ref_of(rig)
    g = torch.Generator().manual_seed(seed)
    vol = torch.randn((1, 8, 16, 24, 32), generator=g)
    pts, dirs, _, ndc, z, ro, _ = O.build_rays(rig["images_raw"], pose, rig["near_fars"], n_rays, n_samples, pad=4,
                                               t_rand=torch.rand((n_rays, n_samples), generator=g), generator=g)
    ndc = ndc * 1.2 - 0.1
    mlp_sd, _ = load_weights()
    R = torch.randn((n_rays, 3), generator=g); Q = torch.randn((n_rays,), generator=g)
    Wt = torch.randn((n_rays, n_samples), generator=g) * 0.1; A = torch.randn((n_rays, n_samples), generator=g) * 0.1
    return rig, pose, vol, pts, dirs, ndc, z, ro, mlp_sd, (R, Q, Wt, A)


@pytest.mark.parametrize("n_rays,n_samples,white", [(64, 32, False), (37, 16, True), (8, 128, False), (130, 3, False)])
def test_raymarch_backward_vs_autograd(n_rays, n_samples, white):
    import types
    from mvsnerf_amd import models, renderer
    from oracle import mvsnerf_oracle as O
    rig, pose, vol, pts, dirs, ndc, z, ro, mlp_sd, (R, Q, Wt, A) = _setup(n_rays, n_samples, 5 + n_rays, white)

    # ---- reference gradients: autograd through the CPU oracle
    sd = {k: v.clone().requires_grad_(True) for k, v in mlp_sd.items()}
    vol_ref = vol.clone().requires_grad_(True)
    out = O.rendering(pose, pts, ndc, z, dirs, vol_ref, rig["images_raw"][:, :3], sd, white_bkgd=white)
    loss_ref = (out[0] * R).sum() + (out[3] * Q).sum() + (out[2] * Wt).sum() + (out[4] * A).sum()
    loss_ref.backward()

    # ---- HIP path
    args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0,
                                 pts_dim=3, multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024,
                                 ckpt=None, perturb=1.0, N_samples=n_samples, use_viewdirs=True, white_bkgd=white, raw_noise_std=0.0)
    kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(mlp_sd)
    vol_g = models.RefVolume(vol.to(DEV))            # the fine-tuning setup: learnable volume + MLP
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    rgb, feat, w, depth, alpha, _ = renderer.rendering(args, pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV),
                                                       vol_g, rig["images_raw"][:, :3].to(DEV), network_fn=net,
                                                       network_query_fn=kw["network_query_fn"], white_bkgd=white)
    loss = (rgb * R.to(DEV)).sum() + (depth * Q.to(DEV)).sum() + (w * Wt.to(DEV)).sum() + (alpha * A.to(DEV)).sum()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-3 * max(1.0, abs(float(loss_ref.detach())))
    loss.backward()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    errs = {"volume": rel(vol_g.feat_volume.grad, vol_ref.grad)}
    for name, p in net.named_parameters():
        errs[name] = rel(p.grad, sd[name].grad)
    bad = {k: v for k, v in errs.items() if not v < 2e-3}
    assert not bad, f"gradient mismatches (rel. to max |ref|): {bad}\nall: {errs}"


def test_composite_backward_alone():
    from mvsnerf_amd import _lib, ops
    from oracle import mvsnerf_oracle as O
    g = torch.Generator().manual_seed(2)
    for S in (1, 64, 128, 300):
        N = 9
        raw = (torch.rand((N, S, 4), generator=g) * 2).requires_grad_(True)
        z = torch.sort(torch.rand((N, S), generator=g) + 2, -1)[0]
        rgb, disp, acc, w, depth, alpha = O.raw2outputs(raw, z, white_bkgd=True)
        G = [torch.randn(t.shape, generator=g) for t in (rgb, depth, w, alpha)]
        (rgb * G[0]).sum().add((depth * G[1]).sum()).add((w * G[2]).sum()).add((alpha * G[3]).sum()).backward()
        d_raw = torch.empty((N, S, 4), device=DEV)
        keep = [raw.detach().to(DEV), z.to(DEV)] + [t.to(DEV).contiguous() for t in G]     # keep the device copies alive
        rc = _lib.lib().mvsnerf_composite_bwd(keep[0].data_ptr(), keep[1].data_ptr(), N, S, 1, keep[2].data_ptr(),
                                              keep[3].data_ptr(), 0, keep[4].data_ptr(), keep[5].data_ptr(), d_raw.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        err = float((d_raw.cpu() - raw.grad).abs().max())
        assert err < 1e-4 * max(1.0, float(raw.grad.abs().max())), (S, err)


@pytest.mark.parametrize("fused", [False, True])
def test_encoder_backward_vs_autograd(fused):
    """Plane sweep + CostRegNet backward (grads of all conv / ABN parameters and of the source features) against
    PyTorch autograd through the CPU oracle.  Small shapes: features 16x24, pad 4, D=16.
    fused: the single autograd node MVSNet.forward uses in training (cost volume in channel blocks of four, conv0's weight gradient
    on the matrix cores, data gradient of the variance channels only) instead of the two public modules."""
    from mvsnerf_amd import encoder as E, models
    from mvsnerf_amd.synth import make_rig
    from oracle import mvsnerf_oracle as O
    _, sd0 = load_weights()
    rig = make_rig(64, 96, seed=31, rot_deg=2.0, smooth=True)
    pad, D = 4, 16
    imgs, proj = rig["images"][:, :3], rig["proj_mats"][:, :3]
    g = torch.Generator().manual_seed(4)
    feats0 = O.feature_net(imgs[0], sd0)[None].detach()
    dv = O.depth_planes(2.125, 4.525, D)
    Rw = torch.randn((1, 8, D, 16 + 2 * pad, 24 + 2 * pad), generator=g)

    # reference: autograd through the oracle
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd0.items()}
    f_ref = feats0.clone().requires_grad_(True)
    cost_ref, _ = O.build_volume_costvar_img(imgs, f_ref, proj, dv, pad)
    vol_ref = O.cost_reg_net(cost_ref, sd)
    (vol_ref * Rw).sum().backward()

    net = models.MVSNet()
    net.load_state_dict(sd0)
    net = net.to(DEV).train()
    f = feats0.clone().to(DEV).requires_grad_(True)
    if fused:
        vol = E._SweepRegFunction.apply(f, imgs.to(DEV), proj.to(DEV), dv.to(DEV), pad, net.cost_reg_2, *E._costreg_params(net.cost_reg_2))
    else:
        cost, _ = net.build_volume_costvar_img(imgs.to(DEV), f, proj.to(DEV), dv.to(DEV), pad=pad)
        vol = net.cost_reg_2(cost)
    assert float((vol.detach().cpu() - vol_ref.detach()).abs().max()) < 2e-3
    (vol * Rw.to(DEV)).sum().backward()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    errs = {"feats": rel(f.grad, f_ref.grad)}
    for name, p in net.cost_reg_2.named_parameters():
        errs[name] = rel(p.grad, sd["cost_reg_2." + name].grad)
    bad = {k: v for k, v in errs.items() if not v < 5e-3}
    assert not bad, f"encoder gradient mismatches: {bad}\nall: {errs}"


def test_color_volume_backward_vs_autograd():
    """--use_color_volume fine-tuning (reference train_mvs_nerf_finetuning_pl.py:72-82, renderer.py:134-135): the learnable volume has
    8 + 4V = 20 channels and ALL 20 per-sample features come from it, so the MLP backward must return every feature gradient and the
    trilinear scatter runs on 20 channels.  Gradients vs autograd through the oracle (the same lookup + MLP + compositing)."""
    import types
    from mvsnerf_amd import models, renderer
    from oracle import mvsnerf_oracle as O
    n_rays, n_samples = 48, 24
    rig, pose, vol8, pts, dirs, ndc, z, ro, mlp_sd, (R, Q, Wt, A) = _setup(n_rays, n_samples, 77)
    g = torch.Generator().manual_seed(9)
    vol = torch.cat([vol8, torch.rand((1, 12, *vol8.shape[2:]), generator=g)], 1)          # 20 channels

    sd = {k: v.clone().requires_grad_(True) for k, v in mlp_sd.items()}
    vol_ref = vol.clone().requires_grad_(True)
    feat = O.index_point_feature(vol_ref, ndc)                                              # (N,S,20): one lookup, no colour projection
    cos = torch.norm(dirs, dim=-1)
    angle = O.gen_dir_feature(pose["w2cs"][0], dirs / cos.unsqueeze(-1))
    raw = O.run_network_mvs(ndc, angle, feat, sd)
    rgb_r, _, _, w_r, depth_r, alpha_r = O.raw2outputs(raw, z, False)
    loss_ref = (rgb_r * R).sum() + (depth_r * Q).sum() + (w_r * Wt).sum() + (alpha_r * A).sum()
    loss_ref.backward()

    args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=True, net_type="v0", multires=10, i_embed=0,
                                 pts_dim=3, multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024,
                                 ckpt=None, perturb=1.0, N_samples=n_samples, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0)
    kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(mlp_sd)
    vol_g = models.RefVolume(vol.to(DEV))
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    rgb, feat_g, w, depth, alpha, _ = renderer.rendering(args, pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV),
                                                         vol_g, rig["images_raw"][:, :3].to(DEV), network_fn=net,
                                                         network_query_fn=kw["network_query_fn"])
    assert float((feat_g.detach().cpu() - feat.detach()).abs().max()) < 1e-5
    loss = (rgb * R.to(DEV)).sum() + (depth * Q.to(DEV)).sum() + (w * Wt.to(DEV)).sum() + (alpha * A.to(DEV)).sum()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-3 * max(1.0, abs(float(loss_ref.detach())))
    loss.backward()

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    gv, gr = vol_g.feat_volume.grad.cpu(), vol_ref.grad
    errs = {"volume[0:8]": rel(gv[:, :8], gr[:, :8]), "volume[8:20]": rel(gv[:, 8:], gr[:, 8:])}
    assert float(gr[:, 8:].abs().max()) > 0                                                # the colour channels do receive gradient
    for name, p in net.named_parameters():
        errs[name] = rel(p.grad, sd[name].grad)
    bad = {k: v for k, v in errs.items() if not v < 2e-3}
    assert not bad, f"gradient mismatches: {bad}\nall: {errs}"


# ------------------------------------------------------------------ bf16 training (BASELINE config 3; reference AMP switch)
class _BF16Linear(torch.autograd.Function):
    """y = x W^T + b with the operands of every product rounded to bf16 and fp32 accumulation - forward, data gradient and weight
    gradient alike (what v_mfma_f32_32x32x16_bf16 computes); the bias gradient is an fp32 sum.  round_* switch the rounding per GEMM."""

    @staticmethod
    def forward(ctx, x, w, b, round_fwd, round_dgrad, round_wgrad):
        q = lambda t: t.to(torch.bfloat16).to(torch.float32)
        ctx.save_for_backward(x, w)
        ctx.flags = (round_dgrad, round_wgrad)
        return (q(x) if round_fwd else x) @ (q(w) if round_fwd else w).t() + b

    @staticmethod
    def backward(ctx, gy):
        q = lambda t: t.to(torch.bfloat16).to(torch.float32)
        x, w = ctx.saved_tensors
        rd, rw = ctx.flags
        gx = (q(gy) if rd else gy) @ (q(w) if rd else w)
        g2, x2 = gy.reshape(-1, gy.shape[-1]), x.reshape(-1, x.shape[-1])
        gw = (q(g2) if rw else g2).t() @ (q(x2) if rw else x2)
        return gx, gw, g2.sum(0), None, None, None


class _ModReLU16(torch.autograd.Function):
    """h = relu(p * b) as the bf16 training kernels keep it for the backward pass: h and b are stored as bf16 slots (round 4: the fp32
    slots made the three MLP training kernels HBM-bound), the mask comes from the stored h and the gradient of the multiplicative bias
    is g * h / b on the stored values (mlp_bwd.hip, `gbm[q] += gq * (hq[q] / bm[q])`)."""

    @staticmethod
    def forward(ctx, p, b):
        q = lambda t: t.to(torch.bfloat16).to(torch.float32)
        h = torch.relu(p * b)
        ctx.save_for_backward(q(h), q(b))
        return h

    @staticmethod
    def backward(ctx, g):
        h, b = ctx.saved_tensors
        on = h > 0
        gq = torch.where(on, g, torch.zeros_like(g))
        return gq * b, torch.where(on, gq * (h / b), torch.zeros_like(g))


def _renderer_bf16_emulation(ndc, angle, feat, sd):
    """Renderer_ours (models.py:194-222) with the rounding of the bf16 training kernels: the nine wide layers are bf16 GEMMs in
    all three passes; the two heads (alpha_linear, rgb_linear) are fp32 dot products forward and in the data gradient, and go
    through the same bf16 point contraction as every other weight gradient."""
    from oracle import mvsnerf_oracle as O
    big = lambda name, h: _BF16Linear.apply(h, sd[f"nerf.{name}.weight"], sd[f"nerf.{name}.bias"], True, True, True)
    head = lambda name, h: _BF16Linear.apply(h, sd[f"nerf.{name}.weight"], sd[f"nerf.{name}.bias"], False, False, True)
    pts = O.embed(ndc)
    bias = big("pts_bias", feat)
    h = pts
    for i in range(6):
        h = _ModReLU16.apply(big(f"pts_linears.{i}", h), bias)
        if i == 4:
            h = torch.cat([pts, h], -1)
    alpha = torch.relu(head("alpha_linear", h))
    h = torch.cat([big("feature_linear", h), angle[:, None].expand(-1, ndc.shape[1], -1)], -1)
    h = torch.relu(big("views_linears.0", h))
    return torch.cat([torch.sigmoid(head("rgb_linear", h)), alpha], -1)


def test_bf16_training_vs_torch_emulation():
    """ops.set_mlp_precision('bf16') in training: bf16-MFMA forward with activation store, bf16-MFMA data- and weight-gradient
    GEMMs, fp32 accumulation, fp32 master weights / gradients.  Checked against (1) autograd through a torch emulation of exactly
    that rounding - tight - and (2) the fp32 oracle - loose (bf16 keeps 8 mantissa bits)."""
    import types
    from mvsnerf_amd import models, renderer, ops
    from oracle import mvsnerf_oracle as O
    n_rays, n_samples = 96, 32
    rig, pose, vol, pts, dirs, ndc, z, ro, mlp_sd, (R, Q, Wt, A) = _setup(n_rays, n_samples, 123)

    def reference(emulate):
        sd = {k: v.clone().requires_grad_(True) for k, v in mlp_sd.items()}
        vol_ref = vol.clone().requires_grad_(True)
        feat = O.gen_pts_feats(rig["images_raw"][:, :3], vol_ref, pts, pose, ndc)
        angle = O.gen_dir_feature(pose["w2cs"][0], dirs / torch.norm(dirs, dim=-1, keepdim=True))
        raw = _renderer_bf16_emulation(ndc, angle, feat, sd) if emulate else O.run_network_mvs(ndc, angle, feat, sd)
        rgb, _, _, w, depth, alpha = O.raw2outputs(raw, z, False)
        loss = (rgb * R).sum() + (depth * Q).sum() + (w * Wt).sum() + (alpha * A).sum()
        loss.backward()
        return float(loss.detach()), sd, vol_ref
    loss_e, sd_e, vol_e = reference(True)
    loss_f, sd_f, vol_f = reference(False)

    args = types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0,
                                 pts_dim=3, multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024,
                                 ckpt=None, perturb=1.0, N_samples=n_samples, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0)
    kw, _, _, _ = models.create_nerf_mvs(args, use_mvs=False, dir_embedder=False, pts_embedder=True)
    net = kw["network_fn"]
    net.load_state_dict(mlp_sd)
    vol_g = models.RefVolume(vol.to(DEV))
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    ops.set_mlp_precision("bf16")
    try:
        rgb, feat, w, depth, alpha, _ = renderer.rendering(args, pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV),
                                                           vol_g, rig["images_raw"][:, :3].to(DEV), network_fn=net,
                                                           network_query_fn=kw["network_query_fn"])
        loss = (rgb * R.to(DEV)).sum() + (depth * Q.to(DEV)).sum() + (w * Wt.to(DEV)).sum() + (alpha * A.to(DEV)).sum()
        loss.backward()
    finally:
        ops.set_mlp_precision("fp32")
    for p in net.parameters():
        assert p.grad.dtype == torch.float32 and p.dtype == torch.float32          # fp32 master weights and gradients

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))
    assert abs(float(loss.detach()) - loss_e) < 2e-3 * max(1.0, abs(loss_e)), (float(loss.detach()), loss_e)
    errs_e = {"volume": rel(vol_g.feat_volume.grad, vol_e.grad)}
    errs_f = {"volume": rel(vol_g.feat_volume.grad, vol_f.grad)}
    for name, p in net.named_parameters():
        errs_e[name] = rel(p.grad, sd_e[name].grad)
        errs_f[name] = rel(p.grad, sd_f[name].grad)
    print("bf16 training: max rel. gradient error vs bf16 emulation %.2e, vs fp32 oracle %.2e" % (max(errs_e.values()), max(errs_f.values())))
    # the emulation rounds at the same places but sums in another order; a bf16 rounding flips on an fp32-ulp difference, so
    # single operands differ by one bf16 ulp now and then
    bad = {k: v for k, v in errs_e.items() if not v < 5e-3}          # measured 8.7e-4
    assert not bad, f"vs bf16 emulation: {bad}\nall: {errs_e}"
    bad = {k: v for k, v in errs_f.items() if not v < 0.15}          # measured 6.8e-2: the price of bf16 operands
    assert not bad, f"vs fp32 oracle: {bad}\nall: {errs_f}"


@pytest.mark.parametrize("V,H,W,pad,D,with_img,bscale", [(3, 30, 41, 3, 10, True, 1.0), (5, 16, 24, 4, 19, True, 1.0), (2, 32, 32, 0, 8, False, 1.0),
                                                        (8, 20, 28, 2, 9, True, 1.0), (3, 24, 33, 2, 37, False, 6.0), (3, 128, 160, 24, 128, False, 1.0)])
def test_planesweep_bwd_vs_float64_autograd(V, H, W, pad, D, with_img, bscale):
    """The plane-sweep backward (column form: a thread walks the depth planes of one voxel column and channel with the taps and the
    gradient sums of every source view in registers, memory is touched only when the tap set changes) against float64 autograd through
    a torch restatement of models.py:839-893 on the same GPU.  Cases: 1, 2, 4 and 7 source views; widths that are not a multiple of the
    8-column workgroup; depths that are not a multiple of the 16-plane geometry batch; bscale = 6: baselines six times wider, so the taps
    move by more than a pixel per plane and the send / gather path runs on nearly every plane.  The last case is the training shape
    (timed)."""
    import torch.nn.functional as F
    from mvsnerf_amd import _lib
    from mvsnerf_amd.ops import stream_ptr
    from mvsnerf_amd.synth import make_rig
    base = tuple(b * bscale for b in (0.0, 0.25, -0.25, 0.12, -0.12, 0.1, -0.3, 0.3, 0.2))
    rig = make_rig(H * 4, W * 4, n_views=V + 1, seed=77, baselines=base[:V + 1], rot_deg=2.0, smooth=True)
    proj = rig["proj_mats"][0, :V].contiguous().to(DEV)
    nf = rig["near_fars"][0, 0]
    depth = torch.linspace(float(nf[0]), float(nf[1]), D).to(DEV)
    g = torch.Generator(DEV).manual_seed(V * 100 + D)
    feats = torch.randn((V, H, W, 32), device=DEV, generator=g)
    CP = (32 + 3 * V + 3) // 4 * 4 if with_img else 32
    c_var = 3 * V if with_img else 0
    Hp, Wp = H + 2 * pad, W + 2 * pad
    g_cost = torch.randn((D, Hp, Wp, CP), device=DEV, generator=g)
    L = _lib.lib()
    for rep in range(3):
        gf = torch.zeros((V, H, W, 32), device=DEV)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.mvsnerf_planesweep_costvar_bwd(feats.data_ptr(), proj.data_ptr(), depth.data_ptr(), V, 32, H, W, D, pad, g_cost.data_ptr(), CP,
                                              int(with_img), gf.data_ptr(), stream_ptr())
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
    ms = e0.elapsed_time(e1)
    # ---- float64 reference (grid coordinates in fp32 as the kernel computes them, everything downstream in float64)
    f = feats.double().permute(0, 3, 1, 2).contiguous().requires_grad_()                    # (V,32,H,W)
    ys, xs = torch.meshgrid(torch.arange(Hp, device=DEV, dtype=torch.float32) - pad, torch.arange(Wp, device=DEV, dtype=torch.float32) - pad, indexing="ij")
    uv1 = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(Hp * Wp, device=DEV)], 0)   # (3,N)
    ref = F.pad(f[0], (pad, pad, pad, pad))[:, None].expand(-1, D, -1, -1).reshape(32, D, Hp * Wp)
    # the view count per voxel from the HIP forward kernel (a step function of the fp32 grid: torch's own fp32 arithmetic may round a
    # coordinate at the frustum border the other way, and a flipped count is an O(1) difference that says nothing about the gradient)
    cost_tmp, cnt_k = torch.empty((D, Hp, Wp, 32), device=DEV), torch.empty((D, Hp, Wp), device=DEV)
    assert L.mvsnerf_planesweep_costvar_fwd(feats.data_ptr(), 0, proj.data_ptr(), depth.data_ptr(), V, 32, H, W, D, pad, cost_tmp.data_ptr(), 32,
                                            cnt_k.data_ptr(), 0, stream_ptr()) == 0
    del cost_tmp
    cnt = cnt_k.double().reshape(D, Hp * Wp)
    s, s2 = ref, ref ** 2
    for v in range(1, V):
        pr = (proj[v, :, :3] @ uv1)[None] + (proj[v, :, 3:][None] / depth[:, None, None])   # (D,3,N) fp32
        gx = pr[:, 0] / pr[:, 2] / ((W - 1) / 2) - 1
        gy = pr[:, 1] / pr[:, 2] / ((H - 1) / 2) - 1
        grid = torch.stack([gx, gy], -1).double()[None]                                   # (1,D,N,2)
        w = F.grid_sample(f[v:v + 1], grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0]   # (32,D,N)
        s, s2 = s + w, s2 + w ** 2
    var = s2 / cnt - (s / cnt) ** 2
    gv = g_cost[..., c_var:c_var + 32].double().permute(3, 0, 1, 2).reshape(32, D, Hp * Wp)
    (gref,) = torch.autograd.grad((var * gv).sum(), f)
    gref = gref.permute(0, 2, 3, 1)
    scale = float(gref.abs().max())
    err = float((gf.double() - gref).abs().max())
    print(f"[planesweep bwd V={V} {D}x{Hp}x{Wp}] {ms:.3f} ms; max err vs float64 autograd {err:.2e} (|g| max {scale:.1f})")
    assert scale > 0 and err < 2e-5 * scale


@pytest.mark.parametrize("V,H,W,pad,D,with_img,bscale", [(3, 30, 41, 3, 10, True, 1.0), (3, 24, 33, 2, 37, False, 6.0), (4, 128, 160, 24, 128, False, 1.0)])
def test_planesweep_bwd_deterministic_variant(V, H, W, pad, D, with_img, bscale):
    """mvsnerf_planesweep_costvar_bwd_det (64-bit fixed-point accumulators): three runs are bit-identical, the result agrees
    with the float-atomic kernel to the float atomics' own run-to-run noise, a gradient 1e6 times larger (another fixed-point scale) works
    the same, and encoder.PSW_BWD_DETERMINISTIC routes the autograd node through it."""
    from mvsnerf_amd import _lib, encoder
    from mvsnerf_amd.ops import stream_ptr
    from mvsnerf_amd.synth import make_rig
    from tests.util import record_err
    base = tuple(b * bscale for b in (0.0, 0.25, -0.25, 0.12, -0.12, 0.1, -0.3, 0.3, 0.2))
    rig = make_rig(H * 4, W * 4, n_views=V + 1, seed=77, baselines=base[:V + 1], rot_deg=2.0, smooth=True)
    proj = rig["proj_mats"][0, :V].contiguous().to(DEV)
    nf = rig["near_fars"][0, 0]
    depth = torch.linspace(float(nf[0]), float(nf[1]), D).to(DEV)
    g = torch.Generator(DEV).manual_seed(V * 100 + D)
    feats = torch.randn((V, H, W, 32), device=DEV, generator=g)
    CP = (32 + 3 * V + 3) // 4 * 4 if with_img else 32
    g_cost = torch.randn((D, H + 2 * pad, W + 2 * pad, CP), device=DEV, generator=g)
    L = _lib.lib()
    words = L.mvsnerf_planesweep_costvar_bwd_det_workspace_words(V, 32, H, W)
    assert words == (V - 1) * H * W * 32 + 1

    def det(gc):
        gf = torch.zeros((V, H, W, 32), device=DEV)
        ws = torch.zeros(words, device=DEV, dtype=torch.int64)
        assert L.mvsnerf_planesweep_costvar_bwd_det(feats.data_ptr(), proj.data_ptr(), depth.data_ptr(), V, 32, H, W, D, pad, gc.data_ptr(), CP,
                                                    int(with_img), gf.data_ptr(), ws.data_ptr(), stream_ptr()) == 0
        return gf

    def atomic(gc):
        gf = torch.zeros((V, H, W, 32), device=DEV)
        assert L.mvsnerf_planesweep_costvar_bwd(feats.data_ptr(), proj.data_ptr(), depth.data_ptr(), V, 32, H, W, D, pad, gc.data_ptr(), CP,
                                                int(with_img), gf.data_ptr(), stream_ptr()) == 0
        return gf

    a = det(g_cost)
    for _ in range(2):
        assert torch.equal(det(g_cost), a), "two runs of the deterministic variant differ"
    ref = atomic(g_cost)
    scale = float(ref.abs().max())
    err = float((a - ref).abs().max()) / scale
    record_err(f"planesweep_bwd_det_vs_atomic:V{V}:D{D}", err, tol=3e-6)
    assert scale > 0 and err < 3e-6, err                                  # the float atomics' summation-order noise
    big = det(g_cost * 1.0e6)
    assert torch.equal(det(g_cost * 1.0e6), big)
    assert float((big / 1.0e6 - a).abs().max()) / scale < 3e-6
    assert float(det(torch.zeros_like(g_cost)).abs().max()) == 0.0       # all-zero gradient: scale falls back to 1
    # the autograd node
    f0 = feats.permute(0, 3, 1, 2).unsqueeze(0).contiguous()
    imgs = torch.rand((1, V, 3, H * 4, W * 4), device=DEV)
    net = encoder.MVSNet().to(DEV)
    outs = []
    for flag in (True, True, False):
        encoder.PSW_BWD_DETERMINISTIC = flag
        try:
            f = f0.clone().requires_grad_()
            if with_img:
                vol, _ = net.build_volume_costvar_img(imgs, f, proj[None], depth[None], pad)
            else:
                vol, _ = net.build_volume_costvar(f, proj[None], depth[None], pad)
            (vol * vol).sum().backward()
            outs.append(f.grad.clone())
        finally:
            encoder.PSW_BWD_DETERMINISTIC = False
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - outs[2]).abs().max()) <= 3e-6 * float(outs[2].abs().max())


@pytest.mark.parametrize("C,D,H,W,P", [(8, 16, 24, 32, 4096), (8, 128, 176, 208, 131072), (20, 12, 20, 28, 3001)])
def test_volume_sample_bwd_deterministic_variant(C, D, H, W, P):
    """mvsnerf_volume_sample_bwd_det (64-bit fixed-point accumulators): three runs are bit-identical - the float-atomic kernel's
    are not at the headline size -, the result agrees with the float-atomic kernel to its own summation-order noise, a gradient 1e6 times larger (another
    fixed-point scale) works the same, an all-zero gradient adds nothing, and ops.VOLUME_BWD_DETERMINISTIC routes the ray march's autograd node through it."""
    from mvsnerf_amd import _lib, ops
    from mvsnerf_amd.ops import stream_ptr
    from tests.util import record_err
    g = torch.Generator(DEV).manual_seed(C * 1000 + D)
    ndc = (torch.rand((P, 3), device=DEV, generator=g) * 1.1 - 0.05).contiguous()        # a few samples outside the volume
    ndc[: P // 2, :2] = ndc[:1, :2]                                                       # ... and half of them in ONE column: many contributions per voxel
    gf = torch.randn((P, C), device=DEV, generator=g)
    L = _lib.lib()
    words = L.mvsnerf_volume_sample_bwd_det_workspace_words(D, H, W, C)
    assert words == 8 + D * H * W * C

    def det(gfeat):
        gv = torch.zeros((D, H, W, C), device=DEV)
        ws = torch.zeros(words, device=DEV, dtype=torch.int64)
        assert L.mvsnerf_volume_sample_bwd_det(D, H, W, C, ndc.data_ptr(), P, gfeat.data_ptr(), C, gv.data_ptr(), ws.data_ptr(), stream_ptr()) == 0
        return gv

    def atomic(gfeat):
        gv = torch.zeros((D, H, W, C), device=DEV)
        assert L.mvsnerf_volume_sample_bwd(D, H, W, C, ndc.data_ptr(), P, gfeat.data_ptr(), C, gv.data_ptr(), stream_ptr()) == 0
        return gv

    a = det(gf)
    for _ in range(2):
        assert torch.equal(det(gf), a), "two runs of the deterministic variant differ"
    ref = atomic(gf)
    scale = float(ref.abs().max())
    # an exact sum of the kernel's own fp32 contributions (fp32 weights and products as the kernels form them, float64 accumulation by index_add_)
    ix, iy, iz = [((ndc[:, k] * 2.0 - 1.0 + 1.0) / 2.0) * float(n - 1) for k, n in ((0, W), (1, H), (2, D))]
    fx, fy, fz = ix.floor(), iy.floor(), iz.floor()
    exact = torch.zeros(D * H * W * C, device=DEV, dtype=torch.float64)
    for zc in (0, 1):
        for yc in (0, 1):
            for xc in (0, 1):
                cx, cy, cz = fx + xc, fy + yc, fz + zc
                ok = (cx >= 0) & (cx <= W - 1) & (cy >= 0) & (cy <= H - 1) & (cz >= 0) & (cz <= D - 1)
                w = ((ix - fx) if xc else ((fx + 1.0) - ix)) * ((iy - fy) if yc else ((fy + 1.0) - iy)) * ((iz - fz) if zc else ((fz + 1.0) - iz))
                base = ((cz.long() * H + cy.long()) * W + cx.long()) * C
                idx = (base[ok, None] + torch.arange(C, device=DEV)[None]).reshape(-1)
                exact.index_add_(0, idx, (gf[ok] * w[ok, None]).double().reshape(-1))
    exact = exact.view(D, H, W, C)
    err = float((a.double() - exact).abs().max()) / scale
    err_atomic = float((ref.double() - exact).abs().max()) / scale
    record_err(f"volume_sample_bwd_det_vs_exact:C{C}:P{P}", err, tol=2e-5)
    record_err(f"volume_sample_bwd_atomic_vs_exact:C{C}:P{P}", err_atomic, tol=5e-5)
    # (torch forms the fp32 weights / products with its own roundings: ~1 ulp per contribution, up to ~500 contributions per voxel in the crowded column: measured
    # 4e-7 at P = 4096, 4e-6 at P = 131072 - the float-atomic kernel sits at the same distance from this reference; what pins the deterministic variant are the
    # bit-identical repeats above and the scale invariance below)
    assert scale > 0 and err < 2e-5, err
    assert err_atomic < 5e-5, err_atomic                                  # the float atomics: summation-order noise of up to ~500 contributions per voxel here
    big = det(gf * 1.0e6)
    assert torch.equal(det(gf * 1.0e6), big)
    assert float((big / 1.0e6 - a).abs().max()) / scale < 3e-6
    assert float(det(torch.zeros_like(gf)).abs().max()) == 0.0
    # the switch of the Python layer (what RayMarchFunction.backward calls)
    outs = []
    for flag in (True, True, False):
        ops.VOLUME_BWD_DETERMINISTIC = flag
        try:
            gv = torch.zeros((D, H, W, C), device=DEV)
            ops._scatter_hip(gv, ndc, gf)
            outs.append(gv)
        finally:
            ops.VOLUME_BWD_DETERMINISTIC = False
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], a)
    assert float((outs[0] - outs[2]).abs().max()) <= 5e-5 * scale
