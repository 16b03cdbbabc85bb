"""Drop-in replacements for the hot-path helpers of the reference's utils.py (same names, argument
order, defaults and return layouts - SURVEY.md 8b), backed by libmvsnerf_hip.so.

Ray generation: the RNG draws stay here, exactly as in the reference, because the ray indices must be bit-exact
(pixel ids: CPU RNG, utils.py:93; jitter: device RNG, utils.py:220); everything downstream of them in build_rays /
build_rays_test is one HIP kernel (mvsnerf_raygen_fwd).  get_rays_mvs / get_ndc_coordinate keep host-side torch bodies
for callers that use them on their own (and for CPU tensors).
"""
import torch

from . import ops


# ------------------------------------------------------------------ metrics used by training_step
def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10.0 * torch.log(x) / torch.log(torch.tensor([10.0], device=x.device if torch.is_tensor(x) else None))


# ------------------------------------------------------------------ rays (reference utils.py:86-297)
def get_rays_mvs(H, W, intrinsic, c2w, N=1024, isRandom=True, is_precrop_iters=False, chunk=-1, idx=-1):
    """utils.py:86-108.  Pixel ids are drawn on the CPU generator in the order xs then ys, so a seeded run
    picks bit-identical rays to the reference.  Returns rays_o (3,), rays_d (N,3), pixel_coordinates (2,N)=[row,col]."""
    dev = c2w.device
    if isRandom:
        if is_precrop_iters and torch.rand((1,)) > 0.3:
            xs = torch.randint(W // 6, W - W // 6, (N,)).float().to(dev)
            ys = torch.randint(H // 6, H - H // 6, (N,)).float().to(dev)
        else:
            xs = torch.randint(0, W, (N,)).float().to(dev)
            ys = torch.randint(0, H, (N,)).float().to(dev)
    else:
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        ys, xs = gy.reshape(-1), gx.reshape(-1)
        if chunk > 0:
            ys, xs = ys[idx * chunk:(idx + 1) * chunk], xs[idx * chunk:(idx + 1) * chunk]
        ys, xs = ys.to(dev), xs.to(dev)
    cam_dirs = torch.stack([(xs - intrinsic[0, 2]) / intrinsic[0, 0],
                            (ys - intrinsic[1, 2]) / intrinsic[1, 1],
                            torch.ones_like(xs)], -1)
    return c2w[:3, -1].clone(), cam_dirs @ c2w[:3, :3].t(), torch.stack((ys, xs))


def get_ndc_coordinate(w2c_ref, intrinsic_ref, point_samples, inv_scale, near=2, far=6, pad=0, lindisp=False):
    """utils.py:112-146: world points (N_rays,N_samples,3) -> reference-view NDC in [0,1]."""
    n_rays, n_samples = point_samples.shape[:2]
    p = point_samples.reshape(-1, 3)
    if w2c_ref is not None:
        p = torch.matmul(p, w2c_ref[:3, :3].t()) + w2c_ref[:3, 3:].reshape(1, 3)
    if intrinsic_ref is not None:
        q = p @ intrinsic_ref.t()
        xy = (q[:, :2] / q[:, -1:] + 0.0) / inv_scale.reshape(1, 2)
        z = (q[:, 2] - near) / (far - near) if not lindisp else (1.0 / q[:, 2] - 1.0 / near) / (1.0 / far - 1.0 / near)
        q = torch.cat([xy, z[:, None]], -1)
    else:
        q = (p - near.view(1, 3)) / (far.view(1, 3) - near.view(1, 3))
    if pad > 0:
        w_feat, h_feat = (inv_scale + 1) / 4.0
        x = q[:, 0] * w_feat / (w_feat + pad * 2) + pad / (w_feat + pad * 2)
        y = q[:, 1] * h_feat / (h_feat + pad * 2) + pad / (h_feat + pad * 2)
        q = torch.stack([x, y, q[:, 2]], -1)
    return q.view(n_rays, n_samples, 3)


def _stratified(near, far, n_rays, n_samples, device, jitter):
    t = torch.linspace(0.0, 1.0, steps=n_samples).view(1, n_samples).to(device)
    z = (near * (1.0 - t) + far * t).expand([n_rays, n_samples])
    if not jitter:
        return z
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * torch.rand(z.shape, device=device)       # device RNG, as utils.py:220


_INV_SCALE = {}


def _inv_scale(W, H, dev):
    """tensor([W-1, H-1]) on `dev`, uploaded once (a torch.tensor(list).to(dev) per step is a synchronous copy)."""
    key = (W, H, str(dev))
    if key not in _INV_SCALE:
        _INV_SCALE[key] = torch.tensor([W - 1, H - 1]).to(dev)
    return _INV_SCALE[key]


def _nf_pair(near_fars_row):
    return near_fars_row.reshape(-1)[:2].to(torch.float32).contiguous()


def build_rays(imgs, depths, pose_ref, w2cs, c2ws, intrinsics, near_fars, N_rays, N_samples, pad=0,
               is_precrop_iters=False, ref_idx=0, importanceSampling=False, with_depth=False, is_volume=False):
    """utils.py:148-241: random target-view rays for one training step (target = last view).
    Returns (rays_pts, rays_dir, colors, rays_NDC, depth_candidates, rays_o, rays_depth, ndc_parameters).
    The RNG draws happen here exactly as in the reference (pixel ids: CPU generator, xs then ys; jitter: device
    generator), everything downstream is one HIP kernel (mvsnerf_raygen_fwd) on GPU tensors."""
    dev = imgs.device
    _, V, _, H, W = imgs.shape
    w2c_ref, k_ref = pose_ref["w2cs"][ref_idx], pose_ref["intrinsics"][ref_idx]
    inv_scale = _inv_scale(W, H, dev)
    near_ref, far_ref = pose_ref["near_fars"][ref_idx, 0], pose_ref["near_fars"][ref_idx, 1]
    i = V - 1
    if not imgs.is_cuda:
        return _build_rays_torch(imgs, depths, pose_ref, w2cs, c2ws, intrinsics, near_fars, N_rays, N_samples, pad,
                                 is_precrop_iters, ref_idx, importanceSampling, with_depth)
    if is_precrop_iters and torch.rand((1,)) > 0.3:                                  # utils.py:90-93
        xs = torch.randint(W // 6, W - W // 6, (N_rays,))
        ys = torch.randint(H // 6, H - H // 6, (N_rays,))
    else:
        xs = torch.randint(0, W, (N_rays,))
        ys = torch.randint(0, H, (N_rays,))
    # one asynchronous copy from pinned memory (a pageable .to(dev) makes the host wait for everything already enqueued)
    xy = torch.stack((xs, ys)).float().pin_memory().to(dev, non_blocking=True)
    xs, ys = xy[0], xy[1]
    has_depth = depths.shape[2] != 1                                                  # utils.py:194
    if importanceSampling and not has_depth:
        raise RuntimeError("build_rays(importanceSampling=True) needs the ground-truth depth maps (utils.py:203 reads rays_depth)")
    # with_depth (:199-200): the reference indexes `near_fars` itself with the pixel ids, i.e. expects an (H,W) depth map there
    z_map = near_fars.to(torch.float32) if with_depth else None
    if with_depth and tuple(z_map.shape) != (H, W):
        raise RuntimeError(f"build_rays(with_depth=True): near_fars must be the (H,W) = ({H},{W}) depth map (utils.py:200), got {tuple(near_fars.shape)}")
    # utils.py:220: the jitter is drawn only on the stratified branches (not for with_depth)
    t_rand = None if with_depth else torch.rand((N_rays, N_samples), device=dev)
    nf_tgt = _nf_pair(pose_ref["near_fars"][ref_idx]) if with_depth else _nf_pair(near_fars[0, i])   # unused by depth modes 1 / 2
    pts, rays_d, ndc, z, pix, colors, rays_depth = ops.raygen_train(
        H, W, intrinsics[i], c2ws[i], k_ref, w2c_ref, nf_tgt, _nf_pair(pose_ref["near_fars"][ref_idx]), N_samples, xs, ys, t_rand,
        imgs[0, i], depth_map=depths[0, i].to(torch.float32) if has_depth else None, z_map=z_map,
        depth_mode=2 if with_depth else (1 if importanceSampling else 0), pad=pad)
    rays_o = c2ws[i][:3, -1].reshape(3, 1).expand(3, N_rays)
    ndc_parameters = {"w2c_ref": w2c_ref, "intrinsic_ref": k_ref, "inv_scale": inv_scale, "near": near_ref, "far": far_ref}
    return pts, rays_d, colors, ndc, z, rays_o, rays_depth, ndc_parameters


def _build_rays_torch(imgs, depths, pose_ref, w2cs, c2ws, intrinsics, near_fars, N_rays, N_samples, pad,
                      is_precrop_iters, ref_idx, importanceSampling, with_depth):
    """Host-side torch version of build_rays for the rarely used variants (per-pixel depth ranges) and CPU tensors."""
    dev = imgs.device
    _, V, _, H, W = imgs.shape
    w2c_ref, k_ref = pose_ref["w2cs"][ref_idx], pose_ref["intrinsics"][ref_idx]
    inv_scale = torch.tensor([W - 1, H - 1]).to(dev)
    near_ref, far_ref = pose_ref["near_fars"][ref_idx, 0], pose_ref["near_fars"][ref_idx, 1]
    i = V - 1
    rays_o, rays_d, pix = get_rays_mvs(H, W, intrinsics[i], c2ws[i].clone(), N_rays, is_precrop_iters=is_precrop_iters)
    pix_i = pix.long()
    colors = imgs[0, i, :, pix_i[0], pix_i[1]].permute(1, 0)
    rays_depth = depths[0, i, pix_i[0], pix_i[1]] if depths.shape[2] != 1 else None
    if with_depth:
        z = near_fars[pix_i[0], pix_i[1]].reshape(-1, 1)
    else:
        if importanceSampling:
            near, far = (rays_depth - 0.1).view(N_rays, 1), (rays_depth + 0.1).view(N_rays, 1)
        else:
            near, far = near_fars[0, i, 0], near_fars[0, i, 1]
        z = _stratified(near, far, N_rays, N_samples, dev, jitter=True)
    o = rays_o.reshape(1, 3).expand(N_rays, -1)
    pts = o.unsqueeze(1) + z.unsqueeze(-1) * rays_d.unsqueeze(1)
    ndc = get_ndc_coordinate(w2c_ref, k_ref, pts, inv_scale, near=near_ref, far=far_ref, pad=pad)
    ndc_parameters = {"w2c_ref": w2c_ref, "intrinsic_ref": k_ref, "inv_scale": inv_scale, "near": near_ref, "far": far_ref}
    return pts, rays_d, colors, ndc, z, o.permute(1, 0), rays_depth, ndc_parameters


def build_rays_test(H, W, tgt_to_world, world_to_ref, intrinsic, near_fars_ref, near_fars, N_samples, pad=0, ref_idx=0,
                    use_cpu=False, chunk=-1, idx=-1):
    """utils.py:243-297: deterministic row-major rays of one chunk, no jitter.  On the GPU: one HIP kernel (the
    reference rebuilds a full-frame meshgrid for every chunk, utils.py:95-98)."""
    dev = torch.device("cpu") if use_cpu else tgt_to_world.device
    if use_cpu or dev.type != "cuda":
        return _build_rays_test_torch(H, W, tgt_to_world, world_to_ref, intrinsic, near_fars_ref, near_fars, N_samples, pad, ref_idx, use_cpu, chunk, idx)
    inv_scale = torch.tensor([W - 1, H - 1]).to(dev)
    k_render = intrinsic if intrinsic.dim() == 2 else intrinsic.mean(0)
    first = 0 if chunk < 0 else idx * chunk
    n = H * W if chunk < 0 else max(0, min(chunk, H * W - first))
    pts, rays_d, ndc, z, _ = ops.raygen(H, W, k_render, tgt_to_world, k_render, world_to_ref, _nf_pair(near_fars),
                                        _nf_pair(near_fars_ref[ref_idx]), N_samples, pad=pad, first_pixel=first, n_rays=n)
    o = tgt_to_world[:3, -1].reshape(1, 3).expand(n, -1)
    near, far = near_fars_ref[ref_idx, 0], near_fars_ref[ref_idx, 1]
    ndc_parameters = {"w2c_ref": world_to_ref, "intrinsic_ref": intrinsic, "inv_scale": inv_scale, "near": near, "far": far}
    return pts, rays_d, ndc, z, o, ndc_parameters


def _build_rays_test_torch(H, W, tgt_to_world, world_to_ref, intrinsic, near_fars_ref, near_fars, N_samples, pad=0, ref_idx=0,
                           use_cpu=False, chunk=-1, idx=-1):
    dev = torch.device("cpu") if use_cpu else tgt_to_world.device
    if use_cpu:
        tgt_to_world, world_to_ref, intrinsic = tgt_to_world.cpu(), world_to_ref.cpu(), intrinsic.cpu()
        near_fars_ref, near_fars = near_fars_ref.cpu(), near_fars.cpu()
    inv_scale = torch.tensor([W - 1, H - 1]).to(dev)
    k_render = intrinsic if intrinsic.dim() == 2 else intrinsic.mean(0)
    rays_o, rays_d, pix = get_rays_mvs(H, W, k_render, tgt_to_world, isRandom=False, chunk=chunk, idx=idx)
    n = H * W if chunk < 0 else pix.shape[-1]
    z = _stratified(near_fars[0], near_fars[1], n, N_samples, dev, jitter=False)
    o = rays_o.reshape(1, 3).expand(n, -1)
    pts = o.unsqueeze(1) + z.unsqueeze(-1) * rays_d.unsqueeze(1)
    near, far = near_fars_ref[ref_idx, 0], near_fars_ref[ref_idx, 1]
    ndc = get_ndc_coordinate(world_to_ref, intrinsic, pts, inv_scale, near=near, far=far, pad=pad)
    ndc_parameters = {"w2c_ref": world_to_ref, "intrinsic_ref": intrinsic, "inv_scale": inv_scale, "near": near, "far": far}
    return pts, rays_d, ndc, z, o, ndc_parameters


# ------------------------------------------------------------------ gathers (HIP)
def index_point_feature(volume_feature, ray_coordinate_ref, chunk=-1):
    """utils.py:357-383: trilinear lookup of the (1,C,D,h,w) volume at NDC (x,y,z) in [0,1].
    Returns (N_rays, N_samples, C) (squeezed like the reference).  `chunk` is accepted and ignored:
    the kernel never materialises anything chunk-sized."""
    ops._need_no_grad(volume_feature, op="index_point_feature (volume_sample)")         # on the caller's tensor: the channel-last view is always detached
    vol_cl = ops.channels_last_volume(volume_feature)
    ndc = ray_coordinate_ref.to(vol_cl.device, torch.float32).contiguous()
    return ops.volume_sample(vol_cl, ndc).squeeze()


def build_color_volume(point_samples, pose_ref, imgs, img_feat=None, downscale=1.0, with_mask=False):
    """utils.py:300-332: per-view projected colours [+ img_feat channels] (+ strict in-frustum mask) -> (N_rays,N_samples,V*C).
    `downscale` is accepted and unused, as in the reference (its interpolate line is commented out, utils.py:319)."""
    V = imgs.shape[1]
    if img_feat is not None:          # (1,V,Cf,Hf,Wf): sampled at the same grid with zeros padding (:322)
        return ops.color_feat_sample(imgs[0].contiguous(), img_feat[0, :V].contiguous(), pose_ref["w2cs"][:V].contiguous(),
                                     pose_ref["intrinsics"][:V].contiguous(), point_samples.contiguous(), with_mask=with_mask)
    return ops.color_sample(imgs[0].contiguous(), pose_ref["w2cs"][:V].contiguous(), pose_ref["intrinsics"][:V].contiguous(),
                            point_samples.contiguous(), with_mask=with_mask)


def get_ptsvolume(H, W, D, pad, near_far, intrinsic, c2w):
    """utils.py:338-355: world positions of the voxel centres of the (D, H+2p, W+2p) reference-frustum volume, as
    (D*(H+2p), W+2p, 3).  Once-per-scene set-up arithmetic (host-side torch like the reference)."""
    dev = intrinsic.device
    near, far = near_far
    corners = torch.tensor([[-pad, -pad, 1.0], [W + pad, -pad, 1.0], [-pad, H + pad, 1.0], [W + pad, H + pad, 1.0]], device=dev)
    corners = torch.matmul(corners, torch.inverse(intrinsic).t())
    xs_l = torch.linspace(float(corners[0, 0]), float(corners[1, 0]), W + 2 * pad)
    ys_l = torch.linspace(float(corners[0, 1]), float(corners[2, 1]), H + 2 * pad)
    ys, xs = torch.meshgrid(ys_l, xs_l, indexing="ij")
    plane = torch.stack((xs, ys, torch.ones_like(xs)), dim=-1).to(dev)
    lz = torch.linspace(1.0, 0.0, D).view(D, 1, 1, 1).to(dev)
    pts = lz * (plane * near) + (1.0 - lz) * (plane * far)
    pts = torch.matmul(pts.view(-1, 3), c2w[:3, :3].t()) + c2w[:3, 3].view(1, 3)
    return pts.view(D * (H + pad * 2), W + pad * 2, 3)


def normal_vect(vect, dim=-1):
    return vect / (torch.sqrt(torch.sum(vect ** 2, dim=dim, keepdim=True)) + 1e-7)


from .encoder import homo_warp  # noqa: E402,F401  (reference utils.py:580; implemented next to the plane-sweep kernels)
