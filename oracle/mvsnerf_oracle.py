"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.

A functional fp32 restatement (PyTorch *CPU* ops, no modules, no autograd state) of the
MVSNeRF rendering hot path of apchenstu/mvsnerf.  Every function cites the reference
file:line (under /root/reference) whose arithmetic it follows.  Only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import this file - and only as the
checker.  The product path (mvsnerf_amd/) never imports it and has no CPU fallback.

Pinning: tests/test_oracle_golden.py checks every function here against fixtures in
tests/golden/*.npz that were produced by running the reference's own code (imported
behind oracle/ref_shim.py, script oracle/gen_golden.py) on seeded synthetic inputs.

Third-party arithmetic not in the reference tree: `inplace_abn.InPlaceABN` (mapillary,
version unpinned by the reference).  Restated from its published semantics
(gamma=|w|+eps, eps=1e-5, leaky-ReLU 0.01, batch statistics in train mode) - PARITY
UNPINNED for that one op; see DESIGN.md.

Weights are passed as a flat dict name->tensor with the checkpoint's key names
(`nerf.pts_linears.0.weight`, `cost_reg_2.conv0.bn.weight`, ...).

Two conventions the reference leaves undefined are *defined* here:
  * build_volume_costvar_img leaves the padded border of channels 0:3 uninitialised
    (torch.empty, models.py:858-860); the oracle zero-fills it.
  * InPlaceABN: gamma = |weight| + eps.
"""
import math

import torch
import torch.nn.functional as F

ABN_EPS = 1e-5
ABN_SLOPE = 0.01


def _dt():
    return torch.get_default_dtype()


class precision:
    """`with precision(torch.float64):` evaluates the same restatement in double precision (inputs and weights must be passed
    as doubles).  Not the reference's arithmetic - the reference runs fp32 - but the yardstick for it: the distance between
    the fp32 oracle and the fp64 evaluation is the rounding noise of the reference's own fp32 path, and a HIP result that
    is as close to the fp64 evaluation as the fp32 oracle is has nothing left to fix (tests/test_gpu_headline_parity.py)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = torch.get_default_dtype()
        torch.set_default_dtype(self.dtype)

    def __exit__(self, *a):
        torch.set_default_dtype(self.prev)


def _conv3d_s1(x, w):
    """F.conv3d(stride 1, padding 1).  fp64 on CPU has no oneDNN kernel and ATen's fallback unfolds the whole input
    (41 x 27 x 4.7 M doubles = 41 GB at the headline shape): evaluate in depth slabs with a one-plane halo."""
    if x.dtype != torch.float64 or x.shape[1] * x.shape[2] * x.shape[3] * x.shape[4] < (1 << 26):
        return F.conv3d(x, w, None, stride=1, padding=1)
    D = x.shape[2]
    step = max(1, (1 << 26) // (x.shape[1] * x.shape[3] * x.shape[4]))
    out = []
    for d0 in range(0, D, step):
        d1 = min(D, d0 + step)
        lo, hi = max(0, d0 - 1), min(D, d1 + 1)
        slab = F.pad(x[:, :, lo:hi], (0, 0, 0, 0, 1 if d0 == 0 else 0, 1 if d1 == D else 0))
        out.append(F.conv3d(slab, w, None, stride=1, padding=(0, 1, 1)))
    return torch.cat(out, 2)


# --------------------------------------------------------------------------- encoder (L1a)

def abn(x, sd, prefix, training=True, update_running=False, momentum=0.1):
    """InPlaceABN forward (third-party, see header).  Call sites models.py:668,681,742,747,752.
    training=True is what the reference uses even at inference (train_mvs_nerf_pl.py:182)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not update_running:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, w.abs() + ABN_EPS, b, training, momentum, ABN_EPS)
    return F.leaky_relu(y, ABN_SLOPE)


def feature_net(x, sd, prefix="feature."):
    """FeatureNet.forward models.py:715-722 (2-D CNN, ConvBnReLU models.py:661-672)."""
    def cbr(x, name, k, s, p):
        x = F.conv2d(x, sd[prefix + name + ".conv.weight"], None, stride=s, padding=p)
        return abn(x, sd, prefix + name + ".bn")
    x = cbr(x, "conv0.0", 3, 1, 1)
    x = cbr(x, "conv0.1", 3, 1, 1)
    x = cbr(x, "conv1.0", 5, 2, 2)
    x = cbr(x, "conv1.1", 3, 1, 1)
    x = cbr(x, "conv1.2", 3, 1, 1)
    x = cbr(x, "conv2.0", 5, 2, 2)
    x = cbr(x, "conv2.1", 3, 1, 1)
    x = cbr(x, "conv2.2", 3, 1, 1)
    return F.conv2d(x, sd[prefix + "toplayer.weight"], sd[prefix + "toplayer.bias"])


def _fma32(a, b, c):
    """fl32(a*b + c) with ONE rounding: the product of two fp32 numbers is exact in float64."""
    return (a.double() * b.double() + c.double()).to(torch.float32)


def _sgemm_k3(R, X):
    """R (B,3,3) @ X (B,3,N) in the arithmetic the reference's bmm (utils.py:612) has on the authoring host, the one the golden fixtures
    pin: per output element the k-ordered chain  r0*x0 -> fma(r1, x1, .) -> fma(r2, x2, .)  (Intel MKL sgemm on the Xeon that ran
    oracle/gen_golden.py; tests/test_oracle_golden.py compares the resulting sampling grid with the reference-generated one BIT FOR BIT).
    The bits of `R @ X` itself depend on the host's BLAS code path - on the GPU box's AMD EPYC the same MKL rounds the two products and
    the sums separately, 8 % of the grid values of a rotated view then differ in the last bit (scratch/keep/grid_bits_gpu.py,
    gpurun_out/r3_grid_bits.txt) - so the oracle spells the pinned arithmetic out instead of inheriting whatever the host it runs on does.
    float64 mode (`precision`): a plain matmul."""
    if R.dtype != torch.float32:
        return R @ X
    acc = R[:, :, 0:1] * X[:, 0:1]
    acc = _fma32(R[:, :, 1:2], X[:, 1:2], acc)
    return _fma32(R[:, :, 2:3], X[:, 2:3], acc)


def _rows_times_mat3_t(p, M):
    """p (n,3) @ M(3,3).t() in the pinned arithmetic of the authoring host's sgemm (see _sgemm_k3): out[:, j] =
    fma(p2, M[j,2], fma(p1, M[j,1], p0 * M[j,0])).  tests/test_oracle_golden.py: the reference-generated NDC coordinates
    (utils.py:124,128 inside build_rays) are reproduced bit for bit; scratch/keep/cpu_lookup_probe.py lists the alternatives that are not."""
    if p.dtype != torch.float32 or M.dtype != torch.float32:
        return p @ M.t()
    cols = []
    for j in range(3):
        acc = p[:, 0] * M[j, 0]
        acc = _fma32(p[:, 1], M[j, 1], acc)
        cols.append(_fma32(p[:, 2], M[j, 2], acc))
    return torch.stack(cols, -1)


def homo_warp(src_feat, proj_mat, depth_values, src_grid=None, pad=0):
    """utils.py:580-630.  src_feat (B,C,H,W); proj_mat (B,3,4); depth_values (B,D).
    Returns warped (B,C,D,H+2p,W+2p) and the sampling grid (B,D,(W+2p),(H+2p),2) [x,y in -1..1]
    (the reference's view naming at :622 swaps H/W but the flat order is d,y,x)."""
    B, C, H, W = src_feat.shape
    if src_grid is None:
        Hp, Wp = H + 2 * pad, W + 2 * pad
        D = depth_values.shape[1]
        R, T = proj_mat[:, :, :3], proj_mat[:, :, 3:]                       # :600-601
        ys, xs = torch.meshgrid(torch.arange(Hp, dtype=_dt()) - pad,
                                torch.arange(Wp, dtype=_dt()) - pad, indexing="ij")  # :603-605
        uv1 = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(Hp * Wp)], 0)[None].expand(B, -1, -1)
        uv1 = uv1.repeat(1, 1, D)                                            # :611  (B,3,D*Hp*Wp) order d,y,x
        dv = depth_values[:, :, None].expand(B, D, Hp * Wp).reshape(B, 1, -1)
        p = _sgemm_k3(R, uv1) + T / dv                                       # :612
        g = p[:, :2] / p[:, 2:]                                              # :617
        gx = g[:, 0] / ((W - 1) / 2) - 1                                     # :619 un-padded W
        gy = g[:, 1] / ((H - 1) / 2) - 1                                     # :620
        src_grid = torch.stack([gx, gy], -1).view(B, D, Wp, Hp, 2)          # :621-622
    B, D, Wp, Hp = src_grid.shape[:4]
    warped = F.grid_sample(src_feat, src_grid.view(B, D, Wp * Hp, 2), mode="bilinear",
                           padding_mode="zeros", align_corners=True)         # :625
    return warped.view(B, -1, D, Hp, Wp), src_grid


def build_volume_costvar(feats, proj_mats, depth_values, pad=0):
    """MVSNet.build_volume_costvar models.py:787-837 (no colours, mask summed)."""
    B, V, C, H, W = feats.shape
    D = depth_values.shape[1]
    ref = feats[:, 0]
    if pad > 0:
        ref = F.pad(ref, (pad, pad, pad, pad))
    ref = ref.unsqueeze(2).expand(-1, -1, D, -1, -1)
    s, s2 = ref.clone(), ref ** 2
    cnt = torch.ones(B, 1, D, H + 2 * pad, W + 2 * pad)
    for v in range(1, V):
        warped, grid = homo_warp(feats[:, v], proj_mats[:, v], depth_values, pad=pad)
        grid = grid.view(B, 1, D, H + 2 * pad, W + 2 * pad, 2)
        inb = ((grid > -1.0) & (grid < 1.0)).all(-1).to(_dt())               # :819-820
        cnt = cnt + inb
        s, s2 = s + warped, s2 + warped ** 2
    inv = 1.0 / cnt
    return s2 * inv - (s * inv) ** 2, cnt                                   # :833-834


def build_volume_costvar_img(imgs, feats, proj_mats, depth_values, pad=0):
    """MVSNet.build_volume_costvar_img models.py:839-893 - the variant forward() calls.
    imgs (B,V,3,Hi,Wi) ImageNet-normalised; feats (B,V,32,H,W).  Returns img_feat (B,3V+32,D,Hp,Wp)
    [0:3 ref rgb (border := 0, see header), 3:3V warped src rgb, last 32 variance], in_masks (B,V,D,Hp,Wp)."""
    B, V, C, H, W = feats.shape
    D = depth_values.shape[1]
    Hp, Wp = H + 2 * pad, W + 2 * pad
    small = F.interpolate(imgs.reshape(B * V, *imgs.shape[2:]), (H, W), mode="bilinear",
                          align_corners=False).view(B, V, -1, H, W)         # :859
    out = torch.zeros(B, 3 * V + C, D, Hp, Wp)
    out[:, :3, :, pad:H + pad, pad:W + pad] = small[:, 0].unsqueeze(2)      # :860
    ref = feats[:, 0]
    if pad > 0:
        ref = F.pad(ref, (pad, pad, pad, pad))                               # :856
    ref = ref.unsqueeze(2).expand(-1, -1, D, -1, -1)
    s, s2 = ref.clone(), ref ** 2                                            # :862-865
    masks = torch.ones(B, V, D, Hp, Wp)
    for v in range(1, V):
        warped, grid = homo_warp(feats[:, v], proj_mats[:, v], depth_values, pad=pad)       # :871
        out[:, 3 * v:3 * v + 3], _ = homo_warp(small[:, v], proj_mats[:, v], depth_values, src_grid=grid, pad=pad)  # :872
        g = grid.view(B, D, Hp, Wp, 2)
        masks[:, v] = ((g > -1.0) & (g < 1.0)).all(-1).to(_dt())              # :875-877
        s, s2 = s + warped, s2 + warped ** 2                                 # :880-881
    inv = 1.0 / masks.sum(1, keepdim=True)                                   # :889
    out[:, -C:] = s2 * inv - (s * inv) ** 2                                  # :890
    return out, masks


def cost_reg_net(x, sd, prefix="cost_reg_2."):
    """CostRegNet.forward models.py:756-769; ConvBnReLU3D :674-685.  ABN follows every conv
    including the three transposed ones, *before* the skip add (:762-766)."""
    def cbr(x, name, stride=1):
        w = sd[prefix + name + ".conv.weight"]
        x = _conv3d_s1(x, w) if stride == 1 else F.conv3d(x, w, None, stride=stride, padding=1)
        return abn(x, sd, prefix + name + ".bn")

    def up(x, name):
        x = F.conv_transpose3d(x, sd[prefix + name + ".0.weight"], None, stride=2, padding=1, output_padding=1)
        return abn(x, sd, prefix + name + ".1")
    c0 = cbr(x, "conv0")
    c2 = cbr(cbr(c0, "conv1", 2), "conv2")
    c4 = cbr(cbr(c2, "conv3", 2), "conv4")
    y = cbr(cbr(c4, "conv5", 2), "conv6")
    y = c4 + up(y, "conv7")
    y = c2 + up(y, "conv9")
    return c0 + up(y, "conv11")


def depth_planes(near, far, D, lindisp=False):
    """models.py:914-922 (D is hard-coded 128 there)."""
    t = torch.linspace(0.0, 1.0, D)
    if lindisp:
        return (1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t))[None]
    return (near * (1.0 - t) + far * t)[None]


def mvsnet_forward(imgs, proj_mats, near_far, sd, pad=0, D=128, lindisp=False):
    """MVSNet.forward models.py:895-932.  Returns (volume (1,8,D,h,w), feats (B,V,32,h,w), depth_values (1,D),
    cost volume (B,3V+32,D,h,w), in_masks)."""
    B, V, _, H, W = imgs.shape
    feats = feature_net(imgs.reshape(B * V, 3, H, W), sd)
    feats = feats.view(B, V, *feats.shape[1:])
    dv = depth_planes(float(near_far[0]), float(near_far[1]), D, lindisp)
    cost, masks = build_volume_costvar_img(imgs, feats, proj_mats, dv, pad)
    vol = cost_reg_net(cost, sd)
    return vol.reshape(1, -1, *vol.shape[2:]), feats, dv, cost, masks


# --------------------------------------------------------------------------- rays (L2, host-side in reference)

def get_rays_mvs(H, W, intrinsic, c2w, N=1024, isRandom=True, chunk=-1, idx=-1, generator=None):
    """utils.py:86-108.  Pixel ids come from the CPU RNG (torch.randint, :93); pass `generator`
    (or seed the global RNG) so both sides draw identical ids.  Returns rays_o (3,), rays_d (N,3), pix (2,N)=[row,col]."""
    if isRandom:
        xs = torch.randint(0, W, (N,), generator=generator).float()
        ys = torch.randint(0, H, (N,), generator=generator).float()
    else:
        ys, xs = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        if chunk > 0:
            ys, xs = ys[idx * chunk:(idx + 1) * chunk], xs[idx * chunk:(idx + 1) * chunk]
    dirs = torch.stack([(xs - intrinsic[0, 2]) / intrinsic[0, 0], (ys - intrinsic[1, 2]) / intrinsic[1, 1],
                        torch.ones_like(xs)], -1)                            # :101
    return c2w[:3, -1].clone(), dirs @ c2w[:3, :3].t(), torch.stack((ys, xs))


def get_ndc_coordinate(w2c_ref, intrinsic_ref, pts, inv_scale, near=2, far=6, pad=0, lindisp=False):
    """utils.py:112-146.  pts (N,S,3) world -> (N,S,3) in [0,1]: x/(W-1), y/(H-1), (z-near)/(far-near),
    then the pad re-scaling of x,y (:140-143)."""
    N, S = pts.shape[:2]
    p = pts.reshape(-1, 3)
    if w2c_ref is not None:
        p = _rows_times_mat3_t(p, w2c_ref[:3, :3]) + w2c_ref[:3, 3].reshape(1, 3)   # :124
    q = _rows_times_mat3_t(p, intrinsic_ref)                                 # :128
    xy = q[:, :2] / q[:, 2:] / inv_scale.reshape(1, 2)                       # :129
    if lindisp:
        z = (1.0 / q[:, 2] - 1.0 / near) / (1.0 / far - 1.0 / near)
    else:
        z = (q[:, 2] - near) / (far - near)                                  # :131
    x, y = xy[:, 0], xy[:, 1]
    if pad > 0:
        Wf, Hf = (inv_scale + 1) / 4.0                                       # :141
        y = y * Hf / (Hf + pad * 2) + pad / (Hf + pad * 2)
        x = x * Wf / (Wf + pad * 2) + pad / (Wf + pad * 2)
    return torch.stack([x, y, z], -1).view(N, S, 3)


def stratified_depths(near, far, N_rays, N_samples, t_rand=None):
    """utils.py:211-221.  t_rand (N_rays,N_samples) in [0,1) supplied by the caller (the reference draws it
    with the *device* RNG at :220); None = no jitter (build_rays_test, utils.py:279-282)."""
    t = torch.linspace(0.0, 1.0, N_samples).view(1, N_samples)
    z = (near * (1.0 - t) + far * t).expand(N_rays, N_samples)
    if t_rand is None:
        return z
    mids = 0.5 * (z[:, 1:] + z[:, :-1])
    upper = torch.cat([mids, z[:, -1:]], -1)
    lower = torch.cat([z[:, :1], mids], -1)
    return lower + (upper - lower) * t_rand


def build_rays(imgs, pose_ref, near_fars, N_rays, N_samples, pad=0, t_rand=None, generator=None, tgt=-1,
               depths=None, importanceSampling=False, with_depth=False):
    """utils.py:148-241.  imgs (1,V,3,H,W) un-normalised; pose_ref dict of w2cs/c2ws/intrinsics (V,..), near_fars (1,V,2)
    (with_depth=True: the (H,W) map the reference indexes with the pixel ids, :200); depths None or (1,V,H,W).
    Returns rays_pts, rays_dir, target_rgb, rays_ndc, depth_candidates, rays_o (3,N) , pixel ids (2,N) long
    [, rays_depth (N,) when depths is given]."""
    _, V, _, H, W = imgs.shape
    tgt = tgt % V
    inv_scale = torch.tensor([W - 1, H - 1], dtype=_dt())
    rays_o, rays_d, pix = get_rays_mvs(H, W, pose_ref["intrinsics"][tgt], pose_ref["c2ws"][tgt], N_rays, generator=generator)
    pix_i = pix.long()
    target = imgs[0, tgt][:, pix_i[0], pix_i[1]].permute(1, 0)              # :194,233
    rays_depth = None if depths is None else depths[0, tgt, pix_i[0], pix_i[1]]        # :194-196
    if with_depth:
        z = near_fars[pix_i[0], pix_i[1]].reshape(-1, 1)                   # :199-200
    else:
        if importanceSampling:
            near, far = (rays_depth - 0.1).view(N_rays, 1), (rays_depth + 0.1).view(N_rays, 1)   # :202-204
        else:
            near, far = near_fars[0, tgt, 0], near_fars[0, tgt, 1]          # :206
        z = stratified_depths(near, far, N_rays, N_samples, t_rand)
    ro = rays_o.reshape(1, 3).expand(N_rays, -1)
    pts = ro.unsqueeze(1) + z.unsqueeze(-1) * rays_d.unsqueeze(1)           # :223
    nr, fr = pose_ref["near_fars"][0, 0], pose_ref["near_fars"][0, 1]       # :173 ref view 0
    ndc = get_ndc_coordinate(pose_ref["w2cs"][0], pose_ref["intrinsics"][0], pts, inv_scale, near=nr, far=fr, pad=pad)
    if depths is not None:
        return pts, rays_d, target, ndc, z, ro.permute(1, 0), pix_i, rays_depth
    return pts, rays_d, target, ndc, z, ro.permute(1, 0), pix_i


def build_rays_test(H, W, tgt_to_world, world_to_ref, intrinsic, near_fars_ref, near_fars, N_samples, pad=0, chunk=-1, idx=-1,
                    ref_intrinsic=None, ref_hw=None):
    """utils.py:243-297: deterministic row-major pixels, no jitter.
    ref_intrinsic / ref_hw are NOT in the reference (it normalises the reference-view NDC with the target's intrinsics and
    size, utils.py:252-253,288): they restate the same arithmetic for a target grid that differs from the source views
    (BASELINE config 5); None reproduces the reference."""
    Hr, Wr = (H, W) if ref_hw is None else ref_hw
    inv_scale = torch.tensor([Wr - 1, Hr - 1], dtype=_dt())
    rays_o, rays_d, pix = get_rays_mvs(H, W, intrinsic, tgt_to_world, isRandom=False, chunk=chunk, idx=idx)
    n = pix.shape[-1]
    z = stratified_depths(near_fars[0], near_fars[1], n, N_samples, None)
    ro = rays_o.reshape(1, 3).expand(n, -1)
    pts = ro.unsqueeze(1) + z.unsqueeze(-1) * rays_d.unsqueeze(1)
    ndc = get_ndc_coordinate(world_to_ref, intrinsic if ref_intrinsic is None else ref_intrinsic, pts, inv_scale,
                             near=near_fars_ref[0, 0], far=near_fars_ref[0, 1], pad=pad)
    return pts, rays_d, ndc, z, ro


# --------------------------------------------------------------------------- ray march (L1b)

def gen_dir_feature(w2c_ref, rays_dir):
    """renderer.py:111-122."""
    return rays_dir @ w2c_ref[:3, :3].t()


def index_point_feature(volume, ndc):
    """utils.py:357-383 / RefVolume.forward models.py:941-950: trilinear, zeros padding, align_corners.
    volume (1,C,D,h,w); ndc (N,S,3) [x->w, y->h, z->D] in [0,1]  ->  (N,S,C)."""
    N, S = ndc.shape[:2]
    grid = ndc.view(-1, 1, N, S, 3) * 2 - 1.0                                # :381
    f = F.grid_sample(volume, grid, align_corners=True, mode="bilinear")     # :382
    return f[:, :, 0].permute(2, 3, 0, 1).reshape(N, S, -1)


def build_color_volume(pts, pose_ref, imgs, with_mask=True, img_feat=None):
    """utils.py:300-332.  pts (N,S,3) world; imgs (1,V,3,H,W) un-normalised; img_feat None or (1,V,Cf,Hf,Wf).
    Per view: project (:316), bilinear with *border* padding (:320), [img_feat at the same grid with *zeros* padding (:322),]
    strict in-bounds mask (:325-326).  -> (N,S,V*(3+Cf+mask))."""
    _, V, C, H, W = imgs.shape
    inv_scale = torch.tensor([W - 1, H - 1], dtype=_dt())
    Cv = C + int(with_mask) + (0 if img_feat is None else img_feat.shape[2])
    out = torch.empty(*pts.shape[:2], V * Cv)
    for v in range(V):
        ndc = get_ndc_coordinate(pose_ref["w2cs"][v], pose_ref["intrinsics"][v], pts, inv_scale)[None]
        grid = ndc[..., :2] * 2.0 - 1.0                                      # :317
        data = F.grid_sample(imgs[:, v], grid, align_corners=True, mode="bilinear", padding_mode="border")
        if img_feat is not None:
            data = torch.cat((data, F.grid_sample(img_feat[:, v], grid, align_corners=True, mode="bilinear", padding_mode="zeros")), 1)
        if with_mask:
            m = ((grid > -1.0) & (grid < 1.0)).all(-1).to(_dt())
            data = torch.cat((data, m.unsqueeze(1)), 1)
        out[..., v * Cv:(v + 1) * Cv] = data[0].permute(1, 2, 0)            # :329
    return out


def gen_pts_feats(imgs, volume, rays_pts, pose_ref, rays_ndc):
    """renderer.py:124-136 (use_color_volume=False, img_feat=None): [vol 8 | V x (r,g,b,mask)]."""
    return torch.cat([index_point_feature(volume, rays_ndc),
                      build_color_volume(rays_pts, pose_ref, imgs, with_mask=True)], -1)


def embed(x, num_freqs=10):
    """Embedder.embed models.py:47-51 with get_embedder(multires=10) :53-68.
    Layout [x | sin(x f0), sin(x f1).. (xyz fastest) | cos(...)]  (NOT per-frequency sin/cos interleave)."""
    freqs = 2.0 ** torch.linspace(0.0, num_freqs - 1, num_freqs)             # :34
    xs = (x.unsqueeze(-2) * freqs.view(*[1] * (x.dim() - 1), -1, 1)).reshape(*x.shape[:-1], -1)
    return torch.cat((x, torch.sin(xs), torch.cos(xs)), -1)


def renderer_ours(x, sd, prefix="nerf.", in_ch_pts=63, in_ch_views=3, skips=(4,), alpha_only=False):
    """Renderer_ours.forward models.py:194-222 / forward_alpha :176-191.
    x = [pts_embed(63) | feat(20) | dir(3)]; h = relu(W_i h * pts_bias(feat)) (multiplicative, :202)."""
    n_feat = x.shape[-1] - in_ch_pts - (0 if alpha_only else in_ch_views)
    pts, feat = x[..., :in_ch_pts], x[..., in_ch_pts:in_ch_pts + n_feat]
    lin = lambda name, h: F.linear(h, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"])
    bias = lin("pts_bias", feat)                                             # :200
    h = pts
    n_layers = sum(1 for k in sd if k.startswith(prefix + "pts_linears.") and k.endswith(".weight"))
    for i in range(n_layers):
        h = F.relu(lin(f"pts_linears.{i}", h) * bias)                        # :202-203
        if i in skips:
            h = torch.cat([pts, h], -1)                                      # :204-205
    alpha = torch.relu(lin("alpha_linear", h))                               # :209
    if alpha_only:
        return alpha
    h = torch.cat([lin("feature_linear", h), x[..., -in_ch_views:]], -1)     # :210-211
    h = F.relu(lin("views_linears.0", h))                                    # :213-215
    rgb = torch.sigmoid(lin("rgb_linear", h))                                # :217
    return torch.cat([rgb, alpha], -1)


def run_network_mvs(pts_ndc, viewdirs, feat, sd, netchunk=1024):
    """renderer.py:42-63 with embed_fn=Embedder(10), embeddirs_fn=None; batchify over rays (:28-40)."""
    x = torch.cat((embed(pts_ndc), feat), -1)
    if viewdirs is not None:
        x = torch.cat([x, viewdirs[:, None].expand(-1, x.shape[1], -1)], -1)
    outs = [renderer_ours(x[i:i + netchunk], sd, alpha_only=viewdirs is None) for i in range(0, x.shape[0], netchunk)]
    return torch.cat(outs, 0)


def raw2alpha(sigma):
    """renderer.py:18-26: alpha = 1-exp(-sigma) (dist ignored); T = exclusive cumprod(1-alpha+1e-10)."""
    alpha = 1.0 - torch.exp(-sigma)
    T = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    return alpha, alpha * T


def raw2outputs(raw, z_vals, white_bkgd=False):
    """renderer.py:65-92."""
    alpha, weights = raw2alpha(raw[..., 3])
    rgb_map = torch.sum(weights[..., None] * raw[..., :3], -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map, alpha


def rendering(pose_ref, rays_pts, rays_ndc, depth_candidates, rays_dir, volume, imgs, sd, white_bkgd=False):
    """renderer.py:138-165.  Returns the reference's 6-tuple (rgb_map, input_feat, weights, depth_map, alpha, {})
    plus raw (N,S,4) as a 7th element for sigma parity."""
    cos_angle = torch.norm(rays_dir, dim=-1)                                 # :142
    angle = gen_dir_feature(pose_ref["w2cs"][0], rays_dir / cos_angle.unsqueeze(-1))  # :147
    input_feat = gen_pts_feats(imgs, volume, rays_pts, pose_ref, rays_ndc)   # :152
    raw = run_network_mvs(rays_ndc, angle, input_feat, sd)                   # :156
    rgb_map, _, _, weights, depth_map, alpha = raw2outputs(raw, depth_candidates, white_bkgd)  # :162
    return rgb_map, input_feat, weights, depth_map, alpha, {}, raw


# --------------------------------------------------------------------------- importance sampling (fine-tuning option)

def get_ptsvolume(H, W, D, pad, near_far, intrinsic, c2w):
    """utils.py:338-355: world positions of the (D, H+2p, W+2p) voxel centres of the reference frustum volume."""
    near, far = near_far
    corners = torch.tensor([[-pad, -pad, 1.0], [W + pad, -pad, 1.0], [-pad, H + pad, 1.0], [W + pad, H + pad, 1.0]])
    corners = torch.matmul(corners, torch.inverse(intrinsic).t())                 # :343
    xs_l = torch.linspace(float(corners[0, 0]), float(corners[1, 0]), W + 2 * pad)
    ys_l = torch.linspace(float(corners[0, 1]), float(corners[2, 1]), H + 2 * pad)
    ys, xs = torch.meshgrid(ys_l, xs_l, indexing="ij")
    plane = torch.stack((xs, ys, torch.ones_like(xs)), dim=-1)
    near_plane, far_plane = plane * near, plane * far                             # :348-349
    lz = torch.linspace(1.0, 0.0, D).view(D, 1, 1, 1)
    pts = lz * near_plane + (1.0 - lz) * far_plane                                # :352
    pts = torch.matmul(pts.view(-1, 3), c2w[:3, :3].t()) + c2w[:3, 3].view(1, 3)
    return pts.view(D * (H + pad * 2), W + pad * 2, 3)


def render_density(pts, feat, sd, chunk=1024 * 5):
    """renderer.py:167-177: sigma-only MLP queries (forward_alpha); pts rows are embedded as they are (:172-174)."""
    return torch.cat([run_network_mvs(pts[i:i + chunk], None, feat[i:i + chunk], sd) for i in range(0, pts.shape[0], chunk)])


def ray_marcher(rays, N_samples=64, lindisp=False, perturb=0, perturb_rand=None):
    """data/ray_utils.py:152-197 (bbox_3D=None).  rays (N,8) = [o | d | near | far]; perturb_rand = the torch.rand draw of :187."""
    n = rays.shape[0]
    rays_o, rays_d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    steps = torch.linspace(0, 1, N_samples)
    z = near * (1 - steps) + far * steps if not lindisp else 1 / (1 / near * (1 - steps) + 1 / far * steps)
    z = z.expand(n, N_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper, lower = torch.cat([mid, z[:, -1:]], -1), torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * perturb_rand)
    return rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(2), rays_o, rays_d, z


def sample_pdf(bins, weights, u):
    """data/ray_utils.py:96-139 with the uniform draw `u` (N, N_importance) supplied (:105-109)."""
    weights = weights + 1e-5                                                       # :99
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)                    # (N, len(bins))
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)                                  # :126
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bins_b, bins_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)               # :135
    t = (u - cdf_b) / denom
    return bins_b + t * (bins_a - bins_b)


def ray_marcher_fine(rays, density_volume, z_vals, pts_NDC, u):
    """data/ray_utils.py:199-224.  Note the reference's double transform: pts_NDC*2-1 is handed to index_point_feature,
    which maps to [-1,1] once more (:209-210) - restated as is.  u: the torch.rand draw inside sample_pdf."""
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    sigma = index_point_feature(density_volume[None, None], pts_NDC * 2 - 1.0)[..., 0]
    alpha = 1.0 - torch.exp(-torch.relu(sigma))
    weights = alpha * torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    z_mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
    z_samples = sample_pdf(z_mid, weights[:, 1:-1], u)
    z = torch.sort(torch.cat([z_samples, z_vals], -1), -1)[0]
    return rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(2), rays_o, rays_d, z
