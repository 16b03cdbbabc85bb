"""Python faces of the C-ABI entry points (include/mvsnerf_hip.h).  Tensors in, tensors out;
all arithmetic happens in libmvsnerf_hip.so.  No fallbacks."""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import check, dev_f32, stream_ptr


def _need_no_grad(*tensors, op):
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        raise NotImplementedError(
            f"{op}: this stand-alone op has no backward kernel (gradients are provided for the fused paths: "
            "renderer.rendering / RefVolume -> ops.RayMarchFunction, MVSNet.build_volume_costvar* and CostRegNet); "
            "call it under torch.no_grad() or go through those entry points")


# ------------------------------------------------------------------ volume layout
_cl_cache = {}
FUSED_GATHER = True      # ops.raymarch: one gather launch (False: the three stand-alone lookups; same bits)


class _Keep:
    """Pointer helper for one C-ABI call: makes a tensor contiguous fp32 if needed and keeps that (possibly temporary) tensor
    alive until the call has been issued - a temporary's block may otherwise be handed to the next temporary by the caching
    allocator before the kernel is even enqueued."""

    def __init__(self):
        self.alive = []

    def __call__(self, t, name):
        t = t.to(torch.float32).contiguous()
        self.alive.append(t)
        return dev_f32(t, name)


VOL_DHWC, VOL_HWDC = 0, 1      # MVSNERF_VOL_* of include/mvsnerf_hip.h


def vol_ptr_layout(vol_cl, cur=None):
    """(device pointer, MVSNERF_VOL_* layout) of a (D,H,W,C)-shaped volume view as channels_last_volume returns it: contiguous memory is
    vol[d][y][x][c]; a view whose (H,W,D,C) permutation is contiguous is the depth-fastest vol[y][x][d][c] the encoder emits."""
    if vol_cl.is_contiguous():
        return dev_f32(vol_cl, "volume", cur), VOL_DHWC
    m = vol_cl.permute(1, 2, 0, 3)
    if m.is_contiguous():
        return dev_f32(m, "volume", cur), VOL_HWDC
    raise RuntimeError("volume: expected vol[d][y][x][c] or vol[y][x][d][c] memory (ops.channels_last_volume makes one of them)")


def channels_last_volume(volume_feature):
    """(1,C,D,H,W) reference-layout volume -> (D,H,W,C)-shaped tensor the kernels read: a zero-copy VIEW when the memory already is
    channel-last in one of the two orders of include/mvsnerf_hip.h - vol[d][y][x][c] (RefVolume, channels_last_3d tensors) or the
    depth-fastest vol[y][x][d][c] our MVSNet emits; otherwise one HIP transpose to vol[d][y][x][c], cached on (storage, version)."""
    v = volume_feature
    hit = _cl_cache.get("last")          # the same tensor object, unmodified, as in the previous call (a render loop): ~0.3 us instead of ~6
    if hit is not None and hit[0]() is v and hit[1] == (v._version, _lib.weights_epoch()):
        return hit[2]
    out = _channels_last_volume(v)
    # a WEAK reference to the tensor object: an encoder output still carries its grad_fn, and a module-level strong reference would keep the whole
    # encoder graph (its saved activations, GB-scale at config 2/3) alive until the next call.  `out` (a no-grad view or a transposed copy) keeps
    # only the volume's own storage; RayMarchFunction.backward drops the entry.
    # The DETACHED view is what is cached AND returned (a miss used to return the view with its grad_fn, a hit the detached one - whether a stand-alone
    # op saw a gradient-requiring volume depended on the cache).  The differentiable paths take the volume tensor itself (RayMarchFunction), never this view.
    out = out.detach()
    _cl_cache["last"] = (weakref.ref(v), (v._version, _lib.weights_epoch()), out)
    return out


def _channels_last_volume(volume_feature):
    v = volume_feature
    if v.dim() == 5:
        if v.shape[0] != 1:
            raise RuntimeError("volume batch must be 1 (the reference assumes it too, models.py:916)")
        v = v[0]
    if v.dim() != 4:
        raise RuntimeError(f"volume must be (1,C,D,H,W) or (C,D,H,W), got {tuple(volume_feature.shape)}")
    cl = v.permute(1, 2, 3, 0)
    if cl.is_contiguous() or cl.permute(1, 2, 0, 3).is_contiguous():
        vol_ptr_layout(cl)
        return cl
    # cache key: storage identity + version.  The entry keeps the source storage alive, so the caching
    # allocator cannot hand the same address to a different tensor while the entry exists.
    st = v.untyped_storage()
    key = (st.data_ptr(), v.storage_offset(), v._version, _lib.weights_epoch(), tuple(v.shape), tuple(v.stride()))
    hit = _cl_cache.get("k")
    if hit is not None and hit[0] == key:
        return hit[2]
    src = v.detach().contiguous()
    C, D, H, W = src.shape
    dst = torch.empty((D, H, W, C), device=src.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_ncdhw_to_ndhwc(dev_f32(src, "volume"), dst.data_ptr(), C, D, H, W, stream_ptr()), "ncdhw_to_ndhwc")
    _cl_cache["k"] = (key, st, dst)
    return dst


def channels_last_images(imgs):
    """(V,3,H,W) un-normalised source images -> (V,H,W,4) copy (rgb + pad) for the fused gather; one HIP transpose per
    scene, cached on (storage, version) like the volume."""
    if imgs.dim() != 4 or imgs.shape[1] != 3:
        raise RuntimeError(f"imgs must be (V,3,H,W), got {tuple(imgs.shape)}")
    st = imgs.untyped_storage()
    key = (st.data_ptr(), imgs.storage_offset(), imgs._version, tuple(imgs.shape), tuple(imgs.stride()))
    hit = _cl_cache.get("imgs")
    if hit is not None and hit[0] == key:
        return hit[2]
    src = imgs.detach().contiguous()
    V, _, H, W = src.shape
    dst = torch.empty((V, H, W, 4), device=src.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_nchw_to_nhwc(dev_f32(src, "imgs"), dst.data_ptr(), V, 3, H, W, 4, stream_ptr()), "nchw_to_nhwc")
    _cl_cache["imgs"] = (key, st, dst)
    return dst


def gather(vol_cl, imgs, w2cs, intrinsics, rays_pts, rays_ndc, rays_dir=None):
    """Fused gen_pts_feats (+ gen_dir_feature when rays_dir is given): -> (input_feat (N,S,8+4V), dirs (N,3) | None)."""
    _need_no_grad(vol_cl, imgs, rays_pts, rays_ndc, op="gather")
    N, S = rays_ndc.shape[:2]
    V = imgs.shape[0]
    F = 8 + 4 * V
    D, H, W, C = vol_cl.shape
    if C != 8:
        raise RuntimeError("gather: the neural volume must have 8 channels")
    feat = torch.empty((N, S, F), device=rays_ndc.device, dtype=torch.float32)
    dirs = None if rays_dir is None else torch.empty((N, 3), device=rays_ndc.device, dtype=torch.float32)
    icl = channels_last_images(imgs)
    vp, vl = vol_ptr_layout(vol_cl)
    check(_lib.lib().mvsnerf_gather_fwd(vp, D, H, W, icl.data_ptr(), V, imgs.shape[2], imgs.shape[3],
                                        dev_f32(w2cs, "w2cs"), dev_f32(intrinsics, "intrinsics"), dev_f32(rays_pts, "rays_pts"),
                                        dev_f32(rays_ndc, "rays_ndc"), N, S, 0 if rays_dir is None else dev_f32(rays_dir, "rays_dir"),
                                        feat.data_ptr(), F, 0 if dirs is None else dirs.data_ptr(), vl, stream_ptr()), "gather_fwd")
    return feat, dirs


def ndhwc_to_ncdhw(vol_cl):
    """vol[d][y][x][c] memory -> a contiguous (C,D,H,W) tensor (boundary helper for callers that want the reference's layout)."""
    D, H, W, C = vol_cl.shape
    if not vol_cl.is_contiguous():
        raise RuntimeError("ndhwc_to_ncdhw: vol[d][y][x][c] memory expected (a depth-fastest volume converts with tensor.contiguous())")
    dst = torch.empty((C, D, H, W), device=vol_cl.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_ndhwc_to_ncdhw(dev_f32(vol_cl, "volume"), dst.data_ptr(), C, D, H, W, stream_ptr()), "ndhwc_to_ncdhw")
    return dst


# ------------------------------------------------------------------ gathers
def volume_sample(vol_cl, ndc, out=None, out_stride=None):
    """vol_cl (D,H,W,C) ; ndc (...,3) -> (..., C) (or written into `out` rows of stride out_stride)."""
    _need_no_grad(vol_cl, ndc, op="volume_sample")
    D, H, W, C = vol_cl.shape
    P = ndc.numel() // 3
    if out is None:
        out = torch.empty((*ndc.shape[:-1], C), device=ndc.device, dtype=torch.float32)
        out_stride = C
    vp, vl = vol_ptr_layout(vol_cl)
    check(_lib.lib().mvsnerf_volume_sample_fwd(vp, D, H, W, C, dev_f32(ndc, "ndc"), P,
                                               out.data_ptr(), out_stride, vl, stream_ptr()), "volume_sample_fwd")
    return out


def color_sample(imgs, w2cs, intrinsics, pts, with_mask=True, out=None, out_ptr=None, out_stride=None):
    """imgs (V,3,H,W) ; w2cs (V,4,4) ; intrinsics (V,3,3) ; pts (...,3) -> (..., V*(3+mask))."""
    _need_no_grad(imgs, pts, op="color_sample")
    V, _, H, W = imgs.shape
    Cv = 3 + int(bool(with_mask))
    P = pts.numel() // 3
    if out is None:
        out = torch.empty((*pts.shape[:-1], V * Cv), device=pts.device, dtype=torch.float32)
        out_ptr, out_stride = out.data_ptr(), V * Cv
    check(_lib.lib().mvsnerf_color_sample_fwd(dev_f32(imgs, "imgs"), V, H, W, dev_f32(w2cs, "w2cs"), dev_f32(intrinsics, "intrinsics"),
                                              dev_f32(pts, "pts"), P, int(bool(with_mask)), out_ptr, out_stride, stream_ptr()),
          "color_sample_fwd")
    return out


def color_feat_sample(imgs, img_feat, w2cs, intrinsics, pts, with_mask=True, out=None, out_ptr=None, out_stride=None):
    """build_color_volume's img_feat branch: imgs (V,3,H,W), img_feat (V,Cf,Hf,Wf) -> (..., V*(3+Cf+mask))."""
    _need_no_grad(imgs, img_feat, pts, op="color_feat_sample")
    V, _, H, W = imgs.shape
    if img_feat.dim() != 4 or img_feat.shape[0] != V:
        raise RuntimeError(f"img_feat must be (V={V},Cf,Hf,Wf), got {tuple(img_feat.shape)}")
    Cf, Hf, Wf = img_feat.shape[1:]
    Cv = 3 + Cf + int(bool(with_mask))
    P = pts.numel() // 3
    if out is None:
        out = torch.empty((*pts.shape[:-1], V * Cv), device=pts.device, dtype=torch.float32)
        out_ptr, out_stride = out.data_ptr(), V * Cv
    check(_lib.lib().mvsnerf_color_feat_sample_fwd(dev_f32(imgs, "imgs"), V, H, W, dev_f32(img_feat, "img_feat"), Cf, Hf, Wf,
                                                   dev_f32(w2cs, "w2cs"), dev_f32(intrinsics, "intrinsics"), dev_f32(pts, "pts"), P,
                                                   int(bool(with_mask)), out_ptr, out_stride, stream_ptr()), "color_feat_sample_fwd")
    return out


def dir_feature(rays_dir, w2c_ref=None, normalize=True):
    _need_no_grad(rays_dir, op="dir_feature")
    out = torch.empty_like(rays_dir)
    check(_lib.lib().mvsnerf_dir_feature_fwd(dev_f32(rays_dir, "rays_dir"), 0 if w2c_ref is None else dev_f32(w2c_ref, "w2c_ref"),
                                             rays_dir.shape[0], int(bool(normalize)), out.data_ptr(), stream_ptr()), "dir_feature_fwd")
    return out


def posenc(x, num_freqs):
    """Embedder.embed (models.py:47-51) as a stand-alone op: (...,d) -> (..., d*(1+2L))."""
    _need_no_grad(x, op="posenc")
    d = x.shape[-1]
    out = torch.empty((*x.shape[:-1], d * (1 + 2 * num_freqs)), device=x.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_posenc_fwd(dev_f32(x, "x"), x.numel() // d, d, num_freqs, out.data_ptr(), stream_ptr()), "posenc_fwd")
    return out


def raygen(H, W, K_tgt, c2w_tgt, K_ref, w2c_ref, nf_tgt, nf_ref, N_samples, pad=0, lindisp=False,
           xs=None, ys=None, first_pixel=0, n_rays=None, t_rand=None, ref_hw=None):
    """Ray generation kernel (utils.build_rays / build_rays_test downstream of the RNG draws).
    Either xs/ys (float pixel ids, (N,)) or first_pixel + n_rays (row-major ids).  All camera tensors stay on the device.
    Returns rays_pts (N,S,3), rays_dir (N,3), rays_ndc (N,S,3), z_vals (N,S), pix (2,N)."""
    dev = c2w_tgt.device
    N = int(xs.shape[0]) if xs is not None else int(n_rays)
    f32 = dict(device=dev, dtype=torch.float32)
    pts = torch.empty((N, N_samples, 3), **f32)
    ndc = torch.empty((N, N_samples, 3), **f32)
    dirs = torch.empty((N, 3), **f32)
    z = torch.empty((N, N_samples), **f32)
    pix = torch.empty((2, N), **f32)
    c = _Keep()
    check(_lib.lib().mvsnerf_raygen_fwd(0 if xs is None else c(xs, "xs"), 0 if ys is None else c(ys, "ys"), int(first_pixel), W, H,
                                        0 if ref_hw is None else int(ref_hw[1]), 0 if ref_hw is None else int(ref_hw[0]),
                                        c(K_tgt, "K_tgt"), c(c2w_tgt, "c2w_tgt"), c(K_ref, "K_ref"), c(w2c_ref, "w2c_ref"),
                                        c(nf_tgt, "near_far_tgt"), c(nf_ref, "near_far_ref"), int(pad), int(bool(lindisp)),
                                        0 if t_rand is None else c(t_rand, "t_rand"), N, N_samples,
                                        pts.data_ptr(), dirs.data_ptr(), ndc.data_ptr(), z.data_ptr(), pix.data_ptr(), stream_ptr()), "raygen_fwd")
    return pts, dirs, ndc, z, pix


def raygen_train(H, W, K_tgt, c2w_tgt, K_ref, w2c_ref, nf_tgt, nf_ref, N_samples, xs, ys, t_rand, tgt_img, depth_map=None, z_map=None,
                 depth_mode=0, pad=0, lindisp=False):
    """build_rays of a training step in one launch (mvsnerf_raygen_train_fwd): ray generation + target-colour gather (+ ground-truth depth
    gather, + the per-pixel depth ranges of importanceSampling (depth_mode 1) / with_depth (depth_mode 2, one candidate per ray)).
    Returns rays_pts (N,S,3), rays_dir (N,3), rays_ndc (N,S,3), z_vals (N,S), pix (2,N), colors (N,3), rays_depth (N,) | None."""
    dev = c2w_tgt.device
    N = int(xs.shape[0])
    S = 1 if depth_mode == 2 else int(N_samples)
    f32 = dict(device=dev, dtype=torch.float32)
    pts, ndc, dirs = torch.empty((N, S, 3), **f32), torch.empty((N, S, 3), **f32), torch.empty((N, 3), **f32)
    z, pix, colors = torch.empty((N, S), **f32), torch.empty((2, N), **f32), torch.empty((N, 3), **f32)
    rd = None if depth_map is None else torch.empty((N,), **f32)
    c = _Keep()
    if tuple(tgt_img.shape) != (3, H, W) or (depth_map is not None and tuple(depth_map.shape) != (H, W)) or (z_map is not None and tuple(z_map.shape) != (H, W)):
        raise RuntimeError("raygen_train: tgt_img must be (3,H,W), depth_map / z_map (H,W)")
    check(_lib.lib().mvsnerf_raygen_train_fwd(c(xs, "xs"), c(ys, "ys"), W, H, 0, 0, c(K_tgt, "K_tgt"), c(c2w_tgt, "c2w_tgt"), c(K_ref, "K_ref"),
                                              c(w2c_ref, "w2c_ref"), c(nf_tgt, "near_far_tgt"), c(nf_ref, "near_far_ref"), int(pad), int(bool(lindisp)),
                                              0 if t_rand is None else c(t_rand, "t_rand"), N, S, c(tgt_img, "tgt_img"),
                                              0 if depth_map is None else c(depth_map, "depth_map"), 0 if z_map is None else c(z_map, "z_map"),
                                              int(depth_mode), pts.data_ptr(), dirs.data_ptr(), ndc.data_ptr(), z.data_ptr(), pix.data_ptr(),
                                              colors.data_ptr(), 0 if rd is None else rd.data_ptr(), stream_ptr()), "raygen_train_fwd")
    return pts, dirs, ndc, z, pix, colors, rd


def ray_points(rays_o, rays_d, z_vals, w2c_ref=None, K_ref=None, near_far_ref=None, ref_hw=None, pad=0, lindisp=False):
    """pts = o + d*z (N,S,3) and, when the reference camera is given, their NDC coordinates (get_ndc_coordinate)."""
    _need_no_grad(rays_o, rays_d, z_vals, op="ray_points")
    N, S = z_vals.shape
    dev = z_vals.device
    c = _Keep()
    per_ray = int(rays_o.dim() == 2 and rays_o.shape[0] == N and N > 1)
    pts = torch.empty((N, S, 3), device=dev, dtype=torch.float32)
    ndc = None if w2c_ref is None else torch.empty((N, S, 3), device=dev, dtype=torch.float32)
    check(_lib.lib().mvsnerf_ray_points_fwd(c(rays_o.reshape(-1, 3), "rays_o"), per_ray, c(rays_d, "rays_d"), c(z_vals, "z_vals"),
                                            0 if ndc is None else c(w2c_ref, "w2c_ref"), 0 if ndc is None else c(K_ref, "K_ref"),
                                            0 if ndc is None else c(near_far_ref.reshape(-1)[:2], "near_far_ref"),
                                            0 if ndc is None else int(ref_hw[1]), 0 if ndc is None else int(ref_hw[0]), int(pad), int(bool(lindisp)),
                                            N, S, pts.data_ptr(), 0 if ndc is None else ndc.data_ptr(), stream_ptr()), "ray_points_fwd")
    return pts, ndc


def sample_pdf(bins, weights, u):
    """data/ray_utils.py:96-139 with the uniform draws supplied: bins (N,nb), weights (N,nb-1), u (N,NI) -> (N,NI)."""
    _need_no_grad(bins, weights, u, op="sample_pdf")
    N, nb = bins.shape
    out = torch.empty(tuple(u.shape), device=u.device, dtype=torch.float32)
    c = _Keep()
    check(_lib.lib().mvsnerf_sample_pdf_fwd(c(bins, "bins"), c(weights, "weights"), c(u, "u"),
                                            N, nb, u.shape[1], out.data_ptr(), stream_ptr()), "sample_pdf_fwd")
    return out


def ray_marcher_fine_z(density_volume, rays_ndc, z_vals, u):
    """data/ray_utils.py:207-219: density (D,H,W), ndc (N,S,3), z (N,S), u (N,NI) -> sorted depths (N,S+NI)."""
    _need_no_grad(density_volume, rays_ndc, z_vals, u, op="ray_marcher_fine")
    N, S = z_vals.shape
    D, H, W = density_volume.shape
    out = torch.empty((N, S + u.shape[1]), device=z_vals.device, dtype=torch.float32)
    c = _Keep()
    check(_lib.lib().mvsnerf_ray_marcher_fine_fwd(c(density_volume, "density_volume"), D, H, W, c(rays_ndc, "rays_ndc"),
                                                  c(z_vals, "z_vals"), c(u, "u"), N, S, u.shape[1],
                                                  out.data_ptr(), stream_ptr()), "ray_marcher_fine_fwd")
    return out


# ------------------------------------------------------------------ MLP
MLP_ORDER = [f"pts_linears.{i}" for i in range(6)] + ["pts_bias", "feature_linear", "alpha_linear", "views_linears.0", "rgb_linear"]


def mlp_pack(weights, biases, F):
    """weights/biases: 11 contiguous fp32 GPU tensors in MLP_ORDER -> packed fragment-ordered buffer."""
    n = _lib.lib().mvsnerf_mlp_packed_floats(F)
    if n == 0:
        raise RuntimeError(f"mlp_pack: feat_dim {F} unsupported (must be even, <= 40)")
    expect = [(128, 63)] + [(128, 128)] * 4 + [(128, 191), (128, F), (128, 128), (1, 128), (64, 131), (3, 64)]
    for name, w, e in zip(MLP_ORDER, weights, expect):
        if tuple(w.shape) != e:
            raise RuntimeError(f"mlp_pack: {name}.weight has shape {tuple(w.shape)}, kernel is specialised for {e} "
                               "(netdepth 6, netwidth 128, skips [4], multires 10, raw 3-d view dirs)")
    packed = torch.empty(n, device=weights[0].device, dtype=torch.float32)
    wp = (ctypes.c_void_p * 11)(*[dev_f32(w, "weight") for w in weights])
    bp = (ctypes.c_void_p * 11)(*[dev_f32(b, "bias") for b in biases])
    check(_lib.lib().mvsnerf_mlp_pack(wp, bp, F, packed.data_ptr(), stream_ptr()), "mlp_pack")
    return packed


MLP_PRECISION = "auto"      # "auto" (default) | "fp32" | "bf16" | "bf16x3" | "bf16x6" | "fp16x3"; see set_mlp_precision
N_SPLIT = {"bf16x3": 2, "bf16x6": 3, "fp16x3": 18}      # n_split of mvsnerf_mlp_*_split; 18 = MVSNERF_SPLIT_FP16 (include/mvsnerf_hip.h)
_MODES = ("auto", "fp32", "bf16", "bf16x3", "bf16x6", "fp16x3")


def set_mlp_precision(mode):
    """Select the matrix-core arithmetic of the MLP:
      "auto"    (default) a step that needs gradients runs "fp32"; a no-grad rendering / network query runs the GUARDED fp16x3 sequence:
                the two-piece fp16 kernel (fp32-grade results, 2.5x the fp32-MFMA rate; operands and weights carry exact power-of-two
                scales, so fp16's exponent range is not a limit) reports what it cannot represent - a non-finite weight or value -
                through a device-side guard word, and the fp32-MFMA kernel enqueued right behind it - predicated on that word -
                recomputes the batch when it is set.  No host synchronisation (include/mvsnerf_hip.h, "guarded 16-bit sequences");
                ops.guard_fallbacks() tells how often the fp32 kernel had to step in.
      "fp32"    v_mfma_f32_32x32x2_f32 everywhere (the arithmetic of bench.py's headline and of the parity tests that pin the fp32 kernel)
      "bf16"    v_mfma_f32_32x32x16_bf16, operands rounded to bf16 (BASELINE configs 3/4; ~1e-2 errors); in training this is the
                reference's AMP switch (train_mvs_nerf_pl.py:317-318): forward, data- and weight-gradient GEMMs on bf16 operands
                with fp32 accumulation, fp32 master weights and fp32 gradients
      (the split modes below are inference-only)
      "bf16x6"  split-bf16 fp32 emulation: operands as 3 bf16 pieces, 6 bf16 MFMAs per product (fp32-grade results, fp32's range)
      "bf16x3"  2 bf16 pieces, 3 MFMAs (~1e-5 relative)
      "fp16x3"  the UNGUARDED two-piece fp16 kernel alone (csrc/mlp_f16x3.hip)"""
    global MLP_PRECISION
    if mode not in _MODES:
        raise ValueError(f"mlp precision must be one of {_MODES}")
    MLP_PRECISION = mode


def inference_mlp_mode():
    """What a no-grad query runs under the current MLP_PRECISION ("guarded" for "auto")."""
    return "guarded" if MLP_PRECISION == "auto" else MLP_PRECISION


def training_mlp_mode():
    return "fp32" if MLP_PRECISION == "auto" else MLP_PRECISION


class mlp_precision:
    """`with ops.mlp_precision("bf16"): ...` - set_mlp_precision for the duration of a block (the previous mode is restored)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = MLP_PRECISION
        set_mlp_precision(self.mode)

    def __exit__(self, *exc):
        set_mlp_precision(self.prev)


_guards = {}


def guard_words(device=None):
    """The guard words of the guarded 16-bit sequences enqueued on (this device, the CURRENT stream): int32[4] = {tripped (re-armed by every
    sequence), sequences that fell back to the fp32 kernels so far, 0, 0}.  One buffer per (device, stream), allocated once and owned here (the
    library allocates nothing): a sequence arms, reads and re-arms word 0 in STREAM order only, so two streams (or two host threads on their own
    streams) must never share a buffer - stream B's consume kernel could re-arm the word between stream A's fp16 kernel tripping it and A's
    predicated fp32 kernel reading it (include/mvsnerf_hip.h, "guarded 16-bit sequences": one guard buffer per stream)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    g = _guards.get(key)
    if g is None:
        with torch.cuda.device(idx):
            g = _guards[key] = torch.zeros(4, device=torch.device("cuda", idx), dtype=torch.int32)
    return g


def guard_fallbacks(device=None):
    """Number of guarded sequences (ray-march batches, network queries, scene encodes) on this device - all streams - whose fp32 kernels had to
    take over because a value left fp16's range.  Reading it synchronises - it is for tests and reports, the hot path never looks at it."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    torch.cuda.synchronize(idx)
    return sum(int(g[1].item()) for (d, _), g in list(_guards.items()) if d == idx)


def mlp_pack_split(weights, F, n_split):
    n = _lib.lib().mvsnerf_mlp_packed_split_elems(F, n_split)
    if n == 0:
        raise RuntimeError(f"mlp_pack_split: feat_dim {F} / n_split {n_split} unsupported")
    packed = torch.empty(n, device=weights[0].device, dtype=torch.bfloat16)
    wp = (ctypes.c_void_p * 11)(*[dev_f32(w, "weight") for w in weights])
    check(_lib.lib().mvsnerf_mlp_pack_split(wp, F, n_split, packed.data_ptr(), stream_ptr()), "mlp_pack_split")
    return packed


def mlp_forward_split(packed_split, n_split, packed, F, ndc_ptr, ndc_stride, feat_ptr, feat_stride, dirs_ptr, dirs_stride, N, S, alpha_only, device):
    raw = torch.empty((N * S, 1 if alpha_only else 4), device=device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_mlp_fwd_split(packed_split.data_ptr(), packed.data_ptr(), F, n_split, ndc_ptr, ndc_stride, feat_ptr, feat_stride,
                                           dirs_ptr, dirs_stride, N, S, int(alpha_only), raw.data_ptr(), stream_ptr()), "mlp_fwd_split")
    return raw


def mlp_forward_guarded(packed_h, packed, F, ndc_ptr, ndc_stride, feat_ptr, feat_stride, dirs_ptr, dirs_stride, N, S, alpha_only, device):
    raw = torch.empty((N * S, 1 if alpha_only else 4), device=device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_mlp_fwd_guarded(packed_h.data_ptr(), packed.data_ptr(), F, ndc_ptr, ndc_stride, feat_ptr, feat_stride,
                                             dirs_ptr, dirs_stride, N, S, int(alpha_only), raw.data_ptr(), guard_words(device).data_ptr(), stream_ptr()),
          "mlp_fwd_guarded")
    return raw


def mlp_pack_bf16(weights, F):
    n = _lib.lib().mvsnerf_mlp_packed_bf16_elems(F)
    packed = torch.empty(n, device=weights[0].device, dtype=torch.bfloat16)
    wp = (ctypes.c_void_p * 11)(*[dev_f32(w, "weight") for w in weights])
    check(_lib.lib().mvsnerf_mlp_pack_bf16(wp, F, packed.data_ptr(), stream_ptr()), "mlp_pack_bf16")
    return packed


def mlp_forward_bf16(packed_bf16, packed, F, ndc_ptr, ndc_stride, feat_ptr, feat_stride, dirs_ptr, dirs_stride, N, S, alpha_only, device):
    raw = torch.empty((N * S, 1 if alpha_only else 4), device=device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_mlp_fwd_bf16(packed_bf16.data_ptr(), packed.data_ptr(), F, ndc_ptr, ndc_stride, feat_ptr, feat_stride,
                                          dirs_ptr, dirs_stride, N, S, int(alpha_only), raw.data_ptr(), stream_ptr()), "mlp_fwd_bf16")
    return raw


def mlp_forward(packed, F, ndc_ptr, ndc_stride, feat_ptr, feat_stride, dirs_ptr, dirs_stride, N, S, alpha_only, device):
    raw = torch.empty((N * S, 1 if alpha_only else 4), device=device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_mlp_fwd(packed.data_ptr(), F, ndc_ptr, ndc_stride, feat_ptr, feat_stride, dirs_ptr, dirs_stride,
                                     N, S, int(alpha_only), raw.data_ptr(), stream_ptr()), "mlp_fwd")
    return raw


# ------------------------------------------------------------------ compositing
def composite(raw, z_vals, white_bkgd=False):
    """raw (N,S,4), z (N,S) -> rgb_map, disp, acc, weights, depth, alpha  (renderer.py:65-92)."""
    _need_no_grad(raw, z_vals, op="composite")
    N, S = z_vals.shape
    dev = raw.device
    rgb = torch.empty((N, 3), device=dev, dtype=torch.float32)
    disp = torch.empty((N,), device=dev, dtype=torch.float32)
    acc = torch.empty((N,), device=dev, dtype=torch.float32)
    depth = torch.empty((N,), device=dev, dtype=torch.float32)
    weights = torch.empty((N, S), device=dev, dtype=torch.float32)
    alpha = torch.empty((N, S), device=dev, dtype=torch.float32)
    check(_lib.lib().mvsnerf_composite_fwd(dev_f32(raw, "raw"), dev_f32(z_vals, "z_vals"), N, S, int(bool(white_bkgd)),
                                           rgb.data_ptr(), disp.data_ptr(), acc.data_ptr(), weights.data_ptr(),
                                           depth.data_ptr(), alpha.data_ptr(), stream_ptr()), "composite_fwd")
    return rgb, disp, acc, weights, depth, alpha


# ------------------------------------------------------------------ fused ray march
def _raymarch_block(vol_cl, imgs, w2cs, intrinsics, packed, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd, packed_bf16, packed_split, guard, want, imgs_cl, cur):
    """Output tensors + the filled mvsnerf_raymarch_args of one batch."""
    N, S = z_vals.shape
    V = imgs.shape[0]
    F = 8 + 4 * V
    dev = rays_pts.device
    D, H, W, C = vol_cl.shape
    if C != 8:
        raise RuntimeError("raymarch: the neural volume must have 8 channels")
    empty = torch.empty
    out = {
        "input_feat": empty((N, S, F), device=dev, dtype=torch.float32), "raw": empty((N, S, 4), device=dev, dtype=torch.float32),
        "rgb_map": empty((N, 3), device=dev, dtype=torch.float32), "weights": empty((N, S), device=dev, dtype=torch.float32),
        "depth": empty((N,), device=dev, dtype=torch.float32), "alpha": empty((N, S), device=dev, dtype=torch.float32),
    }
    for k in want:
        out[k] = empty((N,), device=dev, dtype=torch.float32)
    out["_dirs_tmp"] = dirs_tmp = empty((N, 3), device=dev, dtype=torch.float32)
    if imgs_cl is None and FUSED_GATHER:
        imgs_cl = channels_last_images(imgs)
    vp, vl = vol_ptr_layout(vol_cl, cur)
    a = (
        vp, D, H, W, dev_f32(imgs, "imgs", cur), V, imgs.shape[2], imgs.shape[3],
        dev_f32(w2cs, "w2cs", cur), dev_f32(intrinsics, "intrinsics", cur), dev_f32(packed, "packed", cur),
        dev_f32(rays_pts, "rays_pts", cur), dev_f32(rays_ndc, "rays_ndc", cur), dev_f32(z_vals, "z_vals", cur), dev_f32(rays_dir, "rays_dir", cur),
        N, S, int(bool(white_bkgd)), dirs_tmp.data_ptr(), out["input_feat"].data_ptr(), out["raw"].data_ptr(),
        out["rgb_map"].data_ptr(), out["disp"].data_ptr() if "disp" in out else 0, out["acc"].data_ptr() if "acc" in out else 0, out["weights"].data_ptr(),
        out["depth"].data_ptr(), out["alpha"].data_ptr(), 0 if packed_bf16 is None else packed_bf16.data_ptr(),
        imgs_cl.data_ptr() if (FUSED_GATHER and imgs_cl is not None) else 0,
        0 if packed_split is None else packed_split[0].data_ptr(), 0 if packed_split is None else int(packed_split[1]),
        0 if guard is None else guard.data_ptr(), vl)
    return out, a


def raymarch(vol_cl, imgs, w2cs, intrinsics, packed, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd=False, packed_bf16=None, packed_split=None,
             guard=None, want=("disp", "acc"), imgs_cl=None):
    """One FFI call for rendering() (renderer.py:138-165).  Returns dict of outputs.
    want: which of the optional per-ray maps `disp` / `acc` to produce (rendering() returns neither: it passes ()).
    imgs_cl: the channel-last copy of `imgs` when the caller already holds it (renderer's per-scene cache)."""
    _need_no_grad(vol_cl, imgs, rays_pts, rays_ndc, z_vals, rays_dir, op="raymarch")
    out, a = _raymarch_block(vol_cl, imgs, w2cs, intrinsics, packed, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd, packed_bf16, packed_split,
                             guard, want, imgs_cl, torch.cuda.current_device())
    blk = _lib.RaymarchArgs(*a)
    check(_lib.lib().mvsnerf_raymarch_fwd(ctypes.byref(blk), stream_ptr()), "raymarch_fwd")
    return out


def raymarch_batched(vol_cl, imgs, w2cs, intrinsics, packed, ray_batches, white_bkgd=False, packed_bf16=None, packed_split=None, guard=None, want=(),
                     imgs_cl=None):
    """K ray batches of one scene in ONE FFI call (mvsnerf_raymarch_fwd_batched): ray_batches = [(rays_pts, rays_ndc, z_vals, rays_dir), ...].
    Returns the list of per-batch output dicts of raymarch().  One batch is ~0.1 ms of GPU work; issued one call at a time, a render loop is
    paced by the host."""
    cur = torch.cuda.current_device()
    outs, blocks = [], (_lib.RaymarchArgs * len(ray_batches))()
    for k, (pts, ndc, z, rdir) in enumerate(ray_batches):
        _need_no_grad(vol_cl, imgs, pts, ndc, z, rdir, op="raymarch_batched")
        out, a = _raymarch_block(vol_cl, imgs, w2cs, intrinsics, packed, pts, ndc, z, rdir, white_bkgd, packed_bf16, packed_split, guard, want, imgs_cl, cur)
        blocks[k] = _lib.RaymarchArgs(*a)
        outs.append(out)
    check(_lib.lib().mvsnerf_raymarch_fwd_batched(blocks, len(ray_batches), stream_ptr()), "raymarch_fwd_batched")
    return outs


def render_pixels(vol_cl, imgs, w2cs, intrinsics, packed, H, W, K_tgt, c2w_tgt, K_ref, w2c_ref, nf_tgt, nf_ref, N_samples,
                  first_pixel=0, n_pixels=None, pad=0, lindisp=False, white_bkgd=False, packed_bf16=None, batch_rays=4096,
                  want=("depth",), ref_hw=None, packed_split=None, guard=None):
    """Pixel range of one target view in ONE FFI call (the chunk loop of validation_step, train_mvs_nerf_pl.py:198-208).
    Returns dict with rgb (n,3) and the requested extras among depth/acc/disp (n,)."""
    _need_no_grad(vol_cl, imgs, op="render_pixels")
    lib = _lib.lib()
    n = H * W - first_pixel if n_pixels is None else int(n_pixels)
    V = imgs.shape[0]
    D, Hv, Wv, C = vol_cl.shape
    if C != 8:
        raise RuntimeError("render_pixels: the neural volume must have 8 channels")
    dev = vol_cl.device
    f32 = dict(device=dev, dtype=torch.float32)
    B = int(min(batch_rays, max(n, 1)))
    ws_n = lib.mvsnerf_render_workspace_floats(B, N_samples, V)
    # one workspace per (size, device, STREAM): two streams rendering on one device must not share intermediates (same reason as guard_words)
    key = (ws_n, dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _render_ws.get(key)
    if ws is None:
        while len(_render_ws) >= 4:
            _render_ws.pop(next(iter(_render_ws)))
        ws = _render_ws[key] = torch.empty(ws_n, **f32)
    out = {"rgb": torch.empty((n, 3), **f32)}
    for k in ("depth", "acc", "disp"):
        out[k] = torch.empty((n,), **f32) if k in want else None
    if n == 0:
        return {k: v for k, v in out.items() if v is not None}
    c = _Keep()
    vp, vl = vol_ptr_layout(vol_cl)
    a = _lib.RenderArgs(
        vp, D, Hv, Wv, channels_last_images(imgs).data_ptr(), V, imgs.shape[2], imgs.shape[3],
        c(w2cs, "w2cs"), c(intrinsics, "intrinsics"), packed.data_ptr(), 0 if packed_bf16 is None else packed_bf16.data_ptr(),
        c(K_tgt, "K_tgt"), c(c2w_tgt, "c2w_tgt"), c(K_ref, "K_ref"), c(w2c_ref, "w2c_ref"), c(nf_tgt, "near_far_tgt"), c(nf_ref, "near_far_ref"),
        W, H, int(pad), int(bool(lindisp)), 0 if ref_hw is None else int(ref_hw[1]), 0 if ref_hw is None else int(ref_hw[0]), int(first_pixel), n, int(N_samples), int(bool(white_bkgd)), B,
        ws.data_ptr(), ws_n, out["rgb"].data_ptr(), *[0 if out[k] is None else out[k].data_ptr() for k in ("depth", "acc", "disp")],
        0 if packed_split is None else packed_split[0].data_ptr(), 0 if packed_split is None else int(packed_split[1]),
        0 if guard is None else guard.data_ptr(), vl)
    check(lib.mvsnerf_render_pixels_fwd(ctypes.byref(a), stream_ptr()), "render_pixels_fwd")
    return {k: v for k, v in out.items() if v is not None}


_render_ws = {}


# ------------------------------------------------------------------ training path (autograd)
def _act_n(q, h):
    return (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h


_maps_cache = {}


def _mlp_bwd_maps(F, device):
    """int32 table: fragment row -> nn.Linear row/column (layout documented at mvsnerf_mlp_bwd in csrc/mlp_bwd.hip)."""
    key = (F, str(device))
    if key in _maps_cache:
        return _maps_cache[key]
    import numpy as np
    t = -np.ones(1312, dtype=np.int32)
    act128 = np.array([_act_n(r >> 1, r & 1) for r in range(128)], dtype=np.int32)
    act64 = act128[:64].copy()
    pe = np.array([(r & 1) if (r >> 1) == 0 else ((2 if (r & 1) == 0 else -1) if (r >> 1) == 1 else 3 + ((r >> 1) - 2) + 30 * (r & 1))
                   for r in range(64)], dtype=np.int32)
    feat = np.array([((r & 1) * (F // 2) + (r >> 1)) if (r >> 1) < F // 2 else -1 for r in range(32)], dtype=np.int32)
    t[0:128], t[128:192], t[192:256], t[320:352] = act128, act64, pe, feat
    t[512:515] = [0, 1, 2]
    t[547] = 0
    t[576:640], t[640:768] = pe, 63 + act128
    t[768:896] = act128
    t[896:899] = [128, 129, 130]
    t[928:992] = act64
    t[1184:1312] = act128
    out = torch.from_numpy(t).to(device)
    _maps_cache[key] = out
    return out


def mlp_pack_bwd_bf16(weights, F):
    n = _lib.lib().mvsnerf_mlp_packed_bwd_bf16_elems()
    packed = torch.empty(n, device=weights[0].device, dtype=torch.bfloat16)
    wp = (ctypes.c_void_p * 11)(*[dev_f32(w, "weight") for w in weights])
    check(_lib.lib().mvsnerf_mlp_pack_bwd_bf16(wp, F, packed.data_ptr(), stream_ptr()), "mlp_pack_bwd_bf16")
    return packed


def mlp_pack_bwd(weights, F):
    n = _lib.lib().mvsnerf_mlp_packed_bwd_floats()
    packed = torch.empty(n, device=weights[0].device, dtype=torch.float32)
    wp = (ctypes.c_void_p * 11)(*[dev_f32(w, "weight") for w in weights])
    check(_lib.lib().mvsnerf_mlp_pack_bwd(wp, F, packed.data_ptr(), stream_ptr()), "mlp_pack_bwd")
    return packed


class RayMarchFunction(torch.autograd.Function):
    """rendering() (renderer.py:138-165) with gradients to the neural volume and the 22 MLP tensors.
    Inputs that never carry gradients in the reference's losses (rays, source images, cameras) are not differentiated."""

    @staticmethod
    def forward(ctx, volume, imgs, w2cs, intrinsics, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd, packed, dp_samples, *mlp_params):
        """ONE FFI call (mvsnerf_raymarch_train_fwd): lookups -> MLP training forward with activation store -> compositing."""
        lib = _lib.lib()
        vol_cl = channels_last_volume(volume)
        D, H, W, C = vol_cl.shape
        N, S = z_vals.shape
        V = imgs.shape[0]
        F = 8 + 4 * V
        if C not in (8, F):
            raise RuntimeError(f"ray march: the volume has {C} channels; expected 8 or 8 + 4V = {F}")
        if training_mlp_mode() not in ("fp32", "bf16"):
            raise RuntimeError(f"training runs the MLP in 'fp32' or 'bf16' (ops.set_mlp_precision), not {MLP_PRECISION!r}")
        bf16 = training_mlp_mode() == "bf16"
        dev = rays_pts.device
        f32 = dict(device=dev, dtype=torch.float32)
        feat = torch.empty((N, S, F), **f32)
        raw = torch.empty((N, S, 4), **f32)
        saved = torch.empty(lib.mvsnerf_mlp_saved_floats(N * S) // (2 if bf16 else 1), **f32)      # bf16 mode: two-byte slots
        dirs = torch.empty((N, 3), **f32)
        rgb, disp, acc, depth = (torch.empty(sh, **f32) for sh in ((N, 3), (N,), (N,), (N,)))
        weights, alpha = torch.empty((N, S), **f32), torch.empty((N, S), **f32)
        packed_b = mlp_pack_bf16([p.detach() for p in mlp_params[0::2]], F) if bf16 else None
        icl = channels_last_images(imgs) if C == 8 else None
        vp, vl = vol_ptr_layout(vol_cl)
        a = _lib.RaymarchTrainArgs(
            vol=vp, vol_layout=vl, D=D, H=H, W=W, C=C,
            imgs_nhwc4=0 if icl is None else icl.data_ptr(), V=V, IH=imgs.shape[2], IW=imgs.shape[3],
            w2c=dev_f32(w2cs, "w2cs"), K=dev_f32(intrinsics, "intrinsics"), packed_mlp=packed.data_ptr(),
            packed_mlp_bf16=0 if packed_b is None else packed_b.data_ptr(), bf16=int(bf16),
            rays_pts=dev_f32(rays_pts, "rays_pts"), rays_ndc=dev_f32(rays_ndc, "rays_ndc"), z_vals=dev_f32(z_vals, "z_vals"),
            rays_dir=dev_f32(rays_dir, "rays_dir"), N=N, S=S, white_bkgd=int(bool(white_bkgd)),
            dirs_tmp=dirs.data_ptr(), input_feat=feat.data_ptr(), raw=raw.data_ptr(), saved=saved.data_ptr(),
            rgb_map=rgb.data_ptr(), disp=disp.data_ptr(), acc=acc.data_ptr(), weights=weights.data_ptr(), depth=depth.data_ptr(), alpha=alpha.data_ptr())
        check(lib.mvsnerf_raymarch_train_fwd(ctypes.byref(a), stream_ptr()), "raymarch_train_fwd")
        ctx.save_for_backward(rays_ndc, z_vals, raw, saved, packed, *mlp_params)
        ctx.meta = (tuple(volume.shape), (D, H, W, C), N, S, F, bool(white_bkgd), bf16, bool(dp_samples))
        ctx.mark_non_differentiable(feat, raw)
        return rgb, feat, weights, depth, alpha, raw

    @staticmethod
    def backward(ctx, g_rgb, g_feat, g_weights, g_depth, g_alpha, g_raw):
        """The weight re-pack for the transposed products + ONE FFI call (mvsnerf_raymarch_bwd): compositing backward -> MLP data and
        weight gradients -> trilinear scatter into the volume gradient."""
        lib = _lib.lib()
        _cl_cache.pop("last", None)              # do not keep the step's volume storage past its backward (channels_last_volume)
        rays_ndc, z_vals, raw, saved, packed, *mlp_params = ctx.saved_tensors
        vshape, (D, H, W, C), N, S, F, white, bf16, dp_samples = ctx.meta
        dev = raw.device
        f32 = dict(device=dev, dtype=torch.float32)
        grads_in = [None if g is None else g.contiguous() for g in (g_rgb, g_depth, g_weights, g_alpha)]       # kept alive until the launch
        ptr = lambda t: 0 if t is None else dev_f32(t, "grad")
        weights = [p.detach() for p in mlp_params[0::2]]
        packed_bwd = mlp_pack_bwd_bf16(weights, F) if bf16 else mlp_pack_bwd(weights, F)
        d_raw = torch.empty((N, S, 4), **f32)
        gslots = torch.empty(lib.mvsnerf_mlp_gradslot_floats(N * S) // (2 if bf16 else 1), **f32)
        ws = torch.empty(lib.mvsnerf_mlp_bwd_workspace_floats(), **f32)
        d_feat = torch.empty((N * S, C), **f32)         # C = 8: the volume features only; C = F: the colour volume is a parameter too
        # 22 gradient tensors of their own, zeroed by ONE multi-tensor launch.  (They used to be views of one zero-filled buffer: autograd's
        # AccumulateGrad clones a gradient that does not own its storage - 22 device copies per step in the kernel trace.)
        views = [torch.empty_like(p, memory_format=torch.contiguous_format) for p in mlp_params]
        torch._foreach_zero_(views)
        gws, gbs = views[0::2], views[1::2]
        gwp = (ctypes.c_void_p * 11)(*[g.data_ptr() for g in gws])
        gbp = (ctypes.c_void_p * 11)(*[g.data_ptr() for g in gbs])
        maps = _mlp_bwd_maps(F, dev)
        gvol_cl = torch.zeros((D, H, W, C), **f32) if ctx.needs_input_grad[0] else None
        from . import distributed as DD
        # data-parallel fine-tuning of the volume: see volume_grad_from_all_ranks below (the scatter then runs after the call)
        exchange = dp_samples and gvol_cl is not None and DD._collective_needed()
        # VOLUME_BWD_DETERMINISTIC: the scatter runs behind the call too, through the fixed-point accumulators (mvsnerf_volume_sample_bwd_det)
        gvol_arg = None if (exchange or VOLUME_BWD_DETERMINISTIC) else gvol_cl
        a = _lib.RaymarchBwdArgs(
            packed_mlp=packed.data_ptr(), packed_bwd=packed_bwd.data_ptr(), bf16=int(bf16), F=F,
            raw=raw.data_ptr(), saved=saved.data_ptr(), z_vals=z_vals.data_ptr(), rays_ndc=rays_ndc.data_ptr(), N=N, S=S, white_bkgd=int(white),
            g_rgb=ptr(grads_in[0]), g_depth=ptr(grads_in[1]), g_weights=ptr(grads_in[2]), g_alpha=ptr(grads_in[3]),
            d_raw=d_raw.data_ptr(), gslots=gslots.data_ptr(), d_feat=d_feat.data_ptr(), n_feat_out=C,
            gw=ctypes.cast(gwp, ctypes.POINTER(ctypes.c_void_p)), gb=ctypes.cast(gbp, ctypes.POINTER(ctypes.c_void_p)),
            maps=maps.data_ptr(), workspace=ws.data_ptr(),
            gvol=0 if gvol_arg is None else gvol_arg.data_ptr(), D=D, H=H, W=W, C=C)
        check(lib.mvsnerf_raymarch_bwd(ctypes.byref(a), stream_ptr()), "raymarch_bwd")
        if exchange:
            volume_grad_from_all_ranks(d_feat, rays_ndc.reshape(-1, 3), gvol_cl)
        elif VOLUME_BWD_DETERMINISTIC and gvol_cl is not None:
            _scatter_hip(gvol_cl, rays_ndc.reshape(-1, 3), d_feat)
        g_vol = None
        if gvol_cl is not None:
            g_vol = gvol_cl.permute(3, 0, 1, 2)
            if len(vshape) == 5:
                g_vol = g_vol.unsqueeze(0)
        param_grads = []
        for gw, gb in zip(gws, gbs):
            param_grads += [gw, gb]
        return (g_vol, None, None, None, None, None, None, None, None, None, None, *param_grads)


VOLUME_BWD_DETERMINISTIC = False   # True: the volume gradient's scatter through 64-bit fixed-point accumulators (integer atomics commute: two runs, and N ranks
                                   # against one, give bit-identical gradients; mvsnerf_volume_sample_bwd_det) - two extra passes and 8 bytes of zeroed workspace
                                   # per volume element (300 MB at config 2) instead of float atomics


def _scatter_hip(gvol_cl, ndc, g):
    D, H, W, C = gvol_cl.shape
    lib = _lib.lib()
    if VOLUME_BWD_DETERMINISTIC:
        ws = torch.zeros(lib.mvsnerf_volume_sample_bwd_det_workspace_words(D, H, W, C), device=gvol_cl.device, dtype=torch.int64)
        check(lib.mvsnerf_volume_sample_bwd_det(D, H, W, C, ndc.data_ptr(), ndc.shape[0], g.data_ptr(), g.shape[-1], gvol_cl.data_ptr(), ws.data_ptr(), stream_ptr()),
              "volume_sample_bwd_det")
        return
    check(lib.mvsnerf_volume_sample_bwd(D, H, W, C, ndc.data_ptr(), ndc.shape[0], g.data_ptr(), C, gvol_cl.data_ptr(), stream_ptr()), "volume_sample_bwd")


def volume_grad_from_all_ranks(d_feat, ndc, gvol_cl, group=None, scatter=_scatter_hip):
    """Data-parallel gradient of a learnable RefVolume WITHOUT reducing the volume-sized tensor (SURVEY.md 5 / 8e "Fine-tune").

    The volume gradient is the trilinear scatter of the per-sample feature gradients d_feat (P_local x C).  A dense all-reduce of it moves
    2 x (world-1)/world x 150-246 MB per rank and step over xGMI; the SAMPLE gradients of all ranks are 36 B + 12 B (NDC) per sample -
    6 MB for 1024 x 128 samples in total.  So: ONE all_gather of [d_feat | ndc] rows, then every rank scatters the samples of ALL ranks
    into its own gradient volume (0.16 ms for 131 072 samples) and scales by 1/world - the same mean-over-ranks FlatGradAllReduce gives
    the other parameters.  Every rank ends with the full gradient; the volume parameter takes no part in the flat all-reduce.
    (The scatter uses float atomics, so replicas can differ in the last bits; MVSSystemFinetune re-broadcasts the volume from rank 0 every
    `args.dp_volume_resync` steps.)"""
    import torch.distributed as dist
    from . import distributed as DD
    C = d_feat.shape[1]
    P = d_feat.shape[0]
    world = dist.get_world_size(group)
    # shards may differ by one ray: gather the counts, pad to the largest
    cnt = torch.tensor([P], device=d_feat.device, dtype=torch.int64)
    cnts = torch.empty(world, device=d_feat.device, dtype=torch.int64)
    DD.all_gather_into_tensor(cnts, cnt, group=group)
    cnts = [int(c) for c in cnts.tolist()]
    width = max(cnts)
    row = torch.zeros((width, C + 4), device=d_feat.device, dtype=torch.float32)          # [d_feat (C) | ndc (3) | pad] -> 16-byte rows
    row[:P, :C] = d_feat
    row[:P, C:C + 3] = ndc
    allrows = torch.empty((world * width, C + 4), device=d_feat.device, dtype=torch.float32)
    DD.all_gather_into_tensor(allrows, row, group=group)
    for r, n in enumerate(cnts):
        if n == 0:
            continue
        blk = allrows[r * width:r * width + n]
        scatter(gvol_cl, blk[:, C:C + 3].contiguous(), blk[:, :C].contiguous())     # `scatter` is replaceable so that the CPU (gloo) tests can drive this exchange
    gvol_cl.div_(world)
    return gvol_cl


def raymarch_train(volume, imgs, w2cs, intrinsics, net, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd=False, dp_samples=False):
    """Differentiable rendering(): `net` is a models.MVSNeRF; returns the dict of ops.raymarch.
    dp_samples: data-parallel volume gradient by exchanging sample gradients (volume_grad_from_all_ranks)."""
    V = imgs.shape[0]
    lins = net.nerf._linears()
    params = []
    for l in lins:
        params += [l.weight, l.bias]
    packed = net.packed(8 + 4 * V)
    rgb, feat, weights, depth, alpha, raw = RayMarchFunction.apply(
        volume, imgs, w2cs, intrinsics, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd, packed, dp_samples, *params)
    return {"rgb_map": rgb, "input_feat": feat, "weights": weights, "depth": depth, "alpha": alpha, "raw": raw}
