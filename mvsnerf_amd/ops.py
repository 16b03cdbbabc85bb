"""Python faces of the C-ABI entry points (include/mvsnerf_hip.h).  Tensors in, tensors out;
all arithmetic happens in libmvsnerf_hip.so.  No fallbacks."""
import ctypes

import torch

from . import _lib
from ._lib import check, dev_f32, stream_ptr


def _need_no_grad(*tensors, op):
    if torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors):
        raise NotImplementedError(
            f"{op}: the backward HIP kernel for this op is not built yet (round 1 ships the forward/"
            "inference path); call under torch.no_grad()")


# ------------------------------------------------------------------ volume layout
_cl_cache = {}


def channels_last_volume(volume_feature):
    """(1,C,D,H,W) reference-layout volume -> contiguous (D,H,W,C) tensor the kernels read.
    Zero-copy when the tensor is already channels_last_3d (what our MVSNet / RefVolume produce);
    otherwise one HIP transpose, cached on (storage, version)."""
    v = volume_feature
    if v.dim() == 5:
        if v.shape[0] != 1:
            raise RuntimeError("volume batch must be 1 (the reference assumes it too, models.py:916)")
        v = v[0]
    if v.dim() != 4:
        raise RuntimeError(f"volume must be (1,C,D,H,W) or (C,D,H,W), got {tuple(volume_feature.shape)}")
    cl = v.permute(1, 2, 3, 0)
    if cl.is_contiguous():
        dev_f32(cl, "volume")
        return cl
    # cache key: storage identity + version.  The entry keeps the source storage alive, so the caching
    # allocator cannot hand the same address to a different tensor while the entry exists.
    st = v.untyped_storage()
    key = (st.data_ptr(), v.storage_offset(), v._version, tuple(v.shape), tuple(v.stride()))
    hit = _cl_cache.get("k")
    if hit is not None and hit[0] == key:
        return hit[2]
    src = v.detach().contiguous()
    C, D, H, W = src.shape
    dst = torch.empty((D, H, W, C), device=src.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_ncdhw_to_ndhwc(dev_f32(src, "volume"), dst.data_ptr(), C, D, H, W, stream_ptr()), "ncdhw_to_ndhwc")
    _cl_cache["k"] = (key, st, dst)
    return dst


def ndhwc_to_ncdhw(vol_cl):
    D, H, W, C = vol_cl.shape
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
    check(_lib.lib().mvsnerf_volume_sample_fwd(dev_f32(vol_cl, "volume"), D, H, W, C, dev_f32(ndc, "ndc"), P,
                                               out.data_ptr(), out_stride, stream_ptr()), "volume_sample_fwd")
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
def raymarch(vol_cl, imgs, w2cs, intrinsics, packed, rays_pts, rays_ndc, z_vals, rays_dir, white_bkgd=False):
    """One FFI call for rendering() (renderer.py:138-165).  Returns dict of outputs."""
    _need_no_grad(vol_cl, imgs, rays_pts, rays_ndc, z_vals, rays_dir, op="raymarch")
    N, S = z_vals.shape
    V = imgs.shape[0]
    F = 8 + 4 * V
    dev = rays_pts.device
    D, H, W, C = vol_cl.shape
    if C != 8:
        raise RuntimeError("raymarch: the neural volume must have 8 channels")
    f32 = dict(device=dev, dtype=torch.float32)
    out = {
        "input_feat": torch.empty((N, S, F), **f32), "raw": torch.empty((N, S, 4), **f32),
        "rgb_map": torch.empty((N, 3), **f32), "disp": torch.empty((N,), **f32), "acc": torch.empty((N,), **f32),
        "weights": torch.empty((N, S), **f32), "depth": torch.empty((N,), **f32), "alpha": torch.empty((N, S), **f32),
    }
    dirs_tmp = torch.empty((N, 3), **f32)
    a = _lib.RaymarchArgs(
        dev_f32(vol_cl, "volume"), D, H, W, dev_f32(imgs, "imgs"), V, imgs.shape[2], imgs.shape[3],
        dev_f32(w2cs, "w2cs"), dev_f32(intrinsics, "intrinsics"), dev_f32(packed, "packed"),
        dev_f32(rays_pts, "rays_pts"), dev_f32(rays_ndc, "rays_ndc"), dev_f32(z_vals, "z_vals"), dev_f32(rays_dir, "rays_dir"),
        N, S, int(bool(white_bkgd)), dirs_tmp.data_ptr(), out["input_feat"].data_ptr(), out["raw"].data_ptr(),
        out["rgb_map"].data_ptr(), out["disp"].data_ptr(), out["acc"].data_ptr(), out["weights"].data_ptr(),
        out["depth"].data_ptr(), out["alpha"].data_ptr())
    check(_lib.lib().mvsnerf_raymarch_fwd(ctypes.byref(a), stream_ptr()), "raymarch_fwd")
    return out
