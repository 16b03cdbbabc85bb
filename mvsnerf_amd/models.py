"""Drop-in for the hot-path classes of the reference's models.py: same class names, constructor
arguments, sub-module / parameter names (so `ckpts/mvsnerf-v0.tar` loads unchanged) and call
signatures - with every forward running in libmvsnerf_hip.so.

Modules here are parameter *containers* (nn.Linear / nn.Conv3d objects are never called): forward passes
hand raw device pointers to the C ABI.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .renderer import run_network_mvs

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")   # reference models.py:657


def weights_init(m):
    """reference models.py:10-14."""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)


# ------------------------------------------------------------------ positional encoding
class Embedder:
    """reference models.py:17-51.  Callable; layout [x | sin(x 2^f) f-major | cos(...)].
    When (input_dims, num_freqs) == (3, 10) the MLP kernel embeds in registers and this object is only
    a tag (`fusable`); called on its own it runs the stand-alone posenc kernel."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.input_dims = kwargs["input_dims"]
        self.num_freqs = kwargs["num_freqs"]
        if not (kwargs.get("include_input", True) and kwargs.get("log_sampling", True)
                and kwargs["max_freq_log2"] == self.num_freqs - 1):
            raise NotImplementedError("Embedder: only include_input + log-sampled 2^0..2^(L-1) bands (get_embedder's setting)")
        self.out_dim = self.input_dims * (1 + 2 * self.num_freqs)
        self.fusable = (self.input_dims == 3 and self.num_freqs == 10)

    def embed(self, inputs):
        return ops.posenc(inputs.contiguous(), self.num_freqs)

    __call__ = embed


def get_embedder(multires, i=0, input_dims=3):
    """reference models.py:53-68 -> (embed_fn, out_dim)."""
    if i == -1:
        return nn.Identity(), 3
    e = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires,
                 log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return e, e.out_dim


# ------------------------------------------------------------------ radiance MLP
class Renderer_ours(nn.Module):
    """reference models.py:145-222 (net_type 'v0', the architecture of the shipped checkpoint).
    forward(x) / forward_alpha(x) take the reference's concatenated rows
        x = [embed(pts)(63) | feat(F) | dir(3)]      (forward_alpha: no dir)
    and run the fused HIP kernel on them in place: the kernel reads pts from x[..., :3] (the embedding's
    leading copy of the input, models.py:50) and re-derives the sin/cos terms in registers."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, input_ch_feat=8, skips=[4], use_viewdirs=False):
        super().__init__()
        if not use_viewdirs:
            raise NotImplementedError("Renderer_ours(use_viewdirs=False) cannot even be constructed in the reference (models.py:172)")
        self.D, self.W, self.skips = D, W, list(skips)
        self.input_ch, self.input_ch_views, self.use_viewdirs = input_ch, input_ch_views, use_viewdirs
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch, input_ch_views, input_ch_feat
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W + input_ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.pts_bias = nn.Linear(input_ch_feat, W)
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        for m in (self.pts_linears, self.views_linears, self.feature_linear, self.alpha_linear, self.rgb_linear):
            m.apply(weights_init)
        self._packed = None
        self._packed_key = None

    # -- weight staging ---------------------------------------------------------------------
    def _linears(self):
        return list(self.pts_linears) + [self.pts_bias, self.feature_linear, self.alpha_linear, self.views_linears[0], self.rgb_linear]

    def packed(self, feat_dim=None):
        """Fragment-ordered weight buffer for the MFMA kernel; re-packed (one tiny kernel) whenever a
        parameter was modified in place (optimizer step, load_state_dict) or moved."""
        F = self.in_ch_feat if feat_dim is None else feat_dim
        if F != self.in_ch_feat:
            raise RuntimeError(f"feat_dim {F} does not match the network's input_ch_feat {self.in_ch_feat}")
        if self.D != 6 or self.W != 128 or self.skips != [4] or self.in_ch_pts != 63 or self.in_ch_views != 3:
            raise NotImplementedError(
                "the HIP MLP kernel is specialised for netdepth=6, netwidth=128, skips=[4], multires=10, raw view dirs "
                f"(got D={self.D}, W={self.W}, skips={self.skips}, in_ch_pts={self.in_ch_pts}, in_ch_views={self.in_ch_views})")
        key = self._weights_key()
        if self._packed is None or key != self._packed_key:
            lins = self._linears()
            self._packed = ops.mlp_pack([l.weight.detach() for l in lins], [l.bias.detach() for l in lins], F)
            self._packed_key = key
        return self._packed

    def _weights_key(self):
        """(optimizer-step epoch, data_ptr and _version of the 22 tensors).  The Parameter objects are looked up through nn.Module.__getattr__
        once and kept (a module attribute access costs ~0.5 us, and this key is built on every rendering() call: 44 of them were a third
        of the step's host time); the kept list is validated by identity against the 11 layers' own `_parameters` dicts (22 `is` tests, ~1.5 us)
        and re-made when ANY parameter object was replaced (`lin.weight = nn.Parameter(...)`, load_state_dict(assign=True), pruning utilities)."""
        c = self.__dict__.get("_plist")
        ok = c is not None
        if ok:
            pl, lins = c
            i = 0
            for l in lins:
                pr = l._parameters
                if pr["weight"] is not pl[i] or pr["bias"] is not pl[i + 1]:
                    ok = False
                    break
                i += 2
        if not ok:
            lins = self._linears()
            c = ([t for l in lins for t in (l.weight, l.bias)], lins)
            self.__dict__["_plist"] = c
        pl = c[0]
        return (_lib.weights_epoch(), *[p._version for p in pl], *[p.data_ptr() for p in pl])

    def invalidate_packed(self):
        """Drop the packed-weight caches.  The caches key on (data_ptr, tensor._version, optimizer-step epoch - _lib.weights_epoch); writes through `.data` (`p.data.copy_()`,
        EMA updates on `.data`) do not bump `_version`, so code that updates weights that way must call this afterwards
        (MVSNet.invalidate_packed does the same for the encoder's convolution weights)."""
        self._packed = self._packed_key = None
        self._packed_b = self._packed_s = None
        self.__dict__["_plist"] = None

    def packed_bf16(self, feat_dim=None, fresh=False):
        """bf16 fragment-ordered weights for the opt-in bf16-MFMA kernel (same cache policy as packed()).
        fresh: packed() has just been called for these weights (same host call) - its key is current."""
        F = self.in_ch_feat if feat_dim is None else feat_dim
        if not fresh:
            self.packed(F)                  # validates the architecture and keeps the fp32 vectors current
        if getattr(self, "_packed_b", None) is None or self._packed_b_key != self._packed_key:
            self._packed_b = ops.mlp_pack_bf16([l.weight.detach() for l in self._linears()], F)
            self._packed_b_key = self._packed_key
        return self._packed_b

    def packed_split(self, feat_dim=None, n_split=3, fresh=False):
        """(split weight planes, n_split) for the bf16x3 / bf16x6 / fp16x3 kernels (same cache policy as packed(); fresh: see packed_bf16)."""
        F = self.in_ch_feat if feat_dim is None else feat_dim
        if not fresh:
            self.packed(F)
        cache = getattr(self, "_packed_s", None)
        if cache is None or cache[0] != (self._packed_key, n_split):
            self._packed_s = ((self._packed_key, n_split), ops.mlp_pack_split([l.weight.detach() for l in self._linears()], F, n_split))
        return self._packed_s[1], n_split

    def packed_alt(self, feat_dim=None, fresh=False):
        """Keyword arguments selecting the MLP kernel of ops.raymarch / ops.render_pixels for the current ops.MLP_PRECISION."""
        mode = ops.inference_mlp_mode()
        if mode == "bf16":
            return {"packed_bf16": self.packed_bf16(feat_dim, fresh)}
        if mode == "guarded":        # the default: fp16x3 kernel + predicated fp32-MFMA kernel behind it (ops.set_mlp_precision)
            return {"packed_split": self.packed_split(feat_dim, ops.N_SPLIT["fp16x3"], fresh), "guard": ops.guard_words(self._packed.device)}
        if mode in ops.N_SPLIT:
            return {"packed_split": self.packed_split(feat_dim, ops.N_SPLIT[mode], fresh)}
        return {}

    # -- queries ----------------------------------------------------------------------------
    def query(self, pts, feat, viewdirs, N, S):
        """pts (N,S,3) NDC, feat (N,S,F), viewdirs (N,3) per ray or None (sigma only) -> (N*S, 4|1)."""
        ops._need_no_grad(pts, feat, viewdirs, *self.parameters(), op="Renderer_ours")
        pts, feat = pts.contiguous(), feat.contiguous()
        alpha_only = viewdirs is None
        F = feat.shape[-1]
        vd = None if alpha_only else viewdirs.contiguous()        # named: a temporary would be freed before the launch
        dptr = 0 if alpha_only else ops.dev_f32(vd, "viewdirs")
        mode = ops.inference_mlp_mode()
        if mode == "bf16":
            return ops.mlp_forward_bf16(self.packed_bf16(F), self.packed(F), F, ops.dev_f32(pts, "pts"), 3, ops.dev_f32(feat, "feat"), F,
                                        dptr, 3, N, S, alpha_only, pts.device)
        if mode == "guarded":
            ps, _ = self.packed_split(F, ops.N_SPLIT["fp16x3"])
            return ops.mlp_forward_guarded(ps, self.packed(F), F, ops.dev_f32(pts, "pts"), 3, ops.dev_f32(feat, "feat"), F, dptr, 3, N, S, alpha_only, pts.device)
        if mode in ops.N_SPLIT:
            ps, ns = self.packed_split(F, ops.N_SPLIT[mode])
            return ops.mlp_forward_split(ps, ns, self.packed(F), F, ops.dev_f32(pts, "pts"), 3, ops.dev_f32(feat, "feat"), F,
                                         dptr, 3, N, S, alpha_only, pts.device)
        return ops.mlp_forward(self.packed(F), F, ops.dev_f32(pts, "pts"), 3, ops.dev_f32(feat, "feat"), F, dptr, 3, N, S, alpha_only, pts.device)

    def _rows(self, x, alpha_only):
        ops._need_no_grad(x, *self.parameters(), op="Renderer_ours")
        width = x.shape[-1]
        F = width - self.in_ch_pts - (0 if alpha_only else self.in_ch_views)
        x2 = x.reshape(-1, width).contiguous()
        base = ops.dev_f32(x2, "x")
        P = x2.shape[0]
        raw = ops.mlp_forward(self.packed(F), F, base, width, base + 4 * self.in_ch_pts, width,
                              0 if alpha_only else base + 4 * (self.in_ch_pts + F), width, P, 1, alpha_only, x.device)
        return raw.view(*x.shape[:-1], -1)

    def forward_alpha(self, x):
        return self._rows(x, True)

    def forward(self, x):
        return self._rows(x, False)


class MVSNeRF(nn.Module):
    """reference models.py:540-567: wrapper holding the renderer as `.nerf` (checkpoint keys `nerf.*`)."""

    def __init__(self, D=8, W=256, input_ch_pts=3, input_ch_views=3, input_ch_feat=8, skips=[4], net_type="v2"):
        super().__init__()
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch_pts, input_ch_views, input_ch_feat
        if net_type != "v0":
            raise NotImplementedError(
                f"net_type {net_type!r}: only 'v0' (Renderer_ours, the shipped checkpoint) is on the hot path; "
                "v1/v2 have no weights in the reference and v1 is dead code (SURVEY.md 2)")
        self.nerf = Renderer_ours(D=D, W=W, input_ch_feat=input_ch_feat, input_ch=input_ch_pts, output_ch=4, skips=skips,
                                  input_ch_views=input_ch_views, use_viewdirs=True)

    def packed(self, feat_dim=None):
        return self.nerf.packed(feat_dim)

    def packed_bf16(self, feat_dim=None):
        return self.nerf.packed_bf16(feat_dim)

    def packed_split(self, feat_dim=None, n_split=3):
        return self.nerf.packed_split(feat_dim, n_split)

    def packed_alt(self, feat_dim=None, fresh=False):
        return self.nerf.packed_alt(feat_dim, fresh)

    def invalidate_packed(self):
        self.nerf.invalidate_packed()

    def query(self, pts, feat, viewdirs, N, S):
        return self.nerf.query(pts, feat, viewdirs, N, S)

    def forward_alpha(self, x):
        return self.nerf.forward_alpha(x)

    def forward(self, x):
        return self.nerf(x)


# ------------------------------------------------------------------ learnable volume (fine-tuning)
class RefVolume(nn.Module):
    """reference models.py:935-950.  Parameter name `feat_volume` (checkpoint key `volume.feat_volume`),
    logical shape (1,C,D,h,w); stored channels-last in memory so lookups need no transpose."""

    def __init__(self, volume):
        super().__init__()
        self.feat_volume = nn.Parameter(volume.contiguous(memory_format=torch.channels_last_3d))

    def forward(self, ray_coordinate_ref):
        ops._need_no_grad(self.feat_volume, op="RefVolume.forward (volume_sample)")      # checked on the parameter itself: the view below is always detached
        vol_cl = ops.channels_last_volume(self.feat_volume)
        ndc = ray_coordinate_ref.to(vol_cl.device, torch.float32).contiguous()
        return ops.volume_sample(vol_cl, ndc).squeeze()


# ------------------------------------------------------------------ factory
def create_nerf_mvs(args, pts_embedder=True, use_mvs=False, dir_embedder=True):
    """reference models.py:569-654: same return tuple and dict keys; loads `network_fn_state_dict` /
    `network_mvs_state_dict` of a reference checkpoint unchanged."""
    if pts_embedder:
        embed_fn, input_ch = get_embedder(args.multires, args.i_embed, input_dims=args.pts_dim)
    else:
        embed_fn, input_ch = None, args.pts_dim
    if dir_embedder:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed, input_dims=args.dir_dim)
    else:
        embeddirs_fn, input_ch_views = None, args.dir_dim

    skips = [4]
    model = MVSNeRF(D=args.netdepth, W=args.netwidth, input_ch_pts=input_ch, skips=skips, input_ch_views=input_ch_views,
                    input_ch_feat=args.feat_dim, net_type=args.net_type).to(device)
    grad_vars = list(model.parameters())

    model_fine = None
    if args.N_importance > 0:
        model_fine = MVSNeRF(D=args.netdepth, W=args.netwidth, input_ch_pts=input_ch, skips=skips, input_ch_views=input_ch_views,
                             input_ch_feat=args.feat_dim, net_type=args.net_type).to(device)
        grad_vars += list(model_fine.parameters())

    def network_query_fn(pts, viewdirs, rays_feats, network_fn):
        return run_network_mvs(pts, viewdirs, rays_feats, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                               netchunk=args.netchunk)
    # tag read by renderer.rendering: this query fn may be replaced by the one-call fused ray march
    network_query_fn._mvsnerf_fused = isinstance(embed_fn, Embedder) and embed_fn.fusable and embeddirs_fn is None

    EncodingNet = None
    if use_mvs:
        from .encoder import MVSNet
        EncodingNet = MVSNet(n_views=getattr(args, "n_views", 3)).to(device)     # n_views: extension (config 4), default = reference
        grad_vars += list(EncodingNet.parameters())

    start = 0
    ckpt_path = getattr(args, "ckpt", None)
    if ckpt_path is not None and ckpt_path != "None":
        print("Reloading from", ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=device, weights_only=False)
        if use_mvs:
            EncodingNet.load_state_dict(ckpt["network_mvs_state_dict"])
        model.load_state_dict(ckpt["network_fn_state_dict"])

    render_kwargs_train = {
        "network_query_fn": network_query_fn, "perturb": args.perturb, "N_importance": args.N_importance,
        "network_fine": model_fine, "N_samples": args.N_samples, "network_fn": model, "network_mvs": EncodingNet,
        "use_viewdirs": args.use_viewdirs, "white_bkgd": args.white_bkgd, "raw_noise_std": args.raw_noise_std,
    }
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test["perturb"] = False
    return render_kwargs_train, render_kwargs_test, start, grad_vars


# the encoder half of the reference's models.py lives in encoder.py; re-exported here so that
# `from mvsnerf_amd.models import *` offers the same names as `from models import *`
from .encoder import InPlaceABN, ConvBnReLU, ConvBnReLU3D, FeatureNet, CostRegNet, MVSNet  # noqa: E402,F401
