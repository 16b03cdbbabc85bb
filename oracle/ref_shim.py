"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mvsnerf_amd/).

Imports the *real* reference (apchenstu/mvsnerf, mounted read-only at
/root/reference) on a CPU-only host so that golden vectors can be generated
from the reference's own code.  /root/reference exists only in the authoring
container; nothing under tests/ -m gpu, bench.py or smoke() may use this file.

The reference has module-level imports of packages that are not in this image
(cv2, torchvision, kornia, inplace_abn, warmup_scheduler) - SURVEY.md 8(c).
They are stubbed in sys.modules *before* `import models, renderer, utils`:

* kornia.utils.create_meshgrid(h, w, normalized_coordinates=False, device)
    -> (1,h,w,2), [...,0]=x pixel, [...,1]=y pixel       (used at utils.py:603)
* inplace_abn.InPlaceABN(C)  (used at models.py:6,668,681,742,747,752)
    third-party mapillary/inplace_abn, version unpinned by the reference.
    Published semantics restated: y = leaky_relu(batch_norm(x, gamma=|w|+eps), 0.01),
    eps=1e-5, momentum=0.1, buffers weight/bias/running_mean/running_var.
    PARITY UNPINNED for this third-party op (no reference test pins it).
* Tensor.cuda -> no-op on a CPU-only host (models.py:37 hard-codes .cuda()).
"""
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = os.environ.get("MVSNERF_REFERENCE", "/root/reference")


class InPlaceABN(nn.Module):
    """Restatement of mapillary InPlaceABN forward (activation='leaky_relu', slope 0.01)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True,
                 activation="leaky_relu", activation_param=0.01):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.activation, self.activation_param = activation, activation_param
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))  # ckpt has this key

    def forward(self, x):
        y = F.batch_norm(x, self.running_mean, self.running_var,
                         self.weight.abs() + self.eps, self.bias,
                         self.training, self.momentum, self.eps)
        return F.leaky_relu(y, self.activation_param)


def _create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1)[None]


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "cv2" not in sys.modules:
        mod("cv2", COLORMAP_JET=2, INTER_NEAREST=0)
    if "torchvision" not in sys.modules:
        tv = mod("torchvision")
        tv.transforms = mod("torchvision.transforms")
        tv.transforms.functional = mod("torchvision.transforms.functional")
    if "kornia" not in sys.modules:
        k = mod("kornia")
        k.utils = mod("kornia.utils", create_meshgrid=_create_meshgrid)
        k.create_meshgrid = _create_meshgrid
    if "inplace_abn" not in sys.modules:
        mod("inplace_abn", InPlaceABN=InPlaceABN)
    if "warmup_scheduler" not in sys.modules:
        mod("warmup_scheduler", GradualWarmupScheduler=object)


import contextlib


@contextlib.contextmanager
def zero_filled_empty():
    """torch.empty -> torch.zeros while the reference runs: models.py:858 leaves the border of the cost volume's first three
    channels unwritten when pad > 0; the fixtures define that border as 0 (what the HIP path writes)."""
    orig = torch.empty
    torch.empty = lambda *a, **k: torch.zeros(*a, **{kk: v for kk, v in k.items() if kk != "memory_format"})
    try:
        yield
    finally:
        torch.empty = orig


_REF = None


def load_reference():
    """Returns (models, renderer, utils) modules of the real reference, CPU-only."""
    global _REF
    if _REF is not None:
        return _REF
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference not present at {REF_ROOT} (only exists in the authoring container)")
    _install_stubs()
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # models.py:37
    # the reference modules are called `models`, `renderer`, `utils`: import them under
    # those names from REF_ROOT, then remove the path again.
    saved = {n: sys.modules.pop(n, None) for n in ("models", "renderer", "utils")}
    sys.path.insert(0, REF_ROOT)
    try:
        import models as ref_models
        import renderer as ref_renderer
        import utils as ref_utils
    finally:
        sys.path.remove(REF_ROOT)
    torch.autograd.set_detect_anomaly(False)  # models.py:2 turns it on
    # keep them reachable as ref_* and restore whatever was there
    sys.modules["ref_models"], sys.modules["ref_renderer"], sys.modules["ref_utils"] = \
        ref_models, ref_renderer, ref_utils
    for n, m in saved.items():
        if m is not None:
            sys.modules[n] = m
    _REF = (ref_models, ref_renderer, ref_utils)
    return _REF


def reference_args(**over):
    """Namespace with the fields create_nerf_mvs / rendering read (opt.py defaults)."""
    d = dict(multires=10, i_embed=0, pts_dim=3, multires_views=4, dir_dim=3, netdepth=6, netwidth=128,
             feat_dim=20, net_type="v0", N_importance=0, netchunk=1024, ckpt=None, perturb=1.0,
             N_samples=128, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0, img_downscale=1.0,
             use_color_volume=False, pad=24, batch_size=1024, chunk=1024)
    d.update(over)
    return types.SimpleNamespace(**d)


def load_reference_ray_utils():
    """data/ray_utils.py of the real reference (ray_marcher, ray_marcher_fine, sample_pdf).  Loaded from its file so that
    data/__init__.py (dataset classes, PIL/cv2 readers) is not executed; it star-needs `renderer` and `utils` by name."""
    import importlib.util
    ref_models, ref_renderer, ref_utils = load_reference()
    saved = {n: sys.modules.get(n) for n in ("renderer", "utils")}
    sys.modules["renderer"], sys.modules["utils"] = ref_renderer, ref_utils
    try:
        spec = importlib.util.spec_from_file_location("ref_ray_utils", os.path.join(REF_ROOT, "data", "ray_utils.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for n, v in saved.items():
            if v is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = v
    return m


def load_reference_networks(args=None, ckpt=os.path.join(REF_ROOT, "ckpts/mvsnerf-v0.tar")):
    """create_nerf_mvs(use_mvs=True, dir_embedder=False, pts_embedder=True) as train_mvs_nerf_pl.py:45."""
    ref_models, _, _ = load_reference()
    args = args or reference_args()
    args.ckpt = ckpt
    _orig = torch.load
    torch.load = lambda f, *a, **k: _orig(f, map_location="cpu", weights_only=False)
    try:
        kw_train, kw_test, _, _ = ref_models.create_nerf_mvs(args, use_mvs=True, dir_embedder=False, pts_embedder=True)
    finally:
        torch.load = _orig
    return args, kw_train
