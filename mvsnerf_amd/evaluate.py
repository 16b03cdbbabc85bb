"""Evaluation protocol of the reference's notebooks (renderer.ipynb) - host-side arithmetic on finished frames.

* `abs_error`, `acc_threshold`            utils.py:71-82
* `psnr_center_crop`                      renderer.ipynb cell 8, l.95-100 (Blender / LLFF: centre crop to 0.8 of each side)
* `psnr_masked`                           renderer.ipynb cell 16, l.120-124 (DTU: pixels whose GT depth is 0 are background)
* `depth_metrics`                         renderer.ipynb cell 16, l.99-106 (abs error and the 0.01/0.05/0.1 accuracy thresholds)

* `ssim`                                  renderer.ipynb cell 8 l.111 / cell 16: `skimage.metrics.structural_similarity(rgb, img, multichannel=True)` restated on
                                          numpy + scipy (skimage is not in this image, so this restatement is NOT pinned against the library here: it follows the
                                          published algorithm of skimage 0.19, the version the notebook's deprecation warning identifies)

LPIPS needs the lpips package and its pretrained VGG weights (absent, no network); it is not part of the hot path.
"""
import numpy as np
import torch

from .utils import mse2psnr


def abs_error(depth_pred, depth_gt, mask):
    """utils.py:71-74."""
    return (depth_pred[mask] - depth_gt[mask]).abs()


def acc_threshold(depth_pred, depth_gt, mask, threshold):
    """utils.py:76-82: per-pixel indicator |err| < threshold over the masked pixels."""
    return (abs_error(depth_pred, depth_gt, mask) < threshold).float()


def _psnr(mse):
    return float(mse2psnr(torch.as_tensor(mse, dtype=torch.float32)))


def psnr_center_crop(rgb, img):
    """rgb, img: (H,W,3) in [0,1].  Crop H//10 rows and W//10 columns from every side, then PSNR of the mean squared error."""
    rgb, img = torch.as_tensor(rgb, dtype=torch.float32), torch.as_tensor(img, dtype=torch.float32)
    hc, wc = rgb.shape[0] // 10, rgb.shape[1] // 10
    if hc == 0 or wc == 0:
        raise ValueError("psnr_center_crop: the reference's [H_crop:-H_crop] slicing is empty for images smaller than 10 pixels")
    a, b = rgb[hc:-hc, wc:-wc], img[hc:-hc, wc:-wc]
    return _psnr(((a - b) ** 2).mean())


def psnr_masked(rgb, img, depth_gt):
    """DTU protocol: mask = (depth_gt == 0) is background; PSNR over the remaining pixels."""
    rgb, img = torch.as_tensor(rgb, dtype=torch.float32), torch.as_tensor(img, dtype=torch.float32)
    keep = torch.as_tensor(depth_gt) != 0
    return _psnr(((rgb[keep] - img[keep]) ** 2).mean())


def depth_metrics(depth_pred, depth_gt, thresholds=(0.01, 0.05, 0.1), gt_scale=1.0 / 200.0):
    """cell 16 l.99-106: mask = depth_gt > 0; GT is in mm/200 units there (depth_gt/200)."""
    depth_pred = torch.as_tensor(depth_pred, dtype=torch.float32)
    depth_gt = torch.as_tensor(depth_gt, dtype=torch.float32)
    mask = depth_gt > 0
    gt = depth_gt * gt_scale
    err = abs_error(depth_pred, gt, mask)
    out = {"abs_err": float(err.mean())}
    for t in thresholds:
        out[f"acc_l_{t}"] = float((err < t).float().mean())
    return out


def ssim(rgb, img, win_size=7, data_range=None, K1=0.01, K2=0.03):
    """Mean structural similarity of two (H,W,3) float images as the reference's notebooks compute it:
    `structural_similarity(rgb, img, multichannel=True)` of skimage 0.19 with its defaults - 7x7 UNIFORM window (no gaussian weights), sample
    covariance (normalised by N-1), K1 = 0.01, K2 = 0.03, the SSIM map cropped by (win_size-1)/2 at every border, mean over pixels and then over
    channels.  data_range=None reproduces the library's default for float images at that version: the width of the dtype's nominal range [-1, 1],
    i.e. 2 - NOT the 1.0 the images actually span (the numbers in the reference's tables were produced that way); pass data_range=1.0 for the
    conventional value."""
    from scipy.ndimage import uniform_filter
    a = np.asarray(torch.as_tensor(rgb, dtype=torch.float32).cpu().numpy())
    b = np.asarray(torch.as_tensor(img, dtype=torch.float32).cpu().numpy())
    if a.shape != b.shape or a.ndim != 3:
        raise ValueError("ssim: two (H,W,C) images of the same shape")
    if min(a.shape[:2]) < win_size or win_size % 2 == 0:
        raise ValueError("ssim: win_size must be odd and not exceed the image sides")
    R = 2.0 if data_range is None else float(data_range)
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    n_p = win_size ** 2
    cov_norm = n_p / (n_p - 1.0)
    pad = (win_size - 1) // 2
    vals = []
    for c in range(a.shape[2]):
        x, y = a[..., c], b[..., c]
        ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
        uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
        vals.append(S[pad:-pad, pad:-pad].astype(np.float64).mean())
    return float(np.mean(vals))
