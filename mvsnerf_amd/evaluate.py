"""Evaluation protocol of the reference's notebooks (renderer.ipynb) - host-side arithmetic on finished frames.

* `abs_error`, `acc_threshold`            utils.py:71-82
* `psnr_center_crop`                      renderer.ipynb cell 8, l.95-100 (Blender / LLFF: centre crop to 0.8 of each side)
* `psnr_masked`                           renderer.ipynb cell 16, l.120-124 (DTU: pixels whose GT depth is 0 are background)
* `depth_metrics`                         renderer.ipynb cell 16, l.99-106 (abs error and the 0.01/0.05/0.1 accuracy thresholds)

SSIM / LPIPS need skimage / lpips, which are not in this image; they are not part of the hot path.
"""
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
