"""Host-side evaluation protocol (reference notebooks) - plain arithmetic, checked against direct restatements."""
import math

import numpy as np
import pytest
import torch

from mvsnerf_amd import evaluate as E


def test_center_crop_psnr_matches_notebook_arithmetic():
    g = torch.Generator().manual_seed(0)
    rgb, img = torch.rand((50, 70, 3), generator=g), torch.rand((50, 70, 3), generator=g)
    hc, wc = 5, 7
    mse = float(((rgb[hc:-hc, wc:-wc] - img[hc:-hc, wc:-wc]).numpy() ** 2).mean())
    assert abs(E.psnr_center_crop(rgb, img) - (-10.0 * math.log(mse) / math.log(10.0))) < 1e-4
    with pytest.raises(ValueError):
        E.psnr_center_crop(torch.rand(8, 8, 3), torch.rand(8, 8, 3))


def test_masked_psnr_and_depth_metrics():
    g = torch.Generator().manual_seed(1)
    rgb, img = torch.rand((20, 30, 3), generator=g), torch.rand((20, 30, 3), generator=g)
    depth_gt = torch.rand((20, 30), generator=g) * 800 + 400
    depth_gt[:5] = 0
    keep = (depth_gt != 0).numpy()
    mse = float(((rgb.numpy()[keep] - img.numpy()[keep]) ** 2).mean())
    assert abs(E.psnr_masked(rgb, img, depth_gt) - (-10.0 * math.log10(mse))) < 1e-4
    pred = depth_gt / 200 + torch.randn((20, 30), generator=g) * 0.04
    m = E.depth_metrics(pred, depth_gt)
    err = np.abs((pred - depth_gt / 200).numpy()[keep])
    assert abs(m["abs_err"] - err.mean()) < 1e-6
    for t in (0.01, 0.05, 0.1):
        assert abs(m[f"acc_l_{t}"] - (err < t).mean()) < 1e-6
    assert E.acc_threshold(pred, depth_gt / 200, depth_gt > 0, 0.05).shape == (int(keep.sum()),)
