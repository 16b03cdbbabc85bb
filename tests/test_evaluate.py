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


def test_ssim_restatement_against_a_direct_window_evaluation():
    """evaluate.ssim (uniform-filter form of skimage 0.19's structural_similarity defaults) against the definition evaluated window by window in
    float64: for every interior pixel the 7x7 means, sample variances and covariance, the SSIM formula, the mean.  Plus the properties any SSIM has."""
    import numpy as np
    from mvsnerf_amd import evaluate as E
    rng = np.random.default_rng(0)
    a = rng.random((20, 23, 3)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape).astype(np.float32), 0, 1)
    for R in (None, 1.0):
        got = E.ssim(a, b, data_range=R)
        Rv = 2.0 if R is None else R
        C1, C2 = (0.01 * Rv) ** 2, (0.03 * Rv) ** 2
        vals = []
        for c in range(3):
            acc = []
            for i in range(3, 20 - 3):
                for j in range(3, 23 - 3):
                    x = a[i - 3:i + 4, j - 3:j + 4, c].astype(np.float64).ravel()
                    y = b[i - 3:i + 4, j - 3:j + 4, c].astype(np.float64).ravel()
                    ux, uy = x.mean(), y.mean()
                    vx, vy, vxy = x.var(ddof=1), y.var(ddof=1), ((x - ux) * (y - uy)).sum() / 48.0
                    acc.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
            vals.append(np.mean(acc))
        assert abs(got - float(np.mean(vals))) < 2e-5, (R, got, float(np.mean(vals)))
    assert abs(E.ssim(a, a) - 1.0) < 1e-6 and abs(E.ssim(a, b) - E.ssim(b, a)) < 1e-7
    assert E.ssim(a, b) > E.ssim(a, b, data_range=1.0)            # the library default for float images (range 2) flatters the score
    import pytest
    with pytest.raises(ValueError):
        E.ssim(a[:5], b[:5])
