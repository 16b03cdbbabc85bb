"""CPU-only: the arithmetic claim behind the opt-in "fp16x3" kernels (csrc/mlp_f16x3.hip, csrc/conv_f16x3.hip), checked without a GPU.
An fp32 operand as two round-to-nearest fp16 pieces carries 22 significant bits, and a product taken as x0*w0 + x0*w1 + x1*w0 (exact piece
products, wide accumulation) is fp32-grade: on the shipped MLP weights the split path is as close to the float64 result as the torch fp32
path (= the reference's arithmetic, models.py:194-222) is, while the two-piece BF16 split is two orders of magnitude away.
The GPU-side counterparts are tests/test_gpu_fp16x3.py and tests/test_gpu_fp16x3_encoder.py."""
import torch

from tests.util import load_weights


def _pieces(x, dtype):
    a0 = x.to(dtype)
    a1 = (x - a0.float()).to(dtype)
    return a0.double(), a1.double()


def test_two_fp16_pieces_carry_22_bits():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(100000, generator=g) * s for s in (1e-3, 1.0, 30.0, 2000.0)])
    a0, a1 = _pieces(x, torch.float16)
    err = (a0 + a1 - x.double()).abs()
    # normal lo pieces: 2^-22 relative; lo pieces below fp16's normal range are subnormals with an absolute spacing of 2^-24
    assert bool((err <= 2.0 ** -22 * x.double().abs() + 2.0 ** -25).all())
    b0, b1 = _pieces(x, torch.bfloat16)
    assert float(((b0 + b1 - x.double()).abs() / x.double().abs().clamp_min(1e-30)).max()) > 2.0 ** -18      # two bf16 pieces: 16 bits


def _mlp(x, sd, lin):
    p = "nerf."
    pts, feat, dirs = x[..., :63], x[..., 63:-3], x[..., -3:]
    bias = lin(p + "pts_bias", feat)
    h = pts
    for i in range(6):
        h = torch.relu(lin(p + f"pts_linears.{i}", h) * bias)
        if i == 4:
            h = torch.cat([pts, h], -1)
    alpha = torch.relu(lin(p + "alpha_linear", h))
    h = torch.cat([lin(p + "feature_linear", h), dirs], -1)
    h = torch.relu(lin(p + "views_linears.0", h))
    return torch.cat([torch.sigmoid(lin(p + "rgb_linear", h)), alpha], -1)


def test_three_piece_products_are_fp32_grade_on_the_shipped_mlp():
    from oracle import mvsnerf_oracle as O
    sd, _ = load_weights()
    g = torch.Generator().manual_seed(1)
    n = 4096
    ndc = torch.rand((n, 3), generator=g) * 1.1 - 0.05
    feat = torch.cat([torch.randn((n, 8), generator=g) * 1.5, torch.rand((n, 12), generator=g)], -1)     # volume features, colours + masks
    dirs = torch.nn.functional.normalize(torch.randn((n, 3), generator=g), dim=-1)
    x = torch.cat([O.embed(ndc), feat, dirs], -1)

    def split_lin(dtype):
        def lin(name, h):
            W, b = sd[name + ".weight"], sd[name + ".bias"]
            h0, h1 = _pieces(h.float(), dtype)
            w0, w1 = _pieces(W, dtype)
            return ((h0 @ w0.T) + (h0 @ w1.T) + (h1 @ w0.T) + b.double()).float()
        return lin
    with torch.no_grad():
        y64 = _mlp(x.double(), {k: v.double() for k, v in sd.items()}, lambda name, h: h @ sd[name + ".weight"].double().T + sd[name + ".bias"].double()).float()
        y32 = _mlp(x, sd, lambda name, h: torch.nn.functional.linear(h, sd[name + ".weight"], sd[name + ".bias"]))
        y16 = _mlp(x, sd, split_lin(torch.float16))
        yb = _mlp(x, sd, split_lin(torch.bfloat16))
        assert torch.allclose(y32, O.renderer_ours(x, sd), atol=0, rtol=0)                                 # the harness restates the oracle
    e32 = float((y32 - y64).abs().max())
    e16 = float((y16 - y64).abs().max())
    eb = float((yb - y64).abs().max())
    scale = float(y64.abs().max())
    assert e16 < 3 * e32 + 1e-6 * scale, (e16, e32)            # fp32 grade (measured: 2.4e-6 against the fp32 path's 2.9e-6 on config-2-like inputs)
    assert eb > 10 * e16                                       # the two-piece bf16 split is not


def test_two_piece_conv0_is_closer_to_float64_than_the_fp32_convolution():
    """conv0 of CostRegNet (models.py:756) with the shipped weights on cost-volume-like values: the operands as two fp16 pieces of x/16 and 16 w
    (what csrc/conv_f16x3.hip multiplies; piece convolutions in float64 here) against the float64 convolution, next to torch's fp32 convolution."""
    import torch.nn.functional as F
    _, sd = load_weights()
    w = sd["cost_reg_2.conv0.conv.weight"]
    g = torch.Generator().manual_seed(2)
    x = torch.cat([torch.rand((1, 9, 10, 20, 24), generator=g), torch.randn((1, 32, 10, 20, 24), generator=g).abs() * 60], 1)     # thumbnails | variances
    xs, ws = x * 0.0625, w * 16.0
    x0, x1 = _pieces(xs, torch.float16)
    w0, w1 = _pieces(ws, torch.float16)
    with torch.no_grad():
        y64 = F.conv3d(x.double(), w.double(), padding=1)
        y32 = F.conv3d(x, w, padding=1)
        ys = F.conv3d(x0, w0, padding=1) + F.conv3d(x0, w1, padding=1) + F.conv3d(x1, w0, padding=1)
    e32, es, scale = float((y32.double() - y64).abs().max()), float((ys - y64).abs().max()), float(y64.abs().max())
    assert es < 4e-7 * scale, (es, scale)          # 22-bit operands: ~3 x 2^-22 per product, partly cancelling over 41 x 27 terms
    assert es < e32                                # below the rounding noise of an fp32 summation
