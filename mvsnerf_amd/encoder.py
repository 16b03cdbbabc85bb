"""Scene encoder (L1a) of the reference's models.py on libmvsnerf_hip.so: FeatureNet (2-D CNN, csrc/featnet.hip),
plane-sweep variance cost volume (csrc/planesweep.hip) and CostRegNet (csrc/encoder.hip) - no ATen/MIOpen compute between images and volume.

Same class / sub-module / parameter names as the reference (models.py:661-932) so that
`network_mvs_state_dict` of a reference checkpoint loads unchanged.

Internally every 3-D tensor is channel-last in HBM; the tensors handed back to callers are logical
NCDHW views of that memory (no copies), e.g. the neural volume is (1,8,D,h,w) with channels_last_3d
strides and feeds the ray-march kernels without a transpose.
"""
import ctypes
import time

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, dev_f32, stream_ptr

ABN_EPS, ABN_MOMENTUM, ABN_SLOPE = 1e-5, 0.1, 0.01

ENCODER_PRECISION = "auto"      # "auto" | "fp32" | "bf16" | "fp16x3": see encoder_precision


class encoder_precision:
    """`with encoder.encoder_precision(mode): ...` - the arithmetic of conv0 of CostRegNet (models.py:756; 74.5 % of the encoder's FLOPs).  Every
    other layer, the plane sweep's own arithmetic, InPlaceABN statistics, master weights and gradients are fp32 in every mode.
      "auto" (default)  a step that needs gradients takes the fp32-MFMA kernels; a no-grad scene encode (validation_step / render_view / fine-tuning's
                one-off encode) runs the GUARDED fp16 sequence (mvsnerf_sweep_conv0_guarded_fwd): the "fp16x3" pair below, which reports a cost
                value or weight outside fp16's range through a device-side guard word, followed by the fp32 plane sweep + fp32-MFMA conv0
                predicated on that word - they recompute the layer when it is set and cost ~10 us when it is not.  No host synchronisation; the
                volume never saturates (ops.guard_fallbacks() counts the fallbacks).
      "fp32"    conv0 on v_mfma_f32_4x4x1_16B_f32 (csrc/conv_mfma.hip) everywhere
      "fp16x3"  the UNGUARDED pair: the plane sweep stores every cost value as two fp16 pieces of x / 16, conv0 multiplies x0*w0 + x0*w1 + x1*w0 on
                v_mfma_f32_16x16x32_f16 (csrc/conv_f16x3.hip; dropped: <= 2^-22 of a product).  Measured: as far from the reference's CPU results as the fp32 kernel
                (DESIGN.md), 0.53 instead of 0.81 ms; cost values SATURATE at 2^20 (the shipped FeatureNet: < 450)
      "bf16"    the reference's AMP switch (train_mvs_nerf_pl.py:317-318 `precision=16 if args.use_amp`; BASELINE config 3): the cost volume is stored as
                bf16 and conv0 runs forward, data gradient and weight gradient on v_mfma_f32_16x16x32_bf16 (csrc/conv_bf16.hip): operands ROUNDED to bf16."""

    def __init__(self, mode):
        if mode not in ("auto", "fp32", "bf16", "fp16x3"):
            raise ValueError("encoder precision must be 'auto', 'fp32', 'bf16' or 'fp16x3'")
        self.mode = mode

    def __enter__(self):
        global ENCODER_PRECISION
        self.prev, ENCODER_PRECISION = ENCODER_PRECISION, self.mode

    def __exit__(self, *exc):
        global ENCODER_PRECISION
        ENCODER_PRECISION = self.prev


# ------------------------------------------------------------------ InPlaceABN stand-in
class InPlaceABN(nn.Module):
    """Parameter container for mapillary InPlaceABN (third-party, not in the reference tree; semantics restated:
    y = leaky_relu(batch_norm(x, gamma=|w|+eps), 0.01)).  Inside FeatureNet / CostRegNet the normalisation is applied
    lazily by the consuming convolution (abn_stats -> per-channel scale/shift); `forward` is the stand-alone form on the same HIP
    kernels (statistics + apply) for callers that use the layer on its own."""

    def __init__(self, num_features, eps=ABN_EPS, momentum=ABN_MOMENTUM, affine=True, activation="leaky_relu", activation_param=ABN_SLOPE):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.activation_param = num_features, eps, momentum, activation_param
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        """(N,C,H,W) or (1,C,D,H,W) -> same logical shape (channel-last memory).  No autograd (inside the networks the backward
        goes through _abn_bwd); C must be a multiple of 4 (all the reference uses: 8, 16, 32, 64)."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad or self.bias.requires_grad):
            # stand-alone differentiable use (outside FeatureNet / CostRegNet, whose backward runs on _abn_bwd): the same function through
            # autograd-capable torch ops - plumbing for callers that build their own networks from this layer, not a hot-path route
            import torch.nn.functional as F
            y = F.batch_norm(x, self.running_mean, self.running_var, self.weight.abs() + self.eps, self.bias, self.training, self.momentum, self.eps)
            return F.leaky_relu(y, self.activation_param)
        C = self.num_features
        if x.dim() not in (4, 5) or x.shape[1] != C or C % 4 or self.activation_param != ABN_SLOPE:
            raise NotImplementedError(f"InPlaceABN.forward: expected (N,{C},H,W) or (1,{C},D,H,W) with C % 4 == 0 and slope {ABN_SLOPE}")
        if x.dim() == 4:
            buf, ld = _images_channel_last(x, C)
            dims = (x.shape[0], x.shape[2], x.shape[3], C)
        else:
            buf, ld = _as_channel_last(x, C)
            dims = (x.shape[2], x.shape[3], x.shape[4], C)
        if ld != C:
            buf = buf[..., :C].contiguous()
        n = dims[0] * dims[1] * dims[2]
        scale, shift, mean, invstd = _abn_stats(buf, n, self, update_running=self.training)
        out = _apply_add(_Lazy(buf, scale, shift, dims, mean, invstd))
        _flush_nbt()
        return out.permute(0, 3, 1, 2) if x.dim() == 4 else _cl_view_to_ncdhw(out)


class _PackBatch:
    """Weight re-layouts collected while `active` and issued as ONE launch on exit (mvsnerf_pack_weights_multi).  Outside a batch a
    re-layout is its own launch of the same kernel.  MVSNet.forward replays the re-layouts the previous step asked for (`log`, filled by
    the getters on their cache misses) inside a batch, so a training step packs its ~50 weight layouts in one launch up front."""
    active = None

    def __init__(self):
        self.jobs = []

    def __enter__(self):
        _PackBatch.active = self
        return self

    def __exit__(self, *exc):
        _PackBatch.active = None
        jobs, self.jobs = self.jobs, []
        _PackBatch.launch(jobs)        # also on an exception: the getters have already cached these buffers as packed
        return False

    @staticmethod
    def launch(jobs):
        import ctypes
        lib = _lib.lib()
        for i in range(0, len(jobs), 64):
            part = jobs[i:i + 64]
            n = len(part)
            flat = [v for _, _, q in part for v in q]
            check(lib.mvsnerf_pack_weights_multi(n, (ctypes.c_void_p * n)(*[w for w, _, _ in part]), (ctypes.c_void_p * n)(*[d.data_ptr() for _, d, _ in part]),
                                                 (ctypes.c_int * (9 * n))(*flat), stream_ptr()), "pack_weights_multi")


def _pack(w_ptr, dst, kind, ntaps, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip):
    """dst <- layout `kind` of the weights at device address w_ptr (see mvsnerf_pack_weights_multi)."""
    job = (w_ptr, dst, (kind, ntaps, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip))
    if _PackBatch.active is not None:
        _PackBatch.active.jobs.append(job)
    else:
        _PackBatch.launch([job])
    return dst


def _log_pack(pk, method, *args):
    """Remember that `pk.method(*args)` had to pack (replayed in a batch at the start of the next MVSNet.forward)."""
    log = getattr(pk, "log", None)
    if log is not None and _PackBatch.active is None:
        entry = (pk, method, args)
        if entry not in log:
            log.append(entry)


class _PackedConv2d:
    """Caches the [k*k][cin_pad][cout] re-layouts of a Conv2d weight (re-packed when it changes); get_bf16: the bf16 fragments (_get_bf16_2d).
    mode 'fwd': the layer itself; mode 'dgrad': the convolution computing its data gradient (channel roles swapped,
    taps mirrored for stride 1; the stride-2 layers use the gather-form kernel with un-mirrored taps)."""

    def __init__(self, conv):
        self.conv, self.cache = conv, {}
        self.cout, self.cin, self.k = conv.weight.shape[0], conv.weight.shape[1], conv.weight.shape[2]
        self.cin_pad = (self.cin + 3) // 4 * 4

    def get_bf16(self, mode="fwd"):
        return _get_bf16_2d(self, mode)

    def get(self, mode="fwd"):
        w = self.conv.weight
        key = (w.data_ptr(), w._version, _lib.weights_epoch())
        hit = self.cache.get(mode)
        if hit is not None and hit[0] == key:
            return hit[1]
        kk = self.k * self.k
        if mode == "fwd":
            args = (self.cin, self.cout, self.cin_pad, self.cout, kk, self.cin * kk, self.k, 0)
        else:
            args = (self.cout, self.cin, self.cout, self.cin_pad, self.cin * kk, kk, self.k, 1 if self.conv.stride[0] == 1 else 0)
        buf = torch.empty(kk * args[2] * args[3], device=w.device, dtype=torch.float32)
        ci_real, co_real, ci_pad, co_pad, s_ci, s_co, _, flip = args
        _pack(dev_f32(w.detach().contiguous(), "conv weight"), buf, 0, kk, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip)
        _log_pack(self, "get", mode)
        self.cache[mode] = (key, buf)
        return buf


def _get_bf16_2d(pk, mode="fwd"):
    """bf16 B fragments of a Conv2d (or of its stride-1 data gradient) for mvsnerf_conv2d_bf16_fwd; None when that shape has no such kernel."""
    w = pk.conv.weight
    kk = pk.k * pk.k
    if mode == "fwd":
        ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip = pk.cin, pk.cout, pk.cin_pad, pk.cout, kk, pk.cin * kk, 0
    else:
        if pk.conv.stride[0] != 1:
            return None
        ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip = pk.cout, pk.cin, pk.cout, pk.cin_pad, pk.cin * kk, kk, 1
    lib = _lib.lib()
    if co_pad < 8:                       # the data gradient towards the 3-channel images is never needed
        return None
    n = lib.mvsnerf_conv2d_bf16_packed_elems(ci_pad, co_pad, pk.k)
    if n == 0:
        return None
    key = (w.data_ptr(), w._version, _lib.weights_epoch())
    name = mode + "_bf16"
    hit = pk.cache.get(name)
    if hit is not None and hit[0] == key:
        return hit[1]
    buf = torch.empty(n, device=w.device, dtype=torch.bfloat16)
    _pack(dev_f32(w.detach().contiguous(), "conv weight"), buf, 3, kk, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip)
    _log_pack(pk, "get_bf16", mode)
    pk.cache[name] = (key, buf)
    return buf


class _Nchw3:
    """Marker for _conv2d: the input is the caller's contiguous (N,3,H,W) fp32 image tensor itself (no channel-last copy was made)."""

    def __init__(self, x):
        self.x = x


def _conv2d(src, dims_in, cin_ld, wbuf, cin_k, cout_k, ksize, stride, bias=None, want_stats=False, packed=None, mode="fwd"):
    """2-D convolution kernel launch (padding k//2): input (N,H,W) with channel stride cin_ld -> raw (N,Ho,Wo,cout_k).
    want_stats: returns (raw, InPlaceABN partial sums | None) - the matrix-core layers leave them from their own launch."""
    N, H, W, _ = dims_in
    P = ksize // 2
    Ho, Wo = (H + 2 * P - ksize) // stride + 1, (W + 2 * P - ksize) // stride + 1
    dev = (src.x if isinstance(src, (_Lazy, _Nchw3)) else src).device
    out = torch.empty((N, Ho, Wo, cout_k), device=dev, dtype=torch.float32)
    lib = _lib.lib()
    wq = _get_bf16_2d(packed, mode) if (_LAYER_BF16[0] and packed is not None and not isinstance(src, _Nchw3)) else None
    if wq is not None:               # use_amp: FeatureNet on the bf16 matrix cores (csrc/conv3d_bf16.hip), statistics from the same launch
        part, nblk = None, 0
        if want_stats and bias is None and FUSED_ABN_STATS:
            nblk = lib.mvsnerf_conv2d_bf16_tiles(N, H, W, ksize, stride)
            part = torch.empty(nblk * 2 * cout_k, device=dev, dtype=torch.float32)
        check(lib.mvsnerf_conv2d_bf16_fwd(*_ptrs(src), cin_k, cin_ld, N, H, W, wq.data_ptr(), 0 if bias is None else bias.data_ptr(), cout_k, ksize, stride,
                                          out.data_ptr(), 0 if part is None else part.data_ptr(), stream_ptr()), "conv2d_bf16_fwd")
        return (out, None if part is None else (part, nblk)) if want_stats else out
    if callable(wbuf):
        wbuf = wbuf()
    if isinstance(src, _Nchw3):      # FeatureNet's first layer of a no-grad encode: straight from the (N,3,H,W) images (csrc/featnet.hip conv2d_c8_lds_kernel)
        nblk = lib.mvsnerf_conv2d_mfma_tiles(4, 8, N, H, W, 3, 1)
        part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
        check(lib.mvsnerf_conv2d_c3_nchw_fwd_stats(src.x.data_ptr(), N, H, W, wbuf.data_ptr(), out.data_ptr(), part.data_ptr(), stream_ptr()), "conv2d_c3_nchw_fwd_stats")
        return out, (part, nblk)
    if want_stats and bias is None and FUSED_ABN_STATS:
        nblk = lib.mvsnerf_conv2d_mfma_tiles(cin_k, cout_k, N, H, W, ksize, stride)
        if nblk > 0:
            part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
            check(lib.mvsnerf_conv2d_fwd_stats(*_ptrs(src), cin_k, cin_ld, N, H, W, wbuf.data_ptr(), cout_k, ksize, stride, out.data_ptr(),
                                               part.data_ptr(), stream_ptr()), "conv2d_fwd_stats")
            return out, (part, nblk)
    check(lib.mvsnerf_conv2d_fwd(*_ptrs(src), cin_k, cin_ld, N, H, W, wbuf.data_ptr(), 0 if bias is None else bias.data_ptr(), cout_k,
                                 ksize, stride, out.data_ptr(), stream_ptr()), "conv2d_fwd")
    return (out, None) if want_stats else out


class ConvBnReLU(nn.Module):
    """reference models.py:661-672 (2-D): Conv2d(bias=False) + InPlaceABN, on the HIP kernels of csrc/featnet.hip."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, norm_act=InPlaceABN):
        super().__init__()
        if (kernel_size, stride, pad) not in ((3, 1, 1), (5, 2, 2)):
            raise NotImplementedError("ConvBnReLU: the HIP kernels cover (k3,s1,p1) and (k5,s2,p2) - all FeatureNet uses")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)
        self.k, self.stride = kernel_size, stride
        self._packed = _PackedConv2d(self.conv)

    def lazy(self, src, dims_in, cin_ld):
        pk = self._packed
        raw, partials = _conv2d(src, dims_in, cin_ld, pk.get, pk.cin_pad, pk.cout, self.k, self.stride, want_stats=True, packed=pk)
        N, H, W, C = raw.shape
        scale, shift, mean, invstd = _abn_stats(raw, N * H * W, self.bn, update_running=self.bn.training,
                                                partials=partials if self.bn.training else None)
        return _Lazy(raw, scale, shift, (N, H, W, C), mean, invstd)

    def forward(self, x):
        """Stand-alone call on (N,Cin,H,W) -> activated (N,Cout,H',W') (channel-last memory)."""
        ops._need_no_grad(x, *self.parameters(), op="ConvBnReLU")
        buf, ld = _images_channel_last(x, self._packed.cin_pad)
        N, _, H, W = x.shape
        out = _apply_add(self.lazy(buf, (N, H, W, ld), ld)).permute(0, 3, 1, 2)
        _flush_nbt()
        return out


def _images_channel_last(x, cin_pad):
    """(N,C,H,W) -> ([N][H][W][ld] buffer, ld).  Zero-copy for permuted channel-last views (what this module emits)."""
    N, C, H, W = x.shape
    st = x.stride()
    ld = st[3]
    if (x.is_cuda and x.dtype == torch.float32 and st[1] == 1 and ld >= cin_pad and ld % 4 == 0 and st[2] == W * ld and st[0] == H * W * ld
            and x.storage_offset() % 4 == 0):
        return torch.as_strided(x, (N, H, W, ld), (H * W * ld, W * ld, ld, 1)), ld
    dst = torch.empty((N, H, W, cin_pad), device=x.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_nchw_to_nhwc(dev_f32(x.detach().contiguous(), "x"), dst.data_ptr(), N, C, H, W, cin_pad, stream_ptr()), "nchw_to_nhwc")
    return dst, cin_pad


class FeatureNet(nn.Module):
    """reference models.py:688-722.  (N,3,H,W) -> (N,32,H/4,W/4) (channel-last memory behind an NCHW view).
    Eight ConvBnReLU layers with lazily-applied InPlaceABN (statistics over all N images, as the reference's B*V batch)
    and the biased 1x1 `toplayer`, all on HIP kernels; differentiable through _FeatureNetFunction."""

    def __init__(self, norm_act=InPlaceABN):
        super().__init__()
        self.conv0 = nn.Sequential(ConvBnReLU(3, 8, 3, 1, 1, norm_act=norm_act), ConvBnReLU(8, 8, 3, 1, 1, norm_act=norm_act))
        self.conv1 = nn.Sequential(ConvBnReLU(8, 16, 5, 2, 2, norm_act=norm_act), ConvBnReLU(16, 16, 3, 1, 1, norm_act=norm_act),
                                   ConvBnReLU(16, 16, 3, 1, 1, norm_act=norm_act))
        self.conv2 = nn.Sequential(ConvBnReLU(16, 32, 5, 2, 2, norm_act=norm_act), ConvBnReLU(32, 32, 3, 1, 1, norm_act=norm_act),
                                   ConvBnReLU(32, 32, 3, 1, 1, norm_act=norm_act))
        self.toplayer = nn.Conv2d(32, 32, 1)
        self._top_packed = _PackedConv2d(self.toplayer)

    def _layers(self):
        return [*self.conv0, *self.conv1, *self.conv2]

    def _run(self, x, keep_input=True):
        """keep_input = False (a no-grad forward: nobody reads the channel-last copy of the images afterwards): the first layer reads the (N,3,H,W) tensor itself."""
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"FeatureNet: expected (N,3,H,W) images, got {tuple(x.shape)}")
        N, _, H, W = x.shape
        bf16 = ENCODER_PRECISION == "bf16" and BF16_LAYERS
        if (not keep_input and not bf16 and FUSED_ABN_STATS and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
                and x.data_ptr() % 4 == 0 and self.conv0[0].bn.training):
            img, ld = None, 4
            src, dims = _Nchw3(x.detach()), (N, H, W, 4)
        else:
            img, ld = _images_channel_last(x, 4)
            src, dims = img, (N, H, W, ld)
        lz = []
        with _layer_precision(bf16):      # use_amp: the 2-D layers on the bf16 matrix cores as well
            for lay in self._layers():
                z = lay.lazy(src, dims, ld)
                lz.append(z)
                src, dims, ld = z, z.dims, z.dims[3]      # (materialising the activation here, as the 3-D up-blocks do, is 3 % slower)
            top = _conv2d(src, dims, ld, self._top_packed.get, 32, 32, 1, 1, bias=dev_f32_tensor(self.toplayer.bias), packed=self._top_packed)
        _flush_nbt()
        return (img, 4 if img is None else img.shape[3]), lz, top

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            params = []
            for lay in self._layers():
                params += [lay.conv.weight, lay.bn.weight, lay.bn.bias]
            return _FeatureNetFunction.apply(x, self, *params, self.toplayer.weight, self.toplayer.bias)
        _, _, top = self._run(x, keep_input=False)
        return top.permute(0, 3, 1, 2)


def dev_f32_tensor(t):
    t = t.detach()
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError("expected a contiguous fp32 device tensor")
    return t


class _PartialSums:
    """The per-workgroup partial results of the weight-gradient kernels of one backward pass, reduced together at its end
    (mvsnerf_partial_sum_multi: two launches for all of them instead of two per layer)."""

    def __init__(self):
        self.jobs = []

    def add(self, ws, n_part, gw):
        self.jobs.append((ws, int(n_part), gw))

    def flush(self):
        import ctypes
        lib = _lib.lib()
        for i in range(0, len(self.jobs), 32):
            jobs = self.jobs[i:i + 32]
            n = len(jobs)
            n_out = [gw.numel() for _, _, gw in jobs]
            scratch = torch.empty(lib.mvsnerf_partial_sum_multi_scratch_floats(sum(n_out)), device=jobs[0][2].device, dtype=torch.float32)
            check(lib.mvsnerf_partial_sum_multi(n, (ctypes.c_void_p * n)(*[ws.data_ptr() for ws, _, _ in jobs]),
                                                (ctypes.c_int * n)(*[p for _, p, _ in jobs]), (ctypes.c_int64 * n)(*n_out),
                                                (ctypes.c_void_p * n)(*[gw.data_ptr() for _, _, gw in jobs]), scratch.data_ptr(), stream_ptr()),
                  "partial_sum_multi")
        self.jobs = []


def _wgrad2d(gx, A, X, B, ldx, g_dims, x_dims, ksize, stride, shape, sums=None):
    """gW[a][b][tap] = sum_o G[o][a] X[o*stride - k//2 + tap][b]  (mvsnerf_conv2d_wgrad).  sums: a _PartialSums that finishes gw later."""
    lib = _lib.lib()
    gw = torch.empty(shape, device=gx.device, dtype=torch.float32)
    if _LAYER_BF16[0] and BF16_WGRAD:
        n_parts = lib.mvsnerf_conv_wgrad_bf16_parts(A, B, g_dims[0], g_dims[1], g_dims[2], 1, ksize, stride)
        if n_parts > 0:             # use_amp: bf16 operands, v_mfma_f32_16x16x32_bf16 (csrc/wgrad_bf16.hip)
            ws = torch.empty(lib.mvsnerf_conv_wgrad_bf16_workspace_floats(A, B, 1, ksize) if sums is None else n_parts * gw.numel(), device=gx.device, dtype=torch.float32)
            check(lib.mvsnerf_conv_wgrad_bf16(gx.data_ptr(), 0, 0, 0, 0, 0, A, *_ptrs(X), 0, 0, 0, B, ldx, g_dims[0], g_dims[1], g_dims[2],
                                              x_dims[0], x_dims[1], x_dims[2], 1, ksize, stride, 0 if sums is not None else gw.data_ptr(), ws.data_ptr(),
                                              stream_ptr()), "conv_wgrad_bf16")
            if sums is not None:
                sums.add(ws, n_parts, gw)
            return gw
    ws = torch.empty(lib.mvsnerf_conv2d_wgrad_workspace_floats(A, B, ksize), device=gx.device, dtype=torch.float32)
    check(lib.mvsnerf_conv2d_wgrad(gx.data_ptr(), A, *_ptrs(X), B, ldx, g_dims[0], g_dims[1], g_dims[2], x_dims[1], x_dims[2], ksize, stride,
                                   0 if sums is not None else gw.data_ptr(), ws.data_ptr(), stream_ptr()), "conv2d_wgrad")
    if sums is not None:
        sums.add(ws, lib.mvsnerf_conv2d_wgrad_parts(A, B, g_dims[0], g_dims[1], g_dims[2], ksize, stride), gw)
    return gw


class _FeatureNetFunction(torch.autograd.Function):
    """FeatureNet with gradients to its 8 conv weights, 8 ABN weight/bias pairs and the toplayer weight/bias
    (the input images carry no gradient)."""

    @staticmethod
    def forward(ctx, x, net, *params):
        (img, ld), lz, top = net._run(x)
        ctx.net, ctx.img, ctx.ld, ctx.lz = net, img, ld, lz
        ctx.layers_bf16 = ENCODER_PRECISION == "bf16" and BF16_LAYERS
        return top.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_out):
        with _layer_precision(ctx.layers_bf16):
            return _FeatureNetFunction._backward(ctx, g_out)

    @staticmethod
    def _backward(ctx, g_out):
        net, lz, lib = ctx.net, ctx.lz, _lib.lib()
        L = net._layers()
        g = g_out.permute(0, 2, 3, 1)
        g = g if g.is_contiguous() else g.contiguous()                 # (N,h,w,32) grad of the toplayer output
        N, h, w, _ = g.shape
        last = lz[-1]
        # toplayer: bias, weight (against the activated conv2.2 output), data gradient (1x1 conv with W^T)
        gb = torch.empty(32, device=g.device, dtype=torch.float32)
        ws = torch.empty(lib.mvsnerf_channel_sum_workspace_floats(32), device=g.device, dtype=torch.float32)
        check(lib.mvsnerf_channel_sum(g.data_ptr(), N * h * w, 32, gb.data_ptr(), ws.data_ptr(), stream_ptr()), "channel_sum")
        sums = _PartialSums()
        gw_top = _wgrad2d(g, 32, last, 32, 32, (N, h, w), last.dims, 1, 1, tuple(net.toplayer.weight.shape), sums)
        g_act = _conv2d(g, (N, h, w, 32), 32, lambda: net._top_packed.get("dgrad"), 32, 32, 1, 1, packed=net._top_packed, mode="dgrad")
        grads = [None] * len(L)
        for i in range(len(L) - 1, -1, -1):
            lay, out_lz = L[i], lz[i]
            pk = lay._packed
            gx, gbw, gbb = _abn_bwd(out_lz, lay.bn, g_act)
            if i > 0:
                xin, x_dims, x_ld = lz[i - 1], lz[i - 1].dims, lz[i - 1].dims[3]
            else:
                xin, x_dims, x_ld = ctx.img, tuple(ctx.img.shape), ctx.ld
            gw = _wgrad2d(gx, pk.cout, xin, pk.cin, x_ld, out_lz.dims[:3], x_dims, lay.k, lay.stride, tuple(lay.conv.weight.shape), sums)
            grads[i] = (gw, gbw, gbb)
            if i == 0:
                break
            if lay.stride == 1:
                g_act = _conv2d(gx, out_lz.dims, pk.cout, lambda: pk.get("dgrad"), pk.cout, pk.cin, lay.k, 1, packed=pk, mode="dgrad")
            else:
                Nn, Ho, Wo, _ = out_lz.dims
                Hi, Wi = x_dims[1], x_dims[2]
                g_act = torch.empty((Nn, Hi, Wi, pk.cin), device=g.device, dtype=torch.float32)
                check(lib.mvsnerf_conv2d_dgrad_k5s2(gx.data_ptr(), pk.cout, Nn, Ho, Wo, pk.get("dgrad").data_ptr(), pk.cin, Hi, Wi,
                                                    g_act.data_ptr(), stream_ptr()), "conv2d_dgrad_k5s2")
        sums.flush()
        out = [None, None]
        for gw, gbw, gbb in grads:
            out += [gw, gbw, gbb]
        return tuple(out + [gw_top, gb])


# ------------------------------------------------------------------ channel-last plumbing
class _Lazy:
    """A raw conv output x[d][y][x][C] plus the (scale, shift) of its pending InPlaceABN (and the batch statistics
    the backward pass needs)."""
    __slots__ = ("x", "scale", "shift", "dims", "mean", "invstd")

    def __init__(self, x, scale, shift, dims, mean=None, invstd=None):
        self.x, self.scale, self.shift, self.dims, self.mean, self.invstd = x, scale, shift, dims, mean, invstd


def _cl_view_to_ncdhw(x_cl, C=None):
    """(D,H,W,Cpad) channel-last buffer -> logical (1,C,D,H,W) view (no copy)."""
    v = x_cl if C is None else x_cl[..., :C]
    return v.permute(3, 0, 1, 2).unsqueeze(0)


def _as_channel_last(x, cin_pad):
    """Logical (1,C,D,H,W) tensor -> (buffer, ld) with buffer[d][y][x][ld] holding the channels first.
    Zero-copy when x is a view produced by _cl_view_to_ncdhw with a padded stride >= cin_pad; else one HIP transpose."""
    if x.dim() != 5 or x.shape[0] != 1:
        raise RuntimeError(f"expected a (1,C,D,H,W) volume, got {tuple(x.shape)}")
    _, C, D, H, W = x.shape
    st = x.stride()
    ld = st[4]
    if (st[1] == 1 and st[3] == W * ld and st[2] == H * W * ld and ld >= cin_pad and ld % 4 == 0 and x.storage_offset() == 0
            and x.untyped_storage().nbytes() >= D * H * W * ld * 4 and x.dtype == torch.float32 and x.is_cuda):
        buf = torch.as_strided(x, (D, H, W, ld), (H * W * ld, W * ld, ld, 1))
        return buf, ld
    src = x[0].contiguous()
    dst = torch.empty((D, H, W, cin_pad), device=x.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_nchw_to_nhwc(dev_f32(src, "x"), dst.data_ptr(), 1, C, D * H, W, cin_pad, stream_ptr()), "nchw_to_nhwc")
    return dst, cin_pad


class _PackedConv:
    """Caches the [27][cin_pad][cout_pad] re-layouts of a Conv3d / ConvTranspose3d weight (re-packed when it changes).
    mode 'fwd': the layer itself.  mode 'dgrad': the convolution that computes its data gradient -
      Conv3d s1 -> Conv3d s1 with mirrored taps and swapped channel roles; Conv3d s2 -> ConvTranspose3d s2;
      ConvTranspose3d s2 -> Conv3d s2."""

    def __init__(self, conv, transposed):
        self.conv, self.transposed, self.cache = conv, transposed, {}
        w = conv.weight
        self.cin, self.cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        self.cin_pad = (self.cin + 3) // 4 * 4

    def _params(self, mode):
        """(ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip) of the convolution kernel that runs this layer (mode "fwd") or its data
        gradient ("dgrad": kernel input channels = the layer's outputs)."""
        cin, cout = self.cin, self.cout
        if mode == "fwd":
            s_ci, s_co = (cout * 27, 27) if self.transposed else (27, cin * 27)
            return cin, cout, self.cin_pad, cout, s_ci, s_co, 0
        s_ci, s_co = (27, cout * 27) if self.transposed else (cin * 27, 27)
        return cout, cin, cout, self.cin_pad, s_ci, s_co, (1 if (not self.transposed and self.conv.stride[0] == 1) else 0)

    def _cached(self, name, kind, mode, method, *args):
        w = self.conv.weight
        key = (w.data_ptr(), w._version, _lib.weights_epoch())
        hit = self.cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip = self._params(mode)
        buf = torch.empty(27 * ci_pad * co_pad, device=w.device, dtype=torch.float32)
        _pack(dev_f32(w.detach().contiguous(), "conv weight"), buf, kind, 27, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip)
        _log_pack(self, method, *args)
        self.cache[name] = (key, buf)
        return buf

    def get(self, mode="fwd"):
        return self._cached(mode, 0, mode, "get", mode)

    def get_c8(self, mode="fwd"):
        """[ci/4][tap][co][4] layout of get(mode)'s weights: the DMA-staged matrix-core conv0 and the 16 -> 8 transposed layer."""
        return self._cached(mode + "_c8", 1, mode, "get_c8", mode)

    def get_dgrad_slice(self, c0, n):
        """get("dgrad") restricted to the layer inputs c0 .. c0+n-1 (stride-1 Conv3d): [27][cout][n], mirrored taps.  The plane sweep's
        backward needs the gradient of the variance channels only (the warped thumbnails carry no parameters)."""
        w = self.conv.weight
        key = (w.data_ptr(), w._version, _lib.weights_epoch(), c0, n)
        hit = self.cache.get("dgrad_slice")
        if hit is not None and hit[0] == key:
            return hit[1]
        if self.transposed or self.conv.stride[0] != 1 or n % 4 or c0 < 0 or c0 + n > self.cin:
            raise RuntimeError("get_dgrad_slice: stride-1 Conv3d and a channel range inside the layer's inputs (multiple of 4 wide)")
        wc = dev_f32(w.detach().contiguous(), "conv weight")
        buf = torch.empty(27 * self.cout * n, device=w.device, dtype=torch.float32)
        _pack(wc + c0 * 27 * 4, buf, 0, 27, self.cout, n, self.cout, n, self.cin * 27, 27, 1)
        _log_pack(self, "get_dgrad_slice", c0, n)
        self.cache["dgrad_slice"] = (key, buf)
        return buf

    def get_f16x3(self):
        """Two-piece fp16 B fragments of the layer's own weights for csrc/conv_f16x3_tiled.hip (conv1 / conv2 of a no-grad encode):
        mvsnerf_conv3d_f16x3_pack, cached like every other layout."""
        w = self.conv.weight
        key = (w.data_ptr(), w._version, _lib.weights_epoch())
        hit = self.cache.get("f16x3t")
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = _lib.lib()
        buf = torch.empty(lib.mvsnerf_conv3d_f16x3_packed_elems(self.cin), device=w.device, dtype=torch.float16)
        check(lib.mvsnerf_conv3d_f16x3_pack(dev_f32(w.detach().contiguous(), "conv weight"), self.cin, self.cout, buf.data_ptr(), stream_ptr()), "conv3d_f16x3_pack")
        self.cache["f16x3t"] = (key, buf)
        return buf

    def get_mfma(self, mode="fwd"):
        """[tap][ci/8][co][8] layout of get(mode)'s weights for the matrix-core kernels of the 32/64-channel layers."""
        return self._cached(mode + "_m32", 2, mode, "get_mfma", mode)

    def get_bf16(self, mode="fwd"):
        """bf16 B fragments of get(mode)'s convolution for the generic bf16 kernels (csrc/conv3d_bf16.hip); None when the shape has no such kernel.
        The KERNEL that runs `mode` is transposed when exactly one of (layer is transposed, mode is its data gradient of a strided layer) holds:
        a stride-2 Conv3d's data gradient is a transposed convolution and vice versa."""
        w = self.conv.weight
        ci_real, co_real, ci_pad, co_pad, _, _, _ = self._params(mode)
        strided = self.conv.stride[0] == 2
        kernel_t = (self.transposed != (mode == "dgrad")) if strided else False
        lib = _lib.lib()
        n = lib.mvsnerf_conv3d_bf16_packed_elems(ci_pad, co_pad, int(kernel_t))
        if n == 0 or ci_real != ci_pad:
            return None
        key = (w.data_ptr(), w._version, _lib.weights_epoch())
        name = mode + "_bf16"
        hit = self.cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        _, _, _, _, s_ci, s_co, flip = self._params(mode)
        buf = torch.empty(n, device=w.device, dtype=torch.bfloat16)
        # straight from the nn weights (kinds 3 / 4 of mvsnerf_pack_weights_multi): batched with the step's other re-layouts by MVSNet.prepack
        _pack(dev_f32(w.detach().contiguous(), "conv weight"), buf, 4 if kernel_t else 3, 27, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip)
        _log_pack(self, "get_bf16", mode)
        self.cache[name] = (key, buf)
        return buf

    def get_bf16_conv0(self, dgrad=None):
        """bf16 B fragments of the 8-output-channel stride-1 layer (csrc/conv_bf16.hip): dgrad None -> forward; (c_first, n_ci) -> the
        data gradient w.r.t. input channels c_first .. c_first + n_ci - 1."""
        w = self.conv.weight
        key = (w.data_ptr(), w._version, _lib.weights_epoch(), dgrad)
        name = "bf16_fwd" if dgrad is None else "bf16_dgrad"
        hit = self.cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        if self.transposed or self.conv.stride[0] != 1 or self.cout != 8:
            raise RuntimeError("get_bf16_conv0: stride-1 Conv3d with 8 output channels (conv0)")
        lib = _lib.lib()
        wc = dev_f32(w.detach().contiguous(), "conv weight")
        if dgrad is None:
            buf = torch.empty(lib.mvsnerf_conv0_bf16_packed_elems(self.cin), device=w.device, dtype=torch.bfloat16)
            check(lib.mvsnerf_conv0_bf16_pack(wc, self.cin, buf.data_ptr(), stream_ptr()), "conv0_bf16_pack")
        else:
            buf = torch.empty(lib.mvsnerf_conv0_bf16_dgrad_packed_elems(dgrad[1]), device=w.device, dtype=torch.bfloat16)
            check(lib.mvsnerf_conv0_bf16_dgrad_pack(wc, self.cin, dgrad[0], dgrad[1], buf.data_ptr(), stream_ptr()), "conv0_bf16_dgrad_pack")
        self.cache[name] = (key, buf)
        return buf


def _get_f16x3_conv0(pk):
    """fp16 hi / lo B fragments of conv0 (csrc/conv_f16x3.hip), cached like the other layouts of _PackedConv."""
    w = pk.conv.weight
    key = (w.data_ptr(), w._version, _lib.weights_epoch())
    hit = pk.cache.get("f16x3_fwd")
    if hit is not None and hit[0] == key:
        return hit[1]
    if pk.transposed or pk.conv.stride[0] != 1 or pk.cout != 8:
        raise RuntimeError("fp16x3 conv0: stride-1 Conv3d with 8 output channels (conv0)")
    lib = _lib.lib()
    buf = torch.empty(lib.mvsnerf_conv0_f16x3_packed_elems(pk.cin), device=w.device, dtype=torch.float16)
    check(lib.mvsnerf_conv0_f16x3_pack(dev_f32(w.detach().contiguous(), "conv weight"), pk.cin, buf.data_ptr(), stream_ptr()), "conv0_f16x3_pack")
    pk.cache["f16x3_fwd"] = (key, buf)
    return buf


def _abn_stats(raw, n_vox, bn, update_running=True, partials=None):
    """Train-mode InPlaceABN statistics of a raw layer output -> (scale, shift, mean, invstd).  partials = (buffer, n_blocks): the
    producing kernel already left per-workgroup sums (mvsnerf_*_fwd_stats), only stage 2 runs."""
    C = bn.num_features
    dev = raw.device
    out = torch.empty((4, C), device=dev, dtype=torch.float32)        # scale, shift, mean, invstd
    if not bn.training:
        # eval-mode InPlaceABN: running statistics (the reference never leaves .train() for MVSNet -
        # train_mvs_nerf_pl.py:182 - so this is plumbing on C-element vectors, not a hot path; no backward)
        with torch.no_grad():
            out[3] = torch.rsqrt(bn.running_var + bn.eps)
            out[2] = bn.running_mean
            out[0] = (bn.weight.abs() + bn.eps) * out[3]
            out[1] = bn.bias - bn.running_mean * out[0]
        return out[0], out[1], out[2], out[3]
    rm = bn.running_mean.data_ptr() if update_running else 0
    rv = bn.running_var.data_ptr() if update_running else 0
    if partials is not None:
        check(_lib.lib().mvsnerf_abn_finalize(partials[0].data_ptr(), partials[1], C, n_vox, dev_f32(bn.weight.detach(), "bn.weight"),
                                              dev_f32(bn.bias.detach(), "bn.bias"), rm, rv, bn.momentum, bn.eps, out[0].data_ptr(), out[1].data_ptr(),
                                              out[2].data_ptr(), out[3].data_ptr(), stream_ptr()), "abn_finalize")
        if update_running:
            _NBT_PENDING.append(bn.num_batches_tracked)
        return out[0], out[1], out[2], out[3]
    ws = torch.empty(_lib.lib().mvsnerf_abn_workspace_floats(C), device=dev, dtype=torch.float32)
    check(_lib.lib().mvsnerf_abn_stats(raw.data_ptr(), n_vox, C, dev_f32(bn.weight.detach(), "bn.weight"), dev_f32(bn.bias.detach(), "bn.bias"),
                                       rm, rv, bn.momentum, bn.eps, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                       ws.data_ptr(), stream_ptr()), "abn_stats")
    if update_running:
        _NBT_PENDING.append(bn.num_batches_tracked)
    return out[0], out[1], out[2], out[3]


_NBT_PENDING = []
_NBT_DEFER = [0]           # > 0: MVSNet.forward is running FeatureNet - its layers' counters wait for the flush behind CostRegNet (one launch per encode, not two)


def _flush_nbt():
    """`num_batches_tracked += 1` of every layer that ran, as ONE foreach launch per network pass instead of one tiny ATen
    kernel per layer (18 of them per scene encode)."""
    if _NBT_PENDING and not _NBT_DEFER[0]:
        torch._foreach_add_(list(_NBT_PENDING), 1)
        _NBT_PENDING.clear()


def _ptrs(src):
    """(x, scale, shift) pointers of a _Lazy | raw tensor | None."""
    if src is None:
        return 0, 0, 0
    if isinstance(src, _Lazy):
        return src.x.data_ptr(), src.scale.data_ptr(), src.shift.data_ptr()
    return src.data_ptr(), 0, 0


class _ThreadCell:
    """A one-slot cell, one value per host thread: cell[0] reads / writes THIS thread's value (default for a thread that never set it).  The per-launch
    precision switches below are set for the extent of a forward / backward by context managers; two host threads on their own streams (supported: guard
    words and workspaces are per stream) must not see each other's (thread A's `_F16X3_CONSUME[0] = 0` leaking into thread B's conv2 launch)."""

    def __init__(self, default):
        import threading
        self._tls, self._default = threading.local(), default

    def __getitem__(self, i):
        return getattr(self._tls, "v", self._default)

    def __setitem__(self, i, value):
        self._tls.v = value


BF16_WGRAD = True          # A/B switch: the weight gradients of those layers on the bf16 kernel as well (csrc/wgrad_bf16.hip)
BF16_LAYERS = True         # A/B switch (scratch/r4): False keeps conv1 ... conv11 on the fp32 kernels under use_amp (round 3's behaviour)
_LAYER_BF16 = _ThreadCell(False)      # conv1 ... conv11 on the bf16 matrix cores (csrc/conv3d_bf16.hip): set for the extent of a forward / backward by _layer_precision


F16X3_LAYERS = True        # A/B switch: conv1 / conv2 of a no-grad "auto" encode on the guarded fp16x3 LDS-tiled kernel (csrc/conv_f16x3_tiled.hip)
F16X3_MIN_VOXELS = 262144  # ... from this many OUTPUT voxels on (below, the layer is a few microseconds on any kernel and the persistent grid of 512 workgroups is mostly idle)
_LAYER_F16X3 = _ThreadCell(None)      # None, or "guarded" / "plain" for the extent of CostRegNet._run (set by _layers_f16x3)
_F16X3_CONSUME = _ThreadCell(1)       # 0 while conv1 runs inside CostRegNet._run: conv2 (same output size, hence guarded as well) counts / re-arms for both


class _layers_f16x3:
    """conv1 / conv2 on the two-piece fp16 kernel for the enclosed launches: "guarded" (the no-grad default: the layer's fp32 kernel is enqueued
    behind it, predicated on the guard word) or "plain" (encoder_precision("fp16x3"): the kernel alone; NaNs where an operand left fp16's range)."""

    def __init__(self, how):
        self.how = how if F16X3_LAYERS else None

    def __enter__(self):
        self.prev, _LAYER_F16X3[0] = _LAYER_F16X3[0], self.how

    def __exit__(self, *exc):
        _LAYER_F16X3[0] = self.prev


class _layer_precision:
    """The arithmetic of the 3-D layers behind conv0 for the enclosed launches: bf16 operands (use_amp, encoder_precision("bf16")) or fp32.
    A backward pass re-enters it with what its forward ran (the `with encoder_precision(...)` of the caller is long gone by then)."""

    def __init__(self, bf16):
        self.bf16 = bool(bf16)

    def __enter__(self):
        self.prev, _LAYER_BF16[0] = _LAYER_BF16[0], self.bf16

    def __exit__(self, *exc):
        _LAYER_BF16[0] = self.prev


def _conv(src1, src2, dims_in, cin_ld, wbuf, cin_k, cout_k, stride, packed=None, mode="fwd", want_stats=None):
    """k3 p1 convolution kernel launch: input (D,H,W) with channel stride cin_ld -> raw (Do,Ho,Wo,cout_k).
    packed (+ mode): the layer's _PackedConv - lets the 32/64-channel layers take the matrix-core kernel with its own weight layout.
    want_stats: as in _conv_t (None: returns out; True / False: (out, InPlaceABN partial sums | None))."""
    D, H, W, _ = dims_in
    Do, Ho, Wo = (D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1
    dev = (src1.x if isinstance(src1, _Lazy) else src1).device
    out = torch.empty((Do, Ho, Wo, cout_k), device=dev, dtype=torch.float32)
    wq = packed.get_bf16(mode) if (_LAYER_BF16[0] and packed is not None) else None
    if wq is not None:               # use_amp: bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulation, statistics from the same launch
        lib = _lib.lib()
        part, nblk = None, 0
        if want_stats and FUSED_ABN_STATS:
            nblk = lib.mvsnerf_conv3d_bf16_tiles(D, H, W, stride)
            part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
        check(lib.mvsnerf_conv3d_bf16_fwd(*_ptrs(src1), *_ptrs(src2), cin_k, cin_ld, D, H, W, wq.data_ptr(), cout_k, stride, out.data_ptr(),
                                          0 if part is None else part.data_ptr(), stream_ptr()), "conv3d_bf16_fwd")
        return out if want_stats is None else (out, None if part is None else (part, nblk))
    how = _LAYER_F16X3[0]
    if (how is not None and packed is not None and src2 is None and mode == "fwd" and want_stats and FUSED_ABN_STATS and cout_k == 16
            and (cin_k, stride) in ((8, 2), (16, 1)) and Do * Ho * Wo >= F16X3_MIN_VOXELS):
        # conv1 / conv2 of a no-grad encode: fp32-grade results from the fp16 matrix cores (52 / 43 us instead of 107 / 120 at config 2)
        lib = _lib.lib()
        x, sc, sh = _ptrs(src1)
        nblk = lib.mvsnerf_conv3d_f16x3_slots()
        part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
        if how == "guarded":
            w32 = packed.get_mfma("fwd") if cin_k == 8 else (wbuf() if callable(wbuf) else wbuf)      # the layouts of the layer's fp32 kernels
            check(lib.mvsnerf_conv3d_f16x3_guarded_fwd(x, sc, sh, cin_k, cin_ld, D, H, W, packed.get_f16x3().data_ptr(), w32.data_ptr(), cout_k, stride,
                                                       out.data_ptr(), part.data_ptr(), ops.guard_words(out.device).data_ptr(), _F16X3_CONSUME[0], stream_ptr()),
                  "conv3d_f16x3_guarded_fwd")
        else:
            check(lib.mvsnerf_conv3d_f16x3_fwd(x, sc, sh, cin_k, cin_ld, D, H, W, packed.get_f16x3().data_ptr(), cout_k, stride, out.data_ptr(),
                                               part.data_ptr(), stream_ptr()), "conv3d_f16x3_fwd")
        return out, (part, nblk)
    if packed is not None and src2 is None and _lib.lib().mvsnerf_conv3d_mfma_supported(cin_k, cout_k, stride):
        lib = _lib.lib()
        x, sc, sh = _ptrs(src1)
        part, nblk = None, 0
        if want_stats and FUSED_ABN_STATS:
            nblk = lib.mvsnerf_conv3d_mfma_tiles(D, H, W, stride)
            part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
        check(lib.mvsnerf_conv3d_mfma_fwd(x, sc, sh, cin_k, cin_ld, D, H, W, packed.get_mfma(mode).data_ptr(), cout_k, stride,
                                          out.data_ptr(), 0 if part is None else part.data_ptr(), stream_ptr()), "conv3d_mfma_fwd")
        return out if want_stats is None else (out, None if part is None else (part, nblk))
    if callable(wbuf):
        wbuf = wbuf()               # the VALU layout is only packed when this branch runs (no eager pk.get() per step)
    if want_stats and FUSED_ABN_STATS and src2 is None and (cin_k, cout_k, stride) == (16, 16, 1):
        nblk = _lib.lib().mvsnerf_conv3d_tiled_tiles(D, H, W)          # conv2: statistics from the convolution's own launch
        part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
        rc = _lib.lib().mvsnerf_conv3d_fwd_stats(*_ptrs(src1), cin_k, cin_ld, D, H, W, wbuf.data_ptr(), cout_k, stride, out.data_ptr(),
                                                 part.data_ptr(), stream_ptr())
        if rc == 0:
            return out, (part, nblk)
        if rc != -2:                                                   # MVSNERF_EUNSUPPORTED: fall through to the plain launch
            check(rc, "conv3d_fwd_stats")
    check(_lib.lib().mvsnerf_conv3d_fwd(*_ptrs(src1), *_ptrs(src2), cin_k, cin_ld, D, H, W, wbuf.data_ptr(), cout_k, stride,
                                        out.data_ptr(), stream_ptr()), "conv3d_fwd")
    return out if want_stats is None else (out, None)


def _conv_t(src1, src2, dims_in, wbuf, cin_k, cout_k, packed=None, mode="fwd", want_stats=None):
    """k3 s2 p1 op1 transposed-convolution kernel launch: (D,H,W,cin_k) -> raw (2D,2H,2W,cout_k).
    packed (+ mode): the layer's _PackedConv - a plain (materialised) input then takes the matrix-core kernel.
    want_stats None: returns out; True / False: returns (out, partials) - partials = the InPlaceABN partial sums of `out` when the kernel
    could leave them (only asked for with True), else None."""
    D, H, W, _ = dims_in
    dev = (src1.x if isinstance(src1, _Lazy) else src1).device
    out = torch.empty((2 * D, 2 * H, 2 * W, cout_k), device=dev, dtype=torch.float32)
    wq = packed.get_bf16(mode) if (_LAYER_BF16[0] and packed is not None) else None
    if wq is not None:               # use_amp: the eight parity classes as bf16 gather-form convolutions (csrc/conv3d_bf16.hip)
        lib = _lib.lib()
        part, nblk = None, 0
        if want_stats and FUSED_ABN_STATS:
            nblk = lib.mvsnerf_conv_transpose3d_bf16_tiles(D, H, W)
            part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
        check(lib.mvsnerf_conv_transpose3d_bf16_fwd(*_ptrs(src1), *_ptrs(src2), cin_k, D, H, W, wq.data_ptr(), cout_k, out.data_ptr(),
                                                    0 if part is None else part.data_ptr(), stream_ptr()), "conv_transpose3d_bf16_fwd")
        return out if want_stats is None else (out, None if part is None else (part, nblk))
    if packed is not None and _lib.lib().mvsnerf_conv_transpose3d_c8_supported(cin_k, cout_k):
        lib = _lib.lib()            # lazily-activated sources and the skip sum are applied while staging; statistics from the same launch
        part, nblk = None, 0
        if want_stats and FUSED_ABN_STATS:
            nblk = lib.mvsnerf_conv_transpose3d_c8_tiles(D, H, W)
            part = torch.empty(nblk * 16, device=out.device, dtype=torch.float32)
        check(lib.mvsnerf_conv_transpose3d_c8_fwd(*_ptrs(src1), *_ptrs(src2), cin_k, D, H, W, packed.get_c8(mode).data_ptr(), out.data_ptr(),
                                                  0 if part is None else part.data_ptr(), stream_ptr()), "conv_transpose3d_c8_fwd")
        return out if want_stats is None else (out, None if part is None else (part, nblk))
    if (packed is not None and src2 is None and torch.is_tensor(src1) and cin_k % 8 == 0
            and _lib.lib().mvsnerf_conv_transpose3d_mfma_supported(cin_k, cout_k)):
        if want_stats and FUSED_ABN_STATS:                             # conv7 / conv9: the statistics come from the accumulators
            nblk = _lib.lib().mvsnerf_conv_transpose3d_mfma_tiles(cin_k, cout_k, D, H, W)
            part = torch.empty(nblk * 2 * cout_k, device=out.device, dtype=torch.float32)
            check(_lib.lib().mvsnerf_conv_transpose3d_mfma_fwd_stats(src1.data_ptr(), cin_k, D, H, W, packed.get_mfma(mode).data_ptr(), cout_k,
                                                                     out.data_ptr(), part.data_ptr(), stream_ptr()), "conv_transpose3d_mfma_fwd_stats")
            return out, (part, nblk)
        check(_lib.lib().mvsnerf_conv_transpose3d_mfma_fwd(src1.data_ptr(), cin_k, D, H, W, packed.get_mfma(mode).data_ptr(), cout_k,
                                                           out.data_ptr(), stream_ptr()), "conv_transpose3d_mfma_fwd")
        return out if want_stats is None else (out, None)
    if callable(wbuf):
        wbuf = wbuf()
    check(_lib.lib().mvsnerf_conv_transpose3d_fwd(*_ptrs(src1), *_ptrs(src2), cin_k, D, H, W, wbuf.data_ptr(), cout_k,
                                                  out.data_ptr(), stream_ptr()), "conv_transpose3d_fwd")
    return out if want_stats is None else (out, None)


def _apply_add(a, b=None):
    D, H, W, C = a.dims
    out = torch.empty((D, H, W, C), device=a.x.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_abn_apply_add(*_ptrs(a), *_ptrs(b), D * H * W, C, out.data_ptr(), stream_ptr()), "abn_apply_add")
    return out


VOLUME_LAYOUT = "hwdc"      # memory order of the neural volume CostRegNet emits: "hwdc" vol[y][x][d][8] (depth fastest, what the ray march
                            # reads best: include/mvsnerf_hip.h MVSNERF_VOL_HWDC) | "dhwc" vol[d][y][x][8].  The logical tensor is the same.


def _neural_volume(a, b):
    """conv0 + conv11(x) (models.py:766) as the logical (1,8,D,h,w) tensor handed to callers, in the VOLUME_LAYOUT memory order."""
    D, H, W, C = a.dims
    if VOLUME_LAYOUT == "dhwc" or C != 8:
        return _cl_view_to_ncdhw(_apply_add(a, b))
    out = torch.empty((H, W, D, C), device=a.x.device, dtype=torch.float32)
    check(_lib.lib().mvsnerf_abn_apply_add_hwdc(*_ptrs(a), *_ptrs(b), D, H, W, out.data_ptr(), stream_ptr()), "abn_apply_add_hwdc")
    return out.permute(3, 2, 0, 1).unsqueeze(0)


# ------------------------------------------------------------------ 3-D blocks
class ConvBnReLU3D(nn.Module):
    """reference models.py:674-685: Conv3d(k3, bias=False) + InPlaceABN (train-mode statistics)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, norm_act=InPlaceABN):
        super().__init__()
        if kernel_size != 3 or pad != 1 or stride not in (1, 2):
            raise NotImplementedError("ConvBnReLU3D: the HIP kernel is built for k=3, pad=1, stride 1|2 (all the reference uses)")
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)
        self.stride = stride
        self._packed = _PackedConv(self.conv, False)

    def lazy(self, src1, dims_in, cin_ld, src2=None):
        pk = self._packed
        raw, partials = _conv(src1, src2, dims_in, cin_ld, pk.get, pk.cin_pad, pk.cout, self.stride, packed=pk, want_stats=self.bn.training)
        D, H, W, C = raw.shape
        scale, shift, mean, invstd = _abn_stats(raw, D * H * W, self.bn, update_running=self.bn.training, partials=partials)
        return _Lazy(raw, scale, shift, (D, H, W, C), mean, invstd)

    def forward(self, x):
        """Stand-alone call on a logical (1,Cin,D,H,W) tensor -> activated (1,Cout,D',H',W') (channel-last memory)."""
        ops._need_no_grad(x, *self.parameters(), op="ConvBnReLU3D")
        buf, ld = _as_channel_last(x, self._packed_cin_pad())
        D, H, W = x.shape[2:]
        out = _cl_view_to_ncdhw(_apply_add(self.lazy(buf, (D, H, W, ld), ld)))
        _flush_nbt()
        return out

    def _packed_cin_pad(self):
        return self._packed.cin_pad


MATERIALIZE_UP_INPUT = True     # A/B switch (scratch/enc_time.py)
FUSED_ABN_STATS = True          # A/B switch: the 8-channel producers (conv0, conv11) leave their InPlaceABN partial sums themselves
BLOCKED_COST = True             # A/B switch: MVSNet.forward hands conv0 a channel-blocked cost volume on the no-grad path
_BLOCKED_CIN = (32, 36, 40, 44, 48, 52, 56)   # conv0 input widths the matrix-core kernel is instantiated for


class _UpBlock(nn.Sequential):
    """nn.Sequential(ConvTranspose3d, InPlaceABN) of models.py:739-752 (keys `convN.0.weight`, `convN.1.*`)."""

    def __init__(self, cin, cout, norm_act):
        super().__init__(nn.ConvTranspose3d(cin, cout, 3, padding=1, output_padding=1, stride=2, bias=False), norm_act(cout))
        self._packed = _PackedConv(self[0], True)

    def lazy(self, src1, dims_in, src2=None):
        pk = self._packed
        if (MATERIALIZE_UP_INPUT and isinstance(src1, _Lazy)
                and not _lib.lib().mvsnerf_conv_transpose3d_c8_supported(pk.cin_pad, pk.cout)):      # (that kernel activates while staging)
            # every input voxel feeds 27/8 output voxels on average: activate (and sum the skip) once instead of per tap
            src1, src2 = _apply_add(src1, src2), None
        raw, partials = _conv_t(src1, src2, dims_in, pk.get, pk.cin_pad, pk.cout, packed=pk, want_stats=self[1].training)
        D, H, W, C = raw.shape
        scale, shift, mean, invstd = _abn_stats(raw, D * H * W, self[1], update_running=self[1].training, partials=partials)
        return _Lazy(raw, scale, shift, (D, H, W, C), mean, invstd)

    def forward(self, x):
        ops._need_no_grad(x, *self.parameters(), op="ConvTranspose3d+ABN")
        buf, ld = _as_channel_last(x, self._packed.cin_pad)
        if ld != self._packed.cin_pad:
            raise RuntimeError("transposed conv input must be densely channel-last")
        D, H, W = x.shape[2:]
        out = _cl_view_to_ncdhw(_apply_add(self.lazy(buf, (D, H, W, ld))))
        _flush_nbt()
        return out


class CostRegNet(nn.Module):
    """reference models.py:725-769: 3-D U-Net 41->8->16->32->64->32->16->8 with skip sums."""

    def __init__(self, in_channels, norm_act=InPlaceABN):
        super().__init__()
        self.conv0 = ConvBnReLU3D(in_channels, 8, norm_act=norm_act)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2, norm_act=norm_act)
        self.conv2 = ConvBnReLU3D(16, 16, norm_act=norm_act)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2, norm_act=norm_act)
        self.conv4 = ConvBnReLU3D(32, 32, norm_act=norm_act)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2, norm_act=norm_act)
        self.conv6 = ConvBnReLU3D(64, 64, norm_act=norm_act)
        self.conv7 = _UpBlock(64, 32, norm_act)
        self.conv9 = _UpBlock(32, 16, norm_act)
        self.conv11 = _UpBlock(16, 8, norm_act)

    def _layers(self):
        return [self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6, self.conv7, self.conv9, self.conv11]

    def _run(self, x):
        """The lazily-activated U-Net; returns the 10 _Lazy layer outputs and the channel-last input."""
        if isinstance(x, _BlockedCost):
            D, H, W = x.dims
        else:
            _, C, D, H, W = x.shape
        if D % 8 or H % 8 or W % 8:
            raise RuntimeError(f"CostRegNet needs D,h,w divisible by 8 (three stride-2 stages), got {(D, H, W)}")
        # the hand-off says which encode this is: _Conv0Done = the guarded no-grad default, two fp16 planes = encoder_precision("fp16x3")
        f16 = "guarded" if isinstance(x, _Conv0Done) else ("plain" if isinstance(x, _BlockedCostH2) else None)
        if isinstance(x, _Conv0Done):          # conv0 already ran inside the guarded encode head (_plane_sweep, blocked="guarded")
            raw = x.buf
            scale, shift, mean, invstd = _abn_stats(raw, D * H * W, self.conv0.bn, update_running=self.conv0.bn.training,
                                                    partials=None if x.part is None else (x.part, x.nblk))
            c0 = _Lazy(raw, scale, shift, (D, H, W, 8), mean, invstd)
            buf, ld = None, 0
        elif isinstance(x, _BlockedCost16):
            pk = self.conv0._packed
            if x.n_ch != pk.cin:
                raise RuntimeError(f"CostRegNet: 16-bit cost volume has {x.n_ch} channels, conv0 expects {pk.cin}")
            raw = torch.empty((D, H, W, pk.cout), device=x.buf.device, dtype=torch.float32)
            lib = _lib.lib()
            want = self.conv0.bn.training and FUSED_ABN_STATS
            nblk = lib.mvsnerf_conv0_bf16_tiles(D, H, W)                   # the fp16x3 kernel has the same tiles / statistics slots
            part = torch.empty(nblk * 16, device=raw.device, dtype=torch.float32) if want else None
            if isinstance(x, _BlockedCostH2):
                check(lib.mvsnerf_conv0_f16x3_fwd(x.buf.data_ptr(), pk.cin, D, H, W, _get_f16x3_conv0(pk).data_ptr(), raw.data_ptr(),
                                                  0 if part is None else part.data_ptr(), stream_ptr()), "conv0_f16x3_fwd")
            else:
                check(lib.mvsnerf_conv0_bf16_fwd(x.buf.data_ptr(), pk.cin, D, H, W, pk.get_bf16_conv0().data_ptr(), raw.data_ptr(),
                                                 0 if part is None else part.data_ptr(), stream_ptr()), "conv0_bf16_fwd")
            scale, shift, mean, invstd = _abn_stats(raw, D * H * W, self.conv0.bn, update_running=self.conv0.bn.training,
                                                    partials=None if part is None else (part, nblk))
            c0 = _Lazy(raw, scale, shift, (D, H, W, pk.cout), mean, invstd)
            buf, ld = None, 0
        elif isinstance(x, _BlockedCost):
            pk = self.conv0._packed
            if x.cin_pad != pk.cin_pad:
                raise RuntimeError(f"CostRegNet: blocked cost volume has {x.cin_pad} channels, conv0 expects {pk.cin_pad}")
            raw = torch.empty((D, H, W, pk.cout), device=x.buf.device, dtype=torch.float32)
            lib = _lib.lib()
            if self.conv0.bn.training and FUSED_ABN_STATS:
                nblk = lib.mvsnerf_conv3d_c8_blocked_tiles(D, H, W)
                part = torch.empty(nblk * 16, device=raw.device, dtype=torch.float32)
                check(lib.mvsnerf_conv3d_c8_blocked_fwd_stats(x.buf.data_ptr(), pk.cin_pad, pk.cin, D, H, W, pk.get_c8().data_ptr(), raw.data_ptr(),
                                                              part.data_ptr(), stream_ptr()), "conv3d_c8_blocked_fwd_stats")
                partials = (part, nblk)
            else:
                check(lib.mvsnerf_conv3d_c8_blocked_fwd(x.buf.data_ptr(), pk.cin_pad, pk.cin, D, H, W, pk.get_c8().data_ptr(), raw.data_ptr(), stream_ptr()),
                      "conv3d_c8_blocked_fwd")
                partials = None
            scale, shift, mean, invstd = _abn_stats(raw, D * H * W, self.conv0.bn, update_running=self.conv0.bn.training, partials=partials)
            c0 = _Lazy(raw, scale, shift, (D, H, W, pk.cout), mean, invstd)
            buf, ld = None, 0
        else:
            buf, ld = _as_channel_last(x, self.conv0._packed_cin_pad())
            c0 = self.conv0.lazy(buf, (D, H, W, ld), ld)
        with _layer_precision(ENCODER_PRECISION == "bf16" and BF16_LAYERS), _layers_f16x3(f16):     # use_amp: conv1 ... conv11 on the bf16 matrix cores as well (csrc/conv3d_bf16.hip)
            # one guard_consume launch for the conv1 + conv2 pair when conv2 is sure to run guarded as well (same output size, both layers in batch-statistics
            # mode): a guard set by conv1 then also makes conv2 take its fp32 kernel, and conv2's sequence counts the event and re-arms
            _F16X3_CONSUME[0] = 0 if (f16 == "guarded" and self.conv1.bn.training and self.conv2.bn.training and FUSED_ABN_STATS) else 1
            try:
                c1 = self.conv1.lazy(c0, c0.dims, 8)
            finally:
                _F16X3_CONSUME[0] = 1
            c2 = self.conv2.lazy(c1, c1.dims, 16)
            c3 = self.conv3.lazy(c2, c2.dims, 16)
            c4 = self.conv4.lazy(c3, c3.dims, 32)
            c5 = self.conv5.lazy(c4, c4.dims, 32)
            c6 = self.conv6.lazy(c5, c5.dims, 64)
            u7 = self.conv7.lazy(c6, c6.dims)                       # x = conv4 + conv7(x)   (models.py:762)
            u9 = self.conv9.lazy(c4, c4.dims, src2=u7)              # x = conv2 + conv9(x)   (:764)
            u11 = self.conv11.lazy(c2, c2.dims, src2=u9)            # x = conv0 + conv11(x)  (:766)
        _flush_nbt()
        return (buf, ld), [c0, c1, c2, c3, c4, c5, c6, u7, u9, u11]

    def forward(self, x):
        """x: logical (1,Cin,D,h,w) cost volume (D,h,w divisible by 8).  Returns (1,8,D,h,w), channel-last memory."""
        if isinstance(x, _BlockedCost):          # internal no-grad hand-off from MVSNet.forward
            _, lz = self._run(x)
            return _neural_volume(lz[0], lz[9])
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _CostRegFunction.apply(x, self, *_costreg_params(self))
        _, lz = self._run(x)
        return _neural_volume(lz[0], lz[9])


def _grad_cl(g, C):
    """Upstream gradient of a logical (1,C,D,H,W) tensor -> contiguous channel-last (D,H,W,C) buffer."""
    v = g[0].permute(1, 2, 3, 0)
    return v if v.is_contiguous() else v.contiguous()


def _abn_bwd(lz, bn, g1, g2=None):
    """Train-mode InPlaceABN backward of one lazy layer.  g1 (+g2): grads w.r.t. its ACTIVATED output (channel-last).
    Returns (grad w.r.t. the raw conv output, d bn.weight, d bn.bias)."""
    if not bn.training:
        raise RuntimeError("InPlaceABN backward is implemented for train-mode (batch-statistics) layers only - the reference "
                           "keeps MVSNet in .train() (train_mvs_nerf_pl.py:182)")
    D, H, W, C = lz.dims
    dev = lz.x.device
    gx = torch.empty((D, H, W, C), device=dev, dtype=torch.float32)
    # two tensors of their own, not two rows of one: autograd's AccumulateGrad takes a gradient over as `.grad` only when it owns its storage,
    # a view is CLONED - 36 device copies per training step (18 InPlaceABN layers) showed up as __amd_rocclr_copyBuffer in the kernel trace
    gbw, gbb = torch.empty(C, device=dev, dtype=torch.float32), torch.empty(C, device=dev, dtype=torch.float32)
    ws = torch.empty(_lib.lib().mvsnerf_abn_workspace_floats(C), device=dev, dtype=torch.float32)
    check(_lib.lib().mvsnerf_abn_bwd(lz.x.data_ptr(), D * H * W, C, dev_f32(bn.weight.detach(), "bn.weight"), lz.scale.data_ptr(), lz.shift.data_ptr(),
                                     lz.mean.data_ptr(), lz.invstd.data_ptr(), g1.data_ptr(), 0 if g2 is None else g2.data_ptr(),
                                     gx.data_ptr(), gbw.data_ptr(), gbb.data_ptr(), ws.data_ptr(), stream_ptr()), "abn_bwd")
    return gx, gbw, gbb


def _wgrad(G1, G2, A, X1, X2, B, ldx, g_dims, x_dims, stride, shape, sums=None):
    """gW[a][b][tap] = sum_o G[o][a] X[o*stride-1+tap][b]  (see mvsnerf_conv3d_wgrad).  sums: a _PartialSums that finishes gw later."""
    lib = _lib.lib()
    dev = (G1.x if isinstance(G1, _Lazy) else G1).device
    gw = torch.empty(shape, device=dev, dtype=torch.float32)
    if _LAYER_BF16[0] and BF16_WGRAD:
        n_parts = lib.mvsnerf_conv_wgrad_bf16_parts(A, B, g_dims[0], g_dims[1], g_dims[2], 3, 3, stride)
        if n_parts > 0:             # use_amp: bf16 operands, v_mfma_f32_16x16x32_bf16 (csrc/wgrad_bf16.hip)
            ws = torch.empty(lib.mvsnerf_conv_wgrad_bf16_workspace_floats(A, B, 3, 3) if sums is None else n_parts * gw.numel(), device=dev, dtype=torch.float32)
            check(lib.mvsnerf_conv_wgrad_bf16(*_ptrs(G1), *_ptrs(G2), A, *_ptrs(X1), *_ptrs(X2), B, ldx, g_dims[0], g_dims[1], g_dims[2],
                                              x_dims[0], x_dims[1], x_dims[2], 3, 3, stride, 0 if sums is not None else gw.data_ptr(), ws.data_ptr(),
                                              stream_ptr()), "conv_wgrad_bf16")
            if sums is not None:
                sums.add(ws, n_parts, gw)
            return gw
    ws = torch.empty(lib.mvsnerf_conv3d_wgrad_workspace_floats(A, B), device=dev, dtype=torch.float32)
    check(lib.mvsnerf_conv3d_wgrad(*_ptrs(G1), *_ptrs(G2), A, *_ptrs(X1), *_ptrs(X2), B, ldx, g_dims[0], g_dims[1], g_dims[2],
                                   x_dims[0], x_dims[1], x_dims[2], stride, 0 if sums is not None else gw.data_ptr(), ws.data_ptr(), stream_ptr()),
          "conv3d_wgrad")
    if sums is not None:
        sums.add(ws, lib.mvsnerf_conv3d_wgrad_parts(A, B, g_dims[0], g_dims[1], g_dims[2], stride, int(X2 is not None)), gw)
    return gw


def _costreg_backward(net, lz, g_out, conv0_grads, sums=None):
    """Backward of CostRegNet._run + the output sum.  g_out: gradient of the (1,8,D,h,w) result.  conv0_grads(gx) -> (gw, g_input):
    conv0's weight gradient and whatever the caller needs upstream of it, from the gradient gx of conv0's raw output.
    Returns (g_input, [gw, g bn.weight, g bn.bias] x 10 in _layers() order)."""
    c0, c1, c2, c3, c4, c5, c6, u7, u9, u11 = lz
    L = net._layers()
    g = _grad_cl(g_out, 8)                                  # grad w.r.t. A(c0) + A(u11)
    grads = {}

    def up_block(i, lay, out_lz, in1, in2, g_act1, g_act2=None):
        """ConvTranspose3d+ABN `lay` (output out_lz, input A(in1)+A(in2)): returns grad w.r.t. its activated input."""
        gx, gbw, gbb = _abn_bwd(out_lz, lay[1], g_act1, g_act2)
        pk = lay._packed
        gw = _wgrad(in1, in2, pk.cin, gx, None, pk.cout, pk.cout, in1.dims[:3], out_lz.dims[:3], 2, tuple(lay[0].weight.shape), sums)
        g_in = _conv(gx, None, out_lz.dims, pk.cout, lambda: pk.get("dgrad"), pk.cout, pk.cin, 2, packed=pk, mode="dgrad")       # data grad = stride-2 conv
        grads[i] = (gw, gbw, gbb)
        return g_in

    def conv_block(i, lay, out_lz, in1, in_dims, in_ld, g_act1, g_act2=None):
        gx, gbw, gbb = _abn_bwd(out_lz, lay.bn, g_act1, g_act2)
        pk = lay._packed
        gw = _wgrad(gx, None, pk.cout, in1, None, pk.cin, in_ld, out_lz.dims[:3], in_dims[:3], lay.stride, tuple(lay.conv.weight.shape), sums)
        grads[i] = (gw, gbw, gbb)
        if lay.stride == 1:
            return _conv(gx, None, out_lz.dims, pk.cout, lambda: pk.get("dgrad"), pk.cout, pk.cin_pad, 1, packed=pk, mode="dgrad")
        return _conv_t(gx, None, out_lz.dims, lambda: pk.get("dgrad"), pk.cout, pk.cin_pad, packed=pk, mode="dgrad")

    g_u9c2 = up_block(9, L[9], u11, c2, u9, g)               # conv11: grad w.r.t. A(c2)+A(u9)
    g_u7c4 = up_block(8, L[8], u9, c4, u7, g_u9c2)           # conv9:  grad w.r.t. A(c4)+A(u7)
    g_c6 = up_block(7, L[7], u7, c6, None, g_u7c4)           # conv7:  grad w.r.t. A(c6)
    g_c5 = conv_block(6, L[6], c6, c5, c5.dims, 64, g_c6)
    g_c4 = conv_block(5, L[5], c5, c4, c4.dims, 32, g_c5)
    g_c3 = conv_block(4, L[4], c4, c3, c3.dims, 32, g_u7c4, g_c4)     # A(c4) feeds conv5 and the conv9 skip
    g_c2 = conv_block(3, L[3], c3, c2, c2.dims, 16, g_c3)
    g_c1 = conv_block(2, L[2], c2, c1, c1.dims, 16, g_u9c2, g_c2)     # A(c2) feeds conv3 and the conv11 skip
    g_c0 = conv_block(1, L[1], c1, c0, c0.dims, 8, g_c1)
    gx0, gbw0, gbb0 = _abn_bwd(c0, L[0].bn, g, g_c0)                   # A(c0) feeds conv1 and the output sum
    gw0, g_in = conv0_grads(gx0)
    grads[0] = (gw0, gbw0, gbb0)
    if sums is not None:
        sums.flush()
    flat = []
    for i in range(10):
        flat += list(grads[i])
    return g_in, flat


def _costreg_params(net):
    params = []
    for l in net._layers():
        conv, bn = (l.conv, l.bn) if isinstance(l, ConvBnReLU3D) else (l[0], l[1])
        params += [conv.weight, bn.weight, bn.bias]
    return params


class _CostRegFunction(torch.autograd.Function):
    """CostRegNet with gradients to the cost volume, the 10 conv weights and the 10 ABN weight/bias pairs."""

    @staticmethod
    def forward(ctx, x, net, *params):
        (buf, ld), lz = net._run(x)
        ctx.net, ctx.buf, ctx.ld, ctx.lz, ctx.xshape = net, buf, ld, lz, tuple(x.shape)
        ctx.layers_bf16 = ENCODER_PRECISION == "bf16" and BF16_LAYERS
        return _neural_volume(lz[0], lz[9])

    @staticmethod
    def backward(ctx, g_out):
        net, lz, buf, ld = ctx.net, ctx.lz, ctx.buf, ctx.ld
        lay, c0 = net.conv0, lz[0]
        pk = lay._packed
        D, H, W = c0.dims[:3]

        sums = _PartialSums()

        def conv0_grads(gx):
            gw = _wgrad(gx, None, pk.cout, buf, None, pk.cin, ld, (D, H, W), (D, H, W), 1, tuple(lay.conv.weight.shape), sums)
            if not ctx.needs_input_grad[0]:
                return gw, None
            return gw, _conv(gx, None, c0.dims, pk.cout, lambda: pk.get("dgrad"), pk.cout, pk.cin_pad, 1, packed=pk, mode="dgrad")

        with _layer_precision(ctx.layers_bf16):
            g_cost, flat = _costreg_backward(net, lz, g_out, conv0_grads, sums)
        return (None if g_cost is None else _cl_view_to_ncdhw(g_cost, ctx.xshape[1]), None, *flat)


# ------------------------------------------------------------------ MVSNet
def homo_warp(src_feat, proj_mat, depth_values, src_grid=None, pad=0):
    """reference utils.py:580-630.  src_feat (1,C,H,W); proj_mat (1,3,4); depth_values (1,D).
    Returns warped (1,C,D,H+2p,W+2p) and src_grid (1,D,W+2p,H+2p,2) (the reference's axis naming; flat order d,y,x)."""
    ops._need_no_grad(src_feat, op="homo_warp")
    B, C, H, W = src_feat.shape
    if B != 1:
        raise RuntimeError("homo_warp: batch must be 1")
    Hp, Wp = H + 2 * pad, W + 2 * pad
    dev = src_feat.device
    keep = ops._Keep()          # contiguous copies (if any were needed) stay alive until the launch is issued
    if src_grid is None:
        D = depth_values.shape[1]
        grid_in, grid_out = 0, torch.empty((1, D, Wp, Hp, 2), device=dev, dtype=torch.float32)
        proj_p, dep_p = keep(proj_mat[0], "proj_mat"), keep(depth_values[0], "depth_values")
    else:
        D = src_grid.shape[1]
        grid_in, grid_out = keep(src_grid, "src_grid"), src_grid
        proj_p = dep_p = 0
    warped = torch.empty((1, C, D, Hp, Wp), device=dev, dtype=torch.float32)
    check(_lib.lib().mvsnerf_homo_warp_fwd(keep(src_feat, "src_feat"), proj_p, dep_p, grid_in, C, H, W, D, pad,
                                           warped.data_ptr(), 0 if src_grid is not None else grid_out.data_ptr(), stream_ptr()), "homo_warp_fwd")
    return warped, grid_out


class _BlockedCost:
    """Cost volume in channel blocks of four, buf[CP//4][D*H*W][4] (mvsnerf_planesweep_costvar_blocked_fwd): the internal hand-off
    between the plane sweep and the matrix-core conv0 on the no-grad path.  Never handed to callers."""
    __slots__ = ("buf", "n_ch", "cin_pad", "dims")

    def __init__(self, buf, n_ch, cin_pad, dims):
        self.buf, self.n_ch, self.cin_pad, self.dims = buf, n_ch, cin_pad, dims


class _BlockedCost16(_BlockedCost):
    """The same hand-off rounded to bf16, in channel blocks of sixteen: buf[ceil(n_ch/16)][D*H*W][16] torch.bfloat16
    (mvsnerf_planesweep_costvar_bf16_fwd -> the bf16 conv0 kernels of csrc/conv_bf16.hip; `use_amp` training)."""
    __slots__ = ()


def _inference_hand_off():
    """`blocked` of the no-grad plane sweep -> conv0 hand-off for the current encoder precision."""
    return {"bf16": "bf16", "fp16x3": "fp16x2", "auto": "guarded"}.get(ENCODER_PRECISION, True)


class _BlockedCostH2(_BlockedCost16):
    """Two fp16 pieces of cost / 16 in that layout, hi plane then lo plane: buf[2][ceil(n_ch/16)][D*H*W][16] torch.float16
    (mvsnerf_planesweep_costvar_f16x2_fwd -> the fp32-grade fp16 conv0 of csrc/conv_f16x3.hip; encoder_precision("fp16x3"), inference)."""
    __slots__ = ()


class _Conv0Done(_BlockedCost):
    """Hand-off of the guarded encode head (mvsnerf_sweep_conv0_guarded_fwd): the plane sweep AND conv0 have run; `buf` is conv0's raw output
    (D,H,W,8), `part` its InPlaceABN partial sums (or None in eval mode), `nblk` their slot count."""
    __slots__ = ("part", "nblk")

    def __init__(self, raw, n_ch, dims, part, nblk):
        super().__init__(raw, n_ch, 0, dims)
        self.part, self.nblk = part, nblk


def _plane_sweep(imgs, feats, proj_mats, depth_values, pad, with_img, blocked=False, conv0=None):
    """One-pass plane sweep (homo_warp + cost variance [+ warped thumbnails]).  Returns (cost view, masks, saved);
    blocked=True: the cost volume comes back as a _BlockedCost instead of a logical NCDHW view."""
    B, V, C, H, W = feats.shape
    dev = feats.device
    lib = _lib.lib()
    D = depth_values.shape[1]
    Hp, Wp = H + 2 * pad, W + 2 * pad
    feats_cl, f_ld = _images_channel_last(feats[0].detach(), C)       # zero-copy for FeatureNet's channel-last output
    if f_ld != C:
        feats_cl = feats_cl[..., :C].contiguous()
    imgs_cl_p = 0
    if with_img:
        Hi, Wi = imgs.shape[-2:]
        imgs_cl = torch.empty((V, H, W, 4), device=dev, dtype=torch.float32)                     # models.py:859, written channel-last in the same launch
        check(lib.mvsnerf_resize_bilinear_nhwc4(dev_f32(imgs[0].contiguous(), "imgs"), imgs_cl.data_ptr(), V, Hi, Wi, H, W, stream_ptr()), "resize_bilinear_nhwc4")
        imgs_cl_p = imgs_cl.data_ptr()
    n_ch = (3 * V if with_img else 0) + C
    CP = (n_ch + 3) // 4 * 4
    masks = torch.empty((V, D, Hp, Wp) if with_img else (1, D, Hp, Wp), device=dev, dtype=torch.float32)
    proj = proj_mats[0].detach().contiguous()
    depth = depth_values[0].detach().contiguous()
    if blocked == "guarded":
        # no-grad default: two-piece fp16 sweep -> fp16x3 conv0 -> {fp32 sweep, fp32-MFMA conv0, statistics} predicated on the guard word, one call
        from . import ops as _ops
        pk = conv0._packed
        if not with_img or pk.cin != n_ch or pk.cout != 8:
            raise RuntimeError(f"guarded encode head: conv0 expects {pk.cin} channels, the sweep produces {n_ch}")
        nb16 = (n_ch + 15) // 16
        nvox = D * Hp * Wp
        # one buffer for both hand-offs: the fp32 blocks (CP * 4 B per voxel) are only written after the fp16x3 conv0 has consumed the fp16
        # planes (nb16 * 64 B per voxel), in stream order
        scratch = torch.empty(max(2 * nb16 * 16 * 2, CP * 4) * nvox, device=dev, dtype=torch.uint8)
        raw = torch.empty((D, Hp, Wp, 8), device=dev, dtype=torch.float32)
        want = conv0.bn.training and FUSED_ABN_STATS
        nblk = lib.mvsnerf_conv0_bf16_tiles(D, Hp, Wp)
        part = torch.empty(nblk * 16, device=dev, dtype=torch.float32) if want else None
        a = _lib.SweepConv0Args(
            feats_cl=feats_cl.data_ptr(), imgs_cl=imgs_cl_p, proj=dev_f32(proj, "proj_mats"), depth=dev_f32(depth, "depth_values"),
            V=V, H=H, W=W, D=D, pad=pad, CP=CP, masks=masks.data_ptr(), cost16x2=scratch.data_ptr(), cost32=scratch.data_ptr(),
            w_f16x3=_get_f16x3_conv0(pk).data_ptr(), w_c8=pk.get_c8().data_ptr(), Cin=n_ch, out=raw.data_ptr(),
            stats_part=0 if part is None else part.data_ptr(), guard=_ops.guard_words(dev).data_ptr())
        check(lib.mvsnerf_sweep_conv0_guarded_fwd(ctypes.byref(a), stream_ptr()), "sweep_conv0_guarded_fwd")
        return _Conv0Done(raw, n_ch, (D, Hp, Wp), part, nblk), masks.unsqueeze(0), (feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch))
    if blocked == "fp16x2":
        nb16 = (n_ch + 15) // 16
        cost = torch.empty((2, nb16, D * Hp * Wp, 16), device=dev, dtype=torch.float16)
        check(lib.mvsnerf_planesweep_costvar_f16x2_fwd(feats_cl.data_ptr(), imgs_cl_p, dev_f32(proj, "proj_mats"), dev_f32(depth, "depth_values"),
                                                       V, C, H, W, D, pad, cost.data_ptr(), CP, masks.data_ptr(), int(with_img), stream_ptr()),
              "planesweep_costvar_f16x2_fwd")
        return _BlockedCostH2(cost, n_ch, nb16 * 16, (D, Hp, Wp)), masks.unsqueeze(0), (feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch))
    if blocked == "bf16":
        nb16 = (n_ch + 15) // 16
        cost = torch.empty((nb16, D * Hp * Wp, 16), device=dev, dtype=torch.bfloat16)
        check(lib.mvsnerf_planesweep_costvar_bf16_fwd(feats_cl.data_ptr(), imgs_cl_p, dev_f32(proj, "proj_mats"), dev_f32(depth, "depth_values"),
                                                      V, C, H, W, D, pad, cost.data_ptr(), CP, masks.data_ptr(), int(with_img), stream_ptr()),
              "planesweep_costvar_bf16_fwd")
        return _BlockedCost16(cost, n_ch, nb16 * 16, (D, Hp, Wp)), masks.unsqueeze(0), (feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch))
    if blocked:
        cost = torch.empty((CP // 4, D * Hp * Wp, 4), device=dev, dtype=torch.float32)
        check(lib.mvsnerf_planesweep_costvar_blocked_fwd(feats_cl.data_ptr(), imgs_cl_p, dev_f32(proj, "proj_mats"), dev_f32(depth, "depth_values"),
                                                         V, C, H, W, D, pad, cost.data_ptr(), CP, masks.data_ptr(), int(with_img), stream_ptr()),
              "planesweep_costvar_blocked_fwd")
        return _BlockedCost(cost, n_ch, CP, (D, Hp, Wp)), masks.unsqueeze(0), (feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch))
    cost = torch.empty((D, Hp, Wp, CP), device=dev, dtype=torch.float32)
    check(lib.mvsnerf_planesweep_costvar_fwd(feats_cl.data_ptr(), imgs_cl_p, dev_f32(proj, "proj_mats"), dev_f32(depth, "depth_values"),
                                             V, C, H, W, D, pad, cost.data_ptr(), CP, masks.data_ptr(), int(with_img), stream_ptr()),
          "planesweep_costvar_fwd")
    return _cl_view_to_ncdhw(cost, n_ch), masks.unsqueeze(0), (feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch))


PSW_BWD_DETERMINISTIC = False   # True: the plane sweep's feature gradient through 64-bit fixed-point accumulators (integer atomics commute: two
                                # runs, or N ranks against one, agree bit for bit; ~0.1 ms slower).  Default: float atomics.


def _planesweep_bwd(feats_cl, proj, depth, V, C, H, W, D, pad, g_cost, ld, with_img):
    """d cost volume (variance channels) -> d feats_cl (V, H, W, C); reference: autograd through models.py:839-893 (homo_warp + variance)."""
    lib = _lib.lib()
    g_feats = torch.zeros((V, H, W, C), device=feats_cl.device, dtype=torch.float32)
    if PSW_BWD_DETERMINISTIC:
        ws = torch.zeros(lib.mvsnerf_planesweep_costvar_bwd_det_workspace_words(V, C, H, W), device=feats_cl.device, dtype=torch.int64)
        check(lib.mvsnerf_planesweep_costvar_bwd_det(feats_cl.data_ptr(), proj.data_ptr(), depth.data_ptr(), V, C, H, W, D, pad,
                                                     g_cost.data_ptr(), ld, with_img, g_feats.data_ptr(), ws.data_ptr(), stream_ptr()),
              "planesweep_costvar_bwd_det")
    else:
        check(lib.mvsnerf_planesweep_costvar_bwd(feats_cl.data_ptr(), proj.data_ptr(), depth.data_ptr(), V, C, H, W, D, pad,
                                                 g_cost.data_ptr(), ld, with_img, g_feats.data_ptr(), stream_ptr()), "planesweep_costvar_bwd")
    return g_feats


class _PlaneSweepFunction(torch.autograd.Function):
    """Plane sweep with the gradient of the variance channels w.r.t. the source feature maps (bilinear scatter)."""

    @staticmethod
    def forward(ctx, feats, imgs, proj_mats, depth_values, pad, with_img):
        cost, masks, saved = _plane_sweep(imgs, feats, proj_mats, depth_values, pad, with_img)
        ctx.saved, ctx.with_img = saved, with_img
        ctx.mark_non_differentiable(masks)
        return cost, masks

    @staticmethod
    def backward(ctx, g_cost, g_masks):
        feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch) = ctx.saved
        buf, ld = _as_channel_last(g_cost, CP)
        g_feats = _planesweep_bwd(feats_cl, proj, depth, V, C, H, W, D, pad, buf, ld, int(ctx.with_img))
        return g_feats.permute(0, 3, 1, 2).unsqueeze(0), None, None, None, None, None


class _SweepRegFunction(torch.autograd.Function):
    """Training form of MVSNet.forward's volume branch (models.py:922-930): plane sweep -> CostRegNet as ONE autograd node, with the
    cost volume in channel blocks of four between them (never in NCDHW / channel-last form).  conv0 and its weight gradient run on the
    matrix cores from the blocked volume; its data gradient is computed for the 32 variance channels only - the warped thumbnails
    (channels 0..3V-1) have no parameters upstream."""

    @staticmethod
    def forward(ctx, feats, imgs, proj_mats, depth_values, pad, net, *params):
        cost, _, saved = _plane_sweep(imgs, feats, proj_mats, depth_values, pad, True, blocked="bf16" if ENCODER_PRECISION == "bf16" else True)
        _, lz = net._run(cost)
        ctx.net, ctx.cost, ctx.lz, ctx.saved = net, cost, lz, saved
        ctx.layers_bf16 = ENCODER_PRECISION == "bf16" and BF16_LAYERS
        return _neural_volume(lz[0], lz[9])

    @staticmethod
    def backward(ctx, g_out):
        net, cost, lz = ctx.net, ctx.cost, ctx.lz
        feats_cl, proj, depth, (V, C, H, W, D, pad, CP, n_ch) = ctx.saved
        lay, c0 = net.conv0, lz[0]
        pk = lay._packed
        Dv, Hv, Wv = cost.dims
        lib = _lib.lib()

        sums = _PartialSums()

        def conv0_grads(gx):
            gw = torch.empty(tuple(lay.conv.weight.shape), device=gx.device, dtype=torch.float32)
            ws = torch.empty(lib.mvsnerf_conv3d_wgrad_workspace_floats(8, pk.cin), device=gx.device, dtype=torch.float32)
            if isinstance(cost, _BlockedCost16):          # use_amp: both conv0 gradients on the bf16 matrix cores (gx is rounded on the way in)
                check(lib.mvsnerf_conv0_bf16_wgrad(cost.buf.data_ptr(), pk.cin, Dv, Hv, Wv, gx.data_ptr(), 0, ws.data_ptr(), stream_ptr()), "conv0_bf16_wgrad")
                sums.add(ws, lib.mvsnerf_conv0_bf16_wgrad_parts(Dv, Hv, Wv), gw)
                if not ctx.needs_input_grad[0]:
                    return gw, None
                g_var = torch.empty((Dv, Hv, Wv, C), device=gx.device, dtype=torch.float32)
                check(lib.mvsnerf_conv0_bf16_dgrad(gx.data_ptr(), Dv, Hv, Wv, pk.get_bf16_conv0(dgrad=(3 * V, C)).data_ptr(), C, g_var.data_ptr(),
                                                   stream_ptr()), "conv0_bf16_dgrad")
                return gw, g_var
            check(lib.mvsnerf_conv3d_c8_blocked_wgrad(cost.buf.data_ptr(), pk.cin_pad, pk.cin, Dv, Hv, Wv, gx.data_ptr(), 0,
                                                      ws.data_ptr(), stream_ptr()), "conv3d_c8_blocked_wgrad")
            sums.add(ws, lib.mvsnerf_conv3d_c8_blocked_wgrad_parts(pk.cin_pad, pk.cin, Dv, Hv, Wv), gw)
            if not ctx.needs_input_grad[0]:
                return gw, None
            return gw, _conv(gx, None, c0.dims, pk.cout, pk.get_dgrad_slice(3 * V, C), pk.cout, C, 1)     # d cost[variance channels]

        with _layer_precision(ctx.layers_bf16):
            g_var, flat = _costreg_backward(net, lz, g_out, conv0_grads, sums)
        g_feats = None
        if g_var is not None:
            g_feats = _planesweep_bwd(feats_cl, proj, depth, V, C, H, W, D, pad, g_var, C, 0).permute(0, 3, 1, 2).unsqueeze(0)
        return (g_feats, None, None, None, None, None, *flat)


class MVSNet(nn.Module):
    """reference models.py:771-932."""

    def __init__(self, num_groups=1, norm_act=InPlaceABN, levels=1, n_views=3):
        super().__init__()
        self.levels = levels
        self.n_depths = [128, 32, 8]
        self.G = num_groups
        self.feature = FeatureNet()
        self.N_importance = 0
        self.chunk = 1024
        # the reference hard-wires 3 source views (32 + 9 channels, models.py:785); n_views is an extension for
        # BASELINE config 4 (5 views => 47 input channels, no shipped checkpoint fits)
        self.cost_reg_2 = CostRegNet(32 + 3 * n_views, norm_act)
        self.D = 128          # number of depth planes (hard-coded `D = 128` at models.py:914; settable here for config 1)
        self._t_vals = {}     # linspace(0, 1, D) per (device, dtype, D)
        self._pack_log = []   # weight re-layouts the last forward/backward asked for (see _PackBatch)
        for m in self.modules():
            for name in ("_packed", "_top_packed"):
                pk = getattr(m, name, None)
                if pk is not None:
                    pk.log = self._pack_log

    def prepack(self):
        """Issue every weight re-layout the previous step needed (and that is stale now) in one launch."""
        if self._pack_log:
            with _PackBatch():
                for pk, method, args in list(self._pack_log):
                    getattr(pk, method)(*args)

    def invalidate_packed(self):
        """Drop every re-packed convolution weight (see MVSNeRF.invalidate_packed: needed after writes through `.data`)."""
        for m in self.modules():
            pk = getattr(m, "_packed", None)
            if pk is not None and hasattr(pk, "cache"):
                pk.cache.clear()
        tp = getattr(self.feature, "_top_packed", None)
        if tp is not None:
            tp.cache.clear()

    def _depth_values(self, near_far, imgs, lindisp):
        """models.py:903-906.  The linear case with near_far on the device is ONE launch (mvsnerf_depth_values: the same four roundings as linspace / rsub / mul / mul /
        add, without their five launches); t_vals is cached per (device, dtype, D)."""
        key = (imgs.device, imgs.dtype, self.D)
        t_vals = self._t_vals.get(key)
        if t_vals is None:
            t_vals = self._t_vals[key] = torch.linspace(0.0, 1.0, steps=self.D, device=imgs.device, dtype=imgs.dtype)
        if (not lindisp and torch.is_tensor(near_far) and near_far.is_cuda and near_far.dtype == torch.float32 and near_far.numel() == 2 and near_far.is_contiguous()
                and imgs.dtype == torch.float32 and not near_far.requires_grad):
            out = torch.empty(self.D, device=imgs.device, dtype=torch.float32)
            check(_lib.lib().mvsnerf_depth_values(t_vals.data_ptr(), near_far.data_ptr(), self.D, out.data_ptr(), stream_ptr()), "depth_values")
            return out
        near, far = near_far
        return (near * (1.0 - t_vals) + far * t_vals) if not lindisp else 1.0 / (1.0 / near * (1.0 - t_vals) + 1.0 / far * t_vals)

    def _sweep(self, imgs, feats, proj_mats, depth_values, pad, with_img, blocked=False):
        if feats.shape[0] != 1:
            raise RuntimeError("MVSNet: batch size must be 1 (the reference assumes it, models.py:916)")
        if torch.is_grad_enabled() and feats.requires_grad:
            return _PlaneSweepFunction.apply(feats, imgs, proj_mats, depth_values, pad, with_img)
        cost, masks, _ = _plane_sweep(imgs, feats, proj_mats, depth_values, pad, with_img, blocked=blocked,
                                      conv0=self.cost_reg_2.conv0 if blocked == "guarded" else None)
        return cost, masks

    def build_volume_costvar(self, feats, proj_mats, depth_values, pad=0):
        """reference models.py:787-837 -> (variance (B,32,D,h,w), in_masks (B,1,D,h,w) view count)."""
        return self._sweep(None, feats, proj_mats, depth_values, pad, False)

    def build_volume_costvar_img(self, imgs, feats, proj_mats, depth_values, pad=0):
        """reference models.py:839-893 -> (img_feat (B,3V+32,D,h,w), in_masks (B,V,D,h,w)).
        The border of channels 0:3 (uninitialised in the reference, models.py:858) is defined as 0."""
        return self._sweep(imgs, feats, proj_mats, depth_values, pad, True)

    def forward(self, imgs, proj_mats, near_far, pad=0, return_color=False, lindisp=False):
        """reference models.py:895-932.  imgs (B,V,3,H,W) normalised; proj_mats (B,V,3,4); near_far (2,)."""
        B, V, _, H, W = imgs.shape
        self.prepack()
        _NBT_DEFER[0] += 1
        try:
            feats = self.feature(imgs.reshape(B * V, 3, H, W))
        finally:
            _NBT_DEFER[0] -= 1
        try:
            return self._forward_behind_features(imgs, feats, proj_mats, near_far, pad, return_color, lindisp)
        finally:
            _flush_nbt()                 # (a no-op when CostRegNet's own flush took FeatureNet's counters along)

    def _forward_behind_features(self, imgs, feats, proj_mats, near_far, pad, return_color, lindisp):
        B, V, _, H, W = imgs.shape
        feats_l = feats.view(B, V, *feats.shape[1:])
        depth_values = self._depth_values(near_far, imgs, lindisp).unsqueeze(0)
        # inference (no gradient anywhere): the cost volume goes to conv0 in channel blocks of 8 and never takes its NCDHW form
        fast = (BLOCKED_COST and not return_color and not feats_l.requires_grad
                and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.cost_reg_2.parameters()))
                and (32 + 3 * V + 3) // 4 * 4 in _BLOCKED_CIN)
        if fast:
            cost, _ = self._sweep(imgs, feats_l, proj_mats, depth_values, pad, True, blocked=_inference_hand_off())
            return self.cost_reg_2(cost), feats_l, depth_values
        if BLOCKED_COST and not return_color and torch.is_grad_enabled() and (32 + 3 * V + 3) // 4 * 4 in _BLOCKED_CIN and B == 1:
            # training: the same blocked hand-off inside one autograd node
            vol = _SweepRegFunction.apply(feats_l, imgs, proj_mats, depth_values, pad, self.cost_reg_2, *_costreg_params(self.cost_reg_2))
            return vol, feats_l, depth_values
        volume_feat, in_masks = self.build_volume_costvar_img(imgs, feats_l, proj_mats, depth_values, pad=pad)
        if return_color:
            feats_l = torch.cat((volume_feat[:, :V * 3].reshape(B, V, 3, *volume_feat.shape[2:]), in_masks.unsqueeze(2)), dim=2)
        volume_feat = self.cost_reg_2(volume_feat)
        return volume_feat, feats_l, depth_values


def bench_encode(rig, dev, pad, iters=6):
    """Used by bench.py: build the neural volume of the synthetic scene; returns (volume, per-stage ms)."""
    import numpy as np
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mvsnerf_v0_weights.npz"))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")}
    net = MVSNet().to(dev)
    net.load_state_dict(sd)
    net.train()                                   # the reference keeps MVSNet in train mode at inference (train_mvs_nerf_pl.py:182)
    imgs = rig["images"][:, :3].to(dev)
    proj = rig["proj_mats"][:, :3].to(dev)
    nf = rig["near_fars"][0, 0].to(dev)
    rec = []
    with torch.no_grad():
        for it in range(iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            B, V, _, H, W = imgs.shape
            feats = net.feature(imgs.reshape(B * V, 3, H, W))
            feats_l = feats.view(B, V, *feats.shape[1:])
            torch.cuda.synchronize(); t1 = time.perf_counter()
            t_vals = torch.linspace(0.0, 1.0, steps=net.D, device=dev)
            dv = (nf[0] * (1.0 - t_vals) + nf[1] * t_vals).unsqueeze(0)
            if BLOCKED_COST:
                cost, _ = net._sweep(imgs, feats_l, proj, dv, pad, True, blocked=_inference_hand_off())   # what MVSNet.forward does without gradients
            else:
                cost, _ = net.build_volume_costvar_img(imgs, feats_l, proj, dv, pad=pad)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            vol = net.cost_reg_2(cost)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            rec.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    # median over the iterations after the first (the first pays lazy code-object loads; a single iteration can also hit a
    # caching-allocator refill for the 825 MB cost volume)
    med = [sorted(r[i] for r in rec[1:] or rec)[len(rec[1:] or rec) // 2] for i in range(4)]
    times = {"feature_net": round(med[0] * 1e3, 3), "planesweep_costvar": round(med[1] * 1e3, 3), "cost_reg_net": round(med[2] * 1e3, 3),
             "total": round(med[3] * 1e3, 3), "iters": len(rec)}
    # the product call, free-running: MVSNet.forward back to back, host synchronisation only around the whole batch (the per-stage numbers
    # above stop the queue three times per encode, and the first launches after each stop start late)
    with torch.no_grad():
        net(imgs, proj, nf, pad=pad)
        single = []
        for _ in range(iters):                                       # one isolated call: launch to completion, nothing overlapped
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net(imgs, proj, nf, pad=pad)
            torch.cuda.synchronize(); single.append(time.perf_counter() - t0)
        times["forward_single_call"] = round(sorted(single)[len(single) // 2] * 1e3, 3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            vol_f = net(imgs, proj, nf, pad=pad)[0]
        torch.cuda.synchronize()
        times["forward_free_running"] = round((time.perf_counter() - t0) / iters * 1e3, 3)
        times["note"] = ("feature_net / planesweep_costvar / cost_reg_net / total: stage by stage with a host synchronisation after each stage; in the guarded default "
                         "the plane-sweep stage is mvsnerf_sweep_conv0_guarded_fwd and therefore INCLUDES conv0 (and cost_reg_net starts at conv0's statistics) "
                         "(comparable with earlier rounds); forward_single_call: MVSNet.forward, the product call, one isolated call; "
                         "forward_free_running: the same call back to back")
        # the same forward captured ONCE into a hipGraph and replayed (torch.cuda.CUDAGraph: every launch of the encode - ~80 kernels -
        # is issued by the GPU's own scheduler, no Python / ctypes launch path in between)
        try:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    vol_g = net(imgs, proj, nf, pad=pad)[0]
            torch.cuda.current_stream().wait_stream(side)
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                g.replay()
            torch.cuda.synchronize()
            times["forward_hipgraph_replay"] = round((time.perf_counter() - t0) / iters * 1e3, 3)
            times["hipgraph_volume_equals_eager"] = bool(torch.equal(vol_g, vol_f))
        except Exception as e:                                       # capture is an extra: the eager numbers stand on their own
            times["forward_hipgraph_replay"] = None
            times["hipgraph_error"] = str(e)[:200]
    return vol, times
