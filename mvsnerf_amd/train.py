"""Training entry points of the reference on the HIP hot path, without pytorch_lightning (not in this image).

`MVSSystem` mirrors `train_mvs_nerf_pl.py:34-288`: same constructor argument (`args` from opt.py), same
`decode_batch / unpreprocess / configure_optimizers / training_step(batch, batch_nb) -> {'loss': loss} / save_ckpt`
methods, same batch dict schema (data/dtu.py:199-211 collated with B=1), same `self.log` keys, same checkpoint keys.
A real `pytorch_lightning.LightningModule` can be mixed in unchanged; `_ModuleShim` provides the three attributes
the step uses (`log`, `global_step`, `device`) when Lightning is absent.

Multi-GPU (one process per GPU, gradients averaged by ONE flat-buffer all-reduce -
mvsnerf_amd.distributed.FlatGradAllReduce, RCCL over xGMI on GPUs), two modes selected by `args.dp_mode`:
  "scene" (DEFAULT for generalizable training; north_star's "source-view triplets" sharding) each rank takes a DIFFERENT
          (scan, ref view) sample and its own 1024 rays - what Lightning DDP + DistributedSampler would give the reference
          (train_mvs_nerf_pl.py:306,313): the effective batch is `world` scenes per step and the WHOLE step (encoder included) is
          divided; the only exchange is the 1.9 MB flat all-reduce.  BN batch statistics are per rank (DDP without SyncBN); the
          running-stat buffers are broadcast from rank 0 at the end of fit_steps and before save_ckpt (DDP's broadcast_buffers).
  "ray"   every rank encodes the SAME scene sample and draws the same pixel ids / jitter (ONE base seed broadcast from rank 0 at
          the start of fit_steps, step i uses base + i), renders its slice of the rays and back-propagates through its slice AND
          the replicated encoder: the step equals the 1-GPU step on the same batch, but the encoder's ~8.5 of ~10 ms are repeated on
          every rank (bounded speed-up, <= ~1.2x at 8 GPUs by construction).  It is the right mode where no encoder runs per step
          (per-scene fine-tuning, MVSSystemFinetune), not for this class.
Nothing here has run on more than one GPU (the dev boxes have one): every multi-GPU statement is by construction + world-size-2 gloo
tests, UNMEASURED on hardware.

`args.use_amp` (the reference's `precision=16 if args.use_amp`, train_mvs_nerf_pl.py:317-318; BASELINE config 3 "bf16"): the ray-march MLP
trains on v_mfma_f32_32x32x16_bf16 - forward with a 16-bit activation store, data- and weight-gradient GEMMs - and the encoder on
v_mfma_f32_16x16x32_bf16: conv0 of CostRegNet (74.5 % of the encoder's FLOPs) from a bf16 cost volume (csrc/conv_bf16.hip), conv1 .. conv11 and
FeatureNet from their fp32 activations rounded on load (csrc/conv3d_bf16.hip, csrc/wgrad_bf16.hip) - forward, data and weight gradients alike -
with fp32 accumulation, fp32 master weights and an fp32 gradient all-reduce.  The plane sweep's arithmetic and InPlaceABN (statistics,
normalisation, its backward) stay fp32, as autocast keeps grid_sample and batch norm.
"""
import os

import torch
import torch.nn as nn

from . import distributed as D
from . import ops
from .models import MVSNeRF, create_nerf_mvs
from .renderer import rendering
from .utils import build_rays, build_rays_test, img2mse


_UNPRE = {}


def _adam_kw(params):
    """torch's single-kernel Adam when every parameter lives on the GPU (same update rule as the reference's torch.optim.Adam; the default
    multi-tensor path makes ~8 passes over the parameters - 0.6 ms per fine-tuning step for the 150 MB learnable volume)."""
    ps = list(params)
    return {"fused": True} if ps and all(p.is_cuda and p.dtype == torch.float32 for p in ps) else {}


def _adam(params, lr):
    """The reference's torch.optim.Adam(betas=(0.9, 0.999)).  fp32 parameters on the GPU: mvsnerf_amd.optim.Adam - the same update, state_dict and
    hooks with ONE launch per step for all tensors (torch's fused Adam: three ~25 us launches for the 78 tensors of the generalizable step)."""
    ps = list(params)
    if ps and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps):
        from .optim import Adam
        return Adam(ps, lr=lr, betas=(0.9, 0.999))
    return torch.optim.Adam(ps, lr=lr, betas=(0.9, 0.999), **_adam_kw(ps))


def mse2psnr2(x):
    """utils.py:28-30.  A device tensor stays on the device (no host synchronisation inside the training step)."""
    import math
    if torch.is_tensor(x):
        return -10.0 * torch.log(x.detach().clamp_min(1e-20)) / math.log(10.0)
    return -10.0 * math.log(max(x, 1e-20)) / math.log(10.0)


class SL1Loss(nn.Module):
    """reference train_mvs_nerf_pl.py:22-32."""

    def __init__(self, levels=3):
        super().__init__()
        self.loss = nn.SmoothL1Loss(reduction="mean")

    def forward(self, depth_pred, depth_gt, mask=None):
        if mask is None:
            mask = depth_gt > 0
        return self.loss(depth_pred[mask], depth_gt[mask]) * 2 ** (1 - 2)


class _ModuleShim(nn.Module):
    """What `training_step` needs from LightningModule."""

    def __init__(self):
        super().__init__()
        self.global_step = 0
        self.logged = {}

    def log(self, key, value, prog_bar=False, **kw):
        # tensors are kept as (detached) device tensors: reading one is the only host synchronisation, and it is the reader's
        self.logged[key] = value.detach() if torch.is_tensor(value) else float(value)

    def logged_values(self):
        """The last logged metrics as Python floats (synchronises with the device)."""
        return {k: float(v) for k, v in self.logged.items()}

    @property
    def device(self):
        return next(self.parameters()).device


class MVSSystem(_ModuleShim):
    """reference train_mvs_nerf_pl.py:34-288 (generalizable training)."""

    def __init__(self, args, n_depth_planes=128):
        super().__init__()
        self.args = args
        self.n_views = getattr(args, "n_views", 3)                       # extension (config 4: 5 views); the reference fixes 3
        self.args.feat_dim = 8 + self.n_views * 4                        # :38
        self.idx = 0
        self.loss = SL1Loss()
        self.learning_rate = args.lrate
        kw_train, kw_test, _, self.grad_vars = create_nerf_mvs(args, use_mvs=True, dir_embedder=False, pts_embedder=True)   # :45
        for k in ("N_samples", "ndc", "lindisp"):                        # filter_keys, utils.py:418-424
            kw_train.pop(k, None)
        self.render_kwargs_train, self.render_kwargs_test = kw_train, kw_test
        self.MVSNet = kw_train.pop("network_mvs")                        # registered as sub-module `MVSNet` like the reference
        self.MVSNet.D = n_depth_planes
        self.network_fn = kw_train["network_fn"]                         # registers the MLP parameters
        self.render_kwargs_train["NDC_local"] = False
        self.eval_metric = [0.01, 0.05, 0.1]
        self._allreduce = None

    # -- data plumbing -------------------------------------------------------------------------
    def decode_batch(self, batch):
        """:56-62 - move to device, squeeze the B=1 dim of the pose tensors."""
        dev = self.device
        data = {k: (v.to(dev, torch.float32, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        pose_ref = {k: data[k].squeeze(0) if data[k].dim() > 3 or k == "near_fars" else data[k]
                    for k in ("w2cs", "intrinsics", "c2ws", "near_fars")}
        return data, pose_ref

    @staticmethod
    def unpreprocess(data, shape=(1, 1, 3, 1, 1)):
        """:64-71 - undo the ImageNet normalisation (colour lookups use raw [0,1] images)."""
        key = (data.device, data.dtype)
        if key not in _UNPRE:                     # built once per device: a torch.tensor(list, device=...) is a synchronous upload
            _UNPRE[key] = (torch.tensor([-0.485 / 0.229, -0.456 / 0.224, -0.406 / 0.225], device=data.device, dtype=data.dtype),
                           torch.tensor([1 / 0.229, 1 / 0.224, 1 / 0.225], device=data.device, dtype=data.dtype))
        mean, std = _UNPRE[key]
        return (data - mean.view(*shape)) / std.view(*shape)

    def configure_optimizers(self):
        """:84-88."""
        self.optimizer = _adam(self.grad_vars, self.learning_rate)
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=self.args.num_epochs, eta_min=1e-7)
        return [self.optimizer], [sched]

    # -- the step ------------------------------------------------------------------------------
    def training_step(self, batch, batch_nb):
        """:104-168.  Returns {'loss': loss}; gradients are left to the caller (Lightning or `fit_steps`)."""
        args = self.args
        batch = dict(batch)
        batch.pop("scan", None)
        data_mvs, pose_ref = self.decode_batch(batch)
        imgs, proj_mats = data_mvs["images"], data_mvs["proj_mats"]
        near_fars, depths_h = data_mvs["near_fars"], data_mvs["depths_h"]

        nv = self.n_views
        from . import encoder as _enc
        with _enc.encoder_precision("bf16" if getattr(args, "use_amp", False) else _enc.ENCODER_PRECISION):   # :317-318 precision=16: conv0 .. conv11 and FeatureNet on bf16
            volume_feature, _, _ = self.MVSNet(imgs[:, :nv], proj_mats[:, :nv], near_fars[0, 0], pad=args.pad)      # :113
        imgs = self.unpreprocess(imgs)
        N_rays, N_samples = args.batch_size, args.N_samples
        rays_pts, rays_dir, target_s, rays_NDC, depth_candidates, rays_o, rays_depth, _ = build_rays(
            imgs, depths_h, pose_ref, pose_ref["w2cs"], pose_ref["c2ws"], pose_ref["intrinsics"], near_fars, N_rays, N_samples, pad=args.pad)  # :119
        loss_scale, depth_scale, depth_term = 1.0, None, None
        n_masked_all = (rays_depth > 0).sum() if (rays_depth is not None and getattr(args, "with_depth", False)) else None
        if self.dp_mode() == "ray":                                      # same draw on all ranks (seeded in fit_steps), local slice
            (rays_pts, rays_dir, target_s, rays_NDC, depth_candidates, rays_depth, rays_o), loss_scale = D.shard_ray_batch(
                (rays_pts, rays_dir, target_s, rays_NDC, depth_candidates, rays_depth, rays_o), N_rays)     # rays_o is (3, N): sliced in dim 1
        # --use_amp (train_mvs_nerf_pl.py:317-318 `precision=16`): the MLP's forward / backward GEMMs on the bf16 matrix cores with
        # fp32 accumulation, master weights and gradients; the encoder runs its bf16 kernels under the context above (encoder.py)
        with ops.mlp_precision("bf16" if getattr(args, "use_amp", False) else ops.MLP_PRECISION):
            rgb, disp, acc, depth_pred, alpha, ret = rendering(args, pose_ref, rays_pts, rays_NDC, depth_candidates, rays_o, rays_dir,
                                                               volume_feature, imgs[:, :-1], img_feat=None, **self.render_kwargs_train)   # :123
        loss = 0
        if getattr(args, "with_depth", False):
            mask = rays_depth > 0
            if getattr(args, "with_depth_loss", False):
                if self.dp_mode() == "ray" and D.world_rank()[0] > 1:
                    # The depth term is a mean over the MASKED rays of the whole draw.  A rank's slice (128 rays at 8 GPUs) may hold none of
                    # them: a local masked mean would then be NaN, and the flat all-reduce would hand that NaN to every rank.  So: local masked
                    # SUM (0 for an empty slice) x world / global masked count - every rank holds the full draw before slicing, so the global
                    # count is local knowledge (device tensors, no host sync); the rank-averaged gradient is that of the global mean.
                    dsum = torch.nn.functional.smooth_l1_loss(depth_pred[mask], rays_depth[mask], reduction="sum") * 2 ** (1 - 2)
                    depth_term = dsum / mask.sum().clamp_min(1)                     # reported: local mean (0 when the slice has no masked ray)
                    depth_scale = dsum * float(D.world_rank()[0]) / n_masked_all.clamp_min(1)   # back-propagated
                else:
                    depth_term = self.loss(depth_pred, rays_depth, mask)
                loss = loss + depth_term
            with torch.no_grad():                                        # :130-138
                err = (depth_pred - rays_depth)[mask].abs()
                for t in self.eval_metric:
                    self.log(f"train/acc_l_{t}mm", (err < t).float().mean(), prog_bar=False)
                self.log("train/abs_err", err.mean(), prog_bar=True)
        img_loss = img2mse(rgb, target_s)                                # :143
        loss = loss + img_loss
        with torch.no_grad():
            self.log("train/loss", loss, prog_bar=True)
            self.log("train/img_mse_loss", img_loss)
            if getattr(args, "with_depth", False):                       # :152-155: PSNR over the masked rays, PSNR_out over the rest
                self.log("train/PSNR", mse2psnr2(img2mse(rgb[mask], target_s[mask])), prog_bar=True)
                self.log("train/PSNR_out", mse2psnr2(img2mse(rgb[~mask], target_s[~mask])), prog_bar=True)
            else:
                self.log("train/PSNR", mse2psnr2(img_loss), prog_bar=True)
        if self.global_step % 20000 == 19999:
            if self.dp_mode() == "scene":
                self.sync_buffers()                                       # collective: every rank takes part, rank 0 writes
            if D.world_rank()[1] == 0:                                    # one writer (ranks hold identical weights after the all-reduce)
                self.save_ckpt(f"{self.global_step}")
        # ray mode with N_rays % world != 0: the tensor that is back-propagated is weighted so that the rank-AVERAGED gradient is that of
        # the global mean; what is reported / logged stays the plain local mean ('loss_unscaled')
        if loss_scale == 1.0 and depth_scale is None:
            return {"loss": loss}
        scaled = img_loss * loss_scale + (0 if depth_term is None else (depth_scale if depth_scale is not None else depth_term * loss_scale))
        return {"loss": scaled, "loss_unscaled": loss.detach()}

    @torch.no_grad()
    def encode_scene(self, batch):
        """The scene encode of render_view alone: MVSNet on the batch's source views -> the neural volume (1,8,D,h,w).  The reference's video path encodes a scene
        ONCE and renders every camera pose of the path from that volume (renderer_video.ipynb cell 8: `volume_feature` outside the pose loop); pass the result to
        render_view(..., volume=...) for the same.  Under tile-parallel inference this is what removes the replicated encode from the per-frame time."""
        data_mvs, pose_ref = self.decode_batch(dict(batch))
        imgs, proj_mats, near_fars = data_mvs["images"], data_mvs["proj_mats"], pose_ref["near_fars"]
        V = imgs.shape[1] - 1
        self.MVSNet.train()                                              # :182 batch-statistics ABN also at inference
        return self.MVSNet(imgs[:, :V], proj_mats[:, :V], near_fars[0], pad=self.args.pad)[0]

    @torch.no_grad()
    def render_view(self, batch, chunk=None, whole_frame_off=False, target=None, batch_rays=16384, volume=None):
        """The rendering part of validation_step (:172-254): encode once, then the chunk loop over the target view's
        pixels - tile-parallel over ranks (contiguous chunk ranges + one all_gather).  Returns (rgb (H,W,3), depth (H,W)).
        whole_frame_off=True keeps the per-chunk Python loop (build_rays_test + rendering per chunk) instead of the single
        mvsnerf_render_pixels_fwd call; both produce the same pixels.
        target (extension, BASELINE config 5): dict(hw=(H,W), intrinsic (3,3), c2w (4,4)[, near_far (2,)]) renders a camera whose
        pixel grid differs from the source views' (e.g. 1008x756 rays over 960x640 sources).  The reference normalises the NDC
        coordinates with the *target* size and intrinsics (utils.py:252-253, fine when all views share both); with `target` the
        reference view's own intrinsics and size are used, which is what the volume is aligned with.
        batch_rays: rays per sub-batch inside the library call (free parameter: the pixels do not depend on it; measured on a 512x640
        frame: 1024 -> 85.5 ms, 4096 -> 81.7, 16384 -> 80.5, 65536 -> 80.9; the workspace is 16 KB per ray).
        volume: the result of encode_scene(batch) - the encode is skipped (a camera path over one scene; the pixels are the same)."""
        args = self.args
        chunk = chunk or args.chunk
        data_mvs, pose_ref = self.decode_batch(dict(batch))
        imgs, proj_mats, near_fars = data_mvs["images"], data_mvs["proj_mats"], pose_ref["near_fars"]
        H, W = int(imgs.shape[-2]), int(imgs.shape[-1])
        V = imgs.shape[1] - 1                                            # source views (3 in the reference's batches, :193)
        if volume is None:
            self.MVSNet.train()                                          # :182 batch-statistics ABN also at inference
            volume_feature, _, _ = self.MVSNet(imgs[:, :V], proj_mats[:, :V], near_fars[0], pad=args.pad)
        else:
            volume_feature = volume
        imgs = self.unpreprocess(imgs)
        world_to_ref, tgt_to_world, intrinsic = pose_ref["w2cs"][0], pose_ref["c2ws"][-1], pose_ref["intrinsics"][-1]
        k_ref, ref_hw, nf_tgt = None, None, near_fars[-1]
        if target is not None:
            dev = imgs.device
            ref_hw, (H, W) = (H, W), (int(target["hw"][0]), int(target["hw"][1]))
            intrinsic, tgt_to_world = target["intrinsic"].to(dev, torch.float32), target["c2w"].to(dev, torch.float32)
            k_ref = pose_ref["intrinsics"][0]
            if "near_far" in target:
                nf_tgt = target["near_far"].to(dev, torch.float32)

        kw = self.render_kwargs_train
        net = kw["network_fn"]
        fused = (isinstance(net, MVSNeRF) and getattr(kw.get("network_query_fn"), "_mvsnerf_fused", False)
                 and not getattr(args, "use_color_volume", False) and args.feat_dim == 8 + 4 * V and not whole_frame_off)
        if target is not None and not fused:
            raise RuntimeError("render_view(target=...) needs the fused ray-march path (MVSNeRF + fused network_query_fn)")
        if fused:
            # one FFI call per rank: the chunk loop (build_rays_test + rendering per chunk) runs inside the library
            k_render = intrinsic if intrinsic.dim() == 2 else intrinsic.mean(0)
            nf_t, nf_r = nf_tgt.reshape(-1)[:2].contiguous(), near_fars[0].reshape(-1)[:2].contiguous()
            vol_cl = ops.channels_last_volume(volume_feature)
            src = imgs[0, :-1].contiguous()

            def render_range(first, n):
                o = ops.render_pixels(vol_cl, src, pose_ref["w2cs"][:V].contiguous(), pose_ref["intrinsics"][:V].contiguous(),
                                      net.packed(args.feat_dim), H, W, k_render, tgt_to_world, k_render if k_ref is None else k_ref,
                                      world_to_ref, nf_t, nf_r, args.N_samples, first_pixel=first, n_pixels=n, pad=args.pad,
                                      white_bkgd=kw.get("white_bkgd", False), ref_hw=ref_hw, batch_rays=batch_rays,
                                      **net.packed_alt(args.feat_dim))
                return o["rgb"], o["depth"]
            rgb, depth = D.render_frame_pixels(render_range, H, W, chunk, device=imgs.device)
            return rgb.reshape(H, W, 3), depth.reshape(H, W)

        def render_chunk(idx):
            rays_pts, rays_dir, rays_NDC, depth_candidates, rays_o, _ = build_rays_test(
                H, W, tgt_to_world, world_to_ref, intrinsic, near_fars, near_fars[-1], args.N_samples, pad=args.pad, chunk=chunk, idx=idx)
            rgb, _, _, depth_pred, _, _ = rendering(args, pose_ref, rays_pts, rays_NDC, depth_candidates, rays_o, rays_dir,
                                                    volume_feature, imgs[:, :-1], img_feat=None, **self.render_kwargs_train)
            return rgb, depth_pred
        rgb, depth = D.render_frame(render_chunk, H, W, chunk)
        return rgb.reshape(H, W, 3), depth.reshape(H, W)

    @torch.no_grad()
    def validation_step(self, batch, batch_nb):
        """:172-254.  Renders the batch's last view from the first three (render_view: encode + the chunk loop, tile-parallel over the
        ranks) and returns the reference's per-sample log dict: 'val_psnr', 'val_depth_loss_r', 'val_abs_err', 'mask_sum',
        'val_acc_{t}mm' for t in eval_metric (sums over the mask, as the reference - validation_epoch_end divides by mask_sum).
        Left out: the TensorBoard image grids (`self.logger.experiment.add_images`) and the PNG dump (cv2 colour maps / imageio are
        not in this image); `args.img_downscale = rand*0.75+0.25` (:187) is drawn like the reference but, like there, consumed by
        nothing on the img_feat=None path."""
        args = self.args
        batch = dict(batch)
        batch.pop("scan", None)
        from .evaluate import abs_error, acc_threshold
        from .utils import mse2psnr
        log = {k: torch.tensor([0.0], dtype=torch.float64) for k in
               ["val_psnr", "val_depth_loss_r", "val_abs_err", "mask_sum"] + [f"val_acc_{i}mm" for i in self.eval_metric]}      # init_log, utils.py:23-26
        args.img_downscale = float(torch.rand((1,)) * 0.75 + 0.25)                                       # :187
        rgb, depth_r = self.render_view(batch)                                                          # (H,W,3), (H,W)
        rgb = torch.clamp(rgb.permute(2, 0, 1), 0, 1).cpu()                                             # :207
        depth_r = depth_r.cpu()
        tgt = self.unpreprocess(batch["images"].to(torch.float32).cpu())[0, -1]                            # :193 + :208
        img_err_abs = (rgb - tgt).abs()
        if getattr(args, "with_depth", False):
            depth_gt = batch["depths_h"][0, -1].to(torch.float32).cpu()
            mask = depth_gt > 0
            log["val_psnr"] = mse2psnr(torch.mean(img_err_abs[:, mask] ** 2))                             # :213
            log["val_depth_loss_r"] = self.loss(depth_r, depth_gt, mask)                                # :220
            log["val_abs_err"] = abs_error(depth_r, depth_gt, mask).sum()                               # :229
            for t in self.eval_metric:
                log[f"val_acc_{t}mm"] = acc_threshold(depth_r, depth_gt, mask, t).sum()                 # :230-232
            log["mask_sum"] = mask.float().sum()
        else:
            log["val_psnr"] = mse2psnr(torch.mean(img_err_abs ** 2))                                      # :215
        self.idx += 1
        self.last_val_images = {"rgb": rgb, "depth": depth_r, "err": img_err_abs}                       # what the reference's image grids show
        return log

    def validation_epoch_end(self, outputs):
        """:256-275: the 'val/*' keys."""
        st = lambda k: torch.stack([torch.as_tensor(x[k], dtype=torch.float64).reshape(()) for x in outputs])
        mask_sum = st("mask_sum").sum()
        self.log("val/d_loss_r", st("val_depth_loss_r").mean(), prog_bar=False)
        self.log("val/PSNR", st("val_psnr").mean(), prog_bar=False)
        self.log("val/abs_err", st("val_abs_err").sum() / mask_sum, prog_bar=False)
        for t in self.eval_metric:
            self.log(f"val/acc_{t}mm", st(f"val_acc_{t}mm").sum() / mask_sum, prog_bar=False)

    def save_ckpt(self, name="latest"):
        """:277-288 - same dict keys as the reference's .tar checkpoints."""
        save_dir = f"runs_new/{getattr(self.args, 'expname', 'exp')}/ckpts/"
        os.makedirs(save_dir, exist_ok=True)
        path = f"{save_dir}/{name}.tar"
        torch.save({"global_step": self.global_step, "network_fn_state_dict": self.render_kwargs_train["network_fn"].state_dict(),
                    "network_mvs_state_dict": self.MVSNet.state_dict()}, path)
        return path

    # -- minimal trainer -----------------------------------------------------------------------
    def dp_mode(self):
        """"ray" | "scene" (module docstring); anything else is rejected loudly."""
        mode = getattr(self.args, "dp_mode", "scene")
        if mode not in ("ray", "scene"):
            raise ValueError(f"args.dp_mode must be 'ray' or 'scene', got {mode!r}")
        return mode

    def fit_steps(self, batches, optimizer=None):
        """Lightning-free loop: training_step -> backward -> (flat-buffer all-reduce) -> Adam step.
        dp_mode "ray": `batches` is the same list on every rank; "scene": the list is sharded round-robin over the ranks
        (distributed.scene_shard) and every rank seeds its own pixel-id / jitter streams."""
        if optimizer is None:
            optimizer = self.configure_optimizers()[0][0]
        if self._allreduce is None:
            self._allreduce = D.FlatGradAllReduce(self.grad_vars)
        world, rank = D.world_rank()
        mode = self.dp_mode()
        if world > 1 and mode == "scene":
            batches = D.scene_shard(list(batches))
            if not getattr(self, "_scene_seeded", False):
                torch.manual_seed(torch.initial_seed() + 7919 * rank)      # independent draws per rank from here on
                self._scene_seeded = True
        losses = []
        base_seed = D.common_seed(self.device) if (world > 1 and mode == "ray") else None   # ONE broadcast + host read per call, not per step
        for i, batch in enumerate(batches):
            if base_seed is not None:
                torch.manual_seed(base_seed + i)                             # same pixel ids (CPU RNG) and jitter (device RNG) everywhere
            optimizer.zero_grad(set_to_none=True)
            out = self.training_step(batch, i)
            out["loss"].backward()
            self._allreduce()
            optimizer.step()
            self.global_step += 1
            losses.append(out.get("loss_unscaled", out["loss"]).detach())
        if world > 1 and mode == "scene":
            self.sync_buffers()
        return [float(l) for l in losses]                                   # one host synchronisation, after the last step is enqueued

    def sync_buffers(self):
        """Scene-sharded DP: every rank updates the InPlaceABN running statistics from its own scenes.  Like DDP's default
        broadcast_buffers=True, rank 0's buffers become everybody's (one flat broadcast); the forward pass never reads them
        (MVSNet runs in train mode, batch statistics), so it only matters for what save_ckpt writes."""
        if not D._collective_needed():
            return
        bufs = [b for b in self.MVSNet.buffers() if b.is_floating_point()]
        if not bufs:
            return
        flat = torch.cat([b.reshape(-1) for b in bufs])
        D.broadcast(flat, src=0)
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()


def synthetic_batch(H=512, W=640, seed=1234, **rig_kw):
    """A `MVSDatasetDTU.__getitem__`-shaped batch (collated, B=1) from the seeded synthetic rig."""
    from .synth import make_rig
    rig = make_rig(H, W, seed=seed, **rig_kw)
    return {"images": rig["images"], "proj_mats": rig["proj_mats"], "w2cs": rig["w2cs"], "c2ws": rig["c2ws"],
            "intrinsics": rig["intrinsics"], "near_fars": rig["near_fars"], "depths_h": torch.zeros(1, 4, 1, 1)}


def batch_to_device(batch, device):
    """The collated batch with its tensors on `device` (what a pinned-memory DataLoader + prefetcher hands the step): training_step then
    issues no host-to-device copy of its own."""
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def default_args(**over):
    """opt.py defaults of the fields the hot path reads (the reference parses them with configargparse)."""
    import types
    d = dict(expname="exp", pad=24, batch_size=1024, num_epochs=8, pts_dim=3, dir_dim=3, net_type="v0", netdepth=6, netwidth=128,
             lrate=5e-4, chunk=1024, netchunk=1024, ckpt=None, N_samples=128, N_importance=0, perturb=1.0, use_viewdirs=True,
             i_embed=0, multires=10, multires_views=4, raw_noise_std=0.0, white_bkgd=False, img_downscale=1.0,
             use_color_volume=False, with_depth=False, with_depth_loss=False, feat_dim=20, dp_mode="scene", use_amp=False)
    d.update(over)
    return types.SimpleNamespace(**d)


# ------------------------------------------------------------------ per-scene fine-tuning (reference train_mvs_nerf_finetuning_pl.py)
def ray_marcher(rays, N_samples=64, lindisp=False, perturb=0):
    """reference data/ray_utils.py:152-197 (bbox_3D=None): rays (N,8) = [o(3) | d(3) | near | far] -> z-sampled points.
    Host-side torch like the reference (it owns the jitter draw)."""
    n = rays.shape[0]
    rays_o, rays_d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    steps = torch.linspace(0, 1, N_samples, device=rays.device)
    z = near * (1 - steps) + far * steps if not lindisp else 1 / (1 / near * (1 - steps) + 1 / far * steps)
    z = z.expand(n, N_samples)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper, lower = torch.cat([mid, z[:, -1:]], -1), torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * torch.rand(z.shape, device=rays.device))
    return rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(2), rays_o, rays_d, z


def sample_pdf(bins, weights, N_samples, det=False, pytest=False, u=None):
    """reference data/ray_utils.py:96-139.  The uniform draw stays here (torch.rand on the device like :109, or `u` supplied);
    CDF construction + inversion is one HIP kernel (one wave per ray)."""
    if pytest:
        raise NotImplementedError("sample_pdf(pytest=True) overwrites u with numpy's RNG in the reference; pass u= instead")
    shape = list(weights.shape[:-1]) + [N_samples]
    if u is None:
        u = (torch.linspace(0.0, 1.0, steps=N_samples, device=weights.device).expand(shape) if det
             else torch.rand(shape, device=weights.device))
    return ops.sample_pdf(bins.detach(), weights.detach(), u.contiguous())


def ray_marcher_fine(rays, density_volume, z_vals, pts_NDC, N_importance=64, lindisp=False, u=None):
    """reference data/ray_utils.py:199-224: importance samples from the density volume merged into the coarse depths.
    Returns (xyz (N,S+NI,3), rays_o, rays_d, z_vals (N,S+NI)).  Density lookup, weights, sample_pdf and the sort are one
    kernel; `u` is the torch.rand draw of sample_pdf (drawn here when not supplied)."""
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    if u is None:
        u = torch.rand((rays.shape[0], N_importance), device=rays.device)
    with torch.no_grad():
        z = ops.ray_marcher_fine_z(density_volume.detach(), pts_NDC.detach(), z_vals.detach(), u)
        xyz, _ = ops.ray_points(rays_o, rays_d, z)
    return xyz, rays_o, rays_d, z


class MVSSystemFinetune(_ModuleShim):
    """reference train_mvs_nerf_finetuning_pl.py:32-189: the scene is encoded ONCE (`init_volume`), the 8-channel
    volume becomes a learnable `RefVolume` (checkpoint key `volume.feat_volume`) and only the ray march runs per step;
    gradients flow to the MLP and to the volume (trilinear scatter kernel)."""

    def __init__(self, args, source_views, n_depth_planes=128):
        """source_views = (imgs (1,V,3,H,W) normalised, proj_mats (1,V,3,4), near_far (2,), pose_source dict) -
        what `dataset.read_source_views()` returns in the reference."""
        super().__init__()
        from .models import RefVolume
        self.args = args
        self.args.feat_dim = 8 + 4 * int(getattr(args, "n_views", 3))       # :39 hard-wires 3 source views (8 + 3*4); n_views is the config-4 extension
        if getattr(args, "use_color_volume", False) and getattr(args, "use_density_volume", False):
            raise NotImplementedError("--use_color_volume together with --use_density_volume: update_density_volume would concatenate the "
                                      "colours to a volume that already holds them (train_mvs_nerf_finetuning_pl.py:96); not supported")
        kw_train, _, _, self.grad_vars = create_nerf_mvs(args, use_mvs=True, dir_embedder=False, pts_embedder=True)
        for k in ("N_samples", "ndc", "lindisp"):
            kw_train.pop(k, None)
        self.render_kwargs_train = kw_train
        self.MVSNet = kw_train.pop("network_mvs")
        self.MVSNet.D = n_depth_planes
        self.network_fn = kw_train["network_fn"]
        imgs, proj_mats, near_far, pose = source_views
        dev = next(self.MVSNet.parameters()).device
        self.near_far_source = near_far.to(dev)
        self.pose_source = {k: v.to(dev) for k, v in pose.items()}
        # init_volume :57-66: a checkpoint written by save_ckpt of THIS class carries the optimised volume - resume from it
        # instead of re-encoding (the reference does the same); otherwise encode the scene once
        vol = None
        ck_path = getattr(args, "ckpt", None)
        if ck_path and ck_path != "None" and os.path.exists(ck_path):
            ckpts = torch.load(ck_path, map_location="cpu", weights_only=False)
            if "volume" in ckpts:
                vol = ckpts["volume"]["feat_volume"].to(dev, torch.float32)
                self.volume_from_ckpt = True
        if vol is None:
            self.volume_from_ckpt = False
            self.MVSNet.train()                                           # :62
            with torch.no_grad():
                vol, _, _ = self.MVSNet(imgs.to(dev), proj_mats.to(dev), self.near_far_source, pad=args.pad, lindisp=getattr(args, "use_disp", False))
        self.imgs = MVSSystem.unpreprocess(imgs.to(dev))
        # importance sampling from a density volume (:73-86): voxel positions + per-voxel colour features, once per scene
        self.density_volume = None
        use_cv, use_dv = bool(getattr(args, "use_color_volume", False)), bool(getattr(args, "use_density_volume", False))
        if use_cv or use_dv:
            from .utils import get_ptsvolume, build_color_volume
            Dv, Hv, Wv = vol.shape[-3:]
            intrinsic, c2w = self.pose_source["intrinsics"][0].clone(), self.pose_source["c2ws"][0]
            intrinsic[:2] /= 4
            vox_pts = get_ptsvolume(Hv - 2 * args.pad, Wv - 2 * args.pad, Dv, args.pad, self.near_far_source, intrinsic, c2w).contiguous()
            with torch.no_grad():
                self.color_feature = build_color_volume(vox_pts, self.pose_source, self.imgs, with_mask=True)   # (D*H, W, 4V)
            if use_cv and vol.shape[1] == 8:
                # :79-80: the projected colours become 4V extra channels of the learnable volume (a checkpoint written by this class
                # already holds all 8+4V channels; the reference would concatenate a second copy there, :66 + :80)
                cf = self.color_feature.reshape(Dv, Hv, Wv, -1).permute(3, 0, 1, 2).unsqueeze(0)
                vol = torch.cat((vol, cf), dim=1)
            if use_dv:
                self.vox_pts = vox_pts
        self.volume = RefVolume(vol.detach())
        self.grad_vars = [p for p in self.network_fn.parameters()] + list(self.volume.parameters())    # MVSNet stays frozen here
        self._allreduce = None
        # Data parallelism (rays of the scene's all-rays buffer sharded over the ranks): args.dp_volume_grad selects how the gradient of
        # the 150-246 MB volume is combined - "samples" (default): all_gather of the per-sample feature gradients + a local scatter on
        # every rank (ops.volume_grad_from_all_ranks: 6 MB per step instead of a volume-sized all-reduce); "allreduce": the volume joins
        # the flat all-reduce of the MLP gradients.
        if not hasattr(args, "dp_volume_grad"):
            args.dp_volume_grad = "samples"
        if args.dp_volume_grad not in ("samples", "allreduce"):
            raise ValueError("args.dp_volume_grad must be 'samples' or 'allreduce'")
        if args.dp_volume_grad == "samples":
            self._allreduce = D.FlatGradAllReduce([p for p in self.network_fn.parameters()])           # the volume gradient arrives complete

    def update_density_volume(self):
        """:91-99: sigma of every voxel centre from the current volume + MLP (forward_alpha queries), -> (D,H,W)."""
        from .renderer import render_density
        with torch.no_grad():
            Dv, Hv, Wv = self.volume.feat_volume.shape[-3:]
            vol_cl = ops.channels_last_volume(self.volume.feat_volume.detach())                     # (D,H,W,8)
            features = torch.cat((vol_cl.reshape(Dv * Hv, Wv, 8), self.color_feature), -1)           # = cat(...).permute(0,2,3,4,1) :96
            self.density_volume = render_density(self.network_fn, self.vox_pts, features,
                                                 self.render_kwargs_train["network_query_fn"]).reshape(Dv, Hv, Wv)

    def training_step(self, batch, batch_nb):
        """:140-189.  batch = {'rays': (1,B,8), 'rgbs': (1,B,3)} from the all-rays buffer."""
        args = self.args
        rays, target = batch["rays"].squeeze(0).to(self.imgs.device), batch["rgbs"].squeeze(0).to(self.imgs.device)
        if getattr(args, "use_density_volume", False) and 0 == self.global_step % 200:              # :144-145
            self.update_density_volume()
        lindisp = getattr(args, "use_disp", False)
        pts, rays_o, rays_d, z_vals = ray_marcher(rays, N_samples=args.N_samples, lindisp=lindisp, perturb=args.perturb)
        H, W = self.imgs.shape[-2:]

        def to_ndc(z):     # o + d*z and get_ndc_coordinate (:150-156) in one kernel
            return ops.ray_points(rays_o, rays_d, z, self.pose_source["w2cs"][0], self.pose_source["intrinsics"][0], self.near_far_source,
                                  ref_hw=(H, W), pad=args.pad, lindisp=lindisp)
        pts, ndc = to_ndc(z_vals)
        if self.density_volume is not None and getattr(args, "N_importance", 0) > 0:                # :158-163
            pts, rays_o, rays_d, z_vals = ray_marcher_fine(rays, self.density_volume, z_vals, ndc, N_importance=args.N_importance)
            pts, ndc = to_ndc(z_vals)
        with ops.mlp_precision("bf16" if getattr(args, "use_amp", False) else ops.MLP_PRECISION):
            rgbs, _, _, depth_pred, _, _ = rendering(args, self.pose_source, pts, ndc, z_vals, rays_o, rays_d, self.volume, self.imgs,
                                                     **self.render_kwargs_train)
        img_loss = img2mse(rgbs, target)
        with torch.no_grad():
            self.log("train/loss", img_loss, prog_bar=True)
            self.log("train/PSNR", mse2psnr2(img_loss), prog_bar=True)
        return {"loss": img_loss}

    def save_ckpt(self, name="latest"):
        """:277-291 of the fine-tuning script: adds the `volume` state dict."""
        save_dir = f"runs_fine_tuning/{getattr(self.args, 'expname', 'exp')}/ckpts/"
        os.makedirs(save_dir, exist_ok=True)
        path = f"{save_dir}/{name}.tar"
        torch.save({"global_step": self.global_step, "network_fn_state_dict": self.network_fn.state_dict(),
                    "volume": self.volume.state_dict(), "network_mvs_state_dict": self.MVSNet.state_dict()}, path)
        return path

    def dp_mode(self):
        return "data"          # every rank feeds its own slice of the all-rays buffer; nothing to slice or re-seed in the step

    def fit_steps(self, batches, optimizer=None):
        """training_step -> backward -> gradient exchange -> Adam; with args.dp_volume_grad == "samples" the volume is re-broadcast from
        rank 0 every args.dp_volume_resync steps (default 200) so that last-bit differences of the atomics-ordered scatters cannot add up."""
        if optimizer is None:
            optimizer = self.configure_optimizers()[0][0]
        if self._allreduce is None:
            self._allreduce = D.FlatGradAllReduce(self.grad_vars)
        resync = int(getattr(self.args, "dp_volume_resync", 200))
        losses = []
        for i, batch in enumerate(batches):
            optimizer.zero_grad(set_to_none=True)
            out = self.training_step(batch, i)
            out["loss"].backward()
            self._allreduce()
            optimizer.step()
            self.global_step += 1
            if self.args.dp_volume_grad == "samples" and resync > 0 and self.global_step % resync == 0 and D._collective_needed():
                D.broadcast(ops.channels_last_volume(self.volume.feat_volume.data), src=0)     # the (D,H,W,C) view of the same memory
            losses.append(out["loss"].detach())
        return [float(l) for l in losses]                                   # one host synchronisation, after the last step is enqueued

    def configure_optimizers(self):
        self.optimizer = _adam(self.grad_vars, self.args.lrate)
        return [self.optimizer], []
