"""Shared-GPU mismatch, second probe: WHICH stage of the no-grad encode differs when K processes share the GPU.  race_probe.py found the blocked fp32
plane sweep and the fp32 matrix-core conv0 clean (0 of 1920 sweeps); here every process repeats, mode after mode (the processes change mode together):
  h2        FeatureNet -> two-piece fp16 plane sweep (compare both planes) -> fp16x3 conv0 (compare the raw output)
  guarded   FeatureNet -> mvsnerf_sweep_conv0_guarded_fwd (the default head; compare conv0's raw output and its partial sums)
  full      MVSNet.forward under the defaults (compare the neural volume)
  full32    MVSNet.forward under encoder_precision("fp32")
  featnet   FeatureNet alone (compare the features)
and compares with its own reference bit for bit; the host looks at the flags once per window of sweeps so that the queue stays full.
usage: race_probe2.py [K] [iters per mode] [modes, comma separated]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import torch.multiprocessing as mp

ALL = ("h2", "guarded", "full", "full32", "featnet")
# python-only isolation of the mode-3 (two-piece fp16) sweep: m3_only (fixed features, the sweep alone), m3_feat (FeatureNet -> sweep), m3_conv (fixed features,
# sweep -> fp16x3 conv0), m1_conv16 (fixed features, blocked fp32 sweep, then the fp16x3 conv0 on FIXED planes), m2_only (bf16 sweep alone), m1_only


def body(rank, K, iters, modes, q, bar):
    try:
        _body(rank, K, iters, modes, q, bar)
    except BaseException:
        import traceback
        q.put((rank, {"error": traceback.format_exc()[-1500:]}))
        try:
            bar.abort()
        except Exception:
            pass


def _body(rank, K, iters, modes, q, bar):
    import numpy as np
    from mvsnerf_amd import _lib as _L
    if os.environ.get("PSW_LIB"):                                    # a variant library (scratch/r4/variants/build_variants.sh)
        _L.LIB_PATH = os.path.abspath(os.environ["PSW_LIB"])
    from mvsnerf_amd import train, encoder, ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    args = train.default_args(pad=24, batch_size=1024, N_samples=128, chunk=1024)
    system = train.MVSSystem(args).to(dev)
    z = np.load('tests/golden/mvsnerf_v0_weights.npz')
    system.render_kwargs_train["network_fn"].load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")})
    system.MVSNet.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")})
    batch = train.batch_to_device(train.synthetic_batch(512, 640, seed=1234), dev)
    net = system.MVSNet
    data_mvs, _ = system.decode_batch(dict(batch))
    imgs, proj, nf = data_mvs["images"][:, :3], data_mvs["proj_mats"][:, :3], data_mvs["near_fars"][0, 0]
    B, V, _, Hi, Wi = imgs.shape
    t_vals = torch.linspace(0.0, 1.0, steps=net.D, device=dev)
    dv = (nf[0] * (1.0 - t_vals) + nf[1] * t_vals).unsqueeze(0)
    net.prepack()

    NVOX_PLANE = [0]
    FIXED = {}

    def feats_l():
        f = net.feature(imgs.reshape(B * V, 3, Hi, Wi))
        return f.view(B, V, *f.shape[1:])

    def run(mode):
        with torch.no_grad():
            if mode == "featnet":
                return {"feats": feats_l()}
            if mode == "h2":
                cost, _ = net._sweep(imgs, feats_l(), proj, dv, 24, True, blocked="fp16x2")
                D, H, W = cost.dims
                pk = net.cost_reg_2.conv0._packed
                raw = torch.empty((D, H, W, 8), device=dev)
                lib = encoder._lib.lib()
                encoder.check(lib.mvsnerf_conv0_f16x3_fwd(cost.buf.data_ptr(), pk.cin, D, H, W, encoder._get_f16x3_conv0(pk).data_ptr(), raw.data_ptr(), 0,
                                                          encoder.stream_ptr()), "conv0_f16x3_fwd")
                NVOX_PLANE[0] = H * W
                return {"planes": cost.buf.view(torch.int16), "raw": raw}
            if mode in ("m3_only", "m3_feat", "m3_conv", "m2_only", "m1_only", "m1_conv16"):
                f = feats_l() if mode == "m3_feat" else FIXED.setdefault("feats", feats_l())
                blk = {"m2_only": "bf16", "m1_only": True, "m1_conv16": True}.get(mode, "fp16x2")
                cost, _ = net._sweep(imgs, f, proj, dv, 24, True, blocked=blk)
                out = {"planes": cost.buf.view(torch.int16) if blk != True else cost.buf}
                if mode in ("m3_conv", "m1_conv16"):
                    D, H, W = cost.dims
                    pk = net.cost_reg_2.conv0._packed
                    raw = torch.empty((D, H, W, 8), device=dev)
                    lib = encoder._lib.lib()
                    if mode == "m1_conv16" and "planes16" not in FIXED:
                        FIXED["planes16"] = net._sweep(imgs, f, proj, dv, 24, True, blocked="fp16x2")[0].buf
                    src = cost.buf if mode == "m3_conv" else FIXED["planes16"]
                    encoder.check(lib.mvsnerf_conv0_f16x3_fwd(src.data_ptr(), pk.cin, D, H, W, encoder._get_f16x3_conv0(pk).data_ptr(), raw.data_ptr(), 0,
                                                              encoder.stream_ptr()), "conv0_f16x3_fwd")
                    out["raw"] = raw
                NVOX_PLANE[0] = cost.dims[1] * cost.dims[2]
                return out
            if mode in ("m1_conv32", "m2_convbf"):                  # which conv0 kernel in the loop makes the sweep before / after it differ?
                f = FIXED.setdefault("feats", feats_l())
                cost, _ = net._sweep(imgs, f, proj, dv, 24, True, blocked=True if mode == "m1_conv32" else "bf16")
                D, H, W = cost.dims
                pk = net.cost_reg_2.conv0._packed
                raw = torch.empty((D, H, W, 8), device=dev)
                lib = encoder._lib.lib()
                if mode == "m1_conv32":
                    encoder.check(lib.mvsnerf_conv3d_c8_blocked_fwd(cost.buf.data_ptr(), pk.cin_pad, pk.cin, D, H, W, pk.get_c8().data_ptr(), raw.data_ptr(),
                                                                    encoder.stream_ptr()), "conv0 fp32")
                    return {"planes": cost.buf, "raw": raw}
                encoder.check(lib.mvsnerf_conv0_bf16_fwd(cost.buf.data_ptr(), pk.cin, D, H, W, pk.get_bf16_conv0().data_ptr(), raw.data_ptr(), 0,
                                                         encoder.stream_ptr()), "conv0 bf16")
                return {"planes": cost.buf.view(torch.int16), "raw": raw}
            if mode in ("feat_conv16", "m0_conv16"):                # other victims?  FeatureNet / the channel-last fp32 sweep with the fp16x3 conv0 in the loop
                f = FIXED.setdefault("feats", feats_l())
                if "cost16" not in FIXED:
                    FIXED["cost16"] = net._sweep(imgs, f, proj, dv, 24, True, blocked="fp16x2")[0]
                c16 = FIXED["cost16"]
                D, H, W = c16.dims
                pk = net.cost_reg_2.conv0._packed
                raw = torch.empty((D, H, W, 8), device=dev)
                lib = encoder._lib.lib()
                out = {"feats": feats_l()} if mode == "feat_conv16" else {"cost": net._sweep(imgs, f, proj, dv, 24, True, blocked=False)[0]}
                encoder.check(lib.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), pk.cin, D, H, W, encoder._get_f16x3_conv0(pk).data_ptr(), raw.data_ptr(), 0,
                                                          encoder.stream_ptr()), "conv0_f16x3_fwd")
                out["raw"] = raw
                return out
            if mode in ("m1_2stream", "m1_2stream32"):              # ONE process, two streams: the conv0 on a side stream while the sweep runs on the main stream
                f = FIXED.setdefault("feats", feats_l())
                if "cost16" not in FIXED:
                    FIXED["cost16"] = net._sweep(imgs, f, proj, dv, 24, True, blocked="fp16x2")[0]
                    FIXED["cost32"] = net._sweep(imgs, f, proj, dv, 24, True, blocked=True)[0]
                    FIXED["side"] = torch.cuda.Stream()
                    FIXED["raw_side"] = torch.empty((*FIXED["cost16"].dims, 8), device=dev)
                    torch.cuda.synchronize()
                c16, c32 = FIXED["cost16"], FIXED["cost32"]
                D, H, W = c16.dims
                pk = net.cost_reg_2.conv0._packed
                lib = encoder._lib.lib()
                with torch.cuda.stream(FIXED["side"]):
                    for _ in range(2):
                        if mode == "m1_2stream":
                            encoder.check(lib.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), pk.cin, D, H, W, encoder._get_f16x3_conv0(pk).data_ptr(),
                                                                      FIXED["raw_side"].data_ptr(), 0, encoder.stream_ptr()), "conv0_f16x3_fwd")
                        else:
                            encoder.check(lib.mvsnerf_conv3d_c8_blocked_fwd(c32.buf.data_ptr(), pk.cin_pad, pk.cin, D, H, W, pk.get_c8().data_ptr(),
                                                                            FIXED["raw_side"].data_ptr(), encoder.stream_ptr()), "conv0 fp32")
                cost, _ = net._sweep(imgs, f, proj, dv, 24, True, blocked=True)
                return {"planes": cost.buf}
            if mode == "self_bf16":                                  # ONE stream: kernels that hold packed fp32 VALU and 16-bit MFMAs themselves
                with encoder.encoder_precision("bf16"):
                    return {"volume": net(imgs, proj, nf, pad=24)[0]}
            if mode == "self_frame":                                 # the default frame: guarded encode + guarded fp16x3 MLP
                rgb, depth = system.render_view(batch)
                return {"rgb": rgb, "depth": depth}
            if mode in ("bwd_2stream", "bwd_only"):                  # the plane sweep's BACKWARD (order-independent fixed-point form: bit-comparable) as a victim
                f = FIXED.setdefault("feats", feats_l())
                if "cost16" not in FIXED:
                    FIXED["cost16"] = net._sweep(imgs, f, proj, dv, 24, True, blocked="fp16x2")[0]
                if "gcost" not in FIXED:
                    FIXED["side"] = torch.cuda.Stream()
                    FIXED["raw_side"] = torch.empty((*FIXED["cost16"].dims, 8), device=dev)
                    Dp, Hp, Wp = FIXED["cost16"].dims
                    FIXED["gcost"] = torch.randn((Dp, Hp, Wp, 44), device=dev, generator=torch.Generator(dev).manual_seed(3)) * 1e-3
                    FIXED["fcl"] = encoder._images_channel_last(f[0], 32)[0].contiguous()
                    torch.cuda.synchronize()
                c16 = FIXED["cost16"]
                D, H, W = c16.dims
                pk = net.cost_reg_2.conv0._packed
                lib = encoder._lib.lib()
                if mode == "bwd_2stream":
                    with torch.cuda.stream(FIXED["side"]):
                        for _ in range(2):
                            encoder.check(lib.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), pk.cin, D, H, W, encoder._get_f16x3_conv0(pk).data_ptr(),
                                                                      FIXED["raw_side"].data_ptr(), 0, encoder.stream_ptr()), "conv0_f16x3_fwd")
                fcl = FIXED["fcl"]
                Vv, Hh, Ww, Cc = fcl.shape
                encoder.PSW_BWD_DETERMINISTIC = True
                g = encoder._planesweep_bwd(fcl, proj[0].contiguous(), dv[0].contiguous(), Vv, Cc, Hh, Ww, net.D, 24, FIXED["gcost"], 44, 1)
                return {"g_feats": g}
            if mode == "frame_2stream":                              # the whole default frame (encode + 20 sub-batches of the ray march) next to a foreign fp16 MFMA stream
                f = FIXED.setdefault("feats", feats_l())
                if "cost16" not in FIXED:
                    FIXED["cost16"] = net._sweep(imgs, f, proj, dv, 24, True, blocked="fp16x2")[0]
                if "side" not in FIXED:
                    FIXED["side"] = torch.cuda.Stream()
                    FIXED["raw_side"] = torch.empty((*FIXED["cost16"].dims, 8), device=dev)
                    torch.cuda.synchronize()
                c16 = FIXED["cost16"]
                D, H, W = c16.dims
                pk = net.cost_reg_2.conv0._packed
                lib = encoder._lib.lib()
                with torch.cuda.stream(FIXED["side"]):
                    for _ in range(80):                              # ~36 ms of conv0 launches: the frame takes ~33 ms alone
                        encoder.check(lib.mvsnerf_conv0_f16x3_fwd(c16.buf.data_ptr(), pk.cin, D, H, W, encoder._get_f16x3_conv0(pk).data_ptr(),
                                                                  FIXED["raw_side"].data_ptr(), 0, encoder.stream_ptr()), "conv0_f16x3_fwd")
                rgb, depth = system.render_view(batch)
                return {"rgb": rgb, "depth": depth}
            if mode == "guarded":
                cost, _ = net._sweep(imgs, feats_l(), proj, dv, 24, True, blocked="guarded")
                out = {"raw": cost.buf}
                if cost.part is not None:
                    out["partials"] = cost.part
                return out
            if mode == "full":
                return {"volume": net(imgs, proj, nf, pad=24)[0]}
            if mode == "full32":
                with encoder.encoder_precision("fp32"):
                    return {"volume": net(imgs, proj, nf, pad=24)[0]}
        raise ValueError(mode)

    def bits(t):
        return t if t.dtype in (torch.int16, torch.int32) else t.contiguous().view(torch.int32)

    report = {}
    for mode in modes:
        if mode == "time_sweep":                                    # kernel time of the blocked fp32 sweep (HIP events around 50 launches)
            f = FIXED.setdefault("feats", feats_l())
            with torch.no_grad():
                for _ in range(5):
                    net._sweep(imgs, f, proj, dv, 24, True, blocked=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(50):
                    c = net._sweep(imgs, f, proj, dv, 24, True, blocked=True)
                e1.record(); torch.cuda.synchronize()
            bar.wait()
            report[mode] = {"iters": 50, "mismatching": 0, "seconds": 0, "events": [], "guard_fallbacks": 0, "by_key": {"ms_per_sweep": round(e0.elapsed_time(e1) / 50, 4)}, "checksum": {}}
            continue
        ref = None
        for _ in range(4):
            cur = {k: bits(v).clone() for k, v in run(mode).items()}
            if ref is not None and all(torch.equal(cur[k], ref[k]) for k in ref):
                break
            ref = cur
        torch.cuda.synchronize()
        checksum = {k: int(v.to(torch.int64).sum()) for k, v in ref.items()}
        bar.wait()
        events, t0, WIN = [], time.time(), 8
        for w0 in range(0, iters, WIN):
            held = []
            for it in range(w0, min(iters, w0 + WIN)):
                cur = run(mode)
                held.append((it, cur, {k: (bits(v) != ref[k]).any() for k, v in cur.items()}))
            keys = list(ref)
            flags = torch.stack([h[2][k] for h in held for k in keys]).tolist()
            for i, (it, cur, _) in enumerate(held):
                for j, k in enumerate(keys):
                    if flags[i * len(keys) + j]:
                        d = bits(cur[k]) != ref[k]
                        ev = {"it": it, "what": k, "bad_words": int(d.sum())}
                        if k in ("raw", "volume", "feats") and cur[k].dim() >= 4:
                            t = cur[k]
                            if k == "volume":                    # logical (1,8,D,h,w) -> voxels
                                dvx = d.view(t.shape) if d.shape == t.shape else None
                                if dvx is not None:
                                    vox = dvx[0].any(0).nonzero()
                                    ev["bad_voxels"] = int(vox.shape[0]); ev["z"] = (int(vox[:, 0].min()), int(vox[:, 0].max()))
                            elif k == "raw":
                                vox = d.view(t.shape).any(-1).nonzero()
                                ev["bad_voxels"] = int(vox.shape[0]); ev["z"] = (int(vox[:, 0].min()), int(vox[:, 0].max()))
                                ev["y"] = (int(vox[:, 1].min()), int(vox[:, 1].max())); ev["x"] = (int(vox[:, 2].min()), int(vox[:, 2].max()))
                                ev["tiles_4x8x16"] = int(torch.unique(torch.stack([vox[:, 0] // 4, vox[:, 1] // 8, vox[:, 2] // 16], 1), dim=0).shape[0])
                        if k == "planes" and cur[k].dtype == torch.int16 and cur[k].dim() == 4 and cur[k].shape[0] == 2:   # (2, nb16, nvox, 16) int16
                            vox = d.any(-1).nonzero()
                            ev["bad_voxel_rows"] = int(vox.shape[0])
                            rows = []
                            cf, rf = cur[k].view(torch.float16), ref[k].view(torch.float16)
                            plane_vox = cost.dims[1] * cost.dims[2] if False else None
                            npl = NVOX_PLANE[0]
                            for (pl, cb, vx) in vox[:10].tolist():
                                if pl != 0:
                                    continue
                                ch = d[pl, cb, vx].nonzero()[:, 0].tolist()
                                x_got = ((cf[0, cb, vx, ch].float() + cf[1, cb, vx, ch].float()) * 16).tolist()
                                x_want = ((rf[0, cb, vx, ch].float() + rf[1, cb, vx, ch].float()) * 16).tolist()
                                lo_same = bool((cf[1, cb, vx, ch] == rf[1, cb, vx, ch]).all())
                                nb = {}
                                for dz in (-1, 1, -2, 2, -3, 3):
                                    v2 = vx + dz * npl
                                    if 0 <= v2 < cf.shape[2]:
                                        nb[dz] = ((rf[0, cb, v2, ch].float() + rf[1, cb, v2, ch].float()) * 16).tolist()
                                hit = [dz for dz, vals in nb.items() if vals == x_got]
                                rows.append({"vox": vx, "z": vx // npl, "vox%16": vx % 16, "ch": [cb * 16 + c for c in ch], "x_got": x_got, "x_want": x_want,
                                             "lo_piece_unchanged": lo_same, "x_got_equals_reference_of_plane_offset": hit})
                            ev["rows"] = rows
                            ev["bad_rows_by_plane"] = [int((vox[:, 0] == 0).sum()), int((vox[:, 0] == 1).sum())]
                            ev["bad_rows_by_block"] = [int((vox[:, 1] == b).sum()) for b in range(3)]
                            ev["bad_rows_vox_mod_16"] = sorted(set((vox[:, 2] % 16).tolist()))
                        events.append(ev)
            del held, cur
        torch.cuda.synchronize()
        report[mode] = {"iters": iters, "mismatching": len(events), "seconds": round(time.time() - t0, 2), "events": [e for e in events if e["what"] != "raw"][:5], "guard_fallbacks": int(ops.guard_fallbacks()),
                        "by_key": {k: sum(1 for e in events if e["what"] == k) for k in ref}, "checksum": checksum}
    q.put((rank, report))


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    modes = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ALL
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    bar = ctx.Barrier(K)
    ps = [ctx.Process(target=body, args=(r, K, iters, modes, q, bar)) for r in range(K)]
    t0 = time.time()
    [p.start() for p in ps]
    res = sorted((q.get(timeout=900) for _ in ps), key=lambda r: r[0])
    [p.join() for p in ps]
    print(f"race_probe2: K={K} processes on one GPU, {iters} iterations per mode and process, {time.time() - t0:.1f} s")
    for rank, rep in res:
        if "error" in rep:
            print(f"process {rank} FAILED:\n{rep['error']}")
            continue
        for mode in modes:
            r = rep[mode]
            print(f"process {rank} {mode:8s}: {r['by_key']} {r['mismatching']} mismatching results in {r['iters']} iterations ({r['seconds']} s; guard fallbacks so far {r['guard_fallbacks']}; reference checksums {r['checksum']})")
            for ev in r["events"]:
                print("    ", ev)
