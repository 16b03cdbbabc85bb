"""Shared-GPU mismatch of the plane sweep: WHAT differs when K processes share the GPU.
Each process repeats {FeatureNet -> blocked fp32 plane sweep} and compares the cost volume bit for bit with its own reference, in four modes that
run one after the other (the processes move from mode to mode together):
  base      as the product runs it
  prefill   the cost buffer is filled with 0xFFFFFFFF before the sweep          -> a voxel that still holds the pattern is a LOST STORE
  poison    FeatureNet's output / the thumbnails of the previous iteration are filled with NaN before they are freed (the next iteration's
            allocations reuse the blocks) -> a NaN in the cost volume is a STALE READ of the sweep's inputs (or a launch that overtook the fill)
  syncpre   a host synchronisation between FeatureNet and the sweep              -> mismatches gone = inter-kernel visibility / ordering
For every mismatching sweep: bad voxels (count, z range, a sample with got / want), how many hold the prefill pattern or NaN, whether the inputs
still equal the reference inputs, and whether re-running the sweep on the same inputs gives the reference.
usage: race_probe.py [K] [iters per mode]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import torch
import torch.multiprocessing as mp

MODES = ("base", "prefill", "poison", "syncpre")


def body(rank, K, iters, q, bar):
    try:
        _body(rank, K, iters, q, bar)
    except BaseException as e:                                       # the parent must not wait for a dead child
        import traceback
        q.put((rank, {"error": traceback.format_exc()[-1500:]}))
        try:
            bar.abort()
        except Exception:
            pass


def _body(rank, K, iters, q, bar):
    import numpy as np
    from mvsnerf_amd import train, encoder, _lib
    from mvsnerf_amd._lib import check, stream_ptr
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.lib()
    args = train.default_args(pad=24, batch_size=1024, N_samples=128, chunk=1024)
    system = train.MVSSystem(args).to(dev)
    z = np.load('tests/golden/mvsnerf_v0_weights.npz')
    system.MVSNet.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mvs/")})
    batch = train.batch_to_device(train.synthetic_batch(512, 640, seed=1234), dev)
    net = system.MVSNet
    data_mvs, _ = system.decode_batch(dict(batch))
    imgs, proj_mats, nf = data_mvs["images"][:, :3], data_mvs["proj_mats"][:, :3], data_mvs["near_fars"][0, 0]
    B, V, _, Hi, Wi = imgs.shape
    pad, D, C = 24, net.D, 32
    t_vals = torch.linspace(0.0, 1.0, steps=D, device=dev)
    depth = (nf[0] * (1.0 - t_vals) + nf[1] * t_vals).contiguous()
    proj = proj_mats[0].contiguous()
    net.prepack()

    def featnet():
        with torch.no_grad():
            f = net.feature(imgs.reshape(B * V, 3, Hi, Wi))
        fc, ld = encoder._images_channel_last(f, C)
        assert ld == C
        return fc                                                    # (V, H, W, 32) channel-last buffer (zero-copy view of FeatureNet's output)

    f0 = featnet()
    _, H, W, _ = f0.shape
    Hp, Wp = H + 2 * pad, W + 2 * pad
    nvox = D * Hp * Wp
    n_ch = 3 * V + C
    CP = (n_ch + 3) // 4 * 4

    def thumbs():
        small = torch.empty((V, 3, H, W), device=dev)
        check(lib.mvsnerf_resize_bilinear(imgs[0].contiguous().data_ptr(), small.data_ptr(), V * 3, Hi, Wi, H, W, stream_ptr()), "resize")
        cl = torch.empty((V, H, W, 4), device=dev)
        check(lib.mvsnerf_nchw_to_nhwc(small.data_ptr(), cl.data_ptr(), V, 3, H, W, 4, stream_ptr()), "nhwc")
        return cl

    def sweep(fc, icl, cost, masks):
        check(lib.mvsnerf_planesweep_costvar_blocked_fwd(fc.data_ptr(), icl.data_ptr(), proj.data_ptr(), depth.data_ptr(), V, C, H, W, D, pad,
                                                         cost.data_ptr(), CP, masks.data_ptr(), 1, stream_ptr()), "sweep")

    def new_cost():
        return torch.empty((CP // 4, nvox, 4), device=dev), torch.empty((V, D, Hp, Wp), device=dev)

    pk = net.cost_reg_2.conv0._packed
    w_c8 = pk.get_c8()

    def conv0(cost):                                                  # the fp32 matrix-core conv0 on the blocked cost volume (what encoder_precision("fp32") runs)
        raw = torch.empty((D, Hp, Wp, 8), device=dev)
        check(lib.mvsnerf_conv3d_c8_blocked_fwd(cost.data_ptr(), pk.cin_pad, pk.cin, D, Hp, Wp, w_c8.data_ptr(), raw.data_ptr(), stream_ptr()), "conv0")
        return raw

    # reference: a result that a second pass reproduces
    ref = None
    for _ in range(4):
        fc, icl = featnet(), thumbs()
        cost, masks = new_cost()
        sweep(fc, icl, cost, masks)
        if ref is not None and torch.equal(cost.view(torch.int32), ref[0].view(torch.int32)) and torch.equal(fc, ref[2]):
            break
        ref = (cost, masks, fc.clone(), icl.clone())
    ref_cost_i = ref[0].view(torch.int32)
    ref_raw = conv0(ref[0])
    for _ in range(3):
        assert torch.equal(conv0(ref[0]), ref_raw), "conv0 of the reference cost volume does not reproduce"
    torch.cuda.synchronize()
    report = {}
    for mode in MODES:
        bar.wait()
        events, conv0_only, t0 = [], [], time.time()
        WIN = 8                                                      # the host looks at the comparison flags once per WIN sweeps: the queue stays full
        for w0 in range(0, iters, WIN):
            held = []
            for it in range(w0, min(iters, w0 + WIN)):
                fc, icl = featnet(), thumbs()
                cost, masks = new_cost()
                if mode == "prefill":
                    cost.view(torch.int32).fill_(-1)
                if mode == "syncpre":
                    torch.cuda.synchronize()
                sweep(fc, icl, cost, masks)
                raw = conv0(cost)
                held.append((it, fc, icl, cost, masks, (cost.view(torch.int32) != ref_cost_i).any(), raw, (raw != ref_raw).any()))
            flags = torch.stack([h[5] for h in held]).tolist()       # one synchronisation per window
            flags_raw = torch.stack([h[7] for h in held]).tolist()
            for (it, fc, icl, cost, masks, _, raw, _), bad, bad_raw in zip(held, flags, flags_raw):
                if bad_raw and not bad:                               # conv0 differs although its input is the reference cost volume bit for bit
                    dv = (raw != ref_raw).any(-1).nonzero()
                    conv0_only.append({"it": it, "bad_conv0_voxels": int(dv.shape[0]), "z": (int(dv[:, 0].min()), int(dv[:, 0].max())),
                                       "tiles_4x8x16": int(torch.unique(torch.stack([dv[:, 0] // 4, dv[:, 1] // 8, dv[:, 2] // 16], 1), dim=0).shape[0]),
                                       "conv0_rerun_equals_reference": bool(torch.equal(conv0(cost), ref_raw))})
                if not bad:
                    continue
                ci = cost.view(torch.int32)
                neq = (ci != ref_cost_i)                              # (CP/4, nvox, 4)
                badvox = neq.any(2).any(0).nonzero()[:, 0]
                zs = torch.div(badvox, Hp * Wp, rounding_mode="floor")
                rem = badvox - zs * Hp * Wp
                ys = torch.div(rem, Wp, rounding_mode="floor"); xs = rem - ys * Wp
                ev = {"it": it, "bad_voxels": int(badvox.numel()), "bad_words": int(neq.sum()),
                      "words_holding_prefill_pattern": int((neq & (ci == -1)).sum()), "nan_words": int((neq & torch.isnan(cost)).sum()),
                      "z_values": sorted(set(zs.tolist()))[:40], "y_range": (int(ys.min()), int(ys.max())), "x_range": (int(xs.min()), int(xs.max())),
                      "blocks_bad_per_voxel_max": int(neq.any(2).sum(0).max()),
                      "inputs_equal_reference": bool(torch.equal(fc, ref[2]) and torch.equal(icl, ref[3])),
                      "masks_equal_reference": bool(torch.equal(masks, ref[1]))}
                smp = []
                for v in badvox[:6].tolist():
                    cb = int(neq[:, v].any(1).nonzero()[0, 0])
                    smp.append(((v // (Hp * Wp), (v % (Hp * Wp)) // Wp, v % Wp), cb, [float(t) for t in cost[cb, v].tolist()], [float(t) for t in ref[0][cb, v].tolist()]))
                ev["sample (z,y,x), block, got, want"] = smp
                c2, m2 = new_cost()
                sweep(fc, icl, c2, m2)                                # the same inputs again, on an idle queue
                ev["rerun_equals_reference"] = bool(torch.equal(c2.view(torch.int32), ref_cost_i))
                ev["conv0_output_differs_too"] = bool(bad_raw)
                del c2, m2
                events.append(ev)
            if mode == "poison":
                for (_, fc, icl, *_r) in held:                       # the blocks go back to the allocator holding NaN
                    fc.fill_(float("nan")); icl.fill_(float("nan"))
            del held, fc, icl, cost, masks, raw
        torch.cuda.synchronize()
        report[mode] = {"sweeps": iters, "mismatching": len(events), "seconds": round(time.time() - t0, 2), "events": events[:4],
                        "conv0_differs_on_a_good_cost_volume": len(conv0_only), "conv0_events": conv0_only[:4]}
    q.put((rank, report))


if __name__ == "__main__":
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    bar = ctx.Barrier(K)
    ps = [ctx.Process(target=body, args=(r, K, iters, q, bar)) for r in range(K)]
    t0 = time.time()
    [p.start() for p in ps]
    res = sorted(q.get(timeout=900) for _ in ps)
    [p.join() for p in ps]
    print(f"race_probe: K={K} processes on one GPU, {iters} sweeps per mode and process, {time.time() - t0:.1f} s")
    for rank, rep in res:
        if "error" in rep:
            print(f"process {rank} FAILED:\n{rep['error']}")
            continue
        for mode in MODES:
            r = rep[mode]
            print(f"process {rank} {mode:8s}: {r['mismatching']} of {r['sweeps']} sweeps differ ({r['seconds']} s)")
            for ev in r["events"]:
                print("    ", ev)
            print(f"          conv0 (fp32 matrix-core kernel) differing on a bit-exact cost volume: {r['conv0_differs_on_a_good_cost_volume']}")
            for ev in r["conv0_events"]:
                print("    ", ev)
