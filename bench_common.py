"""bench_common.py - constants and timing helpers shared by bench.py (the headline line) and bench_extras.py (every other leg)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS, N_SAMPLES, PAD, D_PLANES = 1024, 128, 24, 128
H_IMG, W_IMG, N_SRC = 512, 640, 3
FLOP_PER_SAMPLE = 251392          # Renderer_ours MACs*2
VOL_BYTES_PER_SAMPLE = 300        # 8 corners x 32 B + 12 B coord + 32 B out
COL_BYTES_PER_SAMPLE = 204        # 3 views x 48 B + 12 B + 48 B out
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
PEAK_16BIT_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 matrix peak (v_mfma_f32_32x32x16_{bf16,f16})
PMC_FILE = "profiles/r06_pmc_summary.json"


def csrc_sha16():
    """Hash of the kernel sources: the committed PMC summary carries the hash of the tree it was measured on (`_csrc_sha16`), and
    `traffic` is reported only while it still matches - a kernel change can no longer keep an old traffic number alive silently."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mvsnerf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def _pmc_summary():
    try:
        d = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except Exception:
        return {}, "no PMC summary committed"
    if d.get("_csrc_sha16") != csrc_sha16():
        return {}, f"{PMC_FILE} was measured on other kernel sources (csrc hash {d.get('_csrc_sha16')} != {csrc_sha16()}): traffic withheld as stale"
    return d, PMC_FILE + " (rocprofv3 --pmc passes of this command on these kernel sources, committed; not re-measured in this run)"


def _load_pmc_traffic():
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE collected in
    separate --pmc passes over this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950's
    16-B/lane reads; counters are in KiB).  bench.py cannot run rocprof on itself, so `traffic` cites that measurement."""
    d, _ = _pmc_summary()
    out = {}
    for k, v in d.items():
        if isinstance(v, dict) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[k] = int((2 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024)
    return out


def pmc_mfma_busy_frac(kernel_prefix):
    """Fraction of the kernel's duration the matrix pipes were busy, from the committed PMC passes:
    (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)."""
    d, _ = _pmc_summary()
    for k, v in d.items():
        if isinstance(v, dict) and k.startswith(kernel_prefix) and "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            return round((v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0) / (v["GRBM_GUI_ACTIVE"]["mean"] / 8.0), 4)
    return None


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the first profiled kernel whose name starts with `kernel_prefix` (None if not profiled)."""
    for k, v in PMC_TRAFFIC.items():
        if k.startswith(kernel_prefix):
            return v
    return None


PMC_TRAFFIC = _load_pmc_traffic()


def load_mlp_weights():
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "mvsnerf_v0_weights.npz"))
    return {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("mlp/")}


def settle(fn, ms, indexed=False):
    """Untimed launches of fn for >= `ms` of wall time.  The GPU leaves its idle power state over the first ~25 ms of sustained
    load (scratch/ramp.py: the MLP kernel runs 0.288 -> 0.239 ms/launch over the first 100 launches after a 1 s pause), so
    every timed region is preceded by this much of the same work; what is reported is the steady state."""
    if ms <= 0:
        return
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            fn(i) if indexed else fn()
            i += 1
        torch.cuda.synchronize()


def event_time(fn, iters, warm=3, graph_batch=0, settle_ms=40):
    """Average duration (ms) of one call of fn, HIP events on the current (= launch) stream.
    graph_batch > 0: the launches are captured into a hipGraph of `graph_batch` back-to-back calls and replayed, so that
    kernels of a few microseconds are not timed through the ~8 us of Python/ctypes launch overhead (the ~1.5 us
    dependent-launch boundary between kernels remains included)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if graph_batch:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(graph_batch):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        settle(g.replay, settle_ms)
        reps = max(1, iters // graph_batch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * graph_batch)
    settle(fn, settle_ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def sustained_clock_ghz(run_mlp_census, n_wg, dev):
    """Shader clock while the MLP kernel runs: every workgroup records s_memtime ticks and the 100 MHz wall clock
    (mvsnerf_mlp_fwd_census: the same launch with a caller-owned timing record)."""
    cen = torch.zeros((n_wg, 16), dtype=torch.int64, device=dev)
    run_mlp_census(cen)
    torch.cuda.synchronize()
    c = cen.cpu().numpy().astype("float64")
    dur_us = (c[:, 1] - c[:, 0]) / 100.0
    ok = dur_us > 0
    return float((c[ok, 13] / dur_us[ok] / 1e3).mean())


def _ms_events(fn, iters=5, warm=2):
    """Mean duration (ms) of fn() with HIP events on the current stream, after `warm` untimed calls."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _psnr(a, b):
    import math
    return round(10 * math.log10(1.0 / max(float(((a.double() - b.double()) ** 2).mean()), 1e-30)), 1)
