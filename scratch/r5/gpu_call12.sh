#!/bin/bash
# anti-phase fp16x3 MLP kernel (MVS_H3_AP=1) against the shipped one: parity tests, then the bench's MLP-mode legs
o=gpurun_out/r5l; mkdir -p $o
MVS_H3_AP=1 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_guard.py tests/test_gpu_raymarch.py tests/test_gpu_views.py tests/test_gpu_encoder.py -m gpu -q -x 2>&1 | grep -v "^$" > $o/pytest_ap.log
grep -n "^E  .*Error\|passed\|failed\|^FAILED" $o/pytest_ap.log | cut -c1-300 | tail -12
for ap in 0 1; do
  MVS_H3_AP=$ap python bench.py --cpu-batches 0 > $o/bench_ap$ap.json 2> $o/bench_ap$ap.err
  python - <<PY
import json
d=json.load(open("$o/bench_ap$ap.json"))
e=d["extras"]
print("AP=$ap headline", d["value"], "guarded", e["guarded_default_mlp_mode"]["mlp_kernel_ms"], e["guarded_default_mlp_mode"]["rays_per_s"], e["guarded_default_mlp_mode"]["roofline"]["frac"], "fp16x3", e["fp16x3_mlp_mode"]["mlp_kernel_ms"], e["fp16x3_mlp_mode"]["rays_per_s"], "frame", e["frame_512x640"]["seconds"], "c5", e["config5"]["frame_guarded_default_mlp"]["seconds"], "sigma diff vs fp32", e["fp16x3_mlp_mode"]["max_abs_sigma_diff_vs_fp32_kernel"])
PY
done
