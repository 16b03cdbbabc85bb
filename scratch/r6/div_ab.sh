#!/bin/bash
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2 3; do
  python scratch/r6/h3_ab.py fp16x3
  MVS_LIB=scratch/lib/libmvsnerf_hip_prep.so python scratch/r6/h3_ab.py fp16x3
done 2>&1 | grep -v amdgpu.ids > $out/ab.txt
cat $out/ab.txt
timeout 1200 python -m pytest tests/test_gpu_raymarch.py tests/test_gpu_views.py tests/test_gpu_layout.py tests/test_gpu_fp16x3.py tests/test_gpu_guard.py tests/test_gpu_signatures.py tests/test_gpu_configs45.py tests/test_gpu_importance.py -x -q 2>&1 | tail -3 > $out/tests.txt
cat $out/tests.txt
python bench.py --cpu-batches 0 > $out/bench.json 2>/dev/null
python - "$out/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: v for k, v in d.items() if k in ("value", "default_step_ms", "default_mlp_kernel_ms", "fp16x3_mlp_frac", "frame_512x640_ms", "config5_default_frame_ms", "config4_default_frame_ms")})
PY
