#!/bin/bash
o=gpurun_out/r5k; mkdir -p $o
python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_guard.py tests/test_gpu_raymarch.py tests/test_gpu_views.py -m gpu -q -s 2>&1 | grep -v "^$" > $o/pytest.log
grep -n "^E  .*Error\|passed\|failed\|^FAILED\|activations x" $o/pytest.log | cut -c1-300 | tail -24
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --cpu-batches 0 > $o/bench.json 2> $o/bench.err
python - <<PY
import json
d=json.load(open("$o/bench.json"))
print("headline", d["value"], d["roofline"]["frac"])
e=d["extras"]
print("guarded", e["guarded_default_mlp_mode"]["mlp_kernel_ms"], e["guarded_default_mlp_mode"]["rays_per_s"], e["guarded_default_mlp_mode"]["roofline"]["frac"], "fp16x3", e["fp16x3_mlp_mode"]["mlp_kernel_ms"], "frame", e["frame_512x640"]["seconds"], "fallbacks", e["guarded_default_mlp_mode"].get("guard_fallbacks"), e["frame_512x640"]["guard_fallbacks_during_the_four_frames"])
print("c4 frame", e["config4"]["frame_800x800_guarded_default_mlp"], "c5", e["config5"]["frame_guarded_default_mlp"])
PY
