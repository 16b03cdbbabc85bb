#!/bin/bash
o=gpurun_out/r5j; mkdir -p $o
python -m pytest tests/test_gpu_fp16x3_encoder.py tests/test_gpu_encoder.py tests/test_gpu_headline_parity.py tests/test_gpu_guard.py tests/test_gpu_configs45.py tests/test_gpu_raymarch.py::test_config1_end_to_end_vs_oracle -m gpu -q -s 2>&1 | grep -v "^$" > $o/pytest.log
grep -n "^E  .*Error\|passed\|failed\|^FAILED\|conv f16x3 tiled" $o/pytest.log | cut -c1-300 | tail -24
scratch/r5/prof_enc.sh r5j | head -8
python bench.py --cpu-batches 0 --no-extras > $o/bench.json 2> $o/bench.err
python - <<PY
import json
d=json.load(open("$o/bench.json"))
print("headline", d["value"], d["roofline"]["frac"], "encode", d["encode_ms"]["forward_free_running"], d["encode_ms"]["forward_single_call"], "stages", d["encode_ms"]["feature_net"], d["encode_ms"]["planesweep_costvar"], d["encode_ms"]["cost_reg_net"])
PY
