#!/bin/bash
o=gpurun_out/r5g; mkdir -p $o
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fp16x3_encoder.py tests/test_gpu_headline_parity.py tests/test_gpu_guard.py tests/test_gpu_bf16_encoder.py tests/test_gpu_bf16_layers.py tests/test_gpu_shared.py tests/test_gpu_configs45.py tests/test_gpu_train.py -m gpu -q -s 2>&1 | grep -v "^$" > $o/pytest.log
grep -n "^E  .*Error\|passed\|failed\|^FAILED\|two guarded" $o/pytest.log | cut -c1-500 | tail -20
scratch/r5/prof_enc.sh r5g
python bench.py --cpu-batches 0 > $o/bench.json 2> $o/bench.err
python - <<PY
import json
d=json.load(open("$o/bench.json"))
print("headline", d["value"], d["roofline"]["frac"], "encode", d["encode_ms"]["forward_free_running"], "fp32", d["encode_ms_fp32_conv0"]["forward_free_running"], "bf16", d["encode_ms_bf16_conv0"]["forward_free_running"])
for k in ("frame_512x640","train_step","train_step_bf16"):
    print(k, {kk:v for kk,v in d["extras"][k].items() if kk!="note"})
PY
