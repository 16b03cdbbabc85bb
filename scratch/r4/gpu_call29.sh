#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train.py -q --tb=short -p no:cacheprovider -x -k "adam or fit_steps or fused_optimizer" > gpurun_out/c29_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/c29_tests.log
timeout 600 python scratch/r4/train_host_time.py amp 2>&1 | grep -v amdgpu.ids | head -8
timeout 900 python bench.py > gpurun_out/c29_bench.json 2> gpurun_out/c29_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c29_bench.json') if l.startswith('{')][-1])
ex=d['extras']
for k in ('train_step','train_step_bf16','frame_512x640'):
    print(k,{kk:vv for kk,vv in ex[k].items() if kk in ('ms','ms_all_reps','seconds')})
print(d['value'], d['ms_per_step'])
PY
