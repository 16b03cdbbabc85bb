#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_shared.py -q --tb=long -p no:cacheprovider > gpurun_out/c32_shared_$i.log 2>&1; echo "run $i rc $?"; tail -3 gpurun_out/c32_shared_$i.log | cut -c1-200
done
