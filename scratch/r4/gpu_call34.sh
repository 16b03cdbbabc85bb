#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_shared.py -q --tb=short -p no:cacheprovider -s > gpurun_out/c34_shared.log 2>&1; echo "rc $?"
grep "^rank [01]:\|passed\|failed" gpurun_out/c34_shared.log | cut -c1-330
rocm-smi --showproductname 2>/dev/null | grep -i "card series\|GUID" | head -2; nproc; grep -m1 "model name" /proc/cpuinfo
