#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_shared.py -q --tb=short -p no:cacheprovider -s > gpurun_out/c38_shared.log 2>&1; echo "rc $?"
grep "NOTE\|passed\|failed" gpurun_out/c38_shared.log | cut -c1-300
