#!/bin/bash
# round 4, call 3: zfast gather with the (row, channel half) lane mapping; configs 4/5 measured errors; bench
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_layout.py tests/test_gpu_encoder.py tests/test_gpu_raymarch.py tests/test_gpu_configs45.py tests/test_gpu_guard.py -q --tb=short -p no:cacheprovider > gpurun_out/c3_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c3_tests.log
tail -8 gpurun_out/c3_tests.log
timeout 600 python bench.py > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err; echo "bench rc $?"
