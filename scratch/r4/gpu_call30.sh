#!/bin/bash
# round 4, call 30: plane-sweep backward: v_rcp for 1 / count, two compares per tap set
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py -q --tb=short -p no:cacheprovider -x -s -k "planesweep" > gpurun_out/c30_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c30_tests.log
grep "planesweep bwd\|passed\|failed" gpurun_out/c30_tests.log | tail -12
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_bf16_encoder.py -q --tb=short -p no:cacheprovider -x > gpurun_out/c30_tests2.log 2>&1; echo "tests2 rc $?"; tail -2 gpurun_out/c30_tests2.log
timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c30_ab.txt
