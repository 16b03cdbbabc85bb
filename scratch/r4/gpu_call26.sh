#!/bin/bash
# round 4, call 26: one-launch Adam
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_shared.py -q --tb=short -p no:cacheprovider -x > gpurun_out/c26_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c26_tests.log
tail -12 gpurun_out/c26_tests.log
timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c26_ab.txt
timeout 300 python scratch/r3/train_prof.py fp32 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c26_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c26_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c26_prof.log" 2>&1; echo "prof rc $?")
grep -i "adam\|multi_tensor" gpurun_out/c26_prof/amp_kernel_stats.csv | cut -c1-200
