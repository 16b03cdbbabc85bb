#!/bin/bash
# round 4, call 14: bf16 slot storage for the MLP training path (forward save, dgrad, wgrad), deterministic plane-sweep backward
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py -q --tb=short -p no:cacheprovider -x > gpurun_out/c14_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c14_tests.log
tail -25 gpurun_out/c14_tests.log
timeout 900 python -m pytest tests/test_gpu_shared.py -q --tb=short -p no:cacheprovider -x -k two_ranks > gpurun_out/c14_shared.log 2>&1; echo "shared rc $?" | tee -a gpurun_out/c14_shared.log
tail -5 gpurun_out/c14_shared.log
timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c14_ab.txt
timeout 300 python scratch/r3/train_prof.py fp32 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c14_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c14_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c14_prof.log" 2>&1; echo "prof rc $?")
grep "mlp_" gpurun_out/c14_prof/amp_kernel_stats.csv | cut -c1-160
