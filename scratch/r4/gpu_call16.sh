#!/bin/bash
# round 4, call 16: vectorised partial sums, small pipelined wgrad tiles adopted
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_bf16_layers.py tests/test_gpu_featnet.py -q --tb=short -p no:cacheprovider -k "partial_sum or wgrad" > gpurun_out/c16_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c16_tests.log
tail -6 gpurun_out/c16_tests.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -q --tb=short -p no:cacheprovider > gpurun_out/c16_tests2.log 2>&1; echo "tests2 rc $?" | tee -a gpurun_out/c16_tests2.log
tail -6 gpurun_out/c16_tests2.log
timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c16_ab.txt
timeout 300 python scratch/r3/train_prof.py fp32 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c16_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c16_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c16_prof.log" 2>&1; echo "prof rc $?")
grep "partial_sum" gpurun_out/c16_prof/amp_kernel_stats.csv | cut -c1-200
