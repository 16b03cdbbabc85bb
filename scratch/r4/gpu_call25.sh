#!/bin/bash
# round 4, call 25: InPlaceABN backward of the small layers as one launch (device-wide barrier)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_backward.py -q --tb=short -p no:cacheprovider -x -k "abn_bwd_fused" > gpurun_out/c25_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c25_tests.log
tail -8 gpurun_out/c25_tests.log
echo skip

timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c25_ab.txt
timeout 300 python scratch/r3/train_prof.py fp32 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c25_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c25_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c25_prof.log" 2>&1; echo "prof rc $?")
grep "abn_bwd" gpurun_out/c25_prof/amp_kernel_stats.csv | cut -c1-200
