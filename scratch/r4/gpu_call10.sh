#!/bin/bash
# round 4, call 10: bf16 wgrad with batched staging loads and the 2x2x32 stride-2 tile; A/B
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py -q --tb=short -p no:cacheprovider -k "wgrad" > gpurun_out/c10_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c10_tests.log
tail -12 gpurun_out/c10_tests.log
MVS_BF16_WGRAD=0 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c10_ab.txt
MVS_BF16_WGRAD=1 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c10_ab.txt
cat gpurun_out/c10_ab.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c10_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 5 > "$GRAFT_REPO_ROOT/gpurun_out/c10_prof.log" 2>&1; echo "prof rc $?"
grep "wgrad" "$GRAFT_REPO_ROOT/gpurun_out/c10_prof/amp_kernel_stats.csv" | cut -c1-200
