#!/bin/bash
# round 4, call 17: LDS-tiled bf16 conv for conv1 / conv2 (+ the data gradients of conv2 / conv11)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py -q --tb=short -p no:cacheprovider -x -k "layer" > gpurun_out/c17_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c17_tests.log
tail -15 gpurun_out/c17_tests.log
timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c17_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c17_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c17_prof.log" 2>&1; echo "prof rc $?")
grep "conv_bf16" gpurun_out/c17_prof/amp_kernel_stats.csv | cut -c1-200
