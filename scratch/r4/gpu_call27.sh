#!/bin/bash
# round 4, call 27: FeatureNet's full- and half-resolution 3x3 layers on the LDS-tiled bf16 kernel
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py tests/test_gpu_train.py -q --tb=short -p no:cacheprovider -x > gpurun_out/c27_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c27_tests.log
tail -8 gpurun_out/c27_tests.log
timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c27_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c27_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c27_prof.log" 2>&1; echo "prof rc $?")
grep "conv_bf16" gpurun_out/c27_prof/amp_kernel_stats.csv | cut -c1-170
