#!/bin/bash
# round 4, call 6: transposed bf16 layers with both x parities in the columns, batched bf16 weight packs; use_amp step time and kernel trace
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py tests/test_gpu_bf16_encoder.py tests/test_gpu_train.py -q --tb=short -p no:cacheprovider > gpurun_out/c6_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c6_tests.log
tail -25 gpurun_out/c6_tests.log
timeout 300 python scratch/r3/train_prof.py amp 5 > gpurun_out/c6_train_amp.txt 2>&1; tail -1 gpurun_out/c6_train_amp.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c6_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 7 > "$GRAFT_REPO_ROOT/gpurun_out/c6_prof.log" 2>&1; echo "prof rc $?"
