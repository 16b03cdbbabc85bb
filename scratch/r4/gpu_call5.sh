#!/bin/bash
# round 4, call 5: conv1..conv11 on the bf16 matrix cores (csrc/conv3d_bf16.hip) - layer tests, the bf16 / training files, kernel trace of the use_amp
# training step, the torch.profiler listing of the small ATen launches of a step
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py -q --tb=short -p no:cacheprovider -x > gpurun_out/c5_layers.log 2>&1; echo "layers rc $?" >> gpurun_out/c5_layers.log
tail -30 gpurun_out/c5_layers.log
timeout 900 python -m pytest tests/test_gpu_bf16_encoder.py tests/test_gpu_train.py tests/test_gpu_backward.py -q --tb=short -p no:cacheprovider > gpurun_out/c5_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c5_tests.log
tail -8 gpurun_out/c5_tests.log
timeout 300 python scratch/r3/train_prof.py amp 5 > gpurun_out/c5_train_amp.txt 2>&1; tail -1 gpurun_out/c5_train_amp.txt
timeout 300 python scratch/r3/train_prof.py fp32 5 > gpurun_out/c5_train_fp32.txt 2>&1; tail -1 gpurun_out/c5_train_fp32.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c5_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 7 > "$GRAFT_REPO_ROOT/gpurun_out/c5_prof.log" 2>&1; echo "prof rc $?"
cd "$GRAFT_REPO_ROOT"; timeout 300 python scratch/r3/torch_prof.py > gpurun_out/c5_torch_prof.txt 2>&1; echo "torch prof rc $?"
