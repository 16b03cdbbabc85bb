#!/bin/bash
# round 4, call 8: FeatureNet on the bf16 kernels; tests + same-box A/B of the use_amp step
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py tests/test_gpu_bf16_encoder.py tests/test_gpu_train.py tests/test_gpu_featnet.py -q --tb=short -p no:cacheprovider > gpurun_out/c8_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c8_tests.log
tail -30 gpurun_out/c8_tests.log
MVS_BF16_LAYERS=0 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c8_ab.txt
MVS_BF16_LAYERS=1 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c8_ab.txt
cat gpurun_out/c8_ab.txt
