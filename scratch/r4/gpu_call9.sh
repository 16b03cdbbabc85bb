#!/bin/bash
# round 4, call 9: bf16 weight gradients (csrc/wgrad_bf16.hip); tests + same-box A/B of the use_amp step (wgrad bf16 on/off)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_layers.py -q --tb=short -p no:cacheprovider -k "wgrad or featurenet or node" > gpurun_out/c9_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c9_tests.log
tail -40 gpurun_out/c9_tests.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_bf16_encoder.py -q --tb=short -p no:cacheprovider > gpurun_out/c9_tests2.log 2>&1; echo "tests rc $?" >> gpurun_out/c9_tests2.log
tail -8 gpurun_out/c9_tests2.log
MVS_BF16_WGRAD=0 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c9_ab.txt
MVS_BF16_WGRAD=1 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c9_ab.txt
cat gpurun_out/c9_ab.txt
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c9_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 5 > "$GRAFT_REPO_ROOT/gpurun_out/c9_prof.log" 2>&1; echo "prof rc $?"
