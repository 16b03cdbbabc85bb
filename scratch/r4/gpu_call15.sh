#!/bin/bash
# round 4, call 15: small pipelined tiles for the 16-channel 3-D weight gradients (MVS_WGRAD_AB=1) vs the 4x6 / 2x2 tiles; bf16-slot emulation test
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py -q --tb=short -p no:cacheprovider -k "bf16_training" > gpurun_out/c15_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c15_tests.log
tail -8 gpurun_out/c15_tests.log
for v in 1 0; do
  MVS_WGRAD_AB=$v timeout 600 python -m pytest tests/test_gpu_bf16_layers.py -q --tb=short -p no:cacheprovider -k "wgrad" > gpurun_out/c15_tests_v$v.log 2>&1; echo "variant $v tests rc $?" | tee -a gpurun_out/c15_tests_v$v.log
  tail -3 gpurun_out/c15_tests_v$v.log
done
for v in 0 1 0 1; do
  echo "== MVS_WGRAD_AB=$v" >> gpurun_out/c15_ab.txt
  MVS_WGRAD_AB=$v timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c15_ab.txt
done
cat gpurun_out/c15_ab.txt
(cd /tmp && MVS_WGRAD_AB=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c15_prof" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c15_prof.log" 2>&1; echo "prof rc $?")
grep "wgrad_bf16" gpurun_out/c15_prof/amp_kernel_stats.csv | cut -c1-200
