#!/bin/bash
# round 4, call 11: bf16 wgrad, 8-channel X rows + software-pipelined tile loop; three variants A/B (MVS_WGRAD_AB = 0 plain, 1 pipelined at one wave/SIMD, 2 pipelined at two)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 2 1 0; do
  MVS_WGRAD_AB=$v timeout 600 python -m pytest tests/test_gpu_bf16_layers.py -q --tb=short -p no:cacheprovider -k "wgrad" > gpurun_out/c11_tests_v$v.log 2>&1; echo "variant $v tests rc $?" | tee -a gpurun_out/c11_tests_v$v.log
  tail -3 gpurun_out/c11_tests_v$v.log
done
for v in 0 1 2 0 1 2; do
  echo "== MVS_WGRAD_AB=$v" >> gpurun_out/c11_ab.txt
  MVS_WGRAD_AB=$v timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c11_ab.txt
done
echo "== bf16 wgrad off" >> gpurun_out/c11_ab.txt
MVS_BF16_WGRAD=0 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c11_ab.txt
cat gpurun_out/c11_ab.txt
for v in 1 2; do
  (cd /tmp && MVS_WGRAD_AB=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c11_prof_v$v" -o amp -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" amp 4 > "$GRAFT_REPO_ROOT/gpurun_out/c11_prof_v$v.log" 2>&1; echo "prof $v rc $?")
  grep "wgrad_bf16" "gpurun_out/c11_prof_v$v/amp_kernel_stats.csv" | cut -c1-220
done
