#!/bin/bash
# round 4, call 7: same-box A/B of the use_amp training step with conv1..conv11 on bf16 (new) vs fp32 (round 3) kernels
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
  MVS_BF16_LAYERS=0 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c7_ab.txt
  MVS_BF16_LAYERS=1 timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/c7_ab.txt
done
cat gpurun_out/c7_ab.txt
