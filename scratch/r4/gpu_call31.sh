#!/bin/bash
# same-box A/B: the committed library against the plane-sweep backward with v_rcp + two compares
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for v in new head new head; do
  echo "== $v" | tee -a gpurun_out/c31_ab.txt
  if [ $v = head ]; then export MVS_LIB=$GRAFT_REPO_ROOT/mvsnerf_amd/lib/libmvsnerf_hip_head.so; else unset MVS_LIB; fi
  timeout 300 python scratch/r3/train_prof.py amp 10 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/c31_ab.txt
done
