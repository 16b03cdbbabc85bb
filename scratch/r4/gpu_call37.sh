#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for k in 1 1 2; do
  echo "== K=$k" | tee -a gpurun_out/c37_race.txt
  timeout 600 python scratch/r4/race_hunt.py $k 60 2>&1 | grep -v "amdgpu.ids" | cut -c1-400 | tee -a gpurun_out/c37_race.txt
done
