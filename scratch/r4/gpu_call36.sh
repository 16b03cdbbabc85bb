#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 900 python scratch/r4/race_hunt.py 3 40 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/c36_race.txt | cut -c1-700
