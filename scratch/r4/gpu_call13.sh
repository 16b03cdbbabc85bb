#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python scratch/r4/fill_copy_census.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" > gpurun_out/c13_census.txt
cat gpurun_out/c13_census.txt
