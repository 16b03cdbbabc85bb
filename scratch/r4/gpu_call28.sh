#!/bin/bash
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
timeout 600 python scratch/r4/train_host_time.py amp 2>&1 | grep -v amdgpu.ids > gpurun_out/c28_host.txt; head -60 gpurun_out/c28_host.txt | cut -c1-200
