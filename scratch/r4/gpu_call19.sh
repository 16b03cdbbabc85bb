#!/bin/bash
# round 4, call 19: kernel profiles of the fp32 training step and of the default encode on the current tree
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c19_prof_fp32" -o fp32 -- python "$GRAFT_REPO_ROOT/scratch/r3/train_prof.py" fp32 4 > "$GRAFT_REPO_ROOT/gpurun_out/c19_prof_fp32.log" 2>&1; echo "prof rc $?")
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c19_prof_enc" -o enc -- python "$GRAFT_REPO_ROOT/scratch/r3/enc_only.py" > "$GRAFT_REPO_ROOT/gpurun_out/c19_prof_enc.log" 2>&1; echo "prof rc $?")
tail -5 gpurun_out/c19_prof_fp32.log gpurun_out/c19_prof_enc.log
