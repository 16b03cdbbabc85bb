#!/bin/bash
# hunt the intermittent "2-rank frame != 1-rank frame" of the shared-GPU dry run: up to eight dry runs, stop at the first failure
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python bench.py --gpus 2 --shared-gpu-dry-run > gpurun_out/c35_dry_$i.json 2> gpurun_out/c35_dry_$i.err; rc=$?
  echo "dry run $i rc $rc"
  if [ $rc -ne 0 ]; then grep "differs" gpurun_out/c35_dry_$i.err | cut -c1-600; break; fi
done
