#!/bin/bash
# round 4, call 20: the whole GPU suite + the default bench line on the current tree
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/c20_gputests.log 2>&1; echo "gpu tests rc $?" | tee -a gpurun_out/c20_gputests.log
tail -15 gpurun_out/c20_gputests.log
timeout 900 python bench.py > gpurun_out/c20_bench.json 2> gpurun_out/c20_bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/c20_bench.json
