#!/bin/bash
# round 4, call 1: the whole GPU suite on the guarded defaults (no -x: the full list of what the new default moves), the default bench line,
# the host-path diagnostics of a ray-march step, and the batched entry
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/c1_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c1_tests.log
tail -5 gpurun_out/c1_tests.log
timeout 600 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; echo "bench rc $?"
timeout 300 python scratch/r4/step_diag.py > gpurun_out/c1_step_diag.txt 2>&1; echo "diag rc $?"
tail -c 1500 gpurun_out/c1_bench.json
