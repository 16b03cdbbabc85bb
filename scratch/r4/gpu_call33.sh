#!/bin/bash
# the whole GPU suite twice (the shared-GPU tests failed once in the full suite, never alone): diagnostics
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
for i in 1 2; do
  timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > gpurun_out/c33_full_$i.log 2>&1; echo "full run $i rc $?"
  grep "passed\|failed" gpurun_out/c33_full_$i.log | tail -2
  grep "^rank [01]:" gpurun_out/c33_full_$i.log | cut -c1-600
done
ls gpurun_out/shared_dry_run_failure.txt 2>/dev/null && tail -40 gpurun_out/shared_dry_run_failure.txt | cut -c1-300
