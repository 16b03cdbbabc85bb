#!/bin/bash
# round 4: the whole GPU suite, then the evidence run (scratch/r4/final_profiles.sh), on the final tree
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/final_gputests.log 2>&1; echo "gpu tests rc $?" | tee -a gpurun_out/final_gputests.log
tail -8 gpurun_out/final_gputests.log
timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/final_smoke.log
bash scratch/r4/final_profiles.sh > gpurun_out/final_profiles.log 2>&1; echo "profiles rc $?"
tail -c 300 gpurun_out/r4_final/r04_bench.json; cat gpurun_out/r4_final/r04_train_step_ms.txt
