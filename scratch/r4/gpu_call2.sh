#!/bin/bash
# round 4, call 2: depth-fastest volume layout - layout tests first, then the whole suite, the bench line and a kernel trace of the bench
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_layout.py -q --tb=short -p no:cacheprovider > gpurun_out/c2_layout.log 2>&1; echo "layout rc $?" >> gpurun_out/c2_layout.log
tail -15 gpurun_out/c2_layout.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_gpu_layout.py > gpurun_out/c2_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c2_tests.log
tail -8 gpurun_out/c2_tests.log
timeout 600 python bench.py > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err; echo "bench rc $?"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c2_prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-extras --cpu-batches 0 > "$GRAFT_REPO_ROOT/gpurun_out/c2_bench_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/c2_prof.err"; echo "prof rc $?"
cd "$GRAFT_REPO_ROOT"; find gpurun_out/c2_prof -name "*kernel_stats*" | head; for f in $(find gpurun_out/c2_prof -name "*kernel_stats.csv"); do head -25 $f; done
