#!/bin/bash
# kernel-trace + stats of the default bench command (the evidence the roofline numbers cite)
export TMPDIR=/tmp
rm -rf gpurun_out/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o b -- python bench.py > gpurun_out/prof_bench.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_bench.log | tail -1 > gpurun_out/bench_under_rocprof.json
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head
