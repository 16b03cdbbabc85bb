#!/bin/bash
# the kernel trace of the default bench command and the un-profiled bench line of the final tree (the PMC summary of these kernel sources is already committed)
export TMPDIR=/tmp
O=gpurun_out/r3b_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o b -- python bench.py > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/r03_bench_under_rocprof.json
find $O/bench_trace -name "*kernel_stats.csv" -exec cp {} $O/r03_bench_kernel_stats.csv \;
rm -rf $O/bench_trace
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
ls -la $O
