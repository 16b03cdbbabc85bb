#!/bin/bash
# full GPU suite + default bench -> gpurun_out/
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/gpu_all.log
timeout 900 python bench.py > gpurun_out/bench_r2.json 2> gpurun_out/bench_r2.err
tail -3 gpurun_out/bench_r2.err
