#!/bin/bash
# everything profiles/r02_* is made from: kernel stats of the default bench command, PMC passes, the training step, the encoder
mkdir -p gpurun_out
bash scratch/prof_bench.sh > gpurun_out/prof_bench_sh.log 2>&1
bash scratch/pmc_run.sh > gpurun_out/pmc_run_sh.log 2>&1
bash scratch/prof_train.sh > gpurun_out/prof_train_sh.log 2>&1
tail -3 gpurun_out/prof_bench_sh.log gpurun_out/prof_train_sh.log
tail -25 gpurun_out/pmc_run_sh.log | cut -c1-400
