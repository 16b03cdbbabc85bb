#!/bin/bash
# rocprofv3 kernel trace of the generalizable-training step (config-3 shapes), fp32 and use_amp -> gpurun_out/r3_prof_train_{fp32,amp}/ + per-kernel summaries
export TMPDIR=/tmp
for mode in fp32 amp; do
  rm -rf gpurun_out/r3_prof_train_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_train_$mode -o t -- python scratch/r3/train_prof.py $mode 5 > gpurun_out/r3_prof_train_$mode.log 2>&1
  grep "train step" gpurun_out/r3_prof_train_$mode.log
  python scratch/r3/prof_summary.py gpurun_out/r3_prof_train_$mode 60 > gpurun_out/r3_train_kernel_summary_$mode.txt
  find gpurun_out/r3_prof_train_$mode -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_train_kernel_stats_$mode.csv \;
  find gpurun_out/r3_prof_train_$mode -name "*kernel_trace.csv" -delete
done
head -45 gpurun_out/r3_train_kernel_summary_amp.txt
