#!/bin/bash
# training-step time without the profiler, host syncs, then the kernel trace -> gpurun_out/train_time.log, gpurun_out/prof_train_r2/
mkdir -p gpurun_out
python scratch/train_prof.py > gpurun_out/train_time.log 2>&1
if [ "$1" == "sync" ]; then python scratch/sync_debug.py >> gpurun_out/train_time.log 2>&1; fi
if [ "$1" != "noprof" ]; then
bash scratch/prof_train.sh >> gpurun_out/train_time.log 2>&1
python scratch/prof_summary.py gpurun_out/prof_train_r2 ${2:-45} >> gpurun_out/train_time.log 2>&1
fi
