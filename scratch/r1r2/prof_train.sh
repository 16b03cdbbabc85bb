#!/bin/bash
# kernel-trace + stats of three generalizable-training steps (config-3 shapes) -> gpurun_out/prof_train_r2/
export TMPDIR=/tmp
rm -rf gpurun_out/prof_train_r2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_r2 -o t -- python scratch/train_prof.py > gpurun_out/prof_train_r2.log 2>&1
grep "train step" gpurun_out/prof_train_r2.log
