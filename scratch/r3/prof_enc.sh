#!/bin/bash
# rocprofv3 kernel trace of three scene encodes (config 2) -> gpurun_out/r3_enc_kernel_summary.txt
export TMPDIR=/tmp
rm -rf gpurun_out/r3_prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_enc -o e -- python scratch/r3/enc_only.py > gpurun_out/r3_prof_enc.log 2>&1
tail -1 gpurun_out/r3_prof_enc.log
python scratch/r3/prof_summary.py gpurun_out/r3_prof_enc 50 > gpurun_out/r3_enc_kernel_summary.txt
find gpurun_out/r3_prof_enc -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_enc_kernel_stats.csv \;
find gpurun_out/r3_prof_enc -name "*kernel_trace.csv" -delete
cut -c1-160 gpurun_out/r3_enc_kernel_summary.txt | head -45
