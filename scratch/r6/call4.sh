#!/bin/bash
# bash scratch/r6/call4.sh <outdir> <variants...>: A/B (product and r5 around the variants) + census of "cen"
mkdir -p gpurun_out/$1; o=gpurun_out/$1; shift
{
python scratch/r6/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_r5.so python scratch/r6/h3_ab.py fp16x3
for v in "$@"; do MVS_LIB=scratch/lib/libmvsnerf_hip_$v.so python scratch/r6/h3_ab.py fp16x3; done
python scratch/r6/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_r5.so python scratch/r6/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_cen.so python scratch/r6/h3_census.py
} 2>&1 | grep -v amdgpu.ids > $o/ab.txt
cat $o/ab.txt
