#!/bin/bash
# bash scratch/r6/census.sh <outdir> <census variant names...>
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do echo "== $v"; MVS_LIB=scratch/lib/libmvsnerf_hip_$v.so python scratch/r6/h3_census.py; done 2>&1 | grep -v amdgpu.ids > $out/census.txt
cat $out/census.txt
