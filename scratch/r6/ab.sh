#!/bin/bash
# bash scratch/r6/ab.sh <outdir> <variant names...>   ("product" = the product library); same-box A/B of the fp16x3 MLP kernel
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  if [ "$v" = product ]; then python scratch/r6/h3_ab.py fp16x3; else MVS_LIB=scratch/lib/libmvsnerf_hip_$v.so python scratch/r6/h3_ab.py fp16x3; fi
done 2>&1 | grep -v amdgpu.ids > $out/ab.txt
cat $out/ab.txt
