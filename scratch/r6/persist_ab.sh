#!/bin/bash
# same-box A/B of the persistent fp16x3 MLP kernel (product) against the one-tile-per-workgroup build (scratch/lib/libmvsnerf_hip_prep.so) at two launch sizes
out=gpurun_out/$1; mkdir -p $out
for n in 1024 4096 16384; do
  for rep in 1 2; do
    H3_N=$n python scratch/r6/h3_ab.py fp16x3
    H3_N=$n MVS_LIB=scratch/lib/libmvsnerf_hip_prep.so python scratch/r6/h3_ab.py fp16x3
  done
done 2>&1 | grep -v amdgpu.ids > $out/ab.txt
cat $out/ab.txt
timeout 1200 python -m pytest tests/test_gpu_guard.py tests/test_gpu_fp16x3.py tests/test_gpu_raymarch.py tests/test_gpu_costream.py -x -q 2>&1 | tail -5 > $out/tests.txt
cat $out/tests.txt
