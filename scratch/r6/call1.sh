#!/bin/bash
# round 6, GPU call 1: the rewritten fp16x3 MLP kernel against round 5's on one box (timing, census), then its parity tests
mkdir -p gpurun_out/r6a
{
python scratch/r3/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_r5.so python scratch/r3/h3_ab.py fp16x3
python scratch/r3/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_r5.so python scratch/r3/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_cen.so python scratch/r3/h3_census.py
} > gpurun_out/r6a/ab.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_guard.py tests/test_gpu_raymarch.py -q -m gpu -x > gpurun_out/r6a/tests.txt 2>&1
tail -5 gpurun_out/r6a/tests.txt
cat gpurun_out/r6a/ab.txt
