#!/bin/bash
mkdir -p gpurun_out/$1
{
python scratch/r6/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_r5.so python scratch/r6/h3_ab.py fp16x3
python scratch/r6/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_r5.so python scratch/r6/h3_ab.py fp16x3
MVS_LIB=scratch/lib/libmvsnerf_hip_cen.so python scratch/r6/h3_census.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/$1/ab.txt
timeout 1200 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_guard.py tests/test_gpu_raymarch.py -q -m gpu > gpurun_out/$1/tests.txt 2>&1
tail -4 gpurun_out/$1/tests.txt
cat gpurun_out/$1/ab.txt
