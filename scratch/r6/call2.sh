#!/bin/bash
mkdir -p gpurun_out/r6f
timeout 1200 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_guard.py tests/test_gpu_raymarch.py -q -m gpu > gpurun_out/r6f/tests.txt 2>&1
tail -40 gpurun_out/r6f/tests.txt
python scratch/r3/h3_ab.py fp16x3 2>&1 | grep -v amdgpu.ids
