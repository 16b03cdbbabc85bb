#!/bin/bash
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b/h3_ab.txt
: > $O
python scratch/r3/h3_ab.py fp16x3 bf16x6 >> $O 2>&1
for v in h3_epi h3_i4 h3_e4; do MVS_LIB=scratch/lib/libmvsnerf_hip_$v.so python scratch/r3/h3_ab.py fp16x3 >> $O 2>&1; done
python scratch/r3/h3_ab.py fp16x3 >> $O 2>&1
grep -v amdgpu.ids $O
