#!/bin/bash
# encoder tests + stage times + a kernel trace of the encode -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_headline_parity.py tests/test_gpu_views.py tests/test_gpu_backward.py -q -x 2>&1 | tail -5 > gpurun_out/enc1.log
rm -rf gpurun_out/prof_enc_r2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_enc_r2 -o enc -- python scratch/enc_layers.py > gpurun_out/enc_time.log 2>&1
