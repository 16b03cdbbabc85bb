#!/bin/bash
o=gpurun_out/r5h; mkdir -p $o
python -m pytest tests/test_gpu_encoder.py tests/test_gpu_fp16x3_encoder.py tests/test_gpu_headline_parity.py tests/test_gpu_guard.py tests/test_gpu_costream.py -m gpu -q 2>&1 | grep -v "^$" > $o/pytest.log
grep -n "^E  .*Error\|passed\|failed\|^FAILED" $o/pytest.log | cut -c1-400 | tail -20
scratch/r5/prof_enc.sh r5h | head -6
