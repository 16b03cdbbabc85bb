#!/bin/bash
mkdir -p gpurun_out/$1
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/$1/tests_full.txt 2>&1
tail -15 gpurun_out/$1/tests_full.txt
