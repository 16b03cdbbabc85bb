#!/bin/bash
# the round-end GPU suite, as the driver runs it; log -> gpurun_out/gpu_all.log
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/gpu_all.log
