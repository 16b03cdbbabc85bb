#!/bin/bash
# builds and runs scratch/r2/wgrad_bench.hip on the GPU box -> gpurun_out/wgrad_bench.log
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-pass-failed -w scratch/r2/wgrad_bench.hip -o /tmp/wgrad_bench > gpurun_out/wgrad_bench.log 2>&1
timeout 300 /tmp/wgrad_bench >> gpurun_out/wgrad_bench.log 2>&1
