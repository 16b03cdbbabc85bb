#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scratch/r2/lds_atomic_rate.hip -o /tmp/lds_atomic_rate > gpurun_out/lds_atomic_rate.log 2>&1
timeout 120 /tmp/lds_atomic_rate >> gpurun_out/lds_atomic_rate.log 2>&1
