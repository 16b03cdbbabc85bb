#!/bin/bash
# round 4, call 22: LDS-tiled fp32-MFMA kernels for conv1 / conv2 / conv11 (encode and fp32 training step)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_encoder.py -q --tb=short -p no:cacheprovider -x > gpurun_out/c22_tests.log 2>&1; echo "tests rc $?" | tee -a gpurun_out/c22_tests.log
tail -8 gpurun_out/c22_tests.log
timeout 300 python scratch/r3/enc_only.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c22_enc.txt
timeout 300 python scratch/r3/train_prof.py fp32 10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c22_ab.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c22_prof" -o enc -- python "$GRAFT_REPO_ROOT/scratch/r3/enc_only.py" > "$GRAFT_REPO_ROOT/gpurun_out/c22_prof.log" 2>&1; echo "prof rc $?")
grep "tiled\|mfma16\|c16to8" gpurun_out/c22_prof/enc_kernel_stats.csv | cut -c1-200
