#!/bin/bash
# round 4, call 4: persistent predicated fp32 kernels of the guarded sequences (MLP, plane sweep), 32x32 transposing epilogue, frame with product defaults
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_guard.py tests/test_gpu_layout.py tests/test_gpu_encoder.py tests/test_gpu_fp16x3_encoder.py tests/test_gpu_fp16x3.py -q --tb=short -p no:cacheprovider > gpurun_out/c4_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c4_tests.log
tail -8 gpurun_out/c4_tests.log
timeout 600 python bench.py > gpurun_out/c4_bench.json 2> gpurun_out/c4_bench.err; echo "bench rc $?"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/c4_prof" -o enc -- python "$GRAFT_REPO_ROOT/scratch/r3/enc_only.py" > "$GRAFT_REPO_ROOT/gpurun_out/c4_enc_prof.log" 2>&1; echo "prof rc $?"
ls "$GRAFT_REPO_ROOT/gpurun_out/c4_prof" | head
