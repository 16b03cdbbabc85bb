mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nccl.py -q -x 2>&1 | tail -25 > gpurun_out/nccl1.log
timeout 900 python bench.py --multi-gpu-legs > gpurun_out/bench_r2_1.json 2> gpurun_out/bench_r2_1.err
tail -5 gpurun_out/bench_r2_1.err
