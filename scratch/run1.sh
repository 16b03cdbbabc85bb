mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_headline_parity.py -q -s 2>&1 | tail -40 > gpurun_out/headline1.log
