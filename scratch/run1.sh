mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_headline_parity.py tests/test_gpu_views.py -q -x 2>&1 | tail -5 > gpurun_out/enc1.log
timeout 600 python scratch/enc_layers.py > gpurun_out/enc_time.log 2>&1
