"""tests/test_gpu_headline_parity.py (config 2 end to end: all HIP against all oracle, unchanged bounds) and tests/test_gpu_encoder.py with the scene
encoder's conv0 on the opt-in fp32-grade fp16 kernel (encoder_precision "fp16x3") instead of the fp32-MFMA one: what the parity file of a default
switch would look like.  python scratch/r3/parity_with_fp16_conv0.py"""
import os
import sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from mvsnerf_amd import encoder
encoder.ENCODER_PRECISION = "fp16x3"
sys.exit(pytest.main(["tests/test_gpu_headline_parity.py", "tests/test_gpu_encoder.py", "-q", "-s", "-m", "gpu"]))
