"""A/B tests of the measured-and-dropped kernel variants: they run against the DEV build of the library
(`make -C mvsnerf_amd/csrc dev` -> scratch/lib/libmvsnerf_hip_dev.so, which exports mvsnerf_tune), never against the product library.

    make -C mvsnerf_amd/csrc dev && python -m pytest scratch/dev_tests -m gpu -q

The files here are the round-2 versions of the tests that flipped a switch (mvsnerf_tune "mlp_variant" / "mlp_gather" / "conv_mfma" /
"conv_xcd" / "psw_bwd_tiles"); tests/ holds their product counterparts (one kernel each, against the oracle or a float64 reference)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X")
    from mvsnerf_amd import _lib
    dev = os.path.join(ROOT, "scratch", "lib", "libmvsnerf_hip_dev.so")
    if not os.path.exists(dev):
        raise RuntimeError(f"{dev} is missing: make -C mvsnerf_amd/csrc dev")
    _lib.LIB_PATH = dev
    _lib._lib = None
    _lib.SIGNATURES["mvsnerf_tune"] = (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int])
    _lib.SIGNATURES["mvsnerf_debug_mlp_occupancy"] = (ctypes.c_int, [ctypes.c_int])
