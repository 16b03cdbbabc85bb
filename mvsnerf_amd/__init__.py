"""mvsnerf_amd - MI355X-native (gfx950) implementation of the MVSNeRF rendering hot path.

Module names mirror the reference so that `from mvsnerf_amd.models import *; from mvsnerf_amd.renderer
import *; from mvsnerf_amd.utils import *` replaces the reference's star-imports
(train_mvs_nerf_pl.py:8-10).  All compute lives in mvsnerf_amd/lib/libmvsnerf_hip.so (C ABI:
include/mvsnerf_hip.h); there is no PyTorch/CPU fallback.
"""
__version__ = "0.1.0"
