"""CPU-only: error behaviour of the boundary.  Argument validation happens before any launch, so the negative MVSNERF_E*
codes can be exercised without a GPU; the Python layer must raise (never fall back) on a missing library, on host
tensors and on unsupported shapes."""
import ctypes

import pytest
import torch

from mvsnerf_amd import _lib, ops

EINVAL, EUNSUPPORTED, EALIGN = -1, -2, -3


def test_c_abi_rejects_bad_arguments_without_launching():
    l = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    assert l.mvsnerf_volume_sample_fwd(0, 4, 4, 4, 8, p, 1, p, 8, 0, 0) == EINVAL                # null volume
    assert l.mvsnerf_volume_sample_fwd(p, 4, 4, 4, 8, p, -1, p, 8, 0, 0) == EINVAL               # negative count
    assert l.mvsnerf_volume_sample_fwd(p, 4, 4, 4, 8, p, 0, p, 8, 1, 0) == 0 and l.mvsnerf_volume_sample_fwd(p, 4, 4, 4, 8, p, 1, p, 8, 2, 0) == EINVAL                     # empty input is a no-op
    assert l.mvsnerf_volume_sample_fwd(p + 4, 4, 4, 4, 8, p, 1, p, 8, 0, 0) == EALIGN            # misaligned volume (16-byte channel vectors)
    assert l.mvsnerf_composite_fwd(0, p, 1, 4, 0, p, p, p, p, p, p, 0) == EINVAL
    assert l.mvsnerf_composite_fwd(p, p, 0, 4, 0, p, p, p, p, p, p, 0) == 0
    assert l.mvsnerf_mlp_fwd(p, 21, p, 3, p, 21, p, 3, 1, 1, 0, p, 0) == EUNSUPPORTED         # odd feat_dim
    assert l.mvsnerf_conv3d_fwd(p, 0, 0, 0, 0, 0, 12, 12, 8, 8, 8, p, 8, 1, p, 0) == EUNSUPPORTED     # no kernel for 12->8
    assert l.mvsnerf_conv3d_fwd(p, 0, 0, 0, 0, 0, 8, 8, 8, 8, 8, p, 16, 3, p, 0) == EUNSUPPORTED      # stride 3
    assert l.mvsnerf_conv2d_fwd(p, p, 0, 8, 8, 1, 8, 8, p, 0, 8, 3, 1, p, 0) == EINVAL                 # scale without shift
    assert l.mvsnerf_planesweep_costvar_fwd(p, 0, p, p, 3, 16, 8, 8, 4, 0, p, 16, p, 0, 0) == EUNSUPPORTED   # C != 32
    assert l.mvsnerf_raymarch_fwd(None, 0) == EINVAL and l.mvsnerf_render_pixels_fwd(None, 0) == EINVAL
    assert l.mvsnerf_render_workspace_floats(0, 128, 3) == 0
    # split-MLP weight buffers: n_split 1..3 = bf16 pieces, MVSNERF_SPLIT_FP16 (18) = two fp16 pieces per operand (hi plane + lo plane per layer)
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mvsnerf_hip.h")).read()
    assert int(re.search(r"#define MVSNERF_SPLIT_FP16 (\d+)", hdr).group(1)) == ops.N_SPLIT["fp16x3"] == 18      # the Python mode table follows the header
    seg = lambda steps, nb: steps * nb * 512
    assert l.mvsnerf_mlp_packed_split_elems(20, 18) == 2 * (seg(2, 4) + seg(4, 4) + 4 * seg(8, 4) + seg(4, 4) + 2 * seg(8, 4) + seg(9, 2)) + 32 == 256032     # + 32 status elements (64 bytes: "a weight was not finite", the K-blocks' largest |w| and scale exponents; ABI 12)
    assert l.mvsnerf_mlp_packed_split_elems(20, 4) == 0 and l.mvsnerf_mlp_packed_split_elems(21, 18) == 0 and l.mvsnerf_mlp_packed_split_elems(20, 3) > 0
    wp = (ctypes.c_void_p * 11)(*[p] * 11)
    assert l.mvsnerf_mlp_pack_split(wp, 20, 4, p, 0) == EUNSUPPORTED and l.mvsnerf_mlp_pack_split(wp, 20, 18, p + 4, 0) == EALIGN
    assert l.mvsnerf_mlp_fwd_split(p, p, 20, 7, p, 3, p, 20, p, 3, 1, 1, 0, p, 0) == EUNSUPPORTED
    assert l.mvsnerf_mlp_fwd_split(p, p, 20, 18, p, 3, p, 20, p, 3, 0, 1, 0, p, 0) == 0                # empty batch: no launch
    # guarded sequences (ABI 10): a guard needs the fp16 weight planes; null guard / null struct are rejected before anything is launched
    assert l.mvsnerf_mlp_fwd_guarded(p, p, 20, p, 3, p, 20, p, 3, 1, 1, 0, p, 0, 0) == EINVAL             # no guard words
    assert l.mvsnerf_mlp_fwd_guarded(p, p, 20, p, 3, p, 20, p, 3, 0, 1, 0, p, p, 0) == 0                  # empty batch: no launch
    assert l.mvsnerf_sweep_conv0_guarded_fwd(None, 0) == EINVAL
    assert l.mvsnerf_conv0_f16x3_packed_elems(41) == 3 * 7680 + 8
    assert not hasattr(l, "mvsnerf_tune") and not hasattr(l, "mvsnerf_debug_set_census")     # the library has no A/B switches (csrc/knobs.h: constants)


def test_python_layer_raises_and_names_the_op():
    with pytest.raises(RuntimeError, match="volume_sample_fwd failed: invalid argument"):
        _lib.check(EINVAL, "volume_sample_fwd")
    with pytest.raises(RuntimeError, match="hipError 719"):
        _lib.check(719, "mlp_fwd")
    with pytest.raises(RuntimeError, match="expected a contiguous float32 tensor on the GPU"):
        ops.volume_sample(torch.zeros(4, 4, 4, 8), torch.zeros(2, 3))                         # host tensors: no CPU fallback
    with pytest.raises(RuntimeError, match="feat_dim 21 unsupported"):
        ops.mlp_pack([], [], 21)


def test_missing_library_is_loud(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmvsnerf_hip.so")
    with pytest.raises(RuntimeError, match="has no fallback"):
        _lib.lib()
