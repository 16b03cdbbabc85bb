"""The no-grad product path (scene encode + a rendered frame, library defaults = guarded fp16x3 kernels; then the same with every kernel fp32) next to the aggressor of
tests/test_gpu_costream.py, bit for bit against the quiet run.  With MVSNERF_TEST_MFMA_HOG=scratch/r5/libpk_hog.so the aggressor is the distilled trigger (waves spinning on
v_mfma_f32_16x16x32_f16)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsnerf_amd import train, ops
from mvsnerf_amd import encoder as E
from tests.test_gpu_costream import _aggressor, _with_aggressor, DEV
from tests.test_gpu_train import _system
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
aggress = _aggressor()
sys_, args, _, _ = _system(8, 512, 64, 32)
batch = train.synthetic_batch(128, 160, seed=3, rot_deg=2.0, smooth=True)
mvs = sys_.MVSNet
data, _ = sys_.decode_batch(dict(batch))
imgs, proj, nf = data["images"][:, :3], data["proj_mats"][:, :3], data["near_fars"][0, 0]
for mode in ("auto", "fp32"):
    def frame():
        with torch.no_grad(), ops.mlp_precision(mode), E.encoder_precision(mode):
            vol, _, _ = mvs(imgs, proj, nf, pad=args.pad)
            rgb, depth = sys_.render_view(batch, batch_rays=1024)
        return [vol.clone(), rgb.clone(), depth.clone()]
    quiet = _with_aggressor(frame, None)
    assert all(torch.equal(a, b) for a, b in zip(quiet, _with_aggressor(frame, None))), "quiet runs differ"
    bad = 0
    for it in range(N):
        got = _with_aggressor(frame, aggress, n=8 + (it % 5) * 4)
        bad += any(not torch.equal(a, b) for a, b in zip(quiet, got))
    print(f"mode {mode}: encode + 128x160 frame: {bad} of {N} aggressed runs differ from the quiet run in any bit; guard fallbacks so far {ops.guard_fallbacks()}")
