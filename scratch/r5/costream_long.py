"""tests/test_gpu_costream.py::test_training_kernels_bit_for_bit_... with 60 aggressed runs per stage instead of 6 (evidence run; profiles/r05_costream_ab.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mvsnerf_amd import train
from mvsnerf_amd import encoder as E
from tests.test_gpu_costream import _aggressor, _with_aggressor, DEV
from tests.test_gpu_train import _system
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
aggress = _aggressor()
E.PSW_BWD_DETERMINISTIC = True
for amp in (True, False):
    sys_, args, _, _ = _system(8, 512, 64, 32)
    args.use_amp = amp
    batch = train.synthetic_batch(128, 160, seed=3, rot_deg=2.0, smooth=True)
    net, mvs = sys_.render_kwargs_train["network_fn"], sys_.MVSNet
    mlp_params, enc_params = list(net.named_parameters()), list(mvs.named_parameters())

    def raymarch_step():
        for _, p in mlp_params:
            p.grad = None
        torch.manual_seed(11)
        out = sys_.training_step(batch, 0)
        out["loss"].backward()
        return [out["loss"].detach().clone()] + [p.grad.detach().clone() for _, p in mlp_params]

    data, _ = sys_.decode_batch(dict(batch))
    imgs, proj, nf = data["images"][:, :3], data["proj_mats"][:, :3], data["near_fars"][0, 0]
    G = [None]

    def encoder_node():
        for _, p in enc_params:
            p.grad = None
        with E.encoder_precision("bf16" if amp else "auto"):
            vol, _, _ = mvs(imgs, proj, nf, pad=args.pad)
            if G[0] is None:
                G[0] = torch.randn(vol.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(7))
            (vol * G[0]).sum().backward()
        return [vol.detach().clone()] + [p.grad.detach().clone() for _, p in enc_params]

    for name, fn in (("ray march (loss + 22 MLP gradients)", raymarch_step), ("encoder node (volume + 56 gradients)", encoder_node)):
        quiet = _with_aggressor(fn, None)
        bad_runs, bad_tensors = 0, 0
        for it in range(N):
            got = _with_aggressor(fn, aggress, n=12 + (it % 5) * 4)        # 12 .. 28 aggressor launches: the overlap shifts from run to run
            nb = sum(0 if torch.equal(a, b) else 1 for a, b in zip(quiet, got))
            bad_runs += nb > 0
            bad_tensors += nb
        print(f"use_amp={amp}: {name}: {bad_runs} of {N} aggressed runs differ from the quiet run in any bit ({bad_tensors} tensors)")
