"""CPU: the reference's shipped checkpoint loads into this package's modules UNCHANGED (same state_dict keys and shapes), through
the reference's own entry point create_nerf_mvs(args) with args.ckpt set.  The .tar only exists in the authoring container
(/root/reference); on machines without it the key/shape contract is checked against the committed weights fixture instead."""
import os
import types

import numpy as np
import pytest
import torch

from tests.util import load_weights

REF_CKPT = "/root/reference/ckpts/mvsnerf-v0.tar"


def _args(ckpt):
    return types.SimpleNamespace(feat_dim=20, img_downscale=1.0, use_color_volume=False, net_type="v0", multires=10, i_embed=0, pts_dim=3,
                                 multires_views=4, dir_dim=3, netdepth=6, netwidth=128, N_importance=0, netchunk=1024, ckpt=ckpt,
                                 perturb=1.0, N_samples=128, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.0, pad=24)


def test_state_dict_contract_matches_the_fixture():
    from mvsnerf_amd import models
    mlp_sd, mvs_sd = load_weights()
    net = models.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0")
    mvs = models.MVSNet()
    for mod, sd in ((net, mlp_sd), (mvs, mvs_sd)):
        own = mod.state_dict()
        assert set(own) == set(sd), set(own) ^ set(sd)
        assert all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)
        mod.load_state_dict(sd, strict=True)
    # the fine-tuning checkpoint adds `volume.feat_volume` (train_mvs_nerf_finetuning_pl.py:66,286)
    assert list(models.RefVolume(torch.zeros(1, 8, 8, 8, 8)).state_dict()) == ["feat_volume"]


@pytest.mark.skipif(not os.path.exists(REF_CKPT), reason="reference checkpoint only exists in the authoring container")
def test_reference_checkpoint_loads_through_create_nerf_mvs():
    from mvsnerf_amd import models
    kw_train, kw_test, start, grad_vars = models.create_nerf_mvs(_args(REF_CKPT), use_mvs=True, dir_embedder=False, pts_embedder=True)
    assert {"network_query_fn", "perturb", "N_importance", "network_fine", "N_samples", "network_fn", "network_mvs", "use_viewdirs",
            "white_bkgd", "raw_noise_std"} <= set(kw_train)
    ck = torch.load(REF_CKPT, map_location="cpu", weights_only=False)
    mlp_sd, mvs_sd = load_weights()
    for k, v in kw_train["network_fn"].state_dict().items():
        assert torch.equal(v.cpu(), ck["network_fn_state_dict"][k]) and np.array_equal(v.cpu().numpy(), mlp_sd[k].numpy())
    for k, v in kw_train["network_mvs"].state_dict().items():
        assert torch.equal(v.cpu(), ck["network_mvs_state_dict"][k])
    assert len(grad_vars) == len(list(kw_train["network_fn"].parameters())) + len(list(kw_train["network_mvs"].parameters()))
