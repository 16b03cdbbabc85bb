import sys, torch
sys.path.insert(0,'.')
from tests.test_gpu_raymarch import _config2_inputs, _args, DEV
from tests.util import load_weights
from mvsnerf_amd import renderer as R, models as M
from oracle import mvsnerf_oracle as O
mlp_sd,_ = load_weights()
net = M.MVSNeRF(D=6, W=128, input_ch_pts=63, input_ch_views=3, input_ch_feat=20, skips=[4], net_type="v0"); net.load_state_dict(mlp_sd); net=net.to(DEV)
for (n_rays,n_samples) in [(3,300),(3,299),(4,300),(40,200)]:
    rig, pose, vol, pts, dirs, ndc, z, ro = _config2_inputs(n_rays, n_samples, D=16, h=24, w=32, H=64, W=96, seed=n_rays * 7 + n_samples)
    ndc = ndc * 1.3 - 0.15
    ref = O.rendering(pose, pts, ndc, z, dirs, vol, rig["images_raw"][:, :3], mlp_sd)
    emb,_ = M.get_embedder(10,0,3)
    qfn = lambda p, vd, f, fn: R.run_network_mvs(p, vd, f, fn, emb, None)
    qfn._mvsnerf_fused = True
    pose_d = {k: v.to(DEV) for k, v in pose.items()}
    with torch.no_grad():
        out = R.rendering(_args(), pose_d, pts.to(DEV), ndc.to(DEV), z.to(DEV), ro.to(DEV), dirs.to(DEV), vol.to(DEV), rig["images_raw"][:, :3].to(DEV), network_fn=net, network_query_fn=qfn)
        raw = R.rendering.last_raw
    names = ["rgb","feat","w","depth","alpha"]
    print(n_rays, n_samples, {k: float((a.cpu()-b).abs().max()) for k,a,b in zip(names, out[:5], ref[:5])}, "raw", float((raw.cpu()-ref[6]).abs().max()))
    e = (raw.cpu()-ref[6]).abs().amax(-1)
    idx = torch.nonzero(e > 1e-3)
    print(" bad points:", idx[:10].tolist(), "of", e.numel())
    fe = (out[1].cpu()-ref[1]).abs()
    print(" feat err per channel:", fe.amax((0,1)).tolist())
