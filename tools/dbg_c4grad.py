#!/usr/bin/env python3
"""Debug aid (GPU): BASELINE configs[3]'s whole training step, every parameter gradient against float64 autograd on the
oracle, with the sampler's density pass in both inference arithmetics (act_scaling True / False) and the float32 oracle
end to end (own coarse pass and resampling) as yardstick."""
import os, sys, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
from oracle import fsnerf_oracle as O
from test_parity_fp64 import cfg_of, make_sd, ndc_rays
from fs_nerf_amd.core.models import NeRF
from fs_nerf_amd.render import rendering as Rm
dev = torch.device("cuda:0")
Lx, Dx, R, S, NI = 8, 256, 256, 64, 128
sd = make_sd(Lx, Dx, 42)
o, d, gen = ndc_rays(R, 7)
u, uf = torch.rand(R, generator=gen), torch.rand(R, NI, generator=gen)
gt = torch.rand(R, 3, generator=gen)
kw = dict(near=0.0, far=1.0, n_samples=S, n_importance=NI, u=u.double(), u_fine=uf.double(), white_bkgd=True)
sd64 = {k: v.double() for k, v in sd.items()}
with torch.no_grad():
    edges = O.render_rays_oracle(o.double(), d.double(), sd64, None, cfg_of(Lx), **kw)[0][3]["edges"]
sdg = {k: v.double().requires_grad_(True) for k, v in sd.items()}
(wrgb, _, _, _), _, wtv = O.render_rays_oracle(o.double(), d.double(), sdg, None, cfg_of(Lx), edges_override=edges, **kw)
torch.nn.functional.mse_loss(wrgb, gt.double()).backward()
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
kw32 = dict(kw, u=u, u_fine=uf)
with torch.no_grad():
    e32 = O.render_rays_oracle(o, d, sd, None, cfg_of(Lx), **kw32)[0][3]["edges"]
sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
(rgb32, _, _, _), _, _ = O.render_rays_oracle(o, d, sd32, None, cfg_of(Lx), edges_override=e32, **kw32)
torch.nn.functional.mse_loss(rgb32, gt).backward()
res = {}
for scaling in (True, False):
    m = NeRF(3, 3, Lx, Dx, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    m.act_scaling = scaling
    m = m.to(dev).train()
    est = Rm.StratifiedEstimator(0.0, 1.0, S, NI).train()
    (rgb, opacity, depth, ex), ri, tv = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev, u=u.to(dev), u_fine=uf.to(dev))
    torch.nn.functional.mse_loss(rgb, gt.to(dev)).backward()
    print(f"act_scaling={scaling}: t_vals max err vs f64 {float((tv.cpu().double() - wtv).abs().max()):.2e} "
          f"(f32 oracle e2e: {float(((e32[:, :-1] + e32[:, 1:]) / 2).reshape(-1).double().sub(wtv).abs().max()):.2e})  exps {m._act_exps}")
    res[scaling] = {n: p.grad.clone() for n, p in m.named_parameters()}
for k in sd:
    print(f"{k:20s} hip scaled {rel(res[True][k], sdg[k].grad):.2e}  hip r3 {rel(res[False][k], sdg[k].grad):.2e}  f32 oracle e2e {rel(sd32[k].grad, sdg[k].grad):.2e}")
