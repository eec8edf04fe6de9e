#!/usr/bin/env python3
"""Experiment (GPU): per-tensor error of the parameter gradients against float64 autograd on the oracle, for the library
FSN_LIB_PATH points at (variants of k_wgrad that drop low-part products, profiles/EXPERIMENTS_r4.md section 11).
Cases: the 8x256 gradient test's shape (N = 777 samples, random d(out)), the same with N = 20,000, and BASELINE
configs[3]'s whole step through the renderer at 256 rays (an MSE loss: coherent d(out))."""
import json, os, sys, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
from oracle import fsnerf_oracle as O
from test_parity_fp64 import cfg_of, make_sd, ndc_rays
from test_train_step import _rel, _relu_margin
from fs_nerf_amd.core.models import NeRF
from fs_nerf_amd.render import rendering as Rm

dev = torch.device("cuda:0")
L, D = 8, 256
out = {}


def model(sd):
    m = NeRF(3, 3, L, D, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    return m.to(dev).train()


def mlp_case(N, seed):
    sd = O.init_nerf_state_dict(L, D, [4], 10, 4, seed=3)
    sd["sigma.weight"] *= 16.0
    m = model(sd)
    gen = torch.Generator().manual_seed(seed)
    x = torch.rand(12 * N, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(12 * N, 3, generator=gen), dim=-1)
    keep = _relu_margin(sd, x, d, L, [4], 10, 4) > 2e-5
    x, d = x[keep][:N].contiguous(), d[keep][:N].contiguous()
    c = torch.randn(N, 4, generator=gen)
    (m(x.to(dev), d.to(dev)) * c.to(dev)).sum().backward()
    sdr = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    (O.nerf_forward(sdr, x.double(), d.double(), **cfg_of(L)) * c.double()).sum().backward()
    sd32 = {k: v.detach().float().clone().requires_grad_(True) for k, v in sd.items()}
    (O.nerf_forward(sd32, x, d, **cfg_of(L)) * c).sum().backward()
    return {n: (_rel(p.grad, sdr[n].grad), _rel(sd32[n].grad, sdr[n].grad)) for n, p in m.named_parameters()}


def step_case(R):
    S, NI = 64, 128
    sd = make_sd(L, D, 42)
    o, d, gen = ndc_rays(R, 7)
    u, uf = torch.rand(R, generator=gen), torch.rand(R, NI, generator=gen)
    gt = torch.rand(R, 3, generator=gen)
    kw = dict(near=0.0, far=1.0, n_samples=S, n_importance=NI, u=u.double(), u_fine=uf.double(), white_bkgd=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        edges = O.render_rays_oracle(o.double(), d.double(), sd64, None, cfg_of(L), **kw)[0][3]["edges"]
    sdg = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (wrgb, _, _, _), _, _ = O.render_rays_oracle(o.double(), d.double(), sdg, None, cfg_of(L), edges_override=edges, **kw)
    torch.nn.functional.mse_loss(wrgb, gt.double()).backward()
    m = model(sd)
    est = Rm.StratifiedEstimator(0.0, 1.0, S, NI).train()
    (rgb, _, _, _), _, _ = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev, u=u.to(dev), u_fine=uf.to(dev))
    torch.nn.functional.mse_loss(rgb, gt.to(dev)).backward()
    return {n: (_rel(p.grad, sdg[n].grad), 0.0) for n, p in m.named_parameters()}


for name, fn in (("mlp777", lambda: mlp_case(777, 0)), ("mlp20000", lambda: mlp_case(20000, 1)), ("step256", lambda: step_case(256))):
    r = fn()
    out[name] = {k: [float(f"{a:.3e}"), float(f"{b:.3e}")] for k, (a, b) in r.items()}
    worst = max(r, key=lambda k: r[k][0])
    big = [k for k in r if k.endswith("weight") and k.split(".")[0] in ("layers", "connection") and k != "layers.0.weight"]
    print(f"{os.path.basename(os.environ.get('FSN_LIB_PATH', 'in-tree')):14s} {name:9s} worst {r[worst][0]:.2e} at {worst:18s} "
          f"(f32 autograd there {r[worst][1]:.1e}); big-job tensors max {max(r[k][0] for k in big):.2e}; "
          f"layers.0.weight {r['layers.0.weight'][0]:.2e}", flush=True)
dst = os.path.join(R_, "gpurun_out", "wgdrop_" + os.path.basename(os.environ.get("FSN_LIB_PATH", "base")).replace(".so", "") + ".json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
