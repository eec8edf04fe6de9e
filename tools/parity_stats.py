#!/usr/bin/env python3
"""Error budget of the fused HIP path against a float64 evaluation of the oracle (the truth), beside the float32
oracle ("the reference PyTorch CPU path") on the same inputs: max / percentile absolute errors of rgb_map, depth_map,
weights for every precision mode, on the BASELINE configurations' network shapes.
usage: python tools/parity_stats.py [--rays 256] [--json out.json]      (needs the GPU)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fsnerf_oracle as O  # noqa: E402


def nets(tag, seed):
    L, D = {"4x128": (4, 128), "8x256": (8, 256)}[tag]
    sd = O.init_nerf_state_dict(L, D, [4], 10, 4, seed=seed)
    sd["sigma.weight"] = sd["sigma.weight"] * 64.0
    sd["sigma.bias"] = sd["sigma.bias"] + 3.0
    return sd, dict(n_layers=L, skip=[4], n_freqs=10, n_freqs_dir=4, log_space=True), (L, D)


def rays(R, seed, hw, focal, ndc=False):
    gen = torch.Generator().manual_seed(seed)
    if ndc:
        pose = torch.eye(4)
        pose[:3, 3] = torch.tensor([0.1, -0.05, 0.0])
        o, d = O.get_rays(pose, (378, 504, 407.6))
        o, d = O.to_ndc(o.reshape(-1, 3), d.reshape(-1, 3), (378, 504, 407.6), 1.0)
    else:
        pose = O.pose_from_spherical(4.0311289, 50.0, float(torch.rand(1, generator=gen)) * 360.0)
        o, d = O.get_rays(pose, (hw, hw, focal))
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    idx = torch.randperm(o.shape[0], generator=gen)[:R]
    return o[idx].contiguous(), d[idx].contiguous(), gen


CONFIGS = {  # name: (net, S, NI, two nets, hw, focal, near, far, ndc, mask ratio)
    "C1": ("4x128", 64, 0, False, 100, 138.88887889922103, 2.0, 6.0, False, None),
    "C2": ("8x256", 64, 0, False, 400, 555.5555, 2.0, 6.0, False, 0.5),
    "C3": ("8x256", 64, 128, True, 800, 1111.111, 2.0, 6.0, False, None),
    "C4": ("8x256", 64, 128, False, 0, 0.0, 0.0, 1.0, True, None),
}


def oracle_all(cfgname, R, dtype):
    tag, S, NI, two, hw, focal, near, far, ndc, mr = CONFIGS[cfgname]
    sd_c, cfg, _ = nets(tag, 42)
    sd_f = nets(tag, 43)[0] if two else None
    o, d, gen = rays(R, 7, hw, focal, ndc)
    u = torch.rand(R, generator=gen)
    uf = torch.rand(R, NI, generator=gen) if NI else None
    pm = O.freq_mask(3, 10, mr) if mr else None
    dm = O.freq_mask(3, 4, mr) if mr else None
    cast = (lambda t: None if t is None else t.to(dtype))
    sdc = {k: v.to(dtype) for k, v in sd_c.items()}
    sdf = None if sd_f is None else {k: v.to(dtype) for k, v in sd_f.items()}
    out = O.render_rays_oracle(cast(o), cast(d), sdc, sdf, cfg, near=near, far=far, n_samples=S, n_importance=NI,
                               u=cast(u), u_fine=cast(uf), white_bkgd=True, pos_mask=cast(pm), dir_mask=cast(dm))
    return out, (o, d, u, uf, pm, dm, sd_c, sd_f)


def hip_all(cfgname, R, prec, inputs):
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.render import rendering as Rm
    tag, S, NI, two, hw, focal, near, far, ndc, mr = CONFIGS[cfgname]
    o, d, u, uf, pm, dm, sd_c, sd_f = inputs
    L, D = {"4x128": (4, 128), "8x256": (8, 256)}[tag]
    dev = torch.device("cuda:0")

    def mk(sd):
        m = NeRF(3, 3, L, D, (4,), precision=prec, pos_fn={"n_freqs": 10, "log_space": True},
                 dir_fn={"n_freqs": 4, "log_space": True})
        m.load_state_dict(sd)
        m.set_freq_mask(pm, dm)
        return m.to(dev).eval()

    mc = mk(sd_c)
    mf = mk(sd_f) if sd_f is not None else None
    est = Rm.StratifiedEstimator(near, far, S, NI)
    with torch.no_grad():
        out = Rm.render_rays(o, d, est, mc, white_bkgd=True, device=dev, model_fine=mf, u=u.to(dev),
                             u_fine=None if uf is None else uf.to(dev))
    return out


def errs(out, truth):
    (rgb, op, dep, ex), _, _ = out
    (trgb, top, tdep, tex), _, _ = truth
    f = lambda t: t.detach().cpu().double().numpy()
    R = trgb.shape[0]
    return {"rgb_map": np.abs(f(rgb) - f(trgb)).ravel(), "depth_map": np.abs(f(dep) - f(tdep)).ravel(),
            "weights": np.abs(f(ex["weights"]).reshape(R, -1) - f(tex["weights"])).ravel()}


def summary(e):
    return {k: {"max": float(v.max()), "p99": float(np.percentile(v, 99)), "p50": float(np.percentile(v, 50))}
            for k, v in e.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=256)
    ap.add_argument("--json", default=None)
    ap.add_argument("--modes", default="fp16x3,bf16x3,fp16x2,fp16,bf16")
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    res = {}
    for cname in CONFIGS:
        truth, inputs = oracle_all(cname, a.rays, torch.float64)
        o32, _ = oracle_all(cname, a.rays, torch.float32)
        res[cname] = {"fp32 oracle": summary(errs(o32, truth))}
        for prec in a.modes.split(","):
            res[cname][prec] = summary(errs(hip_all(cname, a.rays, prec, inputs), truth))
        print(f"== {cname}: {CONFIGS[cname][0]}, {CONFIGS[cname][1]}+{CONFIGS[cname][2]} samples, {a.rays} rays; "
              f"absolute errors vs float64 (max / p99 / median)")
        for mode, s in res[cname].items():
            print(f"  {mode:12s} " + "  ".join(f"{k}: {v['max']:.2e} / {v['p99']:.2e} / {v['p50']:.2e}" for k, v in s.items()),
                  flush=True)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
