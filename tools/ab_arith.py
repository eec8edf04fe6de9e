#!/usr/bin/env python3
"""A/B of the arithmetic variants of the parity mode in ONE process on ONE device, interleaved rounds: the headline
frame (800x800, 64+128, two 8x256 nets) as
  fp16x3-scaled   FSN_PREC_FP16X3U: unscaled low parts, one accumulator, per-layer activation scales (round 4 default)
  fp16x3-r3       FSN_PREC_FP16X3: low parts scaled by 2^11, own correction accumulator (round 3; act_scaling = False)
  bf16x3          the fall-back mode
and the calibration's exponents / the scaled network's measured layer maxima.
usage: python tools/ab_arith.py [--rounds 5] [--modes fp16x3-scaled,fp16x3-r3,bf16x3]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fs_nerf_amd import _lib as L, ops  # noqa: E402
from fs_nerf_amd.render import rendering as Rm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--modes", default="fp16x3-scaled,fp16x3-r3,bf16x3")
args = ap.parse_args()
dev = torch.device("cuda:0")
near, far = 2.0, 6.0
cam = (bench.orbit_pose(0.0), bench.H, bench.W, bench.FOCAL, 0, bench.H, dev)
probe = lambda: Rm._probe_on_rays(None, None, cam, near, far)


def nets(mode):
    coarse, fine = bench.init_sd(42), bench.init_sd(43)
    for m in (coarse, fine):
        m.to(dev).eval()
        m.precision = "bf16x3" if mode == "bf16x3" else "fp16x3"
        m.act_scaling = mode == "fp16x3-scaled"
    return coarse.packed(probe), fine.packed(probe), coarse, fine


packs = {m: nets(m) for m in args.modes.split(",")}
for name, (pc, pf, c, f) in packs.items():
    if c.infer_prec() == L.FSN_PREC_FP16X3U:
        x, d = probe()
        for tag, net, pm in (("coarse", c, pc), ("fine", f, pf)):
            mx = ops.mlp_layer_maxima(pm, x, d).cpu().tolist()
            print(f"{name} {tag}: exps {net._act_exps}  scaled layer maxima {[round(v, 1) for v in mx]}", flush=True)


def run(pc, pf):
    return ops.render_fused(pc, pf, None, None, near=near, far=far, n_samples=64, n_importance=128, bkgd=(1, 1, 1),
                            want_extras=False, camera=cam, two_phase=True)


outs = {}
for name, (pc, pf, _, _) in packs.items():
    outs[name] = run(pc, pf)
torch.cuda.synchronize()
print("status word after the warm-up launches:", int(ops.status_word(dev).item()), flush=True)
ref = outs.get("bf16x3")
for name, o in outs.items():
    if ref is not None and name != "bf16x3":
        print(f"{name}: max |rgb - bf16x3 rgb| = {(o[0] - ref[0]).abs().max().item():.3e}", flush=True)
times = {n: [] for n in packs}
for r in range(args.rounds):
    for name, (pc, pf, _, _) in packs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(pc, pf)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1))
for name in packs:
    t = sorted(times[name])
    print(f"{name:16s} min {t[0]:8.2f} ms  median {t[len(t) // 2]:8.2f} ms  -> {bench.H * bench.W / t[len(t) // 2] / 1e3:6.3f} Mrays/s "
          f"({bench.FLOP_PER_RAY * bench.H * bench.W / t[len(t) // 2] / 1e9 / 2500 * 100:5.1f} % of 2.5 PF)", flush=True)
