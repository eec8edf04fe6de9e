#!/usr/bin/env python3
"""A/B timing of k_render_fused across several builds of libfsnerf_hip.so in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24): the headline frame (800x800, 64+128, two 8x256 nets).
usage: python tools/ab_fused.py [--prec fp16x3] [--rounds 3] name=path.so [name=path.so ...]
Timing only: ablation builds may return garbage."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fs_nerf_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="fp16x3")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--two-phase", type=int, default=1)
ap.add_argument("--camera", type=int, default=1)
ap.add_argument("libs", nargs="+")
args = ap.parse_args()
dev = torch.device("cuda:0")
coarse, fine = bench.init_sd(42), bench.init_sd(43)
for m in (coarse, fine):
    m.precision = args.prec
    m.to(dev).eval()
pc, pf = coarse.packed(), fine.packed()  # packed with the default library (the blob format is common)
o, d = ops.get_rays(bench.orbit_pose(0.0), bench.H, bench.W, bench.FOCAL, dev)
torch.cuda.synchronize()
libs = []
for spec in args.libs:
    name, path = spec.split("=", 1)
    l = C.CDLL(os.path.abspath(path))
    for fn, (res, at) in _lib.SIGNATURES.items():
        f = getattr(l, fn)
        f.restype, f.argtypes = res, at
    libs.append((name, l))
default = _lib._lib


def run(l):
    _lib._lib = l
    try:
        cam = (bench.orbit_pose(0.0), bench.H, bench.W, bench.FOCAL, 0, bench.H, dev) if args.camera else None
        return ops.render_fused(pc, pf, None if cam else o, None if cam else d, near=2.0, far=6.0, n_samples=64,
                                n_importance=128, bkgd=(1, 1, 1), want_extras=False, camera=cam,
                                two_phase=bool(args.two_phase))
    finally:
        _lib._lib = default


times = {n: [] for n, _ in libs}
for n, l in libs:
    run(l)
torch.cuda.synchronize()
for r in range(args.rounds):
    for n, l in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(l)
        e1.record()
        torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1))
for n, _ in libs:
    t = sorted(times[n])
    print(f"{n:24s} min {t[0]:8.2f} ms  median {t[len(t) // 2]:8.2f} ms  -> {bench.H * bench.W / t[0] / 1e3:6.3f} Mrays/s "
          f"({bench.FLOP_PER_RAY * bench.H * bench.W / t[0] / 1e9 / 2500 * 100:5.1f} % of 2.5 PF)", flush=True)
