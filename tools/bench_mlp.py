#!/usr/bin/env python3
"""Micro-benchmark of fsn_mlp_fwd (the MFMA MLP kernel alone) for kernel tuning.
   python tools/bench_mlp.py [--n 1048576] [--precision fp16x3] [--density] [--reps 5]
Prints samples/s, algorithmic TFLOP/s and the matrix-pipe utilisation this implies
(x3 modes issue 3 MFMA passes per algorithmic product)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--precision", default="fp16x3")
    ap.add_argument("--density", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--small", action="store_true", help="4x128 network")
    a = ap.parse_args()
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, D = (4, 128) if a.small else (8, 256)
    m = NeRF(3, 3, L, D, (4,), precision=a.precision, pos_fn={"n_freqs": 10, "log_space": True},
             dir_fn={"n_freqs": 4, "log_space": True}).to(dev).eval()
    x = torch.rand(a.n, 3, device=dev) * 3 - 1.5
    d = torch.nn.functional.normalize(torch.randn(a.n, 3, device=dev), dim=-1)
    pm = m.packed()
    dd = None if a.density else d
    ops.mlp_fwd(pm, x, dd)
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mlp_fwd(pm, x, dd)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    flop = {(8, False): 1186816, (8, True): 982528, (4, False): 167680, (4, True): 114688}[(L, a.density)]
    tf = a.n * flop / t / 1e12
    passes = 3 if a.precision.endswith("x3") else 1
    print(f"{a.precision} {'density' if a.density else 'full'} {L}x{D} n={a.n}: {t*1e3:.2f} ms  "
          f"{a.n/t/1e6:.1f} Msamples/s  {tf:.1f} TFLOP/s algorithmic  "
          f"matrix-pipe {tf*passes/2500*100:.1f}% of 2.5 PF (min {min(ts)*1e3:.2f} ms)", flush=True)


if __name__ == "__main__":
    main()
