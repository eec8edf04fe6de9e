#!/usr/bin/env python3
"""One-GPU timings of the five BASELINE.json configurations -> a markdown table.
C3 is what bench.py reports; C4 is `bench.py --workload train`; the others are parity-test shapes timed here for the
record.  Frames are rendered with get_rays + one fused launch; `value` = rays / wall time of the timed frames."""
import json
import math
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fs_nerf_amd import ops  # noqa: E402
from fs_nerf_amd.core.models import NeRF  # noqa: E402
from bench import orbit_pose  # noqa: E402

dev = torch.device("cuda:0")
ANG = 0.6911112


def net(seed, L, D, skip, prec):
    torch.manual_seed(seed)
    m = NeRF(3, 3, L, D, skip, precision=prec, pos_fn={"n_freqs": 10, "log_space": True},
             dir_fn={"n_freqs": 4, "log_space": True})
    with torch.no_grad():
        m.sigma.weight.mul_(64.0)
        m.sigma.bias.add_(3.0)
    return m.to(dev).eval()


def freq_mask(d, n, ratio):  # FreeNeRF-style linear schedule (oracle.freq_mask), identity band always on
    k = torch.arange(n, dtype=torch.float32)
    w = torch.clamp(n * ratio - k, 0.0, 1.0)
    return torch.cat([torch.ones(d), w.repeat_interleave(2 * d)])


def run(name, hw, S, NI, L, D, skip, prec, mask, frames=3):
    focal = 0.5 * hw / math.tan(0.5 * ANG)
    fine = net(43, L, D, skip, prec)
    coarse = net(42, L, D, skip, prec) if NI else None
    pm = freq_mask(3, 10, 0.5).to(dev) if mask else None
    dm = freq_mask(3, 4, 0.5).to(dev) if mask else None
    pc, pf = (coarse.packed() if coarse else None), fine.packed()

    def frame(i):
        o, d = ops.get_rays(orbit_pose(4.0 * i), hw, hw, focal, dev)
        return ops.render_fused(pc, pf, o, d, near=2.0, far=6.0, n_samples=S, n_importance=NI, bkgd=(1.0, 1.0, 1.0),
                                pos_mask=pm, dir_mask=dm, want_extras=False)

    frame(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        out = frame(1 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    assert bool(torch.isfinite(out[0]).all())
    fl_full = {(8, 256): 1186816, (4, 128): 167680}[(L, D)]
    fl_dens = {(8, 256): 982528, (4, 128): 114688}[(L, D)]
    flop = (S * fl_dens if NI else 0) + (S + NI) * fl_full
    rays = hw * hw
    return f"| {name} | {hw}x{hw}, {S}+{NI}, {L}x{D}, {prec}{', mask' if mask else ''} | {rays / dt / 1e6:.2f} M rays/s | " \
           f"{dt * 1e3:.1f} ms/frame | {flop * rays / dt / 1e12:.0f} TFLOP/s |"


rows = ["| config | shape | rate | time | algorithmic rate |", "|---|---|---|---|---|"]
rows.append(run("C1 (reference's CPU case)", 100, 64, 0, 4, 128, (), "fp16x3", False, frames=20))
rows.append(run("C2", 400, 64, 0, 8, 256, (4,), "fp16x3", True, frames=5))
rows.append(run("C3 (headline)", 800, 64, 128, 8, 256, (4,), "fp16x3", False))
rows.append(run("C5 (parity mode)", 1600, 128, 256, 8, 256, (4,), "fp16x3", False, frames=1))
rows.append(run("C5 (bf16 as specified)", 1600, 128, 256, 8, 256, (4,), "bf16", False, frames=1))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "train"], capture_output=True, text=True)
j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
rows.append(f"| C4 (training) | 4096 NDC rays x (64+128), 8x256, fp16x3, fwd+bwd+Adam | {j['value'] / 1e6:.3f} M rays/s | "
            f"{j['ms_per_step']:.1f} ms/step | {j['roofline']['achieved']:.0f} TFLOP/s |")
print("\n".join(rows))
