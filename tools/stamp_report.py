#!/usr/bin/env python3
"""Diagnostic (FSN_STAMP build of libfsnerf_hip.so): per-wave cycle totals of the hidden-layer k-loop blocks and the
pair epilogues of one fused frame.  usage: FSN_LIB_PATH=.../libfsnerf_stamp.so python tools/stamp_report.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from fs_nerf_amd import _lib, ops  # noqa: E402
from fs_nerf_amd.render import rendering as Rm  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
dev = torch.device("cuda:0")
coarse, fine = bench.init_sd(42), bench.init_sd(43)
for m in (coarse, fine):
    m.precision = prec
    m.to(dev).eval()
pc, pf = coarse.packed(), fine.packed()
o, d = ops.get_rays(bench.orbit_pose(0.0), bench.H, bench.W, bench.FOCAL, dev)
for _ in range(2):
    ops.render_fused(pc, pf, o, d, near=2.0, far=6.0, n_samples=64, n_importance=128, bkgd=(1, 1, 1), want_extras=False)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (256 * 8 * 8))()
fn = _lib.lib().fsn_dbg_stamps
fn.restype = C.c_int
assert fn(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8, 8).astype(np.float64)
tot, tk, te, n, tko, teo = (a[..., i] for i in range(6))
print(f"precision {prec}: per-wave totals over one frame (s_memtime ticks = shader cycles), median over 256 CUs")
for w in range(8):
    print(f" wave {w}: total {np.median(tot[:, w]):.3e}  k-loop blocks {np.median(tk[:, w] / tot[:, w]) * 100:5.1f} % "
          f"({np.median(tk[:, w] / n[:, w]):7.1f} cyc/block, ideal 768..1536)  epilogues {np.median(te[:, w] / tot[:, w]) * 100:5.1f} % "
          f"({np.median(te[:, w] / n[:, w]):6.1f} cyc each) | other GEMMs: blocks {np.median(tko[:, w] / tot[:, w]) * 100:5.1f} % "
          f"epilogues {np.median(teo[:, w] / tot[:, w]) * 100:4.1f} % | outside pairs "
          f"{np.median(1 - (tk + te + tko + teo)[:, w] / tot[:, w]) * 100:5.1f} %")
