#!/usr/bin/env python3
"""Fused-kernel timing for a few sampler configurations (tuning aid): ns per MLP sample evaluation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fs_nerf_amd import ops
from fs_nerf_amd.core.models import NeRF
dev = torch.device("cuda:0")
torch.manual_seed(0)
kw = dict(pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
mc, mf = NeRF(3, 3, 8, 256, (4,), **kw).to(dev).eval(), NeRF(3, 3, 8, 256, (4,), **kw).to(dev).eval()
with torch.no_grad():
    for m in (mc, mf):
        m.sigma.weight.mul_(64); m.sigma.bias.add_(3)
pc, pf = mc.packed(), mf.packed()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
o = torch.tensor([[0.0, 0.0, 4.0]], device=dev).expand(R, 3).contiguous()
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev) * 0.2 + torch.tensor([0, 0, -1.0], device=dev), dim=-1)
for S, NI in ((128, 0), (64, 0), (64, 128), (64, 64), (128, 256)):
    def run():
        return ops.render_fused(pc if NI else None, pf, o, d, near=2.0, far=6.0, n_samples=S, n_importance=NI,
                                bkgd=(1, 1, 1), want_extras=False)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3
    evals = R * ((S if NI else 0) + S + NI)
    print(f"S={S:3d} n_imp={NI:3d}: {t*1e3:8.2f} ms  {R/t/1e6:6.3f} Mrays/s  {t/evals*1e9:6.3f} ns per MLP evaluation", flush=True)
