"""Training-step timing (BASELINE config 4 shape: n_rays rays x (64+128) samples, 8x256 NeRF,
MSE + Adam).  step = render_rays(train=True) -> mse -> backward -> [all-reduce] -> Adam.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import fs_nerf_amd  # noqa: F401
from fs_nerf_amd.core.models import NeRF
from fs_nerf_amd.render import rendering as R
from fs_nerf_amd import shard


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-opt", action="store_true", help="skip the optimizer update (timing experiments on kernels whose gradients are not meaningful)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True}).to(dev).train()
    with torch.no_grad():
        m.sigma.weight.mul_(64.0)
        m.sigma.bias.add_(3.0)
    est = R.StratifiedEstimator(2.0, 6.0, 64, 128).train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    o = torch.zeros(a.rays, 3, device=dev) + torch.tensor([0.0, 0.0, 4.0], device=dev)
    d = torch.nn.functional.normalize(torch.randn(a.rays, 3, device=dev) * 0.3 + torch.tensor([0.0, 0.0, -1.0], device=dev), dim=-1)
    gt = torch.rand(a.rays, 3, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        (rgb, _, _, _), _, _ = R.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev)
        loss = torch.nn.functional.mse_loss(rgb, gt)
        loss.backward()
        shard.allreduce_grads(m.parameters())
        if not a.no_opt:
            opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "train_rays_per_second", "value": a.rays / dt, "ms_per_step": dt * 1e3, "rays": a.rays,
                      "samples_per_ray": 192, "loss": float(loss)}))


if __name__ == "__main__":
    main()
