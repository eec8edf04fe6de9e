#!/usr/bin/env python3
"""A run-nerf.py-shaped driver on the drop-in modules (no dataset files, no network): a teacher NeRF renders the
"photographs" of a Lego-style orbit, a student is trained on them with the reference's loop structure
(src/run-nerf.py:216-299: ray batch -> render_rays(train=True) -> MSE -> backward -> Adam -> ExponentialDecay ->
estimator.update_every_n_steps) and evaluated with render_frame + PSNR (run-nerf.py:140-190).

    python examples/train_synthetic.py [--estimator occgrid|stratified] [--iters 400] [--hw 64]
"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fs_nerf_amd  # noqa: E402,F401
from fs_nerf_amd.core import models as M  # noqa: E402
from fs_nerf_amd.core.loss import WeightNormRegularizer  # noqa: E402
from fs_nerf_amd.core.optim import FusedAdam  # noqa: E402
from fs_nerf_amd.core.scheduler import ExponentialDecay  # noqa: E402
from fs_nerf_amd.render import rendering as R  # noqa: E402
from fs_nerf_amd.render.occgrid import OccGridEstimator  # noqa: E402
from fs_nerf_amd.utils import utilities as U  # noqa: E402


def orbit_pose(phi_deg, theta_deg=50.0, radius=4.0311289):
    th, ph = theta_deg / 180.0 * math.pi, phi_deg / 180.0 * math.pi
    tr = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1.0]])
    rt = torch.tensor([[1, 0, 0, 0], [0, math.cos(th), -math.sin(th), 0], [0, math.sin(th), math.cos(th), 0], [0, 0, 0, 1.0]])
    rp = torch.tensor([[math.cos(ph), -math.sin(ph), 0, 0], [math.sin(ph), math.cos(ph), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    return rp @ (rt @ tr)


def make_model(seed, dev):
    torch.manual_seed(seed)
    m = M.NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    with torch.no_grad():
        m.sigma.weight.mul_(64.0)
        m.sigma.bias.add_(3.0)
    return m.to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--estimator", choices=("occgrid", "stratified"), default="occgrid")
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--cull-bf16", action="store_true",
                    help="opt-in: the occupancy estimator's visibility cull in single-pass bf16 (NeRF.cull_precision; not the parity mode)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    hwf = (a.hw, a.hw, 0.5 * a.hw / math.tan(0.5 * 0.6911112))
    near, far, step = 2.0, 6.0, 2e-2
    teacher = make_model(1, dev).eval()
    t_est = R.StratifiedEstimator(near, far, 64, 128)
    poses = [orbit_pose(phi) for phi in range(0, 360, 45)]
    ro, rd, gt = [], [], []
    with torch.no_grad():  # the "dataset": rays of every view (blender.py:174-191) and their colours
        for p in poses:
            o, d = U.get_rays(p, hwf, dev)
            ro.append(o.reshape(-1, 3))
            rd.append(d.reshape(-1, 3))
            gt.append(R.render_frame(hwf, near, far, p, 1 << 20, t_est, teacher, white_bkgd=True, device=dev)[0].reshape(-1, 3))
    ro, rd, gt = torch.cat(ro), torch.cat(rd), torch.cat(gt)

    model = make_model(2, dev).train()
    if a.cull_bf16:
        model.cull_precision = "bf16"
    if a.estimator == "occgrid":
        estimator = OccGridEstimator(roi_aabb=torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]), resolution=64, levels=1).to(dev)
    else:
        estimator = R.StratifiedEstimator(near, far, 64, 128)
    estimator.train()
    optimizer = FusedAdam(model.parameters(), lr=5e-4)  # torch.optim.Adam's arithmetic, one launch over flat arenas
    wnorm = WeightNormRegularizer(model.named_parameters(), reg="l2", reg_ratio=0.5, Td=a.iters)  # run-nerf.py:266-279
    alpha = 1e-5
    scheduler = ExponentialDecay(optimizer, a.iters, 5e-4, r=0.1)
    gen = torch.Generator(device=dev).manual_seed(0)

    def occ_eval_fn(x):
        return model(x) * step

    t0 = time.perf_counter()
    for k in range(a.iters):
        idx = torch.randint(0, ro.shape[0], (a.batch,), device=dev, generator=gen)
        (rgb, _, depth, _), _, _ = R.render_rays(ro[idx], rd[idx], estimator, model, train=True, white_bkgd=True,
                                                 render_step_size=step, device=dev)
        loss = torch.nn.functional.mse_loss(rgb, gt[idx])
        if wnorm.active(k):
            loss = loss + alpha * wnorm()
        loss.backward()
        optimizer.step()
        scheduler.step()
        optimizer.zero_grad()
        estimator.update_every_n_steps(step=k, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)
        if k % 100 == 0 or k == a.iters - 1:
            lv = float(loss.detach())
            print(f"iter {k:5d}  loss {lv:.5f}  psnr {-10 * math.log10(max(lv, 1e-10)):.2f} dB", flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.eval()
    estimator.eval()
    with torch.no_grad():
        img, _ = R.render_frame(hwf, near, far, orbit_pose(22.5), 1 << 20, estimator, model, white_bkgd=True,
                                render_step_size=step, device=dev)
        ref, _ = R.render_frame(hwf, near, far, orbit_pose(22.5), 1 << 20, t_est, teacher, white_bkgd=True, device=dev)
    mse = float(torch.nn.functional.mse_loss(img, ref))
    print(f"{a.iters} iterations of {a.batch} rays in {dt:.1f} s ({a.iters * a.batch / dt:,.0f} rays/s); held-out view PSNR "
          f"{-10 * math.log10(max(mse, 1e-10)):.2f} dB; frame as uint8: {tuple(R.to8b(img).shape)}")


if __name__ == "__main__":
    main()
