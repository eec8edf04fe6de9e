#!/usr/bin/env python3
"""Experiment (GPU): the opt-in `NeRF.cull_precision = "bf16"` (density pass of the occupancy estimator's visibility cull
in single-pass bf16, everything kept evaluated in the model's own mode) against the default on `bench.py --workload
occgrid`'s frame (800x800 orbit, half-full 128^3 grid, step 5e-3, opaque 8x256 net) and on a thin medium: frame time,
kept samples, deviation of the image.  profiles/EXPERIMENTS_r4.md section 12."""
import json, os, sys, time, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import bench as B
from fs_nerf_amd.render import rendering as Rm
from fs_nerf_amd.render.occgrid import OccGridEstimator

dev = torch.device("cuda:0")
est = OccGridEstimator(roi_aabb=torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]), resolution=B.OCC_RES, levels=1).to(dev)
ax = (torch.arange(B.OCC_RES) + 0.5) / B.OCC_RES * 3.0 - 1.5
x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
est.set_binaries(((x * x + y * y + z * z).sqrt() < B.OCC_RADIUS)[None])
est.eval()
out = {}
for medium, bias in (("opaque (sigma ~ 30)", 27.0), ("thin (sigma ~ 3)", 0.0)):
    res = {}
    for cull in (None, "bf16"):
        m = B.init_sd(42)
        with torch.no_grad():
            m.sigma.bias.add_(bias)
        m.cull_precision = cull
        m.to(dev).eval()
        frames, ts = [], []
        for i in range(4):
            pose = B.orbit_pose(4.0 * i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                img, depth = Rm.render_frame((B.H, B.W, B.FOCAL), B.NEAR, B.FAR, pose, 1 << 30, est, m, white_bkgd=True,
                                             render_step_size=B.OCC_STEP, device=dev)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            frames.append((img, depth))
        res[cull] = (frames, min(ts[1:]))
    d_img = max(float((a[0] - b[0]).abs().max()) for a, b in zip(res[None][0], res["bf16"][0]))
    m_img = sum(float((a[0] - b[0]).abs().mean()) for a, b in zip(res[None][0], res["bf16"][0])) / 4
    d_dep = max(float((a[1] - b[1]).abs().max()) for a, b in zip(res[None][0], res["bf16"][0]))
    px = sum(float(((a[0] - b[0]).abs().amax(-1) > 1e-4).float().mean()) for a, b in zip(res[None][0], res["bf16"][0])) / 4
    out[medium] = {"frame_ms_default": res[None][1], "frame_ms_bf16_cull": res["bf16"][1], "max_abs_rgb": d_img, "mean_abs_rgb": m_img,
                   "max_abs_depth": d_dep, "pixels_above_1e-4": px}
    print(medium, json.dumps(out[medium]), flush=True)
os.makedirs(os.path.join(R_, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R_, "gpurun_out", "exp_cull.json"), "w"), indent=1)
