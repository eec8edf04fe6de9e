#!/usr/bin/env python3
"""bench.py — rendered rays/sec of the fused HIP ray-rendering path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N=1 runs in this process.  N>1: under `python -m torch.distributed.run --nproc-per-node N ...` (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) every process is one rank; started plainly, `bench.py --gpus N` is its own
launcher: before anything touches the GPU it starts N fresh child processes (one per GPU, rendezvous on 127.0.0.1),
relays rank 0's JSON line and exits non-zero if any rank failed.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): synthetic Lego-style
orbit, 800x800 frames (focal 1111.11, near 2, far 6), 64 coarse + 128 importance samples per ray
(192 fine intervals), two 8x256 NeRF networks (coarse seed 42, fine seed 43; default nn.Linear init,
sigma head scaled so that alpha is not ~0), per-ray jitter off (inference).  One "step" = one whole
frame: ONE fused launch (ray generation from the pose -> sampler -> coarse density pass -> resampling -> fine
pass -> compositing) over the 640,000 rays, outputs resident in HBM.  Rays shard across ranks with
no data-path collective: every rank renders its own frames (weak scaling); value = all rays of all
ranks / max-over-ranks wall time.

`--scaling strong` (N>1): all ranks render row blocks of the SAME frame (`shard.shard_rows`, `cam_row0`): total work
fixed, value = frame rays / max-over-ranks time.  `--extras`: the timed launch also writes weights / alphas / trans /
sigmas / rgbs / edges per sample (the default frame path writes rgb_map, depth_map, opacity only, as the reference's
render_frame consumes only those, rendering.py:169-171).

The JSON line also carries
  roofline     - the fused kernel against the dense 16-bit MFMA peak (2.5 PFLOP/s), from HIP-event
                 timing of the launches inside the timed region; algorithmic FLOPs per ray =
                 64 x 982,528 (density-only coarse pass) + 192 x 1,186,816 (full fine pass), Linear
                 layers only, 2 FLOP per MAC; `traffic` = HBM bytes per launch from the committed
                 rocprofv3 PMC pass (profiles/), or null;
                 `clock_ghz` = the clock the chip held DURING the timed launches (s_memtime / s_memrealtime stamps of
                 workgroup 0, fsn_render_args.clock_out), `mfma_busy_derived` = issued MFMAs x 16 cycles / (1024 SIMDs x
                 that clock x the kernel time); `bare_stream` = the kernel's own GEMM code with nothing else in the
                 kernel (fsn_bench_bare_stream: 256 -> 256 layers back to back, weight stream walking the real blob),
                 launched on the same device right after the timed region, and `frac_of_bare_stream`; `fill` = the
                 second ceiling: L2 -> LDS bytes the weight streams move per launch (every 128-sample tile streams a
                 whole network) against the LDS-DMA fill rate MI355X_MICROARCH.md states and the bare stream reaches;
  other_workloads - (default N=1 run only) the other rows' workloads timed in-process after the headline: `train`
                 (configs[3]), `occgrid` (the reference's own render path), `train-occ` (its training loop body), `bf16`
                 (the headline frame in config 5's dtype) and `bf16_c5` (configs[4]: 1600x1600, 128+256): value,
                 ms_per_step and fraction of the MFMA peak each, so that the driver's one run records them;
  cpu_baseline - the CPU oracle (PyTorch CPU ops, same operator sequence as the reference) timed on
                 this box's host cores on a bounded sample of the same workload (rank 0, N=1 only);
  cpu_baseline_c1 - BASELINE.json configs[0] exactly (100x100 orbit pool, 4096-ray batch drawn with
                 manual_seed(42), 64 coarse samples only, 4x128 MLP, mask off) on the host cores, all threads and
                 one thread, with the fused HIP launch on the same batch beside it.

`--workload train` times the TRAINING step instead (BASELINE.json configs[3]): forward-facing
LLFF-style scene, 8 views of 378x504 (focal 407.6), NDC rays, near 0 / far 1, 64+128 samples, one 8x256 NeRF,
4096 rays per rank and step, step = ray batch -> render_rays(train=True) -> MSE -> backward -> ONE all-reduce of
the flat fp32 gradient bucket (RCCL) -> Adam -> ExponentialDecay; metric = trained rays/sec over all ranks.
"""
import argparse
import json
import math
import os
import sys
import time

import torch


def _gc_quiet():
    """Timed regions run without Python's cyclic collector, as `timeit` does: after the headline and the other workloads
    the process holds enough objects that ONE generation-2 pass took 81-117 ms - inside a 150-ms timed region of 30
    training steps it read as +3 ms per step (found with FSN_BENCH_DEBUG_STEPS=1; profiles/EXPERIMENTS_r4.md section 12).
    The garbage is collected before the region instead."""
    import gc
    gc.collect()
    gc.disable()


def _gc_restore():
    import gc
    gc.enable()

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
FOCAL = 0.5 * W / math.tan(0.5 * 0.6911112)  # = 1111.11 (camera_angle_x of the Lego set, blender.py:250-252)
NEAR, FAR = 2.0, 6.0
S, NI = 64, 128
FLOP_DENSITY = 982528
FLOP_FULL = 1186816
FLOP_PER_RAY = S * FLOP_DENSITY + (S + NI) * FLOP_FULL
PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, MI355X_MICROARCH.md


def init_sd(seed):
    """Reference-order default init (src/core/models.py:96-108) with torch only (no oracle import:
    the product path must not depend on oracle/)."""
    from fs_nerf_amd.core.models import NeRF
    torch.manual_seed(seed)
    m = NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True},
             dir_fn={"n_freqs": 4, "log_space": True})
    with torch.no_grad():
        m.sigma.weight.mul_(64.0)
        m.sigma.bias.add_(3.0)
    return m


def orbit_pose(phi_deg):
    th, ph = 50.0 / 180.0 * math.pi, phi_deg / 180.0 * math.pi
    tr = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0311289], [0, 0, 0, 1.0]])
    rt = torch.tensor([[1, 0, 0, 0], [0, math.cos(th), -math.sin(th), 0], [0, math.sin(th), math.cos(th), 0],
                       [0, 0, 0, 1.0]])
    rp = torch.tensor([[math.cos(ph), -math.sin(ph), 0, 0], [math.sin(ph), math.cos(ph), 0, 0], [0, 0, 1, 0],
                       [0, 0, 0, 1.0]])
    return rp @ (rt @ tr)


def cpu_model():
    """CPU model string of the host ('core count and CPU model printed')."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(target_s=15.0):
    """Oracle (kind 'port') on the host cores, bounded sample of the same workload."""
    from oracle import fsnerf_oracle as O
    # the GPU box gives one GPU's share of the host: 16 cores (os.cpu_count() reports the whole machine)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(int(os.environ.get("FSN_CPU_THREADS", min(avail, 16))))
    sd_c = O.init_nerf_state_dict(8, 256, [4], 10, 4, seed=42)
    sd_f = O.init_nerf_state_dict(8, 256, [4], 10, 4, seed=43)
    for sd in (sd_c, sd_f):
        sd["sigma.weight"] = sd["sigma.weight"] * 64.0
        sd["sigma.bias"] = sd["sigma.bias"] + 3.0
    cfg = dict(n_layers=8, skip=[4], n_freqs=10, n_freqs_dir=4, log_space=True)
    o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, 0.0), (H, W, FOCAL))
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)

    def run(n):
        idx = torch.arange(0, H * W, (H * W) // n)[:n]
        t0 = time.perf_counter()
        with torch.no_grad():
            O.render_rays_oracle(o[idx].contiguous(), d[idx].contiguous(), sd_c, sd_f, cfg, near=NEAR, far=FAR,
                                 n_samples=S, n_importance=NI, white_bkgd=True)
        return time.perf_counter() - t0

    run(256)  # warm-up
    t_probe = run(512)
    n = int(min(max(512, 512 * target_s / max(t_probe, 1e-3)), 65536))
    t = run(n)
    return {"value": n / t, "unit": "rays/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{n} rays of the 800x800 frame, 64+128 samples, two 8x256 nets, oracle/fsnerf_oracle.py "
                      f"(PyTorch CPU fp32), {t:.1f} s"}


def cpu_baseline_c1(dev=None, target_s=4.0):
    """BASELINE.json configs[0] as BASELINE.md section 3 / SURVEY 8d C1 specify it: the reference's own CPU-runnable
    case.  Oracle on the host cores with all threads and with one; the fused HIP launch on the same batch beside it."""
    from oracle import fsnerf_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthr = int(os.environ.get("FSN_CPU_THREADS", min(avail, 16)))
    hw, focal = 100, 0.5 * 100 / math.tan(0.5 * 0.6911112)  # 138.89 (blender.py:250-252)
    gen = torch.Generator().manual_seed(42)
    pool_o, pool_d = [], []
    for phi in torch.linspace(0.0, 360.0, 90).tolist():  # blender.py:260-277
        o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, phi), (hw, hw, focal))
        pool_o.append(o.reshape(-1, 3))
        pool_d.append(d.reshape(-1, 3))
    pool_o, pool_d = torch.cat(pool_o), torch.cat(pool_d)
    idx = torch.randint(0, pool_o.shape[0], (4096,), generator=gen)
    o, d = pool_o[idx].contiguous(), pool_d[idx].contiguous()
    u = torch.rand(4096, generator=gen)
    sd = O.init_nerf_state_dict(4, 128, [4], 10, 4, seed=42)  # n_layers = 4: the skip index is never reached
    sd["sigma.weight"] = sd["sigma.weight"] * 64.0
    sd["sigma.bias"] = sd["sigma.bias"] + 1.0
    cfg = dict(n_layers=4, skip=[4], n_freqs=10, n_freqs_dir=4, log_space=True)

    def timed(threads):
        torch.set_num_threads(threads)
        run = lambda: O.render_rays_oracle(o, d, sd, None, cfg, near=NEAR, far=FAR, n_samples=64, n_importance=0, u=u,
                                           white_bkgd=True)
        with torch.no_grad():
            run()
            t0 = time.perf_counter()
            run()
            t1 = time.perf_counter() - t0
            n = int(max(1, min(40, target_s / max(t1, 1e-3))))
            t0 = time.perf_counter()
            for _ in range(n):
                run()
            t = (time.perf_counter() - t0) / n
        return 4096 / t, n

    v_all, n_all = timed(nthr)
    v_one, n_one = timed(1)
    torch.set_num_threads(nthr)
    out = {"value": v_all, "unit": "rays/s", "cores": nthr, "cpu_model": cpu_model(), "value_1_thread": v_one, "kind": "port",
           "sample": f"configs[0]: 4096-ray batch of the 90 x 100x100 orbit pool (seed 42), 64 coarse samples, 4x128 "
                     f"MLP, mask off, oracle/fsnerf_oracle.py (PyTorch CPU fp32); {n_all} batches at {nthr} threads, "
                     f"{n_one} at 1 thread"}
    if dev is not None:  # the same batch through the fused launch (launch-bound at this size)
        from fs_nerf_amd.core.models import NeRF
        from fs_nerf_amd.render import rendering as Rm
        m = NeRF(3, 3, 4, 128, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
        m.load_state_dict(sd)
        m = m.to(dev).eval()
        est = Rm.StratifiedEstimator(NEAR, FAR, 64, 0)
        od, dd, ud = o.to(dev), d.to(dev), u.to(dev)
        with torch.no_grad():
            for _ in range(3):
                Rm.render_rays(od, dd, est, m, white_bkgd=True, device=dev, u=ud, want_extras=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                Rm.render_rays(od, dd, est, m, white_bkgd=True, device=dev, u=ud, want_extras=False)
            torch.cuda.synchronize()
        out["hip_value"] = 4096 * 50 / (time.perf_counter() - t0)
    return out


# ---------------------------------------------------------------- training workload (configs[3])
T_H, T_W, T_FOCAL, T_RAYS = 378, 504, 407.6, 4096  # LLFF images_8 geometry
FLOP_PER_RAY_TRAIN = S * FLOP_DENSITY + 3 * (S + NI) * FLOP_FULL  # density pass (no grad) + fwd + dgrad + wgrad
TRAIN_BYTES_PER_SAMPLE = {True: 10.4e3 + 9.9e3 + 21.8e3, False: 5.2e3 + 5.1e3 + 10.9e3}  # fwd + bwd + wgrad/heads, x3 / single-pass


def train_poses():
    """8 cameras on a 3x3-minus-centre planar grid at z = 0 looking down -z (forward-facing)."""
    out = []
    for iy in (-1, 0, 1):
        for ix in (-1, 0, 1):
            if ix == 0 and iy == 0:
                continue
            p = torch.eye(4)
            p[0, 3], p[1, 3] = 0.25 * ix, 0.25 * iy
            out.append(p)
    return out


def train_main(args, rank, world, dev, dist, backend):
    from fs_nerf_amd import ops, shard
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.core.scheduler import ExponentialDecay
    from fs_nerf_amd.render import rendering as Rm
    from fs_nerf_amd.utils import utilities as U
    torch.manual_seed(42)  # identical replicas on every rank
    model = NeRF(3, 3, 8, 256, (4,), precision=args.precision, pos_fn={"n_freqs": 10, "log_space": True},
                 dir_fn={"n_freqs": 4, "log_space": True})
    with torch.no_grad():
        model.sigma.weight.mul_(64.0)
        model.sigma.bias.add_(3.0)
    model.to(dev).train()
    est = Rm.StratifiedEstimator(0.0, 1.0, S, NI).train()  # NDC: near 0, far 1 (llff.py:51-53)
    est.generator = torch.Generator(device=dev).manual_seed(1000 + rank)
    from fs_nerf_amd.core.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=5e-4)  # one launch over the flat parameter / gradient arenas
    sched = ExponentialDecay(opt, 10000, 5e-4, r=0.1)
    ro, rd, _ = U.build_rays(train_poses(), (T_H, T_W, T_FOCAL), dev, ndc=True)  # dataset precompute (llff.py:59-90)
    gt = torch.rand(ro.shape[0], 3, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    gen = torch.Generator(device=dev).manual_seed(2000 + rank)  # every rank draws its own slice of the global batch

    def step():
        idx = torch.randint(0, ro.shape[0], (T_RAYS,), device=dev, generator=gen)
        opt.zero_grad()  # one fill of the flat gradient bucket; every p.grad is a view into it
        (rgb, _, _, _), _, _ = Rm.render_rays(ro[idx], rd[idx], est, model, train=True, white_bkgd=True, device=dev)
        loss = torch.nn.functional.mse_loss(rgb, gt[idx])
        loss.backward()
        opt.grads.allreduce(average=False)  # ONE RCCL all-reduce (sum) of the bucket; nothing else crosses GPUs
        opt.step(grad_div=float(world))     # ... the 1/world of the mean is folded into the Adam launch
        sched.step()
        return loss.detach()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _gc_quiet()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    _gc_restore()
    assert bool(torch.isfinite(loss))
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        value = world * args.steps * T_RAYS / dt
        achieved = FLOP_PER_RAY_TRAIN * value / world / 1e12  # per GPU, whole step (not one kernel)
        line = {
            "metric": "trained rays/sec (64+128 samples/ray, 8x256 MLP, fwd+bwd+Adam)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": "LLFF-style forward-facing scene, 8 views 378x504 (focal 407.6), NDC rays, near 0 / "
                                   "far 1, 64 coarse + 128 importance samples, one 8x256 NeRF, MSE + Adam + "
                                   "ExponentialDecay, flat-bucket gradient all-reduce",
                       "rays_per_step": T_RAYS, "parallelism": f"dp{world} (ray-batch data parallel)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS, "traffic": None, "kernel": "whole step",
                         "flop_per_ray": FLOP_PER_RAY_TRAIN,
                         # the step is co-limited by HBM: algorithmic bytes of the saved 16-bit parts
                         # (forward writes, backward writes, weight-gradient reads), float32-sized in the x3 modes
                         "hbm": {"bytes_per_step": TRAIN_BYTES_PER_SAMPLE[args.precision in ("fp16x3", "bf16x3")] * T_RAYS * (S + NI),
                                 "achieved_tbps": TRAIN_BYTES_PER_SAMPLE[args.precision in ("fp16x3", "bf16x3")] * (S + NI) * value / world / 1e12,
                                 "peak_tbps": 8.0}},
            "loss": float(loss),
        }
        return line
    return None


# ---------------------------------------------------------------- occupancy-grid workload (the reference's own render path)
OCC_RES, OCC_STEP, OCC_RADIUS = 128, 5e-3, 1.477  # sphere holding half of the +-1.5 box's cells; run-nerf.py's step size


def occ_main(args, rank, world, dev, dist, backend):
    """`--workload occgrid`: render_frame with the occupancy estimator in the slot (rendering.py:58-107 as the reference
    runs it, run-nerf.py:96-98) on a synthetic HALF-EMPTY grid (128^3 cells, the cells inside a sphere of radius 1.477 =
    50 % occupied), 800x800 orbit frames, render_step_size 5e-3, one 8x256 network: ONE launch per frame
    (csrc/render_occ.hip).  Reports rays/s, the mean marched / kept samples per ray, and the launch against the MFMA
    roofline with algorithmic FLOPs = marched samples x 982,528 (density pass) + kept samples x 1,186,816 (full pass);
    the unfused sequence (5 launches + a host sync) is timed beside it."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.render import rendering as Rm
    from fs_nerf_amd.render.occgrid import OccGridEstimator
    from fs_nerf_amd.utils import utilities as U
    model = init_sd(42)
    with torch.no_grad():
        model.sigma.bias.add_(27.0)  # an opaque medium (sigma ~ 30): rays saturate after ~60 of their ~350 marched samples
    model.precision = args.precision
    model.to(dev).eval()
    est = OccGridEstimator(roi_aabb=torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]), resolution=OCC_RES, levels=1).to(dev)
    ax = (torch.arange(OCC_RES) + 0.5) / OCC_RES * 3.0 - 1.5
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    est.set_binaries(((x * x + y * y + z * z).sqrt() < OCC_RADIUS)[None])
    est.eval()
    occupied = float(est.binaries.float().mean())
    ops.launch_timer = ev_raw = []

    def step(i, timed):
        pose = orbit_pose(((i * world + rank) % 90) * 4.0)
        n0 = len(ev_raw)
        with torch.no_grad():
            out = Rm.render_frame((H, W, FOCAL), NEAR, FAR, pose, 1 << 30, est, model, white_bkgd=True,
                                  render_step_size=OCC_STEP, device=dev)
        if not timed:
            del ev_raw[n0:]
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    _gc_quiet()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    _gc_restore()
    assert bool(torch.isfinite(out[0]).all()) and ops.range_ok(dev) and model.precision == args.precision
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev_raw) / max(len(ev_raw), 1)
    ops.launch_timer = None
    if rank == 0:
        # sample counts of the last timed frame + the unfused sequence on the same frame (outside the timed region)
        pose = orbit_pose((((args.warmup + args.steps - 1) * world + rank) % 90) * 4.0)
        cam = (pose, H, W, FOCAL, 0, H, dev)
        _, _, _, cnt = ops.render_occ_fused(model.packed(), None, None, aabb=est.aabb, res=est.resolution, levels=1,
                                            bits=est.bits, near_plane=0.0, far_plane=1e10, step=OCC_STEP,
                                            max_steps=est.max_steps(OCC_STEP), bkgd=(1.0, 1.0, 1.0), camera=cam,
                                            want_counts=True)
        n_cand, n_kept = float(cnt["n_cand"].float().mean()), float(cnt["n_kept"].float().mean())
        o, d = U.get_rays(pose, (H, W, FOCAL), dev)
        o, d = o.reshape(-1, 3), d.reshape(-1, 3)
        torch.cuda.synchronize()
        fused_sampler = Rm.FUSED_OCC_SAMPLER
        Rm.FUSED_OCC_SAMPLER = False  # (with it on, 640,000 rays take the sampler mode of k_render_occ - the
        try:                          # comparison would partly be the kernel against itself)
            tu = time.perf_counter()
            with torch.no_grad():
                (rgb_u, _, _, _), _, _ = Rm.render_rays(o, d, est, model, white_bkgd=True, render_step_size=OCC_STEP, device=dev)
            torch.cuda.synchronize()
            unfused_ms = (time.perf_counter() - tu) * 1e3
        finally:
            Rm.FUSED_OCC_SAMPLER = fused_sampler
        same = bool(torch.equal(rgb_u, out[0].reshape(-1, 3)))
        flop_per_ray = n_cand * FLOP_DENSITY + n_kept * FLOP_FULL
        achieved = flop_per_ray * H * W / (kern_ms * 1e-3) / 1e12
        value = world * args.steps * H * W / dt
        line = {
            "metric": "rendered rays/sec (occupancy-grid estimator, render_step_size 5e-3, 8x256 MLP)", "value": value,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "Lego-style orbit 800x800 (focal 1111.11), occupancy estimator: 128^3 cells over the "
                                   "+-1.5 box, the cells inside a sphere of radius 1.477 occupied, render_step_size 5e-3, "
                                   "early_stop_eps 1e-4, one 8x256 NeRF (seed 42, sigma head x64 + 30: opaque), render_frame = ONE launch (march -> "
                                   "density pass -> visibility -> full pass -> packed integration), rays generated in it",
                       "rays_per_step": H * W, "occupied_cells": occupied, "marched_samples_per_ray": n_cand,
                       "kept_samples_per_ray": n_kept, "parallelism": f"rays x{world} (no data-path collective)",
                       "unfused_sequence_ms": unfused_ms, "fused_equals_unfused_bitwise": same},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS, "traffic": None, "kernel": "k_render_occ", "kernel_ms": kern_ms,
                         "flop_per_ray": flop_per_ray,
                         "passes_per_product": 3 if args.precision.endswith("x3") else 1},
        }
        return line
    return None


def train_occ_main(args, rank, world, dev, dist, backend):
    """`--workload train-occ`: the body of the reference's training loop as it stands (run-nerf.py:243-295): render_rays
    with the OCCUPANCY estimator in the slot, train=True (march -> density pass under no_grad -> visibility cull ->
    model(x, d) with gradients -> packed integration), MSE, backward, Adam, estimator.update_every_n_steps(step,
    occ_eval_fn = model(x) * step_size).  4096 random rays per step out of eight 800x800 orbit views, render_step_size
    5e-3, one 8x256 network.  For a DEFINED workload the grid the rays are sampled from is held fixed (the half-full
    sphere grid of `--workload occgrid`) and the medium is thin (sigma ~ 3: the visibility cull keeps most marched
    samples); the every-16th-step grid update is executed on a second estimator of the same size so that its cost is in the
    timed region.  Unfused: several launches and one host sync (the data-dependent sample count) per step."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.optim import FusedAdam
    from fs_nerf_amd.render import rendering as Rm
    from fs_nerf_amd.render.occgrid import OccGridEstimator
    from fs_nerf_amd.utils import utilities as U
    model = init_sd(42)
    model.precision = args.precision
    model.cull_precision = getattr(args, "cull_precision", None)
    model.to(dev).train()
    box = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    est = OccGridEstimator(roi_aabb=box, resolution=OCC_RES, levels=1).to(dev)
    ax = (torch.arange(OCC_RES) + 0.5) / OCC_RES * 3.0 - 1.5
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    est.set_binaries(((x * x + y * y + z * z).sqrt() < OCC_RADIUS)[None])
    est.train()
    est.generator = torch.Generator(device=dev).manual_seed(1000 + rank)
    shadow = OccGridEstimator(roi_aabb=box, resolution=OCC_RES, levels=1).to(dev).train()
    shadow.generator = torch.Generator(device=dev).manual_seed(3000 + rank)
    opt = FusedAdam(model.parameters(), lr=5e-4)
    rays = [U.get_rays(orbit_pose(45.0 * k), (H, W, FOCAL), dev) for k in range(8)]
    ro = torch.cat([o.reshape(-1, 3) for o, _ in rays])
    rd = torch.cat([d.reshape(-1, 3) for _, d in rays])
    gt = torch.rand(ro.shape[0], 3, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    gen = torch.Generator(device=dev).manual_seed(2000 + rank)
    kept = []

    def occ_eval_fn(xx):
        return model(xx) * OCC_STEP  # (run-nerf.py:288-289)

    def step(i):
        idx = torch.randint(0, ro.shape[0], (T_RAYS,), device=dev, generator=gen)
        opt.zero_grad()
        (rgb, _, _, _), ri, _ = Rm.render_rays(ro[idx], rd[idx], est, model, train=True, white_bkgd=True,
                                               render_step_size=OCC_STEP, device=dev)
        loss = torch.nn.functional.mse_loss(rgb, gt[idx])
        loss.backward()
        opt.grads.allreduce(average=False)
        opt.step(grad_div=float(world))
        with torch.no_grad():
            shadow.update_every_n_steps(step=256 + i, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)  # past the warm-up
        kept.append(ri.numel())
        return loss.detach()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    del kept[:]
    dbg = os.environ.get("FSN_BENCH_DEBUG_STEPS")
    ms0 = torch.cuda.memory_stats(dev) if dbg else None
    if dbg:
        import gc
        gc_log, gc_t = [], [0.0]

        def _gc_cb(phase, info):
            if phase == "start":
                gc_t[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], round((time.perf_counter() - gc_t[0]) * 1e3, 1), info["collected"]))
        gc.callbacks.append(_gc_cb)
    per = []
    if not os.environ.get("FSN_BENCH_KEEP_GC"):
        _gc_quiet()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts = time.perf_counter()
        loss = step(args.warmup + i)
        per.append((time.perf_counter() - ts) * 1e3)
    barrier()
    dt = time.perf_counter() - t0
    _gc_restore()
    if dbg:  # host-side time of every step (no sync) and what the caching allocator did meanwhile
        ms1 = torch.cuda.memory_stats(dev)
        print("[debug] train-occ", model.cull_precision, "host ms per step", [round(v, 1) for v in per], {k: ms1[k] - ms0[k] for k in
              ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated")}, "gc", gc_log, file=sys.stderr, flush=True)
        gc.callbacks.remove(_gc_cb)
    assert bool(torch.isfinite(loss))
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        value = world * args.steps * T_RAYS / dt
        n_kept = sum(kept) / max(len(kept), 1) / T_RAYS
        with torch.no_grad():  # marched samples per ray on a batch drawn like the timed ones
            est.eval()
            idx = torch.randint(0, ro.shape[0], (T_RAYS,), device=dev, generator=gen)
            _, _, _, cnt = ops.render_occ_fused(model.packed(), ro[idx].contiguous(), rd[idx].contiguous(), aabb=est.aabb,
                                                res=est.resolution, levels=1, bits=est.bits, near_plane=0.0, far_plane=1e10,
                                                step=OCC_STEP, max_steps=est.max_steps(OCC_STEP), bkgd=(1.0, 1.0, 1.0),
                                                want_counts=True)
        n_cand = float(cnt["n_cand"].float().mean())
        flop_per_ray = n_cand * FLOP_DENSITY + 3 * n_kept * FLOP_FULL  # density pass + fwd + dgrad + wgrad on the kept samples
        achieved = flop_per_ray * value / world / 1e12
        line = {
            "metric": "trained rays/sec (occupancy-grid estimator, render_step_size 5e-3, 8x256 MLP, fwd+bwd+Adam+grid update)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "run-nerf.py:243-295 as it stands: 4096 random rays of eight 800x800 orbit views, occupancy "
                                   "estimator (128^3 cells, sphere of radius 1.477 occupied, held fixed), render_step_size 5e-3, "
                                   "one 8x256 NeRF (seed 42), MSE + fused Adam, update_every_n_steps on a second estimator",
                       "rays_per_step": T_RAYS, "marched_samples_per_ray": n_cand, "kept_samples_per_ray": n_kept,
                       "cull_precision": model.cull_precision or "the model's (parity)",
                       "parallelism": f"dp{world} (ray-batch data parallel)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS, "traffic": None, "kernel": "whole step", "flop_per_ray": flop_per_ray},
            "loss": float(loss),
        }
        return line
    return None


# ---------------------------------------------------------------- N>1 self-launcher
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout_s=None):
    """Start `n` fresh child processes of this script, one rank each (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set per child), wait for all of them, relay rank 0's stdout.  The parent never touches the GPU (it
    must not: a process that has initialised HIP is not allowed to be replaced or to fork workers on this pool).
    Returns the exit code: 0 iff every rank exited 0."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FSN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_end = None if timeout_s is None else time.time() + timeout_s
    rc = 0
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            break  # a rank died: the others would wait for it in the rendezvous / a collective forever
        if t_end is not None and time.time() > t_end:
            rc = 124
            break
        time.sleep(0.1)
    for p in procs:
        if p.poll() is None:
            p.kill()
            p.wait()
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    for r, p in enumerate(procs):
        if p.returncode not in (0, -9):  # -9: killed above because another rank had failed
            rc = rc or (p.returncode if p.returncode > 0 else 1)
            print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr, flush=True)
    if rc == 0 and any(p.returncode != 0 for p in procs):
        rc = 1
    if rc == 0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    return rc


def dry_run(args, rank, world):
    """Launcher self-test (no GPU work, NOT a measurement): every rank joins a gloo group on the CPU, the max-over-ranks
    reduction used for timing is exercised, rank 0 prints one JSON line describing what each rank saw."""
    import torch.distributed as dist
    seen = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "world": world}
    if os.environ.get("FSN_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    ranks = torch.zeros(world, dtype=torch.int64)
    ranks[rank] = 1 + int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks, op=dist.ReduceOp.SUM)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_over_ranks": float(t.item()), "local_ranks_plus_1": ranks.tolist(), **seen}), flush=True)


FUSED_SOURCES = ("render.hip", "mlp_dev.hpp", "kloop_gen.hpp", "ray_dev.hpp", "mlp_layout.hpp", "common.hpp")


def csrc_sha():
    """Hash of the sources k_render_fused is built from (render.hip and the headers it includes): a PMC summary is only
    quoted when it was collected on exactly this code."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "fs-nerf_amd", "csrc")
    for f in FUSED_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]




def measured_traffic(precision):
    """HBM-side bytes per fused launch from the committed rocprofv3 PMC passes (tools/run_pmc.sh ->
    tools/summarize_pmc.py -> profiles/*_pmc_summary.json).  Returned only when the summary was collected for this
    precision mode on exactly these kernel sources; otherwise null + the reason (never a stale number)."""
    import glob
    best = None
    for pj in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            j = json.load(open(pj))
        except Exception:
            continue
        if j.get("kernel") != "k_render_fused" or j.get("precision") != precision:
            continue
        best = (pj, j)
        if j.get("csrc_sha") == csrc_sha() and j.get("hbm_bytes_per_launch") is not None:
            src = {"file": os.path.relpath(pj, ROOT), "csrc_sha": j["csrc_sha"], "precision": precision,
                   "l2_hit_rate": j.get("l2_hit_rate"), "clock_ghz": j.get("clock_ghz"),
                   "mfma_busy": j.get("mfma_busy_frac")}
            return j["hbm_bytes_per_launch"], src
    if best is None:
        return None, {"note": f"no PMC summary for precision {precision} under profiles/"}
    return None, {"note": "kernel sources changed since " + os.path.relpath(best[0], ROOT) + " was collected",
                  "stale_value": best[1].get("hbm_bytes_per_launch")}


def bare_stream_measure(ops, pm, dev, layers=4000):
    """roofline.bare_stream, measured on THIS device right after the timed region (VERDICT r3 next #2b/c): the render
    kernels' own GEMM code - hand-scheduled blocks, pair epilogues, weight-stream ring walking the hidden phases of the
    packed fine network - as 256 -> 256 layers back to back (fsn_bench_bare_stream), ~50 ms."""
    ops.bench_bare_stream(pm, 200)  # warm-up (clocks, instruction cache)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n_wg, clk = ops.bench_bare_stream(pm, layers)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = clk.double().sum(0).tolist()
    mfma = n_wg * 8 * layers * 384.0
    return {"what": "the render kernels' own MFMA stream, bare: 256 -> 256 hidden layers back to back through gemm_layer "
                    "(generated GEMM-pair blocks, pair epilogues, LDS-DMA weight ring walking the hidden phases of the "
                    "packed fine network) on fixed activations; no encodings, heads, samplers, compositing, tile tails",
            "mfma_tflops": mfma * 16384.0 / (ms * 1e-3) / 1e12, "algorithmic_tflops": mfma * 16384.0 / 3.0 / (ms * 1e-3) / 1e12,
            "clock_ghz": c[0] / c[1] * 0.1 if c[1] > 0 else None, "ms": ms, "layers_per_workgroup": layers, "workgroups": n_wg,
            "mfma_busy_derived": (8 * layers * 384.0 * 16.0 / 4.0) / (c[0] / (n_wg * 8)) if c[0] > 0 else None,
            "fill_tbps": n_wg * layers * 262144.0 / (ms * 1e-3) / 1e12,
            "measured": "in this run, on this device, after the timed region"}


def other_workloads(args, dev, dist, backend):
    """The other rows' workloads in-process (N=1 default run), so that the driver's ONE run records them (VERDICT r3
    missing #3): a few steps each, ~10 s together.  Each entry: value, unit, ms_per_step, frac (of the 2.5 PFLOP/s MFMA
    peak, algorithmic FLOPs of that workload), steps."""
    import types
    from fs_nerf_amd import ops
    from fs_nerf_amd.render import rendering as Rm
    out = {}

    def run(name, fn, steps, warmup, precision, **extra):
        a = types.SimpleNamespace(steps=steps, warmup=warmup, precision=precision, **extra)
        try:
            line = fn(a, 0, 1, dev, dist, backend)
            out[name] = {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
                         "frac": line["roofline"]["frac"], "steps": steps, "dtype": precision, "metric": line["metric"]}
            for k in ("marched_samples_per_ray", "kept_samples_per_ray", "unfused_sequence_ms", "fused_equals_unfused_bitwise",
                      "cull_precision"):
                if k in line["config"]:
                    out[name][k] = line["config"][k]
        except Exception as e:  # a failing side line must not take the headline down; it is recorded as such
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()

    run("train", train_main, 20, 3, args.precision)
    run("occgrid", occ_main, 1, 1, args.precision)
    run("train-occ", train_occ_main, 30, 5, args.precision)  # (round 3's run length: the kept-sample count - and with it the step time - falls as the network sharpens: 172 per ray after 8 steps, ~50 after 35, ~30 after 48)

    # OPT-IN, not a parity mode (NeRF.cull_precision = "bf16": the visibility cull's density pass in single-pass bf16 as
    # its own launch, everything it keeps in the model's mode): the same two workloads, with the image's deviation from
    # the default path measured here
    run("train-occ_cull-bf16", train_occ_main, 30, 5, args.precision, cull_precision="bf16")

    def occ_frame_cull(name):
        try:
            from fs_nerf_amd.render.occgrid import OccGridEstimator
            est = OccGridEstimator(roi_aabb=torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]), resolution=OCC_RES, levels=1).to(dev)
            ax = (torch.arange(OCC_RES) + 0.5) / OCC_RES * 3.0 - 1.5
            x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
            est.set_binaries(((x * x + y * y + z * z).sqrt() < OCC_RADIUS)[None])
            est.eval()
            imgs, ms = {}, {}
            for cull in (None, "bf16"):
                m = init_sd(42)
                with torch.no_grad():
                    m.sigma.bias.add_(27.0)
                m.precision, m.cull_precision = args.precision, cull
                m.to(dev).eval()
                with torch.no_grad():
                    for i in range(2):  # (the second frame is the timed one)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        imgs[cull] = Rm.render_frame((H, W, FOCAL), NEAR, FAR, orbit_pose(4.0), 1 << 30, est, m, white_bkgd=True,
                                                     render_step_size=OCC_STEP, device=dev)
                        torch.cuda.synchronize()
                        ms[cull] = (time.perf_counter() - t0) * 1e3
            out[name] = {"value": H * W / (ms["bf16"] * 1e-3), "unit": "rays/s", "ms_per_step": ms["bf16"], "steps": 1,
                         "dtype": f"{args.precision}, cull bf16", "default_path_ms_same_frame": ms[None],
                         "max_abs_rgb_vs_default": float((imgs["bf16"][0] - imgs[None][0]).abs().max()),
                         "max_abs_depth_vs_default": float((imgs["bf16"][1] - imgs[None][1]).abs().max()),
                         "metric": "rendered rays/sec (occupancy-grid estimator frame of `occgrid`, opt-in bf16 visibility cull: "
                                   "sampler launch + full pass + integration)"}
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.synchronize()

    occ_frame_cull("occgrid_cull-bf16")

    def frame(name, hw, s_, ni_, steps):
        try:
            coarse, fine = init_sd(42), init_sd(43)
            for m in (coarse, fine):
                m.precision = "bf16"
                m.to(dev).eval()
            est = Rm.StratifiedEstimator(NEAR, FAR, s_, ni_)
            focal = 0.5 * hw / math.tan(0.5 * 0.6911112)
            ops.launch_timer = ev = []
            with torch.no_grad():
                Rm.render_frame((hw, hw, focal), NEAR, FAR, orbit_pose(0.0), 1 << 30, est, coarse, white_bkgd=True, device=dev, model_fine=fine)
                torch.cuda.synchronize()
                del ev[:]
                _gc_quiet()
                t0 = time.perf_counter()
                for i in range(steps):
                    rgb, _ = Rm.render_frame((hw, hw, focal), NEAR, FAR, orbit_pose(4.0 * (i + 1)), 1 << 30, est, coarse,
                                             white_bkgd=True, device=dev, model_fine=fine)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                _gc_restore()
            kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
            ops.launch_timer = None
            assert bool(torch.isfinite(rgb).all())
            flop = s_ * FLOP_DENSITY + (s_ + ni_) * FLOP_FULL
            out[name] = {"value": steps * hw * hw / dt, "unit": "rays/s", "ms_per_step": dt / steps * 1e3,
                         "frac": flop * hw * hw / (kern_ms * 1e-3) / 1e12 / PEAK_TFLOPS, "steps": steps, "dtype": "bf16",
                         "metric": f"rendered rays/sec ({s_}+{ni_} samples/ray, 8x256 MLP), {hw}x{hw} frame, bf16 single pass"}
        except Exception as e:
            ops.launch_timer = None
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    frame("bf16", 800, S, NI, 2)          # the headline frame in config 5's dtype
    frame("bf16_c5", 1600, 128, 256, 1)   # BASELINE configs[4]: 1600x1600, 128+256, bf16 weights / activations
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=("render", "train", "occgrid", "train-occ"), default="render")
    ap.add_argument("--precision", default=os.environ.get("FSN_BENCH_PREC", "fp16x3"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the in-process train / occgrid / train-occ / bf16 lines")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N>1: every rank its own frames (weak), or row blocks of the same frame (strong)")
    ap.add_argument("--cull-precision", choices=("bf16",), default=None,
                    help="train-occ: NeRF.cull_precision (opt-in single-pass density pass of the visibility cull; not a parity mode)")
    ap.add_argument("--extras", action="store_true", help="also write the per-sample outputs (weights, ...) in the timed launch")
    ap.add_argument("--raw-launch", action="store_true",
                    help="time ops.render_fused directly instead of the product path render_frame (A/B of the host path)")
    ap.add_argument("--dry-run", action="store_true", help="launcher self-test on the CPU (gloo); not a measurement")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 20 if args.workload.startswith("train") else 3
    if args.warmup is None:
        args.warmup = 3 if args.workload.startswith("train") else 1

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly: become the launcher (nothing in this process has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (or none, and let "
                         f"bench.py launch them)")
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs the GPU (no CPU fallback)"
    local = local % max(torch.cuda.device_count(), 1)  # (rehearsals put several ranks on one card)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    backend = os.environ.get("FSN_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" for rehearsals
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.workload in ("train", "occgrid", "train-occ"):
        line = {"train": train_main, "occgrid": occ_main, "train-occ": train_occ_main}[args.workload](args, rank, world, dev, dist, backend)
        if line is not None:
            print(json.dumps(line), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    from fs_nerf_amd import ops
    from fs_nerf_amd.render import rendering as Rm
    coarse, fine = init_sd(42), init_sd(43)
    for m in (coarse, fine):
        m.precision = args.precision
        m.to(dev).eval()
    est = Rm.StratifiedEstimator(NEAR, FAR, S, NI)
    # (the per-layer activation scales of the fp16x3 inference path are calibrated on samples of a frame's own rays,
    # as render_frame does on its first call)
    probe = lambda: Rm._probe_on_rays(None, None, (orbit_pose(0.0), H, W, FOCAL, 0, H, dev), NEAR, FAR)
    pc, pf = coarse.packed(probe), fine.packed(probe)
    torch.cuda.synchronize()

    from fs_nerf_amd import shard
    strong = args.scaling == "strong" and world > 1
    row0, nrows = shard.shard_rows(H, rank, world) if strong else (0, H)
    ev = []
    ops.launch_timer = ev_raw = []  # ops.render_fused brackets its launch with HIP events on the launch stream
    clock_buf = torch.zeros(2, dtype=torch.int64, device=dev)  # (s_memtime, s_memrealtime) sums of the timed launches

    def step(i, timed):
        # weak scaling: ranks render different frames of the 90-frame orbit (blender.py:260-277); strong: the same one
        pose = orbit_pose(((i * (1 if strong else world) + (0 if strong else rank)) % 90) * 4.0)
        n0 = len(ev_raw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if args.raw_launch or strong or args.extras:
            # rays are generated inside the launch from (pose, pixel index): a frame (or a rank's row block) is one launch
            rgb, op, depth, _ = ops.render_fused(pc, pf, None, None, near=NEAR, far=FAR, n_samples=S, n_importance=NI,
                                                 bkgd=(1.0, 1.0, 1.0), want_extras=args.extras,
                                                 camera=(pose, H, W, FOCAL, row0, nrows, dev),
                                                 two_phase=os.environ.get("FSN_TWO_PHASE", "1") == "1")
        else:
            # the PRODUCT path (rendering.py:110-177): render_frame -> one fused launch, range-guard read-back, depth clamp
            rgb, depth = Rm.render_frame((H, W, FOCAL), NEAR, FAR, pose, 1 << 30, est, coarse, white_bkgd=True,
                                         device=dev, model_fine=fine)
        e1.record()
        if timed:
            ev.append((e0, e1))
        else:
            del ev_raw[n0:]
        return rgb, depth

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    events0 = coarse.range_events + fine.range_events
    ops.clock_buffer = clock_buf
    _gc_quiet()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i, True)
    barrier()
    dt = time.perf_counter() - t0
    _gc_restore()
    ops.clock_buffer = None
    assert bool(torch.isfinite(out[0]).all())
    assert ops.range_ok(dev), "an fp16-mode launch reported activations outside the fp16 range"
    assert coarse.precision == args.precision and fine.precision == args.precision, "no range fallback in the timed region"
    assert coarse.range_events + fine.range_events == events0, "no re-calibration in the timed region"
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    kern_ms = sum(a.elapsed_time(b) for a, b in ev_raw) / max(len(ev_raw), 1)  # the fused launch alone
    ops.launch_timer = None

    if rank == 0:
        rays = (1 if strong else world) * args.steps * H * W
        value = rays / dt
        achieved = FLOP_PER_RAY * nrows * W / (kern_ms * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(args.precision)
        passes = 3 if args.precision.endswith("x3") else (2 if args.precision.endswith("x2") else 1)
        clk = [int(v) for v in clock_buf.tolist()]
        clock_ghz = clk[0] / clk[1] * 0.1 if clk[1] > 0 else None
        mfma_per_launch = passes * FLOP_PER_RAY * nrows * W / 16384.0  # v_mfma_f32_16x16x32: 16,384 FLOP, 16 pipe cycles
        busy = mfma_per_launch * 16.0 / (1024 * clock_ghz * 1e9 * kern_ms * 1e-3) if clock_ghz else None
        bare = bare_stream_measure(ops, pf, dev) if args.precision in ("fp16x3", "bf16x3") else None
        kern_mfma_tflops = passes * achieved
        # the second ceiling: every 128-sample tile streams a whole network L2 -> LDS
        ub = 2 if passes > 1 else 1
        tiles_c, tiles_f = nrows * W * S // 128, nrows * W * (S + NI) // 128
        fill_bytes = 16384.0 * ub * (tiles_c * 60 + tiles_f * 72.5)  # 960 hidden / 1160 total units of 1 KiB per part
        if passes == 1:
            fill_bytes /= 2  # (single-pass modes: 256-sample tiles)
        fill = {"bytes_per_launch": fill_bytes, "tbps": fill_bytes / (kern_ms * 1e-3) / 1e12,
                "guide_ceiling_tbps": [6.4, 6.8],
                "bare_stream_tbps": None if bare is None else bare["fill_tbps"],
                "note": "L2 -> LDS bytes of the weight streams (LDS-DMA), not HBM traffic: the streams hit in L2"}
        line = {
            "metric": "rendered rays/sec (64+128 samples/ray, 8x256 MLP)", "value": value, "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "Lego-style orbit 800x800 (focal 1111.11, near 2, far 6), 64 coarse + 128 "
                                   "importance samples (192 fine intervals), two 8x256 NeRF nets (seeds 42/43), "
                                   "ONE fused launch per 640,000-ray frame, rays generated in the launch",
                       "rays_per_step": H * W,
                       "path": "ops.render_fused (raw launch)" if (args.raw_launch or strong or args.extras) else
                               "render.rendering.render_frame (the product path: launch + range-guard read-back + depth clamp)",
                       "outputs": "rgb_map, depth_map, opacity + per-sample weights / alphas / trans / sigmas / rgbs / edges"
                                  if args.extras else "rgb_map, depth_map, opacity (per-sample weights are NOT written in "
                                  "the timed launch: frame rendering consumes only rgb / depth, rendering.py:169-171; "
                                  "--extras times them too)",
                       "parallelism": (f"row blocks of one frame x{world}" if strong else f"rays x{world}") +
                                      " (no data-path collective)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_render_fused", "kernel_ms": kern_ms, "step_ms_events": step_ms,
                         "flop_per_ray": FLOP_PER_RAY,
                         # from the PMC summary stamped with these kernel sources (null when stale): the clock the chip
                         # held and the matrix-pipe busy fraction; x3 modes issue 3 MFMAs per algorithmic product
                         # measured IN this run: the clock the chip held during the timed launches (stamps of workgroup
                         # 0) and the matrix-pipe occupancy that follows from it; the PMC pass's values beside them
                         "clock_ghz": clock_ghz, "mfma_busy_derived": busy,
                         "pmc_clock_ghz": traffic_src.get("clock_ghz"), "pmc_mfma_busy": traffic_src.get("mfma_busy"),
                         "passes_per_product": passes, "mfma_tflops_issued": kern_mfma_tflops,
                         "bare_stream": bare,
                         "frac_of_bare_stream": None if bare is None else kern_mfma_tflops / bare["mfma_tflops"],
                         "fill": fill},
        }
        if world == 1 and not args.no_cpu_baseline and not args.no_other_workloads:
            line["other_workloads"] = other_workloads(args, dev, dist, backend)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(10.0)
            line["cpu_baseline_c1"] = cpu_baseline_c1(dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
