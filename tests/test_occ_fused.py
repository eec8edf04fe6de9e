"""The reference's own render path - occupancy estimator in the `estimator` slot (src/render/rendering.py:58-107,
src/run-nerf.py:96-98, 288-295) - as ONE launch (csrc/render_occ.hip, VERDICT r2 missing #1): grid march -> density
pass -> visibility cull -> full pass -> packed integration per batch of rays inside persistent workgroups, no host sync.
The fused launch must be the unfused sequence (fsn_occgrid_march, fsn_mlp_fwd, fsn_packed_visibility, fsn_mlp_fwd,
fsn_composite_packed_fwd - itself checked against the oracle in test_occgrid.py / test_trained_parity.py) bit for
bit: same sampling rule, every sample evaluated independently of its tile position, every ray integrated by one wave
with the same arithmetic - whatever the batches look like (rays carried over, empty rays, ragged ends)."""
import numpy as np
import pytest
import torch

from oracle import fsnerf_oracle as O

pytestmark = pytest.mark.gpu
AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import fs_nerf_amd  # noqa: F401
    return torch.device("cuda:0")


def make_model(L, D, seed, dev, precision="fp16x3", gain=64.0, shift=3.0):
    from fs_nerf_amd.core.models import NeRF
    sd = O.init_nerf_state_dict(L, D, [4], 10, 4, seed=seed)
    sd["sigma.weight"] *= gain
    sd["sigma.bias"] += shift
    m = NeRF(3, 3, L, D, (4,), precision=precision, pos_fn={"n_freqs": 10, "log_space": True},
             dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    return m.to(dev).eval()


def sphere_grid(res, levels, dev, radius=0.9):
    from fs_nerf_amd.render.occgrid import OccGridEstimator
    est = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=res, levels=levels).to(dev)
    b = torch.zeros(levels, res, res, res, dtype=torch.bool)
    for lvl in range(levels):
        ax = (torch.arange(res) + 0.5) / res * 3.0 * 2 ** lvl - 1.5 * 2 ** lvl
        x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
        b[lvl] = (x * x + y * y + z * z).sqrt() < radius * (1.0 + 0.5 * lvl)
    est.set_binaries(b)
    return est.eval()


def orbit_rays(n, seed, hw=64):
    gen = torch.Generator().manual_seed(seed)
    o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, float(torch.rand(1, generator=gen)) * 360.0), (hw, hw, hw * 1.39))
    idx = torch.randperm(hw * hw, generator=gen)[:n]
    return o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous()


def unfused(fn):
    """`fn()` with every fused form of the occupancy path switched off: march -> density pass -> visibility -> compaction
    -> full pass -> packed integration as separate launches (the sequence the fused forms must reproduce bit for bit)."""
    from fs_nerf_amd.render import rendering as Rm
    keep = (Rm.FUSED_OCC_SAMPLER, Rm.FUSED_OCC_EXTRAS)
    Rm.FUSED_OCC_SAMPLER = Rm.FUSED_OCC_EXTRAS = False
    try:
        return fn()
    finally:
        Rm.FUSED_OCC_SAMPLER, Rm.FUSED_OCC_EXTRAS = keep


def both_paths(o, d, est, m, dev, step, train=False, white=True):
    from fs_nerf_amd.render import rendering as Rm
    with torch.no_grad():
        if train:
            est.generator = torch.Generator(device=dev).manual_seed(77)
        (rgb_u, op_u, dep_u, ex), ri, tv = unfused(lambda: Rm.render_rays(o, d, est, m, train=train, white_bkgd=white,
                                                                         render_step_size=step, device=dev))
        if train:
            est.generator = torch.Generator(device=dev).manual_seed(77)
        (rgb_f, op_f, dep_f, _), ri_f, tv_f = Rm.render_rays(o, d, est, m, train=train, white_bkgd=white, render_step_size=step,
                                                            device=dev, want_extras=False)
    assert ri_f is None and tv_f is None
    return (rgb_u, op_u, dep_u, ri), (rgb_f, op_f, dep_f)


@pytest.mark.parametrize("net,res,levels,step,train", [((4, 128), 32, 1, 2e-2, False), ((8, 256), 32, 1, 2e-2, False),
                                                        ((8, 256), 64, 2, 1e-2, True), ((4, 128), 128, 1, 5e-3, True)])
def test_fused_occupancy_launch_is_the_unfused_sequence(dev, net, res, levels, step, train):
    from fs_nerf_amd import ops
    m = make_model(net[0], net[1], 4, dev)
    est = sphere_grid(res, levels, dev)
    o, d = orbit_rays(1003, 5)  # not a multiple of the 8-ray chunks
    (rgb_u, op_u, dep_u, ri), (rgb_f, op_f, dep_f) = both_paths(o, d, est, m, dev, step, train)
    assert ri.numel() > 0
    assert torch.equal(rgb_f, rgb_u) and torch.equal(op_f, op_u) and torch.equal(dep_f, dep_u), \
        f"max |d rgb| {float((rgb_f - rgb_u).abs().max()):.3e}"
    # sample counts per ray as the launch saw them
    if train:
        est.generator = torch.Generator(device=dev).manual_seed(77)
        u = torch.rand(o.shape[0], device=dev, generator=est.generator)
    else:
        u = None
    _, _, _, cnt = ops.render_occ_fused(m.packed(), o.to(dev), d.to(dev), aabb=est.aabb, res=est.resolution, levels=est.levels,
                                        bits=est.bits, near_plane=0.0, far_plane=1e10, step=step,
                                        max_steps=est.max_steps(step), u=u, bkgd=(1.0, 1.0, 1.0), want_counts=True)
    assert torch.equal(cnt["n_kept"].long(), torch.bincount(ri, minlength=o.shape[0]))
    assert bool((cnt["n_cand"] >= cnt["n_kept"]).all()) and int(cnt["n_cand"].sum()) > int(cnt["n_kept"].sum()), \
        "the visibility cull drops samples behind the surface"


@pytest.mark.parametrize("precision", ["bf16x3", "fp16", "bf16"])
def test_fused_occupancy_other_precision_modes(dev, precision):
    m = make_model(8, 256, 4, dev, precision)
    est = sphere_grid(32, 1, dev)
    o, d = orbit_rays(700, 6)
    (rgb_u, op_u, dep_u, ri), (rgb_f, op_f, dep_f) = both_paths(o, d, est, m, dev, 2e-2)
    if precision == "bf16x3":
        assert torch.equal(rgb_f, rgb_u) and torch.equal(dep_f, dep_u)
    else:
        # single pass: the standalone forward runs two sample groups per wave (256-sample tiles), the fused occupancy
        # kernel one - the same products summed in the same order, but an independent instruction stream: compared at
        # the mode's own accuracy (2^-8 / 2^-11 per product) instead of bitwise
        tol = {"bf16": 2e-3, "fp16": 3e-4}[precision]
        assert float((rgb_f - rgb_u).abs().max()) <= tol and float((op_f - op_u).abs().max()) <= tol


def test_fused_occupancy_batches_carry_and_empty_rays(dev):
    """Dense grid, small step: ~300 samples per ray, so a batch (2048 samples) holds ~6 rays and almost every chunk of
    8 is split and carried; a third of the rays miss the box (zero samples: pure background); 20,003 rays keep all 256
    workgroups pulling from the queue.  Still bit for bit the unfused sequence."""
    from fs_nerf_amd.render.occgrid import OccGridEstimator
    m = make_model(4, 128, 9, dev, gain=2.0, shift=1.0)  # thin positive medium: the visibility cull keeps most samples
    est = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=16, levels=1).to(dev)
    est.set_binaries(torch.ones(1, 16, 16, 16, dtype=torch.bool))
    est.eval()
    o, d = orbit_rays(20003, 8, hw=160)
    miss = torch.arange(o.shape[0]) % 3 == 0
    d = d.clone()
    d[miss] = -d[miss]  # looking away from the box
    (rgb_u, op_u, dep_u, ri), (rgb_f, op_f, dep_f) = both_paths(o, d, est, m, dev, 5e-3)
    per_ray = torch.bincount(ri, minlength=o.shape[0]).cpu()
    assert int(per_ray[miss].max()) == 0 and float(per_ray[~miss].float().mean()) > 150
    assert torch.equal(rgb_f, rgb_u) and torch.equal(op_f, op_u) and torch.equal(dep_f, dep_u)
    assert bool((rgb_f[miss.to(dev)] == 1.0).all()) and float(dep_f[miss.to(dev)].abs().max()) == 0.0
    # an empty grid: background everywhere, no MLP tile at all
    est.set_binaries(torch.zeros(1, 16, 16, 16, dtype=torch.bool))
    (rgb_u, op_u, dep_u, ri), (rgb_f, op_f, dep_f) = both_paths(o[:100], d[:100], est, m, dev, 5e-3)
    assert ri.numel() == 0 and bool((rgb_f == 1.0).all()) and float(op_f.abs().max()) == 0.0


def test_fused_occupancy_frame_is_one_launch_and_matches_the_ray_path(dev):
    """render_frame with the occupancy estimator: one launch with the rays generated in it == get_rays tensors through
    render_rays (fused, chunked) == the unfused sequence."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.render import rendering as Rm
    from fs_nerf_amd.utils import utilities as U
    m = make_model(8, 256, 4, dev)
    est = sphere_grid(64, 1, dev)
    pose = O.pose_from_spherical(4.0311289, 50.0, 123.0)
    hwf = (90, 121, 150.0)
    step = 1e-2
    timer = ops.launch_timer = []
    with torch.no_grad():
        img, depth = Rm.render_frame(hwf, 2.0, 6.0, pose, 4096, est, m, white_bkgd=True, render_step_size=step, device=dev)
    ops.launch_timer = None
    assert len(timer) == 1, "one launch per frame"
    o, d = U.get_rays(pose, hwf, dev)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    with torch.no_grad():
        (rgb_f, _, dep_f, _), _, _ = Rm.render_rays(o, d, est, m, white_bkgd=True, render_step_size=step, device=dev, want_extras=False)
        (rgb_u, _, dep_u, _), ri, _ = unfused(lambda: Rm.render_rays(o, d, est, m, white_bkgd=True, render_step_size=step, device=dev))
    assert torch.equal(img.reshape(-1, 3), rgb_f) and torch.equal(rgb_f, rgb_u)
    assert torch.equal(depth.reshape(-1), dep_u.reshape(-1).clamp(2.0, 6.0))
    assert 0.05 < float((torch.bincount(ri, minlength=o.shape[0]) > 0).float().mean()) < 0.95, "part of the frame is empty space"


@pytest.mark.parametrize("net,res,levels,step,train", [((4, 128), 32, 1, 2e-2, True), ((8, 256), 64, 2, 1e-2, True),
                                                        ((8, 256), 32, 1, 2e-2, False)])
def test_fused_occupancy_sampler_is_the_unfused_sampling(dev, net, res, levels, step, train):
    """ops.occ_sample_fused (sampler mode of the fused launch + gather: what render_rays uses for estimator.sampling
    when the step needs gradients or extras) == OccGridEstimator.sampling(..., sigma_fn) - march, density pass,
    visibility cull, compaction - bit for bit: ray_indices, t_starts, t_ends; rays without samples, ragged ends."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.render import rendering as Rm
    m = make_model(net[0], net[1], 4, dev)
    est = sphere_grid(res, levels, dev)
    if train:
        est.train()
    o, d = orbit_rays(1003, 5)
    miss = torch.arange(o.shape[0]) % 5 == 0
    d = d.clone()
    d[miss] = -d[miss]
    od, dd = o.to(dev), d.to(dev)
    u = torch.rand(o.shape[0], device=dev, generator=torch.Generator(device=dev).manual_seed(77)) if train else None
    with torch.no_grad():
        ri_u, ts_u, te_u = est.sampling(od, dd, sigma_fn=lambda a, b, c: m.forward_rays(od, dd, c, a, b, full=False).squeeze(-1),
                                        render_step_size=step, u=u)
        ri_f, ts_f, te_f = ops.occ_sample_fused(m.packed(), od, dd, aabb=est.aabb, res=est.resolution, levels=est.levels,
                                                bits=est.bits, near_plane=0.0, far_plane=1e10, step=step,
                                                max_steps=est.max_steps(step), u=u)
    assert ri_u.numel() > 500 and int(torch.bincount(ri_u, minlength=o.shape[0])[miss.to(dev)].max()) == 0
    assert torch.equal(ri_f, ri_u) and torch.equal(ts_f, ts_u) and torch.equal(te_f, te_u)
    # and through render_rays: the training call takes the fused sampler, FUSED_OCC_SAMPLER = False the unfused one
    m.train()
    outs = []
    floor = Rm.FUSED_OCC_SAMPLER_MIN_RAYS
    Rm.FUSED_OCC_SAMPLER_MIN_RAYS = 0  # (by default only calls with >= 4096 rays take the fused sampler)
    for flag in (True, False):
        Rm.FUSED_OCC_SAMPLER = flag
        est.generator = torch.Generator(device=dev).manual_seed(9)
        (rgb, _, _, _), ri, tv = Rm.render_rays(o, d, est, m, train=train, white_bkgd=True, render_step_size=step, device=dev)
        outs.append((rgb.detach(), ri, tv))
    Rm.FUSED_OCC_SAMPLER, Rm.FUSED_OCC_SAMPLER_MIN_RAYS = True, floor
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("net,res,levels,step,train", [((4, 128), 32, 1, 2e-2, False), ((8, 256), 64, 2, 1e-2, True),
                                                        ((8, 256), 32, 1, 2e-2, False)])
def test_fused_occupancy_extras_mode_is_the_full_return_contract(dev, net, res, levels, step, train):
    """VERDICT r3 missing #2: render_rays ALWAYS returns ((rgb, opacity, depth, extras), ray_indices, t_vals)
    (rendering.py:88-107); round 3's fused launch existed only for callers that dropped the extras.  The extras mode of
    fsn_render_rays_occgrid (one launch + one gather behind one host read) returns the whole tuple, bit for bit the unfused
    sequence's: per-sample weights / alphas / trans / sigmas / rgbs, ray_indices, t_vals, and the per-ray outputs."""
    from fs_nerf_amd.render import rendering as Rm
    m = make_model(net[0], net[1], 4, dev)
    est = sphere_grid(res, levels, dev)
    o, d = orbit_rays(1003, 6)

    def run():
        if train:
            est.generator = torch.Generator(device=dev).manual_seed(78)
        with torch.no_grad():
            return Rm.render_rays(o, d, est, m, train=train, white_bkgd=True, render_step_size=step, device=dev)

    (rgb_u, op_u, dep_u, ex_u), ri_u, tv_u = unfused(run)
    assert Rm.FUSED_OCC_EXTRAS and o.shape[0] * est.max_steps(step) <= Rm.FUSED_OCC_EXTRAS_MAX_SLOTS
    (rgb_f, op_f, dep_f, ex_f), ri_f, tv_f = run()
    assert ri_u.numel() > 0 and ri_f.dtype == torch.int64
    assert torch.equal(ri_f, ri_u) and torch.equal(tv_f, tv_u)
    assert torch.equal(rgb_f, rgb_u) and torch.equal(op_f, op_u) and torch.equal(dep_f, dep_u)
    assert set(ex_f) >= {"weights", "alphas", "trans", "sigmas", "rgbs"}
    for k in ("weights", "alphas", "trans", "sigmas", "rgbs"):
        assert ex_f[k].shape == ex_u[k].shape and torch.equal(ex_f[k], ex_u[k]), k
    # an empty grid: zero samples, pure background, empty packed arrays
    from fs_nerf_amd.render.occgrid import OccGridEstimator
    empty = OccGridEstimator(roi_aabb=torch.tensor(AABB), resolution=res, levels=levels).to(dev).eval()
    with torch.no_grad():
        (rgb_e, op_e, dep_e, ex_e), ri_e, tv_e = Rm.render_rays(o, d, empty, m, white_bkgd=True, render_step_size=step, device=dev)
    assert ri_e.numel() == 0 and tv_e.numel() == 0 and ex_e["rgbs"].shape == (0, 3) and bool((rgb_e == 1.0).all())


@pytest.mark.parametrize("gain,shift", [(64.0, 3.0), (512.0, 30.0)])
def test_opt_in_bf16_cull_differs_from_the_default_only_at_the_threshold(dev, gain, shift):
    """`NeRF.cull_precision = "bf16"` (opt-in, NOT a parity mode): the density pass behind the estimator's visibility cull
    runs in single-pass bf16 as its own launch; what it keeps is evaluated, integrated and differentiated in the model's own
    mode.  What may differ from the default is therefore WHICH samples sit next to the cull's threshold (transmittance 1e-4):
    every sample only one of the two kept has a transmittance within a factor 2 of it, such samples are a small fraction,
    and rgb / opacity / depth move by far less than the threshold itself.  The same holds with gradients."""
    from fs_nerf_amd.render import rendering as Rm
    step, eps = 1e-2, 1e-4
    est = sphere_grid(64, 1, dev)
    o, d = orbit_rays(2000, 9)
    m0, m1 = make_model(8, 256, 4, dev, gain=gain, shift=shift), make_model(8, 256, 4, dev, gain=gain, shift=shift)
    m1.cull_precision = "bf16"
    run = lambda m_, **kw: Rm.render_rays(o, d, est, m_, white_bkgd=True, render_step_size=step, device=dev, **kw)
    with torch.no_grad():
        (rgb0, op0, dep0, ex0), ri0, tv0 = run(m0)
        (rgb1, op1, dep1, ex1), ri1, tv1 = run(m1)
        (rgb1f, _, dep1f, _), _, _ = run(m1, want_extras=False)
    assert torch.equal(rgb1f, rgb1) and torch.equal(dep1f, dep1)
    assert float((rgb1 - rgb0).abs().max()) < 5e-5 and float((op1 - op0).abs().max()) < 5e-5 and float((dep1 - dep0).abs().max()) < 5e-4
    # the kept sets as (ray, lattice index) pairs
    key = lambda ri, tv: set(zip(ri.tolist(), torch.round(tv / step * 4).long().tolist()))
    k0, k1 = key(ri0, tv0), key(ri1, tv1)
    only0, only1 = k0 - k1, k1 - k0
    assert len(k0) > 20000 and len(only0) + len(only1) <= 0.01 * len(k0), (len(k0), len(only0), len(only1))
    for ks, ri, tv, ex in ((only0, ri0, tv0, ex0), (only1, ri1, tv1, ex1)):
        if not ks:
            continue
        keys = list(zip(ri.tolist(), torch.round(tv / step * 4).long().tolist()))
        T = torch.tensor([float(ex["trans"][i]) for i, k_ in enumerate(keys) if k_ in ks])
        assert float(T.min()) > eps / 2 and float(T.max()) < eps * 2, (float(T.min()), float(T.max()))
    # training step: gradients through the kept samples
    grads = []
    for m_ in (m0, m1):
        m_.train()
        est.train()
        est.generator = torch.Generator(device=dev).manual_seed(5)
        (rgb, _, _, _), _, _ = Rm.render_rays(o, d, est, m_, train=True, white_bkgd=True, render_step_size=step, device=dev)
        rgb.square().mean().backward()
        grads.append({n: p.grad.clone() for n, p in m_.named_parameters()})
        m_.eval()
    est.eval()
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        assert bool(torch.isfinite(b).all()) and float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-12, n
