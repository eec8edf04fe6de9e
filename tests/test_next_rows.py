"""SURVEY.md 8f "next" rows built so far: occlusion regulariser (f4), lr schedules (f1), to8b / render_path
(f3).  Expected values come from the reference itself (tests/golden/g5_next.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import fs_nerf_amd  # noqa: F401
from oracle import fsnerf_oracle as O


def test_scheduler_matches_reference_golden(golden_dir):
    from fs_nerf_amd.core.scheduler import Constant, ExponentialDecay
    g = np.load(os.path.join(golden_dir, "g5_next.npz"))
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=5e-4)
    sch = ExponentialDecay(opt, 8000, 5e-4, r=0.1)
    for t, want in zip(g["sched_t"], g["sched_lr"]):
        sch.t = int(t)
        assert sch.lr == pytest.approx(float(want), rel=1e-12)
    sch.t = 0
    sch.step()
    assert opt.param_groups[0]["lr"] == sch.lr and sch.t == 1
    assert Constant(opt, 10, 3e-4).lr == 3e-4
    with pytest.raises(ValueError):
        Constant(opt, 10, -1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("func", ["linear", "exp"])
def test_occlusion_regularizer_golden_and_large(golden_dir, func):
    from fs_nerf_amd.core.loss import OcclusionRegularizer
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "g5_next.npz"))
    reg = OcclusionRegularizer(0.5, 3.0, func)
    got = reg(torch.from_numpy(g["occ_sigmas"]).to(dev), torch.from_numpy(g["occ_t"]).to(dev),
              torch.from_numpy(g["occ_ray_idx"]).to(dev))
    assert got.shape == ()
    np.testing.assert_allclose(float(got), float(g["occ_" + func]), rtol=1e-5)
    # a packed batch of the fused renderer: 4096 rays x 192 samples, against the formula in float64
    gen = torch.Generator().manual_seed(0)
    R, S = 4096, 192
    ri = torch.arange(R).repeat_interleave(S)
    sig, t = torch.rand(R * S, generator=gen) * 5, torch.rand(R * S, generator=gen) * 4 + 2
    w = (-0.5 * t.double() + 3.0) if func == "linear" else 0.5 * torch.exp(-3.0 * t.double())
    want = (w * sig.double()).reshape(R, S).sum(-1).mean()
    got = reg(sig.to(dev), t.to(dev), ri.to(dev))
    np.testing.assert_allclose(float(got), float(want), rtol=1e-5)
    with pytest.raises(ValueError):
        OcclusionRegularizer(0.5, 3.0, "cubic")(sig.to(dev), t.to(dev), ri.to(dev))


@pytest.mark.gpu
def test_to8b_and_render_path(golden_dir):
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.render import rendering as Rm
    dev = torch.device("cuda:0")
    x = torch.tensor([-0.5, 0.0, 0.2, 0.999, 1.0, 1.7, 127.5 / 255, 128 / 255], device=dev)
    want = (255 * np.clip(x.cpu().numpy(), 0, 1)).astype(np.uint8)
    assert Rm.to8b(x).cpu().numpy().tolist() == want.tolist()
    assert Rm.to8b(x.cpu().numpy()).tolist() == want.tolist()
    torch.manual_seed(1)
    m = NeRF(3, 3, 4, 128, (4,), pos_fn={"n_freqs": 10, "log_space": True},
             dir_fn={"n_freqs": 4, "log_space": True}).to(dev).eval()
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 128)
    poses = torch.stack([O.pose_from_spherical(4.0311289, 50.0, phi) for phi in (0.0, 120.0)])
    hwf = (20, 24, 30.0)
    frames, d_frames = Rm.render_path(poses, hwf, 2.0, 6.0, 200, m, est, white_bkgd=True, device=dev)
    assert frames.shape == (2, 20, 24, 3) and d_frames.shape == (2, 20, 24) and frames.dtype == np.float32
    img, dep = Rm.render_frame(hwf, 2.0, 6.0, poses[1], 480, est, m, white_bkgd=True, device=dev)
    np.testing.assert_array_equal(frames[1], img.cpu().numpy())
    np.testing.assert_array_equal(d_frames[1], dep.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("ndc", [False, True])
def test_build_rays_matches_the_dataset_recipe(ndc):
    """row f4: all-pose ray precompute (+NDC, +aabb) against the reference's recipe evaluated with the oracle
    (blender.py:174-191 / llff.py:59-90: stack of get_rays per pose, to_ndc(., 1.0), min/max of {o, o+d} / 8)."""
    from fs_nerf_amd.utils import utilities as U
    dev = torch.device("cuda:0")
    hwf = (12, 16, 20.0)
    poses = [O.pose_from_spherical(4.0, 40.0 + 5 * i, 30.0 * i) for i in range(5)]
    ro, rd, aabb = U.build_rays(poses, hwf, dev, ndc=ndc)
    wo = torch.cat([O.get_rays(p, hwf)[0].reshape(-1, 3) for p in poses])
    wd = torch.cat([O.get_rays(p, hwf)[1].reshape(-1, 3) for p in poses])
    if ndc:
        wo, wd = O.to_ndc(wo, wd, hwf, 1.0)
        ends = wo + wd
        want = torch.cat([torch.minimum(wo.amin(0), ends.amin(0)), torch.maximum(wo.amax(0), ends.amax(0))]) / 8
    else:
        want = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    assert ro.shape == (5 * 12 * 16, 3)
    assert torch.allclose(ro.cpu(), wo, rtol=1e-5, atol=1e-5) and torch.allclose(rd.cpu(), wd, rtol=1e-5, atol=1e-5)
    assert torch.allclose(aabb.cpu(), want, rtol=1e-4, atol=1e-5)
    # ONE launch (round 4) = the per-pose sequence it replaces, bit for bit: get_rays per pose, to_ndc over all of them,
    # torch's min / max reductions (min / max are exact in any order); also at a size that is not a multiple of the
    # block, where the last wave's dead lanes must not disturb the reduction
    for hwf2 in (hwf, (7, 9, 11.0)):
        ro2, rd2, aabb2 = U.build_rays(poses, hwf2, dev, ndc=ndc)
        so = torch.cat([U.get_rays(p, hwf2, dev)[0].reshape(-1, 3) for p in poses])
        sd_ = torch.cat([U.get_rays(p, hwf2, dev)[1].reshape(-1, 3) for p in poses])
        if ndc:
            so, sd_ = U.to_ndc(so, sd_, hwf2, 1.0)
            e2 = so + sd_
            a_want = torch.cat([torch.minimum(so.amin(0), e2.amin(0)), torch.maximum(so.amax(0), e2.amax(0))]) / 8
            assert torch.equal(aabb2, a_want)
        assert torch.equal(ro2, so) and torch.equal(rd2, sd_)
    assert U.build_rays([], hwf, dev, ndc=ndc)[0].shape == (0, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("func", ["linear", "exp"])
def test_occlusion_regularizer_gradient(func):
    """loss += occ_reg(sigmas, t_vals, ray_indices) (run-nerf.py:264): gradient w.r.t. sigmas vs autograd on the
    reference formula (loss.py:39-42), rays without samples excluded from the mean."""
    from fs_nerf_amd.core.loss import OcclusionRegularizer
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    counts = torch.tensor([5, 0, 17, 1, 0, 64, 130])
    ri = torch.repeat_interleave(torch.arange(len(counts)), counts)
    N = ri.numel()
    sig = (torch.rand(N, generator=gen) * 3 - 0.5)
    t = torch.rand(N, generator=gen) * 4 + 2
    s_gpu = sig.to(dev).requires_grad_(True)
    reg = OcclusionRegularizer(0.3, 1.7, func)
    out = reg(s_gpu, t.to(dev), ri.to(dev))
    (out * 2.5).backward()
    s64 = sig.double().requires_grad_(True)
    w = (-0.3 * t.double() + 1.7) if func == "linear" else 0.3 * torch.exp(-1.7 * t.double())
    want = torch.stack([(w[ri == v] * s64[ri == v]).sum() for v in torch.unique_consecutive(ri)]).mean()
    (want * 2.5).backward()
    assert abs(float(out) - float(want)) < 1e-5 * max(1.0, abs(float(want)))
    assert torch.allclose(s_gpu.grad.cpu().double(), s64.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("cmap", ["plasma", "viridis"])
def test_render_video_matches_matplotlib_golden(golden_dir, cmap):
    """f3: depth colormap + to8b + NHWC->NCHW on the device against the reference's own lines evaluated with
    matplotlib (tests/golden/make_golden.py::make_g6_video), byte for byte."""
    from fs_nerf_amd.render import rendering as Rm
    g = np.load(os.path.join(golden_dir, "g6_video.npz"))
    f8, d8 = Rm.render_video(g["frames"], g["d_frames"], cmap)  # numpy in (what render_path returns) -> numpy out
    assert f8.dtype == np.uint8 and f8.shape == (2, 3, 5, 7) and d8.shape == (2, 3, 5, 7)
    assert np.array_equal(f8, g[f"{cmap}_frames8"])
    assert np.array_equal(d8, g[f"{cmap}_depth8"])
    dev = torch.device("cuda:0")
    tf, td = Rm.render_video(torch.from_numpy(g["frames"]).to(dev), torch.from_numpy(g["d_frames"]).to(dev), cmap)
    assert tf.is_cuda and np.array_equal(td.cpu().numpy(), g[f"{cmap}_depth8"])
    if cmap == "plasma":  # constant depth: vmin == vmax
        _, fl = Rm.render_video(np.zeros((1, 3, 4, 3), np.float32), g["flat_d"], "plasma")
        assert np.array_equal(fl, g["flat_depth8"])
    with pytest.raises(ValueError):
        Rm.render_video(g["frames"], g["d_frames"], "no-such-map")


def test_colormap_tables_are_matplotlibs():
    """The shipped uint8 tables equal to8b of matplotlib's colormaps (when matplotlib is importable)."""
    mpl = pytest.importorskip("matplotlib")
    from fs_nerf_amd.render.cmaps import TABLES
    for name, raw in TABLES.items():
        cm = mpl.colormaps[name]
        lut = cm(np.arange(256))[:, :3]
        want = (255 * np.clip(lut, 0, 1)).astype(np.uint8)
        assert np.array_equal(np.frombuffer(raw, np.uint8).reshape(256, 3), want), name
