"""Training step (first version): gradients of NeRF.forward and of the volume integration
against torch autograd on the float64 oracle; one optimisation step through render_rays(train=True); the
data-parallel gradient all-reduce over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fs_nerf_amd  # noqa: F401
from fs_nerf_amd import shard
from oracle import fsnerf_oracle as O


def _run(m, tp, x, d):
    """m(x, d) with gradients in training mode `tp`: a precision mode of the product, None (the model's own), or "fp32" =
    the TEST-ONLY plain-fp32 library-GEMM formulation (tests/ref_fp32/binding.py; not reachable from the package)."""
    if tp == "fp32":
        from ref_fp32 import binding
        return binding.forward(m, x, d)
    m.train_precision = tp
    return m(x, d)


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def _relu_margin(sd, x, d, n_layers, skip, nf, nfd):
    """min |pre-activation| over every ReLU unit of the network, per sample (float64, CPU)."""
    sd = {k: v.double() for k, v in sd.items()}
    pe = O.posenc(x.double(), nf, True)
    h, margin = pe, torch.full((x.shape[0],), 1e9, dtype=torch.float64)
    for i in range(n_layers):
        z = torch.nn.functional.linear(h, sd[f"layers.{i}.weight"], sd[f"layers.{i}.bias"])
        margin = torch.minimum(margin, z.abs().amin(dim=1))
        h = torch.relu(z)
        if i in skip:
            h = torch.cat([h, pe], dim=-1)
    f = torch.nn.functional.linear(h, sd["connection.weight"], sd["connection.bias"])
    zb = torch.nn.functional.linear(torch.cat([f, O.posenc(d.double(), nfd, True)], dim=-1),
                                    sd["branch.weight"], sd["branch.bias"])
    return torch.minimum(margin, zb.abs().amin(dim=1))


# train precision -> (gradient tolerance relative to the tensor's largest entry, ReLU margin of the test samples).
# "fp32" = plain library GEMMs; None = the model's mode (fp16x3 MFMA, fp32-grade); bf16x3 carries ~2^-16 per product.
# A ReLU unit whose pre-activation is within the forward's error of zero takes the other branch, which changes that
# sample's whole gradient (torch's own fp32 autograd differs from its fp64 autograd by 1e-3 on layers.0/1 of the
# 8x256 case below for exactly this reason), so the test samples are drawn with every unit at least `margin` away.
TRAIN_MODES = [("fp32", 2e-4, 2e-5), (None, 2e-4, 2e-5), ("bf16x3", 1e-3, 1e-4)]


@pytest.mark.gpu
@pytest.mark.parametrize("train_precision,tol,margin", TRAIN_MODES)
@pytest.mark.parametrize("n_layers,d_hidden,skip,nf,nfd,cscale", [(4, 128, [], 10, 4, 1.0), (8, 256, [4], 10, 4, 1e-7),
                                                                  (6, 128, [1, 3], 7, 3, 1e3)])
def test_nerf_gradients_vs_autograd(n_layers, d_hidden, skip, nf, nfd, cscale, train_precision, tol, margin):
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    sd = O.init_nerf_state_dict(n_layers, d_hidden, skip, nf, nfd, seed=3)
    sd["sigma.weight"] *= 16.0
    m = NeRF(3, 3, n_layers, d_hidden, tuple(skip), pos_fn={"n_freqs": nf, "log_space": True},
             dir_fn={"n_freqs": nfd, "log_space": True})
    m.load_state_dict(sd)
    m = m.to(dev).train()
    gen = torch.Generator().manual_seed(0)
    N = 777
    x = torch.rand(12 * N, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(12 * N, 3, generator=gen), dim=-1)
    keep = _relu_margin(sd, x, d, n_layers, skip, nf, nfd) > margin
    x, d = x[keep][:N].contiguous(), d[keep][:N].contiguous()
    assert x.shape[0] == N
    c = torch.randn(N, 4, generator=gen) * cscale  # d(out): 1e-7 .. 1e3 exercises the fp16 gradient scaling
    out = _run(m, train_precision, x.to(dev), d.to(dev))
    assert out.requires_grad and out.shape == (N, 4)
    (out * c.to(dev)).sum().backward()
    sdr = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.nerf_forward(sdr, x.double(), d.double(), n_layers=n_layers, skip=skip, n_freqs=nf, n_freqs_dir=nfd)
    assert _rel(out, ref) < (1e-4 if train_precision == "bf16x3" else 1e-5)
    (ref * c.double()).sum().backward()
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        err = _rel(p.grad, sdr[name].grad)
        assert err < tol, (name, err)
    with pytest.raises(NotImplementedError):
        m(x.to(dev))  # density-only pass has no backward: the reference runs it under no_grad


@pytest.mark.gpu
@pytest.mark.parametrize("tp", ["bf16", "fp16"])
def test_single_pass_training_modes_run(tp):
    """The single-pass 16-bit modes of the training kernels (BASELINE config 5 style: bf16 weights/activations): not a
    parity mode (2^-8 / 2^-11 per product, ReLU units change branch), so the check is direction and scale of the
    whole gradient against the fp32 path."""
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    m = NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m = m.to(dev).train()
    N = 3000
    x = torch.rand(N, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    c = torch.randn(N, 4, device=dev)
    flat = {}
    for mode in ("fp32", tp):
        m.zero_grad(set_to_none=True)
        (_run(m, mode, x, d) * c).sum().backward()
        flat[mode] = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).double()
    assert bool(torch.isfinite(flat[tp]).all())
    cos = float((flat[tp] * flat["fp32"]).sum() / (flat[tp].norm() * flat["fp32"].norm()))
    assert cos > 0.995 and abs(float(flat[tp].norm() / flat["fp32"].norm()) - 1.0) < 0.05, cos


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 129])
def test_nerf_gradients_tiny_batches(n):
    """Fewer samples than one 128-sample tile / one more than a tile: padded columns must contribute nothing."""
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    m = NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m = m.to(dev).train()
    x = torch.rand(n, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
    c = torch.randn(n, 4, device=dev)
    grads = {}
    for tp in ("fp32", None):
        m.zero_grad(set_to_none=True)
        (_run(m, tp, x, d) * c).sum().backward()
        grads[tp] = {k: p.grad.clone() for k, p in m.named_parameters()}
    for k in grads[None]:
        assert bool(torch.isfinite(grads[None][k]).all())
        assert _rel(grads[None][k], grads["fp32"][k]) < 2e-3, (k, _rel(grads[None][k], grads["fp32"][k]))


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [False, True])
def test_nerf_gradients_mfma_vs_plain_many_tiles(mask):
    """40,001 samples = 313 tiles (more than the wgrad's split count, last tile ragged): the MFMA path against the
    plain fp32-GEMM path on the same device, small d_out (1e-6 scale: exercises the fp16 gradient scaling)."""
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    m = NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m = m.to(dev).train()
    if mask:
        pm = torch.cat([torch.ones(33), torch.full((12,), 0.5), torch.zeros(18)])
        dm = torch.cat([torch.ones(15), torch.zeros(12)])
        m.set_freq_mask(pm, dm)
    N = 40001
    # Keep only samples whose hidden pre-activations all stay 2e-5 away from zero: the two paths' forwards differ
    # by ~3e-6, and a ReLU unit that changes branch changes the sample's whole gradient (see the oracle test), which
    # would mask what this test is about (tiling, split-K ranges, gradient scaling).
    with torch.no_grad():
        cand = torch.rand(2 * N, 3, device=dev) * 2 - 1
        pe = m.pos_encoder(cand, m._mask(m.pos_mask, dev))
        h, margin = pe, torch.full((2 * N,), 1e9, device=dev)
        for i, lin in enumerate(m.layers):
            z = torch.nn.functional.linear(h, lin.weight, lin.bias)
            margin = torch.minimum(margin, z.abs().amin(dim=1))
            h = torch.relu(z)
            if i in m.skip:
                h = torch.cat([h, pe], dim=-1)
        dc = torch.nn.functional.normalize(torch.randn(2 * N, 3, device=dev), dim=-1)
        zb = m.branch(torch.cat([m.connection(h), m.dir_encoder(dc, m._mask(m.dir_mask, dev))], dim=-1))
        margin = torch.minimum(margin, zb.abs().amin(dim=1))
        keep = margin > 2e-5
        x, d = cand[keep][:N].contiguous(), dc[keep][:N].contiguous()
    assert x.shape[0] == N
    c = torch.randn(N, 4, device=dev) * 1e-6
    grads = {}
    for tp in ("fp32", None):
        m.zero_grad(set_to_none=True)
        (_run(m, tp, x, d) * c).sum().backward()
        grads[tp] = {k: p.grad.clone() for k, p in m.named_parameters()}
    for k in grads[None]:
        assert _rel(grads[None][k], grads["fp32"][k]) < 1e-4, (k, _rel(grads[None][k], grads["fp32"][k]))


@pytest.mark.gpu
@pytest.mark.parametrize("S", [64, 192, 5])
def test_composite_gradients_vs_autograd(S):
    from fs_nerf_amd.render import rendering as Rm
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(S)
    R = 70
    edges = O.stratified_edges(2.0, 6.0, S, R, torch.rand(R, generator=gen))
    ri, t0, t1 = O.edges_to_packed(edges)
    sig = (torch.rand(R * S, generator=gen) * 2.0 / (4.0 / S) * 0.3)
    rgb = torch.rand(R * S, 3, generator=gen)
    A, bvec = torch.randn(R, 3, generator=gen), torch.randn(R, 1, generator=gen)
    bk = torch.tensor([1.0, 0.5, 0.25])
    sg, rg = sig.to(dev).requires_grad_(True), rgb.to(dev).requires_grad_(True)
    colors, opacity, depth, ex = Rm.rendering(t0.to(dev), t1.to(dev), ri.to(dev), R, lambda a, b, c: (rg, sg), bk.to(dev))
    ((colors * A.to(dev)).sum() + (opacity * bvec.to(dev)).sum()).backward()
    s64, r64 = sig.double().requires_grad_(True), rgb.double().requires_grad_(True)
    wc, wo, wd, _ = O.rendering_packed(t0.double(), t1.double(), ri, R, lambda a, b, c: (r64, s64), bk.double())
    ((wc * A.double()).sum() + (wo * bvec.double()).sum()).backward()
    assert _rel(colors, wc) < 1e-5
    assert _rel(sg.grad, s64.grad) < 2e-4 and _rel(rg.grad, r64.grad) < 2e-4
    assert not depth.requires_grad and not ex["weights"].requires_grad


@pytest.mark.gpu
def test_render_rays_train_step_matches_oracle():
    """One optimisation step through the reference's call structure (run-nerf.py:243-285): render_rays(train=True)
    -> mse -> backward -> Adam; loss and every gradient against autograd on the float64 oracle."""
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.render import rendering as Rm
    dev = torch.device("cuda:0")
    sd = O.init_nerf_state_dict(4, 128, [], 10, 4, seed=11)
    sd["sigma.weight"] *= 64.0
    sd["sigma.bias"] += 3.0
    m = NeRF(3, 3, 4, 128, (), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    m = m.to(dev).train()
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 128).train()
    o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, 20.0), (16, 16, 20.0))
    o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
    gen = torch.Generator().manual_seed(5)
    R = o.shape[0]
    u, uf = torch.rand(R, generator=gen), torch.rand(R, 128, generator=gen)
    gt = torch.rand(R, 3, generator=gen)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    (rgb, opacity, depth, ex), ri, tv = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev,
                                                       u=u.to(dev), u_fine=uf.to(dev))
    assert rgb.requires_grad and ex["sigmas"].shape == (R * 192,)
    loss = torch.nn.functional.mse_loss(rgb, gt.to(dev))
    loss.backward()
    # oracle on the sample set the GPU path produced (inverse-CDF resampling is ill-conditioned, see test_gpu_parity)
    S = 192
    # rebuild edges from the packed intervals the sampler returned
    with torch.no_grad():
        rays_i, t_s, t_e = est.sampling(o.to(dev), d.to(dev), sigma_fn=lambda a, b, c: m(o.to(dev)[c] + d.to(dev)[c] * (a + b)[:, None] / 2.0).squeeze(-1),
                                        stratified=True, u=u.to(dev), u_fine=uf.to(dev))
    e = torch.cat([t_s.reshape(R, S), t_e.reshape(R, S)[:, -1:]], dim=1).cpu()
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    cfg = dict(n_layers=4, skip=[], n_freqs=10, n_freqs_dir=4, log_space=True)
    (wrgb, _, _, _), _, _ = O.render_rays_oracle(o.double(), d.double(), sd64, None, cfg, near=2.0, far=6.0, n_samples=64,
                                                 n_importance=128, white_bkgd=True, edges_override=e.double())
    wloss = torch.nn.functional.mse_loss(wrgb, gt.double())
    wloss.backward()
    assert abs(float(loss) - float(wloss)) < 1e-5 * max(1.0, float(wloss))
    for name, p in m.named_parameters():
        assert _rel(p.grad, sd64[name].grad) < 2e-3, (name, _rel(p.grad, sd64[name].grad))
    before = m.layers[0].weight.detach().clone()
    opt.step()
    assert float((m.layers[0].weight - before).abs().max()) > 0
    m.eval()
    with torch.no_grad():  # parameters changed -> the MFMA blob is repacked for the fused inference path
        out = Rm.render_rays(o, d, Rm.StratifiedEstimator(2.0, 6.0, 64, 128), m, white_bkgd=True, device=dev)
    assert bool(torch.isfinite(out[0][0]).all())


@pytest.mark.gpu
def test_ray_form_forward_is_the_reference_closures_bit_for_bit():
    """VERDICT r2 missing #2: the training forward (and the sampler's density pass) read the rays and the packed
    intervals; the gathers rays_o[ray_indices] / rays_d[ray_indices] and the midpoint arithmetic of the reference's
    closures (rendering.py:58-64, 76-84) happen inside the launch, no [N,3] tensor is materialised.  Values and
    gradients are those of the closure form exactly."""
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = NeRF(3, 3, 8, 256, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True}).to(dev)
    R, N = 97, 5003
    gen = torch.Generator(device=dev).manual_seed(1)
    ro = torch.rand(R, 3, device=dev, generator=gen) * 2 - 1
    rd = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=gen), dim=-1)
    ri = torch.sort(torch.randint(0, R, (N,), device=dev, generator=gen)).values
    t0 = torch.rand(N, device=dev, generator=gen) * 4 + 2
    t1 = t0 + 0.02
    to, td = ro[ri], rd[ri]
    x = to + td * (t0 + t1)[:, None] / 2.0
    m.eval()
    with torch.no_grad():
        assert torch.equal(m.forward_rays(ro, rd, ri, t0, t1, full=False), m(x))
        assert torch.equal(m.forward_rays(ro, rd, ri, t0, t1, full=True), m(x, td))
        assert torch.equal(m.forward_rays(ro, rd, ri.to(torch.int32), t0, t1, full=True), m(x, td)), "int32 indices are widened"
    m.train()
    c = torch.randn(N, 4, device=dev, generator=gen)
    out_a = m(x, td)
    (out_a * c).sum().backward()
    ga = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    out_b = m.forward_rays(ro, rd, ri, t0, t1, full=True)
    assert out_b.requires_grad and torch.equal(out_a.detach(), out_b.detach())
    (out_b * c).sum().backward()
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, ga[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("net,rays,S,NI,jitter", [((8, 256), "ndc", 64, 128, True), ((8, 256), "orbit", 64, 128, False),
                                                  ((4, 128), "orbit", 96, 32, True), ((8, 256), "orbit", 128, 256, True)])
def test_fused_sampler_is_the_five_launch_sampler(net, rays, S, NI, jitter):
    """VERDICT r2 item 6 ("fuse sampler + density pass"): in front of the training forward the hierarchical sampler is
    ONE launch (the fused render kernel stopped after its coarse stage, fsn_render_args.two_phase = 2) instead of
    stratified edges -> packed intervals -> density pass -> weights -> resampling -> packed intervals.  Same edges and
    coarse weights bit for bit (same arithmetic: stratified_edge, the MLP tile, weights / sample_pdf_merge per ray)."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.render import rendering as Rm
    from test_parity_fp64 import make_sd, ndc_rays, orbit_rays
    dev = torch.device("cuda:0")
    Lx, Dx = net
    m = NeRF(3, 3, Lx, Dx, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(make_sd(Lx, Dx, 42))
    m = m.to(dev).eval()
    R = 1003
    if rays == "ndc":
        o, d, gen = ndc_rays(R, 7)
        near, far = 0.0, 1.0
    else:
        o, d, gen = orbit_rays(R, 7, 400, 555.5)
        near, far = 2.0, 6.0
    o, d = o.to(dev), d.to(dev)
    u = torch.rand(R, generator=gen).to(dev) if jitter else None
    uf = torch.rand(R, NI, generator=gen).to(dev) if jitter else None
    est = Rm.StratifiedEstimator(near, far, S, NI)
    with torch.no_grad():
        ri, t0, t1 = est.sampling(o, d, sigma_fn=lambda a, b, c: m(o[c] + d[c] * (a + b)[:, None] / 2.0).squeeze(-1), u=u, u_fine=uf)
        edges, wc = ops.sample_fused(m.packed(), o, d, near=near, far=far, n_samples=S, n_importance=NI, u=u, u_fine=uf,
                                     want_weights=True)
    want = torch.cat([t0.reshape(R, S + NI), t1.reshape(R, S + NI)[:, -1:]], dim=1)
    assert torch.equal(edges, want), f"max |d edge| {float((edges - want).abs().max()):.3e}"
    # and through render_rays(train=True): the packed intervals the training forward sees
    m.train()
    est.train()
    (rgb, _, _, ex), ri2, tv = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev, u=u if jitter else torch.zeros(R, device=dev),
                                             u_fine=uf if jitter else torch.linspace(0, 1, NI, device=dev)[None].expand(R, NI).contiguous())
    assert rgb.requires_grad and ri2.numel() == R * (S + NI)
    if jitter:
        assert torch.equal(tv, (t0 + t1) / 2.0)


@pytest.mark.gpu
def test_c4_shape_whole_step_gradients_end_to_end():
    """BASELINE configs[3]'s training shape: ONE 8x256 network, 256 forward-facing NDC rays, 64 + 128 samples,
    render_rays(train=True) -> mse -> backward, every parameter gradient against float64 autograd on the oracle -
    END TO END: the oracle runs its OWN coarse pass and resampling (under no_grad, as the reference's sampler does,
    rendering.py:58-74), nothing of the kernel's is handed to it.  Bar: 2e-4 of the tensor's largest entry, or three
    times what torch's own FLOAT32 autograd on the oracle ("the reference PyTorch CPU path") is off by: with 50,000
    uncurated samples some ReLU pre-activations lie within the forward's rounding of zero and take the other branch,
    which moves the first layers' gradients by ~3e-4 in any float32 evaluation (the curated-sample tests above hold
    2e-4 on every tensor)."""
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.render import rendering as Rm
    from test_parity_fp64 import cfg_of, make_sd, ndc_rays
    dev = torch.device("cuda:0")
    Lx, Dx, R, S, NI = 8, 256, 256, 64, 128
    sd = make_sd(Lx, Dx, 42)
    o, d, gen = ndc_rays(R, 7)
    u, uf = torch.rand(R, generator=gen), torch.rand(R, NI, generator=gen)
    gt = torch.rand(R, 3, generator=gen)
    m = NeRF(3, 3, Lx, Dx, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    m = m.to(dev).train()
    est = Rm.StratifiedEstimator(0.0, 1.0, S, NI).train()
    (rgb, opacity, depth, ex), ri, tv = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev,
                                                       u=u.to(dev), u_fine=uf.to(dev))
    loss = torch.nn.functional.mse_loss(rgb, gt.to(dev))
    loss.backward()
    assert m.precision == "fp16x3"
    kw = dict(near=0.0, far=1.0, n_samples=S, n_importance=NI, u=u.double(), u_fine=uf.double(), white_bkgd=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():  # the sampler's density pass carries no gradient (rendering.py:58-64)
        edges = O.render_rays_oracle(o.double(), d.double(), sd64, None, cfg_of(Lx), **kw)[0][3]["edges"]
    sdg = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    (wrgb, _, _, _), _, wtv = O.render_rays_oracle(o.double(), d.double(), sdg, None, cfg_of(Lx), edges_override=edges, **kw)
    wloss = torch.nn.functional.mse_loss(wrgb, gt.double())
    wloss.backward()
    assert float((tv.cpu().double() - wtv).abs().max()) < 2e-5, "the two paths resampled the same positions"
    assert abs(float(loss) - float(wloss)) < 1e-5 * max(1.0, float(wloss))
    # Yardstick: "the reference PyTorch CPU path" END TO END in float32 - its own coarse pass and resampling, then
    # autograd.  (Round 3 handed the float32 oracle the float64 edges, which left the resampler's amplification of a
    # last-bit difference of the coarse pass out of the yardstick while the kernel's gradients contain it: measured
    # tools/dbg_c4grad.py - float32 end to end is off by 3.7e-4 on layers.0.weight and 1.3e-4 .. 2.4e-4 on layers 1, 5
    # and 6, from ONE importance sample displaced by 8.5e-7; which tensor such a sample's ReLU flips land in differs
    # from one float32 evaluation to the next.)  Bar per tensor: 2e-4, or 3 x the float32 oracle's error on that
    # tensor, or 1.5 x its error on its worst tensor.
    sd32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    kw32 = dict(kw, u=u, u_fine=uf)
    with torch.no_grad():
        edges32 = O.render_rays_oracle(o, d, sd, None, cfg_of(Lx), **kw32)[0][3]["edges"]
    (rgb32, _, _, _), _, _ = O.render_rays_oracle(o, d, sd32, None, cfg_of(Lx), edges_override=edges32, **kw32)
    torch.nn.functional.mse_loss(rgb32, gt).backward()
    worst = {}
    e_f32_worst = max(_rel(sd32[name].grad, sdg[name].grad) for name, _ in m.named_parameters())
    for name, p in m.named_parameters():
        e_hip, e_f32 = _rel(p.grad, sdg[name].grad), _rel(sd32[name].grad, sdg[name].grad)
        worst[name] = (e_hip, e_f32)
        assert e_hip < max(2e-4, 3.0 * e_f32, 1.5 * e_f32_worst), \
            f"{name}: {e_hip:.2e} (float32 oracle end to end: {e_f32:.2e} on this tensor, {e_f32_worst:.2e} on its worst)"
    assert sum(e < 2e-4 for e, _ in worst.values()) >= len(worst) - 6, {k: f"{a:.1e}/{b:.1e}" for k, (a, b) in worst.items()}


@pytest.mark.gpu
def test_training_converges_on_rendered_targets():
    """End to end: a student NeRF fitted by Adam to images rendered from a teacher NeRF (reference loop shape,
    run-nerf.py:216-285: ray batch -> render_rays(train=True) -> mse -> backward -> step -> scheduler)."""
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.core.scheduler import ExponentialDecay
    from fs_nerf_amd.render import rendering as Rm
    dev = torch.device("cuda:0")

    def make(seed):
        sd = O.init_nerf_state_dict(4, 128, [], 10, 4, seed=seed)
        sd["sigma.weight"] *= 64.0
        sd["sigma.bias"] += 3.0
        m = NeRF(3, 3, 4, 128, (), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
        m.load_state_dict(sd)
        return m.to(dev)

    teacher, student = make(21).eval(), make(22).train()
    ro, rd = [], []
    for phi in (0.0, 90.0, 180.0, 270.0):
        o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, phi), (48, 48, 66.0))
        ro.append(o.reshape(-1, 3))
        rd.append(d.reshape(-1, 3))
    ro, rd = torch.cat(ro).contiguous().to(dev), torch.cat(rd).contiguous().to(dev)
    est_eval = Rm.StratifiedEstimator(2.0, 6.0, 64, 64)
    with torch.no_grad():
        gt = Rm.render_rays(ro, rd, est_eval, teacher, white_bkgd=True, device=dev)[0][0]
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 64).train()
    est.generator = torch.Generator(device=dev).manual_seed(0)
    opt = torch.optim.Adam(student.parameters(), lr=1e-3)
    sched = ExponentialDecay(opt, 300, 1e-3, r=0.1)
    gen = torch.Generator(device=dev).manual_seed(1)
    losses = []
    for it in range(300):
        idx = torch.randint(0, ro.shape[0], (1024,), device=dev, generator=gen)
        opt.zero_grad(set_to_none=True)
        rgb = Rm.render_rays(ro[idx], rd[idx], est, student, train=True, white_bkgd=True, device=dev)[0][0]
        loss = torch.nn.functional.mse_loss(rgb, gt[idx])
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss))
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    assert all(np.isfinite(losses))
    assert last < 0.25 * first, (first, last)
    student.eval()
    with torch.no_grad():  # the fused inference path sees the trained parameters (blob repacked)
        full = Rm.render_rays(ro, rd, est_eval, student, white_bkgd=True, device=dev)[0][0]
    assert float(torch.nn.functional.mse_loss(full, gt)) < 0.5 * first


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        for i, p in enumerate(net.parameters()):
            p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
        shard.allreduce_grads(net.parameters())
        q.put((rank, [float(p.grad.flatten()[0]) for p in net.parameters()],
               all(bool((p.grad == p.grad.flatten()[0]).all()) for p in net.parameters())))
    finally:
        dist.destroy_process_group()


def _flag_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        b = shard.FlatGrads(list(net.parameters()))
        out = []
        for step, flags in enumerate(((0.0, 0.0), (0.0, 1.0), (1.0, 1.0))):
            for i, p in enumerate(net.parameters()):
                p.grad.fill_(float(rank + 1) * (i + 1))
            b.allreduce(average=False, flag=flags[rank])
            out.append((float(b.flag_slot.item()), float(net[0].weight.grad.flatten()[0]), b.flat.numel()))
            b.flag_slot.zero_()
        bits = [shard.max_bits_over_ranks(pair[rank]) for pair in ((0, 0), (0, 2), (1, 0), (3, 1))]
        q.put((rank, out, bits))
    finally:
        dist.destroy_process_group()


def test_overflow_flag_travels_in_the_gradient_bucket_gloo():
    """in data-parallel training one rank may overflow fp16 while the others do not.  The bucket of
    the step's ONE all-reduce carries a flag slot behind the gradients (sum: > 0 on every rank when any rank raised
    it; `fsn_adam_step(skip_count)` then skips the update everywhere), and the host's amortised look at the range
    word takes the MAX over the ranks, so that all ranks fall back to bf16x3 together."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flag_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, bits in res:
        assert [o[0] for o in out] == [0.0, 1.0, 2.0], "flag slot = number of ranks that raised it, on every rank"
        assert all(o[1] == 3.0 for o in out), "gradients are summed as before"
        assert all(o[2] == 5 * 7 + 7 + 7 * 3 + 3 for o in out), "the gradient view keeps its size"
        assert bits == [0, 2, 1, 3], "MAX over ranks of the range words"


def test_gradient_allreduce_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # mean over ranks of (rank+1)*(i+1) = 1.5*(i+1), identical on both ranks, whole tensors updated
    for rank, vals, uniform in res:
        assert vals == [1.5, 3.0, 4.5, 6.0] and uniform


def _dp_worker_ragged(rank, world, port, q):
    """Ranks whose sets of defined gradients differ must still reduce the same bucket (zeros for the missing ones)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        ps = list(net.parameters())
        for i, p in enumerate(ps):
            if (i + rank) % 2 == 0:  # rank 0 defines grads 0,2; rank 1 defines grads 1,3
                p.grad = torch.full_like(p, float(10 * rank + i + 1))
        b = shard.allreduce_grads(ps)
        views_ok = all(p.grad.data_ptr() == b.view(i).data_ptr() for i, p in enumerate(ps))
        # second step through autograd: accumulation lands in the bucket views, no cat / copy-back
        b.zero()
        net(torch.ones(2, 5)).sum().backward()
        in_place = all(p.grad.data_ptr() == b.view(i).data_ptr() for i, p in enumerate(ps))
        b.allreduce()
        q.put((rank, [float(x) for x in b.flat[[0, 35, 42, 63]]] if False else
               [float(ps[i].grad.flatten()[0]) for i in range(4)], views_ok and in_place, int(b.numel)))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_fixed_bucket_with_missing_grads_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker_ragged, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "ranks disagree after the all-reduce"
    assert all(r[2] for r in res) and all(r[3] == 5 * 7 + 7 + 7 * 3 + 3 for r in res)


def _guard_nets(dev, scale_bad=4e5):
    from fs_nerf_amd.core.models import NeRF
    from test_parity_fp64 import scaled_sd
    L, D = 8, 256

    def mk(sd, prec):
        m = NeRF(3, 3, L, D, (4,), precision=prec, pos_fn={"n_freqs": 10, "log_space": True},
                 dir_fn={"n_freqs": 4, "log_space": True})
        m.load_state_dict(sd)
        m.weight_check = False  # (scaled_sd's 1/s on the connection weights is not what these tests are about)
        return m.to(dev).train()
    return mk, scaled_sd(L, D, 43, scale_bad), scaled_sd(L, D, 42, 1.0)


@pytest.mark.gpu
def test_training_range_guard_zeroes_gradients_on_device_then_continues_in_bf16x3():
    """Hidden activations beyond the fp16 range during training: the step's gradients are written as zeros by the
    backward kernels themselves (no host sync, no inf / NaN reaches the optimizer); at its next amortised look at its
    own accumulated word the host warns and the model continues in bf16x3, where the gradients are right again."""
    from fs_nerf_amd import ops
    from test_parity_fp64 import cfg_of
    dev = torch.device("cuda:0")
    L, N = 8, 600
    mk, sd, _ = _guard_nets(dev)
    m = mk(sd, "fp16x3")
    m.range_check_every = 2
    gen = torch.Generator().manual_seed(2)
    x = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
    c = torch.randn(N, 4, generator=gen).to(dev)
    assert ops.range_ok(dev)
    ops.step_flag(dev).zero_()
    (m(x, d) * c).sum().backward()  # step 1: flagged on the device, not yet seen by the host
    assert m.precision == "fp16x3"
    for name, p in m.named_parameters():
        assert float(p.grad.abs().max()) == 0.0, f"{name}: a flagged step must leave zero gradients"
    assert int(ops.step_flag(dev).item()) & 1, "the device's step flag is raised for the optimizer"
    assert ops.range_ok(dev), "training launches do not touch the inference path's sticky word"
    ops.step_flag(dev).zero_()
    m.zero_grad(set_to_none=True)
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        (m(x, d) * c).sum().backward()  # step 2: the host looks at the model's word
    assert m.precision == "bf16x3"
    ops.step_flag(dev).zero_()
    m.zero_grad(set_to_none=True)
    out = m(x, d)
    (out * c).sum().backward()  # step 3: bf16x3
    sdr = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.nerf_forward(sdr, x.cpu().double(), d.cpu().double(), **cfg_of(L))
    (ref * c.cpu().double()).sum().backward()
    assert bool(torch.isfinite(out).all()) and _rel(out, ref) < 1e-3
    for name, p in m.named_parameters():
        assert bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().max()) > 0.0, name
        if name in ("rgb.weight", "rgb.bias", "branch.bias", "sigma.bias"):
            assert _rel(p.grad, sdr[name].grad) < 5e-2, name
    assert ops.range_ok(dev) and int(ops.step_flag(dev).item()) == 0


@pytest.mark.gpu
def test_training_range_guard_is_per_call_not_a_shared_sticky_word():
    """the skip decision must come from THIS call's word.  (a) A bf16x3 model, and an fp16x3 model
    with range_check=False, keep their gradients although the inference path's sticky device word is raised (by an
    overflowing inference launch) and although another model overflowed in the same step.  (b) Two-model training:
    only the overflowing model's gradients are zeroed, the device's step flag is raised, FusedAdam skips the update
    of that step for BOTH (no half-applied step, no momentum drift on zero gradients) and updates again on the next
    clean step."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.optim import FusedAdam
    dev = torch.device("cuda:0")
    N = 300
    mk, sd_bad, sd_ok = _guard_nets(dev)
    gen = torch.Generator().manual_seed(4)
    x = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
    c = torch.randn(N, 4, generator=gen).to(dev)
    # raise the sticky word with an overflowing INFERENCE launch and leave it raised (nobody reads it)
    bad_inf = mk(sd_bad, "fp16x3").eval()
    bad_inf.range_check = False
    bad_inf.act_scaling = False  # (round 3's inference arithmetic: the scaled network of round 4 does not overflow here)
    with torch.no_grad():
        bad_inf(x, d)
    assert int(ops.status_word(dev).item()) & 1
    ops.step_flag(dev).zero_()
    for prec, rc in (("bf16x3", True), ("fp16x3", False)):
        m = mk(sd_ok, prec)
        m.range_check = rc
        (m(x, d) * c).sum().backward()
        for name, p in m.named_parameters():
            assert float(p.grad.abs().max()) > 0.0, f"{prec} range_check={rc}: {name} lost its gradient to a stale flag"
        assert int(ops.step_flag(dev).item()) == 0
    assert not ops.range_ok(dev)  # (clears the sticky word for the tests that follow)
    # two models, one optimizer, one of them overflows
    good, bad = mk(sd_ok, "fp16x3"), mk(sd_bad, "fp16x3")
    good.range_check_every = bad.range_check_every = 1000  # no host look during this test
    opt = FusedAdam(list(good.parameters()) + list(bad.parameters()), lr=1e-3)
    opt.zero_grad()
    ((good(x, d) * c).sum() + (bad(x, d) * c).nan_to_num(0.0, 0.0, 0.0).sum()).backward()
    for name, p in good.named_parameters():
        assert float(p.grad.abs().max()) > 0.0 and bool(torch.isfinite(p.grad).all()), f"good model {name}"
    for name, p in bad.named_parameters():
        assert float(p.grad.abs().max()) == 0.0, f"overflowing model {name}"
    before = [p.detach().clone() for p in opt.arena.params]
    opt.step()
    for p, b in zip(opt.arena.params, before):
        assert torch.equal(p.detach(), b), "the flagged step is skipped as a whole"
    assert float(opt.exp_avg.abs().max()) == 0.0, "no momentum from a skipped step"
    assert int(ops.step_flag(dev).item()) == 0
    opt.zero_grad()
    (good(x, d) * c).sum().backward()
    opt.step()
    assert any(not torch.equal(p.detach(), b) for p, b in zip(good.parameters(), before)), "clean step updates"


@pytest.mark.gpu
def test_two_optimizers_on_one_device_keep_their_step_flags_apart():
    """the skip word was one per device - an overflow in model A skipped optimizer B's update, and the
    first optimizer to step cleared the flag the second one needed.  The word now belongs to the gradient bucket: A's
    overflowing step skips A's update only, B trains on, in either order of the two `step()` calls."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.optim import FusedAdam
    dev = torch.device("cuda:0")
    N = 300
    mk, sd_bad, sd_ok = _guard_nets(dev)
    gen = torch.Generator().manual_seed(4)
    x = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
    c = torch.randn(N, 4, generator=gen).to(dev)
    for first in ("good", "bad"):
        good, bad = mk(sd_ok, "fp16x3"), mk(sd_bad, "fp16x3")
        good.range_check_every = bad.range_check_every = 1000  # no host look during this test
        og, ob = FusedAdam(good.parameters(), lr=1e-3), FusedAdam(bad.parameters(), lr=1e-3)
        ops.step_flag(dev).zero_()
        og.zero_grad(), ob.zero_grad()
        (good(x, d) * c).sum().backward()
        (bad(x, d) * c).nan_to_num(0.0, 0.0, 0.0).sum().backward()
        assert int(ob.grads.step_flag.item()) & 1 and int(og.grads.step_flag.item()) == 0 and int(ops.step_flag(dev).item()) == 0
        bg, bb = [p.detach().clone() for p in good.parameters()], [p.detach().clone() for p in bad.parameters()]
        for o in ((og, ob) if first == "good" else (ob, og)):
            o.step()
        assert any(not torch.equal(p.detach(), b) for p, b in zip(good.parameters(), bg)), f"{first} first: the clean model trains"
        assert all(torch.equal(p.detach(), b) for p, b in zip(bad.parameters(), bb)), f"{first} first: the flagged one is skipped"
        assert og.steps == 1 and ob.steps == 0 and int(ob.grads.step_flag.item()) == 0


@pytest.mark.gpu
def test_training_sampler_overflow_joins_the_step_guard_without_a_host_read():
    """render_rays(train=True) with the hierarchical sampler: the sampler's density pass reports into a per-call word that
    is OR-ed into the step's guard on the device (no read-back between sampler and forward).  An overflowing pass makes
    the step a skipped one; the host switches the model to bf16x3 at its next periodic look."""
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.optim import FusedAdam
    from fs_nerf_amd.render import rendering as Rm
    dev = torch.device("cuda:0")
    mk, sd_bad, _ = _guard_nets(dev)
    m = mk(sd_bad, "fp16x3")
    m.range_check_every = 2
    opt = FusedAdam(m.parameters(), lr=1e-3)
    est = Rm.StratifiedEstimator(2.0, 6.0, 16, 16).train()
    est.generator = torch.Generator(device=dev).manual_seed(3)
    gen = torch.Generator().manual_seed(5)
    o = torch.tensor([0.0, 0.0, 4.0]).repeat(200, 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(200, 3, generator=gen) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1).to(dev)
    assert ops.range_ok(dev)
    ops.step_flag(dev).zero_()

    def step():
        opt.zero_grad()
        (rgb, _, _, _), _, _ = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev)
        rgb.nan_to_num(0.0, 0.0, 0.0).square().mean().backward()

    before = [p.detach().clone() for p in m.parameters()]
    step()  # flagged on the device (sampler and forward both), not yet seen by the host
    assert m.precision == "fp16x3" and int(opt.grads.step_flag.item()) & 1 and int(ops.step_flag(dev).item()) == 0
    assert ops.range_ok(dev), "the training step does not touch the inference path's sticky word"
    opt.step()
    assert all(torch.equal(p.detach(), b) for p, b in zip(m.parameters(), before)), "a flagged step is skipped"
    with pytest.warns(RuntimeWarning, match="fp16 range"):
        step()
    assert m.precision == "bf16x3"
    opt.step()
    step()  # bf16x3: finite gradients, an update
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    assert int(ops.step_flag(dev).item()) == 0 and int(opt.grads.step_flag.item()) == 0


@pytest.mark.gpu
def test_backward_stage_scales_adapt_and_an_overflowing_stage_is_a_skipped_step():
    """Delayed per-stage gradient scaling (fsn_nerf_train_bwd, stage_scales): (a) the factors settle so that every stage's
    largest stored gradient sits near 2^5 whatever d(out)'s size; (b) factors that are far too large make the backward
    overflow - FSN_STATUS_GRAD_RANGE: zero gradients, the step flag raised for the optimizer, NO range fall-back - and
    the factors drop until the gradients are right again."""
    from fs_nerf_amd import ops, _lib as Lb
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    L, D, N = 8, 256, 640
    sd = O.init_nerf_state_dict(L, D, [4], 10, 4, seed=3)
    sd["sigma.weight"] *= 16.0
    m = NeRF(3, 3, L, D, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    m = m.to(dev).train()
    m.range_check_every = 1
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
    c = torch.randn(N, 4, generator=gen).to(dev)
    ops.step_flag(dev).zero_()

    def grads(cs):
        m.zero_grad(set_to_none=True)
        (m(x, d) * (c * cs)).sum().backward()
        return [p.grad.clone() for p in m.parameters()]

    g1 = grads(1.0)
    scales = m._bwd_stage[0].clone()
    assert float(scales[0]) >= 16.0 * float(scales[L - 1]), scales  # gradients shrink on the way back: earlier layers need more
    g2 = grads(1e-6)  # grad_scale absorbs d(out)'s size: same per-stage factors, gradients scale exactly
    assert bool((torch.log2(m._bwd_stage[0] / scales).abs() <= 1.0).all())
    for a, b in zip(g1, g2):
        assert float((a * 1e-6 - b).abs().max()) <= 2e-4 * float(b.abs().max())
    # (b) absurd factors: the next backward overflows and is a skipped step, not a precision fall-back
    m._bwd_stage[0].fill_(2.0 ** 34)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        bad = grads(1.0)
        assert all(float(g.abs().max()) == 0.0 for g in bad)
        assert int(ops.step_flag(dev).item()) & Lb.FSN_STATUS_GRAD_RANGE and m.precision == "fp16x3"
        ops.step_flag(dev).zero_()
        for _ in range(6):  # 2^34 -> 2^26 -> ... until nothing overflows and the measured maxima take over
            g = grads(1.0)
            ops.step_flag(dev).zero_()
    assert m.precision == "fp16x3" and m.grad_overflow_looks >= 1
    for a, b in zip(g1, g):
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("s", [3e-4, 1e-2, 1e2])
def test_nerf_gradients_over_the_activation_envelope(s):
    """The forward's float32-grade envelope (hidden activations s x the default network's, s = 3e-4 .. 1e2: layer maxima
    from ~1e-3 to ~1e3) holds for the GRADIENTS too: every parameter gradient within 2e-4 of the tensor's largest
    entry against float64 autograd.  (The weight-gradient GEMM keeps the activations' low parts scaled and moves the
    2^-11 to the gradient operand; the dgrad chain scales every stage by its own power of two.)"""
    from fs_nerf_amd.core.models import NeRF
    from test_parity_fp64 import scaled_sd, cfg_of
    import warnings
    dev = torch.device("cuda:0")
    L, D, N = 8, 256, 777
    sd = scaled_sd(L, D, 42, s)
    m = NeRF(3, 3, L, D, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    m = m.to(dev).train()
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(16 * N, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(16 * N, 3, generator=gen), dim=-1)
    keep = _relu_margin(sd, x, d, L, [4], 10, 4) > 2e-5 * s  # (every hidden pre-activation scales by s)
    x, d = x[keep][:N].contiguous(), d[keep][:N].contiguous()
    assert x.shape[0] == N
    c = torch.randn(N, 4, generator=gen)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        out = m(x.to(dev), d.to(dev))
        (out * c.to(dev)).sum().backward()
    assert m.precision == "fp16x3"
    sdr = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.nerf_forward(sdr, x.double(), d.double(), **cfg_of(L))
    (ref * c.double()).sum().backward()
    for name, p in m.named_parameters():
        err = _rel(p.grad, sdr[name].grad)
        assert err < 2e-4, (s, name, err)


@pytest.mark.gpu
def test_c_abi_backward_with_and_without_stage_arrays():
    """fsn_nerf_train_bwd called as a C-ABI user would: without the stage arrays (factors all 1: rounds 1-2's behaviour),
    with fresh ones (first call = calibration) and with calibrated ones, into fresh buffers and accumulating - the same
    gradients within the 2e-4 bar, `accumulate` adds exactly, the arrays are moved by every call."""
    from fs_nerf_amd import ops, _lib as Lb
    from fs_nerf_amd.core.models import NeRF
    dev = torch.device("cuda:0")
    L, D, N = 8, 256, 1000
    sd = O.init_nerf_state_dict(L, D, [4], 10, 4, seed=8)
    sd["sigma.weight"] *= 16.0
    m = NeRF(3, 3, L, D, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    m = m.to(dev)
    ws, bs = m._tensors()
    desc = ops.make_desc(L, D, (4,), m.pos_encoder.freqs, m.dir_encoder.freqs)
    gen = torch.Generator().manual_seed(3)
    x = (torch.rand(N, 3, generator=gen) * 2 - 1).to(dev)
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1).to(dev)
    c = torch.randn(N, 4, generator=gen).to(dev)
    prec = Lb.FSN_PREC_FP16X3

    def bwd(stage, into=None):
        word = torch.zeros(1, dtype=torch.int32, device=dev)
        out, work = ops.nerf_train_fwd(desc, prec, ws, bs, x, d, None, None, status=word)
        dW, db = ops.nerf_train_bwd(desc, prec, ws, work, out, c, status=word, into=into, stage_state=stage)
        assert int(word.item()) == 0
        return [g.clone() for g in dW] + [g.clone().reshape(-1) for g in db]

    plain = bwd(None)
    stage = (torch.ones(L + 2, device=dev), torch.zeros(L + 2, dtype=torch.int32, device=dev))
    first = bwd(stage)  # factors 1 during the call, measured maxima applied afterwards
    assert all(torch.equal(a, b) for a, b in zip(plain, first)), "fresh arrays = factors of 1 = the plain call"
    assert not bool((stage[0] == 1.0).all()) and int(stage[1].abs().max()) == 0, "the call left the next call's factors"
    second = bwd(stage)
    for a, b in zip(plain, second):
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max())
    # accumulate: buffers holding `second` receive the same gradients once more
    bufW = [g.clone() for g in second[:len(ws)]]
    bufb = [g.clone() for g in second[len(ws):]]
    third = bwd(stage, into=(bufW, bufb))
    fresh = bwd(stage)
    for acc, s2, f in zip(third, second, fresh):
        assert float((acc - (s2 + f)).abs().max()) <= 1e-6 * float((s2 + f).abs().max()) + 1e-30
