"""CPU: self-consistency of the oracle parts that the reference does not pin (sampling, compositing,
resampling — SURVEY.md 8c "parity unpinned"): dense vs packed forms agree, known-answer cases, and the
properties the definitions promise."""
import numpy as np
import torch

from oracle import fsnerf_oracle as O


def test_composite_known_answers():
    # constant sigma, unit intervals: w_i = exp(-s i)(1 - exp(-s)); opacity = 1 - exp(-s S)
    S, s = 8, 0.7
    t0 = torch.arange(S, dtype=torch.float64)[None]
    sig = torch.full((1, S), s, dtype=torch.float64)
    rgb = torch.ones(1, S, 3, dtype=torch.float64) * 0.25
    c, o, d, ex = O.composite(sig, rgb, t0, t0 + 1, torch.ones(3, dtype=torch.float64))
    i = torch.arange(S, dtype=torch.float64)
    np.testing.assert_allclose(ex["weights"][0].numpy(), (torch.exp(-s * i) * (1 - np.exp(-s))).numpy(), rtol=1e-12)
    np.testing.assert_allclose(float(o), 1 - np.exp(-s * S), rtol=1e-12)
    np.testing.assert_allclose(c[0].numpy(), 0.25 * float(o) + (1 - float(o)), rtol=1e-12)
    # sigma = 0: pure background, depth 0 (the reference's fallback values, rendering.py:97-103)
    c, o, d, _ = O.composite(torch.zeros(2, S), torch.rand(2, S, 3), t0.float().expand(2, S), t0.float().expand(2, S) + 1,
                             torch.tensor([1.0, 1.0, 1.0]))
    assert float(o.abs().max()) == 0 and float(d.abs().max()) == 0 and float((c - 1).abs().max()) == 0


def test_packed_equals_dense():
    g = torch.Generator().manual_seed(0)
    R, S = 5, 17
    edges = O.stratified_edges(2.0, 6.0, S, R, torch.rand(R, S + 1, generator=g), dtype=torch.float64)
    sig = torch.rand(R, S, generator=g, dtype=torch.float64) * 5
    rgb = torch.rand(R, S, 3, generator=g, dtype=torch.float64)
    bk = torch.tensor([0.2, 0.4, 0.6], dtype=torch.float64)
    dense = O.composite(sig, rgb, edges[:, :-1], edges[:, 1:], bk)
    ri, t0, t1 = O.edges_to_packed(edges)
    packed = O.rendering_packed(t0, t1, ri, R, lambda a, b, c: (rgb.reshape(-1, 3), sig.reshape(-1)), bk)
    for a, b in zip(dense[:3], packed[:3]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(dense[3]["weights"].reshape(-1).numpy(), packed[3]["weights"].numpy(), rtol=1e-12)


def test_stratified_edges_properties():
    g = torch.Generator().manual_seed(1)
    R, S = 50, 64
    e0 = O.stratified_edges(2.0, 6.0, S, R)
    assert float(e0[:, 0].min()) == 2.0 and abs(float(e0[:, -1].max()) - 6.0) < 1e-6
    u = torch.rand(R, generator=g)
    e1 = O.stratified_edges(2.0, 6.0, S, R, u)
    np.testing.assert_allclose((e1 - e0).numpy(), (u[:, None] * (4.0 / S)).expand(R, S + 1).numpy(), atol=1e-6)
    e2 = O.stratified_edges(2.0, 6.0, S, R, torch.rand(R, S + 1, generator=g))
    assert bool((e2[:, 1:] >= e2[:, :-1]).all()) and float(e2.min()) >= 2.0 and float(e2.max()) <= 6.0
    half = O.stratified_edges(2.0, 6.0, S, R, torch.full((R, S + 1), 0.5))  # u = 1/2: interior edges unmoved
    np.testing.assert_allclose(half[:, 1:-1].numpy(), e0[:, 1:-1].numpy(), atol=1e-6)


def test_sample_pdf_properties():
    g = torch.Generator().manual_seed(2)
    R, S, N = 20, 64, 128
    edges = O.stratified_edges(2.0, 6.0, S, R, dtype=torch.float64)
    w = torch.zeros(R, S, dtype=torch.float64)
    w[:, 20] = 1.0                                    # all mass in bin 20
    t = O.sample_pdf(edges, w, N, torch.rand(R, N, generator=g, dtype=torch.float64))
    inside = (t >= edges[:, 20:21]) & (t <= edges[:, 21:22])
    assert inside.double().mean() > 0.99              # 64 * 1e-5 of the mass is the uniform floor
    t_uni = O.sample_pdf(edges, torch.zeros(R, S, dtype=torch.float64), N)  # no mass: uniform, deterministic u
    np.testing.assert_allclose(t_uni[0].numpy(), np.linspace(2.0, 6.0, N), atol=1e-9)
    m = O.merge_edges(edges, t)
    assert m.shape == (R, S + 1 + N) and bool((m[:, 1:] >= m[:, :-1]).all())
    # negative weights are clamped, not allowed to make the cdf non-monotone
    w2 = w.clone()
    w2[:, 40:] = -3.0
    t2 = O.sample_pdf(edges, w2, N, torch.rand(R, N, generator=g, dtype=torch.float64))
    assert bool(torch.isfinite(t2).all()) and float(t2.min()) >= 2.0 and float(t2.max()) <= 6.0


def test_render_rays_oracle_reduces_to_reference_formulas():
    # mask == 1, one network, n_importance = 0: rgb = sum w rgb + bkgd (1 - sum w), positions at midpoints
    sd = O.init_nerf_state_dict(4, 128, [4], 10, 4, seed=42)
    cfg = dict(n_layers=4, skip=[4], n_freqs=10, n_freqs_dir=4, log_space=True)
    o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, 0.0), (4, 4, 5.0))
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    (rgb, op, dep, ex), ri, tv = O.render_rays_oracle(o, d, sd, None, cfg, near=2.0, far=6.0, n_samples=8,
                                                      white_bkgd=True, pos_mask=O.freq_mask(3, 10, 1.0))
    (rgb2, *_), _, _ = O.render_rays_oracle(o, d, sd, None, cfg, near=2.0, far=6.0, n_samples=8, white_bkgd=True)
    assert torch.equal(rgb, rgb2)
    x = o[ri] + d[ri] * tv[:, None]
    out = O.nerf_forward(sd, x, d[ri], **cfg)
    np.testing.assert_allclose(ex["sigmas"].reshape(-1).numpy(), out[:, 3].numpy(), rtol=1e-5, atol=1e-6)
    w = ex["weights"]
    np.testing.assert_allclose(rgb.numpy(), ((w[..., None] * ex["rgbs"]).sum(1) + (1 - w.sum(1, keepdim=True))).numpy(),
                               rtol=1e-5, atol=1e-6)
