"""GPU parity tests: every HIP entry point (through the C-ABI, via the Python host layer)
against the CPU oracle on the same seeded inputs, and against the committed golden vectors.

Tolerances (north_star: 1e-4 relative fp32 on rgb_map, depth_map, weights):
  RTOL = 1e-4 with a small absolute floor (values that are ~0 have no meaningful relative error):
  rgb/opacity/weights ATOL 1e-5 (quantities in [0,1]), depth ATOL 1e-4 (metres, range ~6),
  raw sigma ATOL 1e-4 * max|sigma| (sigma is a 256-term dot product with cancellation).
"""
import os

import numpy as np
import pytest
import torch

from oracle import fsnerf_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import fs_nerf_amd  # noqa: F401
    from fs_nerf_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def close(a, b, rtol=RTOL, atol=1e-5, what=""):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    if not (err <= 0).all():
        i = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError(f"{what}: max violation at {i}: got {a[i]!r} want {b[i]!r} "
                             f"(|diff| {abs(a[i] - b[i]):.3e}, {(err > 0).sum()} of {err.size} out of tolerance)")


def load_sd(golden_dir, tag, sigma64=True):
    g = np.load(os.path.join(golden_dir, f"g4_nerf_{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    if sigma64:
        sd["sigma.weight"] = sd["sigma.weight"] * 64.0
        sd["sigma.bias"] = sd["sigma.bias"] + 3.0
    return g, sd


def make_model(sd, n_layers, d_hidden, skip, dev, nf=10, nfd=4, precision="fp16x3"):
    from fs_nerf_amd.core.models import NeRF
    m = NeRF(3, 3, n_layers, d_hidden, skip, precision=precision,
             pos_fn={"n_freqs": nf, "log_space": True}, dir_fn={"n_freqs": nfd, "log_space": True})
    m.load_state_dict(sd)
    return m.to(dev).eval()


CFG = {"8x256": dict(n_layers=8, skip=[4], n_freqs=10, n_freqs_dir=4, log_space=True),
       "4x128": dict(n_layers=4, skip=[4], n_freqs=10, n_freqs_dir=4, log_space=True)}
DIMS = {"8x256": (8, 256), "4x128": (4, 128)}


def ndc_error_bound(o, d, hwf, near):
    """float64 evaluation of to_ndc (utilities.py:84-120) and a first-order bound of the float32 rounding error of
    its operation sequence, per element.  eps = 2^-24 (half an ulp) per operation."""
    H, W, focal = hwf
    o, d = o.astype(np.float64), d.astype(np.float64)
    eps = 2.0 ** -24
    with np.errstate(divide="ignore", invalid="ignore"):
        num = near + o[:, 2]
        t = -num / d[:, 2]
        e_t = np.abs(t) * 2 * eps + eps * (np.abs(near) + np.abs(o[:, 2])) / np.abs(d[:, 2])
        op = o + t[:, None] * d
        e_op = eps * (np.abs(o) + 2 * np.abs(t[:, None] * d)) + e_t[:, None] * np.abs(d) + eps * np.abs(op)
        sx, sy = -1.0 / (W / (2.0 * focal)), -1.0 / (H / (2.0 * focal))
        sc = np.array([sx, sy])
        r = op[:, :2] / op[:, 2:3]
        e_r = np.abs(r) * (e_op[:, :2] / np.abs(op[:, :2]).clip(1e-300) + e_op[:, 2:3] / np.abs(op[:, 2:3]) + eps)
        q = d[:, :2] / d[:, 2:3]
        e_q = np.abs(q) * eps
        o01 = sc * r
        e_o01 = np.abs(sc) * e_r + 2 * eps * np.abs(o01)
        o2 = 1.0 + 2.0 * near / op[:, 2]
        e_o2 = np.abs(2.0 * near / op[:, 2]) * (e_op[:, 2] / np.abs(op[:, 2]) + 2 * eps) + eps * np.abs(o2)
        d01 = sc * (q - r)
        e_d01 = np.abs(sc) * (e_q + e_r + eps * np.abs(q - r)) + 2 * eps * np.abs(d01)
        d2 = -2.0 * near / op[:, 2]
        e_d2 = np.abs(d2) * (e_op[:, 2] / np.abs(op[:, 2]) + 3 * eps)
    to = np.concatenate([o01, o2[:, None]], -1)
    td = np.concatenate([d01, d2[:, None]], -1)
    bo = np.concatenate([e_o01, e_o2[:, None]], -1) + 1e-37
    bd = np.concatenate([e_d01, e_d2[:, None]], -1) + 1e-37
    return bo, bd, to, td


# ------------------------------------------------------------------ a1/a2/a3 rays
@pytest.mark.parametrize("pname", ["identity", "orbit0", "orbit3x4", "random"])
@pytest.mark.parametrize("hname", ["small", "lego100"])
def test_get_rays_to_ndc_golden(dev, golden_dir, pname, hname):
    from fs_nerf_amd.utils import utilities as U
    g = np.load(os.path.join(golden_dir, "g1_rays.npz"))
    key = f"{pname}_{hname}"
    hwf = (int(g[key + "_hwf"][0]), int(g[key + "_hwf"][1]), float(g[key + "_hwf"][2]))
    o, d = U.get_rays(torch.from_numpy(g[key + "_pose"]), hwf, dev)
    assert o.shape == (hwf[0], hwf[1], 3) and d.shape == o.shape
    close(o, g[key + "_o"], rtol=0, atol=0, what="origins")
    close(d, g[key + "_d"], rtol=1e-6, atol=1e-7, what="dirs")
    no, nd = U.to_ndc(torch.from_numpy(g[key + "_o"]).to(dev).reshape(-1, 3),
                      torch.from_numpy(g[key + "_d"]).to(dev).reshape(-1, 3), hwf, 1.0)
    # to_ndc is 3 divides and a handful of products; its only ill-conditioning is the SINGULAR SET d_z = 0 (ray
    # parallel to the near plane: t = -(near + o_z) / d_z) and the cancellation in o' = o + t d.  Instead of a blanket
    # tolerance the test evaluates the first-order rounding-error bound of the reference's operation sequence
    # (utilities.py:104-114) per element in float64 and allows 1 x that (the reference-generated golden itself sits at
    # 0.26 x, tools: this test's helper on the CPU): a few float32 ulps wherever the projection is
    # well conditioned (rtol ~1e-6 on the forward-facing rays it is used for), wider only where the float32 reference
    # itself is.  Checked against a float64 evaluation AND against the reference-generated golden.
    bo, bd, to64, td64 = ndc_error_bound(g[key + "_o"].reshape(-1, 3), g[key + "_d"].reshape(-1, 3), hwf, 1.0)
    ok = np.isfinite(bo).all(-1) & np.isfinite(bd).all(-1) & np.isfinite(g[key + "_ndc_o"]).all(-1) & \
        np.isfinite(g[key + "_ndc_d"]).all(-1)
    assert ok.mean() > 0.9 or pname != "identity", "the singular set is a null set"
    for got, t64, bnd, gold, what in ((no, to64, bo, g[key + "_ndc_o"], "ndc_o"), (nd, td64, bd, g[key + "_ndc_d"], "ndc_d")):
        got = got.cpu().double().numpy()[ok]
        assert (np.abs(got - t64[ok]) <= 1.0 * bnd[ok]).all(), \
            f"{what}: worst {float((np.abs(got - t64[ok]) / bnd[ok]).max()):.2f} x the rounding-error bound"
        assert (np.abs(got - gold[ok].astype(np.float64)) <= 1.5 * bnd[ok]).all(), f"{what} vs the reference golden"
    well = ok & (np.abs(g[key + "_d"].reshape(-1, 3)[:, 2]) > 0.1)
    if well.any():  # forward-facing rays: plain relative tolerance
        sc = np.abs(to64[well]).max(-1, keepdims=True) + 1.0
        assert (np.abs(no.cpu().double().numpy()[well] - to64[well]) <= 1e-6 * sc).all(), "ndc_o rtol 1e-6 away from d_z ~ 0"


def test_get_rays_row_blocks_and_800(dev):
    from fs_nerf_amd import ops
    pose = O.pose_from_spherical(4.0311289, 50.0, 77.0)
    H = W = 800
    focal = 1111.111
    o, d = ops.get_rays(pose, H, W, focal, dev)
    ro, rd = O.get_rays(pose, (H, W, focal))
    close(d.reshape(H, W, 3), rd, rtol=1e-6, atol=1e-7, what="800x800 dirs")
    o2, d2 = ops.get_rays(pose, H, W, focal, dev, row0=300, nrows=100)  # a rank's row block
    assert torch.equal(d2, d.reshape(H, W, 3)[300:400].reshape(-1, 3))
    assert torch.equal(o2, o.reshape(H, W, 3)[300:400].reshape(-1, 3))
    from fs_nerf_amd.utils import utilities as U
    assert [c.shape[0] for c in U.get_chunks(o, 250000)] == [250000, 250000, 140000]
    with pytest.raises(RuntimeError):
        U.get_rays(pose, (4, 4, 2.0), torch.device("cpu"))


# ------------------------------------------------------------------ a4 encoder
@pytest.mark.parametrize("n,ls", [(10, True), (4, True), (10, False), (4, False), (0, True)])
def test_posenc(dev, golden_dir, n, ls):
    from fs_nerf_amd.core.models import PositionalEncoder
    g = np.load(os.path.join(golden_dir, "g3_posenc.npz"))
    enc = PositionalEncoder(3, n, ls)
    x = torch.from_numpy(g["x"]).to(dev)
    y = enc(x)
    assert y.shape == (256, enc.d_output)
    if n > 0:
        close(y, g[f"pe_n{n}_log{int(ls)}"], rtol=0, atol=2e-6, what="posenc vs reference golden")
        m = O.freq_mask(3, n, 0.45)
        close(enc(x, m.to(dev)), O.posenc(torch.from_numpy(g["x"]), n, ls, m), rtol=0, atol=2e-6, what="masked")
    else:
        assert torch.equal(y, x)
    assert enc(torch.zeros(0, 3, device=dev)).shape == (0, enc.d_output)


# ------------------------------------------------------------------ a8 sampler
@pytest.mark.parametrize("mode", ["none", "ray", "edge"])
@pytest.mark.parametrize("S", [64, 100, 7])
def test_stratified_edges(dev, mode, S):
    from fs_nerf_amd import ops
    R = 333
    gen = torch.Generator().manual_seed(5)
    u = None if mode == "none" else (torch.rand(R, generator=gen) if mode == "ray" else torch.rand(R, S + 1, generator=gen))
    want = O.stratified_edges(2.0, 6.0, S, R, u)
    got = ops.stratified_edges(2.0, 6.0, S, R, None if u is None else u.to(dev), dev)
    close(got, want, rtol=0, atol=1e-6, what="edges")
    assert bool((got[:, 1:] >= got[:, :-1]).all())
    ri, t0, t1 = ops.edges_to_packed(got)
    wri, wt0, wt1 = O.edges_to_packed(got.cpu())
    assert ri.dtype == torch.int64 and torch.equal(ri.cpu(), wri)
    assert torch.equal(t0.cpu(), wt0) and torch.equal(t1.cpu(), wt1)


@pytest.mark.parametrize("S,NI,det", [(64, 128, True), (64, 128, False), (32, 16, False), (128, 256, True), (5, 3, False)])
def test_sample_pdf_merge(dev, S, NI, det):
    from fs_nerf_amd import ops
    R = 257
    gen = torch.Generator().manual_seed(11)
    edges = O.stratified_edges(2.0, 6.0, S, R, torch.rand(R, generator=gen))
    w = torch.rand(R, S, generator=gen) ** 4
    w[::7] = 0.0        # rays with no mass: uniform pdf
    w[3, S // 2:] = -0.2  # negative weights are clamped
    u = None if det else torch.rand(R, NI, generator=gen)
    got = ops.sample_pdf_merge(edges.to(dev), w.to(dev), NI, None if u is None else u.to(dev))
    assert got.shape == (R, S + 1 + NI)
    assert bool((got[:, 1:] >= got[:, :-1]).all()), "sorted union"
    # Truth = the oracle's arithmetic in float64.  The inverse CDF divides by the mass `denom` of the interval the
    # sample falls into: an error delta of the float32 cdf (an S-term running sum: bounded by 2 S 2^-24) moves the sample
    # by delta * width / denom, so THAT is the tolerance, per sample (+ 4 ulps of t): 2e-6 wherever an interval holds
    # real mass, wider only for the rare samples that land in nearly empty intervals.  The definition is discontinuous
    # only through the branch `denom < 1e-5 -> 1`: (a) a sample whose denom lies within the cdf error of the
    # threshold, (b) a sample whose u lies within the cdf error of the END of such a nearly empty interval (inside it
    # t stays at its left edge, just outside it t continues from its right edge - the deterministic u = 1.0 hits this
    # whenever the last interval is empty, since cdf[S] rounds to either side of 1).  Rays with such a sample (the tie
    # set) are excluded from the exact check - they are counted and bounded by one interval width.
    # Sorting is 1-Lipschitz in the sup norm, so the merged rows are compared with each row's largest tolerance.
    e64, w64 = edges.double(), torch.clamp(w.double(), min=0.0) + 1e-5
    pdf = w64 / w64.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros(R, 1, dtype=torch.float64), torch.cumsum(pdf, -1)], -1)
    uu = (torch.linspace(0.0, 1.0, NI)[None, :].expand(R, NI) if u is None else u).double().contiguous()
    idx = torch.searchsorted(cdf.contiguous(), uu, right=True)
    below, above = torch.clamp(idx - 1, min=0), torch.clamp(idx, max=S)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    e0, e1 = torch.gather(e64, 1, below), torch.gather(e64, 1, above)
    raw = c1 - c0
    den = torch.where(raw < 1e-5, torch.ones_like(raw), raw)
    t64 = e0 + (uu - c0) / den * (e1 - e0)
    want64 = torch.sort(torch.cat([e64, t64], -1), -1).values
    delta = 2.0 * S * 2.0 ** -24  # first-order bound of a float32 running sum of S terms that stays <= 1
    tol_s = 2e-6 + delta * (e1 - e0) / den
    big = torch.full_like(raw, 1.0)  # (no interval beyond the ends of the ray)
    m_prev = torch.where(below > 0, c0 - torch.gather(cdf, 1, torch.clamp(below - 1, min=0)), big)
    thin = 1e-5 + delta
    m_cur = torch.where(above > below, raw, big)  # (u >= cdf[S]: below == above == S, no interval of its own)
    # (u just below c0 sits in the previous interval: discontinuous there iff THAT one is thin; u just below c1 sits
    # in the current one: discontinuous iff the current one is thin)
    tie_s = ((m_cur - 1e-5).abs() < delta) | (((uu - c0).abs() < delta) & (m_prev < thin)) | \
        (((uu - c1).abs() < delta) & (m_cur < thin))
    tie = tie_s.any(-1)
    err = (got.cpu().double() - want64).abs().amax(-1)
    row_tol = tol_s.amax(-1)
    # (deterministic u ends with u = 1.0 exactly: every ray whose LAST interval is thin - a quarter of these random
    # rays - is a tie there; with random u ties are rare)
    assert float(tie.float().mean()) < (0.30 if det else 0.02), "tie rays"
    bad = (~tie) & (err > row_tol)
    assert not bool(bad.any()), f"{int(bad.sum())} rays outside the per-sample bound (worst {float((err / row_tol)[~tie].max()):.2f} x)"
    assert bool((err[tie] <= (6.0 - 2.0) / S + 1e-6).all()), "tie rays: within one coarse interval"
    # and the float32 oracle sits inside the same bound (the bound is about the arithmetic, not about this kernel)
    want32 = O.merge_edges(edges, O.sample_pdf(edges, w, NI, u)).double()
    e32 = (want32 - want64).abs().amax(-1)
    assert not bool(((~tie) & (e32 > row_tol)).any())


# ------------------------------------------------------------------ a7 compositing
@pytest.mark.parametrize("S", [64, 192, 5, 384, 1])
@pytest.mark.parametrize("white", [False, True])
def test_composite_dense(dev, S, white):
    from fs_nerf_amd import ops
    R = 130
    gen = torch.Generator().manual_seed(S)
    edges = O.stratified_edges(2.0, 6.0, S, R, torch.rand(R, generator=gen))
    t0, t1 = edges[:, :-1].contiguous(), edges[:, 1:].contiguous()
    # raw sigma, negatives allowed (models.py:127); sigma*dt ~ N(0.25, 0.5) whatever S is
    sig = (torch.randn(R, S, generator=gen) * 0.5 + 0.25) / (4.0 / S)
    rgb = torch.rand(R, S, 3, generator=gen)
    bk = torch.ones(3) * float(white)
    wc, wo, wd, wex = O.composite(sig.double(), rgb.double(), t0.double(), t1.double(), bk.double())
    c, o, d, ex = ops.composite(sig.to(dev), rgb.to(dev), t0.to(dev), t1.to(dev), bk)
    assert c.shape == (R, 3) and o.shape == (R, 1) and d.shape == (R, 1)
    scale = float(wex["weights"].abs().max())  # negative sigma can push T above 1
    close(ex["weights"], wex["weights"], atol=1e-5 * max(scale, 1.0), what="weights")
    close(ex["alphas"], wex["alphas"], atol=1e-5, what="alphas")
    close(ex["trans"], wex["trans"], atol=1e-5 * float(wex["trans"].abs().max()), what="trans")
    close(c, wc, atol=1e-5 * max(scale, 1.0), what="colors")
    close(o, wo, atol=1e-5 * max(scale, 1.0), what="opacity")
    ok = (wo > 1e-2).squeeze(-1)  # depth divides by max(opacity, eps): ill-conditioned near / below 0
    # with raw (signed) sigma the weights have mixed signs: condition number of depth = sum|w| / |sum w|
    cond = (wex["weights"].abs().sum(-1, keepdim=True) / wo.abs()).clamp(min=1.0)
    close(d[ok.to(dev)], wd[ok], atol=(1e-4 * cond[ok]).numpy(), what="depth")


def test_composite_packed_ragged(dev):
    from fs_nerf_amd.render import rendering as Rm
    gen = torch.Generator().manual_seed(2)
    counts = torch.tensor([0, 3, 64, 0, 1, 200, 17, 0])  # empty rays in front, middle and at the end
    R = len(counts)
    ri = torch.repeat_interleave(torch.arange(R), counts)
    N = int(counts.sum())
    t0 = torch.rand(N, generator=gen) * 0.01 + torch.arange(N) * 0.02
    t1 = t0 + 0.02
    sig = torch.rand(N, generator=gen) * 30.0
    rgb = torch.rand(N, 3, generator=gen)
    bk = torch.tensor([1.0, 1.0, 1.0])
    want = O.rendering_packed(t0, t1, ri, R, lambda a, b, c: (rgb, sig), bk)
    got = Rm.rendering(t0.to(dev), t1.to(dev), ri.to(dev), R, lambda a, b, c: (rgb.to(dev), sig.to(dev)), bk.to(dev))
    for k in range(3):
        close(got[k], want[k], atol=1e-5 if k != 2 else 1e-4, what=f"packed out {k}")
    close(got[3]["weights"], want[3]["weights"], what="packed weights")
    # rays without samples: background colour, zero opacity and depth (rendering.py:97-103)
    for r in (0, 3, 7):
        assert got[0][r].tolist() == [1.0, 1.0, 1.0] and float(got[1][r]) == 0.0 and float(got[2][r]) == 0.0
    with pytest.raises(AssertionError):
        Rm.rendering(t0.to(dev), t1.to(dev), ri.to(dev), R, lambda a, b, c: (rgb.to(dev)[:, :2], sig.to(dev)), None)
    # zero samples in total
    e = torch.zeros(0, device=dev)
    out = Rm.rendering(e, e, torch.zeros(0, dtype=torch.int64, device=dev), 4,
                       lambda a, b, c: (torch.zeros(0, 3, device=dev), e), torch.zeros(3, device=dev))
    assert float(out[0].abs().max()) == 0.0 and out[0].shape == (4, 3)


# ------------------------------------------------------------------ a5 MLP
@pytest.mark.parametrize("tag", ["8x256", "4x128"])
def test_nerf_forward_golden(dev, golden_dir, tag):
    g, sd = load_sd(golden_dir, tag, sigma64=False)
    m = make_model(sd, *DIMS[tag], [4], dev)
    x, d = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["dirs"]).to(dev)
    with torch.no_grad():
        y4, y1 = m(x, d), m(x)
    assert y4.shape == (256, 4) and y1.shape == (256, 1)
    s_atol = 1e-4 * float(np.abs(g["y_sigma"]).max())
    close(y4[:, :3], g["y_full"][:, :3], atol=1e-5, what="rgb vs reference golden")
    close(y4[:, 3], g["y_full"][:, 3], atol=s_atol, what="sigma vs reference golden")
    close(y1, g["y_sigma"], atol=s_atol, what="density-only vs reference golden")
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["sigma.weight"] *= 64.0
    sd2["sigma.bias"] += 1.0
    m.load_state_dict(sd2)  # parameters changed -> repacked
    with torch.no_grad():
        y4b = m(x, d)
    close(y4b[:, 3], g["y_full_sigma64"][:, 3], atol=1e-4 * float(np.abs(g["y_full_sigma64"][:, 3]).max()),
          what="sigma64 vs reference golden")
    assert set(m.state_dict().keys()) == set(sd.keys())


@pytest.mark.parametrize("tag,n", [("8x256", 1), ("8x256", 127), ("8x256", 129), ("4x128", 1000), ("8x256", 70001)])
def test_nerf_forward_sizes_vs_oracle(dev, golden_dir, tag, n):
    _, sd = load_sd(golden_dir, tag)
    m = make_model(sd, *DIMS[tag], [4], dev)
    gen = torch.Generator().manual_seed(n)
    x = torch.rand(n, 3, generator=gen) * 3.0 - 1.5
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    sd64 = {k: v.double() for k, v in sd.items()}
    want = O.nerf_forward(sd64, x.double(), d.double(), **CFG[tag])
    with torch.no_grad():
        got = m(x.to(dev), d.to(dev))
        got1 = m(x.to(dev))
    s_atol = 1e-4 * float(want[:, 3].abs().max())
    close(got[:, :3], want[:, :3], atol=1e-5, what="rgb")
    close(got[:, 3], want[:, 3], atol=s_atol, what="sigma")
    close(got1[:, 0], want[:, 3], atol=s_atol, what="density-only")
    assert m(torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev)).shape == (0, 4)


def test_nerf_forward_mask_skips_and_bf16(dev):
    sd = O.init_nerf_state_dict(6, 128, [1, 3], 7, 3, seed=7)
    m = make_model(sd, 6, 128, [1, 3], dev, nf=7, nfd=3)
    pm, dm = O.freq_mask(3, 7, 0.6), O.freq_mask(3, 3, 0.5)
    m.set_freq_mask(pm, dm)
    gen = torch.Generator().manual_seed(9)
    x = torch.rand(500, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(500, 3, generator=gen), dim=-1)
    want = O.nerf_forward({k: v.double() for k, v in sd.items()}, x.double(), d.double(), n_layers=6, skip=[1, 3],
                          n_freqs=7, n_freqs_dir=3, pos_mask=pm.double(), dir_mask=dm.double())
    with torch.no_grad():
        got = m(x.to(dev), d.to(dev))
    close(got[:, :3], want[:, :3], atol=1e-5, what="masked rgb")
    close(got[:, 3], want[:, 3], atol=1e-4 * float(want[:, 3].abs().max()), what="masked sigma")
    # single-pass bf16 mode (BASELINE config 5): relaxed tolerance 2e-2 absolute on rgb
    m16 = make_model(sd, 6, 128, [1, 3], dev, nf=7, nfd=3, precision="bf16")
    m16.set_freq_mask(pm, dm)
    with torch.no_grad():
        got16 = m16(x.to(dev), d.to(dev))
    close(got16[:, :3], want[:, :3], rtol=0, atol=2e-2, what="bf16 rgb")
    m.train()
    assert m(x.to(dev), d.to(dev)).requires_grad  # training forward (tests/test_train_step.py checks the gradients)
    with pytest.raises(NotImplementedError):
        m(x.to(dev))  # density-only forward has no backward


# ------------------------------------------------------------------ a6 whole path
def _rays(R, seed, hw=100, focal=138.88887889922103):
    gen = torch.Generator().manual_seed(seed)
    pose = O.pose_from_spherical(4.0311289, 50.0, float(torch.rand(1, generator=gen)) * 360.0)
    o, d = O.get_rays(pose, (hw, hw, focal))
    idx = torch.randperm(hw * hw, generator=gen)[:R]
    return o.reshape(-1, 3)[idx].contiguous(), d.reshape(-1, 3)[idx].contiguous(), gen


def _check_render(out, want, what):
    """Per-sample and per-ray outputs; `want` was computed on the same sample set (same edges)."""
    (rgb, op, dep, ex), ri, tv = out
    (wrgb, wop, wdep, wex), wri, wtv = want
    close(ex["edges"], wex["edges"], rtol=0, atol=0, what=what + " edges")
    # the test nets emit raw sigma of both signs (models.py:127), so T can exceed 1: scale the absolute
    # floor by the largest weight
    wmax = max(float(wex["weights"].abs().max()), 1.0)
    close(ex["weights"].reshape(wex["weights"].shape), wex["weights"], atol=1e-5 * wmax, what=what + " weights")
    close(rgb, wrgb, atol=1e-5 * wmax, what=what + " rgb_map")
    close(op, wop, atol=1e-5 * wmax, what=what + " opacity")
    # depth = sum(w t)/max(sum w, eps): compare where it is well conditioned
    cond = wex["weights"].abs().sum(-1, keepdim=True) / wop.abs().clamp(min=1e-12)
    # depth = sum(w t) / max(sum w, eps): the absolute tolerance scales with the ray's condition number
    # sum|w| / |sum w| (raw sigma of both signs gives mixed-sign weights); rays of ~zero opacity divide by ~eps
    dep_ok = (wop > 1e-3).squeeze(-1)
    assert dep_ok.float().mean().item() > 0.95, "test scene: nearly every ray must carry opacity"
    close(dep.cpu()[dep_ok], wdep[dep_ok], atol=(1e-4 * cond[dep_ok].clamp(min=1.0)).numpy(), what=what + " depth_map")
    assert torch.equal(ri.cpu(), wri)
    close(tv, wtv, rtol=0, atol=0, what=what + " t_vals")


@pytest.mark.parametrize("tag,R,S,white,jit", [("4x128", 4096, 64, False, "ray"), ("4x128", 100, 64, True, "none"),
                                               ("8x256", 257, 64, True, "edge"), ("8x256", 64, 100, False, "ray")])
def test_render_rays_coarse_only(dev, golden_dir, tag, R, S, white, jit):
    from fs_nerf_amd.render import rendering as Rm
    _, sd = load_sd(golden_dir, tag)
    m = make_model(sd, *DIMS[tag], [4], dev)
    o, d, gen = _rays(R, 42)
    u = None if jit == "none" else (torch.rand(R, generator=gen) if jit == "ray" else torch.rand(R, S + 1, generator=gen))
    # float32 oracle = "the reference PyTorch CPU path": sample positions o + d*(t0+t1)/2 are formed in
    # float32 by both sides with the same op sequence (their low bits matter: the encoder multiplies by 2^9)
    est = Rm.StratifiedEstimator(2.0, 6.0, S, 0)
    out = Rm.render_rays(o, d, est, m, train=False, white_bkgd=white, device=dev, u=None if u is None else u.to(dev))
    want = O.render_rays_oracle(o, d, sd, None, CFG[tag], near=2.0, far=6.0, n_samples=S, u=u, white_bkgd=white)
    close(out[0][3]["edges"], want[0][3]["edges"], rtol=0, atol=5e-7, what="stratified edges")  # <= 1 ulp
    want = O.render_rays_oracle(o, d, sd, None, CFG[tag], near=2.0, far=6.0, n_samples=S, u=u, white_bkgd=white,
                                edges_override=out[0][3]["edges"].cpu())
    _check_render(out, want, f"{tag} S={S}")
    assert out[0][3]["sigmas"].shape == (R * S,) and out[0][3]["rgbs"].shape == (R * S, 3)


@pytest.mark.parametrize("tag,R,S,NI,two_nets", [("8x256", 130, 64, 128, True), ("4x128", 1001, 64, 128, False),
                                                 ("4x128", 33, 128, 256, True)])
def test_render_rays_hierarchical(dev, golden_dir, tag, R, S, NI, two_nets):
    """End to end (coarse pass -> weights -> resampling -> fine pass -> integration in one launch, nothing substituted)
    against the float64 truth with the float32 oracle as yardstick (tests/test_parity_fp64.py: error <= 2 x the
    float32 oracle's, inside 1e-4 wherever the float32 oracle is), plus the intermediate results the path exposes."""
    from fs_nerf_amd.render import rendering as Rm
    from test_parity_fp64 import assert_parity, oracle as run_oracle
    _, sd = load_sd(golden_dir, tag)
    sd_f = None
    if two_nets:
        sd_f = O.init_nerf_state_dict(DIMS[tag][0], DIMS[tag][1], [4], 10, 4, seed=43)
        sd_f["sigma.weight"] = sd_f["sigma.weight"] * 64.0
        sd_f["sigma.bias"] = sd_f["sigma.bias"] + 3.0
    mc = make_model(sd, *DIMS[tag], [4], dev)
    mf = make_model(sd_f, *DIMS[tag], [4], dev) if two_nets else None
    o, d, gen = _rays(R, 7)
    u = torch.rand(R, generator=gen)
    uf = torch.rand(R, NI, generator=gen)
    kw = dict(near=2.0, far=6.0, n_samples=S, n_importance=NI, u=u, u_fine=uf, white_bkgd=True)
    truth = run_oracle(o, d, sd, sd_f, CFG[tag], torch.float64, **kw)
    want = run_oracle(o, d, sd, sd_f, CFG[tag], torch.float32, **kw)
    est = Rm.StratifiedEstimator(2.0, 6.0, S, NI)
    out = Rm.render_rays(o, d, est, mc, train=False, white_bkgd=True, device=dev, model_fine=mf, u=u.to(dev),
                         u_fine=uf.to(dev))
    assert_parity(out, want, truth, f"{tag} {S}+{NI}")
    # stage 1 on its own: coarse weights and the resampled interval edges, same yardstick
    wt = truth[0][3]["weights_coarse"]
    e_hip = (out[0][3]["weights_coarse"].cpu().double() - wt).abs()
    e_o32 = (want[0][3]["weights_coarse"].double() - wt).abs()
    assert float(e_hip.max()) <= 2.0 * float(e_o32.max()) + 3e-7, (float(e_hip.max()), float(e_o32.max()))
    e, et = out[0][3]["edges"].cpu().double(), truth[0][3]["edges"]
    assert bool((e[:, 1:] >= e[:, :-1]).all())
    ee_hip, ee_o32 = (e - et).abs(), (want[0][3]["edges"].double() - et).abs()
    assert float(ee_hip.max()) <= 2.0 * float(ee_o32.max()) + 3e-6, (float(ee_hip.max()), float(ee_o32.max()))
    assert float(np.percentile(ee_hip.numpy(), 99)) <= 2.0 * float(np.percentile(ee_o32.numpy(), 99)) + 1e-6
    assert torch.equal(out[1].cpu(), want[1])


def test_render_rays_generic_model_matches_fused(dev, golden_dir):
    """Any callable model goes through the unfused HIP sampler/compositor kernels (the reference's
    closure structure); it must agree with the fused launch."""
    from fs_nerf_amd.render import rendering as Rm
    _, sd = load_sd(golden_dir, "4x128")
    m = make_model(sd, 4, 128, [4], dev)

    class Wrapped(torch.nn.Module):  # not a fs_nerf_amd NeRF -> generic path
        def forward(self, x, dirs=None):
            return m(x, dirs)

    o, d, gen = _rays(300, 3)
    u, uf = torch.rand(300, generator=gen).to(dev), torch.rand(300, 128, generator=gen).to(dev)
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 128)
    with torch.no_grad():
        a = Rm.render_rays(o, d, est, m, device=dev, u=u, u_fine=uf)
        b = Rm.render_rays(o, d, est, Wrapped().eval(), device=dev, u=u, u_fine=uf)
    close(a[0][0], b[0][0], atol=1e-5, what="rgb fused vs generic")
    close(a[0][2], b[0][2], atol=1e-4, what="depth fused vs generic")
    close(a[2], b[2], rtol=1e-5, atol=1e-5, what="t_vals fused vs generic")
    assert torch.equal(a[1], b[1])


def test_fused_vs_oracle_random_shapes(dev, golden_dir):
    """Seeded sweep against the float32 oracle: ragged sample counts (1..200 per ray, so tiles are partly filled and
    ray groups vary from 1 to 4 rays), 1..300 rays, every jitter mode, both backgrounds, with and without resampling
    (fine pass checked on the kernel's own sample set, see test_render_rays_hierarchical)."""
    from fs_nerf_amd.render import rendering as Rm
    _, sd = load_sd(golden_dir, "4x128")
    m = make_model(sd, *DIMS["4x128"], [4], dev)
    gen = torch.Generator().manual_seed(4321)
    for case in range(16):
        S = int(torch.randint(1, 201, (1,), generator=gen))
        NI = int(torch.randint(1, 129, (1,), generator=gen)) if case % 4 == 3 else 0
        n2 = 1
        while n2 < NI:
            n2 <<= 1
        if S + n2 > 384:
            S = 384 - n2
        R = int(torch.randint(1, 301, (1,), generator=gen))
        white = bool(case & 1)
        jit = ("none", "ray", "edge")[case % 3]
        o, d, g2 = _rays(R, 500 + case)
        u = None if jit == "none" else (torch.rand(R, generator=g2) if jit == "ray" else torch.rand(R, S + 1, generator=g2))
        uf = torch.rand(R, NI, generator=g2) if NI else None
        est = Rm.StratifiedEstimator(2.0, 6.0, S, NI)
        out = Rm.render_rays(o, d, est, m, white_bkgd=white, device=dev, u=None if u is None else u.to(dev),
                             u_fine=None if uf is None else uf.to(dev))
        want = O.render_rays_oracle(o, d, sd, None, CFG["4x128"], near=2.0, far=6.0, n_samples=S, n_importance=NI, u=u,
                                    u_fine=uf, white_bkgd=white, edges_override=out[0][3]["edges"].cpu())
        _check_render(out, want, f"case {case}: S={S} NI={NI} R={R} {jit}")


def test_fused_vs_generic_random_shapes(dev, golden_dir):
    """Seeded sweep over ragged sampler shapes (sample counts that do not fill the 128-sample tiles, one to a few
    hundred rays, with / without resampling and jitter): fused launch == unfused kernel sequence.  Where the two
    paths' edges agree (the inverse CDF is ill-conditioned in flat regions) the outputs must agree tightly."""
    from fs_nerf_amd.render import rendering as Rm
    _, sd = load_sd(golden_dir, "4x128")
    m = make_model(sd, 4, 128, [4], dev)

    class Wrapped(torch.nn.Module):
        def forward(self, x, dirs=None):
            return m(x, dirs)

    gen = torch.Generator().manual_seed(1234)
    w = Wrapped().eval()
    n_checked = 0
    for case in range(24):
        S = int(torch.randint(1, 129, (1,), generator=gen))
        NI = 0 if case % 3 == 0 else int(torch.randint(1, 201, (1,), generator=gen))
        n2 = 1
        while n2 < NI:
            n2 <<= 1
        if S + n2 > 384:
            NI = 0
        R = int(torch.randint(1, 260, (1,), generator=gen))
        o, d, g2 = _rays(R, 100 + case)
        jitter = case % 2 == 0
        u = torch.rand(R, generator=g2).to(dev) if jitter else None
        uf = torch.rand(R, NI, generator=g2).to(dev) if (jitter and NI) else None
        est = Rm.StratifiedEstimator(2.0, 6.0, S, NI)
        with torch.no_grad():
            a = Rm.render_rays(o, d, est, m, white_bkgd=bool(case & 4), device=dev, u=u, u_fine=uf)
            b = Rm.render_rays(o, d, est, w, white_bkgd=bool(case & 4), device=dev, u=u, u_fine=uf)
        assert a[1].shape == b[1].shape == (R * (S + NI),) and torch.equal(a[1], b[1]), (S, NI, R)
        ea, eb = a[0][3]["edges"], b[2].reshape(R, S + NI)
        same = ((a[2].reshape(R, S + NI) - eb).abs().amax(dim=1) < 1e-5)
        assert float(same.float().mean()) > 0.9, (S, NI, R, float(same.float().mean()))
        close(a[0][0][same], b[0][0][same], atol=2e-5, what=f"rgb S={S} NI={NI} R={R}")
        close(a[0][1][same], b[0][1][same], atol=2e-5, what=f"opacity S={S} NI={NI} R={R}")
        assert bool(torch.isfinite(a[0][0]).all()) and bool(torch.isfinite(a[0][2]).all())
        n_checked += int(same.sum())
    assert n_checked > 1000


def test_render_frame_and_properties(dev, golden_dir):
    """Full-size properties: an 800x800 frame (BASELINE config 3 geometry, 64+128 samples) — finite,
    depth clamped to [near, far], opacity = sum of weights, deterministic, chunking-invariant."""
    from fs_nerf_amd.render import rendering as Rm
    _, sd = load_sd(golden_dir, "4x128")
    m = make_model(sd, 4, 128, [4], dev)
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 128)
    pose = O.pose_from_spherical(4.0311289, 50.0, 30.0)
    hwf = (800, 800, 1111.111)
    with torch.no_grad():
        img, depth = Rm.render_frame(hwf, 2.0, 6.0, pose, 320000, est, m, white_bkgd=True, device=dev)
        img2, depth2 = Rm.render_frame(hwf, 2.0, 6.0, pose, 100001, est, m, white_bkgd=True, device=dev)
    assert img.shape == (800, 800, 3) and depth.shape == (800, 800)
    assert bool(torch.isfinite(img).all()) and bool(torch.isfinite(depth).all())
    assert torch.equal(img, img2) and torch.equal(depth, depth2), "chunking must not change results"
    assert float(depth.min()) >= 2.0 and float(depth.max()) <= 6.0
    o, d = Rm.U.get_rays(pose, hwf, dev)
    out = Rm.render_rays(o.reshape(-1, 3)[:5000], d.reshape(-1, 3)[:5000], est, m, white_bkgd=True, device=dev)
    (rgb, op, _, ex), _, _ = out
    close(ex["weights"].reshape(5000, 192).sum(-1, keepdim=True), op, atol=1e-5, what="opacity = sum weights")
    assert torch.equal(rgb, img.reshape(-1, 3)[:5000])


# ------------------------------------------------------------------ BASELINE.json configs 2, 4, 5 (shapes)
def test_config2_freq_mask_on_coarse_only(dev, golden_dir):
    """configs[1]: 400x400-style rays, 64 coarse samples, 8x256 MLP, frequency mask on (ratio 0.5)."""
    from fs_nerf_amd.render import rendering as Rm
    _, sd = load_sd(golden_dir, "8x256")
    m = make_model(sd, 8, 256, [4], dev)
    pm, dm = O.freq_mask(3, 10, 0.5), O.freq_mask(3, 4, 0.5)
    m.set_freq_mask(pm, dm)
    o, d, gen = _rays(300, 11, hw=400, focal=555.5555)
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 0)
    out = Rm.render_rays(o, d, est, m, white_bkgd=False, device=dev)
    want = O.render_rays_oracle(o, d, sd, None, CFG["8x256"], near=2.0, far=6.0, n_samples=64, white_bkgd=False,
                                pos_mask=pm, dir_mask=dm, edges_override=out[0][3]["edges"].cpu())
    _check_render(out, want, "config 2 (mask on)")
    m.set_freq_mask(None, None)
    out2 = Rm.render_rays(o, d, est, m, white_bkgd=False, device=dev)
    assert float((out2[0][0] - out[0][0]).abs().max()) > 1e-3, "the mask must change the image"


def test_config4_ndc_forward_facing(dev, golden_dir):
    """configs[3] geometry (inference part): LLFF-style forward-facing rays through to_ndc(near=1), near 0 / far 1,
    64+128 samples (llff.py:51-53,75-76).  NDC directions are not unit length."""
    from fs_nerf_amd.render import rendering as Rm
    from fs_nerf_amd.utils import utilities as U
    _, sd = load_sd(golden_dir, "8x256")
    m = make_model(sd, 8, 256, [4], dev)
    hwf = (378, 504, 407.6)
    pose = torch.eye(4)
    pose[:3, 3] = torch.tensor([0.1, -0.05, 0.0])
    o, d = U.get_rays(pose, hwf, dev)
    no, nd = U.to_ndc(o.reshape(-1, 3), d.reshape(-1, 3), hwf, 1.0)
    idx = torch.randperm(378 * 504, generator=torch.Generator().manual_seed(0))[:200].to(dev)
    no, nd = no[idx].contiguous(), nd[idx].contiguous()
    est = Rm.StratifiedEstimator(0.0, 1.0, 64, 128)
    out = Rm.render_rays(no, nd, est, m, white_bkgd=True, device=dev)
    want = O.render_rays_oracle(no.cpu(), nd.cpu(), sd, None, CFG["8x256"], near=0.0, far=1.0, n_samples=64,
                                n_importance=128, white_bkgd=True, edges_override=out[0][3]["edges"].cpu())
    _check_render(out, want, "config 4 (ndc)")


# (BASELINE configs[4], bf16 / fp16 single pass at 128+256: tests/test_parity_fp64.py, against the rounding-emulating oracle)
