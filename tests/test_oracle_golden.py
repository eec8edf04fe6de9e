"""CPU: the oracle restatement against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import fsnerf_oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("pname", ["identity", "orbit0", "orbit3x4", "random"])
@pytest.mark.parametrize("hname", ["small", "lego100"])
def test_get_rays_and_ndc(golden_dir, pname, hname):
    g = _load(golden_dir, "g1_rays.npz")
    key = f"{pname}_{hname}"
    hwf = g[key + "_hwf"]
    hwf = (int(hwf[0]), int(hwf[1]), float(hwf[2]))
    o, d = O.get_rays(torch.from_numpy(g[key + "_pose"]), hwf)
    # same op sequence -> bitwise on the same CPU; allow 1 ulp for other hosts
    np.testing.assert_allclose(o.numpy(), g[key + "_o"], rtol=0, atol=0)
    np.testing.assert_allclose(d.numpy(), g[key + "_d"], rtol=2e-7, atol=1e-7)
    no, nd = O.to_ndc(torch.from_numpy(g[key + "_o"]).reshape(-1, 3),
                      torch.from_numpy(g[key + "_d"]).reshape(-1, 3), hwf, 1.0)
    ok = np.isfinite(g[key + "_ndc_o"]).all(-1) & np.isfinite(g[key + "_ndc_d"]).all(-1)
    np.testing.assert_allclose(no.numpy()[ok], g[key + "_ndc_o"][ok], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(nd.numpy()[ok], g[key + "_ndc_d"][ok], rtol=1e-6, atol=1e-6)


def test_get_rays_properties():
    pose = O.pose_from_spherical(4.0311289, 50.0, 40.0)
    o, d = O.get_rays(pose, (8, 10, 7.5))
    assert o.shape == (8, 10, 3) and d.shape == (8, 10, 3)
    np.testing.assert_allclose(d.norm(dim=-1).numpy(), 1.0, rtol=1e-6)
    # centre pixel looks down -z of the camera
    np.testing.assert_allclose(d[4, 5].numpy(), (pose[:3, :3] @ torch.tensor([0.0, 0.0, -1.0])).numpy(), atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(pose[:3, 3].numpy()), 4.0311289, rtol=1e-6)


def test_get_chunks(golden_dir):
    g = _load(golden_dir, "g1_rays.npz")
    ch = O.get_chunks(torch.zeros(10, 3), 4)
    assert [c.shape[0] for c in ch] == list(g["chunks_lens"])


@pytest.mark.parametrize("n", [10, 4])
@pytest.mark.parametrize("ls", [True, False])
def test_posenc(golden_dir, n, ls):
    g = _load(golden_dir, "g3_posenc.npz")
    y = O.posenc(torch.from_numpy(g["x"]), n, ls)
    assert y.shape[1] == int(g[f"pe_n{n}_log{int(ls)}_dout"])
    np.testing.assert_allclose(y.numpy(), g[f"pe_n{n}_log{int(ls)}"], rtol=0, atol=1e-6)
    # mask == 1 is the reference; a zero band removes exactly that band
    m = O.freq_mask(3, n, 1.0)
    assert torch.equal(O.posenc(torch.from_numpy(g["x"]), n, ls, m), y)
    m0 = O.freq_mask(3, n, 0.5)
    ym = O.posenc(torch.from_numpy(g["x"]), n, ls, m0)
    assert torch.equal(ym[:, :3], y[:, :3]) and float(ym[:, -6:].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["8x256", "4x128"])
def test_nerf_forward(golden_dir, tag):
    g = _load(golden_dir, f"g4_nerf_{tag}.npz")
    n_layers, d_hidden, nf, nfd = [int(v) for v in g["cfg"]]
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    kw = dict(n_layers=n_layers, skip=[int(s) for s in g["skip"]], n_freqs=nf, n_freqs_dir=nfd, log_space=True)
    x, d = torch.from_numpy(g["x"]), torch.from_numpy(g["dirs"])
    y1 = O.nerf_forward(sd, x, None, **kw)
    y4 = O.nerf_forward(sd, x, d, **kw)
    np.testing.assert_allclose(y1.numpy(), g["y_sigma"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y4.numpy(), g["y_full"], rtol=1e-5, atol=1e-6)
    sd2 = dict(sd)
    sd2["sigma.weight"] = sd["sigma.weight"] * 64.0
    sd2["sigma.bias"] = sd["sigma.bias"] + 1.0
    np.testing.assert_allclose(O.nerf_forward(sd2, x, d, **kw).numpy(), g["y_full_sigma64"], rtol=1e-5, atol=2e-5)
    # the reference's construction order reproduces its parameters from the seed
    sd_init = O.init_nerf_state_dict(n_layers, d_hidden, [int(s) for s in g["skip"]], nf, nfd, seed=42)
    assert set(sd_init) == set(sd)
    for k in sd:
        assert sd_init[k].shape == sd[k].shape
        assert torch.equal(sd_init[k], sd[k]), k


def test_state_dict_shapes_8x256(golden_dir):
    g = _load(golden_dir, "g4_nerf_8x256.npz")
    shp = {k[3:]: g[k].shape for k in g.files if k.startswith("sd.")}
    assert shp["layers.0.weight"] == (256, 63) and shp["layers.5.weight"] == (256, 319)
    assert shp["branch.weight"] == (128, 283) and shp["rgb.weight"] == (3, 128) and shp["sigma.weight"] == (1, 256)
    assert sum(int(np.prod(s)) for s in shp.values()) == 595844
