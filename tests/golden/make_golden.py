#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by IMPORTING the reference.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py

What is imported from /root/reference/src and how:
  * core.models      - as is (PositionalEncoder, NeRF)
  * core.scheduler   - as is (ExponentialDecay)
  * core.loss        - as is (OcclusionRegularizer)
  * utils.utilities  - get_rays / to_ndc / get_chunks.  Its top-of-file
    `from nerfacc... import ...` lines (utilities.py:6-7) name a package that is not
    installed; none of the three functions uses it, so two EMPTY module objects are
    registered under those names for the duration of the import.  No nerfacc arithmetic
    is emulated; everything that depends on nerfacc stays "parity unpinned".
Outputs are data only (inputs + the reference's outputs); no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.path.insert(0, REF)
    import core.models as M
    import core.scheduler as S
    import core.loss as L
    placeholders = {}
    for name, attrs in (("nerfacc", ()), ("nerfacc.volrend", ("rendering",)),
                        ("nerfacc.estimators", ()), ("nerfacc.estimators.occ_grid", ("OccGridEstimator",))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        placeholders[name] = m
    sys.modules.update(placeholders)
    try:
        import utils.utilities as U
    finally:
        for name in placeholders:
            sys.modules.pop(name, None)
    return M, S, L, U


def orbit_pose(radius, theta_deg, phi_deg):
    # same construction as blender.py:20-69, written independently (see oracle.pose_from_spherical)
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle.fsnerf_oracle import pose_from_spherical
    return pose_from_spherical(radius, theta_deg, phi_deg)


def main():
    M, S, L, U = _import_reference()
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1234)

    # ---- G1 / G2: get_rays, to_ndc --------------------------------------------------
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    rand_pose = torch.eye(4)
    rand_pose[:3, :3] = q
    rand_pose[:3, 3] = torch.tensor([0.3, -1.2, 2.5])
    poses = {"identity": torch.eye(4), "orbit0": orbit_pose(4.0311289, 50.0, 0.0),
             "orbit3x4": orbit_pose(4.0311289, 50.0, 123.0)[:3, :4].clone(), "random": rand_pose}
    out = {}
    for pname, pose in poses.items():
        for hname, hwf in (("small", (4, 6, 5.0)), ("lego100", (100, 100, 138.88887889922103))):
            o, d = U.get_rays(pose, hwf)
            key = f"{pname}_{hname}"
            out[key + "_pose"] = pose.numpy()
            out[key + "_hwf"] = np.array(hwf, dtype=np.float64)
            out[key + "_o"] = o.contiguous().numpy()
            out[key + "_d"] = d.contiguous().numpy()
            no, nd = U.to_ndc(o.reshape(-1, 3), d.reshape(-1, 3), hwf, 1.0)
            out[key + "_ndc_o"] = no.numpy()
            out[key + "_ndc_d"] = nd.numpy()
    ch = U.get_chunks(torch.arange(10 * 3, dtype=torch.float32).reshape(10, 3), 4)
    out["chunks_lens"] = np.array([c.shape[0] for c in ch])
    np.savez_compressed(os.path.join(HERE, "g1_rays.npz"), **out)

    # ---- G3: positional encoder -----------------------------------------------------
    x = (torch.rand(256, 3, generator=g) * 3.0 - 1.5)
    out = {"x": x.numpy()}
    for n in (10, 4):
        for ls in (True, False):
            enc = M.PositionalEncoder(3, n, ls)
            out[f"pe_n{n}_log{int(ls)}"] = enc(x).numpy()
            out[f"pe_n{n}_log{int(ls)}_dout"] = np.array(enc.d_output)
    np.savez_compressed(os.path.join(HERE, "g3_posenc.npz"), **out)

    # ---- G4: NeRF MLP -----------------------------------------------------------------
    dirs = torch.randn(256, 3, generator=g)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    for tag, n_layers, d_hidden, skip in (("8x256", 8, 256, [4]), ("4x128", 4, 128, [4])):
        torch.manual_seed(42)
        kw = {"pos_fn": {"n_freqs": 10, "log_space": True}, "dir_fn": {"n_freqs": 4, "log_space": True}}
        net = M.NeRF(3, 3, n_layers, d_hidden, skip, **kw)
        with torch.no_grad():
            y1 = net(x)
            y4 = net(x, dirs)
            # a second parameter set with a large sigma head so that alpha is not ~0
            sd2 = {k: v.clone() for k, v in net.state_dict().items()}
            sd2["sigma.weight"] *= 64.0
            sd2["sigma.bias"] += 1.0
            net.load_state_dict(sd2)
            y4b = net(x, dirs)
        out = {"x": x.numpy(), "dirs": dirs.numpy(), "y_sigma": y1.numpy(), "y_full": y4.numpy(),
               "y_full_sigma64": y4b.numpy(),
               "cfg": np.array([n_layers, d_hidden, 10, 4], dtype=np.int64), "skip": np.array(skip)}
        torch.manual_seed(42)
        net = M.NeRF(3, 3, n_layers, d_hidden, skip, **kw)
        for k, v in net.state_dict().items():
            out["sd." + k] = v.numpy()
        np.savez_compressed(os.path.join(HERE, f"g4_nerf_{tag}.npz"), **out)

    # ---- G5: "next" rows (scheduler, occlusion regulariser) ---------------------------
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=5e-4)
    sch = S.ExponentialDecay(opt, 8000, 5e-4, r=0.1)
    lrs, ts = [], [0, 1, 4000, 7999, 8000, 9000]
    for t in ts:
        sch.t = t
        lrs.append(sch.lr)
    ri = torch.tensor([0, 0, 0, 2, 2, 5, 5, 5, 5], dtype=torch.int64)
    sig = torch.rand(9, generator=g)
    tv = torch.rand(9, generator=g) * 4 + 2
    occ_lin = L.OcclusionRegularizer(0.5, 3.0, "linear")(sig, tv, ri)
    occ_exp = L.OcclusionRegularizer(0.5, 3.0, "exp")(sig, tv, ri)
    np.savez_compressed(os.path.join(HERE, "g5_next.npz"), sched_t=np.array(ts), sched_lr=np.array(lrs),
                        occ_ray_idx=ri.numpy(), occ_sigmas=sig.numpy(), occ_t=tv.numpy(),
                        occ_linear=occ_lin.numpy(), occ_exp=occ_exp.numpy())
    print("golden vectors written to", HERE)




def make_g6_video():
    """render_video (src/render/rendering.py:240-266).  `render.rendering` itself cannot be imported here (imageio and
    nerfacc are missing), so its body - six matplotlib / numpy calls - is evaluated with the same calls:
    Normalize over all frames, ScalarMappable(cmap).to_rgba on the flattened depths, to8b, NHWC -> NCHW."""
    import matplotlib
    from matplotlib import cm
    rng = np.random.default_rng(6)
    frames = rng.uniform(-0.2, 1.2, size=(2, 5, 7, 3)).astype(np.float32)   # values outside [0,1] exercise the clip
    d_frames = rng.uniform(2.0, 6.0, size=(2, 5, 7)).astype(np.float32)
    d_frames[0, 0, 0], d_frames[1, 4, 6] = 2.0, 6.0
    to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)
    out = {"frames": frames, "d_frames": d_frames}
    for cmap in ("plasma", "viridis"):
        norm = matplotlib.colors.Normalize(vmin=np.amin(d_frames), vmax=np.amax(d_frames))
        mapper = cm.ScalarMappable(norm=norm, cmap=cmap)
        d_rgba = mapper.to_rgba(d_frames.flatten())
        d_rgba = np.reshape(d_rgba, list(d_frames.shape[:3]) + [-1])
        out[f"{cmap}_frames8"] = np.transpose(to8b(frames), (0, 3, 1, 2))
        out[f"{cmap}_depth8"] = np.transpose(to8b(d_rgba[..., :3]), (0, 3, 1, 2))
    # a constant depth map: vmin == vmax -> everything maps to the first entry
    flat = np.full((1, 3, 4), 3.5, np.float32)
    norm = matplotlib.colors.Normalize(vmin=np.amin(flat), vmax=np.amax(flat))
    out["flat_d"] = flat
    out["flat_depth8"] = np.transpose(to8b(np.reshape(cm.ScalarMappable(norm=norm, cmap="plasma").to_rgba(flat.flatten()),
                                                      [1, 3, 4, -1])[..., :3]), (0, 3, 1, 2))
    np.savez_compressed(os.path.join(HERE, "g6_video.npz"), **out)


if __name__ == "__main__":
    if "--video" in sys.argv:  # only the video fixture (needs matplotlib, not the reference)
        make_g6_video()
    else:
        main()
        make_g6_video()
