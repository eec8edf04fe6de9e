"""CPU (no GPU): the packed-weights blob and the kernel's register dataflow.

`fsn_mlp_pack_host` (the same index code as the device packer) packs a reference-format
state_dict; a NumPy model of v_mfma_f32_16x16x32_{f16,bf16}'s documented lane layouts
(cdna_hip_programming.md section 3) then replays exactly what fs-nerf_amd/csrc/mlp_dev.hpp does for one
wavefront of 16 samples — encoding slots, unit order, accumulator-as-B-operand chaining,
fp32 heads — and the result must equal the oracle's NeRF forward.  This pins the blob
format and the chaining rule without a GPU."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import fs_nerf_amd
from fs_nerf_amd import _lib as L
from fs_nerf_amd import ops
from oracle import fsnerf_oracle as O


def bf16_to_f32(u16: np.ndarray) -> np.ndarray:
    return (u16.astype(np.uint32) << 16).view(np.float32)


def pack_host(sd, n_layers, d_hidden, skip, nf, nfd, prec, exps=None):
    fp = O.pe_freqs(nf, True).tolist()
    fd = O.pe_freqs(nfd, True).tolist()
    desc = ops.make_desc(n_layers, d_hidden, skip, fp, fd)
    nbytes = L.lib().fsn_mlp_blob_bytes(C.byref(desc), prec)
    assert nbytes > 0
    ws, bs = ops.sd_tensor_lists(sd, n_layers)
    ws = [np.ascontiguousarray(w.numpy(), dtype=np.float32) for w in ws]
    bs = [np.ascontiguousarray(b.numpy(), dtype=np.float32) for b in bs]
    n = n_layers + 4
    Wp = (C.c_void_p * n)(*[w.ctypes.data for w in ws])
    Bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
    blob = np.zeros(nbytes, dtype=np.uint8)
    if exps is None:
        L.check(L.lib().fsn_mlp_pack_host(C.byref(desc), prec, Wp, Bp, blob.ctypes.data), "fsn_mlp_pack_host")
    else:
        ex = (C.c_int32 * (n_layers + 2))(*exps)
        L.check(L.lib().fsn_mlp_pack_scaled_host(C.byref(desc), prec, Wp, Bp, ex, blob.ctypes.data), "fsn_mlp_pack_scaled_host")
    return blob


class Emu:
    """One wavefront (64 lanes: sample c = lane & 15, lane group g = lane >> 4) of mlp_dev.hpp."""

    def __init__(self, blob, prec):
        hw = blob[:256].view(np.uint32)
        assert hw[0] == 0x4E53460A
        assert hw[1] == 3, "blob layout version 3: low parts scaled by lo_scale(prec), per-layer exponents at words 16.."
        self.prec, self.L, self.D = int(hw[2]), int(hw[3]), int(hw[4])
        self.skip_mask, self.nf, self.nfd = int(hw[5]), int(hw[6]), int(hw[7])
        self.units_total, self.nph_full, self.nph_density = int(hw[8]), int(hw[9]), int(hw[10])
        aux_off, aux_floats, stream_off = int(hw[11]), int(hw[12]), int(hw[13])
        self.aux = blob[aux_off:aux_off + 4 * aux_floats].view(np.float32)
        self.stream = blob[stream_off:]
        self.ub = 2048 if prec in (0, 2, 4) else 1024
        self.exps = [int(np.int32(v)) for v in hw[16:16 + self.L + 2]]
        self.unit = 0
        self.NT = self.D // 32

    def a_frag(self):
        """A operand of the next unit as float64 [lane, j] (hi + lo)."""
        base = self.unit * self.ub
        dec = (lambda b: b.view(np.float16).astype(np.float32)) if self.prec >= 2 else \
            (lambda b: bf16_to_f32(b.view(np.uint16)))
        hi = dec(self.stream[base:base + 1024]).reshape(64, 8).astype(np.float64)
        if self.prec in (0, 2, 4):
            # FSN_PREC_FP16X3: fp16 low parts are stored as fp16((w - hi) * 2^11); FSN_PREC_FP16X3U (4): unscaled
            lo_scale = 2048.0 if self.prec == 2 else 1.0
            hi = hi + dec(self.stream[base + 1024:base + 2048]).reshape(64, 8).astype(np.float64) / lo_scale
        self.unit += 1
        return hi

    @staticmethod
    def mfma(afrag, bfrag, acc):
        """acc [4 regs, 64 lanes] += A.B with the 16x16x32 lane maps (cdna_hip_programming.md section 3):
        A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], D col = l&15, row = 4(l>>4)+reg."""
        A = np.zeros((16, 32))
        B = np.zeros((32, 16))
        for lane in range(64):
            r, g = lane & 15, lane >> 4
            A[r, 8 * g:8 * g + 8] = afrag[lane]
            B[8 * g:8 * g + 8, r] = bfrag[lane]
        Dm = A @ B
        for lane in range(64):
            col, g = lane & 15, lane >> 4
            for reg in range(4):
                acc[reg, lane] += Dm[4 * g + reg, col]
        return acc

    def encode(self, xyz, n_freqs, freqs, mask, nks):
        slots = 8 * nks
        v = np.zeros((64, slots))
        P = 3 * n_freqs
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            x = xyz[c]
            for i in range(slots // 2):
                p = 4 * i + g
                if p < P:
                    band, coord = divmod(p, 3)
                    a = np.float32(x[coord]) * np.float32(freqs[band])
                    v[lane, 2 * i] = np.sin(np.float64(a)) * mask[3 + band * 6 + coord]
                    v[lane, 2 * i + 1] = np.cos(np.float64(a)) * mask[3 + band * 6 + 3 + coord]
            if g == 2:
                v[lane, slots - 2], v[lane, slots - 1] = x[0] * mask[0], x[1] * mask[1]
            if g == 3:
                v[lane, slots - 2] = x[2] * mask[2]
        return [v[:, 8 * k:8 * k + 8] for k in range(nks)]

    def layer(self, np_out, act, enc, bias_off, relu):
        """-> (B operands of the next layer, one per output pair; the pairs' 8 fp32 values per lane)"""
        outs, vals = [], []
        for tp in range(np_out):
            acc = [np.zeros((4, 64)), np.zeros((4, 64))]
            for lane in range(64):
                g = lane >> 4
                for reg in range(4):
                    acc[0][reg, lane] = self.aux[bias_off + 32 * tp + 4 * g + reg]
                    acc[1][reg, lane] = self.aux[bias_off + 32 * tp + 16 + 4 * g + reg]
            for b in list(act) + list(enc):
                for sub in range(2):
                    acc[sub] = self.mfma(self.a_frag(), b, acc[sub])
            v = np.concatenate([acc[0], acc[1]], axis=0)  # element j = register j&3 of tile j>>2
            if relu:
                v = np.maximum(v, 0.0)
            vals.append(v)
            outs.append(v.T.copy())
        return outs, vals

    def head(self, vals, w_off):
        tot = np.zeros(64)
        for tp, v in enumerate(vals):
            for lane in range(64):
                g = lane >> 4
                for j in range(8):
                    tot[lane] += self.aux[w_off + 32 * tp + 16 * (j >> 2) + 4 * g + (j & 3)] * v[j, lane]
        return tot[:16] + tot[16:32] + tot[32:48] + tot[48:]

    def forward(self, x, dirs, pos_mask, dir_mask):
        L_, D = self.L, self.D
        misc = (L_ + 5) * D
        self.unit = 0
        pe = self.encode(x, self.nf, self.aux[misc + 4:misc + 20], pos_mask, 2)
        act, _ = self.layer(self.NT, [], pe, 0, True)
        for l in range(1, L_):
            wide = (self.skip_mask >> (l - 1)) & 1
            act, vals = self.layer(self.NT, act, pe if wide else [], l * D, True)
        assert self.unit * self.ub == self.nph_density * 16384
        sigma = self.head(vals, (L_ + 2) * D) + self.aux[misc]
        if dirs is None:
            return sigma[:, None]
        act, _ = self.layer(self.NT, act, [], L_ * D, False)
        de = self.encode(dirs, self.nfd, self.aux[misc + 20:misc + 36], dir_mask, 1)
        _, vals = self.layer(self.NT // 2, act, de, (L_ + 1) * D, True)
        assert self.unit == self.units_total
        rgb = [1.0 / (1.0 + np.exp(-(self.head(vals, (L_ + 3) * D + c * (D // 2)) + self.aux[misc + 1 + c])))
               for c in range(3)]
        return np.stack(rgb + [sigma], axis=-1)


def _load_sd(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"g4_nerf_{tag}.npz"))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    return g, sd


@pytest.mark.parametrize("tag,n_layers,d_hidden", [("4x128", 4, 128), ("8x256", 8, 256)])
@pytest.mark.parametrize("prec", [0, 1, 2, 3, 4])
def test_blob_replays_to_reference_forward(golden_dir, tag, n_layers, d_hidden, prec):
    g, sd = _load_sd(golden_dir, tag)
    blob = pack_host(sd, n_layers, d_hidden, [4], 10, 4, prec)
    emu = Emu(blob, prec)
    assert emu.L == n_layers and emu.D == d_hidden
    assert emu.skip_mask == (0b10000 if n_layers > 5 else 0)
    x, d = g["x"][:16], g["dirs"][:16]
    ones_p, ones_d = np.ones(64), np.ones(32)
    y = emu.forward(x, d, ones_p, ones_d)
    ref = g["y_full"][:16]
    # x3 modes: weights are hi+lo (16 / 22 mantissa bits), activations exact in this model;
    # single-pass modes: 8 / 11-bit weights
    tol = {0: 2e-5, 1: 2e-2, 2: 2e-6, 3: 3e-3, 4: 2e-6}[prec]
    np.testing.assert_allclose(y[:, :3], ref[:, :3], rtol=0, atol=tol)
    np.testing.assert_allclose(y[:, 3], ref[:, 3], rtol=0, atol=tol)
    ys = emu.forward(x, None, ones_p, ones_d)
    np.testing.assert_allclose(ys[:, 0], g["y_sigma"][:16, 0], rtol=0, atol=tol)


@pytest.mark.parametrize("tag,n_layers,d_hidden", [("4x128", 4, 128), ("8x256", 8, 256)])
@pytest.mark.parametrize("prec", [4, 0])
def test_scaled_blob_is_the_same_network(golden_dir, tag, n_layers, d_hidden, prec):
    """fsn_mlp_pack_scaled (round 4): per-layer powers of two folded into weight columns (activation columns by the
    ratio of neighbouring scales, encoding columns by the layer's own), biases and the two float32 heads.  The replayed
    blob must give the REFERENCE's outputs whatever the exponents, its hidden activations must be 2^e x the plain blob's."""
    g, sd = _load_sd(golden_dir, tag)
    rng = np.random.default_rng(5)
    if prec == 0:  # bf16 has float32's exponent range: any exponents
        exps = [int(e) for e in rng.integers(-6, 12, n_layers + 2)]
    else:          # fp16, unscaled low parts: exponents that scale no weight DOWN (a weight of 0.05 x 2^-10 would be an
        # fp16 subnormal; calibrated scales of a real network keep neighbouring layers within a few octaves)
        exps = [int(e) for e in np.cumsum(rng.integers(0, 3, n_layers + 2)) + 4]
        exps[n_layers] = exps[n_layers - 1] + 1       # connection reads layer L-1, branch the connection
        exps[n_layers + 1] = exps[n_layers] + 2
    blob = pack_host(sd, n_layers, d_hidden, [4], 10, 4, prec, exps)
    emu = Emu(blob, prec)
    assert emu.exps == exps
    x, d = g["x"][:16], g["dirs"][:16]
    ones_p, ones_d = np.ones(64), np.ones(32)
    y = emu.forward(x, d, ones_p, ones_d)
    ref = g["y_full"][:16]
    tol = {0: 2e-5, 4: 2e-6}[prec]
    np.testing.assert_allclose(y[:, :3], ref[:, :3], rtol=0, atol=tol)
    np.testing.assert_allclose(y[:, 3], ref[:, 3], rtol=0, atol=tol)
    ys = emu.forward(x, None, ones_p, ones_d)
    np.testing.assert_allclose(ys[:, 0], g["y_sigma"][:16, 0], rtol=0, atol=tol)
    # layer 0 of the scaled blob = 2^e0 x layer 0 of the plain one (weights on encoding columns, bias)
    plain = Emu(pack_host(sd, n_layers, d_hidden, [4], 10, 4, prec), prec)
    assert plain.exps == [0] * (n_layers + 2)
    for e_, scale in ((emu, 2.0 ** exps[0]), (plain, 1.0)):
        e_.unit = 0
        pe = e_.encode(x, e_.nf, e_.aux[(e_.L + 5) * e_.D + 4:(e_.L + 5) * e_.D + 20], ones_p, 2)
        _, vals = e_.layer(e_.NT, [], pe, 0, True)
        e_.h0 = np.stack(vals) / scale
    np.testing.assert_allclose(emu.h0, plain.h0, rtol=2e-5 if prec == 0 else 2e-6, atol=1e-6)  # (plain fp16 blob: subnormal low parts)
    # exponents outside [-60, 60] are rejected
    with pytest.raises(RuntimeError):
        pack_host(sd, n_layers, d_hidden, [4], 10, 4, prec, [61] + [0] * (n_layers + 1))


@pytest.mark.parametrize("n_layers,d_hidden,skip,nf,nfd", [(6, 128, [1, 3], 7, 3), (2, 128, [0], 10, 4), (12, 256, [2, 5, 9], 10, 4)])
def test_scaled_blob_other_topologies(n_layers, d_hidden, skip, nf, nfd):
    """The scaling's column rules on other network shapes: several skip layers (their encoding columns take the layer's
    own scale, their activation columns the ratio to the layer before), the shallowest network the kernels accept, a deep
    one - scaled blob == the oracle's forward of the UNSCALED network, and == the oracle's own restatement of the
    transformation (oracle.scale_state_dict) packed plainly."""
    sd = O.init_nerf_state_dict(n_layers, d_hidden, skip, nf, nfd, seed=11)
    rng = np.random.default_rng(n_layers)
    exps = [int(e) for e in np.cumsum(rng.integers(0, 3, n_layers + 2)) + 2]
    exps[n_layers] = exps[n_layers - 1] + 1
    exps[n_layers + 1] = exps[n_layers] + 1
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(16, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(16, 3, generator=gen), dim=-1)
    ref = O.nerf_forward({k: v.double() for k, v in sd.items()}, x.double(), d.double(), n_layers=n_layers, skip=skip,
                         n_freqs=nf, n_freqs_dir=nfd).numpy()
    ones_p, ones_d = np.ones(64), np.ones(32)
    emu = Emu(pack_host(sd, n_layers, d_hidden, skip, nf, nfd, 4, exps), 4)
    y = emu.forward(x.numpy(), d.numpy(), ones_p, ones_d)
    np.testing.assert_allclose(y, ref, rtol=0, atol=3e-6)
    sds = O.scale_state_dict(sd, exps, n_layers=n_layers, skip=skip, d_hidden=d_hidden)
    plain = pack_host(sds, n_layers, d_hidden, skip, nf, nfd, 4)
    scaled = pack_host(sd, n_layers, d_hidden, skip, nf, nfd, 4, exps)
    hw_p, hw_s = plain[:256].view(np.uint32).copy(), scaled[:256].view(np.uint32).copy()
    hw_p[16:16 + n_layers + 2] = hw_s[16:16 + n_layers + 2]  # (the header records the exponents; everything else is equal)
    assert np.array_equal(hw_p, hw_s) and np.array_equal(plain[256:], scaled[256:]), "packer's scaling == oracle.scale_state_dict"


def test_blob_frequency_mask_and_wide_variants(golden_dir):
    # skip at two places + a frequency mask: emulator == oracle with the same mask
    sd = O.init_nerf_state_dict(6, 128, [1, 3], 7, 3, seed=7)
    blob = pack_host(sd, 6, 128, [1, 3], 7, 3, 0)
    emu = Emu(blob, 0)
    assert emu.skip_mask == 0b1010
    gen = torch.Generator().manual_seed(3)
    x = torch.rand(16, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(16, 3, generator=gen), dim=-1)
    pm, dm = O.freq_mask(3, 7, 0.6), O.freq_mask(3, 3, 0.5)
    ref = O.nerf_forward(sd, x.double(), d.double(), n_layers=6, skip=[1, 3], n_freqs=7, n_freqs_dir=3,
                         pos_mask=pm.double(), dir_mask=dm.double()).numpy()
    pmk, dmk = np.ones(64), np.ones(32)
    pmk[:pm.numel()], dmk[:dm.numel()] = pm.numpy(), dm.numpy()
    y = emu.forward(x.numpy(), d.numpy(), pmk, dmk)
    np.testing.assert_allclose(y, ref, rtol=0, atol=3e-5)


def test_pack_rejects_unsupported_shapes():
    with pytest.raises(ValueError):
        ops.make_desc(8, 256, [7], [1.0] * 10, [1.0] * 4)
    for kw in (dict(n_layers=8, d_hidden=192), dict(n_layers=1, d_hidden=256), dict(n_layers=8, d_hidden=256, nf=11)):
        desc = ops.make_desc(kw["n_layers"], kw["d_hidden"], [], [1.0] * kw.get("nf", 10), [1.0] * 4)
        assert L.lib().fsn_mlp_blob_bytes(C.byref(desc), 0) < 0
        assert len(L.lib().fsn_last_error()) > 0
