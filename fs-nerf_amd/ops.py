"""Tensor-level wrappers over the C-ABI (include/fsnerf_hip.h).  Every function here launches
HIP kernels on the current torch stream; inputs must live on the GPU (float32, contiguous).
No function in this module computes anything on the CPU."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _i64(t: Tensor, name: str) -> Tensor:
    """ray_indices as the kernels read them: int64 (the reference contract, rendering.py:66,92), on the GPU,
    contiguous.  nerfacc-style callers may hand int32 indices: they are widened here, never reinterpreted."""
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8, torch.int8):
        raise TypeError(f"{name}: expected an integer tensor, got {t.dtype}")
    return t.to(torch.int64).contiguous()


def _p(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _bk(bkgd) -> Optional[C.Array]:
    if bkgd is None:
        return None
    vals = [float(v) for v in (bkgd.detach().cpu().tolist() if isinstance(bkgd, Tensor) else bkgd)]
    return (C.c_float * 3)(*vals)


# ------------------------------------------------------------------ rays
def get_rays(pose: Tensor, H: int, W: int, focal: float, device, row0: int = 0,
             nrows: Optional[int] = None, out: Optional[Tuple[Tensor, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """`out`: optional pre-allocated ([nrows*W,3], [nrows*W,3]) contiguous fp32 slices to write into."""
    nrows = H - row0 if nrows is None else nrows
    p = pose.detach().to("cpu", torch.float32)[:3, :4].contiguous()
    pose_host = (C.c_float * 12)(*p.reshape(-1).tolist())
    if out is None:
        o = torch.empty(nrows * W, 3, device=device, dtype=torch.float32)
        d = torch.empty_like(o)
    else:
        o, d = out
        assert o.is_contiguous() and d.is_contiguous() and o.shape == (nrows * W, 3) and d.shape == o.shape
    with torch.cuda.device(o.device):
        L.check(L.lib().fsn_get_rays(pose_host, H, W, float(focal), row0, nrows, _p(o), _p(d), _stream()),
                "fsn_get_rays")
    return o, d


def build_rays(poses: Tensor, H: int, W: int, focal: float, device, ndc: bool = False, near: float = 1.0,
               want_aabb: bool = False):
    """Rays of all `poses` ([n, 3or4, 4]) in ONE launch (fsn_build_rays) -> (rays_o [n*H*W,3], rays_d [n*H*W,3],
    aabb [6] or None): get_rays per pose, optional NDC mapping, and the min / max region of interest over {o, o + d}
    (/ 2^3, llff.py:77-84) reduced inside the launch."""
    P = torch.as_tensor(poses, dtype=torch.float32)[:, :3, :4].reshape(-1, 12).contiguous().to(device)
    n = P.shape[0]
    o = torch.empty(n * H * W, 3, device=P.device, dtype=torch.float32)
    d = torch.empty_like(o)
    aabb = torch.empty(6, device=P.device, dtype=torch.float32) if want_aabb else None
    keys = torch.empty(6, device=P.device, dtype=torch.int32) if want_aabb else None
    with torch.cuda.device(P.device):
        L.check(L.lib().fsn_build_rays(_p(P), n, int(H), int(W), float(focal), 1 if ndc else 0, float(near), _p(o), _p(d),
                                       _p(aabb), _p(keys), _stream()), "fsn_build_rays")
    return o, d, aabb


def to_ndc(rays_o: Tensor, rays_d: Tensor, H: int, W: int, focal: float, near: float) -> Tuple[Tensor, Tensor]:
    o, d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
    n = o.numel() // 3
    no, nd = torch.empty_like(o), torch.empty_like(d)
    with torch.cuda.device(o.device):
        L.check(L.lib().fsn_to_ndc(_p(o), _p(d), n, H, W, float(focal), float(near), _p(no), _p(nd), _stream()),
                "fsn_to_ndc")
    return no, nd


def posenc(x: Tensor, freqs: Sequence[float], mask: Optional[Tensor] = None) -> Tensor:
    x = _f32(x, "x")
    d_in = x.shape[-1]
    n = x.numel() // d_in
    nf = len(freqs)
    out = torch.empty(*x.shape[:-1], d_in * (1 + 2 * nf), device=x.device, dtype=torch.float32)
    fr = (C.c_float * max(nf, 1))(*[float(f) for f in freqs])
    m = None if mask is None else _f32(mask, "mask")
    with torch.cuda.device(x.device):
        L.check(L.lib().fsn_posenc_fwd(_p(x), n, d_in, nf, fr, _p(m), _p(out), _stream()), "fsn_posenc_fwd")
    return out


# ------------------------------------------------------------------ sampling
def _u_mode(u: Optional[Tensor], R: int, S: int) -> Tuple[int, Optional[Tensor]]:
    if u is None:
        return 0, None
    u = _f32(u, "u")
    if u.numel() == R:
        return 1, u
    if tuple(u.shape) == (R, S + 1):
        return 2, u
    raise ValueError(f"jitter u must have {R} or {R}x{S + 1} elements, got {tuple(u.shape)}")


def stratified_edges(near: float, far: float, S: int, R: int, u: Optional[Tensor], device) -> Tensor:
    mode, u = _u_mode(u, R, S)
    edges = torch.empty(R, S + 1, device=device, dtype=torch.float32)
    with torch.cuda.device(edges.device):
        L.check(L.lib().fsn_stratified_edges(float(near), float(far), S, R, _p(u), mode, _p(edges), _stream()),
                "fsn_stratified_edges")
    return edges


def edges_to_packed(edges: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    edges = _f32(edges, "edges")
    R, S = edges.shape[0], edges.shape[1] - 1
    ri = torch.empty(R * S, device=edges.device, dtype=torch.int64)
    t0 = torch.empty(R * S, device=edges.device, dtype=torch.float32)
    t1 = torch.empty_like(t0)
    with torch.cuda.device(edges.device):
        L.check(L.lib().fsn_edges_to_packed(_p(edges), R, S, _p(ri), _p(t0), _p(t1), _stream()),
                "fsn_edges_to_packed")
    return ri, t0, t1


def sample_pdf_merge(edges: Tensor, weights: Tensor, n_imp: int, u: Optional[Tensor] = None) -> Tensor:
    edges, weights = _f32(edges, "edges"), _f32(weights, "weights")
    R, S = weights.shape
    assert edges.shape == (R, S + 1)
    u = None if u is None else _f32(u, "u")
    out = torch.empty(R, S + 1 + n_imp, device=edges.device, dtype=torch.float32)
    with torch.cuda.device(edges.device):
        L.check(L.lib().fsn_sample_pdf_merge(_p(edges), _p(weights), R, S, n_imp, _p(u), _p(out), _stream()),
                "fsn_sample_pdf_merge")
    return out


# ------------------------------------------------------------------ compositing
def composite(sigmas: Tensor, rgbs: Tensor, t_starts: Tensor, t_ends: Tensor, bkgd=None, extras: bool = True):
    """Dense [R,S] volume integration.  Returns colors [R,3], opacity [R,1], depth [R,1], extras."""
    sig, rgb = _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs")
    t0, t1 = _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    R, S = sig.shape
    dev = sig.device
    colors = torch.empty(R, 3, device=dev)
    opacity = torch.empty(R, 1, device=dev)
    depth = torch.empty(R, 1, device=dev)
    w = torch.empty(R, S, device=dev) if extras else None
    a = torch.empty(R, S, device=dev) if extras else None
    tr = torch.empty(R, S, device=dev) if extras else None
    with torch.cuda.device(dev):
        L.check(L.lib().fsn_composite_fwd(_p(sig), _p(rgb), _p(t0), _p(t1), R, S, _bk(bkgd), _p(colors), _p(opacity),
                                          _p(depth), _p(w), _p(a), _p(tr), _stream()), "fsn_composite_fwd")
    ex = {"weights": w, "alphas": a, "trans": tr, "sigmas": sig, "rgbs": rgb} if extras else {}
    return colors, opacity, depth, ex


def composite_packed(sigmas: Tensor, rgbs: Tensor, t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor,
                     n_rays: int, bkgd=None):
    sig, rgb = _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs")
    t0, t1 = _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    ri = _i64(ray_indices, "ray_indices")
    N = sig.numel()
    dev = t0.device
    colors = torch.empty(n_rays, 3, device=dev)
    opacity = torch.empty(n_rays, 1, device=dev)
    depth = torch.empty(n_rays, 1, device=dev)
    w, a, tr = (torch.empty(N, device=dev) for _ in range(3))
    with torch.cuda.device(dev):
        L.check(L.lib().fsn_composite_packed_fwd(_p(sig), _p(rgb), _p(t0), _p(t1), _p(ri), N, n_rays, _bk(bkgd),
                                                 _p(colors), _p(opacity), _p(depth), _p(w), _p(a), _p(tr), _stream()),
                "fsn_composite_packed_fwd")
    return colors, opacity, depth, {"weights": w, "alphas": a, "trans": tr, "sigmas": sig, "rgbs": rgb}


# ------------------------------------------------------------------ MLP
SD_ORDER_TAIL = ("sigma", "connection", "branch", "rgb")

# Range guard of the fp16 modes (include/fsnerf_hip.h, FSN_STATUS_FP16_RANGE): one sticky device word per GPU that
# every MLP launch of this process reports into.
_status_words: Dict[torch.device, Tensor] = {}


def status_word(device) -> Tensor:
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    w = _status_words.get(device)
    if w is None:
        w = _status_words[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


_step_flags: Dict[torch.device, Tensor] = {}


def step_flag(device) -> Tensor:
    """Per-device word of the current training step for parameters that NO gradient bucket owns (plain torch
    optimizers): a NeRF backward ORs its per-call range word into it (device op).  Parameters in a `shard.FlatGrads`
    bucket (`FusedAdam`) report into the bucket's own `step_flag` instead, which that optimizer hands to its
    Adam launch (bit 0 / bit 2 set: the update is skipped on the device) and clears.  Never read by the host."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    w = _step_flags.get(device)
    if w is None:
        w = _step_flags[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def range_flags(device) -> int:
    """FSN_STATUS_* bits reported by the launches since the last call (0 = all inside the fp16 modes' envelope); the
    word is cleared.  Reads 4 bytes back: one host sync."""
    w = status_word(device)
    bits = int(w.item()) & (L.FSN_STATUS_FP16_RANGE | L.FSN_STATUS_FP16_SMALL)
    if bits:
        w.zero_()
    return bits


# Deferred look at the range word (`NeRF.range_check = "deferred"`): batch rendering in small launches must not wait for
# the GPU after every call.  range_post(): asynchronous copy of the word into pinned host memory + an event, on the
# launch stream.  range_poll(): bits seen by the most recent post (waits for its event - normally long complete when the
# NEXT call polls), clearing the device word when it was raised.
_range_pending: Dict[torch.device, Tuple[Tensor, "torch.cuda.Event"]] = {}


def range_post(device) -> None:
    w = status_word(device)
    ent = _range_pending.get(w.device)
    host = ent[0] if ent is not None else torch.zeros(1, dtype=torch.int32).pin_memory()
    host.copy_(w, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _range_pending[w.device] = (host, ev, True)


def range_poll(device) -> int:
    w = status_word(device)
    ent = _range_pending.get(w.device)
    if ent is None or not ent[2]:
        return 0
    host, ev, _ = ent
    ev.synchronize()
    _range_pending[w.device] = (host, ev, False)
    bits = int(host[0]) & (L.FSN_STATUS_FP16_RANGE | L.FSN_STATUS_FP16_SMALL)
    if bits:
        w.zero_()
    return bits


def range_ok(device) -> bool:
    """False when a launch since the last call reported values outside the fp16 modes' envelope (either end)."""
    return range_flags(device) == 0


def describe_flags(bits: int) -> str:
    what = []
    if bits & L.FSN_STATUS_FP16_RANGE:
        what.append("values reached the top of the fp16 range (|v| >= 65504)")
    if bits & L.FSN_STATUS_FP16_SMALL:
        what.append("a layer's activations were all below 2^-14 (below the split's float32-grade envelope; scaled "
                    "inference: a wavefront's layer maximum below 2^-4)")
    return " and ".join(what) if what else "inside the envelope"



def make_desc(n_layers: int, d_hidden: int, skip: Sequence[int], freqs_pos: Sequence[float],
              freqs_dir: Sequence[float]) -> L.MlpDesc:
    d = L.MlpDesc()
    d.n_layers, d.d_hidden = n_layers, d_hidden
    mask = 0
    for i in skip:
        if 0 <= i < n_layers - 1:
            mask |= 1 << i
        elif i == n_layers - 1:
            raise ValueError(f"skip index {i} == n_layers-1: the reference model cannot run this either "
                             "(sigma expects d_hidden inputs, src/core/models.py:107,123)")
    d.skip_mask = mask
    d.n_freqs_pos, d.n_freqs_dir = len(freqs_pos), len(freqs_dir)
    if len(freqs_pos) > 16 or len(freqs_dir) > 16:
        raise ValueError("too many frequency bands")
    for i, f in enumerate(freqs_pos):
        d.freqs_pos[i] = float(f)
    for i, f in enumerate(freqs_dir):
        d.freqs_dir[i] = float(f)
    return d


def sd_tensor_lists(sd: Dict[str, Tensor], n_layers: int) -> Tuple[List[Tensor], List[Tensor]]:
    names = [f"layers.{i}" for i in range(n_layers)] + list(SD_ORDER_TAIL)
    return [sd[n + ".weight"] for n in names], [sd[n + ".bias"] for n in names]


class PackedMLP:
    """A reference-format state_dict packed for the MFMA kernels (device blob + descriptor)."""

    def __init__(self, desc: L.MlpDesc, prec: int, device):
        self.desc, self.prec = desc, prec
        nbytes = L.lib().fsn_mlp_blob_bytes(C.byref(desc), prec)
        if nbytes < 0:
            L.check(int(nbytes), "fsn_mlp_blob_bytes")
        self.blob = torch.empty(int(nbytes), dtype=torch.uint8, device=device)

    def pack(self, weights: Sequence[Tensor], biases: Sequence[Tensor], exps: Optional[Sequence[int]] = None) -> "PackedMLP":
        """`exps`: n_layers + 2 per-GEMM activation exponents (fsn_mlp_pack_scaled: the blob then holds the network whose
        GEMM g produces 2^exps[g] times the reference's activations, heads unscaled accordingly), or None."""
        n = self.desc.n_layers + 4
        assert len(weights) == n and len(biases) == n
        ws = [_f32(w.detach(), "weight") for w in weights]
        bs = [_f32(b.detach(), "bias") for b in biases]
        Wp = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
        Bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
        ex = None
        if exps is not None:
            assert len(exps) == self.desc.n_layers + 2
            ex = (C.c_int32 * len(exps))(*[int(e) for e in exps])
        self.exps = None if exps is None else tuple(int(e) for e in exps)
        with torch.cuda.device(self.blob.device):
            L.check(L.lib().fsn_mlp_pack_scaled(C.byref(self.desc), self.prec, Wp, Bp, ex, _p(self.blob), _stream()),
                    "fsn_mlp_pack_scaled")
        self._keep = (ws, bs)  # alive until the stream work is queued behind later launches
        return self


def mlp_layer_maxima(pm: PackedMLP, x: Tensor, dirs: Tensor, pos_mask: Optional[Tensor] = None,
                     dir_mask: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """Per-GEMM largest |activation| of NeRF.forward(x, dirs) over the probe samples (fsn_mlp_layer_maxima) ->
    float32 [n_layers + 2] on the device, in the scale of the blob `pm`.  `out`: accumulate into an earlier result."""
    x, d = _f32(x, "x").reshape(-1, 3), _f32(dirs, "dirs").reshape(-1, 3)
    assert x.shape == d.shape
    pmk = None if pos_mask is None else _f32(pos_mask, "pos_mask")
    dmk = None if dir_mask is None else _f32(dir_mask, "dir_mask")
    if out is None:
        out = torch.zeros(pm.desc.n_layers + 2, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        L.check(L.lib().fsn_mlp_layer_maxima(C.byref(pm.desc), pm.prec, _p(pm.blob), _p(x), _p(d), _p(pmk), _p(dmk),
                                             x.shape[0], _p(out), _stream()), "fsn_mlp_layer_maxima")
    return out


def mlp_fwd(pm: PackedMLP, x: Tensor, dirs: Optional[Tensor] = None, pos_mask: Optional[Tensor] = None,
            dir_mask: Optional[Tensor] = None) -> Tensor:
    x = _f32(x, "x")
    lead = x.shape[:-1]
    n = x.numel() // 3
    d = None if dirs is None else _f32(dirs, "dirs")
    pmk = None if pos_mask is None else _f32(pos_mask, "pos_mask")
    dmk = None if dir_mask is None else _f32(dir_mask, "dir_mask")
    out = torch.empty(*lead, 4 if d is not None else 1, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        L.check(L.lib().fsn_mlp_fwd(C.byref(pm.desc), pm.prec, _p(pm.blob), _p(x), _p(d), _p(pmk), _p(dmk), n,
                                    _p(out), _p(status_word(x.device)), _stream()), "fsn_mlp_fwd")
    return out


# profiling hook (bench.py): a list -> every render_fused launch appends a pair of HIP events recorded on the launch
# stream right around the C-ABI call (the kernel's own duration, without the host path around it)
launch_timer: Optional[list] = None
# measurement hook (bench.py): a uint64 [2] device tensor -> every render_fused launch adds its (s_memtime, s_memrealtime)
# differences to it (fsn_render_args.clock_out): the clock the chip held during the launches
clock_buffer: Optional[Tensor] = None

def mlp_fwd_rays(pm: PackedMLP, rays_o: Tensor, rays_d: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor,
                 full: bool, pos_mask: Optional[Tensor] = None, dir_mask: Optional[Tensor] = None) -> Tensor:
    """NeRF.forward on the midpoints of packed intervals (the reference's sigma_fn / rgb_sigma_fn, rendering.py:58-64,
    76-84) without materialising the gathered [N,3] positions / directions: -> [N,4] (full) or [N,1]."""
    o, d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
    ri, t0, t1 = _i64(ray_indices, "ray_indices"), _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    n = ri.numel()
    pmk = None if pos_mask is None else _f32(pos_mask, "pos_mask")
    dmk = None if dir_mask is None else _f32(dir_mask, "dir_mask")
    out = torch.empty(n, 4 if full else 1, device=o.device, dtype=torch.float32)
    with torch.cuda.device(o.device):
        L.check(L.lib().fsn_mlp_fwd_rays(C.byref(pm.desc), pm.prec, _p(pm.blob), _p(o), _p(d), _p(ri), _p(t0), _p(t1),
                                         1 if full else 0, _p(pmk), _p(dmk), n, _p(out), _p(status_word(o.device)),
                                         _stream()), "fsn_mlp_fwd_rays")
    return out


_edges_ws: Dict[Tuple[torch.device, int], Tensor] = {}


def _edges_workspace(dev, n: int) -> Tensor:
    """Hand-over buffer of the two-phase frame mode when the caller does not ask for the edges: one cached tensor per
    device (grown on demand), so that a path of frames does not re-allocate 0.5 GB per frame."""
    dev = torch.device(dev)
    w = _edges_ws.get((dev, 0))
    if w is None or w.numel() < n:
        w = _edges_ws[(dev, 0)] = torch.empty(n, device=dev, dtype=torch.float32)
    return w


def render_fused(pm_coarse: Optional[PackedMLP], pm_fine: PackedMLP, rays_o: Optional[Tensor], rays_d: Optional[Tensor], *,
                 near: float, far: float, n_samples: int, n_importance: int = 0, u: Optional[Tensor] = None,
                 u_fine: Optional[Tensor] = None, bkgd=(0.0, 0.0, 0.0), pos_mask: Optional[Tensor] = None,
                 dir_mask: Optional[Tensor] = None, want_extras: bool = True, camera=None,
                 two_phase: Optional[bool] = None):
    """One launch for the whole path (fsn_render_rays_fused).  Returns colors [R,3], opacity [R,1],
    depth [R,1], extras {weights, alphas, trans, sigmas [R,S'], rgbs [R,S',3], edges [R,S'+1],
    weights_coarse [R,S] (hierarchical only)}.
    `camera` = (pose [3or4,4], H, W, focal, row0, nrows, device) instead of ray tensors: the rays of the image rows
    [row0, row0+nrows) are generated inside the launch (the arithmetic of get_rays), R = nrows*W.
    `two_phase` (default: hierarchical launches of >= 65,536 rays): all coarse passes of a workgroup before its fine
    passes, the resampled edges handed over through HBM (its own kernel instantiation: two simple loops instead of
    one long one, half the register spills); same results, 3 % faster on the 800x800 frame (A/B on one device)."""
    S, NI = n_samples, n_importance
    So = S + NI
    if camera is not None:
        pose, cH, cW, cfocal, crow0, cnrows, dev = camera
        dev = torch.device(dev)
        R = int(cnrows) * int(cW)
        o = d = None
    else:
        o, d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
        R = o.shape[0]
        dev = o.device
    mode, u = _u_mode(u, R, S)
    a = L.RenderArgs()
    a.R = R
    if camera is None:
        a.rays_o, a.rays_d = o.data_ptr(), d.data_ptr()
    else:
        pm = pose.detach().to("cpu", torch.float32)[:3, :4].contiguous().reshape(-1).tolist()
        for i in range(12):
            a.cam_pose[i] = pm[i]
        a.cam_H, a.cam_W, a.cam_row0, a.cam_focal = int(cH), int(cW), int(crow0), float(cfocal)
    a.near, a.far, a.S, a.n_imp, a.u_mode = float(near), float(far), S, NI, mode
    keep = [o, d, u]
    a.u = None if u is None else u.data_ptr()
    if u_fine is not None:
        u_fine = _f32(u_fine, "u_fine")
        assert tuple(u_fine.shape) == (R, NI)
        a.u_fine = u_fine.data_ptr()
    for name, m in (("pos_mask", pos_mask), ("dir_mask", dir_mask)):
        if m is not None:
            m = _f32(m, name)
            keep.append(m)
            setattr(a, name, m.data_ptr())
    for i in range(3):
        a.bkgd[i] = float(bkgd[i])
    colors = torch.empty(R, 3, device=dev)
    opacity = torch.empty(R, 1, device=dev)
    depth = torch.empty(R, 1, device=dev)
    a.colors, a.opacity, a.depth = colors.data_ptr(), opacity.data_ptr(), depth.data_ptr()
    ex: Dict[str, Tensor] = {}
    if want_extras:
        for k, shape in (("weights", (R, So)), ("alphas", (R, So)), ("trans", (R, So)), ("sigmas", (R, So)),
                         ("rgbs", (R, So, 3))):
            ex[k] = torch.empty(*shape, device=dev)
            setattr(a, k, ex[k].data_ptr())
        ex["edges"] = torch.empty(R, So + 1, device=dev)
        a.edges_out = ex["edges"].data_ptr()
        if NI > 0:
            ex["weights_coarse"] = torch.empty(R, S, device=dev)
            a.weights_coarse = ex["weights_coarse"].data_ptr()
    if two_phase is None:
        two_phase = NI > 0 and R >= 65536
    if two_phase and NI > 0:
        if "edges" not in ex:
            keep.append(_edges_workspace(dev, R * (So + 1)))
            a.edges_out = keep[-1].data_ptr()
        a.two_phase = 1
    if NI > 0 and pm_coarse is None:
        pm_coarse = pm_fine
    if pm_coarse is not None and pm_coarse.prec != pm_fine.prec:
        raise ValueError("render_fused: the coarse and the fine network must be packed in the same precision mode")
    a.status = status_word(dev).data_ptr()
    if clock_buffer is not None:
        a.clock_out = clock_buffer.data_ptr()
    with torch.cuda.device(dev):
        if launch_timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(L.lib().fsn_render_rays_fused(C.byref(pm_fine.desc), pm_fine.prec,
                                              _p(pm_coarse.blob) if pm_coarse is not None else None,
                                              _p(pm_fine.blob), C.byref(a), _stream()), "fsn_render_rays_fused")
        if launch_timer is not None:
            e1.record()
            launch_timer.append((e0, e1))
    return colors, opacity, depth, ex


def bench_bare_stream(pm: PackedMLP, layers: int) -> Tuple[int, Tensor]:
    """Measurement aid (fsn_bench_bare_stream): `layers` 256 -> 256 hidden layers back to back per workgroup through the
    render kernels' own GEMM code, the weight stream walking the hidden phases of `pm`'s blob.  -> (workgroups launched,
    uint64 [workgroups * 8, 2] per-wave (s_memtime, s_memrealtime) differences around the loop)."""
    dev = pm.blob.device
    with torch.cuda.device(dev):
        n = L.lib().fsn_device_cus()
        clk = torch.zeros(max(n, 1) * 8, 2, dtype=torch.int64, device=dev)
        rc = L.lib().fsn_bench_bare_stream(C.byref(pm.desc), pm.prec, _p(pm.blob), int(layers), _p(clk), _stream())
        if rc <= 0:
            L.check(rc if rc < 0 else -3, "fsn_bench_bare_stream")
    return rc, clk


def sample_fused(pm_coarse: PackedMLP, rays_o: Tensor, rays_d: Tensor, *, near: float, far: float, n_samples: int,
                 n_importance: int, u: Optional[Tensor] = None, u_fine: Optional[Tensor] = None,
                 pos_mask: Optional[Tensor] = None, dir_mask: Optional[Tensor] = None,
                 want_weights: bool = False, status: Optional[Tensor] = None):
    """The hierarchical sampler as ONE launch (fsn_render_rays_fused with two_phase = 2): stratified edges -> density
    pass of `pm_coarse` -> weights -> inverse-CDF resampling -> sorted union.  -> edges [R, S+NI+1] (and the coarse
    weights [R,S] when asked for): what StratifiedEstimator.sampling builds from five launches around a sigma_fn."""
    S, NI = n_samples, n_importance
    if NI <= 0:
        raise ValueError("sample_fused: hierarchical sampling only (n_importance > 0)")
    o, d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
    R, dev = o.shape[0], o.device
    mode, u = _u_mode(u, R, S)
    a = L.RenderArgs()
    a.R, a.rays_o, a.rays_d = R, o.data_ptr(), d.data_ptr()
    a.near, a.far, a.S, a.n_imp, a.u_mode = float(near), float(far), S, NI, mode
    keep = [o, d, u]
    a.u = None if u is None else u.data_ptr()
    if u_fine is not None:
        u_fine = _f32(u_fine, "u_fine")
        assert tuple(u_fine.shape) == (R, NI)
        a.u_fine = u_fine.data_ptr()
    for name, m in (("pos_mask", pos_mask), ("dir_mask", dir_mask)):
        if m is not None:
            m = _f32(m, name)
            keep.append(m)
            setattr(a, name, m.data_ptr())
    edges = torch.empty(R, S + NI + 1, device=dev)
    a.edges_out = edges.data_ptr()
    wc = torch.empty(R, S, device=dev) if want_weights else None
    if wc is not None:
        a.weights_coarse = wc.data_ptr()
    a.two_phase = 2
    a.status = (status_word(dev) if status is None else status).data_ptr()  # `status`: a per-call word (training steps)
    if R > 0:
        with torch.cuda.device(dev):
            L.check(L.lib().fsn_render_rays_fused(C.byref(pm_coarse.desc), pm_coarse.prec, _p(pm_coarse.blob), None,
                                                  C.byref(a), _stream()), "fsn_render_rays_fused")
    return (edges, wc) if want_weights else edges


# ------------------------------------------------------------------ "next" rows
class _OcclusionRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sig, t, ri, n_rays, a, b, func):
        N = sig.numel()
        sums = torch.empty(max(n_rays, 1), device=sig.device, dtype=torch.float32)
        out = torch.empty(1, device=sig.device, dtype=torch.float32)
        with torch.cuda.device(sig.device):
            L.check(L.lib().fsn_occlusion_reg_fwd(_p(sig), _p(t), _p(ri), N, n_rays, float(a), float(b), func, _p(sums),
                                                  _p(out), _stream()), "fsn_occlusion_reg_fwd")
        ctx.save_for_backward(t, sums)
        ctx.cfg = (N, n_rays, float(a), float(b), func)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        t, sums = ctx.saved_tensors
        N, n_rays, a, b, func = ctx.cfg
        d_sig = torch.zeros(N, device=t.device, dtype=torch.float32)
        cnt = torch.empty(1, device=t.device, dtype=torch.int32)
        g = _f32(g, "grad").reshape(1)
        with torch.cuda.device(t.device):
            L.check(L.lib().fsn_occlusion_reg_bwd(_p(t), N, _p(sums), n_rays, a, b, func, _p(g), _p(cnt), _p(d_sig),
                                                  _stream()), "fsn_occlusion_reg_bwd")
        return d_sig, None, None, None, None, None, None


def occlusion_reg(sigmas: Tensor, t_vals: Tensor, ray_idxs: Tensor, a: float, b: float, func: str = "linear",
                  n_rays: Optional[int] = None) -> Tensor:
    """OcclusionRegularizer (src/core/loss.py:26-60) -> 0-dim tensor; differentiable w.r.t. sigmas."""
    sig, t = _f32(sigmas, "sigmas").reshape(-1), _f32(t_vals.detach(), "t_vals").reshape(-1)
    ri = ray_idxs.contiguous()
    if ri.dtype != torch.int64:
        ri = ri.long()
    N = sig.numel()
    if n_rays is None:
        n_rays = int(ri[-1].item()) + 1 if N > 0 else 0
    if func not in ("linear", "exp"):
        raise ValueError(f"Unknown occlusion regularizer type: {func}")
    return _OcclusionRegFn.apply(sig, t, ri, n_rays, a, b, 0 if func == "linear" else 1)


def to8b(x: Tensor) -> Tensor:
    """(255 * clip(x, 0, 1)).astype(uint8) on the device (src/render/rendering.py:21)."""
    x = _f32(x, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    with torch.cuda.device(x.device):
        L.check(L.lib().fsn_to8b(_p(x), x.numel(), _p(out), _stream()), "fsn_to8b")
    return out


_lut_cache: Dict[Tuple[str, torch.device], Tensor] = {}


def video_tensors(frames: Tensor, d_frames: Tensor, cmap: str = "plasma") -> Tuple[Tensor, Tensor]:
    """render_video (src/render/rendering.py:240-266) on the device: frames [N,H,W,3], d_frames [N,H,W] ->
    (uint8 [N,3,H,W], uint8 [N,3,H,W]): to8b + NHWC->NCHW for the colour frames; global min / max normalisation,
    colormap lookup and to8b for the depth frames."""
    from .render.cmaps import TABLES
    if cmap not in TABLES:
        raise ValueError(f"render_video: colormap {cmap!r} is not in the shipped tables {sorted(TABLES)}")
    fr, dp = _f32(frames, "frames"), _f32(d_frames, "d_frames")
    N, H, W = dp.shape
    assert fr.shape == (N, H, W, 3)
    dev = fr.device
    key = (cmap, dev)
    if key not in _lut_cache:
        _lut_cache[key] = torch.frombuffer(bytearray(TABLES[cmap]), dtype=torch.uint8).to(dev)
    out_f = torch.empty(N, 3, H, W, device=dev, dtype=torch.uint8)
    out_d = torch.empty(N, 3, H, W, device=dev, dtype=torch.uint8)
    if N * H * W == 0:
        return out_f, out_d
    vmm = torch.stack(torch.aminmax(dp)).to(torch.float32).contiguous()  # np.amin / np.amax over ALL frames
    with torch.cuda.device(dev):
        L.check(L.lib().fsn_to8b_nchw(_p(fr), N, H * W, _p(out_f), _stream()), "fsn_to8b_nchw")
        L.check(L.lib().fsn_depth_colormap(_p(dp), N, H * W, _p(vmm), _p(_lut_cache[key]), _p(out_d), _stream()),
                "fsn_depth_colormap")
    return out_f, out_d


# ------------------------------------------------------------------ training step
def _ptr_array(ts: Sequence[Tensor]):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def nerf_train_fwd_rays(desc: L.MlpDesc, prec: int, weights: Sequence[Tensor], biases: Sequence[Tensor], rays_o: Tensor,
                        rays_d: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor, pos_mask: Optional[Tensor],
                        dir_mask: Optional[Tensor], status: Optional[Tensor] = None):
    """nerf_train_fwd with the samples in ray form (midpoints of packed intervals): -> (out [n,4], workspace)."""
    o, d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
    ri, t0, t1 = _i64(ray_indices, "ray_indices"), _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    n = ri.numel()
    ws_ = [_f32(w.detach(), "weight") for w in weights]
    bs_ = [_f32(b.detach(), "bias") for b in biases]
    with torch.cuda.device(o.device):
        pm = None if pos_mask is None else _f32(pos_mask, "pos_mask")
        dm = None if dir_mask is None else _f32(dir_mask, "dir_mask")
        out = torch.empty(n, 4, device=o.device, dtype=torch.float32)
        nfl = L.lib().fsn_nerf_train_workspace_floats(C.byref(desc), prec, n)
        if nfl < 0:
            L.check(int(nfl), "fsn_nerf_train_workspace_floats")
        work = torch.empty(max(int(nfl), 1), device=o.device, dtype=torch.float32)
        L.check(L.lib().fsn_nerf_train_fwd_rays(C.byref(desc), prec, _ptr_array(ws_), _ptr_array(bs_), _p(o), _p(d), _p(ri),
                                                _p(t0), _p(t1), _p(pm), _p(dm), n, _p(work), _p(out), _p(status), _stream()),
                "fsn_nerf_train_fwd_rays")
    return out, work


def nerf_train_fwd(desc: L.MlpDesc, prec: int, weights: Sequence[Tensor], biases: Sequence[Tensor], x: Tensor,
                   dirs: Tensor, pos_mask: Optional[Tensor], dir_mask: Optional[Tensor], status: Optional[Tensor] = None):
    """NeRF.forward keeping what the backward needs: -> (out [n,4], workspace).  prec: L.FSN_PREC_*.  `status`: this
    call's range-guard word (int32[1], zeroed by the caller; the same tensor goes to nerf_train_bwd)."""
    x, d = _f32(x, "x").reshape(-1, 3), _f32(dirs, "dirs").reshape(-1, 3)
    n = x.shape[0]
    ws_ = [_f32(w.detach(), "weight") for w in weights]
    bs_ = [_f32(b.detach(), "bias") for b in biases]
    with torch.cuda.device(x.device):
        pm = None if pos_mask is None else _f32(pos_mask, "pos_mask")
        dm = None if dir_mask is None else _f32(dir_mask, "dir_mask")
        out = torch.empty(n, 4, device=x.device, dtype=torch.float32)
        nfl = L.lib().fsn_nerf_train_workspace_floats(C.byref(desc), prec, n)
        if nfl < 0:
            L.check(int(nfl), "fsn_nerf_train_workspace_floats")
        work = torch.empty(max(int(nfl), 1), device=x.device, dtype=torch.float32)
        L.check(L.lib().fsn_nerf_train_fwd(C.byref(desc), prec, _ptr_array(ws_), _ptr_array(bs_), _p(x), _p(d), _p(pm),
                                           _p(dm), n, _p(work), _p(out), _p(status), _stream()),
                "fsn_nerf_train_fwd")
    return out, work


def grad_scale_for(d_out: Tensor) -> Tensor:
    """Power of two that brings max|d_out| to ~2^10 (device scalar, no host sync): the fp16 MFMA modes of the
    backward multiply d_out by it on entry and divide the gradients by it at the end.  One launch (fsn_grad_scale):
    2^clamp(floor(log2(1024 / max|d_out|)), -40, 60), 1 when the maximum is 0 / inf / NaN."""
    d = _f32(d_out.detach(), "d_out").reshape(-1)
    buf = torch.zeros(4, dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        L.check(L.lib().fsn_grad_scale(_p(d), d.numel(), _p(buf), _stream()), "fsn_grad_scale")
    return buf[:1]


def nerf_train_bwd(desc: L.MlpDesc, prec: int, weights: Sequence[Tensor], work: Tensor, out: Tensor, d_out: Tensor,
                   status: Optional[Tensor] = None, into: Optional[Tuple[Sequence[Tensor], Sequence[Tensor]]] = None,
                   stage_state: Optional[Tuple[Tensor, Tensor]] = None):
    """-> (d_weights, d_biases) lists in state_dict order.  `status`: the word given to nerf_train_fwd.
    `into` = (weight .grad buffers, bias .grad buffers): the gradients are ADDED to them on the device (what autograd's
    AccumulateGrad does with returned tensors, without the temporaries and one add launch per parameter); -> into.
    `stage_state` = (scales float32 [n_layers + 2], maxima int32 [n_layers + 2]), fp16 modes: the backward chain's delayed
    per-stage power-of-two factors (include/fsnerf_hip.h); the call uses the scales and updates them for the next one."""
    ws_ = [_f32(w.detach(), "weight") for w in weights]
    d_out = _f32(d_out, "d_out").reshape(-1, 4)
    n = d_out.shape[0]
    if into is not None:
        for g in list(into[0]) + list(into[1]):
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous()):
                raise ValueError("nerf_train_bwd: `into` buffers must be contiguous float32 GPU tensors (they are written in place)")
        dW, db = list(into[0]), [g.view(-1) for g in into[1]]
        if len(dW) != len(ws_) or len(db) != len(ws_) or any(g.shape != w.shape for g, w in zip(dW, ws_)) or \
                any(g.numel() != w.shape[0] for g, w in zip(db, ws_)):
            raise ValueError("nerf_train_bwd: `into` must hold one gradient buffer per weight / bias, of its shape")
        if n == 0:
            return dW, db
    else:
        if n == 0:  # e.g. the all-background first batch of an empty occupancy grid (run-nerf.py:243 precedes :293)
            return [torch.zeros_like(w) for w in ws_], [torch.zeros(w.shape[0], device=w.device) for w in ws_]
        dW = [torch.empty_like(w) for w in ws_]
        db = [torch.empty(w.shape[0], device=w.device, dtype=torch.float32) for w in ws_]
    scale = grad_scale_for(d_out) if prec in (L.FSN_PREC_FP16X3, L.FSN_PREC_FP16) else None
    with torch.cuda.device(work.device):
        L.check(L.lib().fsn_nerf_train_bwd(C.byref(desc), prec, _ptr_array(ws_), n, _p(work), _p(out), _p(d_out),
                                           _p(scale), _ptr_array(dW), _ptr_array(db), 0 if into is None else 1,
                                           _p(stage_state[0]) if stage_state else None,
                                           _p(stage_state[1]) if stage_state else None, _p(status), _stream()),
                "fsn_nerf_train_bwd")
    return dW, db


def composite_packed_bwd(sigmas, rgbs, t_starts, t_ends, ray_indices, n_rays, bkgd, d_colors, d_opacity):
    sig, rgb = _f32(sigmas, "sigmas"), _f32(rgbs, "rgbs")
    t0, t1 = _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    ri = _i64(ray_indices, "ray_indices")
    N = sig.numel()
    dc = _f32(d_colors, "d_colors")
    dop = None if d_opacity is None else _f32(d_opacity, "d_opacity").reshape(-1)
    ds, dr = torch.empty_like(sig), torch.empty_like(rgb)
    with torch.cuda.device(sig.device):
        L.check(L.lib().fsn_composite_packed_bwd(_p(sig), _p(rgb), _p(t0), _p(t1), _p(ri), N, n_rays, _bk(bkgd), _p(dc),
                                                 _p(dop), _p(ds), _p(dr), _stream()), "fsn_composite_packed_bwd")
    return ds, dr


# ------------------------------------------------------------------ occupancy-grid sampler
def occgrid_march(rays_o: Tensor, rays_d: Tensor, aabb: Sequence[float], res: int, levels: int, bits: Tensor,
                  near_plane: float, far_plane: float, step: float, u: Optional[Tensor], max_steps: int):
    """Lattice march through the occupancy grid -> (ray_indices int64 [N], t_starts [N], t_ends [N], counts [R]).
    Two launches around an exclusive scan of the per-ray counts; one host sync for N (as nerfacc's does)."""
    o, d = _f32(rays_o, "rays_o").reshape(-1, 3), _f32(rays_d, "rays_d").reshape(-1, 3)
    R = o.shape[0]
    ab = (C.c_float * 6)(*[float(v) for v in aabb])
    u_ = None if u is None else _f32(u, "u").reshape(-1)
    if u_ is not None and u_.numel() != R:
        raise ValueError("u must hold one value per ray")
    counts = torch.zeros(R, device=o.device, dtype=torch.int64)
    args = (_p(o), _p(d), R, ab, int(res), int(levels), _p(bits), float(near_plane), float(far_plane), float(step), _p(u_),
            int(max_steps))
    with torch.cuda.device(o.device):
        L.check(L.lib().fsn_occgrid_march(*args, _p(counts), None, None, None, None, _stream()), "fsn_occgrid_march")
        ends = torch.cumsum(counts, 0)
        offsets = (ends - counts).contiguous()
        N = int(ends[-1].item()) if R > 0 else 0
        ri = torch.empty(N, device=o.device, dtype=torch.int64)
        t0 = torch.empty(N, device=o.device, dtype=torch.float32)
        t1 = torch.empty(N, device=o.device, dtype=torch.float32)
        if N > 0:
            L.check(L.lib().fsn_occgrid_march(*args, None, _p(offsets), _p(ri), _p(t0), _p(t1), _stream()), "fsn_occgrid_march")
    return ri, t0, t1, counts


def occ_sample_fused(pm: PackedMLP, rays_o: Tensor, rays_d: Tensor, *, aabb: Sequence[float], res: int, levels: int, bits: Tensor,
                     near_plane: float, far_plane: float, step: float, max_steps: int, u: Optional[Tensor] = None,
                     early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, pos_mask: Optional[Tensor] = None,
                     dir_mask: Optional[Tensor] = None, status: Optional[Tensor] = None):
    """OccGridEstimator.sampling(..., sigma_fn = the model's density) of a training step as ONE launch + one gather:
    grid march -> density pass -> visibility cull inside persistent workgroups (the sampler mode of
    fsn_render_rays_occgrid), the kept samples of every ray in a slot of its own, then an exclusive scan of the counts,
    ONE host read (the sample count sizes the outputs) and fsn_occ_gather_samples.  -> (ray_indices int64 [N], t_starts
    [N], t_ends [N]): what occgrid_march (two launches + a read) -> density pass -> packed_visibility -> compaction
    (a second read) give, bit for bit."""
    o, d = _f32(rays_o, "rays_o").reshape(-1, 3), _f32(rays_d, "rays_d").reshape(-1, 3)
    Rn, dev = o.shape[0], o.device
    a = L.OccRenderArgs()
    a.R, a.rays_o, a.rays_d = Rn, o.data_ptr(), d.data_ptr()
    keep = [o, d]
    for i in range(6):
        a.aabb[i] = float(aabb[i])
    a.res, a.levels, a.bits = int(res), int(levels), bits.data_ptr()
    a.near_plane, a.far_plane, a.step, a.max_steps = float(near_plane), float(far_plane), float(step), int(max_steps)
    if u is not None:
        u = _f32(u, "u").reshape(-1)
        if u.numel() != Rn:
            raise ValueError("u must hold one value per ray")
        keep.append(u)
        a.u = u.data_ptr()
    a.early_stop_eps, a.alpha_thre = float(early_stop_eps), float(alpha_thre)
    for name, m in (("pos_mask", pos_mask), ("dir_mask", dir_mask)):
        if m is not None:
            m = _f32(m, name)
            keep.append(m)
            setattr(a, name, m.data_ptr())
    n_kept = torch.zeros(Rn, dtype=torch.int32, device=dev)
    slots = torch.empty(max(Rn, 1), int(max_steps), dtype=torch.float32, device=dev)
    a.n_kept, a.sample_t0, a.sample_cap = n_kept.data_ptr(), slots.data_ptr(), int(max_steps)
    # (one queue counter PER CALL - ADVICE r3: a per-device buffer let two launches on different streams race on it; the
    # launch zeroes it itself)
    wc = torch.empty(1, dtype=torch.int64, device=dev)
    keep.append(wc)
    a.work_counter = wc.data_ptr()
    a.status = (status_word(dev) if status is None else status).data_ptr()
    with torch.cuda.device(dev):
        if Rn > 0:
            L.check(L.lib().fsn_render_rays_occgrid(C.byref(pm.desc), pm.prec, _p(pm.blob), C.byref(a), _stream()),
                    "fsn_render_rays_occgrid (sampler mode)")
        incl = torch.cumsum(n_kept, 0, dtype=torch.int64)
        N = int(incl[-1].item()) if Rn > 0 else 0  # the one host read: the sample count sizes the outputs
        ri = torch.empty(N, dtype=torch.int64, device=dev)
        ts, te = torch.empty(N, device=dev), torch.empty(N, device=dev)
        if N > 0:
            offs = incl - n_kept
            L.check(L.lib().fsn_occ_gather_samples(_p(n_kept), _p(offs), _p(slots), int(max_steps), Rn, float(step), _p(ri),
                                                   _p(ts), _p(te), _stream()), "fsn_occ_gather_samples")
    return ri, ts, te


def render_occ_fused(pm: PackedMLP, rays_o: Optional[Tensor], rays_d: Optional[Tensor], *, aabb: Sequence[float], res: int,
                     levels: int, bits: Tensor, near_plane: float, far_plane: float, step: float, max_steps: int,
                     u: Optional[Tensor] = None, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                     bkgd=(0.0, 0.0, 0.0), pos_mask: Optional[Tensor] = None, dir_mask: Optional[Tensor] = None,
                     camera=None, want_counts: bool = False, want_extras: bool = False):
    """render_rays with the occupancy estimator in ONE launch (fsn_render_rays_occgrid): grid march -> density pass ->
    visibility cull -> full pass -> packed integration, no host sync.  -> colors [R,3], opacity [R,1], depth [R,1],
    counts {"n_cand", "n_kept"} (int32 [R]) when asked for.  `camera` as in render_fused.
    `want_extras` (EXTRAS mode): the launch also leaves the kept samples' weights / alphas / trans / sigmas / rgbs and
    interval starts in per-ray slot rows; behind an exclusive scan of the kept counts and ONE host read (the sample count
    sizes the outputs) fsn_occ_gather_extras packs them: -> colors, opacity, depth, counts, (ray_indices int64 [N],
    t_starts [N], t_ends [N], extras {weights, alphas, trans, sigmas [N], rgbs [N,3]}) - render_rays' full return
    contract (rendering.py:88-107) from one launch + one gather."""
    if camera is not None:
        pose, cH, cW, cfocal, crow0, cnrows, dev = camera
        dev = torch.device(dev)
        R = int(cnrows) * int(cW)
        o = d = None
    else:
        o, d = _f32(rays_o, "rays_o").reshape(-1, 3), _f32(rays_d, "rays_d").reshape(-1, 3)
        R = o.shape[0]
        dev = o.device
    a = L.OccRenderArgs()
    a.R = R
    keep = [o, d]
    if camera is None:
        a.rays_o, a.rays_d = o.data_ptr(), d.data_ptr()
    else:
        pmx = pose.detach().to("cpu", torch.float32)[:3, :4].contiguous().reshape(-1).tolist()
        for i in range(12):
            a.cam_pose[i] = pmx[i]
        a.cam_H, a.cam_W, a.cam_row0, a.cam_focal = int(cH), int(cW), int(crow0), float(cfocal)
    for i in range(6):
        a.aabb[i] = float(aabb[i])
    a.res, a.levels, a.bits = int(res), int(levels), bits.data_ptr()
    a.near_plane, a.far_plane, a.step, a.max_steps = float(near_plane), float(far_plane), float(step), int(max_steps)
    if u is not None:
        u = _f32(u, "u").reshape(-1)
        if u.numel() != R:
            raise ValueError("u must hold one value per ray")
        keep.append(u)
        a.u = u.data_ptr()
    a.early_stop_eps, a.alpha_thre = float(early_stop_eps), float(alpha_thre)
    for name, m in (("pos_mask", pos_mask), ("dir_mask", dir_mask)):
        if m is not None:
            m = _f32(m, name)
            keep.append(m)
            setattr(a, name, m.data_ptr())
    for i in range(3):
        a.bkgd[i] = float(bkgd[i])
    colors = torch.empty(R, 3, device=dev)
    opacity = torch.empty(R, 1, device=dev)
    depth = torch.empty(R, 1, device=dev)
    a.colors, a.opacity, a.depth = colors.data_ptr(), opacity.data_ptr(), depth.data_ptr()
    counts = {}
    if want_counts or want_extras:
        counts = {"n_cand": torch.zeros(R, dtype=torch.int32, device=dev), "n_kept": torch.zeros(R, dtype=torch.int32, device=dev)}
        a.n_cand, a.n_kept = counts["n_cand"].data_ptr(), counts["n_kept"].data_ptr()
    slots = None
    if want_extras:
        cap = int(max_steps)
        slots = {k: torch.empty(max(R, 1), cap, device=dev) for k in ("t0", "weights", "alphas", "trans", "sigmas")}
        slots["rgbs"] = torch.empty(max(R, 1), cap, 3, device=dev)
        a.sample_t0, a.sample_cap = slots["t0"].data_ptr(), cap
        a.ex_weights, a.ex_alphas, a.ex_trans = slots["weights"].data_ptr(), slots["alphas"].data_ptr(), slots["trans"].data_ptr()
        a.ex_sigmas, a.ex_rgbs = slots["sigmas"].data_ptr(), slots["rgbs"].data_ptr()
    # (one queue counter PER CALL - ADVICE r3: a per-device buffer let two launches on different streams race on it; the
    # launch zeroes it itself)
    wc = torch.empty(1, dtype=torch.int64, device=dev)
    keep.append(wc)
    a.work_counter = wc.data_ptr()
    a.status = status_word(dev).data_ptr()
    with torch.cuda.device(dev):
        if launch_timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        L.check(L.lib().fsn_render_rays_occgrid(C.byref(pm.desc), pm.prec, _p(pm.blob), C.byref(a), _stream()),
                "fsn_render_rays_occgrid")
        if launch_timer is not None:
            e1.record()
            launch_timer.append((e0, e1))
        if want_extras:
            n_kept = counts["n_kept"]
            incl = torch.cumsum(n_kept, 0, dtype=torch.int64)
            N = int(incl[-1].item()) if R > 0 else 0  # the one host read: the sample count sizes the outputs
            ri = torch.empty(N, dtype=torch.int64, device=dev)
            ts, te = torch.empty(N, device=dev), torch.empty(N, device=dev)
            ex = {k: torch.empty(N, device=dev) for k in ("weights", "alphas", "trans", "sigmas")}
            ex["rgbs"] = torch.empty(N, 3, device=dev)
            if N > 0:
                order = ("weights", "alphas", "trans", "sigmas", "rgbs")
                sp = (C.c_void_p * 5)(*[slots[k].data_ptr() for k in order])
                op = (C.c_void_p * 5)(*[ex[k].data_ptr() for k in order])
                L.check(L.lib().fsn_occ_gather_extras(_p(n_kept), _p(incl - n_kept), _p(slots["t0"]), int(max_steps), R, float(step),
                                                      _p(ri), _p(ts), _p(te), sp, op, _stream()), "fsn_occ_gather_extras")
            return colors, opacity, depth, counts, (ri, ts, te, ex)
    return colors, opacity, depth, counts


def packed_visibility(sigmas: Tensor, t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor, n_rays: int,
                      early_stop_eps: float, alpha_thre: float) -> Tensor:
    sig, t0, t1 = _f32(sigmas, "sigmas").reshape(-1), _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends")
    N = sig.numel()
    keep = torch.zeros(N, device=sig.device, dtype=torch.uint8)
    with torch.cuda.device(sig.device):
        L.check(L.lib().fsn_packed_visibility(_p(sig), _p(t0), _p(t1), _p(_i64(ray_indices, "ray_indices")), N, int(n_rays),
                                              float(early_stop_eps), float(alpha_thre), _p(keep), _stream()),
                "fsn_packed_visibility")
    return keep.bool()


def occgrid_update(occs: Tensor, bits: Tensor, cells: Optional[Tensor], vals: Optional[Tensor], decay: float,
                   threshold: Optional[Tensor]) -> None:
    """occs[cells] = max(occs[cells]*decay, vals) (unique cells), then bits = occs > threshold (device scalar)."""
    n = 0 if cells is None else cells.numel()
    thr = None if threshold is None else _f32(threshold, "threshold").reshape(1)
    with torch.cuda.device(occs.device):
        L.check(L.lib().fsn_occgrid_update(_p(occs), occs.numel(), _p(cells), _p(None if vals is None else _f32(vals, "vals")),
                                           n, float(decay), _p(thr), _p(bits), _stream()), "fsn_occgrid_update")


def occgrid_select(bits: Tensor, aabb: Sequence[float], res: int, levels: int, lvl: int, all_cells: bool, n_uniform: int,
                   n_occupied: int, seed: int, scratch: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Cells of level `lvl` an update re-evaluates and a random point inside each (fsn_occgrid_select) ->
    (cells int64 [n] global indices, x [n,3]); n = res^3 (all_cells) or n_uniform + n_occupied.  No host sync."""
    res3 = res ** 3
    n = res3 if all_cells else int(n_uniform) + int(n_occupied)
    dev = bits.device
    cells = torch.empty(n, dtype=torch.int64, device=dev)
    x = torch.empty(n, 3, dtype=torch.float32, device=dev)
    if scratch is None and not all_cells and n_occupied > 0:
        scratch = torch.empty(res3 // 32 + 1, dtype=torch.int32, device=dev)
    ab = (C.c_float * 6)(*[float(v) for v in aabb])
    with torch.cuda.device(dev):
        L.check(L.lib().fsn_occgrid_select(_p(bits), int(res), int(levels), int(lvl), ab, 1 if all_cells else 0, int(n_uniform),
                                           int(n_occupied), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), _p(scratch), _p(cells),
                                           _p(x), _stream()), "fsn_occgrid_select")
    return cells, x


def occgrid_update_multi(occs: Tensor, pending: Tensor, cells: Tensor, vals: Tensor, decay: float) -> None:
    """occs[c] = max(occs[c]*decay, max of the vals drawn for c); `cells` may repeat (fsn_occgrid_update_multi)."""
    n = cells.numel()
    with torch.cuda.device(occs.device):
        L.check(L.lib().fsn_occgrid_update_multi(_p(occs), occs.numel(), _p(pending), _p(cells), _p(_f32(vals, "vals").reshape(-1)),
                                                 n, float(decay), _stream()), "fsn_occgrid_update_multi")
