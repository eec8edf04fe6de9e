"""Host-side mirror of `src/core/loss.py`: the occlusion regulariser as one
segmented reduction on the GPU instead of a Python loop with one kernel launch per ray; and the weight-norm
"frequency" regulariser of the training loop (row f1, src/run-nerf.py:266-279) as one reduction over the flat
parameter arena."""
import ctypes as C

import torch
from torch import Tensor

from .. import _lib as L
from .. import ops


class OcclusionRegularizer:
    """Penalises density near the camera: mean over rays of sum_i w(t_i) sigma_i (loss.py:26-42),
    w = -a t + b ('linear') or a exp(-b t) ('exp') (loss.py:44-60).  Differentiable w.r.t. sigmas
    (the reference adds it to the training loss, run-nerf.py:264)."""

    def __init__(self, a: float, b: float, func: str = "linear"):
        assert a >= 0, "a should be non-negative"
        assert b >= 0, "b should be non-negative"
        self.a, self.b, self.func = a, b, func

    def __call__(self, sigmas: Tensor, t_vals: Tensor, ray_idxs: Tensor) -> Tensor:
        return ops.occlusion_reg(sigmas, t_vals, ray_idxs, self.a, self.b, self.func)


class _WeightNormFn(torch.autograd.Function):
    """out = sum over the selected tensors of |w|_1 or |w|_2, one reduction launch over the flat parameter arena
    (`fsn_weight_norm_fwd`); backward accumulates d_out * d(out)/dw into a gradient tensor of the arena's layout
    (`fsn_weight_norm_bwd`) and hands each selected parameter its slice."""

    @staticmethod
    def forward(ctx, reg, *params):
        flat, offs, lens = reg._arena(params)
        n = len(offs)
        off_t, len_t = (C.c_int64 * n)(*offs), (C.c_int64 * n)(*lens)
        nws = L.lib().fsn_weight_norm_workspace_floats(n, len_t)
        if nws < 0:
            L.check(int(nws), "fsn_weight_norm_workspace_floats")
        ws = torch.empty(int(nws), device=flat.device, dtype=torch.float32)
        out = torch.empty(1, device=flat.device, dtype=torch.float32)
        with torch.cuda.device(flat.device):
            L.check(L.lib().fsn_weight_norm_fwd(ops._p(flat), n, off_t, len_t, int(reg.l2), ops._p(ws), ops._p(out),
                                                ops._stream()), "fsn_weight_norm_fwd")
        ctx.reg, ctx.flat, ctx.ws, ctx.tabs, ctx.shapes = reg, flat, ws, (n, off_t, len_t, offs, lens), [p.shape for p in params]
        return out.reshape(())

    @staticmethod
    def backward(ctx, d_out):
        n, off_t, len_t, offs, lens = ctx.tabs
        flat = ctx.flat
        g = torch.zeros_like(flat)
        d = d_out.reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(flat.device):
            L.check(L.lib().fsn_weight_norm_bwd(ops._p(flat), n, off_t, len_t, int(ctx.reg.l2), ops._p(ctx.ws), ops._p(d),
                                                ops._p(g), ops._stream()), "fsn_weight_norm_bwd")
        return (None,) + tuple(g[o:o + ln].view(sh) for o, ln, sh in zip(offs, lens, ctx.shapes))


class WeightNormRegularizer:
    """The weight-norm "frequency" regulariser the reference writes inline in its training loop
    (src/run-nerf.py:266-279, flags src/utils/parser.py:141-156): over the parameters whose name contains "weight"
    and whose first dimension exceeds 3 (every Linear weight except the 1- and 3-row heads),
        reg "l1":  sum_t |W_t|_1          reg "l2":  sum_t |W_t|_2   (torch.square(p).sum().sqrt() per tensor).
    `reg()` returns that scalar with gradients to the parameters; the caller multiplies by alpha and applies the
    schedule (`active(k)`: k < int(reg_ratio * Td), run-nerf.py:270-271).  One reduction launch over the flat parameter
    arena when the parameters live in one (core/optim.py: FlatParams / FusedAdam); otherwise they are gathered into a
    temporary flat buffer first."""

    def __init__(self, named_parameters, reg: str = "l1", reg_ratio: float = 0.5, Td: int = 0):
        assert reg in ("l1", "l2")
        self.l2 = reg == "l2"
        self.params = [p for name, p in named_parameters if "weight" in name and p.shape[0] > 3]
        self.Ts = int(reg_ratio * Td)

    def active(self, k: int) -> bool:
        return k < self.Ts

    def _arena(self, params):
        """(flat tensor, offsets, lengths): the parameters' own storage when they are views of one contiguous buffer."""
        base = min(p.data_ptr() for p in params)
        offs = [(p.data_ptr() - base) // 4 for p in params]
        lens = [p.numel() for p in params]
        span = max(o + n for o, n in zip(offs, lens))
        st = params[0].untyped_storage()
        same = all(p.is_contiguous() and p.untyped_storage().data_ptr() == st.data_ptr() for p in params)
        if same and span * 4 <= st.nbytes():
            p0 = min(params, key=lambda p: p.data_ptr())
            flat = torch.as_strided(p0.detach(), (span,), (1,), p0.storage_offset())
            return flat, offs, lens
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        offs, o = [], 0
        for n in lens:
            offs.append(o)
            o += n
        return flat, offs, lens

    def __call__(self) -> Tensor:
        if not self.params:
            raise ValueError("WeightNormRegularizer: no weight tensor selected")
        if not all(p.is_cuda for p in self.params):
            raise RuntimeError("WeightNormRegularizer: GPU parameters expected (the HIP path has no CPU fallback)")
        return _WeightNormFn.apply(self, *self.params)
