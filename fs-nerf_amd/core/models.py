"""Host-side mirror of the reference's `core/models.py` call surface, backed by HIP kernels.

  PositionalEncoder(d_input, n_freqs, log_space=False).forward(x)   (src/core/models.py:10-50)
  NeRF(d_pos, d_dir, n_layers, d_hidden, skip, pos_fn=..., dir_fn=...).forward(x, dirs=None)
                                                                      (src/core/models.py:53-143)
Parameters are ordinary nn.Linear modules under the reference's attribute names, so a reference
`state_dict()` loads unchanged (keys layers.N.*, sigma.*, connection.*, branch.*, rgb.*).
`forward` never touches them with torch ops: they are packed into the MFMA streaming layout
(re-packed whenever a parameter's version counter changes) and the whole network runs in the
fused kernel `fsn_mlp_fwd`.  In training mode with autograd enabled, `forward(x, dirs)` runs the training pair
instead (`_NerfTrainFn`: `fsn_nerf_train_fwd` / `_bwd`, hand-written MFMA forward-with-savers, dgrad chain and
split-K wgrad kernels in the same precision mode).  Additions over the reference: an optional frequency mask
(`set_freq_mask`) and the arithmetic mode `precision`:
  "fp16x3" (default) three fp16 MFMA passes on high / scaled-low parts with a separate correction accumulator, fp32
            accumulate: fp32-class accuracy (the 1e-4 parity mode) for hidden-layer scales from 2^-14 to 65504.  The
            kernels raise a device flag at either end (a value reached fp16 infinity; a layer whose largest activation
            over a wavefront is an fp16 subnormal); `forward` / `render_rays` then re-run the call in "bf16x3" and
            keep that mode (a RuntimeWarning is issued) - never silent, inf/NaN are never returned;
  "bf16x3"  the same split on bf16 parts: no range limit, ~1e-5 per product;
  "fp16x2"  two passes (weights high part only): measured accuracy in DESIGN.md, not a parity mode;
  "bf16" / "fp16"  one pass (BASELINE config 5), tolerance stated in the tests.
"""
import warnings
from typing import Optional, Sequence, Tuple

import torch
from torch import Tensor, nn

from .. import _lib as L
from .. import ops


def _freqs(n_freqs: int, log_space: bool) -> Tensor:
    # same expressions as the reference (models.py:31-34), evaluated by torch on the CPU once
    if log_space:
        return 2.0 ** torch.linspace(0.0, n_freqs - 1, n_freqs)
    return torch.linspace(2.0 ** 0.0, 2.0 ** (n_freqs - 1), n_freqs)


class PositionalEncoder(nn.Module):
    def __init__(self, d_input: int, n_freqs: int, log_space: bool = False):
        super().__init__()
        self.d_input, self.n_freqs, self.log_space = d_input, n_freqs, log_space
        self.d_output = d_input * (1 + 2 * n_freqs)
        self.freqs = [float(f) for f in _freqs(n_freqs, log_space)] if n_freqs > 0 else []

    def forward(self, x: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        return ops.posenc(x, self.freqs, mask)


class _NerfTrainFn(torch.autograd.Function):
    """NeRF.forward with gradients.  `fsn_nerf_train_fwd` runs the MFMA kernel of the inference
    path with the fp32 activations saved; `fsn_nerf_train_bwd` the dgrad chain + wgrad GEMMs on the matrix cores
    (csrc/train_fused.hip), in the model's precision mode.  Gradients flow to the parameters only (sample positions /
    directions need none on this path).  (The plain-fp32 library-GEMM formulation these kernels were first checked
    against lives entirely under tests/ref_fp32, binding included; nothing in this package can reach it.)"""

    @staticmethod
    def forward(ctx, model, x, dirs, *params):
        """`x` / `dirs` [..., 3] tensors, or - ray form - x = (rays_o, rays_d, ray_indices, t_starts, t_ends), dirs = None:
        the samples are the midpoints of the packed intervals, formed inside the launch (no gathered [N,3] tensors)."""
        n = len(params) // 2
        weights, biases = params[:n], params[n:]
        desc = ops.make_desc(model.n_layers, model.d_hidden, model.skip, model.pos_encoder.freqs,
                             model.dir_encoder.freqs)
        ray_form = isinstance(x, tuple)
        dev = x[0].device if ray_form else x.device
        prec = model.train_prec()
        # range-guard word of THIS call (one network, one step): the kernels of its forward and backward report into
        # it, and only they consult it - a flag raised by another network or by an inference launch must not zero
        # this network's gradients (the sticky per-device word of the inference path is not used here)
        word = torch.zeros(1, dtype=torch.int32, device=dev) if model.fp16_family(prec) else None
        if ray_form:
            out, work = ops.nerf_train_fwd_rays(desc, prec, weights, biases, *x, model._mask(model.pos_mask, dev),
                                                model._mask(model.dir_mask, dev), status=word)
        else:
            out, work = ops.nerf_train_fwd(desc, prec, weights, biases, x, dirs, model._mask(model.pos_mask, dev),
                                           model._mask(model.dir_mask, dev), status=word)
        ctx.desc, ctx.prec, ctx.work, ctx.out, ctx.model, ctx.word = desc, prec, work, out, model, word
        ctx.weights = [w.detach() for w in weights]
        ctx.params = params  # (the leaf Parameters themselves: backward may accumulate into their .grad buffers)
        return out if ray_form else out.reshape(*x.shape[:-1], 4)

    @staticmethod
    def backward(ctx, d_out):
        if ctx.work is None:
            raise RuntimeError("NeRF backward ran twice on one forward: the saved activations are released after the "
                               "first pass (retain_graph is not supported on this path)")
        # Gradient buffers owned by a flat bucket (shard.FlatGrads marks its parameters): the kernels ADD into p.grad on
        # the device - exactly what autograd's AccumulateGrad would do with returned tensors, minus 24 temporaries and 24
        # add launches per step - and autograd is handed None for them.  Anything else gets the gradients returned.
        n = len(ctx.params) // 2
        sink = all(getattr(p, "_fsn_grad_sink", False) and p.requires_grad and p.grad is not None and p.grad.is_contiguous()
                   and p.grad.dtype == torch.float32 and p.grad.device == d_out.device for p in ctx.params)
        into = ([p.grad for p in ctx.params[:n]], [p.grad for p in ctx.params[n:]]) if sink else None
        model = ctx.model
        d_out = d_out.contiguous()
        stage = model._bwd_stage_state(d_out.device) if model.fp16_family(ctx.prec) else None
        if stage is not None and not model._bwd_calibrated:
            # first backward of this model in an fp16 mode: one throw-away pass to measure the per-stage gradient
            # magnitudes (delayed scaling needs a previous call; its own status word, scratch outputs)
            ops.nerf_train_bwd(ctx.desc, ctx.prec, ctx.weights, ctx.work, ctx.out, d_out,
                               status=torch.zeros_like(ctx.word) if ctx.word is not None else None, stage_state=stage)
            model._bwd_calibrated = True
        dW, db = ops.nerf_train_bwd(ctx.desc, ctx.prec, ctx.weights, ctx.work, ctx.out, d_out, status=ctx.word,
                                    into=into, stage_state=stage)
        ctx.work = None
        ctx.params = None
        # fp16 range guard without a per-step host sync.  When this call's launches reported values outside the fp16
        # range, the backward kernels have written ITS gradients as zeros on the device; the call's word is folded
        # (device ops) into the device's step flag, on which FusedAdam skips the whole update - a skipped step, like a
        # loss scaler's - and into this model's own accumulated word, which the host reads every `range_check_every`
        # training calls; it then warns and continues in bf16x3.
        if ctx.word is not None:
            ops.step_flag(d_out.device).bitwise_or_(ctx.word)
            model._train_status(d_out.device).bitwise_or_(ctx.word)
            model._train_calls += 1
            if model.range_check and model._train_calls % model.range_check_every == 0:
                model._guard_weights("training steps")
                bits = model._read_train_status(d_out.device)
                if bits:
                    model.fall_back("training steps (an overflowing step's gradients were zeroed on the device and its "
                                    "optimizer update skipped)", bits)
        if sink:
            return (None, None, None) + (None,) * (2 * n)
        db = [g.reshape(-1) for g in db]
        return (None, None, None, *dW, *db)


class NeRF(nn.Module):
    PRECISIONS = {"bf16x3": L.FSN_PREC_BF16X3, "bf16": L.FSN_PREC_BF16, "fp16x3": L.FSN_PREC_FP16X3,
                  "fp16": L.FSN_PREC_FP16, "fp16x2": L.FSN_PREC_FP16X2}
    FALLBACK = {"fp16x3": "bf16x3", "fp16x2": "bf16x3", "fp16": "bf16"}  # same pass structure, float32 range

    def __init__(self, d_pos: int = 3, d_dir: int = 3, n_layers: int = 8, d_hidden: int = 256,
                 skip: Tuple[int, ...] = (4,), precision: str = "fp16x3", **kwargs) -> None:
        super().__init__()
        if d_pos != 3 or d_dir != 3:
            raise ValueError("the HIP path is built for 3-D positions and directions")
        self.d_pos, self.d_dir, self.n_layers, self.d_hidden = d_pos, d_dir, n_layers, d_hidden
        self.skip = tuple(skip)
        self.pos_encoder = PositionalEncoder(d_pos, kwargs["pos_fn"]["n_freqs"], kwargs["pos_fn"]["log_space"])
        self.dir_encoder = PositionalEncoder(d_dir, kwargs["dir_fn"]["n_freqs"], kwargs["dir_fn"]["log_space"])
        d_pe, d_de = self.pos_encoder.d_output, self.dir_encoder.d_output
        # construction order follows the reference (models.py:96-108) so that seeding
        # torch's RNG reproduces its initial parameters
        hidden = [nn.Linear(d_hidden + d_pe, d_hidden) if i in self.skip else nn.Linear(d_hidden, d_hidden)
                  for i in range(n_layers - 1)]
        self.layers = nn.ModuleList([nn.Linear(d_pe, d_hidden)] + hidden)
        self.sigma = nn.Linear(d_hidden, 1)
        self.connection = nn.Linear(d_hidden, d_hidden)
        self.branch = nn.Linear(d_hidden + d_de, d_hidden // 2)
        self.rgb = nn.Linear(d_hidden // 2, 3)
        self.precision = precision
        # fp16 modes: True = read the kernels' range flag back after each call (one host sync; a flagged call is re-run in
        # bf16x3 before it returns); "deferred" = asynchronous read-back, looked at by the NEXT call (batch rendering in
        # small launches: no wait per call; render_frame checks once per frame); False = never look
        self.range_check = True
        self.range_check_every = 16  # ... in training: every that many steps (gradients are guarded on the device)
        self._train_calls = 0
        self._pack_calls = 0
        self.weight_check = True     # fp16 modes: look at the layers' largest weights when (re-)packing (_guard_weights)
        self._train_word: Optional[Tensor] = None
        self._bwd_stage = None       # (scales, maxima) of the backward's delayed per-stage gradient scaling (fp16 modes)
        self._bwd_calibrated = False
        self.grad_overflow_looks = 0  # host looks that found a step skipped by the delayed gradient scaling
        self.train_precision: Optional[str] = None  # None: same mode as `precision`
        self.pos_mask: Optional[Tensor] = None
        self.dir_mask: Optional[Tensor] = None
        self._packed = None
        self._packed_key = None

    # -- additions -----------------------------------------------------------------
    def set_freq_mask(self, pos_mask: Optional[Tensor], dir_mask: Optional[Tensor] = None) -> None:
        """Frequency mask multiplied onto the encoded features ([d_pos*(1+2n)], [d_dir*(1+2m)])."""
        self.pos_mask, self.dir_mask = pos_mask, dir_mask

    def train_prec(self) -> int:
        tp = self.train_precision or self.precision
        if tp == "fp16x2":
            tp = "fp16x3"  # the two-pass mode is inference only
        return self.PRECISIONS[tp]

    @staticmethod
    def fp16_family(prec: int) -> bool:
        return prec in (L.FSN_PREC_FP16X3, L.FSN_PREC_FP16, L.FSN_PREC_FP16X2)

    def _bwd_stage_state(self, dev):
        """(scales, maxima) of the backward chain's delayed per-stage scaling (fsn_nerf_train_bwd), device-resident."""
        st = self._bwd_stage
        if st is None or st[0].device != dev:
            n = self.n_layers + 2
            st = self._bwd_stage = (torch.ones(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
            self._bwd_calibrated = False
        return st

    def _train_status(self, dev) -> Tensor:
        """This model's accumulated training range word (device; OR of its calls' words since the last host look)."""
        w = self._train_word
        if w is None or w.device != dev:
            w = self._train_word = torch.zeros(1, dtype=torch.int32, device=dev)
        return w

    def _read_train_status(self, dev) -> int:
        """Host look at the accumulated word (one 4-byte read-back) and clear.  Data-parallel runs take the MAX over
        the ranks first (every rank makes the same number of training calls, so all of them are here together): a
        rank that overflowed and one that did not must not continue in different arithmetic."""
        w = self._train_status(dev)
        from ..shard import max_bits_over_ranks
        local = int(w.item())
        if local & L.FSN_STATUS_GRAD_RANGE:  # delayed gradient scaling skipped a step (and lowered a stage): no fall-back
            self.grad_overflow_looks += 1
        # (MAX of bit masks 0..3: both bits lead to the same fall-back, so the largest mask is enough)
        bits = max_bits_over_ranks(local & (L.FSN_STATUS_FP16_RANGE | L.FSN_STATUS_FP16_SMALL), dev)
        if local or bits:
            w.zero_()
        return bits & (L.FSN_STATUS_FP16_RANGE | L.FSN_STATUS_FP16_SMALL)

    def fall_back(self, what: str, bits: int = L.FSN_STATUS_FP16_RANGE) -> None:
        """An fp16-mode launch reported values outside the mode's envelope (|v| >= 65504, or a layer whose activations
        all sat below 2^-14): continue in the bf16 mode of the same pass structure, which has float32's range.  Never
        silent."""
        new = self.FALLBACK.get(self.precision, "bf16x3")
        warnings.warn(f"fs-nerf HIP path: {what}: hidden activations left the fp16 range envelope of precision "
                      f"'{self.precision}' ({ops.describe_flags(bits)}); re-running / continuing in '{new}'",
                      RuntimeWarning, stacklevel=3)
        self.precision = new
        if self.train_precision in self.FALLBACK:
            self.train_precision = None

    def _tensors(self):
        mods = list(self.layers) + [self.sigma, self.connection, self.branch, self.rgb]
        return [m.weight for m in mods], [m.bias for m in mods]

    def _weights_below_envelope(self) -> bool:
        """fp16 modes: True when some layer's LARGEST weight is below 2^-20.  High parts under 2^-14 are fp16 subnormals
        and the scaled low parts resolve 2^-35: a layer whose largest weight is M keeps 2^-35 / M relative to it -
        2^-18 at the 6e-6 the envelope tests run (fine against the 1e-4 bar), 2^-15 at 2^-20, where this check draws the
        line.  (The kernels' low-end guard looks at activations; tiny weights under large activations would pass it.)
        One small reduction per layer and one 4-byte read-back: on the first pack and every `range_check_every`-th
        re-pack / training call."""
        mods = list(self.layers) + [self.connection, self.branch]  # (the sigma / rgb heads are float32 dot products)
        with torch.no_grad():
            m = torch.stack([mod.weight.detach().abs().amax() for mod in mods])
            return bool(((m > 0) & (m < 2.0 ** -20)).any())

    def _guard_weights(self, what: str) -> None:
        if self.range_check and self.weight_check and self.fp16_family(self.PRECISIONS[self.precision]) and \
                self._weights_below_envelope():
            self.fall_back(what + " (a layer's largest weight is below 2^-20)", L.FSN_STATUS_FP16_SMALL)

    def packed(self) -> ops.PackedMLP:
        """Weights in the MFMA streaming layout; re-packed when any parameter changed."""
        ws, bs = self._tensors()
        key = (self.precision, ws[0].device) + tuple((t.data_ptr(), t._version) for t in ws + bs)
        if self._packed is None or key != self._packed_key:
            if self._pack_calls % max(1, int(self.range_check_every)) == 0:
                self._guard_weights("NeRF.packed")
                key = (self.precision,) + key[1:]
            self._pack_calls += 1
            desc = ops.make_desc(self.n_layers, self.d_hidden, self.skip, self.pos_encoder.freqs,
                                 self.dir_encoder.freqs)
            if self._packed is None or self._packed.prec != self.PRECISIONS[self.precision] \
                    or self._packed.blob.device != ws[0].device:
                self._packed = ops.PackedMLP(desc, self.PRECISIONS[self.precision], ws[0].device)
            self._packed.pack(ws, bs)
            self._packed_key = key
        return self._packed

    def _mask(self, m: Optional[Tensor], dev) -> Optional[Tensor]:
        return None if m is None else m.to(dev, torch.float32)

    def forward_rays(self, rays_o: Tensor, rays_d: Tensor, ray_indices: Tensor, t_starts: Tensor, t_ends: Tensor,
                     full: bool = True) -> Tensor:
        """`self(x, d)` (full) or `self(x)` (density only) on the midpoints of packed intervals - what the reference's
        closures compute as `to = rays_o[ri]; td = rays_d[ri]; x = to + td * (t0 + t1)[:, None] / 2; model(x, td)`
        (rendering.py:58-64, 76-84) - with the gathers and the midpoint arithmetic inside the launch: no [N,3] tensor
        is materialised.  Same values bit for bit; gradients (training mode, full pass) flow to the parameters."""
        if full and self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            ws, bs = self._tensors()
            return _NerfTrainFn.apply(self, (rays_o, rays_d, ray_indices, t_starts, t_ends), None, *ws, *bs)
        dev = rays_o.device
        args = (rays_o, rays_d, ray_indices, t_starts, t_ends, full, self._mask(self.pos_mask, dev), self._mask(self.dir_mask, dev))
        out = ops.mlp_fwd_rays(self.packed(), *args)
        bits = ops.range_flags(dev) if self.range_check is True and self.fp16_family(self.PRECISIONS[self.precision]) else 0
        if bits:
            self.fall_back("NeRF.forward_rays", bits)
            out = ops.mlp_fwd_rays(self.packed(), *args)
        return out

    # -- reference surface ---------------------------------------------------------
    def forward(self, x: Tensor, dirs: Optional[Tensor] = None) -> Tensor:
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if dirs is None:
                raise NotImplementedError(
                    "NeRF.forward(x) with gradients: the density-only pass has no backward (the reference runs it "
                    "under torch.no_grad(), rendering.py:58-64); call it under torch.no_grad()")
            ws, bs = self._tensors()
            return _NerfTrainFn.apply(self, x, dirs, *ws, *bs)
        dev = x.device
        out = ops.mlp_fwd(self.packed(), x, dirs, self._mask(self.pos_mask, dev), self._mask(self.dir_mask, dev))
        bits = ops.range_flags(dev) if self.range_check and self.fp16_family(self.PRECISIONS[self.precision]) else 0
        if bits:
            self.fall_back("NeRF.forward", bits)
            out = ops.mlp_fwd(self.packed(), x, dirs, self._mask(self.pos_mask, dev), self._mask(self.dir_mask, dev))
        return out
