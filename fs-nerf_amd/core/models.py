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
  "fp16x3" (default) three fp16 MFMA passes on high / low parts, fp32 accumulate: fp32-class accuracy (the 1e-4 parity
            mode).  INFERENCE (round 4, `act_scaling = True`) runs the SCALED network in FSN_PREC_FP16X3U: a power of two
            per layer, calibrated from the layers' measured activation maxima on a probe batch (`calibrate`) and folded
            into the packed weights / biases / heads - an exact transformation - keeps every layer's activations at
            2^4 .. 2^10 whatever the network's own scale, so the plain split (unscaled low parts, one accumulator, no
            merge) is float32-grade and 6 % faster.  The kernels raise a device flag when the calibration no longer fits
            the data at either end (a value reached fp16 infinity; a wavefront's layer maximum below 2^-4): the host
            re-calibrates and re-runs the call (never silent, inf/NaN are never returned); only when that does not help
            it continues in "bf16x3" with a RuntimeWarning.  TRAINING (forward-with-savers / backward / weight
            gradients) keeps round 3's arithmetic (FSN_PREC_FP16X3: low parts scaled by 2^11, own correction accumulator:
            float32-grade from 2^-14 to 65504 with no calibration), as does inference with `act_scaling = False`;
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
        if stage is not None and not model._bwd_calibrated and d_out.shape[0] > 0:
            # first backward of this model in an fp16 mode: one throw-away pass to measure the per-stage gradient
            # magnitudes (delayed scaling needs a previous call; its own status word, scratch outputs).  A call without
            # samples - the all-background first batch of an empty occupancy grid - launches nothing and calibrates
            # nothing: the next one with samples does.
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
            for f in model._step_flags(d_out.device):
                f.bitwise_or_(ctx.word)
            model._train_status(d_out.device)[0:1].bitwise_or_(ctx.word)
            model._train_calls += 1
            if model.range_check and model._train_calls % model.range_check_every == 0:
                model._guard_weights("training steps", training=True)
                bits, sampler_bits = model._read_train_status(d_out.device)
                if bits:
                    model.fall_back("training steps (an overflowing step's gradients were zeroed on the device and its "
                                    "optimizer update skipped)", bits, scaled=False)
                # The sampler's density pass in front of the training forward is an INFERENCE launch (scaled fp16x3):
                # a flag there says its per-layer scales no longer fit the weights being trained (the step was skipped
                # on the device); the next sampler call re-calibrates on its own rays.  Also every 4th look, flag or
                # not, so that the scales follow the network long before a guard trips.
                if sampler_bits or (model._train_calls // model.range_check_every) % 4 == 0:
                    model._needs_calibration = True
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
        # OPT-IN, not a parity mode: arithmetic of the density pass behind the occupancy estimator's visibility cull when
        # it runs as the sampler in front of a training forward (None = the model's inference mode; "bf16" = one pass).
        # That pass only decides which marched samples are KEPT (transmittance >= early_stop_eps); the kept samples are
        # then evaluated, integrated and differentiated in the model's own mode.  See `packed_cull`.
        self.cull_precision: Optional[str] = None
        self._packed_cull = None
        self._packed_cull_key = None
        # per-layer activation scaling of the fp16x3 inference path (module docstring)
        self.act_scaling = True
        self.act_target_exp = self.ACT_TARGET_EXP
        self._act_exps: Optional[list] = None   # n_layers + 2 exponents (kernel GEMM order), None = not calibrated
        self._calib_key = None                  # parameter versions the calibration was made with
        self._calib_moves = 0                   # target moves since then (same weights, data did not fit)
        self._target_moved_by = 0               # ... and their sum: undone when the weights change
        self._needs_calibration = False
        self._calib_blob = None
        self.calibrations = 0                   # statistics: calibration launches / range events (fall-backs and
        self.range_events = 0                   # re-calibrations) of this model

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

    def _step_flags(self, dev):
        """The step-flag words a training launch of this model reports into (device int32 [1] each): the word of every
        gradient bucket that owns one of its parameters (`shard.FlatGrads.step_flag`, consumed by that bucket's
        optimizer), and the device's shared word (`ops.step_flag`) for parameters no bucket owns."""
        flags, loose = {}, False
        for p in self.parameters():
            f = getattr(p, "_fsn_step_flag", None)
            if f is None or f.device != dev:
                loose = True
            else:
                flags[id(f)] = f
        out = list(flags.values())
        if loose or not out:
            out.append(ops.step_flag(dev))
        return out

    def _train_status(self, dev) -> Tensor:
        """This model's accumulated training range words (device int32 [2]; OR of its calls' words since the last host
        look): [0] the training kernels' (forward-with-savers / backward), [1] the sampler's density pass (an inference
        launch in front of the training forward, rendering.py)."""
        w = self._train_word
        if w is None or w.device != dev:
            w = self._train_word = torch.zeros(2, dtype=torch.int32, device=dev)
        return w

    def _read_train_status(self, dev) -> Tuple[int, int]:
        """Host look at the accumulated words (one 8-byte read-back) and clear -> (training kernels' bits, sampler's
        bits).  Data-parallel runs take the MAX over the ranks first (every rank makes the same number of training
        calls, so all of them are here together): a rank that overflowed and one that did not must not continue in
        different arithmetic."""
        w = self._train_status(dev)
        from ..shard import max_bits_over_ranks
        local, samp = (int(v) for v in w.tolist())
        if local & L.FSN_STATUS_GRAD_RANGE:  # delayed gradient scaling skipped a step (and lowered a stage): no fall-back
            self.grad_overflow_looks += 1
        env = L.FSN_STATUS_FP16_RANGE | L.FSN_STATUS_FP16_SMALL
        # (MAX of bit masks 0..3: both bits lead to the same action, so the largest mask is enough; the two words are
        # reduced separately - a MAX over a combined mask would lose the smaller word's bits)
        bits = max_bits_over_ranks(local & env, dev)
        sbits = max_bits_over_ranks(samp & env, dev)
        if local or samp or bits or sbits:
            w.zero_()
        return bits & env, sbits & env

    def fall_back(self, what: str, bits: int = L.FSN_STATUS_FP16_RANGE, probe=None, earlier_invalid: bool = False,
                  scaled: bool = True) -> None:
        """An fp16-mode launch reported values outside the mode's envelope; the caller re-runs the call afterwards.
        Scaled inference (`infer_prec() == FSN_PREC_FP16X3U`, and `scaled`: the report came from an inference launch):
        the calibration did not fit the data - re-calibrate on `probe` (`_recalibrate`) and stay in fp16x3.  Otherwise,
        or when that does not help: continue in the bf16 mode of the same pass structure, which has float32's range,
        with a RuntimeWarning.  Never silent: `earlier_invalid` (deferred range checks) warns in the re-calibrated case
        too, because results already handed out were invalid."""
        self.range_events += 1
        if scaled and self.infer_prec() == L.FSN_PREC_FP16X3U and self._recalibrate(bits, probe):
            if earlier_invalid:
                warnings.warn(f"fs-nerf HIP path: {what}: the per-layer activation scales did not fit the data "
                              f"({ops.describe_flags(bits)}); re-calibrated", RuntimeWarning, stacklevel=3)
            return
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

    # -- per-layer activation scaling (fp16x3 inference) ---------------------------------
    ACT_TARGET_EXP = 10  # a layer's largest probe activation is scaled into (2^9, 2^10]: 2^6 of headroom to 65504
    PROBE_MAX = 16384    # probe samples per calibration launch

    def infer_prec(self) -> int:
        """Arithmetic mode of the INFERENCE kernels (fsn_mlp_fwd, fsn_render_rays_fused, fsn_render_rays_occgrid)."""
        if self.precision == "fp16x3" and self.act_scaling:
            return L.FSN_PREC_FP16X3U
        return self.PRECISIONS[self.precision]

    def _param_key(self):
        ws, bs = self._tensors()
        return tuple((t.data_ptr(), t._version) for t in ws + bs)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._act_exps = None  # other weights: calibrate again on the next inference call
        self._forget_target_moves()
        return out

    def _forget_target_moves(self) -> None:
        # target moves answered a probe that did not represent the data under the OLD weights
        self.act_target_exp -= self._target_moved_by
        self._target_moved_by = self._calib_moves = 0

    def _default_probe(self, dev):
        g = torch.Generator(device="cpu").manual_seed(0)
        x = (torch.rand(4096, 3, generator=g) * 4.0 - 2.0).to(dev)
        d = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1).to(dev)
        return x, d

    @torch.no_grad()
    def calibrate(self, probe=None, keep_if_close: bool = False) -> list:
        """Measure every layer's largest |activation| on a probe batch and choose the power-of-two scales of the
        fp16x3 inference path.  `probe`: (x [n,3], dirs [n,3]) tensors or a callable returning them - sample positions
        and directions like those the network is evaluated on (the call sites build them from their rays); None = a
        fixed default (uniform in [-2,2]^3, random unit directions).  One bf16x3 launch of the UNSCALED network on at
        most PROBE_MAX samples (`fsn_mlp_layer_maxima`) and one small read-back.  `keep_if_close`: an existing exponent
        is kept while it still puts the layer's maximum within [2^-3, 2^1] of the target (periodic re-calibration
        during training: no flapping between neighbouring powers of two).  -> the exponents."""
        import math
        ws, bs = self._tensors()
        dev = ws[0].device
        if callable(probe):
            probe = probe()
        x, d = probe if probe is not None else self._default_probe(dev)
        x, d = x.reshape(-1, 3).to(dev, torch.float32), d.reshape(-1, 3).to(dev, torch.float32)
        if x.shape[0] == 0:
            x, d = self._default_probe(dev)
        if x.shape[0] > self.PROBE_MAX:
            idx = torch.linspace(0, x.shape[0] - 1, self.PROBE_MAX, device=dev).long()
            x, d = x[idx], d[idx]
        desc = ops.make_desc(self.n_layers, self.d_hidden, self.skip, self.pos_encoder.freqs, self.dir_encoder.freqs)
        if self._calib_blob is None or self._calib_blob.blob.device != dev:
            self._calib_blob = ops.PackedMLP(desc, L.FSN_PREC_BF16X3, dev)
        self._calib_blob.pack(ws, bs)
        maxima = ops.mlp_layer_maxima(self._calib_blob, x.contiguous(), d.contiguous(), self._mask(self.pos_mask, dev),
                                      self._mask(self.dir_mask, dev))
        # largest |weight| per GEMM, activation columns and encoding columns separately (the scaled weights must stay
        # inside fp16 too): one stacked reduction, read back together with the maxima
        D, n = self.d_hidden, self.n_layers
        mods = list(self.layers) + [self.connection, self.branch]
        wmax = []
        for g, m in enumerate(mods):
            w = m.weight.detach().abs()
            na = 0 if g == 0 else D
            wmax.append(w[:, :na].amax() if na else w.new_zeros(()))
            wmax.append(w[:, na:].amax() if w.shape[1] > na else w.new_zeros(()))
        host = torch.cat([maxima, torch.stack(wmax).to(torch.float32)]).cpu().tolist()  # the one host sync
        amax, wmx = host[:n + 2], host[n + 2:]
        T = int(self.act_target_exp)
        old = self._act_exps
        exps = []
        for g, v in enumerate(amax):
            if not math.isfinite(v):
                raise RuntimeError(f"NeRF.calibrate: layer {g} produced non-finite activations on the probe batch")
            if v <= 0.0:
                e = old[g] if old is not None else 0  # dead layer on the probe: its scale does not matter
            else:
                e = T - math.ceil(math.log2(v))
                if keep_if_close and old is not None and 2.0 ** (T - 3) <= v * 2.0 ** old[g] <= 2.0 ** (T + 1):
                    e = old[g]
            exps.append(max(-60, min(60, e)))
        # scaled weights: W_g * 2^(e_g - e_prev) on activation columns, W_g * 2^e_g on encoding columns, below 2^15
        for g in range(n + 2):
            prev = 0 if g == 0 else (exps[n - 1] if g == n else (exps[n] if g == n + 1 else exps[g - 1]))
            for wm, sh in ((wmx[2 * g], exps[g] - prev), (wmx[2 * g + 1], exps[g])):
                if wm > 0.0 and math.isfinite(wm):
                    over = math.ceil(math.log2(wm)) + sh - 15
                    if over > 0:
                        exps[g] -= over
        self._act_exps = exps
        self._calib_key = self._param_key()
        self._needs_calibration = False
        self.calibrations += 1
        return exps

    def _recalibrate(self, bits: int, probe) -> bool:
        """A scaled-mode launch reported that its calibration does not fit the data.  Other weights than the calibration
        saw: measure again.  Same weights: the probe was not representative - move the target (overflow: 2^4 more
        headroom; small: 2^3 less) and measure again, at most three times.  False: give up (bf16x3)."""
        if self._act_exps is not None and self._calib_key == self._param_key():
            if self._calib_moves >= 3 or bits == (L.FSN_STATUS_FP16_RANGE | L.FSN_STATUS_FP16_SMALL):
                return False
            self._calib_moves += 1
            move = -4 if bits & L.FSN_STATUS_FP16_RANGE else 3
            self.act_target_exp += move
            self._target_moved_by += move
            if not 2 <= self.act_target_exp <= 14:
                return False
        else:
            self._forget_target_moves()
        try:
            self.calibrate(probe)
        except RuntimeError:
            return False
        return True

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

    def _guard_weights(self, what: str, training: bool = False) -> None:
        # (scaled inference brings every layer's weights to the activations' scale: the check is about the unscaled
        # arithmetic of training and of `act_scaling = False`)
        if not training and self.infer_prec() == L.FSN_PREC_FP16X3U:
            return
        if self.range_check and self.weight_check and self.fp16_family(self.PRECISIONS[self.precision]) and \
                self._weights_below_envelope():
            self.fall_back(what + " (a layer's largest weight is below 2^-20)", L.FSN_STATUS_FP16_SMALL, scaled=False)

    def packed(self, probe=None) -> ops.PackedMLP:
        """Weights in the MFMA streaming layout of the inference kernels; re-packed when any parameter changed.  Scaled
        fp16x3 inference: the first call (and the first one after `load_state_dict` or a request of the training
        loop's periodic look) calibrates the per-layer scales on `probe` (see `calibrate`) first."""
        ws, bs = self._tensors()
        scaled = self.infer_prec() == L.FSN_PREC_FP16X3U
        if scaled and (self._act_exps is None or self._needs_calibration):
            self.calibrate(probe, keep_if_close=self._act_exps is not None)
        exps = tuple(self._act_exps) if scaled else None
        key = (self.precision, ws[0].device, exps) + self._param_key()
        if self._packed is None or key != self._packed_key:
            if self._pack_calls % max(1, int(self.range_check_every)) == 0:
                self._guard_weights("NeRF.packed")
                scaled = self.infer_prec() == L.FSN_PREC_FP16X3U
                exps = tuple(self._act_exps) if scaled else None
                key = (self.precision, key[1], exps) + key[3:]
            self._pack_calls += 1
            desc = ops.make_desc(self.n_layers, self.d_hidden, self.skip, self.pos_encoder.freqs,
                                 self.dir_encoder.freqs)
            prec = self.infer_prec()
            if self._packed is None or self._packed.prec != prec or self._packed.blob.device != ws[0].device:
                self._packed = ops.PackedMLP(desc, prec, ws[0].device)
            self._packed.pack(ws, bs, exps)
            self._packed_key = key
        return self._packed

    def packed_cull(self) -> ops.PackedMLP:
        """The weights packed for the mode `cull_precision` names (single-pass bf16: float32's range, no calibration and
        no range flags), re-packed when a parameter changed."""
        if self.cull_precision != "bf16":
            raise ValueError("NeRF.cull_precision: None or 'bf16'")
        ws, bs = self._tensors()
        key = (self.cull_precision, ws[0].device) + self._param_key()
        if self._packed_cull is None or key != self._packed_cull_key:
            desc = ops.make_desc(self.n_layers, self.d_hidden, self.skip, self.pos_encoder.freqs, self.dir_encoder.freqs)
            if self._packed_cull is None or self._packed_cull.blob.device != ws[0].device:
                self._packed_cull = ops.PackedMLP(desc, self.PRECISIONS[self.cull_precision], ws[0].device)
            self._packed_cull.pack(ws, bs)
            self._packed_cull_key = key
        return self._packed_cull

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

        def probe():
            n = ray_indices.numel()
            idx = torch.linspace(0, max(n - 1, 0), min(n, self.PROBE_MAX), device=dev).long()
            ri = ray_indices[idx].long()
            d = rays_d[ri]
            return rays_o[ri] + d * (t_starts[idx] + t_ends[idx])[:, None] / 2.0, d

        return self._guarded(dev, "NeRF.forward_rays", probe, lambda: ops.mlp_fwd_rays(self.packed(probe), *args))

    def _guarded(self, dev, what: str, probe, launch):
        """`launch()` under the fp16 range guard of an inference call.  range_check True: read the launch's word back
        (one host sync); a flagged call is re-run after `fall_back` (re-calibration, or bf16x3).  "deferred": no wait -
        the word of the PREVIOUS launch is looked at now (ops.range_poll), this launch posts its own for the next call /
        the end of the frame (render_frame) to look at.  False: never look."""
        f16 = lambda: self.fp16_family(self.PRECISIONS[self.precision])
        if not self.range_check or not f16():
            return launch()
        if self.range_check == "deferred":
            bits = ops.range_poll(dev)
            if bits:
                self.fall_back("an EARLIER call (deferred range check: its outputs are invalid)", bits, probe,
                               earlier_invalid=True)
            out = launch()
            if f16():
                ops.range_post(dev)
            return out
        out = launch()
        for _ in range(5):
            bits = ops.range_flags(dev)
            if not bits:
                break
            self.fall_back(what, bits, probe)
            out = launch()
            if not f16():
                break
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

        def probe():
            xs = x.reshape(-1, 3)
            if dirs is not None:
                return xs, dirs.reshape(-1, 3)
            return xs, torch.tensor([0.0, 0.0, 1.0], device=dev).expand_as(xs)  # (density only: any direction)

        return self._guarded(dev, "NeRF.forward", probe,
                             lambda: ops.mlp_fwd(self.packed(probe), x, dirs, self._mask(self.pos_mask, dev),
                                                 self._mask(self.dir_mask, dev)))
