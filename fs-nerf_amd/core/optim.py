"""Optimizer side of the training step on the device (reference: `torch.optim.Adam(params, lr)` at
src/run-nerf.py:217, `optimizer.step()` / `zero_grad()` at :283-285).

`FlatParams` re-homes every parameter of a model into ONE flat float32 arena (each `p.data` becomes a view), with the
gradients in a second arena of the same layout (`shard.FlatGrads`: the buffer the RCCL all-reduce runs on).
`FusedAdam` is a `torch.optim.Optimizer` whose `step()` is a single HIP launch over those arenas
(`fsn_adam_step`, torch.optim.Adam's arithmetic operation for operation) - no per-tensor kernels, no `torch.cat`.

Differences from `torch.optim.Adam`, all deliberate:
  * a parameter whose gradient is None takes part with a ZERO gradient (its moments decay, torch skips it): the
    arenas are one flat buffer and one launch;
  * `zero_grad(set_to_none=True)` fills the gradient arena with zeros and keeps `p.grad` bound to it;
  * a step in which an fp16-mode training launch on ITS parameters overflowed is skipped on the device (the gradient
    bucket's own `step_flag` word, or the flag slot of an all-reduced bucket): parameters and moments untouched, and - round 4 - the step COUNTER too: it lives on
    the device (`fsn_adam_step_dev`), is advanced by the launch only when the update runs, and the bias corrections are
    formed from it there, so a skipped step is a step that did not happen, exactly as with torch's GradScaler
    (overflowing ACTIVATIONS: the model leaves fp16 mode at the host's next look; an overflowing GRADIENT under the
    backward's delayed per-stage scaling: the stage's factor drops and training continues in the same mode);
  * the backward kernels add into the gradient arena's views directly (`fsn_nerf_train_bwd(accumulate=1)`): autograd
    sees no gradient tensors for those parameters (parameter hooks do not fire; use `loss.backward()`);
  * `state_dict()` / `load_state_dict()` carry the flat moments and the step count (checkpoint / resume)."""
from typing import Iterable, List, Optional

import torch
from torch import Tensor

from .. import _lib as L
from .. import ops
from ..shard import FlatGrads


class FlatParams:
    """All trainable parameters as views into one contiguous float32 buffer (state_dict keys and shapes unchanged)."""

    def __init__(self, params: Iterable[Tensor]):
        self.params: List[Tensor] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatParams: no trainable parameters")
        dev = self.params[0].device
        if not all(p.is_cuda and p.device == dev and p.dtype == torch.float32 for p in self.params):
            raise RuntimeError("FlatParams: float32 parameters on one GPU expected (the HIP path has no CPU fallback)")
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                v = self.flat[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
        self.grads = FlatGrads(self.params)

    def offset_of(self, p: Tensor) -> int:
        for q, off in zip(self.params, self.offsets):
            if q is p:
                return off
        raise KeyError("parameter is not part of this arena")

    def attached(self) -> bool:
        """False once something (e.g. `model.to(...)`, `load_state_dict(assign=True)`) replaced a parameter's storage."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * off for p, off in zip(self.params, self.offsets))


class FusedAdam(torch.optim.Optimizer):
    """`torch.optim.Adam(params, lr=..., betas=..., eps=..., weight_decay=...)` (no amsgrad) as one launch per step.
    `grad_div`: see `step`.  The parameters are moved into a `FlatParams` arena on construction."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdam: one parameter group (the reference uses one, run-nerf.py:216-217)")
        self.arena = FlatParams(self.param_groups[0]["params"])
        self.exp_avg = torch.zeros_like(self.arena.flat)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=self.arena.flat.device)  # updates APPLIED (device)
        self._tick = torch.zeros(4, dtype=torch.float32, device=self.arena.flat.device)

    @property
    def steps(self) -> int:
        """Updates applied so far (skipped steps do not count).  Reads 4 bytes back: checkpointing / tests only."""
        return int(self.step_count.item())

    @steps.setter
    def steps(self, v: int) -> None:
        self.step_count.fill_(int(v))

    @property
    def grads(self) -> FlatGrads:
        return self.arena.grads

    def zero_grad(self, set_to_none: bool = False) -> None:  # one fill of the gradient arena
        self.arena.grads.zero()

    @torch.no_grad()
    def step(self, closure=None, grad_div: float = 1.0):
        """`grad_div` = world size when the gradient arena holds an all-reduced SUM (`FlatGrads.allreduce(average=
        False)`): the division of the data-parallel mean then costs no pass of its own."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self.arena.attached():
            raise RuntimeError("FusedAdam: a parameter no longer lives in the optimizer's arena (was the model moved or "
                               "re-assigned after the optimizer was built?)")
        g = self.param_groups[0]
        self.arena.grads.bind()
        a = self.arena
        flag = a.grads.step_flag  # the word of THIS optimizer's parameters (shard.FlatGrads), not a per-device one
        with torch.cuda.device(a.flat.device):
            L.check(L.lib().fsn_adam_step_dev(ops._p(a.flat), ops._p(a.grads.flat), ops._p(self.exp_avg),
                                              ops._p(self.exp_avg_sq), a.numel, ops._p(self.step_count), ops._p(self._tick),
                                              float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                              float(g["weight_decay"]), float(grad_div), ops._p(flag),
                                              ops._p(a.grads.flag_slot), ops._stream()), "fsn_adam_step_dev")
        flag.zero_()  # the step's flag is consumed (stream order: after the launch that read it)
        a.grads.flag_slot.zero_()
        self._bump_versions()
        return loss

    # checkpoint / resume: the moments and the step count live outside Optimizer.state (flat arenas)
    def state_dict(self):
        sd = super().state_dict()
        sd["fused_adam"] = {"exp_avg": self.exp_avg.detach().clone(), "exp_avg_sq": self.exp_avg_sq.detach().clone(),
                            "steps": self.steps}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        extra = state_dict.pop("fused_adam", None)
        super().load_state_dict(state_dict)
        if extra is not None:
            if extra["exp_avg"].numel() != self.exp_avg.numel():
                raise ValueError("FusedAdam.load_state_dict: moment arenas of a different parameter list")
            self.exp_avg.copy_(extra["exp_avg"])
            self.exp_avg_sq.copy_(extra["exp_avg_sq"])
            self.steps = int(extra["steps"])

    def _bump_versions(self) -> None:
        # `NeRF.packed()` re-packs when a parameter's version counter changed; the HIP launch wrote through raw
        # pointers, so the write is recorded here (no kernel: the counter only).
        for p in self.arena.params:
            torch.autograd.graph.increment_version(p)
