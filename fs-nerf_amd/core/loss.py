"""Host-side mirror of `src/core/loss.py`: the occlusion regulariser as one
segmented reduction on the GPU instead of a Python loop with one kernel launch per ray."""
from torch import Tensor

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
