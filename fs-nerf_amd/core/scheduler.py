"""Host-side mirror of `src/core/scheduler.py`: learning-rate schedules.
Pure host arithmetic on optimizer param groups - nothing to accelerate."""
from torch.optim import Optimizer


class Scheduler:
    def __init__(self, optim: Optimizer, T: int, lro: float, **kwargs) -> None:
        if lro < 0:
            raise ValueError("lro must be a positive value.")
        self.optim, self.T, self.lro, self.t = optim, T, lro, 0

    def step(self) -> None:  # scheduler.py:34-41
        self.t += 1
        for group in self.optim.param_groups:
            group["lr"] = self.lr


class Constant(Scheduler):
    @property
    def lr(self) -> float:
        return self.lro


class ExponentialDecay(Scheduler):
    """lr = lro * r**(t/T) for t < T, then lro * r (scheduler.py:74-80)."""

    def __init__(self, optim: Optimizer, T: int, lro: float, **kwargs) -> None:
        super().__init__(optim, T, lro)
        self.r = kwargs["r"]
        self.lrf = self.lro * self.r

    @property
    def lr(self) -> float:
        return self.lro * (self.r ** (self.t / self.T)) if self.t < self.T else self.lrf
