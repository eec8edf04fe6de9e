"""Ray sharding across the GPUs of one node (one process per GPU, torch.distributed).

Rays are independent, so the rendering path partitions with NO data-path collective
(SURVEY.md 8e): a rank renders its own frames (bench.py, weak scaling) or its own row block of a
frame (`shard_rows`, then `fsn_get_rays(row0, nrows)` + the fused launch).  The only collectives
are outside the timed path: `gather_rows` to assemble one image, `max_over_ranks` for timing.
The backend is whatever the process group was created with ("nccl" = RCCL on the GPU box,
"gloo" in the CPU tests)."""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(H: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous row block [row0, row0+nrows) of rank `rank`; the first H % world ranks get one more."""
    base, rem = divmod(H, world)
    nrows = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, nrows


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Round-robin frame assignment (rank, rank+world, ...)."""
    return range(rank, n_frames, world)


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def max_bits_over_ranks(bits: int, device=None) -> int:
    """MAX over the ranks of a small non-negative integer (the fp16 range words of a training step: every rank must
    take the same fall-back decision).  One tiny collective; identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(bits)
    on_host = dist.get_backend() == "gloo" or device is None
    t = torch.tensor([int(bits)], dtype=torch.int32, device="cpu" if on_host else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def gather_rows(local: torch.Tensor, H: int) -> torch.Tensor:
    """All-gather row blocks [nrows_r, W, C] produced under `shard_rows` into the [H, W, C] image."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    max_rows = (H + world - 1) // world
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][: shard_rows(H, r, world)[1]] for r in range(world)], dim=0)


class FlatGrads:
    """ONE persistent flat fp32 gradient bucket for a FIXED parameter list (identical on every rank: same model,
    same order), with every `p.grad` a view into it.  Autograd accumulates into the views in place, the data-parallel
    synchronisation is a single all-reduce of `flat` with no concatenation or copy-back, and a flat optimizer
    (core/optim.py:FusedAdam) reads the same memory.  A parameter whose gradient is missing (`None`) contributes
    zeros, so every rank always enters the collective with the same bucket size."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGrads: no trainable parameters")
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatGrads: parameters must be float32 on one device")
            self.offsets.append(n)
            n += p.numel()
        self.numel = n
        # one extra slot behind the gradients: the "an fp16-mode launch of this step overflowed" flag travels in the
        # SAME all-reduce (sum: > 0 on every rank when any rank raised it), so all ranks skip the update together
        self._buf = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.flat = self._buf[:n]
        self.flag_slot = self._buf[n:]
        # THIS bucket's "a training launch of the current step overflowed" word (one word per device let model
        # A's overflow skip optimizer B's update, and the first optimizer to step consumed the flag of the second): the
        # backward of a network whose parameters live in this bucket ORs its per-call range word into it (device op),
        # the optimizer that owns the bucket hands it to its Adam launch and clears it.
        self.step_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self._views = [self.view(i) for i in range(len(self.params))]  # the bucket's views, made once
        self.bind()

    def view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)

    def bind(self) -> None:
        """(Re)attach: gradients that autograd or the caller replaced are copied into the bucket, missing ones
        become zeros, and `p.grad` is pointed at the bucket again."""
        for p, v in zip(self.params, self._views):
            g = p.grad
            if g is v:  # still attached (the usual case: three calls per training step, 24 parameters each)
                continue
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                v.copy_(g)
            p.grad = v
            p._fsn_grad_sink = True  # core/models.py: the backward kernels may accumulate into this buffer directly
            p._fsn_step_flag = self.step_flag

    def zero(self) -> None:
        """`optimizer.zero_grad(set_to_none=False)` for the whole model in one fill."""
        self.flat.zero_()
        self.bind()

    def allreduce(self, average: bool = True, flag: Optional[float] = None) -> None:
        """`flag`: this rank's "skip this step" value for the bucket's extra slot; default = the bucket's own step flag
        (raised by an fp16-mode training launch on its parameters that overflowed)."""
        self.bind()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        if flag is not None:
            self.flag_slot.fill_(float(flag))
        elif self._buf.is_cuda:  # this rank's step flag (word of the training launches on these parameters) into the slot
            self.flag_slot.copy_((self.step_flag & 5).to(torch.float32))  # FSN_STATUS_FP16_RANGE | _GRAD_RANGE
        flat = self._buf
        if flat.is_cuda and dist.get_backend() == "gloo":  # CPU rehearsal backend: stage through host memory
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            self.flat.div_(dist.get_world_size())


_buckets = {}


def allreduce_grads(params, average: bool = True) -> FlatGrads:
    """Data-parallel gradient synchronisation (north_star: "an RCCL all-reduce of the gradient over xGMI and
    nothing else"): ONE all-reduce(sum) of one flat fp32 bucket holding every gradient (595,844 floats = 2.38 MB
    for an 8x256 NeRF; the message is latency-bound on xGMI, so no per-layer hooks or bucketing), then x 1/world.
    The bucket (`FlatGrads`) is built once per parameter list and reused: the list is FIXED (every trainable
    parameter, zeros where a gradient is missing), so all ranks always reduce the same number of elements."""
    params = [p for p in params if p.requires_grad]
    key = tuple(id(p) for p in params)
    b = _buckets.get(key)
    if b is None or any(a is not c for a, c in zip(b.params, params)):
        b = _buckets[key] = FlatGrads(params)
    b.allreduce(average)
    return b
