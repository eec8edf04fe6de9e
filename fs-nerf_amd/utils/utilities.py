"""Host-side mirror of the reference's ray helpers (src/utils/utilities.py:36-134) on HIP."""
from typing import List, Tuple

import torch
from torch import Tensor

from .. import ops


def _need_gpu(device, what: str) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"{what}: the MI355X path has no CPU implementation; pass a GPU device")
    return device


def get_rays(pose: Tensor, hwf: Tuple[int, int, float],
             device: torch.device = torch.device("cuda")) -> Tuple[Tensor, Tensor]:
    """pose [4,4] or [3,4] -> (origins_w [H,W,3], dirs_w [H,W,3]); utilities.py:36-82.
    (The reference returns the origins as a stride-0 view; here they are materialised.)"""
    H, W, focal = hwf
    device = _need_gpu(device, "get_rays")
    o, d = ops.get_rays(pose, int(H), int(W), float(focal), device)
    return o.reshape(H, W, 3), d.reshape(H, W, 3)


def to_ndc(rays_o: Tensor, rays_d: Tensor, hwf: Tuple[int, int, float], near: float) -> Tuple[Tensor, Tensor]:
    """World rays -> normalised device coordinates; utilities.py:84-120."""
    H, W, focal = hwf
    shape = rays_o.shape
    no, nd = ops.to_ndc(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), int(H), int(W), float(focal), float(near))
    return no.reshape(shape), nd.reshape(shape)


def get_chunks(inputs: Tensor, chunksize: int) -> List[Tensor]:
    """Row slices of at most `chunksize`; utilities.py:122-134."""
    return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def build_rays(poses, hwf: Tuple[int, int, float], device: torch.device = torch.device("cuda"),
               ndc: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """The datasets' ray precompute (src/nerfdata/datasets/blender.py:174-191, llff.py:59-90) on the
    device: rays of every pose written straight into one [n*H*W, 3] pair (no per-pose cat / stack), optional NDC
    mapping over all of them, and the region of interest the reference derives for its estimator:
    NDC -> aabb = [min, max] over {o, o + d} / 2^3 (llff.py:77-84), else [-1.5]*3 + [1.5]*3 (blender.py:140)."""
    H, W, focal = hwf
    device = _need_gpu(device, "build_rays")
    n, per = len(poses), int(H) * int(W)
    rays_o = torch.empty(n * per, 3, device=device, dtype=torch.float32)
    rays_d = torch.empty_like(rays_o)
    for i, pose in enumerate(poses):
        ops.get_rays(pose, int(H), int(W), float(focal), device, out=(rays_o[i * per:(i + 1) * per], rays_d[i * per:(i + 1) * per]))
    if ndc:
        rays_o, rays_d = to_ndc(rays_o, rays_d, hwf, 1.0)
        ends = rays_o + rays_d
        lo = torch.minimum(rays_o.amin(dim=0), ends.amin(dim=0))
        hi = torch.maximum(rays_o.amax(dim=0), ends.amax(dim=0))
        aabb = torch.cat([lo, hi]) / 2 ** (4 - 1)
    else:
        aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], device=device)
    return rays_o, rays_d, aabb
