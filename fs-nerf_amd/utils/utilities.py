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
