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
    """The datasets' ray precompute (src/nerfdata/datasets/blender.py:174-191, llff.py:59-90) as ONE
    launch (ops.build_rays / fsn_build_rays): rays of every pose written straight into one [n*H*W, 3] pair (no per-pose
    cat / stack), optional NDC mapping of all of them in the same pass, and the region of interest the reference derives
    for its estimator reduced on the way: NDC -> aabb = [min, max] over {o, o + d} / 2^3 (llff.py:77-84), else
    [-1.5]*3 + [1.5]*3 (blender.py:140).  (Round 3: a Python loop of per-pose launches, a to_ndc launch and five torch
    reductions.)"""
    H, W, focal = hwf
    device = _need_gpu(device, "build_rays")
    P = torch.stack([torch.as_tensor(p, dtype=torch.float32)[:3, :4] for p in poses]) if len(poses) else torch.zeros(0, 3, 4)
    rays_o, rays_d, aabb = ops.build_rays(P, int(H), int(W), float(focal), device, ndc=ndc, near=1.0, want_aabb=ndc)
    if not ndc:
        aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], device=device)
    return rays_o, rays_d, aabb
