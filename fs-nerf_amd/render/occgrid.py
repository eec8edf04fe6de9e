"""Occupancy-grid estimator for the `estimator` slot of render_rays: the duck-typed interface
the reference uses from nerfacc's OccGridEstimator — construction `OccGridEstimator(roi_aabb=, resolution=, levels=)`
(src/run-nerf.py:96-98), `.sampling(rays_o, rays_d, sigma_fn=, render_step_size=, stratified=, near_plane=,
far_plane=)` (src/render/rendering.py:66-74), `.update_every_n_steps(step=, occ_eval_fn=, occ_thre=)`
(src/run-nerf.py:293-295), nn.Module modes / `.to()`.  nerfacc itself is not part of the reference; the sampling
rule is this build's definition of that contract (see csrc/occgrid.hip), on HIP kernels."""
import math
from typing import Callable, Optional, Sequence

import torch
from torch import Tensor, nn

from .. import ops


class OccGridEstimator(nn.Module):
    def __init__(self, roi_aabb, resolution: int = 128, levels: int = 1) -> None:
        super().__init__()
        aabb = [float(v) for v in (roi_aabb.tolist() if isinstance(roi_aabb, Tensor) else roi_aabb)]
        if len(aabb) != 6:
            raise ValueError("roi_aabb must hold 6 values {xmin, ymin, zmin, xmax, ymax, zmax}")
        self.aabb, self.resolution, self.levels = aabb, int(resolution), int(levels)
        n_cells = self.levels * self.resolution ** 3
        if n_cells % 64:
            raise ValueError("levels * resolution^3 must be a multiple of 64")
        self.register_buffer("occs", torch.zeros(n_cells, dtype=torch.float32))
        self.register_buffer("bits", torch.zeros(n_cells // 32, dtype=torch.int32))
        self.generator: Optional[torch.Generator] = None
        self._updates = 0      # update_every_n_steps calls that ran (part of the draws' seed)
        self._pending = None   # scratch of the duplicate-safe EMA / the selection's popcount prefix (device)
        self._prefix = None

    # -- helpers -------------------------------------------------------------------
    @property
    def binaries(self) -> Tensor:
        """[levels, res, res, res] bool view of the bit field (nerfacc's attribute of the same name)."""
        shifts = torch.arange(32, device=self.bits.device, dtype=torch.int32)
        b = ((self.bits[:, None] >> shifts[None, :]) & 1).bool()
        return b.reshape(self.levels, self.resolution, self.resolution, self.resolution)

    def set_binaries(self, binaries: Tensor) -> None:
        """Load an explicit occupancy ([levels, res, res, res] bool); occs become 1 / 0."""
        flat = binaries.reshape(-1).to(self.occs.device)
        self.occs.copy_(flat.float())
        thr = torch.full((1,), 0.5, device=self.occs.device)
        ops.occgrid_update(self.occs, self.bits, None, None, 1.0, thr)

    def level_aabb(self, lvl: int):
        c = [(self.aabb[a] + self.aabb[3 + a]) / 2.0 for a in range(3)]
        h = [(self.aabb[3 + a] - self.aabb[a]) / 2.0 * 2 ** lvl for a in range(3)]
        return [c[a] - h[a] for a in range(3)], [c[a] + h[a] for a in range(3)]

    def max_steps(self, render_step_size: float) -> int:
        """Lattice points a ray can have inside the outermost box (its diagonal / step, + 2)."""
        lo, hi = self.level_aabb(self.levels - 1)
        diag = math.sqrt(sum((hi[a] - lo[a]) ** 2 for a in range(3)))
        return int(min(16384, math.ceil(diag / render_step_size) + 2))

    # -- reference surface ---------------------------------------------------------
    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, u: Optional[Tensor] = None):
        """-> (ray_indices int64 [N], t_starts [N], t_ends [N]), packed and sorted by ray."""
        if alpha_fn is not None or t_min is not None or t_max is not None or cone_angle != 0.0:
            raise NotImplementedError("alpha_fn / t_min / t_max / cone_angle are not used by the reference")
        R = rays_o.shape[0]
        if u is None and stratified:
            u = torch.rand(R, device=rays_o.device, generator=self.generator)
        max_steps = self.max_steps(render_step_size)
        ri, t0, t1, _ = ops.occgrid_march(rays_o, rays_d, self.aabb, self.resolution, self.levels, self.bits, near_plane,
                                          far_plane, render_step_size, u, max_steps)
        if sigma_fn is not None and (early_stop_eps > 0.0 or alpha_thre > 0.0) and ri.numel() > 0:
            sig = sigma_fn(t0, t1, ri)
            keep = ops.packed_visibility(sig.reshape(-1), t0, t1, ri, R, early_stop_eps, alpha_thre)
            ri, t0, t1 = ri[keep], t0[keep], t1[keep]
        return ri, t0, t1

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        """Every n-th training step (run-nerf.py:288-295): re-evaluate cells - all of them during warm-up, else res^3/4
        drawn uniformly + res^3/4 drawn uniformly from the occupied ones, with replacement - at a random point inside
        each, occs = max(occs*decay, occ), binaries = occs > min(mean, thre).
        Round 4: selection, jitter and the duplicate-safe EMA are kernels reading the bit field directly
        (fsn_occgrid_select / fsn_occgrid_update_multi); no host sync, no bool expansion of the grid.  Randomness is a
        counter-based hash of (seed, draw): seed = the estimator generator's (or torch's) initial seed and the number of
        updates made so far - `oracle.occgrid_select` restates it."""
        if not self.training or step % n != 0:
            return
        res, res3 = self.resolution, self.resolution ** 3
        if self._pending is None or self._pending.device != self.occs.device:
            self._pending = torch.zeros(self.occs.numel(), dtype=torch.int32, device=self.occs.device)
            self._prefix = torch.empty(res3 // 32 + 1, dtype=torch.int32, device=self.occs.device)
        for lvl in range(self.levels):
            seed = self.update_seed(lvl)
            warm = step < warmup_steps
            cells, x = ops.occgrid_select(self.bits, self.aabb, res, self.levels, lvl, warm, res3 // 4, res3 // 4, seed,
                                          self._prefix)
            occ = occ_eval_fn(x).reshape(-1).float()
            ops.occgrid_update_multi(self.occs, self._pending, cells, occ, ema_decay)
        self._updates += 1
        thr = torch.clamp(self.occs.mean(), max=occ_thre).reshape(1)
        ops.occgrid_update(self.occs, self.bits, None, None, 1.0, thr)

    def update_seed(self, lvl: int = 0) -> int:
        """64-bit seed of the NEXT update's draws at level `lvl` (see update_every_n_steps)."""
        base = self.generator.initial_seed() if self.generator is not None else torch.initial_seed()
        return (base * 0x9E3779B97F4A7C15 + self._updates * 0x100000001B3 + lvl * 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
