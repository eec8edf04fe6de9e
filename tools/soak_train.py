"""Soak of the training path (run ON the GPU box): three 8x256 students (plain, reference weight-norm regulariser l1 / l2)
fitted for N iterations through render_rays(train=True) -> mse (+ reg) -> backward -> FusedAdam; records range warnings,
steps skipped by the delayed gradient scaling, the per-stage factors it ended on and (round 4) what the scaled fp16x3
inference path of the sampler's density pass did on the way (calibrations, range events, final exponents, updates the
optimizer applied) -> gpurun_out/r04_soak.json.
usage: python tools/soak_train.py [iterations]"""
import sys, os, warnings, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fs_nerf_amd
from oracle import fsnerf_oracle as O
from fs_nerf_amd.core.models import NeRF
from fs_nerf_amd.core.optim import FusedAdam
from fs_nerf_amd.core.loss import WeightNormRegularizer
from fs_nerf_amd.core.scheduler import ExponentialDecay
from fs_nerf_amd.render import rendering as Rm
from fs_nerf_amd import ops
dev = torch.device("cuda:0")
L, D = 8, 256
def make(seed):
    sd = O.init_nerf_state_dict(L, D, [4], 10, 4, seed=seed)
    sd["sigma.weight"] *= 64.0
    sd["sigma.bias"] += 3.0
    m = NeRF(3, 3, L, D, (4,), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    m.load_state_dict(sd)
    return m.to(dev)
teacher = make(21).eval()
ro, rd = [], []
for phi in (0.0, 90.0, 180.0, 270.0):
    o, d = O.get_rays(O.pose_from_spherical(4.0311289, 50.0, phi), (64, 64, 88.0))
    ro.append(o.reshape(-1, 3)); rd.append(d.reshape(-1, 3))
ro, rd = torch.cat(ro).contiguous().to(dev), torch.cat(rd).contiguous().to(dev)
with torch.no_grad():
    gt = Rm.render_rays(ro, rd, Rm.StratifiedEstimator(2.0, 6.0, 64, 64), teacher, white_bkgd=True, device=dev)[0][0]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
out = {}
for name, seed, alpha, norm in (("plain", 22, None, None), ("wnorm_l1", 23, 3e-8, "l1"), ("wnorm_l2", 24, 2e-5, "l2")):
    student = make(seed).train()
    est = Rm.StratifiedEstimator(2.0, 6.0, 64, 64).train()
    est.generator = torch.Generator(device=dev).manual_seed(0)
    opt = FusedAdam(student.parameters(), lr=5e-4)
    sched = ExponentialDecay(opt, iters, 5e-4, r=0.1)
    reg = WeightNormRegularizer(student.named_parameters(), reg=norm, reg_ratio=1.0, Td=iters) if alpha else None
    gen = torch.Generator(device=dev).manual_seed(1)
    warned, losses = [], []
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always", RuntimeWarning)
        for it in range(iters):
            idx = torch.randint(0, ro.shape[0], (1024,), device=dev, generator=gen)
            opt.zero_grad()
            rgb = Rm.render_rays(ro[idx], rd[idx], est, student, train=True, white_bkgd=True, device=dev)[0][0]
            loss = torch.nn.functional.mse_loss(rgb, gt[idx])
            if reg is not None and reg.active(it):
                loss = loss + alpha * reg()
            loss.backward()
            opt.step(); sched.step()
            if it % 100 == 0 or it == iters - 1:
                losses.append(float(loss.detach()))
        warned = [str(x.message)[:160] for x in w if issubclass(x.category, RuntimeWarning)]
    sc = [int(round(math.log2(v))) for v in student._bwd_stage[0].tolist()] if student._bwd_stage else None
    wmax = {k: float(v.detach().abs().max()) for k, v in student.named_parameters() if k.endswith("weight")}
    out[name] = {"precision_at_end": student.precision, "grad_overflow_looks": student.grad_overflow_looks,
                 "range_warnings": warned, "loss_first": losses[0], "loss_last": losses[-1], "stage_scale_log2": sc,
                 "min_layer_max_weight": min(wmax.values()), "calibrations": student.calibrations,
                 "range_events": student.range_events, "act_exps": student._act_exps, "act_target_exp": student.act_target_exp,
                 "optimizer_steps_applied": opt.steps, "iterations": iters}
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_soak.json"), "w"), indent=1)
