"""Worker of tests/test_dp_two_ranks_gpu.py: ONE data-parallel rank of the real HIP training step (VERDICT r3 next
#6 iii).  Started as a fresh process per rank (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment), both
ranks on the one GPU of the box, collectives over gloo - what DESIGN.md's rehearsal did by hand.  Three steps of
render_rays(train=True) -> mse -> backward -> ONE all-reduce of the flat bucket -> FusedAdam; in step 1 (the second)
rank 1 raises its device step flag as an overflowing fp16 launch would.  Writes what the test compares to a JSON file."""
import hashlib
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path = sys.argv[1]
    import fs_nerf_amd  # noqa: F401
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.models import NeRF
    from fs_nerf_amd.core.optim import FusedAdam
    from fs_nerf_amd.render import rendering as Rm
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"all_reduce": 0}
    real = dist.all_reduce

    def counted(*a, **k):
        calls["all_reduce"] += 1
        return real(*a, **k)

    dist.all_reduce = counted
    torch.manual_seed(42)  # identical replicas
    m = NeRF(3, 3, 4, 128, (), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    with torch.no_grad():
        m.sigma.weight.mul_(64.0)
        m.sigma.bias.add_(3.0)
    m = m.to(dev).train()
    est = Rm.StratifiedEstimator(2.0, 6.0, 32, 32).train()
    est.generator = torch.Generator(device=dev).manual_seed(1000 + rank)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    gen = torch.Generator(device=dev).manual_seed(2000 + rank)  # every rank its own rays
    hashes, per_step_calls, steps_applied = [digest(opt.arena.flat)], [], []
    for step in range(3):
        o = torch.tensor([0.0, 0.0, 4.0], device=dev).expand(256, 3).contiguous()
        d = torch.nn.functional.normalize(torch.randn(256, 3, device=dev, generator=gen) * 0.2 + torch.tensor([0.0, 0.0, -1.0], device=dev), dim=-1)
        gt = torch.rand(256, 3, device=dev, generator=gen)
        c0 = calls["all_reduce"]
        opt.zero_grad()
        (rgb, _, _, _), _, _ = Rm.render_rays(o, d, est, m, train=True, white_bkgd=True, device=dev)
        torch.nn.functional.mse_loss(rgb, gt).backward()
        if step == 1 and rank == 1:
            opt.grads.step_flag.fill_(1)  # what an overflowing fp16 training launch of this rank leaves behind
        opt.grads.allreduce(average=False)
        opt.step(grad_div=float(world))
        per_step_calls.append(calls["all_reduce"] - c0)
        hashes.append(digest(opt.arena.flat))
        steps_applied.append(opt.steps)
    json.dump({"rank": rank, "hashes": hashes, "all_reduce_per_step": per_step_calls, "steps_applied": steps_applied,
               "precision": m.precision, "grad_hash": digest(opt.grads.flat)}, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
