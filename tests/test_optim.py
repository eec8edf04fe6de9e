"""Optimizer side of the training step: fused Adam over the flat parameter arena against
torch.optim.Adam, the weight-norm regulariser (run-nerf.py:266-279) against the reference's formula, the arena
plumbing (parameters / gradients as views, NeRF re-packing after a step)."""
import ctypes as C

import pytest
import torch

import fs_nerf_amd  # noqa: F401
from fs_nerf_amd import _lib as L


def test_optimizer_entry_points_validate_without_gpu():
    lib = L.lib()
    assert lib.fsn_adam_step(None, None, None, None, 0, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, None, None, None) == 0  # n = 0
    assert lib.fsn_adam_step(None, None, None, None, 8, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, None, None, None) != 0  # null pointers
    assert lib.fsn_adam_step(None, None, None, None, 8, 0, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, None, None, None) != 0  # step 0
    lens = (C.c_int64 * 2)(10, 20000)
    assert lib.fsn_weight_norm_workspace_floats(2, lens) == 1 + 3 + 40
    assert lib.fsn_weight_norm_workspace_floats(99, lens) < 0
    offs = (C.c_int64 * 2)(0, -5)
    assert lib.fsn_weight_norm_fwd(None, 2, offs, lens, 0, None, None, None) != 0


def _model(dev, seed=0):
    from fs_nerf_amd.core.models import NeRF
    torch.manual_seed(seed)
    m = NeRF(3, 3, 4, 128, (), pos_fn={"n_freqs": 10, "log_space": True}, dir_fn={"n_freqs": 4, "log_space": True})
    return m.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adam_matches_torch_adam(wd):
    from fs_nerf_amd.core.optim import FusedAdam
    dev = torch.device("cuda:0")
    a, b = _model(dev), _model(dev)
    keys = list(a.state_dict().keys())
    oa = FusedAdam(a.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    ob = torch.optim.Adam(b.parameters(), lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, foreach=False,
                          fused=False)
    assert list(a.state_dict().keys()) == keys and oa.arena.attached()
    assert all(p.data_ptr() == oa.arena.flat.data_ptr() + 4 * o for p, o in zip(oa.arena.params, oa.arena.offsets))
    gen = torch.Generator(device=dev).manual_seed(1)
    for step in range(6):
        oa.zero_grad()
        ob.zero_grad(set_to_none=True)
        for pa, pb in zip(a.parameters(), b.parameters()):
            g = torch.randn(pa.shape, device=dev, generator=gen) * (10.0 ** (step - 3))
            pa.grad.copy_(g)     # gradients are views into the flat bucket: written in place
            pb.grad = g.clone()
        assert all(p.grad.data_ptr() == oa.grads.view(i).data_ptr() for i, p in enumerate(oa.arena.params))
        oa.step()
        ob.step()
        for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=1e-9), (step, n, float((pa - pb).abs().max()))
    # grad_div: an all-reduced SUM over `world` ranks steps like the mean
    c = _model(dev)
    oc = FusedAdam(c.parameters(), lr=5e-4)
    d = _model(dev)
    od = FusedAdam(d.parameters(), lr=5e-4)
    oc.zero_grad(); od.zero_grad()
    for pc, pd in zip(c.parameters(), d.parameters()):
        g = torch.randn(pc.shape, device=dev, generator=gen)
        pc.grad.copy_(4.0 * g)
        pd.grad.copy_(g)
    oc.step(grad_div=4.0)
    od.step()
    assert all(torch.equal(pc, pd) for pc, pd in zip(c.parameters(), d.parameters()))


@pytest.mark.gpu
def test_fused_adam_step_is_seen_by_the_packed_weights():
    """NeRF.forward runs on the packed blob: it must be re-packed after the optimizer wrote the arena."""
    from fs_nerf_amd.core.optim import FusedAdam
    dev = torch.device("cuda:0")
    m = _model(dev).train()
    opt = FusedAdam(m.parameters(), lr=1e-2)
    x = torch.rand(300, 3, device=dev) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(300, 3, device=dev), dim=-1)
    with torch.no_grad():
        y0 = m.eval()(x, d).clone()
    m.train()
    opt.zero_grad()
    out = m(x, d)
    out.square().mean().backward()
    assert float(opt.grads.flat.abs().max()) > 0.0, "autograd accumulated into the flat bucket"
    opt.step()
    with torch.no_grad():
        y1 = m.eval()(x, d)
    assert float((y1 - y0).abs().max()) > 1e-4, "the forward must see the stepped parameters"
    ref = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert torch.equal(ref, opt.arena.flat)


@pytest.mark.gpu
@pytest.mark.parametrize("reg", ["l1", "l2"])
@pytest.mark.parametrize("arena", [True, False])
def test_weight_norm_regulariser_matches_the_reference_formula(reg, arena):
    from fs_nerf_amd.core.loss import WeightNormRegularizer
    from fs_nerf_amd.core.optim import FlatParams
    dev = torch.device("cuda:0")
    m = _model(dev, seed=3)
    if arena:
        FlatParams(m.parameters())
    wn = WeightNormRegularizer(m.named_parameters(), reg=reg, reg_ratio=0.5, Td=100)
    names = [n for n, p in m.named_parameters() if "weight" in n and p.shape[0] > 3]
    assert len(wn.params) == len(names) == 6 and "rgb.weight" not in names and "sigma.weight" not in names
    assert wn.active(49) and not wn.active(50)
    val = wn()
    (0.37 * val).backward()
    # the reference's loop (run-nerf.py:272-277), on float64 copies
    ps = {n: p.detach().double().clone().requires_grad_(True) for n, p in m.named_parameters()}
    want = torch.zeros((), dtype=torch.float64, device=dev)
    for n, p in ps.items():
        if "weight" in n and p.shape[0] > 3:
            want = want + (torch.abs(p).sum() if reg == "l1" else torch.square(p).sum().sqrt())
    (0.37 * want).backward()
    assert abs(float(val) - float(want)) <= 2e-6 * float(want)
    for n, p in m.named_parameters():
        if n in names:
            assert torch.allclose(p.grad.double(), ps[n].grad, rtol=1e-5, atol=1e-7), n
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
    # bit-reproducible (fixed summation order, no atomics)
    assert float(wn()) == float(val)


@pytest.mark.gpu
def test_fused_adam_skips_a_flagged_step_and_checkpoints_its_state():
    """(1) The bucket's device-side step flag (FlatGrads.step_flag: an fp16-mode training launch of this step overflowed) or the flag
    slot of the all-reduced bucket makes the Adam launch a no-op - parameters and both moments untouched - and is
    consumed by the step.  (2) state_dict / load_state_dict carry the flat moments and the step count: a resumed
    optimizer continues bit for bit."""
    import copy
    from fs_nerf_amd import ops
    from fs_nerf_amd.core.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    opt = FusedAdam(net.parameters(), lr=1e-2)
    x = torch.randn(16, 7, device=dev)

    def one_step(o, n):
        o.zero_grad()
        n(x).square().sum().backward()
        o.step()

    one_step(opt, net)
    before = [p.detach().clone() for p in net.parameters()]
    m0, v0 = opt.exp_avg.clone(), opt.exp_avg_sq.clone()
    for which in ("word", "slot"):
        opt.zero_grad()
        net(x).square().sum().backward()
        if which == "word":
            opt.grads.step_flag.fill_(1)
        else:
            opt.grads.flag_slot.fill_(2.0)  # two ranks flagged
        opt.step()
        for p, b in zip(net.parameters(), before):
            assert torch.equal(p.detach(), b), f"{which}: a flagged step must not move the parameters"
        assert torch.equal(opt.exp_avg, m0) and torch.equal(opt.exp_avg_sq, v0), f"{which}: nor the moments"
        assert int(opt.grads.step_flag.item()) == 0 and float(opt.grads.flag_slot.item()) == 0.0, "flag consumed"
        assert opt.steps == 1, f"{which}: a skipped step does not advance the (device-side) step counter"
    # ... so the bias corrections of the next clean step are those of step 2: the same update torch.optim.Adam makes when
    # a GradScaler withheld the two flagged steps (round 3 counted them on the host)
    ref_net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    ref_net.load_state_dict({k: v.clone() for k, v in zip(ref_net.state_dict(), before)})
    ref_opt = torch.optim.Adam(ref_net.parameters(), lr=1e-2)
    ref_opt.zero_grad()
    ref_net(x).square().sum().backward()
    for p, m_, v_ in zip(ref_net.parameters(), torch.split(m0, [p.numel() for p in ref_net.parameters()]),
                         torch.split(v0, [p.numel() for p in ref_net.parameters()])):
        ref_opt.state[p] = {"step": torch.tensor(1.0), "exp_avg": m_.view_as(p).clone(), "exp_avg_sq": v_.view_as(p).clone()}
    ref_opt.step()
    one_step(opt, net)
    assert opt.steps == 2
    assert not torch.equal(next(net.parameters()).detach(), before[0]), "the next clean step updates again"
    for p, q in zip(net.parameters(), ref_net.parameters()):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-7 * max(1.0, float(q.detach().abs().max())), "step 2's bias corrections"
    # checkpoint / resume
    sd_opt, sd_net = copy.deepcopy(opt.state_dict()), copy.deepcopy(net.state_dict())
    one_step(opt, net)
    want = [p.detach().clone() for p in net.parameters()]
    net2 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3)).to(dev)
    net2.load_state_dict(sd_net)
    opt2 = FusedAdam(net2.parameters(), lr=1e-2)
    opt2.load_state_dict(sd_opt)
    assert opt2.steps == sd_opt["fused_adam"]["steps"]
    one_step(opt2, net2)
    for p, w in zip(net2.parameters(), want):
        assert torch.equal(p.detach(), w), "resumed optimizer continues bit for bit"


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
def test_backward_accumulates_into_the_flat_bucket_like_autograd(precision):
    """Parameters whose .grad lives in a FlatGrads bucket: the backward kernels ADD into the buffers (no returned
    temporaries, no add launch per parameter).  Same gradients as the returning path, and a second backward without
    zero_grad accumulates exactly like AccumulateGrad does (g1 + g2 in fp32)."""
    from fs_nerf_amd.core.optim import FusedAdam
    dev = torch.device("cuda:0")
    a, b = _model(dev).train(), _model(dev).train()
    a.precision = b.precision = precision
    opt = FusedAdam(a.parameters(), lr=1e-3)  # a: gradients in the bucket; b: plain autograd
    assert all(getattr(p, "_fsn_grad_sink", False) for p in a.parameters())
    gen = torch.Generator(device=dev).manual_seed(5)
    x1, x2 = (torch.rand(777, 3, device=dev, generator=gen) * 2 - 1 for _ in range(2))
    d1, d2 = (torch.nn.functional.normalize(torch.randn(777, 3, device=dev, generator=gen), dim=-1) for _ in range(2))
    opt.zero_grad()
    a(x1, d1).square().mean().backward()
    b(x1, d1).square().mean().backward()
    ptrs = [p.grad.data_ptr() for p in a.parameters()]
    assert ptrs == [opt.grads.view(i).data_ptr() for i in range(len(ptrs))], "the gradients stayed views of the bucket"
    g1 = [p.grad.clone() for p in b.parameters()]
    for (n, pa), pb in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(pa.grad, pb.grad), (n, float((pa.grad - pb.grad).abs().max()))
    # second backward, no zero_grad: both paths accumulate
    (3.0 * a(x2, d2).square().mean()).backward()
    for p in b.parameters():
        p.grad = None
    (3.0 * b(x2, d2).square().mean()).backward()
    for (n, pa), pb, g in zip(a.named_parameters(), b.parameters(), g1):
        assert torch.equal(pa.grad, g + pb.grad), (n, float((pa.grad - (g + pb.grad)).abs().max()))
    # a frozen parameter (no bucket view): the returning path takes over, nothing is written behind autograd's back
    c = _model(dev).train()
    c.precision = precision
    FusedAdam(c.parameters(), lr=1e-3).zero_grad()
    c.rgb.bias.requires_grad_(False)
    c(x1, d1).square().mean().backward()
    for (n, pc), g in zip(c.named_parameters(), g1):
        if pc.requires_grad:
            assert torch.equal(pc.grad, g), n


@pytest.mark.gpu
def test_grad_scale_is_one_launch_of_the_elementwise_chain():
    from fs_nerf_amd import ops
    dev = torch.device("cuda:0")

    def chain(d_out):  # what nine elementwise / reduction launches computed before
        amax = d_out.detach().abs().amax().to(torch.float32)
        e = torch.floor(torch.log2(1024.0 / amax))
        e = torch.nan_to_num(e, nan=0.0, posinf=0.0, neginf=0.0).clamp(-40.0, 60.0)
        return torch.exp2(e).reshape(1)

    gen = torch.Generator(device=dev).manual_seed(2)
    for n in (1, 63, 4096 * 192 * 4 + 3):
        for s in (1.0, 3e-9, 7e-31, 1e-44, 5e4, 3e17, 3e38):
            d = torch.randn(n, device=dev, generator=gen) * s
            got, want = ops.grad_scale_for(d), chain(d)
            assert got.shape == (1,) and float(got) == float(want), (n, s, float(got), float(want))
    for special in (0.0, float("inf"), float("nan"), -float("inf")):
        d = torch.randn(1000, device=dev, generator=gen)
        d[517] = special
        if special == 0.0:
            d.zero_()
        assert float(ops.grad_scale_for(d)) == 1.0 == float(chain(d))
