"""cProfile of the host side of the reference's training loop body (bench.py --workload train-occ) in steady state."""
import cProfile, os, pstats, sys, types, torch
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import bench as B
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cull = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "none" else None
a = types.SimpleNamespace(steps=60, warmup=40, precision="fp16x3", cull_precision=cull)
# profile only the timed region: wrap time.perf_counter boundaries by patching _gc_quiet/_gc_restore
pr = cProfile.Profile()
q0, r0 = B._gc_quiet, B._gc_restore
def q():
    q0(); pr.enable()
def r():
    pr.disable(); r0()
B._gc_quiet, B._gc_restore = q, r
line = B.train_occ_main(a, 0, 1, dev, None, "nccl")
print("ms_per_step", line["ms_per_step"], "kept", line["config"]["kept_samples_per_ray"])
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)

# second run: wall time inside the python backward functions (they run on autograd's thread, outside cProfile's view)
import time
from fs_nerf_amd.core import models as M
from fs_nerf_amd.render import rendering as Rm
acc = {}
def timed(name, fn):
    def w(*a_, **k_):
        t = time.perf_counter()
        try:
            return fn(*a_, **k_)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return staticmethod(w)
M._NerfTrainFn.backward = timed("NerfTrainFn.backward", M._NerfTrainFn.backward)
M._NerfTrainFn.forward = timed("NerfTrainFn.forward", M._NerfTrainFn.forward)
Rm._CompositeFn.backward = timed("CompositeFn.backward", Rm._CompositeFn.backward)
Rm._CompositeFn.forward = timed("CompositeFn.forward", Rm._CompositeFn.forward)
from fs_nerf_amd import ops
for nm in ("nerf_train_bwd", "nerf_train_fwd_rays", "occ_sample_fused", "composite_packed"):
    orig = getattr(ops, nm)
    def mk(nm, orig):
        def w(*a_, **k_):
            t = time.perf_counter()
            try:
                return orig(*a_, **k_)
            finally:
                acc["ops." + nm] = acc.get("ops." + nm, 0.0) + time.perf_counter() - t
        return w
    setattr(ops, nm, mk(nm, orig))
B._gc_quiet, B._gc_restore = q0, r0
a = types.SimpleNamespace(steps=60, warmup=40, precision="fp16x3", cull_precision=cull)
acc.clear()
line = B.train_occ_main(a, 0, 1, dev, None, "nccl")
print("ms_per_step", line["ms_per_step"])
print({k: round(v / 100 * 1e3, 3) for k, v in acc.items()}, "ms per step (100 steps incl. warm-up)")
