import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from fs_nerf_amd import _lib, ops
dev = torch.device("cuda:0")
coarse, fine = bench.init_sd(42), bench.init_sd(43)
for m in (coarse, fine):
    m.to(dev).eval()
pc, pf = coarse.packed(), fine.packed()
outs = {}
default = _lib.lib()
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    l = C.CDLL(os.path.abspath(path))
    for fn, (res, at) in _lib.SIGNATURES.items():
        f = getattr(l, fn); f.restype, f.argtypes = res, at
    _lib._lib = l
    cam = (bench.orbit_pose(8.0), 200, 200, 277.0, 0, 200, dev)
    outs[name] = ops.render_fused(pc, pf, None, None, near=2.0, far=6.0, n_samples=64, n_importance=128, bkgd=(1, 1, 1), want_extras=False, camera=cam, two_phase=True)
    _lib._lib = default
torch.cuda.synchronize()
names = list(outs)
for n in names[1:]:
    print(n, "== ", names[0], all(torch.equal(a, b) for a, b in zip(outs[n][:3], outs[names[0]][:3])), "status", int(ops.status_word(dev).item()))
