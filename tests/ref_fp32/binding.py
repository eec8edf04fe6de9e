"""TEST-ONLY binding of tests/ref_fp32/libfsnerf_ref_fp32.so: the plain-fp32 formulation of NeRF.forward / backward
(rocBLAS sgemm + elementwise kernels) the MFMA training kernels were first checked against.  Nothing under
fs-nerf_amd/ imports or can reach this module; the product has no fp32 training mode."""
import ctypes as C
import os

import torch

from fs_nerf_amd import _lib as L
from fs_nerf_amd import ops

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfsnerf_ref_fp32.so")
_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
_PD = C.POINTER(L.MlpDesc)
_ref = None


def lib() -> C.CDLL:
    global _ref
    if _ref is None:
        if not os.path.exists(PATH):
            raise RuntimeError(f"{PATH} not found: run `make -C tests/ref_fp32` (__graft_entry__.build() does)")
        l = C.CDLL(PATH)
        l.fsn_last_error.restype = C.c_char_p
        l.fsnref_train_workspace_floats.restype, l.fsnref_train_workspace_floats.argtypes = _i64, [_PD, _i64]
        l.fsnref_train_fwd.restype = _i
        l.fsnref_train_fwd.argtypes = [_PD, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]
        l.fsnref_train_bwd.restype = _i
        l.fsnref_train_bwd.argtypes = [_PD, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
        _ref = l
    return _ref


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {lib().fsn_last_error().decode('utf-8', 'replace')}")


class _RefTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, dirs, *params):
        n_p = len(params) // 2
        ws = [ops._f32(w.detach(), "weight") for w in params[:n_p]]
        bs = [ops._f32(b.detach(), "bias") for b in params[n_p:]]
        desc = ops.make_desc(model.n_layers, model.d_hidden, model.skip, model.pos_encoder.freqs, model.dir_encoder.freqs)
        x, d = ops._f32(x, "x").reshape(-1, 3), ops._f32(dirs, "dirs").reshape(-1, 3)
        n = x.shape[0]
        dev = x.device
        pm, dm = model._mask(model.pos_mask, dev), model._mask(model.dir_mask, dev)
        out = torch.empty(n, 4, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            nfl = lib().fsnref_train_workspace_floats(C.byref(desc), n)
            if nfl < 0:
                _check(int(nfl), "fsnref_train_workspace_floats")
            work = torch.empty(max(int(nfl), 1), device=dev, dtype=torch.float32)
            _check(lib().fsnref_train_fwd(C.byref(desc), ops._ptr_array(ws), ops._ptr_array(bs), ops._p(x), ops._p(d),
                                          ops._p(pm), ops._p(dm), n, ops._p(work), ops._p(out), ops._stream()),
                   "fsnref_train_fwd")
        ctx.desc, ctx.ws, ctx.work, ctx.out, ctx.keep = desc, ws, work, out, (x, d, pm, dm)
        return out.reshape(*x.shape[:-1], 4)

    @staticmethod
    def backward(ctx, d_out):
        d_out = ops._f32(d_out, "d_out").reshape(-1, 4)
        n = d_out.shape[0]
        dW = [torch.empty_like(w) for w in ctx.ws]
        db = [torch.empty(w.shape[0], device=w.device, dtype=torch.float32) for w in ctx.ws]
        with torch.cuda.device(d_out.device):
            _check(lib().fsnref_train_bwd(C.byref(ctx.desc), ops._ptr_array(ctx.ws), n, ops._p(ctx.work), ops._p(ctx.out),
                                          ops._p(d_out), ops._ptr_array(dW), ops._ptr_array(db), ops._stream()),
                   "fsnref_train_bwd")
        return (None, None, None, *dW, *[g.reshape(-1) for g in db])


def forward(model, x, dirs):
    """`model(x, dirs)` with gradients, evaluated by the plain-fp32 reference formulation instead of the MFMA kernels."""
    ws, bs = model._tensors()
    return _RefTrainFn.apply(model, x, dirs, *ws, *bs)
