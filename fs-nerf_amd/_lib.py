"""ctypes binding of libfsnerf_hip.so (C-ABI: include/fsnerf_hip.h).

There is no CPU fallback: if the library is missing or a call fails, a RuntimeError
is raised.  PyTorch is only used for device memory and streams."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("FSN_LIB_PATH", os.path.join(CSRC, "libfsnerf_hip.so"))  # override: A/B builds

FSN_PREC_BF16X3 = 0
FSN_PREC_BF16 = 1
FSN_PREC_FP16X3 = 2
FSN_PREC_FP16 = 3
FSN_PREC_FP16X3U = 4  # inference: fp16 x 3 passes, unscaled low parts, per-layer activation scales folded into the blob
FSN_PREC_FP16X2 = 6  # two passes (weights high part only), inference only
FSN_STATUS_FP16_RANGE = 1  # a value reached fp16 infinity
FSN_STATUS_FP16_SMALL = 2  # a layer's activations were all below 2^-14: outside the split's float32-grade envelope
FSN_STATUS_GRAD_RANGE = 4  # backward with per-stage scales: a stored gradient overflowed (skipped step, scale drops; no fallback)


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("d_hidden", C.c_int32), ("skip_mask", C.c_uint32),
                ("n_freqs_pos", C.c_int32), ("n_freqs_dir", C.c_int32),
                ("freqs_pos", C.c_float * 16), ("freqs_dir", C.c_float * 16)]


class RenderArgs(C.Structure):
    _fields_ = [("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("R", C.c_int64),
                ("near", C.c_float), ("far", C.c_float), ("S", C.c_int32), ("n_imp", C.c_int32),
                ("u_mode", C.c_int32), ("u", C.c_void_p), ("u_fine", C.c_void_p),
                ("pos_mask", C.c_void_p), ("dir_mask", C.c_void_p), ("bkgd", C.c_float * 3),
                ("colors", C.c_void_p), ("opacity", C.c_void_p), ("depth", C.c_void_p),
                ("weights", C.c_void_p), ("alphas", C.c_void_p), ("trans", C.c_void_p),
                ("sigmas", C.c_void_p), ("rgbs", C.c_void_p), ("edges_out", C.c_void_p),
                ("weights_coarse", C.c_void_p), ("status", C.c_void_p),
                ("cam_pose", C.c_float * 12), ("cam_H", C.c_int32), ("cam_W", C.c_int32), ("cam_row0", C.c_int32),
                ("cam_focal", C.c_double), ("two_phase", C.c_int32), ("clock_out", C.c_void_p)]


class OccRenderArgs(C.Structure):
    _fields_ = [("rays_o", C.c_void_p), ("rays_d", C.c_void_p), ("R", C.c_int64), ("aabb", C.c_float * 6),
                ("res", C.c_int32), ("levels", C.c_int32), ("bits", C.c_void_p),
                ("near_plane", C.c_float), ("far_plane", C.c_float), ("step", C.c_float), ("u", C.c_void_p),
                ("max_steps", C.c_int32), ("early_stop_eps", C.c_float), ("alpha_thre", C.c_float),
                ("pos_mask", C.c_void_p), ("dir_mask", C.c_void_p), ("bkgd", C.c_float * 3),
                ("colors", C.c_void_p), ("opacity", C.c_void_p), ("depth", C.c_void_p),
                ("n_cand", C.c_void_p), ("n_kept", C.c_void_p), ("status", C.c_void_p), ("work_counter", C.c_void_p),
                ("cam_pose", C.c_float * 12), ("cam_H", C.c_int32), ("cam_W", C.c_int32), ("cam_row0", C.c_int32),
                ("cam_focal", C.c_double), ("sample_t0", C.c_void_p), ("sample_cap", C.c_int32),
                ("ex_weights", C.c_void_p), ("ex_alphas", C.c_void_p), ("ex_trans", C.c_void_p), ("ex_sigmas", C.c_void_p),
                ("ex_rgbs", C.c_void_p)]


_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_PD = C.POINTER(MlpDesc)

# name -> (restype, argtypes); every symbol declared in include/fsnerf_hip.h
SIGNATURES = {
    "fsn_version": (_i, []),
    "fsn_last_error": (C.c_char_p, []),
    "fsn_device_cus": (_i, []),
    "fsn_debug_report": (_i, [_vp]),
    "fsn_debug_selftest": (_i, []),
    "fsn_get_rays": (_i, [_vp, _i, _i, _d, _i, _i, _vp, _vp, _vp]),
    "fsn_to_ndc": (_i, [_vp, _vp, _i64, _i, _i, _d, _d, _vp, _vp, _vp]),
    "fsn_build_rays": (_i, [_vp, _i64, _i, _i, _d, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "fsn_posenc_fwd": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "fsn_stratified_edges": (_i, [_f, _f, _i, _i64, _vp, _i, _vp, _vp]),
    "fsn_edges_to_packed": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "fsn_sample_pdf_merge": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    "fsn_composite_fwd": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsn_composite_packed_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsn_mlp_blob_bytes": (_i64, [_PD, _i]),
    "fsn_mlp_pack": (_i, [_PD, _i, _vp, _vp, _vp, _vp]),
    "fsn_mlp_pack_host": (_i, [_PD, _i, _vp, _vp, _vp]),
    "fsn_mlp_pack_scaled": (_i, [_PD, _i, _vp, _vp, _vp, _vp, _vp]),
    "fsn_mlp_pack_scaled_host": (_i, [_PD, _i, _vp, _vp, _vp, _vp]),
    "fsn_mlp_layer_maxima": (_i, [_PD, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    "fsn_mlp_fwd": (_i, [_PD, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "fsn_mlp_fwd_rays": (_i, [_PD, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _vp, _vp, _vp]),
    "fsn_render_rays_fused": (_i, [_PD, _i, _vp, _vp, C.POINTER(RenderArgs), _vp]),
    "fsn_bench_bare_stream": (_i, [_PD, _i, _vp, _i, _vp, _vp]),
    "fsn_render_rays_occgrid": (_i, [_PD, _i, _vp, C.POINTER(OccRenderArgs), _vp]),
    "fsn_occlusion_reg_fwd": (_i, [_vp, _vp, _vp, _i64, _i64, _f, _f, _i, _vp, _vp, _vp]),
    "fsn_to8b": (_i, [_vp, _i64, _vp, _vp]),
    "fsn_to8b_nchw": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "fsn_depth_colormap": (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "fsn_nerf_train_workspace_floats": (_i64, [_PD, _i, _i64]),
    "fsn_nerf_train_fwd": (_i, [_PD, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fsn_nerf_train_fwd_rays": (_i, [_PD, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fsn_nerf_train_bwd": (_i, [_PD, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fsn_grad_scale": (_i, [_vp, _i64, _vp, _vp]),
    "fsn_occlusion_reg_bwd": (_i, [_vp, _i64, _vp, _i64, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "fsn_occ_gather_samples": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp, _vp]),
    "fsn_occ_gather_extras": (_i, [_vp, _vp, _vp, _i, _i64, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsn_occgrid_march": (_i, [_vp, _vp, _i64, _vp, _i, _i, _vp, _f, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fsn_packed_visibility": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _f, _f, _vp, _vp]),
    "fsn_occgrid_update": (_i, [_vp, _i64, _vp, _vp, _i64, _f, _vp, _vp, _vp]),
    "fsn_occgrid_select": (_i, [_vp, _i, _i, _i, _vp, _i, _i64, _i64, C.c_uint64, _vp, _vp, _vp, _vp]),
    "fsn_occgrid_update_multi": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _f, _vp]),
    "fsn_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _d, _d, _d, _d, _d, _d, _vp, _vp, _vp]),
    "fsn_adam_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _d, _d, _d, _d, _d, _d, _vp, _vp, _vp]),
    "fsn_weight_norm_workspace_floats": (_i64, [_i, _vp]),
    "fsn_weight_norm_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "fsn_weight_norm_bwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fsn_composite_packed_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libfsnerf_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HIP path)")
        l = C.CDLL(LIB_PATH)
        partial = os.environ.get("FSN_LIB_PARTIAL") == "1"  # the sanitizer build of the host side exports a subset
        for name, (res, args) in SIGNATURES.items():
            if partial and not hasattr(l, name):
                continue
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fsn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
