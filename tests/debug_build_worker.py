"""Worker of tests/test_debug_build.py: runs small-shape parity cases of the fused render kernels and of the occupancy
kernel through the DEBUG library (FSN_LIB_PATH -> libfsnerf_hip_dbg.so: every LDS index of k_render_fused / k_render_occ
range-checked, csrc/common.hpp) and prints the violation records as JSON."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def report(L):
    buf = (C.c_uint32 * 8)()
    L.check(L.lib().fsn_debug_report(buf), "fsn_debug_report")
    return list(buf)


def main():
    import fs_nerf_amd  # noqa: F401
    from fs_nerf_amd import _lib as L
    assert "dbg" in os.path.basename(L.LIB_PATH), L.LIB_PATH
    dev = torch.device("cuda:0")
    out = {}
    # negative control first: the checks record
    L.check(L.lib().fsn_debug_selftest(), "fsn_debug_selftest")
    out["selftest"] = report(L)
    out["after_selftest"] = report(L)  # cleared
    import test_parity_fp64 as T
    import test_occ_fused as F
    for name in ("C1", "C2", "C3", "C4", "C5"):
        for jitter in (True, False):
            hip, o32, truth = T.run_case(name, dev, "fp16x3", R=64, jitter=jitter)
            T.assert_parity(hip, o32, truth, f"debug build {name}")
    hip, o32, truth = T.run_case("C3", dev, "bf16x3", R=64)
    T.assert_parity(hip, o32, truth, "debug build C3 bf16x3", factor=10.0, atol_scale=3.0)
    T.test_single_pass_two_groups_per_wave_ragged_shapes(dev, "bf16")
    T.test_two_phase_and_camera_modes_are_bitwise_the_plain_launch(dev)
    out["render"] = report(L)
    F.test_fused_occupancy_launch_is_the_unfused_sequence(dev, (4, 128), 32, 1, 2e-2, False)
    F.test_fused_occupancy_launch_is_the_unfused_sequence(dev, (8, 256), 32, 1, 2e-2, False)
    F.test_fused_occupancy_batches_carry_and_empty_rays(dev)
    F.test_fused_occupancy_sampler_is_the_unfused_sampling(dev, (4, 128), 32, 1, 2e-2, True)
    out["occ"] = report(L)
    print("DEBUG_REPORT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
