"""The host side of the weight packer (blob geometry + every index computation of the packed format) under
AddressSanitizer + UndefinedBehaviorSanitizer.  GPU ASan is not available on this pool, so this is the part of the
library that can run under a sanitizer: `make -C fs-nerf_amd/csrc sanitize` builds csrc/host_pack.cpp (plain C++, the
same headers the device packer uses) and tests/test_pack_layout.py runs against it in a child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fs-nerf_amd", "csrc")


def test_pack_layout_under_asan_ubsan():
    r = subprocess.run(["make", "-C", CSRC, "sanitize"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, FSN_LIB_PATH=os.path.join(CSRC, "libfsnerf_host_san.so"), FSN_LIB_PARTIAL="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_pack_layout.py"), "-x", "-q",
                          "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
