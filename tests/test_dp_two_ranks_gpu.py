"""-m gpu: two data-parallel ranks of the REAL HIP training step (VERDICT r3 next #6 iii; SURVEY 8e: rays shard, ONE
all-reduce of the flat gradient bucket per step, nothing else crosses ranks).  Two fresh child processes share the one
GPU of the box and talk over gloo on 127.0.0.1 - the rehearsal DESIGN.md 8 describes, as a test the driver runs.  (RCCL
itself needs two devices; the bucket, the flag slot, the fused optimizer and the kernels are the ones an 8-GPU run uses.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_two_ranks_train_in_lockstep_with_one_allreduce_per_step(tmp_path):
    port = _free_port()
    procs, outs = [], []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        outs.append(str(tmp_path / f"rank{r}.json"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_rank_worker.py"), outs[-1]], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    r0, r1 = (json.load(open(o)) for o in outs)
    assert r0["rank"] == 0 and r1["rank"] == 1
    # identical replicas at the start, bit-identical parameters on both ranks after every step (the ranks saw different
    # rays: only the all-reduced bucket can have made them equal)
    assert r0["hashes"] == r1["hashes"]
    h = r0["hashes"]
    assert h[1] != h[0] and h[3] != h[2], "clean steps update"
    assert h[2] == h[1], "the step rank 1 flagged is skipped on BOTH ranks (the flag travels in the bucket's extra slot)"
    assert r0["steps_applied"] == r1["steps_applied"] == [1, 1, 2], "... and does not advance the device-side step counters"
    assert r0["all_reduce_per_step"] == r1["all_reduce_per_step"] == [1, 1, 1], "ONE collective per step"
    assert r0["grad_hash"] == r1["grad_hash"] and r0["precision"] == r1["precision"] == "fp16x3"
