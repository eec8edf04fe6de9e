"""CPU, world_size 2 over gloo: the N>1 host path (row/frame sharding, image gather, max-over-ranks
timing) is correct by construction.  The HIP kernels themselves need no collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fs_nerf_amd  # noqa: F401
from fs_nerf_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        row0, nrows = shard.shard_rows(H, rank, world)
        # a "rendered" row block whose value encodes the global pixel index
        rows = torch.arange(row0, row0 + nrows, dtype=torch.float32)[:, None, None]
        cols = torch.arange(W, dtype=torch.float32)[None, :, None]
        local = (rows * W + cols).expand(nrows, W, 3).contiguous()
        img = shard.gather_rows(local, H)
        want = (torch.arange(H * W, dtype=torch.float32).reshape(H, W, 1)).expand(H, W, 3)
        ok = bool(torch.equal(img, want))
        t = shard.max_over_ranks(1.0 + rank, torch.device("cpu"))
        frames = list(shard.shard_frames(7, rank, world))
        q.put((rank, ok, t, row0, nrows, frames))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H", [8, 9])
def test_two_ranks_gloo(H):
    world, W = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "gathered image differs"
    assert all(r[2] == 2.0 for r in res), "max over ranks"
    assert sum(r[4] for r in res) == H and res[0][3] == 0 and res[1][3] == res[0][4]
    assert res[0][5] == [0, 2, 4, 6] and res[1][5] == [1, 3, 5]


def test_shard_rows_cover_exactly():
    for H in (1, 7, 800, 801, 1600):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard.shard_rows(H, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == H
            for (a, n), (b, _) in zip(blocks, blocks[1:]):
                assert a + n == b
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
