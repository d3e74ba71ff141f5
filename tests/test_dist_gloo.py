"""World-size-2 gloo test (CPU) of the frame sharding + area-waveform all-gather used at N>1."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openglottal_amd.dist import all_gather_areas, shard_range


def test_shard_range_covers_every_frame_once():
    for n in (0, 1, 7, 8, 9, 80, 502, 10000):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                got.extend(range(lo, hi))
            assert got == list(range(n)), (n, world)


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = (np.arange(n) * 37 % 1009).astype(np.int32)  # stand-in for per-frame areas
    lo, hi = shard_range(n, rank, world)
    wave = all_gather_areas(torch.from_numpy(full[lo:hi].copy()), n)
    q.put((rank, wave.numpy().tolist() == full.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [11, 502])
def test_all_gather_areas_world2_equals_single_process(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_passthrough():
    a = torch.arange(5, dtype=torch.int32)
    assert all_gather_areas(a, 5).tolist() == [0, 1, 2, 3, 4]
