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


# ── gated pipeline: all-gather of per-frame best boxes, state machine replayed on every rank ──────────


class _FakeModel:
    """Stand-in for UNet.segment on CPU: 'area' = number of pixels of a fixed pattern inside the box."""

    def segment(self, gray, boxes=None, want_mask=False, **kw):
        out = np.zeros(len(gray), np.int32)
        for i, (g, b) in enumerate(zip(gray, boxes)):
            if b[0] >= 0:
                out[i] = int((g[b[1]:b[3], b[0]:b[2]] > 100).sum())
        return None, out, None


def _fake_detect(frames, conf):
    best = np.full((len(frames), 5), -1.0, np.float32)
    for i, f in enumerate(frames):
        k = int(f[0, 0, 0])           # frame id encoded in a pixel
        if k % 4 != 3:                # every 4th frame: no detection
            best[i] = [40 + k % 7, 50 + k % 5, 120 + k % 9, 160 + k % 11, 0.9]
    return best


def _frames(n):
    rs = np.random.RandomState(0)
    fr = rs.randint(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    fr[:, 0, 0, 0] = np.arange(n) % 256
    return fr


def _gated_worker(rank, world, port, n, q):
    import openglottal_amd as og
    from openglottal_amd.dist import sharded_gated_area_waveform
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wave, boxes = sharded_gated_area_waveform(list(_frames(n)), _fake_detect, lambda: og.TemporalDetector(lambda f, c: None),
                                              _FakeModel(), rank, world)
    q.put((rank, wave.tolist(), boxes.tolist()))
    dist.destroy_process_group()


def test_gated_waveform_world2_equals_single_process():
    import openglottal_amd as og
    from openglottal_amd.dist import sharded_gated_area_waveform
    n = 23
    ref_wave, ref_boxes = sharded_gated_area_waveform(list(_frames(n)), _fake_detect, lambda: og.TemporalDetector(lambda f, c: None),
                                                      _FakeModel(), 0, 1)
    assert (ref_wave > 0).any() and (ref_boxes[:, 0] >= 0).any()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_gated_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, wave, boxes in res:
        assert wave == ref_wave.tolist() and boxes == ref_boxes.tolist(), rank
