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
    """Stand-in for UNet.segment / UNet.segment_stream on CPU: 'area' = number of pixels of a fixed pattern inside the box."""

    def segment(self, gray, boxes=None, want_mask=False, **kw):
        out = np.zeros(len(gray), np.int32)
        for i, (g, b) in enumerate(zip(gray, boxes)):
            if b[0] >= 0:
                out[i] = int((g[b[1]:b[3], b[0]:b[2]] > 100).sum())
        return None, out, None

    def segment_stream(self, frames, threshold=0.5, boxes=None, want_mask=False):   # BGR in, as the streaming engine takes it
        return None, self.segment([f[..., 1] for f in frames], boxes=boxes)[1]


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


# ── C5: stateless (per-frame reset) evaluation sharded over ranks: one all-gather of integer count rows ──────────────


def _fake_counts(lo, hi):
    k = np.arange(lo, hi, dtype=np.int64)
    return np.stack([k * 7 % 1000, k * 7 % 1000 + k % 13, k * 5 % 900, k % 50, k % 60, k % 70, k % 80, k % 2, k % 3 == 0, k % 9 != 8], axis=1)


def _eval_worker(rank, world, port, n, q):
    from openglottal_amd.dist import sharded_eval_counts
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = sharded_eval_counts(n, _fake_counts, rank, world)
    q.put((rank, np.array_equal(got, _fake_counts(0, n))))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [3500, 37, 1])
def test_sharded_eval_counts_world2_equals_single_process(n):
    from openglottal_amd import evaluate as E
    from openglottal_amd.dist import sharded_eval_counts
    one = sharded_eval_counts(n, _fake_counts, 0, 1)
    assert np.array_equal(one, _fake_counts(0, n))
    agg, st = E.agg_from_counts(one, True, True)          # the table is a pure function of the gathered rows
    assert agg["yolo-crop+unet"]["n_total"] == n and len(agg["unet-only"]["dice"]) == n
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_eval_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


# ── world 8 (one rank per GPU of the node) rehearsed on CPU: BASELINE config C4's N = 10 000 and a ragged N ─────────


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _w8_worker(rank, world, port, n, q):
    import openglottal_amd as og
    from openglottal_amd.dist import all_gather_rows, sharded_gated_area_waveform
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = (np.arange(n, dtype=np.int64) * 37 % 65537).astype(np.int32)
    lo, hi = shard_range(n, rank, world)
    wave = all_gather_areas(torch.from_numpy(full[lo:hi].copy()), n)          # plain: int32 areas (C4)
    rows = all_gather_rows(torch.from_numpy(np.stack([full[lo:hi]] * 5, 1).astype(np.float32) / 4), n, 5)   # gated: 5-float boxes
    ok = np.array_equal(wave.numpy(), full) and np.array_equal(rows.numpy()[:, 3], full.astype(np.float32) / 4)
    # gated pipeline end to end on a short ragged video (state machine replayed on every rank)
    m = 43
    gw, gb = sharded_gated_area_waveform(list(_frames(m)), _fake_detect, lambda: og.TemporalDetector(lambda f, c: None), _FakeModel(), rank, world)
    q.put((rank, bool(ok), gw.tolist(), gb.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [10000, 10003])
def test_world8_plain_and_gated_gathers_equal_single_process(n):
    import openglottal_amd as og
    from openglottal_amd.dist import sharded_gated_area_waveform
    ref_w, ref_b = sharded_gated_area_waveform(list(_frames(43)), _fake_detect, lambda: og.TemporalDetector(lambda f, c: None), _FakeModel(), 0, 1)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_w8_worker, args=(r, 8, port, n, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, ok, gw, gb in res:
        assert ok, rank
        assert gw == ref_w.tolist() and gb == ref_b.tolist(), rank
