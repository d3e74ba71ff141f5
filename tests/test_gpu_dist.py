"""-m gpu: the sharded frame loops with REAL device code on every rank.  The GPU box has one GPU, so the ranks
are separate processes sharing it and the (tiny) collectives go over gloo; on an 8-GPU node the same functions run one rank
per GPU over RCCL.  Equivalence bar: the gathered waveform equals the single-process waveform bit for bit, ragged N included."""
import os
import socket

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector

pytestmark = pytest.mark.gpu

N = 53   # ragged over 2 and 3 ranks


def _model():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    m = og.UNet(1, 1, tuple(int(f) for f in g["features"]))
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    m.set_chunk(8)
    return m, g


def _video():
    frames, _ = synth.glottis_frames(4, 20, seed=99)
    gray = frames[np.arange(N) % 80]
    rs = np.random.RandomState(1)
    bgr = np.clip(gray[..., None].astype(np.int32) + rs.randint(-3, 4, (N, 256, 256, 3)), 0, 255).astype(np.uint8)
    return gray, bgr


def _record_route(m):
    """Which engine entry a frame loop takes: [(method, input shape tail, boxes given)] per call into the model."""
    route = []
    for name in ("segment", "segment_stream"):
        orig = getattr(m, name)

        def wrapped(x, *a, _orig=orig, _name=name, **k):
            route.append((_name, int(np.asarray(x[0]).ndim), k.get("boxes") is not None))
            return _orig(x, *a, **k)

        setattr(m, name, wrapped)
    return route


def _rank(rank, world, port, q):
    import torch.distributed as dist

    from openglottal_amd.dist import sharded_area_waveform, sharded_gated_area_waveform

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, _ = _model()
    gray, bgr = _video()
    plain = sharded_area_waveform(gray, m, rank, world)
    y = YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device="cuda:0")
    route = _record_route(m)
    gated, boxes = sharded_gated_area_waveform(list(bgr), y.detect_frames, lambda: og.TemporalDetector(y), m, rank, world)
    q.put((rank, plain.tolist(), gated.tolist(), boxes.tolist(), route))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_waveforms_equal_single_process(world):
    import torch.multiprocessing as mp

    from openglottal_amd.features import area_waveform

    m, g = _model()
    gray, bgr = _video()
    ref_plain = area_waveform(gray, None, m)
    assert np.array_equal(ref_plain.astype(np.int64), g["areas"][np.arange(N) % 80])     # = the reference's integers
    y = YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device="cuda:0")
    route1 = _record_route(m)
    ref_gated = area_waveform(list(bgr), og.TemporalDetector(y), m)
    assert (ref_gated > 0).any()
    assert route1 == [("segment_stream", 3, True)], route1      # BGR frames + boxes into the streaming engine, one call
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(120)
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, plain, gated, boxes, route in res:
        assert plain == ref_plain.astype(np.int64).tolist(), rank
        assert gated == ref_gated.astype(np.int64).tolist(), rank
        assert [tuple(r) for r in route] == route1, (rank, route)      # the sharded gated loop takes the single-process loop's route
