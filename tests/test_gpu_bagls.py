"""-m gpu: BASELINE config C5's front end and evaluation loop on the device (scripts/eval_bagls.py).

* `k_canvas_letterbox` (2-D NEAREST / BGR LINEAR, mixed frame sizes) == `geometry.letterbox` pixel for pixel.  Both restate
  OpenCV's published resize rules (cv2 is absent from the image and the reference holds no fixture for it): PARITY UNPINNED
  against a real cv2; what is pinned is that the device path and the host path the other tests use are the same function.
* `og_mask_stats_dev` == numpy confusion counts; `evaluate_device` (everything resident in HBM, 40 bytes per frame back)
  == the host-orchestrated `evaluate(canvas=256, reset_every_frame=True)` per frame, for all three pipelines.
"""
import json
import os

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import evaluate as E
from openglottal_amd import synth
from openglottal_amd.geometry import letterbox
from openglottal_amd.yolo import YoloV8Detector

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trained(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    m = og.UNet(1, 1, tuple(int(f) for f in g["features"]))
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()


def test_device_canvas_letterbox_equals_host_geometry(trained):
    m = trained
    rs = np.random.RandomState(1)
    sizes = [(256, 256), (256, 512), (128, 512), (208, 352), (512, 256), (512, 128), (352, 208),   # BAGLS sizes, both orientations
             (301, 217), (17, 400), (255, 257), (1, 1), (600, 600), (256, 255), (3, 2), (1024, 768)]
    for canvas in (256, 320):
        gray = [rs.randint(0, 256, s, dtype=np.uint8) for s in sizes]
        bgr = [rs.randint(0, 256, s + (3,), dtype=np.uint8) for s in sizes]
        masks = [(rs.rand(*s) > 0.7).astype(np.uint8) * 255 for s in sizes]
        for name, imgs in (("gray", gray), ("bgr", bgr), ("mask", masks)):
            dev = m.canvas_letterbox(imgs, canvas)
            for i, im in enumerate(imgs):
                want = letterbox(im, canvas)
                assert np.array_equal(dev[i], want), (name, canvas, sizes[i], int((dev[i] != want).sum()))
    assert m.canvas_letterbox([], 256).shape == (0, 256, 256)
    out = m.canvas_letterbox([np.full((10, 40), 200, np.uint8)], 64, value=7)      # pad value
    assert out[0, 0, 0] == 7 and out[0, 32, 32] == 200
    with pytest.raises(og.OpenGlottalHipError):
        m.canvas_letterbox([np.zeros((4, 4), np.uint8), np.zeros((4, 4, 3), np.uint8)], 64)


def test_mask_stats_dev_equals_numpy(trained):
    import torch

    from openglottal_amd._lib import check, lib, ptr
    from openglottal_amd.utils import frame_metrics

    m = trained
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(2)
    pred = (rs.rand(9, 48, 80) > 0.6).astype(np.uint8) * 255
    gt = (rs.rand(9, 48, 80) > 0.7).astype(np.uint8) * 255
    pred[3] = 0; gt[3] = 0                     # both empty -> Dice = IoU = 1
    pred[4] = 0
    boxes = np.array([[5, 3, 60, 33]] * 9, np.int32)
    boxes[2] = -1
    d_p, d_g = torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev)
    for bx in (None, boxes):
        st = torch.empty((9, 3), dtype=torch.int32, device=dev)
        check(lib().og_mask_stats_dev(m._h, ptr(d_p), ptr(d_g), 9, 48, 80, None if bx is None else ptr(torch.from_numpy(bx).to(dev)), ptr(st)), "stats")
        m.sync()
        got = st.cpu().numpy()
        for i in range(9):
            p = pred[i]
            if bx is not None:
                p = np.zeros_like(pred[i])
                if bx[i][0] >= 0:
                    x1, y1, x2, y2 = bx[i]
                    p[y1:y2, x1:x2] = pred[i][y1:y2, x1:x2]
            want = [int(((p > 0) & (gt[i] > 0)).sum()), int((p > 0).sum()), int((gt[i] > 0).sum())]
            assert got[i].tolist() == want, (i, bx is not None)
            assert E.metrics_from_counts(*want) == frame_metrics(p, gt[i])      # the reference's float32 arithmetic


def test_evaluate_device_equals_host_orchestrated_evaluate(trained, tmp_path):
    m = trained
    frames, gts = synth.bagls_standin(61, seed=5)
    det = og.TemporalDetector(YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device="cuda:0"), conf=0.25)
    for crop_pad in (0, 6):
        agg_h, _, st_h = E.evaluate(frames, gts, m, detector=det, canvas=256, reset_every_frame=True, crop_pad=crop_pad)
        agg_d, st_d = E.evaluate_device(frames, gts, m, detector=det, canvas=256, crop_pad=crop_pad)
        assert st_d == st_h
        for p in E.PIPELINES:
            assert agg_d[p]["n_total"] == 61 and agg_d[p]["n_det"] == agg_h[p]["n_det"], p
            assert agg_d[p]["dice"] == agg_h[p]["dice"] and agg_d[p]["iou"] == agg_h[p]["iou"], (p, crop_pad)
        assert 0 < agg_d["yolo+unet"]["n_det"] <= 61
    # U-Net only (no detector): the other rows stay empty, as in the reference when --yolo-weights is not given
    agg_u, st_u = E.evaluate_device(frames[:10], gts[:10], m)
    assert agg_u["unet-only"]["n_total"] == 10 and agg_u["yolo+unet"]["n_total"] == 0 and st_u["n_pos_gt"] == 0
    # gray frames (2-D) take the NEAREST letterbox; a scripted (host) detector works through the same loop
    gframes = [f[..., 1] for f in frames[:12]]
    scripted = og.TemporalDetector(lambda f, c: (np.array([[60.0, 70.0, 190.0, 200.0]], np.float32), np.array([0.9], np.float32)))
    agg_g, _, st_g = E.evaluate(gframes, gts[:12], m, detector=scripted, canvas=256, reset_every_frame=True)
    agg_gd, st_gd = E.evaluate_device(gframes, gts[:12], m, detector=scripted, canvas=256)
    assert st_gd == st_g and all(agg_gd[p]["dice"] == agg_g[p]["dice"] for p in E.PIPELINES)
    # JSON in the shape of the reference's results/bagls_eval.json (eval_bagls.py:369-391)
    path = str(tmp_path / "out" / "bagls_eval.json")
    E.dump_json(path, agg_d, st_d, meta={"bagls_dir": "synthetic stand-in"})
    got = json.load(open(path))
    assert set(got) == {"unet-only", "yolo+unet", "yolo-crop+unet", "_meta"}
    for p in E.PIPELINES:
        assert set(got[p]) == {"dice", "iou", "n_det", "n_total"} and len(got[p]["dice"]) == got[p]["n_total"] == 61
    assert got["_meta"]["crop_letterbox"] is True
    E.print_table(agg_d)


def _c5_rank(rank, world, port, n, q):
    """One rank of the sharded BAGLS evaluation: own process, own handles on the (shared) GPU, gloo for the gather."""
    import torch.distributed as dist

    from openglottal_amd.dist import shard_range, sharded_eval_counts

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    m = og.UNet(1, 1, tuple(int(f) for f in g["features"]))
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    det = og.TemporalDetector(YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device="cuda:0"), conf=0.25)
    frames, gts = synth.bagls_standin(n, seed=5)
    counts = sharded_eval_counts(n, lambda lo, hi: E.evaluate_counts_device(frames[lo:hi], gts[lo:hi], m, det, None, 256, 0), rank, world)
    q.put((rank, counts.tolist()))
    dist.destroy_process_group()


def test_c5_sharded_over_two_ranks_equals_single_process(trained):
    """BASELINE config C5 sharded (here: two processes sharing the one GPU, gloo for the 40-byte-per-frame gather): every
    rank ends with the whole per-frame count table, equal to the single-process one, hence the same three-row table."""
    import socket

    import torch.multiprocessing as mp

    m = trained
    n = 37
    frames, gts = synth.bagls_standin(n, seed=5)
    det = og.TemporalDetector(YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device="cuda:0"), conf=0.25)
    ref = E.evaluate_counts_device(frames, gts, m, det, None, 256, 0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_c5_rank, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(120)
    assert sorted(r[0] for r in res) == [0, 1]
    for rank, counts in res:
        assert np.array_equal(np.array(counts), ref), rank
    agg, st = E.agg_from_counts(ref, True, True)
    agg1, st1 = E.evaluate_device(frames, gts, m, det, None, 256, 0)
    assert st == st1 and all(agg[p]["dice"] == agg1[p]["dice"] for p in E.PIPELINES)
