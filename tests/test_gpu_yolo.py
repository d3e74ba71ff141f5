"""GPU tests of the YOLOv8 detector path vs the CPU oracle (oracle/yolo_oracle.py).

PARITY UNPINNED: both sides restate ultralytics' published architecture (third-party, absent);
these tests prove the HIP implementation and the independent torch restatement agree, layer by
layer, and that TemporalDetector over the native backend behaves like the scripted reference runs.
Tolerance: activations abs <= 2e-4 * max(1,|ref|max) (fp32, ~60 layers deep, SiLU); boxes <= 2e-2 px; conf <= 1e-4.
"""
import os

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector, nms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    sd = synth.make_yolov8_state_dict(seed=7)
    return sd, YoloV8Detector(sd, device="cuda:0")


def frames(n, h=256, w=256, seed=5):
    return np.random.RandomState(seed).randint(0, 256, (n, h, w, 3), dtype=np.uint8)


def test_every_module_output_matches_oracle(det):
    import torch
    from oracle import yolo_oracle as Y
    sd, d = det
    fr = frames(2)
    best, pred = d.detect_batch(fr, conf=0.25, want_pred=True)
    with torch.no_grad():
        ref_pred, taps = Y.forward(sd, Y.preprocess_bgr(fr))
    names = ["model.0", "model.1", "model.2", "model.3", "model.4", "model.5", "model.6", "model.7", "model.8", "model.9",
             "model.12", "model.15", "model.16", "model.18", "model.19", "model.21", "box0", "cls0", "box1", "cls1", "box2", "cls2"]
    for n in names:
        ref = taps[n].numpy()
        got = d.activation(n, 2)
        assert got.shape == ref.shape, (n, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (n, err)
    cand = Y.candidates(sd, fr)
    assert pred.shape == cand.shape == (2, 1344, 5)
    assert np.abs(pred[..., :4] - cand[..., :4]).max() <= 2e-2
    assert np.abs(pred[..., 4] - cand[..., 4]).max() <= 1e-4
    for b in range(2):
        i = int(np.argmax(pred[b, :, 4]))
        assert pred[b, i, 4] > 0.25
        assert np.array_equal(best[b], pred[b, i])                      # device arg-max == arg-max of its own candidates
        assert abs(best[b, 4] - cand[b, :, 4].max()) <= 1e-4             # and agrees with the oracle's top confidence
        k = nms(pred[b, :, :4], pred[b, :, 4])
        assert k[0] == i                                                 # NMS never changes the top-1 box


def test_threshold_and_fused_weights_and_rect_shapes():
    from oracle import yolo_oracle as Y
    sd = synth.make_yolov8_state_dict(seed=3, cls_bias=-6.0, fused=True)   # confidences ~0.0025: nothing passes 0.25
    d = YoloV8Detector(sd, device="cuda:0")
    fr = frames(3, 160, 256, seed=9)                                       # rectangular (ultralytics auto letterbox shapes)
    best, pred = d.detect_batch(fr, conf=0.25, want_pred=True)
    assert np.all(best[:, 4] == -1.0)
    cand = Y.candidates(sd, fr)
    assert pred.shape == cand.shape == (3, 20 * 32 + 10 * 16 + 5 * 8, 5)
    assert np.abs(pred[..., :4] - cand[..., :4]).max() <= 2e-2 and np.abs(pred[..., 4] - cand[..., 4]).max() <= 1e-4
    best0 = d.detect_batch(fr, conf=0.0)
    assert np.all(best0[:, 4] > 0)
    xy, cf = d(fr[0], 0.25)
    assert xy.shape == (0, 4) and cf.shape == (0,)


def test_temporal_detector_over_native_backend(det):
    sd, d = det
    td = og.TemporalDetector(d, conf=0.25)
    fr = frames(6, seed=11)
    raw = d.detect_batch(fr, 0.25)
    outs = [td.detect(f) for f in fr]
    # replay the same raw detections through the (golden-tested) state machine
    td2 = og.TemporalDetector(lambda f, c: (np.zeros((0, 4), np.float32), np.zeros(0, np.float32)))
    exp = []
    for b in raw:
        if b[4] < 0:
            exp.append(td2.update(None, None, 256, 256))
        else:
            exp.append(td2.update(b[None, :4], b[4:5], 256, 256))
    assert outs == exp
    assert any(o is not None for o in outs)


def test_gated_pipeline_end_to_end(det):
    """YOLO+UNet (config C3): detector boxes gate the fused area count; equals mask[y1:y2,x1:x2] sums."""
    sd, d = det
    feats = (32, 64, 128, 256)
    m = og.UNet(1, 1, feats)
    m.load_state_dict(synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-2.5))
    m.to("cuda:0").eval()
    fr = frames(5, seed=21)
    from openglottal_amd.features import area_waveform
    from openglottal_amd.utils import bgr_to_gray
    wave = area_waveform(list(fr), og.TemporalDetector(d), m)
    masks, _, _ = m.segment(np.stack([bgr_to_gray(f) for f in fr]))
    td = og.TemporalDetector(d)
    for i, f in enumerate(fr):
        b = td.detect(f)
        want = 0.0 if b is None else float(np.sum(masks[i][b[1]:b[3], b[0]:b[2]] > 0))
        assert wave[i] == want


def test_large_batch_takes_the_occupancy_kernels_and_agrees_with_small_batches(det):
    """At >= 3 workgroups per CU a 3x3 conv launch switches from the persistent kernel to the occupancy kernel
    (lean epilogue: SiLU, residual, bounds-check clipping).  Same arithmetic order -> same predictions, frame by frame."""
    _, d = det
    f = frames(192, seed=11)
    best_big, pred_big = d.detect_batch(f, 0.25, want_pred=True)
    for lo in (0, 95, 190):
        b, p = d.detect_batch(f[lo:lo + 2], 0.25, want_pred=True)
        np.testing.assert_array_equal(pred_big[lo:lo + 2], p)
        np.testing.assert_array_equal(best_big[lo:lo + 2], b)
    # rectangular input whose deepest maps have an odd height (5 x 8): partial tiles and the half-row clipping
    f = frames(160, 160, 256, seed=12)
    best_big, pred_big = d.detect_batch(f, 0.25, want_pred=True)
    for lo in (0, 77, 158):
        b, p = d.detect_batch(f[lo:lo + 2], 0.25, want_pred=True)
        np.testing.assert_array_equal(pred_big[lo:lo + 2], p)
        np.testing.assert_array_equal(best_big[lo:lo + 2], b)


def test_device_entry_point_accepts_any_batch(det):
    """og_yolo_detect_u8_dev chains at most 512 frames per launch internally; 600 frames in one call == per-chunk calls."""
    import torch
    from openglottal_amd._lib import check, lib, ptr
    _, d = det
    f = torch.from_numpy(frames(600, seed=21)).cuda()
    best = torch.empty((600, 5), dtype=torch.float32, device="cuda")
    check(lib().og_yolo_detect_u8_dev(d._h, ptr(f), 600, 256, 256, 0.25, ptr(best), None), "detect")
    check(lib().og_yolo_sync(d._h), "sync")
    ref = d.detect_batch(f[:8].cpu().numpy(), 0.25)
    ref_tail = d.detect_batch(f[592:].cpu().numpy(), 0.25)
    got = best.cpu().numpy()
    np.testing.assert_array_equal(got[:8], ref)
    np.testing.assert_array_equal(got[592:], ref_tail)


def test_eval_harness_batches_the_detector_without_changing_results(det):
    """evaluate() runs the YOLO network for all frames in one device pass when the backend offers detect_batch; the
    per-frame path (backend wrapped so that it only has __call__) must give the same boxes, metrics and det stats."""
    from openglottal_amd import evaluate as E
    sd, d = det
    feats = (32, 64, 128, 256)
    m = og.UNet(1, 1, feats)
    m.load_state_dict(synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-2.5))
    m.to("cuda:0").eval()
    f = frames(24, seed=31)
    gt = (np.random.RandomState(2).rand(24, 256, 256) > 0.97).astype(np.uint8) * 255
    patients = ["a"] * 10 + ["b"] * 14
    batched = E.evaluate(list(f), list(gt), m, detector=og.TemporalDetector(d, conf=0.25), patients=patients)
    per_frame = E.evaluate(list(f), list(gt), m, detector=og.TemporalDetector(lambda fr, c: d(fr, c), conf=0.25), patients=patients)
    assert batched[2] == per_frame[2]
    for pipe in E.PIPELINES:
        assert batched[0][pipe]["dice"] == per_frame[0][pipe]["dice"] and batched[0][pipe]["n_det"] == per_frame[0][pipe]["n_det"]


def test_cli_run_both_pipelines_end_to_end(tmp_path):
    """`python -m openglottal_amd.cli run` (counterpart of openglottal/cli.py:46-103): .npy video + torch state_dict +
    detector .npz in, features.json out, for the U-Net-only and the detection-gated pipeline."""
    import json
    import torch
    from openglottal_amd import cli
    from openglottal_amd.features import extract_features_unet
    feats = (32, 64, 128, 256)
    sd = synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-2.5)
    torch.save(synth.state_dict_to_torch(sd), tmp_path / "unet.pt")
    ysd = synth.make_yolov8_state_dict(seed=7)
    np.savez(tmp_path / "yolo.npz", **ysd)
    gray, _ = synth.glottis_frames(2, 12)
    video = np.repeat(gray[..., None], 3, axis=-1)
    np.save(tmp_path / "video.npy", video)
    for pipe, extra in (("unet-only", []), ("unet", ["--yolo-weights", str(tmp_path / "yolo.npz")])):
        out = tmp_path / pipe
        rc = cli.main(["run", str(tmp_path / "video.npy"), "--pipeline", pipe, "--unet-weights", str(tmp_path / "unet.pt"),
                       "--device", "cuda:0", "-o", str(out)] + extra)
        assert rc in (0, 1)
        if rc == 0:
            got = json.load(open(out / "features.json"))
            # exactly the reference's payload (cli.py:97 dumps every key of the feature dict, `_area` as a list): the key
            # set of the reference-generated kinematic fixture + `_area`
            ref_keys = set(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kinematic.json")))["periodic"]["out"])
            assert set(got) == ref_keys | {"_area"}, set(got) ^ (ref_keys | {"_area"})
            assert isinstance(got["_area"], list) and len(got["_area"]) == 24
            m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
            det = og.TemporalDetector(str(tmp_path / "yolo.npz")) if pipe == "unet" else None
            ref = extract_features_unet(str(tmp_path / "video.npy"), det, m, "cuda:0")
            for k in ("area_mean", "area_std", "area_range", "open_quotient", "periodicity", "cv"):
                assert got[k] == pytest.approx(float(ref[k]), rel=0, abs=0)
            assert got["_area"] == ref["_area"].tolist()
    rc = cli.main(["run", str(tmp_path / "video.npy"), "--pipeline", "unet-only", "--unet-weights", str(tmp_path / "unet.pt"),
                   "--device", "cuda:0", "-o", str(tmp_path / "ann"), "--annotate"])
    if rc == 0:
        assert json.load(open(tmp_path / "ann" / "features.json"))["pipeline"] == "unet-only"


@pytest.mark.parametrize("shape", [(512, 512), (128, 128), (224, 352), (256, 256), (250, 300)])
def test_batched_and_per_frame_detector_paths_agree_on_any_frame_size(det, shape):
    """`area_waveform` / `evaluate` batch the YOLO network; `TemporalDetector.detect` runs it per frame.  ultralytics
    letterboxes every frame to the trained imgsz (256) and scales the boxes back, so both paths must do that — also
    where the letterbox is not the identity (512x512 -> gain 0.5, 128x128 -> gain 2, 352x224 -> 256x192 + no pad,
    300x250 -> padding)."""
    from openglottal_amd import evaluate as E
    from openglottal_amd.features import area_waveform
    sd, d = det
    H, W = shape
    fr = frames(6, H, W, seed=H + W)
    best = d.detect_frames(fr, 0.25)
    d.set_option("latency_batch", 0)                # one-frame calls on the batched kernels: bit for bit
    try:
        td = og.TemporalDetector(d, conf=0.25)
        per_frame = [td.detect(f) for f in fr]
        td2 = og.TemporalDetector(lambda f, c: None)
        batched = [td2.update(b[None, :4], b[4:5], W, H) if b[4] >= 0 else td2.update(None, None, W, H) for b in best]
        assert per_frame == batched
        for i, f in enumerate(fr):                      # raw boxes too, bit for bit (same letterbox, same scale-back)
            xy, cf = d(f, 0.25)
            assert (len(cf) == 0) == (best[i, 4] < 0)
            if len(cf):
                assert np.array_equal(xy[0], best[i, :4]) and cf[0] == best[i, 4]
                assert 0 <= xy[0, 0] <= xy[0, 2] <= W and 0 <= xy[0, 1] <= xy[0, 3] <= H
    finally:
        d.set_option("latency_batch", 1)
    # default: one-frame calls take the latency path (K split over workgroups -> float sums in another order):
    # the same boxes to rounding (<= 1e-3 px at the frame's scale, conf <= 1e-5)
    for i, f in enumerate(fr):
        xy, cf = d(f, 0.25)
        assert (len(cf) == 0) == (best[i, 4] < 0)
        if len(cf):
            assert np.abs(xy[0] - best[i, :4]).max() <= 1e-3 * max(1.0, max(H, W) / 256) and abs(cf[0] - best[i, 4]) <= 1e-5
    # the two callers: gated area waveform and the eval harness give what the per-frame loop gives
    feats = (4, 8, 16, 32)
    m = og.UNet(1, 1, feats)
    m.load_state_dict(synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-0.4))
    m.to("cuda:0").eval()
    wave = area_waveform(list(fr), og.TemporalDetector(d), m)
    td3 = og.TemporalDetector(d)
    from openglottal_amd.utils import bgr_to_gray
    for i, f in enumerate(fr):
        b = td3.detect(f)
        mk = og.unet_segment_frame(bgr_to_gray(f), m)
        assert wave[i] == (0.0 if b is None else float(np.sum(mk[b[1]:b[3], b[0]:b[2]] > 0))), i
    gts = np.zeros((6, H, W), np.uint8)
    agg, _, st = E.evaluate(list(fr), gts, m, detector=og.TemporalDetector(d), reset_every_frame=True)
    stateless = []
    for f in fr:
        t = og.TemporalDetector(d)
        stateless.append(t.detect(f))
    assert agg["yolo+unet"]["n_det"] == sum(b is not None for b in stateless)


NAMES = ["model.0", "model.1", "model.2", "model.3", "model.4", "model.5", "model.6", "model.7", "model.8", "model.9",
         "model.12", "model.15", "model.16", "model.18", "model.19", "model.21", "box0", "cls0", "box1", "cls1", "box2", "cls2"]


@pytest.mark.parametrize("shape,nb,latency_batch", [((256, 256), 1, 1), ((96, 160), 1, 1), ((320, 256), 1, 1), ((32, 32), 1, 1),
                                                     ((256, 256), 3, 4), ((512, 512), 1, 1)])
def test_latency_path_matches_oracle_and_batched_path(det, shape, nb, latency_batch):
    """One-frame calls (detector.py:58's pattern) take their own path: K split over workgroups with a fused reduce on every
    small conv, 32-column tiles, the Detect level's two branches as one chain (block-diagonal weights), one-launch SPPF
    pools, a multi-workgroup decode.  Every module output against the oracle at the batched path's tolerance, and the
    candidates / best box against the batched kernels on the same frames (they differ only in float summation order)."""
    import torch
    from oracle import yolo_oracle as Y
    sd, d = det
    H, W = shape
    fr = frames(nb, H, W, seed=H * 3 + W + nb)
    try:
        d.set_option("latency_batch", 0)
        best0, pred0 = d.detect_batch(fr, conf=0.25, want_pred=True)
        d.set_option("latency_batch", latency_batch)
        best1, pred1 = d.detect_batch(fr, conf=0.25, want_pred=True)
        with torch.no_grad():
            _, taps = Y.forward(sd, Y.preprocess_bgr(fr))
        for n in NAMES:
            ref = taps[n].numpy()
            got = d.activation(n, nb)
            assert got.shape == ref.shape, (n, got.shape, ref.shape)
            err = np.abs(got - ref).max()
            assert err <= 2e-4 * max(1.0, np.abs(ref).max()), (n, err)
    finally:
        d.set_option("latency_batch", 1)
    assert np.abs(pred1[..., :4] - pred0[..., :4]).max() <= 1e-3 * max(1.0, max(H, W) / 256)
    assert np.abs(pred1[..., 4] - pred0[..., 4]).max() <= 1e-5
    for b in range(nb):
        i = int(np.argmax(pred1[b, :, 4]))
        if pred1[b, i, 4] > 0.25:
            assert np.array_equal(best1[b], pred1[b, i])      # multi-workgroup arg-max == arg-max of its own candidates
        else:
            assert best1[b, 4] == -1
        assert abs(best1[b, 4] - best0[b, 4]) <= 1e-5


def test_latency_path_option_validation(det):
    sd, d = det
    for name, bad in (("latency_batch", 65), ("splitk_max", 0), ("splitk_min_steps", 2), ("nonsense", 1)):
        with pytest.raises(Exception):
            d.set_option(name, bad)
    d.set_option("latency_batch", 1)


def test_latency_path_interleaved_with_batched_calls_and_no_detection(det):
    """One handle, one arena: one-frame calls (fused Detect chain: other buffers than the batched layout) between batched calls
    give what they give on their own; a threshold nothing passes returns conf = -1 from the multi-workgroup decode too; the
    device entry point takes the same path."""
    import torch
    sd, d = det
    fr = frames(5, 256, 256, seed=77)
    big = d.detect_batch(fr, 0.25)
    one = [d.detect_batch(fr[i:i + 1], 0.25)[0] for i in range(5)]
    for rep in range(2):
        assert np.array_equal(d.detect_batch(fr, 0.25), big)
        for i in (3, 0, 4):
            assert np.array_equal(d.detect_batch(fr[i:i + 1], 0.25)[0], one[i])      # deterministic, whatever ran before
    assert np.abs(np.stack(one)[:, :4] - big[:, :4]).max() <= 1e-3 and np.abs(np.stack(one)[:, 4] - big[:, 4]).max() <= 1e-5
    none = d.detect_batch(fr[:1], 0.9999)
    assert none[0, 4] == -1 and not none[0, :4].any()
    dev = torch.from_numpy(fr).to("cuda:0")
    for i in range(3):
        assert np.array_equal(d.detect_dev(dev[i], 1, 256, 256, 0.25)[0], one[i])


def test_per_frame_and_batched_detector_are_bit_identical_on_2000_frames(det):
    """the detector's arithmetic is a property of the handle.  `TemporalDetector.detect(frame)` (one frame per
    call: K parts on separate workgroups, fused reduce) and `detect_frames(video)[i]` (batched: the same parts summed in registers,
    k_conv_mfma_o<..., VS>) return the same five floats BIT FOR BIT on 2 000 seeded frames -- so the truncations of
    openglottal/models/detector.py:68-69 (`int(x2 - x1)`) and :94-95 (`int(np.clip(...))`) see the same numbers, the integer boxes
    after `update()` are equal, and the gated areas of the per-frame loop equal the streamed ones (features.py:240-245)."""
    import openglottal_amd as og
    from openglottal_amd.features import area_waveform
    from openglottal_amd.utils import bgr_to_gray, normalize_box, unet_segment_frame

    sd, d = det
    n = 2000
    fr = np.stack([synth.bench_frame_bgr(i) for i in range(n)])
    fr[::7] = frames(len(fr[::7]), 256, 256, seed=5)              # every 7th frame structured: other boxes than the noise frames'
    batched = d.detect_frames(fr, 0.25)
    per_frame = np.stack([d.detect_frames(fr[i:i + 1], 0.25)[0] for i in range(n)])
    assert np.array_equal(batched, per_frame), int((batched != per_frame).any(axis=1).sum())
    assert int((batched[:, 4] >= 0).sum()) >= n // 2                      # the comparison is about real boxes
    for lb in (0, 4):                                                       # another scheduling of the same sums
        d.set_option("latency_batch", lb)
        try:
            assert np.array_equal(d.detect_frames(fr[:64], 0.25), batched[:64])
            assert np.array_equal(np.stack([d.detect_frames(fr[i:i + 3], 0.25) for i in range(0, 63, 3)]).reshape(-1, 5), batched[:63])
        finally:
            d.set_option("latency_batch", 1)
    ta, tb = og.TemporalDetector(d, conf=0.25), og.TemporalDetector(d, conf=0.25)
    boxes_a = [ta.detect(f) for f in fr]                                    # the reference's call, frame by frame
    boxes_b = [tb.update(batched[i:i + 1, :4], batched[i:i + 1, 4], 256, 256) if batched[i, 4] >= 0 else tb.update(None, None, 256, 256) for i in range(n)]
    assert boxes_a == boxes_b
    feats = (32, 64, 128, 256)
    m = og.UNet(1, 1, feats)
    m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506))
    m.to("cuda:0").eval()
    m.set_chunk(64)
    streamed = area_waveform(fr, og.TemporalDetector(d, conf=0.25), m)
    tc = og.TemporalDetector(d, conf=0.25)
    loop = np.zeros(n)
    for i, f in enumerate(fr):                                              # features.py:234-245, one call each per frame
        mask = unet_segment_frame(bgr_to_gray(f), m, "cuda:0")
        box = tc.detect(f)
        if box is not None:
            x1, y1, x2, y2 = normalize_box(box, 256, 256)
            loop[i] = float(np.sum(mask[y1:y2, x1:x2] > 0))
    assert np.array_equal(loop, streamed), int((loop != streamed).sum())
