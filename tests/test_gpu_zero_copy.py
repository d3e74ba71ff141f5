"""-m gpu: the per-frame call's zero-copy path (option "zero_copy", default 1) against the copy-command path, bit for bit.

A call that is ONE micro-batch of at most 4 frames (the reference's `unet_segment_frame(gray)` per frame, utils.py:235-237, and
`TemporalDetector.detect(frame)`, detector.py:58) lets the kernels read the frame from, and write the mask / area / best box to, the
engine's pinned host buffers directly.  Everything the call returns must be what the H2D / D2H path returns: gray and BGR input,
with and without boxes, mask-only / area-only calls, frame lists, a caller-pinned source at an offset, and calls of 5 frames
(which take the copy path) in between.
"""
import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray, unet_segment_frame

pytestmark = pytest.mark.gpu

FEATS = (32, 64, 128, 256)


@pytest.fixture(scope="module")
def net():
    sd = synth.make_unet_state_dict(FEATS, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)   # bench.py's net
    m = og.UNet(1, 1, FEATS)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    bgr = np.stack([synth.bench_frame_bgr(i) for i in range(12)])
    return m, bgr, bgr_to_gray(bgr)


def _both(m, fn):
    out = []
    for zc in (0, 1):
        m.set_option("zero_copy", zc)
        out.append(fn())
    m.set_option("zero_copy", 1)
    return out


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            assert np.array_equal(x, y)


def test_host_batch_calls_of_one_to_five_frames(net):
    m, bgr, gray = net
    boxes = np.array([[10, 20, 200, 220], [-1, -1, -1, -1], [0, 0, 256, 256], [100, 100, 101, 101], [30, 5, 90, 250]], np.int32)
    for B in (1, 2, 4, 5):
        for bx in (None, boxes[:B]):
            for want_mask, want_area in ((True, True), (True, False), (False, True)):
                a, b = _both(m, lambda: m.segment(gray[:B], boxes=bx, want_mask=want_mask, want_area=want_area))
                _same(a, b)
    ref_mask, ref_area, _ = m.segment(gray[:5])
    for i in range(5):     # frame by frame == batched (the canonical form), through the zero-copy path
        mk, ar, _ = m.segment(gray[i:i + 1])
        assert np.array_equal(mk[0], ref_mask[i]) and ar[0] == ref_area[i]
        assert np.array_equal(unet_segment_frame(gray[i], m), ref_mask[i])


def test_streamed_bgr_calls_lists_and_a_pinned_source(net):
    import torch

    m, bgr, gray = net
    ref_mask, ref_area, _ = m.segment(gray)
    for B in (1, 3, 4, 5):
        a, b = _both(m, lambda: m.segment_stream(bgr[:B], want_mask=True))
        _same(a, b)
        assert np.array_equal(a[0], ref_mask[:B]) and np.array_equal(a[1], ref_area[:B])
        a, b = _both(m, lambda: m.segment_stream([f for f in bgr[:B]], want_mask=True))           # the reference's `frames_bgr` list
        _same(a, b)
        assert np.array_equal(a[1], ref_area[:B])
    pinned = torch.from_numpy(bgr).pin_memory()
    for lo, B in ((0, 1), (3, 2), (7, 4)):       # DMA'd in place by the copy path, read in place by the zero-copy path
        a, b = _both(m, lambda: m.segment_stream(pinned[lo:lo + B], want_mask=True))
        _same(a, b)
        assert np.array_equal(a[0], ref_mask[lo:lo + B]) and np.array_equal(a[1], ref_area[lo:lo + B])
    # interleaved with a longer call (ring of several slots) and back
    _, ar = m.segment_stream(bgr)
    assert np.array_equal(ar, ref_area)
    _, ar1 = m.segment_stream(bgr[5:6])
    assert ar1[0] == ref_area[5]


def test_detector_one_frame_call(net):
    from openglottal_amd.yolo import YoloV8Detector

    m, bgr, gray = net
    det = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
    outs = []
    for zc in (0, 1):
        det.set_option("zero_copy", zc)
        outs.append([det.detect_batch(bgr[i:i + 1], conf=0.001) for i in range(4)])      # one frame per call: the latency path
    det.set_option("zero_copy", 1)
    batched = det.detect_batch(bgr[:4], conf=0.001)
    for i, (x, y) in enumerate(zip(*outs)):
        assert np.array_equal(np.asarray(x), np.asarray(y))
        assert np.array_equal(np.asarray(y)[0], np.asarray(batched)[i])                 # and == the batched call


def test_detector_submit_result_equals_detect_with_a_unet_call_in_between(net):
    """`TemporalDetector.submit(frame)` ... `result()` (og_yolo_detect_u8_begin / _end) returns what `detect(frame)` returns, state
    machine included, with the U-Net call of the same frame between the two halves; misuse is an error, not a hang."""
    from openglottal_amd import TemporalDetector
    from openglottal_amd.yolo import YoloV8Detector

    m, bgr, gray = net
    sd = synth.make_yolov8_state_dict(seed=7)
    a = TemporalDetector(YoloV8Detector(sd, device="cuda:0"), conf=0.001)
    b = TemporalDetector(YoloV8Detector(sd, device="cuda:0"), conf=0.001)
    ref_mask, _, _ = m.segment(gray)
    for i in range(len(bgr)):
        want = a.detect(bgr[i])
        b.submit(bgr[i])
        mask = unet_segment_frame(gray[i], m)
        got = b.result()
        assert got == want, (i, got, want)
        assert np.array_equal(mask, ref_mask[i])
    odd = np.random.RandomState(3).randint(0, 256, (200, 312, 3), dtype=np.uint8)     # a frame the detector letterboxes
    want = a.detect(odd)
    b.submit(odd)
    assert b.result() == want
    b.model.submit(bgr[0], 0.25)
    with pytest.raises(og.OpenGlottalHipError):
        b.model.submit(bgr[1], 0.25)                 # one call in flight per handle
    with pytest.raises(og.OpenGlottalHipError):
        b.model.detect_batch(bgr[:1])
    b.model.result()
    with pytest.raises(og.OpenGlottalHipError):
        b.model.result()
