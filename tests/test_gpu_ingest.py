"""-m gpu: device-side frame ingest and the gated recount on resident masks.

`og_bgr2gray_dev` must equal the host restatement `utils.bgr_to_gray` bit for bit (both restate OpenCV's published
u8 BGR2GRAY; OpenCV itself is absent: parity unpinned, exact for R = G = B by construction).  `og_mask_area_dev` must
equal numpy's `mask[y1:y2, x1:x2] > 0` count (features.py:244-245), incl. the reference-generated gated areas.
"""
import json
import os

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray, normalize_box

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trained(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    m = og.UNet(1, 1, tuple(int(f) for f in g["features"]))
    m.load_state_dict(sd)
    frames, gt = synth.glottis_frames(4, 20, seed=99)
    return g, m.to("cuda:0").eval(), frames


def test_bgr2gray_dev_equals_host_restatement(trained):
    import torch

    g, m, frames = trained
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(12)
    for (B, H, W) in [(5, 256, 256), (3, 208, 352), (1, 1, 7), (2, 33, 65)]:
        bgr = rs.randint(0, 256, (B, H, W, 3), dtype=np.uint8)
        bgr[0, 0, :3] = [[0, 0, 0], [255, 255, 255], [255, 0, 0]][:min(3, W)]      # extremes
        d_bgr = torch.from_numpy(bgr).to(dev)
        d_gray = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
        m.bgr2gray_dev(d_bgr, B, H, W, d_gray)
        m.sync()
        assert np.array_equal(d_gray.cpu().numpy(), bgr_to_gray(bgr)), (B, H, W)
    # every (b,g,r) with b=g=r maps to itself; all 256^2 (g, r) pairs at b = 77 against the host formula
    gg, rr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    bgr = np.stack([np.full_like(gg, 77), gg, rr], axis=-1)[None]
    d_gray = torch.zeros((1, 256, 256), dtype=torch.uint8, device=dev)
    m.bgr2gray_dev(torch.from_numpy(bgr).to(dev), 1, 256, 256, d_gray)
    m.sync()
    assert np.array_equal(d_gray.cpu().numpy(), bgr_to_gray(bgr))
    v = np.arange(256, dtype=np.uint8)
    same = np.stack([v, v, v], axis=-1)[None, None]
    d_g = torch.zeros((1, 1, 256), dtype=torch.uint8, device=dev)
    m.bgr2gray_dev(torch.from_numpy(same).to(dev), 1, 1, 256, d_g)
    m.sync()
    assert np.array_equal(d_g.cpu().numpy().ravel(), v)


def test_mask_area_dev_equals_numpy_slices_and_golden_gated(trained, golden_dir):
    import torch

    g, m, frames = trained
    dev = torch.device("cuda", 0)
    masks, areas, _ = m.segment(frames[:8])
    d_masks = torch.from_numpy(masks).to(dev)
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["gated"]
    for bi, b in enumerate(meta["boxes"]):       # reference-generated: features.py:244-245 on the reference's masks
        bx = torch.tensor([normalize_box(b, 256, 256)] * 8, dtype=torch.int32, device=dev)
        out = torch.full((8,), -7, dtype=torch.int32, device=dev)
        m.mask_area_dev(d_masks, 8, 256, 256, bx, out)
        m.sync()
        assert out.cpu().tolist() == [row[bi] for row in meta["areas_first8"]], b
    rs = np.random.RandomState(9)
    boxes = []
    for i in range(8):
        x1, y1 = rs.randint(-20, 250, 2)
        boxes.append((int(x1), int(y1), int(x1 + rs.randint(-5, 300)), int(y1 + rs.randint(-5, 300))))
    boxes[3] = None
    nb = np.array([normalize_box(b, 256, 256) for b in boxes], np.int32)
    out = torch.zeros(8, dtype=torch.int32, device=dev)
    m.mask_area_dev(d_masks, 8, 256, 256, torch.from_numpy(nb).to(dev), out)
    m.sync()
    # python slice semantics on the raw boxes (negative indices wrap, as mask[y1:y2, x1:x2] does in features.py:244-245)
    want = [0 if b is None else int((masks[i][b[1]:b[3], b[0]:b[2]] > 0).sum()) for i, b in enumerate(boxes)]
    assert out.cpu().tolist() == want
    # no boxes: the plain full-frame count
    out2 = torch.zeros(8, dtype=torch.int32, device=dev)
    m.mask_area_dev(d_masks, 8, 256, 256, None, out2)
    m.sync()
    assert out2.cpu().tolist() == areas.tolist()
    # non-square masks
    mk = (rs.rand(3, 40, 72) > 0.6).astype(np.uint8) * 255
    bx = np.array([[5, 3, 60, 33], [0, 0, 72, 40], [-1, -1, -1, -1]], np.int32)
    out3 = torch.zeros(3, dtype=torch.int32, device=dev)
    m.mask_area_dev(torch.from_numpy(mk).to(dev), 3, 40, 72, torch.from_numpy(bx).to(dev), out3)
    m.sync()
    assert out3.cpu().tolist() == [int((mk[0][3:33, 5:60] > 0).sum()), int((mk[1] > 0).sum()), 0]


def test_crop_entry_points_refuse_or_neutralise_out_of_frame_boxes(trained):
    """A ctypes caller may hand raw TemporalDetector boxes (not clamped) to the crop entry points: the host variant must
    refuse them (OG_EINVAL), the device variant must not index out of bounds (such a record counts as "no detection")."""
    import torch

    from openglottal_amd._lib import check, lib, ptr

    g, m, frames = trained
    fr = np.ascontiguousarray(frames[:4])
    boxes = np.array([[10, 10, 100, 120], [200, 200, 300, 260], [-5, 10, 50, 60], [0, 0, 256, 256]], np.int32)   # 1: outside; 2: "none"
    geom = np.array([[0, 23, 256, 209], [0, 0, 256, 256], [0, 0, 256, 256], [0, 0, 256, 256]], np.int32)
    out = np.empty_like(fr)
    rc = lib().og_unet_segment_crops_u8(m._h, ptr(fr), 4, 256, 256, ptr(boxes), ptr(geom), 256, 0.5, ptr(out))
    assert rc == -1 and "outside the frame" in lib().og_last_error().decode()
    bad_geom = geom.copy(); bad_geom[0] = [200, 0, 100, 256]     # top + content_h > size
    ok_boxes = boxes.copy(); ok_boxes[1] = [200, 200, 256, 256]
    rc = lib().og_unet_segment_crops_u8(m._h, ptr(fr), 4, 256, 256, ptr(ok_boxes), ptr(bad_geom), 256, 0.5, ptr(out))
    assert rc == -1 and "does not fit" in lib().og_last_error().decode()
    # device variant with the same bad records: runs, and the bad frames come back all-zero
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(v).to(dev) for k, v in dict(fr=fr, boxes=boxes, geom=bad_geom).items()}
    tiles = torch.zeros((4, 256, 256), dtype=torch.uint8, device=dev)
    tmask = torch.zeros_like(tiles)
    outd = torch.full((4, 256, 256), 7, dtype=torch.uint8, device=dev)
    check(lib().og_unet_segment_crops_u8_dev(m._h, ptr(d["fr"]), 4, 256, 256, ptr(d["boxes"]), ptr(d["geom"]), 256, 0.5,
                                             ptr(tiles), ptr(tmask), ptr(outd)), "crops_dev")
    m.sync()
    o = outd.cpu().numpy()
    assert not o[0].any() and not o[1].any() and not o[2].any()          # bad geom / outside the frame / no detection
    full, _, _ = m.segment(fr[3:4])
    assert np.array_equal(o[3], full[0])                                  # the valid record is untouched by its neighbours


def test_streaming_engine_equals_one_shot_staging(trained):
    """og_unet_segment_u8 on the streaming engine (pinned ring, chunked async H2D / D2H under the kernel chains) must give
    bit for bit what staging the whole batch at once gives: masks, areas, logits, with and without boxes, ragged batch."""
    g, m, frames = trained
    idx = np.arange(203) % 80
    fr = np.ascontiguousarray(frames[idx])
    rs = np.random.RandomState(5)
    boxes = np.array([normalize_box((int(x), int(y), int(x + 60), int(y + 90)), 256, 256) for x, y in rs.randint(0, 200, (203, 2))], np.int32)
    boxes[7] = -1
    try:
        for chunk in (16, 64, 7):
            m.set_chunk(chunk)
            m.set_option("stream", 0)
            mk0, ar0, lg0 = m.segment(fr, want_logits=True)
            _, arb0, _ = m.segment(fr, boxes=boxes, want_mask=False)
            m.set_option("stream", 1)
            mk1, ar1, lg1 = m.segment(fr, want_logits=True)
            _, arb1, _ = m.segment(fr, boxes=boxes, want_mask=False)
            assert np.array_equal(mk0, mk1) and np.array_equal(ar0, ar1) and np.array_equal(lg0, lg1), chunk
            assert np.array_equal(arb0, arb1) and arb1[7] == 0, chunk
            assert np.array_equal(ar1.astype(np.int64), g["areas"][idx])       # and both equal the reference's integers
            mk2, ar2 = m.segment_stream(fr, want_mask=True)
            assert np.array_equal(mk2, mk1) and np.array_equal(ar2, ar1)
    finally:
        m.set_chunk(32)
        m.set_option("stream", 1)


def test_streaming_bgr_frames_pinned_and_pageable(trained):
    """BGR frames go up once and `k_bgr2gray` feeds the chain (features.py:235 on the device): equals host BGR->gray +
    gray segmentation; a pinned torch tensor (DMA straight from it) equals pageable numpy memory."""
    import torch

    g, m, frames = trained
    rs = np.random.RandomState(3)
    bgr = np.clip(frames[:70, ..., None].astype(np.int32) + rs.randint(-30, 30, (70, 256, 256, 3)), 0, 255).astype(np.uint8)
    gray = bgr_to_gray(bgr)
    m.set_chunk(16)
    try:
        mk_ref, ar_ref, _ = m.segment(gray)
        mk, ar = m.segment_stream(bgr, want_mask=True)
        assert np.array_equal(mk, mk_ref) and np.array_equal(ar, ar_ref)
        pinned = torch.from_numpy(bgr).pin_memory()
        assert pinned.is_pinned()
        mk_p, ar_p = m.segment_stream(pinned, want_mask=True)
        assert np.array_equal(mk_p, mk_ref) and np.array_equal(ar_p, ar_ref)
        # the callers: extract_features_unet / area_waveform on a BGR video use the device conversion
        from openglottal_amd.features import area_waveform
        assert np.array_equal(area_waveform(bgr, None, m), ar_ref.astype(np.float64))
        assert np.array_equal(area_waveform(list(bgr), None, m), ar_ref.astype(np.float64))
    finally:
        m.set_chunk(32)


def test_streaming_keeps_device_memory_bounded_for_a_long_video(trained, tmp_path):
    """A 6 000-frame video (memory-mapped .npy, 393 MB of frames) through extract_features_unet: device memory in use
    grows by the ring and one arena per lane, not by the video; the waveform repeats the golden integers."""
    import torch

    from openglottal_amd.features import area_waveform

    g, m, frames = trained
    idx = np.arange(6000) % 80
    path = str(tmp_path / "long.npy")
    np.save(path, frames[idx])
    m.set_chunk(32)
    area_waveform(frames[:64], None, m)            # ring + arenas exist now
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    wave = area_waveform(path, None, m)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert np.array_equal(wave.astype(np.int64), g["areas"][idx])
    assert free0 - free1 < 64 << 20, (free0, free1)   # nothing proportional to 6 000 frames (393 MB) was allocated


def test_mixed_frame_sizes_reuse_the_arena_and_give_the_same_results(trained):
    """A stream of mixed frame sizes re-plans the activation arena in place (capacity is kept in bytes; og_unet_reserve
    allocates it ahead of time): no allocation after the reserve, results equal to those of a fresh handle per size."""
    import torch

    g, m, frames = trained
    m.set_chunk(16)
    m.reserve(16, 256, 256)
    shapes = [(256, 256), (64, 128), (256, 256), (128, 128), (48, 80), (256, 256), (64, 128)]
    rs = np.random.RandomState(8)
    inputs = [frames[:20] if s == (256, 256) else rs.randint(0, 256, (20,) + s, dtype=np.uint8) for s in shapes]
    m.segment(inputs[0]); m.segment(inputs[1])     # ring buffers of the streaming engine exist for both footprints now
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    got = [m.segment(x, want_logits=True) for x in inputs]
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 48 << 20, (free0, free1)     # only ring slots for new (smaller) footprints, never an arena
    for s, x, (mk, ar, lg) in zip(shapes, inputs, got):
        fresh = og.UNet(1, 1, m.features)
        fresh.load_state_dict(m.state_dict())
        fresh.to("cuda:0").eval()
        fresh.set_chunk(16)
        mk2, ar2, lg2 = fresh.segment(x, want_logits=True)
        assert np.array_equal(mk, mk2) and np.array_equal(ar, ar2) and np.array_equal(lg, lg2), s
    assert np.array_equal(got[0][1].astype(np.int64), g["areas"][:20])
    m.set_chunk(32)


def test_frame_list_entry_equals_the_stacked_video(trained):
    """`og_unet_stream_frames_u8` (a LIST of separately allocated frames, as the reference's `frames_bgr`): masks, areas and
    box-gated areas equal those of the same video as one array -- gray and BGR, non-contiguous / non-u8 items converted,
    ragged over the micro-batch, one frame, empty list; mismatched shapes and null entries refused."""
    import ctypes as C

    from openglottal_amd._lib import lib

    g, m, frames = trained
    n = 37
    gray = frames[:n]
    rs = np.random.RandomState(3)
    bgr = np.clip(gray[..., None].astype(np.int32) + rs.randint(-4, 5, (n, 256, 256, 3)), 0, 255).astype(np.uint8)
    boxes = np.array([normalize_box((30 + i, 40, 200 - i, 220), 256, 256) for i in range(n)], np.int32)
    boxes[5] = -1
    m.set_chunk(16)
    try:
        for video in (gray, bgr):
            mk_a, ar_a = m.segment_stream(video, boxes=boxes, want_mask=True)
            items = [np.array(f) for f in video]                                     # separately allocated
            items[3] = np.asfortranarray(items[3]) if items[3].ndim == 2 else items[3][:, ::-1][:, ::-1]   # a non-contiguous view
            items[4] = items[4].astype(np.int32)                                     # a non-u8 item
            mk_l, ar_l = m.segment_stream(items, boxes=boxes, want_mask=True)
            assert np.array_equal(ar_l, ar_a) and np.array_equal(mk_l, mk_a)
            mk_1, ar_1 = m.segment_stream(items[:1], want_mask=True)
            assert np.array_equal(mk_1[0], m.segment_stream(video[:1], want_mask=True)[0][0])
        mk_e, ar_e = m.segment_stream([], want_mask=True)
        assert ar_e.shape == (0,)
        with pytest.raises(og.OpenGlottalHipError):
            m.segment_stream([gray[0], gray[1][:128]])
        ptrs = (C.c_void_p * 2)(gray[0].ctypes.data, None)
        area = np.zeros(2, np.int32)
        assert lib().og_unet_stream_frames_u8(m._h, ptrs, 2, 256, 256, 1, C.c_float(0.5), None, None, area.ctypes.data_as(C.c_void_p)) != 0
        assert b"frame_ptrs[1]" in lib().og_last_error()
    finally:
        m.set_chunk(32)
