"""CPU tests: host-side mirror of the reference logic vs golden vectors captured from the reference."""
import ctypes
import json
import math
import os
import re

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import _lib, synth
from openglottal_amd.features import _kinematic_features
from openglottal_amd.utils import bgr_to_gray, frame_metrics, normalize_box


def test_library_loads_and_exports_every_declared_symbol():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "openglottal_hip.h")).read()
    declared = set(re.findall(r"\b(og_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("og_unet")
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert b"gfx950" in lib.og_version()
    assert b"store_nop=1" in lib.og_version()     # a production build: the loader refuses anything else (next test)


def test_loader_refuses_an_audit_build(monkeypatch):
    """`_lib.lib()` must not hand out a library whose `og_version()` lacks the store-nop marker (the -DOG_STORE_NOP=0 audit build of
    tools/epilogue_fence_audit.sh computes wrong lanes); only OPENGLOTTAL_HIP_ALLOW_AUDIT_BUILD=1 lets it through."""
    import ctypes as C

    from openglottal_amd import _lib

    real = _lib.lib()

    class FakeFn:
        def __init__(self, fn):
            self._fn = fn
        def __call__(self, *a):
            return b"openglottal_hip 0.3 (gfx950; store_nop=0)"

    class Fake:
        def __getattr__(self, name):
            return FakeFn(None) if name == "og_version" else getattr(real, name)

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(C, "CDLL", lambda path: Fake())
    monkeypatch.delenv("OPENGLOTTAL_HIP_ALLOW_AUDIT_BUILD", raising=False)
    with pytest.raises(_lib.OpenGlottalHipError, match="not a production build"):
        _lib.lib()
    monkeypatch.setenv("OPENGLOTTAL_HIP_ALLOW_AUDIT_BUILD", "1")
    assert _lib.lib() is not None
    monkeypatch.setattr(_lib, "_lib", real)


def test_no_cpu_fallback():
    m = og.UNet(1, 1, (4, 8))
    with pytest.raises(og.OpenGlottalHipError):
        m.to("cpu")
    with pytest.raises(og.OpenGlottalHipError):
        m(np.zeros((1, 1, 16, 16), np.float32))  # no weights, no device
    with pytest.raises(og.OpenGlottalHipError):
        og.UNet(3, 1)


def test_product_does_not_import_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, fs in os.walk(os.path.join(root, "openglottal_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_kinematic_features_match_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "kinematic.json")))
    assert set(cases) >= {"periodic", "slow_f0_none", "silent", "short"}
    for name, c in cases.items():
        got = _kinematic_features(list(c["wave"]))
        if c["out"] is None:
            assert got is None, name
            continue
        assert np.array_equal(got["_area"], np.array(c["wave"]))
        for k, v in c["out"].items():
            if v is None:
                assert got[k] is None, (name, k)
            else:
                assert got[k] is not None and math.isclose(float(got[k]), v, rel_tol=1e-12, abs_tol=1e-12), (name, k)


def test_temporal_detector_traces_match_reference(golden_dir):
    traces = json.load(open(os.path.join(golden_dir, "detector_traces.json")))
    assert len(traces) >= 9
    for name, t in traces.items():
        script = t["script"]
        calls = {"i": 0}

        def backend(frame, conf):
            d = [x for x in script[calls["i"]] if x[4] >= conf]
            calls["i"] += 1
            return (np.array([x[:4] for x in d], np.float32).reshape(-1, 4), np.array([x[4] for x in d], np.float32))

        det = og.TemporalDetector(backend, **t["kw"])
        frame = np.zeros(t["shape"], np.uint8)
        outs = []
        for _ in script:
            b = det.detect(frame)
            outs.append(None if b is None else [int(v) for v in b])
        assert outs == t["out"], name
        # the two-halves form (submit ... result: an extension next to the reference's detect) runs the same state machine
        calls["i"] = 0
        det2 = og.TemporalDetector(backend, **t["kw"])
        outs2 = []
        for _ in script:
            det2.submit(frame)
            b = det2.result()
            outs2.append(None if b is None else [int(v) for v in b])
        assert outs2 == t["out"], name


def test_dice_iou_frame_metrics(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_trained_small.npz"))
    frames, gt = synth.glottis_frames(4, 20, seed=99)
    for i in (0, 7, 33, 79):
        m = np.unpackbits(g["masks_packed"][i])[: 256 * 256].reshape(256, 256) * 255
        assert math.isclose(og.dice(m, gt[i]), float(g["dice_vs_gt"][i]), rel_tol=0, abs_tol=1e-7)
        assert math.isclose(og.iou(m, gt[i]), float(g["iou_vs_gt"][i]), rel_tol=0, abs_tol=1e-7)
        d, j = frame_metrics(m, gt[i])
        assert abs(d - og.dice(m, gt[i])) < 1e-6 and abs(j - og.iou(m, gt[i])) < 1e-6
    z = np.zeros((8, 8), np.uint8)
    assert og.dice(z, z) == 1.0 and og.iou(z, z) == 1.0 and frame_metrics(z, z) == (1.0, 1.0)


def test_bgr2gray_exact_on_gray_and_weights():
    v = np.arange(256, dtype=np.uint8)
    f = np.stack([v, v, v], -1)[None]
    assert np.array_equal(bgr_to_gray(f)[0], v)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)  # B, G, R
    assert bgr_to_gray(px).tolist() == [[29, 150, 76]]


def test_normalize_box_python_slice_semantics():
    m = np.arange(30 * 40).reshape(30, 40) % 7 == 0
    for box in [(3, 4, 20, 25), (-5, 2, 10, 12), (0, 0, 100, 100), (10, 10, 10, 20), (30, 5, 20, 9), (-50, -50, -45, 3)]:
        x1, y1, x2, y2 = box
        want = int(m[y1:y2, x1:x2].sum())
        a, b, c, d = normalize_box(box, 40, 30)
        assert int(m[b:d, a:c].sum()) == want, box
    assert normalize_box(None, 40, 30) == (-1, -1, -1, -1)


def test_load_frames_from_npy_and_png_directory(tmp_path):
    from PIL import Image
    from openglottal_amd.features import load_frames_bgr
    rs = np.random.RandomState(2)
    vid = rs.randint(0, 256, (3, 16, 24, 3), dtype=np.uint8)
    np.save(tmp_path / "v.npy", vid)
    got = load_frames_bgr(str(tmp_path / "v.npy"))
    assert len(got) == 3 and np.array_equal(got[1], vid[1])
    d = tmp_path / "seq"
    d.mkdir()
    for i in range(3):
        Image.fromarray(vid[i][..., ::-1].copy()).save(d / f"f_{i:03d}.png")   # files hold RGB
    Image.fromarray(vid[0][..., 0]).save(d / "g_999.png")                        # a grayscale frame
    seq = load_frames_bgr(str(d))
    assert len(seq) == 4 and all(np.array_equal(seq[i], vid[i]) for i in range(3)) and seq[3].ndim == 2
    assert load_frames_bgr(str(tmp_path / "missing.avi")) == []


def test_host_bgr_to_gray_in_the_c_library_equals_the_numpy_restatement():
    """`og_bgr2gray_host` (the per-frame loop's cv2.cvtColor, features.py:235) against the numpy restatement of OpenCV's u8 path:
    every value of one channel against extremes of the others, random frames, odd sizes, R = G = B exact, non-contiguous input."""
    from openglottal_amd.utils import bgr_to_gray, bgr_to_gray_numpy

    rs = np.random.RandomState(3)
    for shape in [(256, 256, 3), (1, 1, 3), (37, 5, 3), (4, 256, 256, 3)]:
        f = rs.randint(0, 256, shape).astype(np.uint8)
        assert np.array_equal(bgr_to_gray(f), bgr_to_gray_numpy(f))
    v = np.arange(256, dtype=np.uint8)
    for c in range(3):
        for lo in (0, 255):
            f = np.full((256, 1, 3), lo, np.uint8)
            f[:, 0, c] = v
            assert np.array_equal(bgr_to_gray(f), bgr_to_gray_numpy(f))
    g = np.repeat(v[:, None, None], 3, axis=2)
    assert np.array_equal(bgr_to_gray(g)[:, 0], v)               # (v, v, v) -> v
    nc = rs.randint(0, 256, (64, 64, 6)).astype(np.uint8)[..., ::2]     # a strided view takes the numpy path: same integers
    assert not nc.flags.c_contiguous and np.array_equal(bgr_to_gray(nc), bgr_to_gray_numpy(nc))
