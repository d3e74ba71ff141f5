"""`extract_features_unet` and `_kinematic_features` (`openglottal/features.py`)."""

from __future__ import annotations

import os

import numpy as np

from ._lib import OpenGlottalHipError
from .utils import NET_SIZE, bgr_to_gray, normalize_box, unet_segment_frame


def _kinematic_features(area_wave) -> dict | None:
    """Seven scalars of a glottal-area waveform (features.py:38-68).

    ``None`` for a silent (all-zero) waveform; ``f0`` is ``None`` when the
    spectral peak sits in the first non-DC bin; periodicity is the largest
    normalised autocorrelation over lags 1..49.
    """
    a = np.array(area_wave)
    if a.max() == 0:
        return None
    mu, sd = a.mean(), a.std()
    centred = a - mu
    spectrum = np.abs(np.fft.rfft(centred))
    k = int(np.argmax(spectrum[1:]) + 1)
    f0 = None if k == 1 else float(np.fft.rfftfreq(len(a))[k])
    full = np.correlate(centred, centred, mode="full")
    ac = full[len(full) // 2:]
    ac /= ac[0] + 1e-8
    return {
        "area_mean": mu,
        "area_std": sd,
        "area_range": a.max() - a.min(),
        "open_quotient": float(np.mean(a > mu * 0.1)),
        "f0": f0,
        "periodicity": float(ac[1:min(50, len(ac))].max()),
        "cv": sd / (mu + 1e-8),
        "_area": a,
    }


def load_frames_bgr(source) -> list[np.ndarray]:
    """Frames of a video as BGR u8 arrays (utils.py:43-54).

    ``source`` may be an in-memory array/list of frames, a ``.npy``/``.npz`` file
    (``[N,H,W,3]`` BGR or ``[N,H,W]`` gray), a directory of PNG/JPEG frames (Pillow), or a
    video file when OpenCV is importable (AVI/MJPG decode is host I/O and stays on cv2).
    """
    if isinstance(source, np.ndarray):
        return list(source)
    if isinstance(source, (list, tuple)):
        return list(source)
    p = str(source)
    if p.endswith(".npy"):
        return list(np.load(p))
    if p.endswith(".npz"):
        z = np.load(p)
        return list(z[z.files[0]])
    if not os.path.exists(p):
        return []
    if os.path.isdir(p):  # image-sequence directory (GIRAFE ships frames as PNGs): decode with Pillow, no OpenCV needed
        from PIL import Image

        frames = []
        for name in sorted(os.listdir(p)):
            if name.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")):
                im = np.asarray(Image.open(os.path.join(p, name)))
                frames.append(np.ascontiguousarray(im[..., 2::-1]) if im.ndim == 3 else im)  # RGB(A) -> BGR, gray stays 2-D
        return frames
    try:
        import cv2  # noqa: F401
    except ImportError as e:
        raise OpenGlottalHipError(f"decoding {p} needs OpenCV; pass frames as .npy/.npz or an array instead") from e
    import cv2

    cap = cv2.VideoCapture(p)
    frames = []
    while True:
        ok, frm = cap.read()
        if not ok:
            break
        frames.append(frm)
    cap.release()
    return frames


BLOCK = 1024  # frames handed to the device engine per call (host side only: the engine itself streams micro-batches)
GATED_BLOCK = 128   # gated pipeline: the detector pass of block k + 1 runs (worker thread, own handle and stream) under the U-Net pass of
                    # block k; rounded up to a whole number of the model's micro-batches so that no block ends in a ragged launch


def gated_block(model) -> int:
    c = max(1, int(getattr(model, "chunk", 32)))
    return -(-GATED_BLOCK // c) * c


def iter_frame_blocks(source, block: int = BLOCK):
    """Blocks ``[n<=block,H,W(,3)]`` of a video without loading it whole where the container allows it: an ``.npy`` file is
    memory-mapped, an in-memory array is sliced (no copy), anything else goes through ``load_frames_bgr``."""
    if isinstance(source, str) and source.endswith(".npy") and os.path.exists(source):
        source = np.load(source, mmap_mode="r")
    if isinstance(source, np.ndarray) and source.ndim >= 3:
        for lo in range(0, len(source), block):
            yield source[lo:lo + block]
        return
    frames = source if isinstance(source, (list, tuple)) else load_frames_bgr(source)
    for lo in range(0, len(frames), block):
        yield frames[lo:lo + block]


def _detect_block(frames, detector) -> np.ndarray:
    """Boxes ``[n,4]`` int32 for one block; the temporal state machine carries over from the previous block."""
    n = len(frames)
    boxes = np.empty((n, 4), np.int32)
    batch = getattr(detector.model, "detect_frames", None)
    shapes = {f.shape for f in frames}
    if batch is not None and len(shapes) == 1:
        # native backend, frames of one size: the YOLO network is per-frame independent, so run it batched on the device
        # (letterboxed to the model's imgsz and scaled back exactly as the per-frame call does); only the O(1)/frame
        # temporal state machine (detector.py:61-96) is sequential
        best = batch(np.asarray(frames) if isinstance(frames, np.ndarray) else np.stack(frames), detector.conf)
        H, W = frames[0].shape[:2]
        for i in range(n):
            b = detector.update(best[i:i + 1, :4], best[i:i + 1, 4], W, H) if best[i, 4] >= 0 else detector.update(None, None, W, H)
            boxes[i] = normalize_box(b, W, H)
    else:
        for i, f in enumerate(frames):
            b = detector.detect(f)
            boxes[i] = normalize_box(b, f.shape[1], f.shape[0])
    return boxes


def _blocks_with_boxes(frames, detector, model=None):
    """``(block, boxes)`` pairs of the gated pipeline.  With the NATIVE detector backend (``detect_frames``: one batched device
    pass per block, the GIL released inside the C-ABI) the detector network and the O(1)-per-frame temporal state machine of block
    k + 1 run on ONE worker thread (so blocks are detected in order: the state machine is sequential, detector.py:61-96) while the
    caller segments block k.  Any other backend calls ``detector.detect`` per frame under the GIL -- a worker would overlap nothing --
    and keeps the plain sequential pass over BLOCK-sized blocks.  Nothing about the results changes either way: same boxes, in the
    same order, as a detect-everything-first pass.  (If the consumer abandons the generator, leaving the ``with`` block waits for
    the detection in flight -- at most one block.)"""
    from concurrent.futures import ThreadPoolExecutor

    if getattr(getattr(detector, "model", None), "detect_frames", None) is None:
        for blk in iter_frame_blocks(frames, BLOCK):
            if len(blk):
                yield blk, _detect_block(blk, detector)
        return
    it = (b for b in iter_frame_blocks(frames, gated_block(model)) if len(b))
    with ThreadPoolExecutor(1) as pool:
        nxt = next(it, None)
        fut = pool.submit(_detect_block, nxt, detector) if nxt is not None else None
        while nxt is not None:
            blk, boxes = nxt, fut.result()
            nxt = next(it, None)
            fut = pool.submit(_detect_block, nxt, detector) if nxt is not None else None
            yield blk, boxes


def area_waveform(frames, detector, model, device=None, threshold: float = 0.5) -> np.ndarray:
    """The frame loop of features.py:234-245 as batched device passes over blocks of the video.

    U-Net-only (``detector is None``): area = #(mask>0) per frame.  Gated: the sequential TemporalDetector pass produces
    one box per frame first (the U-Net does not depend on it), then the fused kernel counts inside the boxes.  Frames at
    network size go to the streaming engine as they are — BGR included: `cv2.cvtColor(BGR2GRAY)` (features.py:235) runs on
    the device — so no per-frame host work is left and device memory does not grow with the video.
    """
    if device is not None and getattr(model, "_device", None) is None:
        model.to(device)
    if detector is not None:
        detector.reset()
    out = []
    done = 0
    pairs = _blocks_with_boxes(frames, detector, model) if detector is not None else ((b, None) for b in iter_frame_blocks(frames))
    for blk, boxes in pairs:
        n = len(blk)
        if n == 0:
            continue
        shapes = {f.shape for f in blk}
        if len(shapes) == 1 and next(iter(shapes))[:2] == (NET_SIZE, NET_SIZE):
            # an array goes up in place; a list of frames (features.py:226's `frames_bgr`) is gathered by the engine itself,
            # frame by frame into its pinned ring -- no np.stack of the block
            _, area = model.segment_stream(blk if isinstance(blk, np.ndarray) else list(blk), threshold=threshold, boxes=boxes)
            out.append(area.astype(np.float64))
        else:   # mixed / non-256 frames: per-frame path incl. host resizes (utils.py:234,239-240)
            a = np.zeros(n, np.float64)
            for i, f in enumerate(blk):
                m = unet_segment_frame(bgr_to_gray(f), model, device, threshold)
                if boxes is None:
                    a[i] = float(np.sum(m > 0))
                elif boxes[i][0] >= 0:
                    x1, y1, x2, y2 = boxes[i]
                    a[i] = float(np.sum(m[y1:y2, x1:x2] > 0))
            out.append(a)
        done += n
    return np.concatenate(out) if out else np.zeros(0, np.float64)


def extract_features_unet(avi_path, detector, model, device=None) -> dict | None:
    """Drop-in for `openglottal/features.py:202-247` (U-Net-only when ``detector is None``)."""
    wave = area_waveform(avi_path, detector, model, device)
    if len(wave) == 0:
        return None   # features.py:227-228
    return _kinematic_features([float(v) for v in wave])
