"""Per-frame helpers with the reference's names and semantics (`openglottal/utils.py`)."""

from __future__ import annotations

import numpy as np

from ._lib import OpenGlottalHipError, check, lib, ptr

NET_SIZE = 256  # unet_segment_frame resizes every frame to 256x256 (utils.py:234)


def bgr_to_gray(frame_bgr: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(frame, COLOR_BGR2GRAY)`` (features.py:235) for u8 ``[...,3]``.

    OpenCV is absent from the build image, so this restates its published u8 path:
    15-bit fixed point, R 9798 / G 19235 / B 3735, ``(x + 2**14) >> 15``.
    Parity unpinned; exact for R=G=B inputs by construction.
    The device twin is ``k_bgr2gray`` (og_kernels.hpp).
    """
    f = np.asarray(frame_bgr)
    if f.ndim >= 3 and f.shape[-1] == 3:
        if f.dtype == np.uint8 and f.flags.c_contiguous:
            # the per-frame loop calls this once per frame (features.py:235): one pass in the C library (og_bgr2gray_host, host only)
            # instead of eight numpy passes -- the same integers (bgr_to_gray_numpy, tests/test_host_logic.py)
            out = np.empty(f.shape[:-1], np.uint8)
            check(lib().og_bgr2gray_host(ptr(f), int(out.size), ptr(out)), "og_bgr2gray_host")
            return out
        return bgr_to_gray_numpy(f)
    return f.astype(np.uint8)


def bgr_to_gray_numpy(f: np.ndarray) -> np.ndarray:
    """The same arithmetic in numpy (any integer dtype / memory layout; the check of the C loop)."""
    b, g, r = (np.asarray(f)[..., i].astype(np.int32) for i in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def unet_segment_frame(frame_gray: np.ndarray, model, device=None, threshold: float = 0.5) -> np.ndarray:
    """Drop-in for `openglottal/utils.py:218-241`: ``(H,W)`` u8 → u8 mask in {0,255}.

    256×256 frames (every BASELINE config) run fully on the device: u8 → /255 →
    U-Net → sigmoid → ``> threshold`` in one kernel chain.  Other sizes go through
    host bilinear resizes like the reference's two ``cv2.resize`` calls
    (utils.py:234,239-240; parity unpinned, see ``resize_linear``).
    """
    g = np.asarray(frame_gray)
    if g.ndim != 2:
        raise OpenGlottalHipError(f"frame_gray must be (H, W), got {g.shape}")
    if device is not None and getattr(model, "_device", None) is None:
        model.to(device)
    H, W = g.shape
    if (H, W) == (NET_SIZE, NET_SIZE):
        mask, _, _ = model.segment(g[None], threshold=threshold, want_mask=True, want_area=False)
        return mask[0]
    from .geometry import resize_linear

    inp = resize_linear(g.astype(np.uint8), NET_SIZE, NET_SIZE)
    _, _, logits = model.segment(inp[None], threshold=threshold, want_mask=False, want_logits=True)
    prob = (np.float32(1) / (np.float32(1) + np.exp(-logits[0]))).astype(np.float32)
    prob = resize_linear(prob, W, H)
    return (prob > threshold).astype(np.uint8) * 255


# ── Segmentation metrics (utils.py:191-206, scripts/eval_girafe.py:113-124) ──


def _binary(a: np.ndarray) -> np.ndarray:
    return (np.asarray(a) > 0).astype(np.float32)


def dice(pred: np.ndarray, gt: np.ndarray) -> float:
    p, g = _binary(pred), _binary(gt)
    total = p.sum() + g.sum()
    if not total > 0:
        return 1.0
    return float(2 * (p * g).sum() / total)


def iou(pred: np.ndarray, gt: np.ndarray) -> float:
    p, g = _binary(pred), _binary(gt)
    inter = (p * g).sum()
    union = p.sum() + g.sum() - inter
    if not union > 0:
        return 1.0
    return float(inter / union)


def frame_metrics(pred: np.ndarray, gt: np.ndarray) -> tuple[float, float]:
    """(Dice, IoU) as the eval harness computes them: from tp/fp/fn, 1.0 when both empty."""
    p, g = _binary(pred).ravel(), _binary(gt).ravel()
    tp = (p * g).sum()
    fp = (p * (1 - g)).sum()
    fn = ((1 - p) * g).sum()
    d_den, i_den = 2 * tp + fp + fn, tp + fp + fn
    return (float(2 * tp / d_den) if d_den > 0 else 1.0, float(tp / i_den) if i_den > 0 else 1.0)


def normalize_box(box, W: int, H: int):
    """Python-slice semantics of ``mask[y1:y2, x1:x2]`` (features.py:244-245) as a
    non-negative half-open box, or ``(-1,-1,-1,-1)`` for ``None`` (area 0, :241-242)."""
    if box is None:
        return (-1, -1, -1, -1)
    x1, y1, x2, y2 = (int(v) for v in box)
    xs, xe, _ = slice(x1, x2).indices(W)
    ys, ye, _ = slice(y1, y2).indices(H)
    if xe <= xs or ye <= ys:
        return (0, 0, 0, 0)
    return (xs, ys, xe, ye)
