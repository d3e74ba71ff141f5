"""Letterbox / crop / resize geometry either side of the U-Net (`openglottal/utils.py:57-186`).

The reference does these with OpenCV (`cv2.resize`, `cv2.copyMakeBorder`).  OpenCV is not
installed in the build image and the reference holds no fixture for it, so the two
interpolators below restate OpenCV's *published* algorithms (modules/imgproc/src/resize.cpp)
and are **parity unpinned** against a real cv2:

* INTER_NEAREST: ``src = min(floor(dst * src_len / dst_len), src_len - 1)``.
* INTER_LINEAR : half-pixel centres ``f = (dst + 0.5) * src_len/dst_len - 0.5``, edge clamp;
  for ``uint8`` the 11-bit fixed-point coefficients and the
  ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`` vertical pass; for ``float32`` plain
  float arithmetic, horizontal pass first.

All of it is host-side index arithmetic on 256×256-class arrays; none of it is on the
BASELINE configs' hot path, where every U-Net input is already 256×256.
"""

from __future__ import annotations

import numpy as np

INTER_NEAREST, INTER_LINEAR = 0, 1
_COEF_BITS = 11
_COEF_ONE = 1 << _COEF_BITS


def _nearest_index(dst_len: int, src_len: int) -> np.ndarray:
    scale = src_len / dst_len
    return np.minimum(np.floor(np.arange(dst_len) * scale).astype(np.int64), src_len - 1)


def resize_nearest(img: np.ndarray, w: int, h: int) -> np.ndarray:
    ys, xs = _nearest_index(h, img.shape[0]), _nearest_index(w, img.shape[1])
    return np.ascontiguousarray(img[ys][:, xs])


def _linear_taps(dst_len: int, src_len: int):
    scale = src_len / dst_len
    f = (np.arange(dst_len) + 0.5) * scale - 0.5
    i0 = np.floor(f).astype(np.int64)
    frac = (f - i0).astype(np.float32)
    lo = i0 < 0
    i0[lo], frac[lo] = 0, 0.0
    hi = i0 >= src_len - 1
    i0[hi], frac[hi] = src_len - 1, 0.0
    i1 = np.minimum(i0 + 1, src_len - 1)
    return i0, i1, frac


def resize_linear(img: np.ndarray, w: int, h: int) -> np.ndarray:
    """``cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)`` for uint8 or float32, 2-D or HxWxC."""
    sh, sw = img.shape[:2]
    if (sh, sw) == (h, w):
        return img.copy()
    x0, x1, fx = _linear_taps(w, sw)
    y0, y1, fy = _linear_taps(h, sh)
    tail = (None,) * (img.ndim - 2)
    if img.dtype == np.uint8:
        ax1 = np.rint(fx * _COEF_ONE).astype(np.int32)
        ax0 = (_COEF_ONE - ax1).astype(np.int32)
        ay1 = np.rint(fy * _COEF_ONE).astype(np.int32)
        ay0 = (_COEF_ONE - ay1).astype(np.int32)
        s = img.astype(np.int32)
        rows = s[:, x0] * ax0[(None, slice(None)) + tail] + s[:, x1] * ax1[(None, slice(None)) + tail]  # [sh, w, ...] scale 2^11
        r0, r1 = rows[y0], rows[y1]
        b0 = ay0[(slice(None), None) + tail]
        b1 = ay1[(slice(None), None) + tail]
        out = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
        return np.clip(out, 0, 255).astype(np.uint8)
    s = img.astype(np.float32)
    fxb = fx[(None, slice(None)) + tail]
    rows = s[:, x0] * (np.float32(1) - fxb) + s[:, x1] * fxb
    fyb = fy[(slice(None), None) + tail]
    return (rows[y0] * (np.float32(1) - fyb) + rows[y1] * fyb).astype(np.float32)


def resize(img: np.ndarray, w: int, h: int, interp: int) -> np.ndarray:
    return resize_nearest(img, w, h) if interp == INTER_NEAREST else resize_linear(img, w, h)


def _pad(img: np.ndarray, top: int, bottom: int, left: int, right: int, value: int) -> np.ndarray:
    pads = ((top, bottom), (left, right)) + ((0, 0),) * (img.ndim - 2)
    return np.pad(img, pads, mode="constant", constant_values=value)


def letterbox_geometry(h: int, w: int, size: int = 256) -> tuple[int, int, int, int]:
    """``(pad_top, pad_left, content_h, content_w)`` of a letterbox to ``size`` (utils.py:108-123, eval_bagls.py:53-63):
    Python ``round()`` (half to even) of the scaled sides, the odd padding pixel goes to the bottom / right."""
    scale = size / max(h, w)
    nh, nw = int(round(h * scale)), int(round(w * scale))
    return (size - nh) // 2, (size - nw) // 2, nh, nw


def pack_frames(frames) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Frames of mixed sizes (all 2-D or all HxWx3, u8) back to back in one buffer for the device front end:
    ``(packed u8 [total], offsets int64 [B], shapes int32 [B,2])``."""
    shapes = np.array([f.shape[:2] for f in frames], np.int32).reshape(-1, 2)
    sizes = np.array([f.size for f in frames], np.int64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64) if len(frames) else np.zeros(0, np.int64)
    packed = np.empty(int(sizes.sum()), np.uint8)
    for f, o, n in zip(frames, offsets, sizes):
        packed[o:o + n] = np.ascontiguousarray(f, dtype=np.uint8).ravel()
    return packed, offsets, shapes


def letterbox_with_info(img: np.ndarray, size: int = 256, value: int = 0):
    """Longest side → ``size`` (aspect preserved), symmetric constant pad (utils.py:97-134).

    2-D inputs (gray frames, masks) are resampled NEAREST, 3-D (BGR) LINEAR, as the reference
    does (utils.py:117).  Returns ``(out, pad_top, pad_left, content_h, content_w)``.
    """
    h, w = img.shape[:2]
    scale = size / max(h, w)
    nh, nw = int(round(h * scale)), int(round(w * scale))
    resized = resize(img, nw, nh, INTER_LINEAR if img.ndim == 3 else INTER_NEAREST)
    ph, pw = size - nh, size - nw
    top, left = ph // 2, pw // 2
    return _pad(resized, top, ph - top, left, pw - left, value), top, left, nh, nw


def letterbox(img: np.ndarray, size: int = 256, value: int = 0) -> np.ndarray:
    return letterbox_with_info(img, size, value)[0]


def letterbox_apply_geometry(img, size, pad_top, pad_left, content_h, content_w, value=0, interp=None):
    """Re-apply a previous letterbox geometry, e.g. to the matching mask (utils.py:137-163)."""
    if interp is None:
        interp = INTER_NEAREST if img.ndim == 2 else INTER_LINEAR
    resized = resize(img, content_w, content_h, interp)
    return _pad(resized, pad_top, size - pad_top - content_h, pad_left, size - pad_left - content_w, value)


def unletterbox(boxed, pad_top, pad_left, content_h, content_w, target_h, target_w, interp=INTER_NEAREST):
    """Cut the content region back out and resample it to the crop's size (utils.py:166-186)."""
    crop = boxed[pad_top:pad_top + content_h, pad_left:pad_left + content_w]
    if (content_h, content_w) == (target_h, target_w):
        return crop
    return resize(crop, target_w, target_h, interp)
