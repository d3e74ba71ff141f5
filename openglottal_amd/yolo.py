"""Native YOLOv8 detector backend behind ``TemporalDetector`` (`openglottal/models/detector.py:31,58`).

``YoloV8Detector`` is what ``ultralytics.YOLO(path)`` is to the reference: constructed from a
weights file, called as ``model(frame_bgr, conf)``.  Differences a user must know:

* weights are a flat ``.npz`` / dict of ultralytics' own state_dict keys (export once, where
  ultralytics is installed: ``np.savez(out, **{k: v.cpu().numpy() for k, v in
  YOLO(pt).model.state_dict().items()})``) — an ultralytics ``.pt`` is a pickle of its class graph
  and cannot be read without the package (SURVEY §7 hard parts);
* the network arithmetic is restated from ultralytics' published sources: PARITY UNPINNED.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import OpenGlottalHipError, check, lib, ptr
from .unet import _device_index

STRIDE = 32


def letterbox_bgr(frame: np.ndarray, imgsz: int = 256, stride: int = STRIDE):
    """ultralytics LetterBox(auto=True): fit the long side to ``imgsz``, pad (114) to a stride multiple.
    Returns ``(img, gain, pad_x, pad_y)``.  Identity for frames whose sides are multiples of 32 and <= imgsz."""
    from .geometry import resize_linear

    h, w = frame.shape[:2]
    r = min(imgsz / h, imgsz / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = (imgsz - nw) % stride, (imgsz - nh) % stride
    img = frame if (nw, nh) == (w, h) else resize_linear(frame, nw, nh)
    top, left = int(round(dh / 2 - 0.1)), int(round(dw / 2 - 0.1))
    bottom, right = int(round(dh / 2 + 0.1)), int(round(dw / 2 + 0.1))
    if top or bottom or left or right:
        img = np.pad(img, ((top, bottom), (left, right), (0, 0)), constant_values=114)
    return np.ascontiguousarray(img), r, left, top


class YoloV8Detector:
    def __init__(self, weights, nc: int = 1, imgsz: int = 256, device="cuda:0") -> None:
        if isinstance(weights, (str, os.PathLike)):
            p = str(weights)
            if p.endswith(".pt"):
                raise OpenGlottalHipError(
                    f"{p}: ultralytics .pt checkpoints are pickles of ultralytics classes; export the state_dict to .npz "
                    "where ultralytics is installed (see openglottal_amd/yolo.py)")
            z = np.load(p)
            weights = {k: z[k] for k in z.files}
        self.imgsz = imgsz
        self.nc = nc
        check(lib().og_init(_device_index(device)), "og_init")
        h = lib().og_yolo_create(nc)
        if not h:
            check(-1, "og_yolo_create")
        try:
            for k, v in weights.items():
                if hasattr(v, "detach"):
                    v = v.detach().cpu().numpy()
                v = np.asarray(v)
                if k.endswith("num_batches_tracked"):
                    v64 = np.ascontiguousarray(v, dtype=np.int64).reshape(-1)
                    check(lib().og_yolo_set_tensor(h, k.encode(), ptr(v64), (C.c_int64 * 1)(0), 0, _lib.OG_DTYPE_I64), k)
                    continue
                v = np.ascontiguousarray(v, dtype=np.float32)
                shp = (C.c_int64 * max(1, v.ndim))(*v.shape)
                check(lib().og_yolo_set_tensor(h, k.encode(), ptr(v), shp, v.ndim, _lib.OG_DTYPE_F32), f"set_tensor({k})")
            check(lib().og_yolo_finalize(h), "og_yolo_finalize")
        except Exception:
            lib().og_yolo_destroy(h)
            raise
        self._h = h

    def set_option(self, name: str, value: int) -> None:
        """Tuning knobs of the C-ABI (``og_yolo_set_option``): ``latency_batch``, ``splitk_slots``, ``splitk_div``."""
        check(lib().og_yolo_set_option(self._h, name.encode(), int(value)), f"og_yolo_set_option({name})")

    def detect_batch(self, frames_bgr: np.ndarray, conf: float = 0.25, want_pred: bool = False):
        """``[B,H,W,3]`` u8 BGR at network size (sides multiples of 32) → ``best [B,5]`` (+ ``pred [B,A,5]``)."""
        f = np.ascontiguousarray(frames_bgr, dtype=np.uint8)
        B, H, W, _ = f.shape
        best = np.empty((B, 5), np.float32)
        A = lib().og_yolo_num_anchors(self._h, H, W)
        if A < 0:
            check(A, "og_yolo_num_anchors")
        pred = np.empty((B, A, 5), np.float32) if want_pred else None
        check(lib().og_yolo_detect_u8(self._h, ptr(f), B, H, W, float(conf), ptr(best), ptr(pred)), "og_yolo_detect_u8")
        return (best, pred) if want_pred else best

    def detect_dev(self, bgr_dev, B: int, H: int, W: int, conf: float = 0.25) -> np.ndarray:
        """``[B,H,W,3]`` u8 BGR frames RESIDENT ON THE DEVICE at network size (sides multiples of 32) → ``best [B,5]`` on the
        host (20 bytes per frame come back)."""
        import torch

        best = torch.empty((B, 5), dtype=torch.float32, device=bgr_dev.device)
        check(lib().og_yolo_detect_u8_dev(self._h, ptr(bgr_dev), B, H, W, float(conf), ptr(best), None), "og_yolo_detect_u8_dev")
        check(lib().og_yolo_sync(self._h), "og_yolo_sync")
        return best.cpu().numpy()

    def detect_frames(self, frames_bgr, conf: float = 0.25) -> np.ndarray:
        """Frames of ONE size ``[B,H,W,3]`` (or ``[B,H,W]`` gray) at their ORIGINAL resolution → ``best [B,5]`` in
        original-frame pixels (conf = -1: no detection): what the ultralytics predictor does per call of
        ``self.model(frame_bgr, conf=...)`` (detector.py:58) — letterbox to ``imgsz``, network, ``scale_boxes`` back,
        clip — for the whole batch in one device pass.  ``__call__`` is this with B = 1, so the batched callers
        (`features.area_waveform`, `evaluate.evaluate`, `dist.sharded_gated_area_waveform`) and the per-frame
        ``TemporalDetector.detect`` see the same boxes for every frame size, not only where the letterbox is the identity."""
        f = np.asarray(frames_bgr)
        if f.ndim == 3:
            f = np.repeat(f[..., None], 3, axis=-1)
        B, H0, W0 = f.shape[:3]
        if B == 0:
            return np.zeros((0, 5), np.float32)
        first, gain, px, py = letterbox_bgr(f[0], self.imgsz)
        if first.shape == f[0].shape and (gain, px, py) == (1.0, 0, 0):
            imgs = f
        else:
            imgs = np.stack([first] + [letterbox_bgr(x, self.imgsz)[0] for x in f[1:]])
        best = self.detect_batch(imgs, conf).copy()
        hit = best[:, 4] >= 0
        if (gain, px, py) != (1.0, 0, 0):  # scale_boxes back to the original frame
            best[:, [0, 2]] = (best[:, [0, 2]] - np.float32(px)) / np.float32(gain)
            best[:, [1, 3]] = (best[:, [1, 3]] - np.float32(py)) / np.float32(gain)
        best[:, [0, 2]] = best[:, [0, 2]].clip(0, W0)
        best[:, [1, 3]] = best[:, [1, 3]].clip(0, H0)
        best[~hit, :4] = 0
        return best.astype(np.float32)

    def __call__(self, frame_bgr: np.ndarray, conf: float = 0.25):
        """Backend protocol of ``TemporalDetector``: → ``(xyxy [n,4] f32, conf [n] f32)``, n ∈ {0,1}: the
        top-confidence detection in ORIGINAL frame pixels (what detector.py:61-64 consumes)."""
        b = self.detect_frames(np.asarray(frame_bgr)[None], conf)[0]
        if b[4] < 0:
            return np.zeros((0, 4), np.float32), np.zeros(0, np.float32)
        return b[None, :4].astype(np.float32), b[4:5].astype(np.float32)

    def submit(self, frame_bgr: np.ndarray, conf: float = 0.25) -> None:
        """First half of ``__call__`` (``og_yolo_detect_u8_begin``): letterbox on the host, enqueue the detector's chain on its own
        stream, return.  ``result()`` delivers what ``__call__`` would have returned.  In the reference's frame loop
        (features.py:235-245) the box only gates the count of the U-Net's mask, so the U-Net call of the same frame fits in between."""
        f = np.asarray(frame_bgr)
        if f.ndim == 2:
            f = np.repeat(f[..., None], 3, axis=-1)
        img, gain, px, py = letterbox_bgr(f, self.imgsz)
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W = img.shape[:2]
        check(lib().og_yolo_detect_u8_begin(self._h, ptr(img), 1, H, W, float(conf)), "og_yolo_detect_u8_begin")
        self._pending = (f.shape[0], f.shape[1], gain, px, py)

    def result(self):
        """Second half of ``__call__``: ``(xyxy [n,4] f32, conf [n] f32)``, n ∈ {0,1}, in ORIGINAL frame pixels."""
        if getattr(self, "_pending", None) is None:
            raise OpenGlottalHipError("result() without submit()")
        H0, W0, gain, px, py = self._pending
        self._pending = None
        best = np.empty((1, 5), np.float32)
        check(lib().og_yolo_detect_u8_end(self._h, ptr(best)), "og_yolo_detect_u8_end")
        b = best[0]
        if b[4] < 0:
            return np.zeros((0, 4), np.float32), np.zeros(0, np.float32)
        if (gain, px, py) != (1.0, 0, 0):  # scale_boxes back to the original frame, as detect_frames
            b[[0, 2]] = (b[[0, 2]] - np.float32(px)) / np.float32(gain)
            b[[1, 3]] = (b[[1, 3]] - np.float32(py)) / np.float32(gain)
        b[[0, 2]] = b[[0, 2]].clip(0, W0)
        b[[1, 3]] = b[[1, 3]].clip(0, H0)
        return b[None, :4].astype(np.float32), b[4:5].astype(np.float32)

    def activation(self, name: str, B: int = 1) -> np.ndarray:
        dims = (C.c_int * 3)()
        cap = 1 << 22
        buf = np.empty(cap, np.float32)
        check(lib().og_yolo_get_activation(self._h, name.encode(), B, ptr(buf), cap, dims), f"og_yolo_get_activation({name})")
        c, h, w = dims[0], dims[1], dims[2]
        return buf[: B * c * h * w].reshape(B, c, h, w).copy()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().og_yolo_destroy(self._h)
                self._h = None
        except Exception:
            pass


def nms(xyxy: np.ndarray, conf: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.7, max_det: int = 300) -> np.ndarray:
    """Greedy single-class NMS over decoded candidates (host side; ultralytics defaults IoU 0.7, 300 dets)."""
    idx = np.flatnonzero(conf > conf_thres)
    idx = idx[np.argsort(-conf[idx], kind="stable")]
    area = (xyxy[:, 2] - xyxy[:, 0]) * (xyxy[:, 3] - xyxy[:, 1])
    keep = []
    while idx.size and len(keep) < max_det:
        i, rest = idx[0], idx[1:]
        keep.append(i)
        iw = np.clip(np.minimum(xyxy[i, 2], xyxy[rest, 2]) - np.maximum(xyxy[i, 0], xyxy[rest, 0]), 0, None)
        ih = np.clip(np.minimum(xyxy[i, 3], xyxy[rest, 3]) - np.maximum(xyxy[i, 1], xyxy[rest, 1]), 0, None)
        inter = iw * ih
        idx = rest[inter / (area[i] + area[rest] - inter + 1e-12) <= iou_thres]
    return np.array(keep, dtype=np.int64)


def load_detector_backend(path: str):
    return YoloV8Detector(path)
