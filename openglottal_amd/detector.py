"""`TemporalDetector` (`openglottal/models/detector.py:9-102`).

The temporal state machine is reference code and is restated here from its
specification; the YOLOv8n network behind it is third-party (ultralytics, not
vendored) — see ``yolo.py`` for the native detector backend.  ``model_path``
may also be a callable ``backend(frame_bgr, conf) -> (xyxy [n,4], conf [n])``,
which is how the parity tests script detections.
"""

from __future__ import annotations

import math

import numpy as np


class TemporalDetector:
    def __init__(self, model_path, conf: float = 0.25, max_shift_px: int = 30, padding: int = 8,
                 max_hold_frames: int = 3) -> None:
        if callable(model_path):
            self.model = model_path
        else:
            from .yolo import load_detector_backend

            self.model = load_detector_backend(str(model_path))
        self.conf = conf
        self.max_shift = max_shift_px
        self.padding = padding
        self.max_hold_frames = max_hold_frames
        self.reset()

    def reset(self) -> None:
        self._centre = None       # (cx, cy) of the last accepted detection
        self._size = None         # (w, h) incl. padding
        self._misses = 0

    @property
    def crop_size(self):
        return None if self._size is None else (self._size[0], self._size[1])

    def detect(self, frame_bgr: np.ndarray):
        H, W = frame_bgr.shape[:2]
        xyxy, confs = self.model(frame_bgr, self.conf)
        return self.update(xyxy, confs, W, H)

    def submit(self, frame_bgr: np.ndarray) -> None:
        """``detect`` in two halves (an extension, not in the reference): ``submit(frame)`` starts the network on the detector's own
        stream, ``result()`` waits for it and runs the state machine.  The reference's loop calls ``detect(frame)`` and then
        ``unet_segment_frame(gray)`` (features.py:235-245); the mask does not depend on the box, so with ``submit`` before the U-Net
        call and ``result`` after it the two networks overlap on the GPU.  Backends without ``submit`` run inside ``result``."""
        self._pending_frame = frame_bgr
        if hasattr(self.model, "submit"):
            self.model.submit(frame_bgr, self.conf)

    def result(self):
        frame_bgr = self._pending_frame
        self._pending_frame = None
        H, W = frame_bgr.shape[:2]
        xyxy, confs = self.model.result() if hasattr(self.model, "submit") else self.model(frame_bgr, self.conf)
        return self.update(xyxy, confs, W, H)

    def update(self, xyxy, confs, W: int, H: int):
        """One step of the state machine given this frame's raw detections."""
        fresh = None
        if confs is not None and len(confs):
            x1, y1, x2, y2 = (np.float32(v) for v in np.asarray(xyxy, dtype=np.float32)[int(np.argmax(confs))])
            centre = ((x1 + x2) / 2, (y1 + y2) / 2)
            size = (int(x2 - x1) + 2 * self.padding, int(y2 - y1) + 2 * self.padding)
            fresh = (centre, size)
            if self._centre is not None:
                jump = np.hypot(centre[0] - self._centre[0], centre[1] - self._centre[1])
                if jump > self.max_shift:
                    fresh = None  # spurious jump: treated as a miss
        if fresh is not None:
            self._centre, self._size = fresh
            self._misses = 0
        elif self._centre is not None:
            self._misses += 1
            if self._misses > self.max_hold_frames:
                self.reset()
                return None
        if self._centre is None:
            return None
        hw, hh = self._size[0] // 2, self._size[1] // 2
        cx = int(np.clip(self._centre[0], hw, W - hw))
        cy = int(np.clip(self._centre[1], hh, H - hh))
        return (cx - hw, cy - hh, cx + hw, cy + hh)

    def crop(self, frame: np.ndarray, box):
        if box is None:
            return frame
        x1, y1, x2, y2 = box
        return frame[y1:y2, x1:x2]
