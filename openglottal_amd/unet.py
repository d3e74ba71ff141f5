"""``UNet``: host-side mirror of the reference model class over the HIP library.

Same constructor, ``.to(device)``, ``.load_state_dict(sd)``, ``.eval()`` and
``__call__`` as `openglottal/models/unet.py:36-88`, so the reference's CLI and
scripts (`cli.py:61-65`, `scripts/benchmark_video_speed.py:42-44`) run
unchanged after swapping the import.  All arithmetic happens in
``libopenglottal_hip.so``; this class only marshals tensors.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import OpenGlottalHipError, check, lib, ptr


def _device_index(device) -> int:
    """'cuda', 'cuda:3', torch.device('cuda', 3), 3  ->  ordinal.  CPU is refused."""
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s.startswith(("cuda", "hip")):
        return int(s.split(":")[1]) if ":" in s else 0
    raise OpenGlottalHipError(
        f"device {s!r}: openglottal_amd runs on MI355X (pass 'cuda' / 'cuda:N'); there is no CPU path"
    )


class UNet:
    """Drop-in for ``openglottal.models.UNet`` (inference only)."""

    def __init__(self, in_ch: int = 1, out_ch: int = 1, features=(32, 64, 128, 256)) -> None:
        self.in_ch, self.out_ch = int(in_ch), int(out_ch)
        self.features = tuple(int(f) for f in features)
        self._h = None
        self._device = None
        self._sd: dict[str, np.ndarray] | None = None
        self.training = True
        if self.in_ch != 1 or self.out_ch != 1:
            raise OpenGlottalHipError("only UNet(1, 1, ...) is implemented (what every reference pipeline constructs)")

    # ── reference surface ────────────────────────────────────────────────────
    def to(self, device):
        idx = _device_index(device)
        if self._device is not None and idx != self._device:
            self._release()
        self._device = idx
        check(lib().og_init(idx), f"og_init({idx})")
        if self._sd is not None and self._h is None:
            self._build()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise OpenGlottalHipError("openglottal_amd.UNet is inference-only (eval-mode BatchNorm is folded)")
        return self.eval()

    def state_dict(self):
        if self._sd is None:
            raise OpenGlottalHipError("no state_dict loaded")
        import torch

        return {k: torch.from_numpy(v.copy()) for k, v in self._sd.items()}

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts torch tensors or numpy arrays, keys as in the reference (118 entries
        at 4 levels incl. ``num_batches_tracked``).  Strict, like the reference's call."""
        sd = {}
        for k, v in state_dict.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            v = np.asarray(v)
            sd[k] = v if k.endswith("num_batches_tracked") else np.ascontiguousarray(v, dtype=np.float32)
        self._sd = sd
        self._release()
        if self._device is not None:
            self._build()
        return self

    def __call__(self, x):
        """``UNet.forward`` (unet.py:74-88): f32 ``[B,1,H,W]`` → logits ``[B,1,H,W]``."""
        self._require()
        is_torch = hasattr(x, "detach")
        if is_torch:
            dev = x.device
            xin = np.ascontiguousarray(x.detach().to("cpu").numpy(), dtype=np.float32)
        else:
            xin = np.ascontiguousarray(x, dtype=np.float32)
        if xin.ndim != 4 or xin.shape[1] != 1:
            raise OpenGlottalHipError(f"expected [B,1,H,W], got {xin.shape}")
        B, _, H, W = xin.shape
        out = np.empty((B, 1, H, W), dtype=np.float32)
        check(lib().og_unet_forward_f32(self._h, ptr(xin), B, H, W, ptr(out)), "og_unet_forward_f32")
        if is_torch:
            import torch

            return torch.from_numpy(out).to(dev)
        return out

    forward = __call__

    # ── batched fast path (what extract_features_unet / bench use) ──────────
    def segment(self, gray, threshold: float = 0.5, boxes=None, want_mask: bool = True, want_logits: bool = False,
                want_area: bool = True):
        """`unet_segment_frame` + area count for host frames ``[B,H,W]`` u8.

        Returns ``(mask u8 {0,255} | None, area int32 [B] | None, logits f32 | None)``.
        ``boxes``: int32 ``[B,4]`` (x1,y1,x2,y2), ``x1 < 0`` ⇒ no detection ⇒ area 0.
        ``want_area=False`` (the mask-only call of utils.py:218-241) spares the count reduction's launch.
        """
        self._require()
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        if g.ndim == 2:
            g = g[None]
        B, H, W = g.shape
        mask = np.empty((B, H, W), np.uint8) if want_mask else None
        area = np.zeros(B, np.int32) if want_area else None
        logits = np.empty((B, H, W), np.float32) if want_logits else None
        bx = None if boxes is None else np.ascontiguousarray(boxes, dtype=np.int32).reshape(B, 4)
        check(lib().og_unet_segment_u8(self._h, ptr(g), B, H, W, float(threshold), ptr(bx), ptr(mask), ptr(area),
                                       ptr(logits)), "og_unet_segment_u8")
        return mask, area, logits

    def segment_stream(self, frames, threshold: float = 0.5, boxes=None, want_mask: bool = False):
        """The frame loop over host frames through the streaming ingest engine (``og_unet_stream_u8``): ``[B,H,W]`` gray or
        ``[B,H,W,3]`` BGR u8 (BGR→gray of features.py:235 on the device), numpy or a pinned torch tensor.  Device memory
        stays bounded by a few micro-batches however long the video is.  Returns ``(mask | None, area int32 [B])``."""
        self._require()
        if isinstance(frames, (list, tuple)):    # the reference's `frames_bgr` list: no stacked copy, the engine gathers
            return self._segment_frame_list(frames, threshold, boxes, want_mask)
        if hasattr(frames, "data_ptr"):          # torch CPU tensor (pinned or not): use its memory in place
            if frames.device.type != "cpu" or str(frames.dtype) != "torch.uint8" or not frames.is_contiguous():
                raise OpenGlottalHipError("segment_stream expects a contiguous uint8 host tensor")
            f, shape = frames, tuple(frames.shape)
        else:
            f = np.ascontiguousarray(frames, dtype=np.uint8)
            shape = f.shape
        if len(shape) == 4 and shape[-1] == 3:
            ch = 3
        elif len(shape) == 3:
            ch = 1
        else:
            raise OpenGlottalHipError(f"expected [B,H,W] or [B,H,W,3] frames, got {shape}")
        B, H, W = shape[:3]
        mask = np.empty((B, H, W), np.uint8) if want_mask else None
        area = np.zeros(B, np.int32)
        bx = None if boxes is None else np.ascontiguousarray(boxes, dtype=np.int32).reshape(B, 4)
        check(lib().og_unet_stream_u8(self._h, ptr(f), B, H, W, ch, float(threshold), ptr(bx), ptr(mask), ptr(area)),
              "og_unet_stream_u8")
        return mask, area

    def _segment_frame_list(self, frames, threshold, boxes, want_mask):
        """``og_unet_stream_frames_u8``: a list of separately allocated ``[H,W]`` / ``[H,W,3]`` u8 frames of one shape; each
        frame is copied once, by the engine, into the pinned slot of its micro-batch (no ``np.stack``)."""
        B = len(frames)
        if B == 0:
            return (np.empty((0, 0, 0), np.uint8) if want_mask else None), np.zeros(0, np.int32)
        keep = [f if (isinstance(f, np.ndarray) and f.dtype == np.uint8 and f.flags.c_contiguous) else np.ascontiguousarray(f, dtype=np.uint8)
                for f in frames]
        shape = keep[0].shape
        if any(f.shape != shape for f in keep) or len(shape) not in (2, 3) or (len(shape) == 3 and shape[2] != 3):
            raise OpenGlottalHipError("segment_stream(list): frames must share one [H,W] or [H,W,3] shape")
        H, W = shape[:2]
        ch = 3 if len(shape) == 3 else 1
        ptrs = (C.c_void_p * B)(*[f.ctypes.data for f in keep])
        mask = np.empty((B, H, W), np.uint8) if want_mask else None
        area = np.zeros(B, np.int32)
        bx = None if boxes is None else np.ascontiguousarray(boxes, dtype=np.int32).reshape(B, 4)
        check(lib().og_unet_stream_frames_u8(self._h, ptrs, B, H, W, ch, float(threshold), ptr(bx), ptr(mask), ptr(area)),
              "og_unet_stream_frames_u8")
        return mask, area

    def segment_dev(self, gray_dev, B: int, H: int, W: int, area_dev, threshold: float = 0.5, boxes_dev=None,
                    mask_dev=None, logits_dev=None) -> None:
        """Device-pointer, asynchronous variant (torch CUDA tensors or raw ints)."""
        self._require()
        check(lib().og_unet_segment_u8_dev(self._h, ptr(gray_dev), B, H, W, float(threshold), ptr(boxes_dev),
                                           ptr(mask_dev), ptr(area_dev), ptr(logits_dev)), "og_unet_segment_u8_dev")

    def segment_crops(self, gray, boxes, crop_size: int = 256, threshold: float = 0.5):
        """YOLO-Crop+UNet for ``[B,H,W]`` frames on the device: crop → letterbox (NEAREST) → U-Net → project back →
        paste (`scripts/eval_girafe.py:127-159`).  ``boxes``: per frame ``(x1,y1,x2,y2)`` inside the frame, or
        ``None``.  Returns full-frame masks ``[B,H,W]`` u8 {0,255}."""
        self._require()
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        B, H, W = g.shape
        bx = np.full((B, 4), -1, np.int32)
        geo = np.zeros((B, 4), np.int32)
        for i, b in enumerate(boxes):
            if b is None:
                continue
            x1, y1, x2, y2 = (int(v) for v in b)
            x1, x2 = max(0, min(W, x1)), max(0, min(W, x2))   # python slice clamping for in-frame boxes
            y1, y2 = max(0, min(H, y1)), max(0, min(H, y2))
            h, w = y2 - y1, x2 - x1
            if h <= 0 or w <= 0:
                continue
            scale = crop_size / max(h, w)
            nh, nw = int(round(h * scale)), int(round(w * scale))
            bx[i] = (x1, y1, x2, y2)
            geo[i] = ((crop_size - nh) // 2, (crop_size - nw) // 2, nh, nw)
        out = np.empty((B, H, W), np.uint8)
        check(lib().og_unet_segment_crops_u8(self._h, ptr(g), B, H, W, ptr(bx), ptr(geo), int(crop_size), float(threshold), ptr(out)),
              "og_unet_segment_crops_u8")
        return out

    def bgr2gray_dev(self, bgr_dev, B: int, H: int, W: int, gray_dev) -> None:
        """``cv2.cvtColor(BGR2GRAY)`` (features.py:235) on device buffers ``[B,H,W,3]`` u8 → ``[B,H,W]`` u8, asynchronous."""
        self._require()
        check(lib().og_bgr2gray_dev(self._h, ptr(bgr_dev), B, H, W, ptr(gray_dev)), "og_bgr2gray_dev")

    def mask_area_dev(self, mask_dev, B: int, H: int, W: int, boxes_dev, area_dev) -> None:
        """Box-gated recount ``sum(mask[y1:y2, x1:x2] > 0)`` (features.py:244-245) on resident masks, asynchronous."""
        self._require()
        check(lib().og_mask_area_dev(self._h, ptr(mask_dev), B, H, W, ptr(boxes_dev), ptr(area_dev)), "og_mask_area_dev")

    def canvas_letterbox(self, frames, size: int = 256, value: int = 0) -> np.ndarray:
        """`letterbox` of scripts/eval_bagls.py:46-70 for a list of frames of MIXED sizes on the device: all 2-D (gray /
        masks: NEAREST) or all ``HxWx3`` (BGR: LINEAR) → ``[B,size,size(,3)]`` u8.  Pixel-identical to
        ``geometry.letterbox`` (tests/test_gpu_bagls.py)."""
        from .geometry import letterbox_geometry, pack_frames

        self._require()
        frames = [np.asarray(f) for f in frames]
        B = len(frames)
        ch = 3 if (B and frames[0].ndim == 3) else 1
        if any((f.ndim == 3) != (ch == 3) or (f.ndim == 3 and f.shape[2] != 3) for f in frames):
            raise OpenGlottalHipError("canvas_letterbox: frames must be all 2-D or all HxWx3")
        out = np.empty((B, size, size, 3) if ch == 3 else (B, size, size), np.uint8)
        if B == 0:
            return out
        packed, offsets, shapes = pack_frames(frames)
        geom = np.array([letterbox_geometry(int(h), int(w), size) for h, w in shapes], np.int32)
        check(lib().og_canvas_letterbox_u8(self._h, ptr(packed), ptr(offsets), ptr(shapes), B, ch, int(size), ptr(geom), int(value), ptr(out)),
              "og_canvas_letterbox_u8")
        return out

    def sync(self) -> None:
        self._require()
        check(lib().og_unet_sync(self._h), "og_unet_sync")

    def reserve(self, frames_per_launch: int, H: int = 256, W: int = 256) -> None:
        """Pre-allocate the activation arenas (all lanes) for micro-batches of this size: later calls at this or any
        smaller footprint (other frame sizes included) neither allocate nor re-capture graphs."""
        self._require()
        check(lib().og_unet_reserve(self._h, int(frames_per_launch), int(H), int(W)), "og_unet_reserve")

    def set_chunk(self, frames_per_launch: int) -> None:
        self._require()
        check(lib().og_unet_set_chunk(self._h, int(frames_per_launch)), "og_unet_set_chunk")
        self._chunk = int(frames_per_launch)

    @property
    def chunk(self) -> int:
        """Frames per kernel chain (micro-batch) the handle is set to (library default 32)."""
        return getattr(self, "_chunk", 32)

    def set_graphs(self, enable: bool) -> None:
        self._require()
        check(lib().og_unet_set_graphs(self._h, int(bool(enable))), "og_unet_set_graphs")

    def set_option(self, name: str, value: int) -> None:
        self._require()
        check(lib().og_unet_set_option(self._h, name.encode(), int(value)), f"og_unet_set_option({name})")

    def timer_start(self) -> None:
        check(lib().og_timer_start(self._h), "og_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        check(lib().og_timer_stop(self._h, C.byref(ms)), "og_timer_stop")
        return float(ms.value)

    def profile(self, gray_dev, B: int, H: int, W: int, reps: int = 5) -> list[dict]:
        """Per-launch HIP-event timings of one chain (bench.py's roofline leg)."""
        self._require()
        n_max = 128
        layers = C.create_string_buffer(64 * n_max)
        kernels = C.create_string_buffer(64 * n_max)
        ms = (C.c_float * n_max)()
        fl = (C.c_double * n_max)()
        n = C.c_int(0)
        check(lib().og_unet_profile(self._h, ptr(gray_dev), B, H, W, reps, n_max, layers, kernels, ms, fl, C.byref(n)),
              "og_unet_profile")
        out = []
        for i in range(n.value):
            out.append({"layer": layers.raw[64 * i:64 * i + 64].split(b"\0")[0].decode(),
                        "kernel": kernels.raw[64 * i:64 * i + 64].split(b"\0")[0].decode(),
                        "ms": float(ms[i]), "flops": float(fl[i])})
        return out

    def clock_probe(self, gray_dev, B: int, H: int, W: int) -> list[float]:
        """Median in-kernel shader clock (MHz) per launch of one chain (diagnostic)."""
        self._require()
        mhz = (C.c_double * 64)()
        n = C.c_int(0)
        check(lib().og_unet_clock_probe(self._h, ptr(gray_dev), B, H, W, 64, mhz, C.byref(n)), "og_unet_clock_probe")
        return [float(mhz[i]) for i in range(n.value)]

    def flops_per_frame(self, H: int = 256, W: int = 256) -> float:
        self._require()
        return float(lib().og_unet_flops_per_frame(self._h, H, W))

    def activation(self, name: str, B: int = 1) -> np.ndarray:
        """Layer-boundary tensor of the last forward, NCHW f32 (parity tests)."""
        self._require()
        dims = (C.c_int * 3)()
        cap = 1 << 20
        while True:
            buf = np.empty(cap, np.float32)
            rc = lib().og_unet_get_activation(self._h, name.encode(), B, ptr(buf), cap, dims)
            if rc == 0:
                c, h, w = dims[0], dims[1], dims[2]
                return buf[: B * c * h * w].reshape(B, c, h, w).copy()
            if dims[0] and cap < B * dims[0] * dims[1] * dims[2]:
                cap = B * dims[0] * dims[1] * dims[2]
                continue
            check(rc, f"og_unet_get_activation({name})")

    # ── internals ────────────────────────────────────────────────────────────
    def _require(self) -> None:
        if self._h is None:
            if self._sd is None:
                raise OpenGlottalHipError("UNet has no weights: call load_state_dict() first")
            if self._device is None:
                raise OpenGlottalHipError("UNet is not on a device: call .to('cuda') first (no CPU path)")
            self._build()

    def _build(self) -> None:
        l = lib()
        feats = (C.c_int * len(self.features))(*self.features)
        h = l.og_unet_create(feats, len(self.features), self.in_ch, self.out_ch)
        if not h:
            check(-1, "og_unet_create")
        try:
            for k, v in self._sd.items():
                if k.endswith("num_batches_tracked"):
                    v64 = np.ascontiguousarray(v, dtype=np.int64).reshape(-1)
                    shape = (C.c_int64 * 1)(0)
                    check(l.og_unet_set_tensor(h, k.encode(), ptr(v64), shape, 0, _lib.OG_DTYPE_I64), f"set_tensor({k})")
                    continue
                shape = (C.c_int64 * max(1, v.ndim))(*v.shape)
                check(l.og_unet_set_tensor(h, k.encode(), ptr(v), shape, v.ndim, _lib.OG_DTYPE_F32), f"set_tensor({k})")
            check(l.og_unet_finalize(h), "og_unet_finalize")
        except Exception:
            l.og_unet_destroy(h)
            raise
        self._h = h

    def _release(self) -> None:
        if self._h is not None:
            lib().og_unet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
