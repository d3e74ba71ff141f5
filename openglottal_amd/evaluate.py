"""Accuracy harness counterpart of `scripts/eval_girafe.py:225-324` / `scripts/eval_bagls.py:120-232`.

Same pipelines, metrics and table as the reference for the rows on the U-Net/YOLO path —
``unet-only``, ``yolo+unet`` (mask zeroed outside the box), ``yolo-crop+unet``
(crop → letterbox 256 → U-Net → unletterbox → paste) — but batched: every frame's U-Net
mask comes from ONE device pass, every crop from a second one.  The classical-CV rows
(``yolo+otsu``, ``yolo+motion``) are out of scope.

Inputs are arrays (``[N,H,W,3]`` BGR or ``[N,H,W]`` gray, GT masks ``[N,H,W]``): image-file
decoding is host I/O behind cv2 and stays with the caller.
"""

from __future__ import annotations

import json
from collections import defaultdict

import numpy as np

from .geometry import INTER_NEAREST, letterbox, letterbox_with_info, unletterbox
from .utils import NET_SIZE, bgr_to_gray, frame_metrics, unet_segment_frame

PIPELINES = ["unet-only", "yolo+unet", "yolo-crop+unet"]


def unet_on_crop(gray: np.ndarray, box, model, device=None, crop_size: int = NET_SIZE) -> np.ndarray:
    """One frame, reference semantics (`scripts/eval_girafe.py:127-159`)."""
    return unet_on_crops(gray[None], [box], model, device, crop_size)[0]


def unet_on_crops(grays: np.ndarray, boxes, model, device=None, crop_size: int = NET_SIZE) -> np.ndarray:
    """Batched YOLO-Crop+UNet: all non-empty crops are letterboxed to ``crop_size`` and
    segmented in one call; masks are projected back NEAREST and pasted into zero frames."""
    if device is not None and getattr(model, "_device", None) is None:
        model.to(device)
    if hasattr(model, "segment_crops") and crop_size % 16 == 0 and grays.ndim == 3 and all(
            b is None or (0 <= b[0] and 0 <= b[1]) for b in boxes):
        return model.segment_crops(grays, boxes, crop_size)   # whole geometry on the device
    out = np.zeros_like(grays)
    jobs, tiles = [], []
    for i, (g, b) in enumerate(zip(grays, boxes)):
        if b is None:
            continue
        x1, y1, x2, y2 = (int(v) for v in b)
        crop = g[y1:y2, x1:x2]
        if crop.size == 0:
            continue
        boxed, pt, pl, ch, cw = letterbox_with_info(crop, crop_size, value=0)
        jobs.append((i, (x1, y1, x2, y2), crop.shape[:2], (pt, pl, ch, cw)))
        tiles.append(boxed)
    if not jobs:
        return out
    if crop_size == NET_SIZE:
        masks, _, _ = model.segment(np.stack(tiles), want_mask=True)
    else:
        masks = np.stack([unet_segment_frame(t, model, device) for t in tiles])
    for (i, (x1, y1, x2, y2), (h, w), (pt, pl, ch, cw)), m in zip(jobs, masks):
        out[i][y1:y2, x1:x2] = unletterbox(m, pt, pl, ch, cw, h, w, interp=INTER_NEAREST)
    return out


def evaluate(frames, gts, unet_model, detector=None, crop_model=None, device=None, patients=None,
             reset_every_frame: bool = False, canvas: int | None = None, crop_pad: int = 0):
    """Returns ``(agg, patient_dice, det_stats)``.

    ``patients``: per-frame patient id → ``detector.reset()`` at every change (GIRAFE,
    eval_girafe.py:246-247).  ``reset_every_frame=True`` is the BAGLS convention
    (eval_bagls.py:164-166).  ``canvas``: letterbox frames and GT to this size first
    (eval_bagls.py:153-155).  ``crop_model`` defaults to ``unet_model`` (eval_girafe.py:302).
    """
    frames = [np.asarray(f) for f in frames]
    gts = [np.asarray(g) for g in gts]
    if canvas is not None:
        frames = [letterbox(f, canvas) for f in frames]
        gts = [letterbox(g, canvas) for g in gts]
    n = len(frames)
    patients = list(patients) if patients is not None else ["all"] * n
    grays = np.stack([bgr_to_gray(f) for f in frames])
    gts = np.stack(gts)
    if device is not None and getattr(unet_model, "_device", None) is None:
        unet_model.to(device)

    boxes = [None] * n
    det_stats = {"tp": 0, "fp": 0, "fn": 0, "n_pos_gt": 0}
    if detector is not None:
        prev = object()
        # native backend and frames at network size: the YOLO network is per-frame independent -> ONE batched device pass;
        # only the O(1)/frame temporal state machine (with its per-patient / per-frame resets) stays sequential
        batch = getattr(getattr(detector, "model", None), "detect_frames", None)
        shapes = {f.shape for f in frames}
        best = None
        if batch is not None and len(shapes) == 1:   # one size: letterbox + network + scale-back for all frames in one pass
            best = batch(np.stack(frames), detector.conf)
        for i, f in enumerate(frames):
            if reset_every_frame or patients[i] != prev:
                detector.reset()
                prev = patients[i]
            if best is not None:
                H, W = f.shape[:2]
                boxes[i] = detector.update(best[i:i + 1, :4], best[i:i + 1, 4], W, H) if best[i, 4] >= 0 else detector.update(None, None, W, H)
            else:
                bgr = f if f.ndim == 3 else np.repeat(f[..., None], 3, axis=-1)
                boxes[i] = detector.detect(bgr)
            gt_pos = bool((gts[i] > 0).any())
            det_stats["n_pos_gt"] += int(gt_pos)
            if boxes[i] is not None:
                H, W = gts[i].shape
                x1, y1, x2, y2 = (int(v) for v in boxes[i])
                x1, x2 = max(0, min(W, x1)), max(0, min(W, x2))
                y1, y2 = max(0, min(H, y1)), max(0, min(H, y2))
                det_stats["tp" if gts[i][y1:y2, x1:x2].any() else "fp"] += 1
            elif gt_pos:
                det_stats["fn"] += 1

    if grays.shape[1:] == (NET_SIZE, NET_SIZE):
        masks_u, _, _ = unet_model.segment(grays, want_mask=True)
    else:
        masks_u = np.stack([unet_segment_frame(g, unet_model, device) for g in grays])

    agg = {p: {"dice": [], "iou": [], "n_det": 0, "n_total": 0} for p in PIPELINES}
    patient_dice: dict = defaultdict(lambda: defaultdict(list))

    def add(pipe, i, mask, detected):
        d, j = frame_metrics(mask, gts[i])
        agg[pipe]["dice"].append(d)
        agg[pipe]["iou"].append(j)
        agg[pipe]["n_total"] += 1
        agg[pipe]["n_det"] += int(detected)
        patient_dice[patients[i]][pipe].append(d)

    for i in range(n):
        add("unet-only", i, masks_u[i], False)
    if detector is not None:
        for i in range(n):
            m = np.zeros_like(masks_u[i])
            if boxes[i] is not None:
                x1, y1, x2, y2 = boxes[i]
                m[y1:y2, x1:x2] = masks_u[i][y1:y2, x1:x2]
            add("yolo+unet", i, m, boxes[i] is not None)
        cboxes = boxes
        if crop_pad:
            H, W = grays.shape[1:]
            cboxes = [None if b is None else (max(0, b[0] - crop_pad), max(0, b[1] - crop_pad),
                                              min(W, b[2] + crop_pad), min(H, b[3] + crop_pad)) for b in boxes]
        masks_c = unet_on_crops(grays, cboxes, crop_model if crop_model is not None else unet_model, device)
        for i in range(n):
            add("yolo-crop+unet", i, masks_c[i], boxes[i] is not None)
    return agg, {k: dict(v) for k, v in patient_dice.items()}, det_stats


def summarize(agg) -> dict:
    """Per pipeline: Det.Recall, mean Dice, mean IoU, Dice≥0.5 % (as `print_table`, eval_girafe.py:355-366)."""
    out = {}
    for p, d in agg.items():
        if not d["n_total"]:
            continue
        out[p] = {
            "det_recall": 1.0 if p == "unet-only" else d["n_det"] / d["n_total"],
            "dice": float(np.mean(d["dice"])),
            "iou": float(np.mean(d["iou"])),
            "dice_ge_0.5_pct": float(np.mean([x >= 0.5 for x in d["dice"]]) * 100),
            "n": d["n_total"],
        }
    return out


def print_table(agg) -> None:
    label = {"unet-only": "U-Net only", "yolo+unet": "YOLO+UNet", "yolo-crop+unet": "YOLO-Crop+UNet"}
    sep = "─" * 76
    print(f"\n{sep}\n  {'Method':<25}  {'Det.Recall':>10}  {'Dice':>8}  {'IoU':>8}  {'Dice≥0.5':>10}\n{sep}")
    for p, r in summarize(agg).items():
        dr = "1.000 *" if p == "unet-only" else f"{r['det_recall']:.3f}"
        print(f"  {label[p]:<25}  {dr:>10}  {r['dice']:>8.3f}  {r['iou']:>8.3f}  {r['dice_ge_0.5_pct']:>9.1f}%")
    print(sep)


def dump_json(path: str, agg, det_stats=None) -> None:
    """Per-frame lists + summary, the shape of `results/bagls_eval.json` (eval_bagls.py:369-391)."""
    with open(path, "w") as f:
        json.dump({"summary": summarize(agg), "per_frame": {p: {"dice": d["dice"], "iou": d["iou"]} for p, d in agg.items()},
                   "det_stats": det_stats}, f, indent=1)
