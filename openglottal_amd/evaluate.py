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
import os
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


def dump_json(path: str, agg, det_stats=None, meta=None) -> None:
    """Per-frame metrics in the shape of the reference's `results/bagls_eval.json` (eval_bagls.py:369-391): one entry per
    pipeline with ``dice`` / ``iou`` lists and the ``n_det`` / ``n_total`` counters, plus ``_meta``."""
    from datetime import datetime

    out = {p: {k: (v if isinstance(v, (int, float)) else [float(x) for x in v]) for k, v in d.items()} for p, d in agg.items()}
    out["_meta"] = dict({"crop_letterbox": True, "written_at": datetime.now().isoformat()}, **(meta or {}))
    if det_stats is not None:
        out["_meta"]["det_stats"] = {k: int(v) for k, v in det_stats.items()}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=2)


# ── C5 on the device: BAGLS front end + three pipelines + metrics without a mask leaving HBM ───────────────────


def metrics_from_counts(tp, n_pred, n_gt) -> tuple[float, float]:
    """`frame_metrics` (eval_bagls.py:75-87) from integer confusion counts, in the reference's float32 arithmetic."""
    tp_, fp, fn = np.float32(tp), np.float32(n_pred - tp), np.float32(n_gt - tp)
    d_den, i_den = 2 * tp_ + fp + fn, tp_ + fp + fn
    return (float(2 * tp_ / d_den) if d_den > 0 else 1.0, float(tp_ / i_den) if i_den > 0 else 1.0)


def evaluate_counts_device(frames, gts, unet_model, detector=None, crop_model=None, canvas: int = NET_SIZE, crop_pad: int = 0,
                           block: int = 512) -> np.ndarray:
    """The BAGLS evaluation loop (eval_bagls.py:120-232) for frames of MIXED sizes, entirely on the device, returning per
    frame ``[tp_u, np_u, ng, tp_yu, np_yu, tp_c, np_c, detected, det_tp, gt_pos]`` (int64 ``[N,10]``): the canvas letterbox of
    frame and GT (``k_canvas_letterbox``), BGR→gray, the stateless detector pass (`detector.reset()` before every frame,
    eval_bagls.py:164-166), the full-frame U-Net, the box-gated row, the crop → 256² → project-back row and the confusion
    counts all stay in HBM; 40 bytes per frame come back.  Frames are independent, so the rows of different shards can
    simply be concatenated (`dist.sharded_eval_counts`)."""
    import torch

    from ._lib import check, lib, ptr
    from .geometry import letterbox_geometry, pack_frames
    from .utils import normalize_box

    n = len(frames)
    out = np.zeros((n, 10), np.int64)
    if n == 0:
        return out
    unet_model._require()
    dev = torch.device("cuda", unet_model._device or 0)
    S = int(canvas)
    native = detector is not None and hasattr(getattr(detector, "model", None), "detect_dev")
    cm = crop_model if crop_model is not None else unet_model
    if crop_model is not None:
        crop_model._require()

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def pack(imgs):   # host side of the front end: one contiguous buffer + per-frame records (runs ahead on a worker thread)
        packed, offsets, shapes = pack_frames(imgs)
        geom = np.array([letterbox_geometry(int(h), int(w), S) for h, w in shapes], np.int32)
        return packed, offsets, shapes, geom

    def letterbox_dev(rec, n_img, ch):
        o = torch.empty((n_img, S, S, 3) if ch == 3 else (n_img, S, S), dtype=torch.uint8, device=dev)
        bufs = [up(a) for a in rec]
        check(lib().og_canvas_letterbox_u8_dev(unet_model._h, ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), n_img, ch, S, ptr(bufs[3]), 0, ptr(o)),
              "og_canvas_letterbox_u8_dev")
        unet_model.sync()   # the staging tensors above may be released now
        return o

    def prepare(lo):
        fb = [np.asarray(f) for f in frames[lo:lo + block]]
        gb = [np.asarray(g) for g in gts[lo:lo + block]]
        return len(fb), fb[0].ndim == 3, pack(fb), pack(gb)

    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(1)   # packs block k+1 on the host while the device works on block k
    nxt = pool.submit(prepare, 0)
    try:
        for lo in range(0, n, block):
            B, bgr, rec_f, rec_g = nxt.result()
            if lo + block < n:
                nxt = pool.submit(prepare, lo + block)
            img = letterbox_dev(rec_f, B, 3 if bgr else 1)                 # img_lb  (eval_bagls.py:153)
            gt = letterbox_dev(rec_g, B, 1)                                 # gt_lb   (:154)
            if bgr:
                gray = torch.empty((B, S, S), dtype=torch.uint8, device=dev)
                unet_model.bgr2gray_dev(img, B, S, S, gray)                 # gray_lb (:155)
            else:
                gray = img
            boxes = [None] * B
            if detector is not None:
                if native:
                    bgr_in = img if bgr else gray[..., None].expand(B, S, S, 3).contiguous()
                    unet_model.sync()
                    best = detector.model.detect_dev(bgr_in, B, S, S, detector.conf)
                for i in range(B):
                    detector.reset()                                        # :164-166: BAGLS frames are not a sequence
                    if native:
                        boxes[i] = detector.update(best[i:i + 1, :4], best[i:i + 1, 4], S, S) if best[i, 4] >= 0 else detector.update(None, None, S, S)
                    else:
                        f_host = img[i].cpu().numpy()
                        boxes[i] = detector.detect(f_host if bgr else np.repeat(f_host[..., None], 3, axis=-1))
            mask_u = torch.empty((B, S, S), dtype=torch.uint8, device=dev)
            area_u = torch.empty(B, dtype=torch.int32, device=dev)
            unet_model.segment_dev(gray, B, S, S, area_u, mask_dev=mask_u)
            st_u = torch.empty((B, 3), dtype=torch.int32, device=dev)
            check(lib().og_mask_stats_dev(unet_model._h, ptr(mask_u), ptr(gt), B, S, S, None, ptr(st_u)), "og_mask_stats_dev")
            res = np.zeros((B, 10), np.int64)
            if detector is not None:
                nb = np.array([normalize_box(b, S, S) for b in boxes], np.int32)          # python-slice semantics of mask[y1:y2, x1:x2]
                d_nb = up(nb)
                st_yu = torch.empty((B, 3), dtype=torch.int32, device=dev)
                check(lib().og_mask_stats_dev(unet_model._h, ptr(mask_u), ptr(gt), B, S, S, ptr(d_nb), ptr(st_yu)), "og_mask_stats_dev")
                clamp = np.array([(-1, -1, -1, -1) if b is None else (max(0, min(S, int(b[0]))), max(0, min(S, int(b[1]))),
                                                                      max(0, min(S, int(b[2]))), max(0, min(S, int(b[3])))) for b in boxes], np.int32)
                d_cl = up(clamp)                                            # :181-186 clamp for the TP/FP bookkeeping
                gt_in = torch.empty(B, dtype=torch.int32, device=dev)
                unet_model.mask_area_dev(gt, B, S, S, d_cl, gt_in)
                # yolo-crop+unet (:89-112, :209-222): optional crop_pad, crop, letterbox NEAREST, U-Net, project back, paste
                cb = np.full((B, 4), -1, np.int32)
                geo = np.zeros((B, 4), np.int32)
                for i, b in enumerate(boxes):
                    if b is None:
                        continue
                    x1, y1, x2, y2 = (int(v) for v in b)
                    if crop_pad:
                        x1, y1, x2, y2 = max(0, x1 - crop_pad), max(0, y1 - crop_pad), min(S, x2 + crop_pad), min(S, y2 + crop_pad)
                    x1, y1, x2, y2 = normalize_box((x1, y1, x2, y2), S, S)
                    if x2 - x1 <= 0 or y2 - y1 <= 0:
                        continue
                    cb[i] = (x1, y1, x2, y2)
                    geo[i] = letterbox_geometry(y2 - y1, x2 - x1, NET_SIZE)
                d_cb, d_geo = up(cb), up(geo)
                tiles = torch.empty((B, NET_SIZE, NET_SIZE), dtype=torch.uint8, device=dev)
                tmask = torch.empty_like(tiles)
                mask_c = torch.empty((B, S, S), dtype=torch.uint8, device=dev)
                unet_model.sync()
                check(lib().og_unet_segment_crops_u8_dev(cm._h, ptr(gray), B, S, S, ptr(d_cb), ptr(d_geo), NET_SIZE, 0.5, ptr(tiles), ptr(tmask),
                                                         ptr(mask_c)), "og_unet_segment_crops_u8_dev")
                cm.sync()
                st_c = torch.empty((B, 3), dtype=torch.int32, device=dev)
                check(lib().og_mask_stats_dev(unet_model._h, ptr(mask_c), ptr(gt), B, S, S, None, ptr(st_c)), "og_mask_stats_dev")
                unet_model.sync()
                yu, c, gin = st_yu.cpu().numpy(), st_c.cpu().numpy(), gt_in.cpu().numpy()
                res[:, 3:5] = yu[:, :2]
                res[:, 5:7] = c[:, :2]
                res[:, 7] = [b is not None for b in boxes]
                res[:, 8] = gin > 0
            unet_model.sync()
            u = st_u.cpu().numpy()
            res[:, 0:3] = u
            res[:, 9] = u[:, 2] > 0
            out[lo:lo + B] = res
    finally:
        pool.shutdown()
    return out


def agg_from_counts(counts: np.ndarray, has_detector: bool, has_crop: bool = True):
    """``(agg, det_stats)`` in the reference's structure (eval_bagls.py:131-135,230) from the per-frame count rows."""
    agg = {p: {"dice": [], "iou": [], "n_det": 0, "n_total": 0} for p in PIPELINES}
    det_stats = {"tp": 0, "fp": 0, "fn": 0, "n_pos_gt": 0}
    for r in counts:
        tp_u, np_u, ng, tp_yu, np_yu, tp_c, np_c, detected, det_tp, gt_pos = (int(v) for v in r)
        rows = [("unet-only", tp_u, np_u)]
        if has_detector:
            rows.append(("yolo+unet", tp_yu, np_yu))
            if has_crop:
                rows.append(("yolo-crop+unet", tp_c, np_c))
            det_stats["n_pos_gt"] += gt_pos
            if detected:
                det_stats["tp" if det_tp else "fp"] += 1
            elif gt_pos:
                det_stats["fn"] += 1
        for pipe, tp, npred in rows:
            d, j = metrics_from_counts(tp, npred, ng)
            agg[pipe]["dice"].append(d)
            agg[pipe]["iou"].append(j)
            agg[pipe]["n_total"] += 1
            agg[pipe]["n_det"] += int(detected and pipe != "unet-only")
    return agg, det_stats


def evaluate_device(frames, gts, unet_model, detector=None, crop_model=None, canvas: int = NET_SIZE, crop_pad: int = 0):
    """`scripts/eval_bagls.py:evaluate` on the device: ``(agg, det_stats)``."""
    counts = evaluate_counts_device(frames, gts, unet_model, detector, crop_model, canvas, crop_pad)
    return agg_from_counts(counts, detector is not None, True)
