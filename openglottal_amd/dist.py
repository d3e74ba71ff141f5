"""Frame sharding across the GPUs of one node + the area-waveform all-gather.

Frames are independent units for the U-Net (features.py:234-238 has no
cross-frame state), so the path shards with NO data-path collective; the only
exchange is one all-gather of per-frame ``int32`` areas (5 KB per rank at 10 k
frames / 8 GPUs) so every rank holds the full glottal-area waveform for
``_kinematic_features``.  One process per GPU, ``torch.distributed`` with the
``nccl`` backend (= RCCL over xGMI on ROCm); ``gloo`` on CPU for tests.
"""

from __future__ import annotations

import os

import numpy as np


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block sharding (keeps waveform order): rank r owns [lo, hi)."""
    per = -(-n_frames // world)
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def env_rank_world() -> tuple[int, int, int]:
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def all_gather_areas(local, n_frames: int, group=None, force: bool = False):
    """All-gather ragged per-rank area slices into the full ``[n_frames]`` waveform.

    ``local``: this rank's areas (torch int32 tensor on the collective's device, or numpy).
    Slices are padded to ceil(n/world) with -1 so ONE fixed-size all-gather suffices.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        t = torch.as_tensor(local)
        return t[:n_frames]
    world = dist.get_world_size(group)
    per = -(-n_frames // world)
    t = torch.as_tensor(local).to(torch.int32)
    pad = torch.full((per,), -1, dtype=torch.int32, device=t.device)
    pad[: t.numel()] = t
    out = torch.empty(world * per, dtype=torch.int32, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_frames, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts)


def sharded_area_waveform(frames_gray: np.ndarray, model, rank: int, world: int, device=None, group=None) -> np.ndarray:
    """U-Net-only area waveform of a whole video with frames sharded over ``world`` ranks."""
    import torch

    n = len(frames_gray)
    lo, hi = shard_range(n, rank, world)
    if hi > lo:
        _, area, _ = model.segment(frames_gray[lo:hi], want_mask=False)
    else:
        area = np.zeros(0, np.int32)
    t = torch.from_numpy(np.ascontiguousarray(area))
    if device is not None:
        t = t.to(device)
    return all_gather_areas(t, n, group).cpu().numpy()


def all_gather_rows(local, n_frames: int, width: int, group=None, force: bool = False):
    """All-gather ragged per-rank ``[n_local, width]`` float32 rows (e.g. best boxes x1,y1,x2,y2,conf) into
    ``[n_frames, width]``; slices are padded to ceil(n/world) rows so one fixed-size collective suffices."""
    import torch
    import torch.distributed as dist

    t = torch.as_tensor(local, dtype=torch.float32).reshape(-1, width)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return t[:n_frames]
    world = dist.get_world_size(group)
    per = -(-n_frames // world)
    pad = torch.full((per, width), -1.0, dtype=torch.float32, device=t.device)
    pad[: t.shape[0]] = t
    out = torch.empty(world * per, width, dtype=torch.float32, device=t.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_frames, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts)


def sharded_gated_area_waveform(frames_bgr, detect_batch, make_detector, model, rank: int, world: int, device=None, group=None,
                                conf: float = 0.25):
    """Detection-gated area waveform (features.py:234-245 with a detector) with frames sharded over ranks.

    SURVEY §8e: (1) every rank runs the per-frame-independent YOLO network on ITS frames and the per-frame
    best boxes (5 floats) are all-gathered; (2) every rank replays the O(N) sequential TemporalDetector state
    machine over the whole video (deterministic, so all ranks agree); (3) every rank segments its own frames
    with its slice of the boxes; (4) the areas are all-gathered.

    ``detect_batch(frames [n,H,W,3], conf) -> best [n,5]`` in ORIGINAL-frame pixels (conf = -1: no detection) —
    pass ``YoloV8Detector.detect_frames`` (letterboxes to the model's imgsz and scales the boxes back, like the
    per-frame call), not the network-resolution ``detect_batch``;
    ``make_detector()`` returns a fresh ``TemporalDetector`` (only its ``update`` is used).
    """
    import torch

    from .utils import normalize_box

    n = len(frames_bgr)
    lo, hi = shard_range(n, rank, world)
    H, W = frames_bgr[0].shape[:2]
    mine = np.stack(frames_bgr[lo:hi]) if hi > lo else np.zeros((0, H, W, 3), np.uint8)
    best_local = detect_batch(mine, conf) if hi > lo else np.zeros((0, 5), np.float32)
    t = torch.from_numpy(np.ascontiguousarray(best_local, dtype=np.float32))
    if device is not None:
        t = t.to(device)
    best = all_gather_rows(t, n, 5, group).cpu().numpy()
    det = make_detector()
    det.reset()
    boxes = np.empty((n, 4), np.int32)
    for i in range(n):
        b = det.update(best[i:i + 1, :4], best[i:i + 1, 4], W, H) if best[i, 4] >= 0 else det.update(None, None, W, H)
        boxes[i] = normalize_box(b, W, H)
    if hi > lo:
        # exactly what the single-process `area_waveform` does with a block: BGR frames + boxes to the streaming engine
        # (BGR->gray of features.py:235 on the device, same kernels, same form)
        _, area = model.segment_stream(mine, boxes=boxes[lo:hi])
    else:
        area = np.zeros(0, np.int32)
    ta = torch.from_numpy(np.ascontiguousarray(area))
    if device is not None:
        ta = ta.to(device)
    return all_gather_areas(ta, n, group).cpu().numpy(), boxes


def sharded_eval_counts(n_frames: int, count_fn, rank: int, world: int, device=None, group=None, width: int = 10) -> np.ndarray:
    """BASELINE config C5 (BAGLS evaluation sharded over the GPUs of a node): the detector is reset before every frame
    (scripts/eval_bagls.py:164-166), so frames are fully independent — each rank evaluates ITS contiguous shard
    (``count_fn(lo, hi) -> int [hi-lo, width]``, e.g. ``evaluate.evaluate_counts_device`` on ``frames[lo:hi]``) and ONE
    all-gather of the per-frame integer count rows gives every rank the whole table (40 bytes per frame; the masks never
    leave their GPU).  Counts stay below 2^24, so the float32 rows of ``all_gather_rows`` carry them exactly.  ``width`` is
    stated by the caller (not inferred) so that a rank with an empty shard posts a collective of the same shape."""
    import torch

    lo, hi = shard_range(n_frames, rank, world)
    local = np.asarray(count_fn(lo, hi)).reshape(-1, width) if hi > lo else np.zeros((0, width), np.int64)
    assert int(local.max(initial=0)) < (1 << 24)
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
    if device is not None:
        t = t.to(device)
    return np.rint(all_gather_rows(t, n_frames, width, group).cpu().numpy()).astype(np.int64)
