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
