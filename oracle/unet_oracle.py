"""CPU oracle for the per-frame glottal segmentation path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.
Nothing under ``openglottal_amd/`` imports it; the product path is the HIP
library and fails loudly without it.

Parity status: **pinned**.  Every function below is checked by
``tests/test_oracle_golden.py`` against vectors captured in the build container
by running the reference's own code (``tests/golden/gen_golden.py``).

Two independent restatements of the same arithmetic are kept:

* ``forward_numpy``  – plain numpy float32 (im2col + matmul; explicit BN
  formula).  Shares no code with torch; slow but independent.
* ``forward_torch``  – the same op sequence through ``torch.nn.functional`` on
  CPU, i.e. the very backend kernels (oneDNN) the reference executes.  This is
  the one timed as ``cpu_baseline`` (kind "port") because it is what a user of
  the reference actually runs on host cores.

Reference lines restated (all under /root/reference):
  openglottal/models/unet.py:18-33   DoubleConv  (conv3x3 no-bias → BN → ReLU)×2
  openglottal/models/unet.py:74-88   UNet.forward (pool, bottleneck, ups, cat[skip,up], head)
  openglottal/utils.py:218-241       unet_segment_frame
  openglottal/features.py:234-245    area = sum(mask > 0) / box-gated
"""

from __future__ import annotations

import numpy as np

BN_EPS = 1e-5  # torch.nn.BatchNorm2d default, used by unet.py:25,28


def n_levels(sd: dict) -> int:
    n = 0
    while f"downs.{n}.net.0.weight" in sd:
        n += 1
    return n


# ───────────────────────────── numpy restatement ─────────────────────────────


def _conv3x3_np(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """x [B,Ci,H,W] f32, w [Co,Ci,3,3] f32 → [B,Co,H,W]; pad 1, cross-correlation."""
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    xp = np.zeros((B, Ci, H + 2, W + 2), dtype=np.float32)
    xp[:, :, 1:-1, 1:-1] = x
    cols = np.empty((B, Ci, 3, 3, H, W), dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            cols[:, :, dy, dx] = xp[:, :, dy:dy + H, dx:dx + W]
    cols = cols.reshape(B, Ci * 9, H * W)
    out = np.matmul(w.reshape(Co, Ci * 9).astype(np.float32), cols)  # [B,Co,HW]
    return out.reshape(B, Co, H, W).astype(np.float32)


def _bn_relu_np(x: np.ndarray, sd: dict, p: str) -> np.ndarray:
    # eval-mode BatchNorm2d: (x-μ)/sqrt(σ²+eps)·γ+β, then ReLU (unet.py:25-26)
    g, b = sd[p + ".weight"], sd[p + ".bias"]
    mu, var = sd[p + ".running_mean"], sd[p + ".running_var"]
    inv = (1.0 / np.sqrt(var.astype(np.float32) + np.float32(BN_EPS))).astype(np.float32)
    y = (x - mu[None, :, None, None]) * inv[None, :, None, None] * g[None, :, None, None] + b[None, :, None, None]
    return np.maximum(y, 0).astype(np.float32)


def _double_conv_np(x, sd, p, taps=None):
    a = _bn_relu_np(_conv3x3_np(x, sd[p + ".net.0.weight"]), sd, p + ".net.1")
    if taps is not None:
        taps[p + ".a"] = a
    b = _bn_relu_np(_conv3x3_np(a, sd[p + ".net.3.weight"]), sd, p + ".net.4")
    if taps is not None:
        taps[p + ".b"] = b
    return b


def _maxpool2_np(x):
    B, C, H, W = x.shape
    return x.reshape(B, C, H // 2, 2, W // 2, 2).max(axis=(3, 5))


def _convT2x2_np(x, w, b):
    """ConvTranspose2d(k=2,s=2): out[co,2y+dy,2x+dx] = b[co] + Σci x[ci,y,x]·w[ci,co,dy,dx]."""
    B, Ci, H, W = x.shape
    Co = w.shape[1]
    out = np.empty((B, Co, 2 * H, 2 * W), dtype=np.float32)
    xf = x.reshape(B, Ci, H * W)
    for dy in range(2):
        for dx in range(2):
            o = np.matmul(w[:, :, dy, dx].T.astype(np.float32), xf).reshape(B, Co, H, W)
            out[:, :, dy::2, dx::2] = o + b[None, :, None, None]
    return out


def forward_numpy(sd: dict, x: np.ndarray, taps: dict | None = None) -> np.ndarray:
    """``UNet.forward`` (unet.py:74-88).  x [B,in_ch,H,W] f32 → logits [B,out_ch,H,W]."""
    L = n_levels(sd)
    assert x.shape[2] % (1 << L) == 0 and x.shape[3] % (1 << L) == 0, "bilinear fallback (unet.py:84-85) not restated"
    x = x.astype(np.float32)
    skips = []
    for i in range(L):
        x = _double_conv_np(x, sd, f"downs.{i}", taps)
        skips.append(x)
        x = _maxpool2_np(x)
        if taps is not None:
            taps[f"pool{i}"] = x
    x = _double_conv_np(x, sd, "bottleneck", taps)
    for j in range(L):
        x = _convT2x2_np(x, sd[f"ups.{2 * j}.weight"], sd[f"ups.{2 * j}.bias"])
        if taps is not None:
            taps[f"ups.{2 * j}"] = x
        x = np.concatenate([skips[-(j + 1)], x], axis=1)  # skip FIRST (unet.py:86)
        x = _double_conv_np(x, sd, f"ups.{2 * j + 1}", taps)
    hw = sd["head.weight"]
    out = np.einsum("oc,bchw->bohw", hw[:, :, 0, 0], x).astype(np.float32) + sd["head.bias"][None, :, None, None]
    if taps is not None:
        taps["head"] = out
    return out.astype(np.float32)


# ───────────────────────────── torch-CPU restatement ─────────────────────────


def forward_torch(sd_t: dict, x_t):
    """Same op sequence through torch.nn.functional (CPU).  sd_t: torch tensors."""
    import torch.nn.functional as F

    def dc(x, p):
        for c, n in (("0", "1"), ("3", "4")):
            x = F.conv2d(x, sd_t[f"{p}.net.{c}.weight"], None, 1, 1)
            x = F.batch_norm(x, sd_t[f"{p}.net.{n}.running_mean"], sd_t[f"{p}.net.{n}.running_var"],
                             sd_t[f"{p}.net.{n}.weight"], sd_t[f"{p}.net.{n}.bias"], False, 0.1, BN_EPS)
            x = F.relu(x)
        return x

    L = n_levels(sd_t)
    skips = []
    x = x_t
    for i in range(L):
        x = dc(x, f"downs.{i}")
        skips.append(x)
        x = F.max_pool2d(x, 2, 2)
    x = dc(x, "bottleneck")
    import torch

    for j in range(L):
        x = F.conv_transpose2d(x, sd_t[f"ups.{2 * j}.weight"], sd_t[f"ups.{2 * j}.bias"], 2)
        x = torch.cat([skips[-(j + 1)], x], dim=1)
        x = dc(x, f"ups.{2 * j + 1}")
    return F.conv2d(x, sd_t["head.weight"], sd_t["head.bias"])


# ───────────────────────────── frame-level helpers ───────────────────────────


def _sigmoid32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.float32)
    return (np.float32(1) / (np.float32(1) + np.exp(-x))).astype(np.float32)


def segment_frames(sd: dict, gray: np.ndarray, threshold: float = 0.5, backend: str = "torch"):
    """``unet_segment_frame`` (utils.py:218-241) for a stack of 256×256-class frames.

    gray [B,H,W] u8 with H,W already at network resolution (both resizes in the
    reference are identities there, utils.py:234,239).  Returns
    ``(mask u8 {0,255} [B,H,W], logits f32 [B,H,W])``.
    """
    x = (gray.astype("float32") / 255.0)[:, None]  # utils.py:235
    if backend == "numpy":
        logits = forward_numpy(sd, x)[:, 0]
        prob = _sigmoid32(logits)
    else:
        import torch

        sd_t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
        with torch.no_grad():
            lt = forward_torch(sd_t, torch.from_numpy(x))
            prob = torch.sigmoid(lt)[:, 0].numpy()
            logits = lt[:, 0].numpy()
    mask = (prob > threshold).astype(np.uint8) * 255  # utils.py:241
    return mask, logits


def areas_from_masks(masks: np.ndarray, boxes=None) -> np.ndarray:
    """features.py:238 (full frame) / :241-245 (box-gated; ``None`` box → 0)."""
    out = np.zeros(len(masks), dtype=np.int64)
    for i, m in enumerate(masks):
        if boxes is None:
            out[i] = int(np.sum(m > 0))
        else:
            b = boxes[i]
            if b is None or b[0] < 0:
                out[i] = 0
            else:
                x1, y1, x2, y2 = (int(v) for v in b)
                out[i] = int(np.sum(m[y1:y2, x1:x2] > 0))
    return out
