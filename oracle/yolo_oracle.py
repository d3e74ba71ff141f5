"""CPU oracle for the YOLOv8 detector path.  TEST INFRASTRUCTURE (see unet_oracle.py header).

Parity status: **UNPINNED**.  The network is third-party: `openglottal/models/detector.py:6,31,58`
calls `ultralytics.YOLO`, dependency `ultralytics>=8.0` (`pyproject.toml:25`; no pin, no lockfile,
not vendored, not installed here), and the reference holds no test vector at that boundary.  This
file restates ultralytics' PUBLISHED architecture and post-processing (cfg/models/v8/yolov8.yaml;
nn/modules/{conv,block,head}.py: Conv = Conv2d(no bias)+BatchNorm2d(eps 1e-3)+SiLU, C2f,
Bottleneck, SPPF, Detect with DFL; utils/ops.non_max_suppression) in plain torch-CPU ops, written
independently of the HIP implementation, so that the two can at least be checked against each
other.  Every number here must be re-verified against an ultralytics checkout before being
relied on as "reference behaviour".
"""

from __future__ import annotations

import numpy as np

BN_EPS = 1e-3


def _t(sd):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _conv(x, sd, p, s=1):
    import torch.nn.functional as F

    w = sd[p + ".conv.weight"]
    k = w.shape[-1]
    if p + ".bn.weight" in sd:
        x = F.conv2d(x, w, None, s, k // 2)
        x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                         False, 0.03, BN_EPS)
    else:
        x = F.conv2d(x, w, sd[p + ".conv.bias"], s, k // 2)
    return F.silu(x)


def _c2f(x, sd, p, shortcut, taps=None):
    import torch

    y = list(_conv(x, sd, p + ".cv1").chunk(2, 1))
    j = 0
    while f"{p}.m.{j}.cv1.conv.weight" in sd:
        z = _conv(_conv(y[-1], sd, f"{p}.m.{j}.cv1"), sd, f"{p}.m.{j}.cv2")
        y.append(y[-1] + z if shortcut else z)
        j += 1
    out = _conv(torch.cat(y, 1), sd, p + ".cv2")
    if taps is not None:
        taps[p] = out
    return out


def forward(sd_np: dict, x, taps: dict | None = None):
    """x: torch f32 [B,3,H,W] RGB in [0,1] → (pred [B,4+nc,A] xywh·stride + sigmoid cls, raw per-level lists)."""
    import torch
    import torch.nn.functional as F

    sd = _t(sd_np)
    T = taps if taps is not None else {}

    def keep(name, v):
        T[name] = v
        return v

    x0 = keep("model.0", _conv(x, sd, "model.0", 2))
    x1 = keep("model.1", _conv(x0, sd, "model.1", 2))
    x2 = _c2f(x1, sd, "model.2", True, T)
    x3 = keep("model.3", _conv(x2, sd, "model.3", 2))
    x4 = _c2f(x3, sd, "model.4", True, T)
    x5 = keep("model.5", _conv(x4, sd, "model.5", 2))
    x6 = _c2f(x5, sd, "model.6", True, T)
    x7 = keep("model.7", _conv(x6, sd, "model.7", 2))
    x8 = _c2f(x7, sd, "model.8", True, T)
    s = _conv(x8, sd, "model.9.cv1")
    y1 = F.max_pool2d(s, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    x9 = keep("model.9", _conv(torch.cat([s, y1, y2, y3], 1), sd, "model.9.cv2"))
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode="nearest")
    x12 = _c2f(torch.cat([up(x9), x6], 1), sd, "model.12", False, T)
    x15 = _c2f(torch.cat([up(x12), x4], 1), sd, "model.15", False, T)
    x16 = keep("model.16", _conv(x15, sd, "model.16", 2))
    x18 = _c2f(torch.cat([x16, x12], 1), sd, "model.18", False, T)
    x19 = keep("model.19", _conv(x18, sd, "model.19", 2))
    x21 = _c2f(torch.cat([x19, x9], 1), sd, "model.21", False, T)
    feats = [x15, x18, x21]
    B = x.shape[0]
    outs, anchors, strides = [], [], []
    for l, f in enumerate(feats):
        pb, pc = f"model.22.cv2.{l}", f"model.22.cv3.{l}"
        box = F.conv2d(_conv(_conv(f, sd, pb + ".0"), sd, pb + ".1"), sd[pb + ".2.weight"], sd[pb + ".2.bias"])
        cls = F.conv2d(_conv(_conv(f, sd, pc + ".0"), sd, pc + ".1"), sd[pc + ".2.weight"], sd[pc + ".2.bias"])
        T[f"box{l}"], T[f"cls{l}"] = box, cls
        h, w = f.shape[-2:]
        stride = x.shape[-1] / w
        sy, sx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
        anchors.append(torch.stack((sx, sy), -1).view(-1, 2))
        strides.append(torch.full((h * w, 1), stride, dtype=torch.float32))
        outs.append(torch.cat([box, cls], 1).view(B, box.shape[1] + cls.shape[1], -1))
    xc = torch.cat(outs, 2)
    nb = 64
    box, cls = xc[:, :nb], xc[:, nb:]
    anc = torch.cat(anchors).transpose(0, 1)[None]     # [1,2,A]
    st = torch.cat(strides).transpose(0, 1)[None]      # [1,1,A]
    A = box.shape[-1]
    dist = (box.view(B, 4, 16, A).transpose(2, 1).softmax(1) * torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)).sum(1)  # DFL
    lt, rb = dist.chunk(2, 1)
    x1y1, x2y2 = anc - lt, anc + rb
    dbox = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], 1) * st
    return torch.cat([dbox, cls.sigmoid()], 1), T


def preprocess_bgr(frames_bgr: np.ndarray):
    """[B,H,W,3] u8 BGR (already at network size) → torch [B,3,H,W] RGB float/255 (predictor.preprocess)."""
    import torch

    rgb = np.ascontiguousarray(frames_bgr[..., ::-1].transpose(0, 3, 1, 2))
    return torch.from_numpy(rgb).float() / 255


def xywh2xyxy(b):
    out = np.empty_like(b)
    out[..., 0] = b[..., 0] - b[..., 2] / 2
    out[..., 1] = b[..., 1] - b[..., 3] / 2
    out[..., 2] = b[..., 0] + b[..., 2] / 2
    out[..., 3] = b[..., 1] + b[..., 3] / 2
    return out


def candidates(sd_np, frames_bgr):
    """→ [B,A,5] xyxy (clipped to the frame) + conf, single-class."""
    import torch

    with torch.no_grad():
        pred, _ = forward(sd_np, preprocess_bgr(frames_bgr))
    p = pred.numpy().transpose(0, 2, 1)  # [B,A,5]
    H, W = frames_bgr.shape[1:3]
    xyxy = xywh2xyxy(p[..., :4].astype(np.float32))
    xyxy[..., [0, 2]] = xyxy[..., [0, 2]].clip(0, W)
    xyxy[..., [1, 3]] = xyxy[..., [1, 3]].clip(0, H)
    return np.concatenate([xyxy, p[..., 4:5]], -1).astype(np.float32)


def nms(xyxy: np.ndarray, conf: np.ndarray, conf_thres=0.25, iou_thres=0.7, max_det=300):
    """Greedy NMS as torchvision.ops.nms / ultralytics non_max_suppression (single class)."""
    keep_mask = conf > conf_thres
    idx = np.flatnonzero(keep_mask)
    idx = idx[np.argsort(-conf[idx], kind="stable")]
    keep = []
    area = (xyxy[:, 2] - xyxy[:, 0]) * (xyxy[:, 3] - xyxy[:, 1])
    while idx.size and len(keep) < max_det:
        i = idx[0]
        keep.append(i)
        r = idx[1:]
        xx1, yy1 = np.maximum(xyxy[i, 0], xyxy[r, 0]), np.maximum(xyxy[i, 1], xyxy[r, 1])
        xx2, yy2 = np.minimum(xyxy[i, 2], xyxy[r, 2]), np.minimum(xyxy[i, 3], xyxy[r, 3])
        inter = np.clip(xx2 - xx1, 0, None) * np.clip(yy2 - yy1, 0, None)
        iou = inter / (area[i] + area[r] - inter + 1e-12)
        idx = r[iou <= iou_thres]
    return np.array(keep, dtype=np.int64)
