"""Throughput of the YOLO-Crop+UNet pipeline (BASELINE config C5, one GPU's share) with the BAGLS convention.

Per frame (`scripts/eval_bagls.py:153-166,195`, `scripts/eval_girafe.py:127-159`): frame already letterboxed to the
256x256 canvas -> YOLOv8n -> stateless TemporalDetector (reset before every frame) -> padded box -> crop ->
NEAREST letterbox to 256x256 -> U-Net -> threshold -> NEAREST back-projection -> paste into a zero frame.
Everything between the resident BGR frames and the resident full-frame masks runs on the device; the host only
turns 5 floats per frame into a box + letterbox geometry.  Random-init detector weights (the reference's are
absent): speed only.
"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd._lib import lib, ptr, check
from openglottal_amd.yolo import YoloV8Detector

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = 256
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-2.5)); m.to("cuda:0").eval()
m.set_chunk(64)
m.set_option("precision", int(os.environ.get("OG_PRECISION", "0")))   # 1: the opt-in split-precision kernels
y = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
bgr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (F, S, S, 3), dtype=np.uint8)).cuda()
gray = torch.empty((F, S, S), dtype=torch.uint8, device="cuda")
best = torch.empty((F, 5), dtype=torch.float32, device="cuda")
boxes = torch.empty((F, 4), dtype=torch.int32, device="cuda")
geom = torch.empty((F, 4), dtype=torch.int32, device="cuda")
tiles = torch.empty((F, S, S), dtype=torch.uint8, device="cuda")
tmask = torch.empty((F, S, S), dtype=torch.uint8, device="cuda")
masks = torch.empty((F, S, S), dtype=torch.uint8, device="cuda")
YC = 256


def yolo_chunk(k):
    b0 = k * YC
    check(lib().og_yolo_detect_u8_dev(y._h, ptr(bgr[b0:]), min(YC, F - b0), S, S, 0.25, ptr(best[b0:]), None), "yolo")


def host_boxes(bh):
    """Stateless detector (reset per frame) + letterbox geometry for a chunk of best boxes."""
    n = len(bh)
    bx = np.full((n, 4), -1, np.int32)
    ge = np.zeros((n, 4), np.int32)
    td = og.TemporalDetector(lambda f, c: None)
    for i in range(n):
        td.reset()
        b = td.update(bh[i:i + 1, :4], bh[i:i + 1, 4], S, S) if bh[i, 4] >= 0 else None
        if b is None:
            continue
        x1, y1, x2, y2 = (max(0, min(S, int(v))) for v in b)
        h, w = y2 - y1, x2 - x1
        if h <= 0 or w <= 0:
            continue
        sc = S / max(h, w)
        nh, nw = int(round(h * sc)), int(round(w * sc))
        bx[i] = (x1, y1, x2, y2)
        ge[i] = ((S - nh) // 2, (S - nw) // 2, nh, nw)
    return bx, ge


def run():
    check(lib().og_bgr2gray_dev(m._h, ptr(bgr), F, S, S, ptr(gray)), "gray")
    nk = (F + YC - 1) // YC
    yolo_chunk(0)
    ndet = 0
    for k in range(nk):
        check(lib().og_yolo_sync(y._h), "sync")
        if k + 1 < nk:
            yolo_chunk(k + 1)
        b0, b1 = k * YC, min(F, (k + 1) * YC)
        bh = np.empty((b1 - b0, 5), np.float32)
        check(lib().og_memcpy_d2h(ptr(bh), ptr(best[b0:]), bh.nbytes), "d2h")
        bx, ge = host_boxes(bh)
        ndet += int((bx[:, 0] >= 0).sum())
        check(lib().og_memcpy_h2d(ptr(boxes[b0:]), ptr(bx), bx.nbytes), "h2d")
        check(lib().og_memcpy_h2d(ptr(geom[b0:]), ptr(ge), ge.nbytes), "h2d")
        check(lib().og_unet_segment_crops_u8_dev(m._h, ptr(gray[b0:]), b1 - b0, S, S, ptr(boxes[b0:]), ptr(geom[b0:]), S, 0.5,
                                                 ptr(tiles[b0:]), ptr(tmask[b0:]), ptr(masks[b0:])), "crops")
    m.sync()
    return ndet


ndet = run()
t0 = time.perf_counter(); n = 3
for _ in range(n):
    run()
el = time.perf_counter() - t0
print(json.dumps({"precision": int(os.environ.get("OG_PRECISION", "0")), "pipeline": "YOLO-Crop+UNet (stateless detector, crop->256->project back), 256x256 canvas, 1xMI355X",
                  "frames": F, "fps": round(n * F / el, 1), "frames_with_box": ndet,
                  "mask_pixels_head": [int(v) for v in (masks[:8] > 0).sum(dim=(1, 2)).tolist()]}))
