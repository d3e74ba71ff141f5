"""Throughput of the detection-gated pipeline (BASELINE config C3: YOLO+UNet, pipeline=unet) on one GPU.

Per video: YOLOv8n on every BGR frame (device, batched) -> per-frame best box -> TemporalDetector state
machine on the host (sequential, O(1)/frame) -> U-Net segment fused with the box-gated area count.
Frames are resident in HBM; random-init detector weights (the reference's are absent), so the numbers
measure speed, not detection quality.
"""
import os, sys, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd._lib import lib, ptr, check
from openglottal_amd.yolo import YoloV8Detector
from openglottal_amd.utils import normalize_box

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-2.5)); m.to("cuda:0").eval()
m.set_chunk(64)
m.set_option("precision", int(os.environ.get("OG_PRECISION", "0")))   # 1: the opt-in split-precision kernels
y = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
bgr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (F, 256, 256, 3), dtype=np.uint8)).cuda()
gray = torch.empty((F, 256, 256), dtype=torch.uint8, device="cuda")
best = torch.empty((F, 5), dtype=torch.float32, device="cuda")
boxes = torch.empty((F, 4), dtype=torch.int32, device="cuda")
area = torch.zeros(F, dtype=torch.int32, device="cuda")

YC = 256  # detector micro-batch

def yolo_chunk(k):
    b0 = k * YC
    check(lib().og_yolo_detect_u8_dev(y._h, ptr(bgr[b0:]), min(YC, F - b0), 256, 256, 0.25, ptr(best[b0:]), None), "yolo")

def run():
    """Pipelined: detector chunk k+1 runs on its stream while the U-Net chain of chunk k runs on the U-Net streams;
    the host replays the O(1)/frame temporal state machine for chunk k in between."""
    td = og.TemporalDetector(lambda f, c: None)
    check(lib().og_bgr2gray_dev(m._h, ptr(bgr), F, 256, 256, ptr(gray)), "gray")
    nk = (F + YC - 1) // YC
    yolo_chunk(0)
    out = np.empty((F, 4), np.int32)
    for k in range(nk):
        check(lib().og_yolo_sync(y._h), "sync")          # chunk k's boxes are on the device
        if k + 1 < nk:
            yolo_chunk(k + 1)                              # overlaps with the U-Net launches below
        b0, b1 = k * YC, min(F, (k + 1) * YC)
        bh = np.empty((b1 - b0, 5), np.float32)
        check(lib().og_memcpy_d2h(ptr(bh), ptr(best[b0:]), bh.nbytes), "d2h")
        for i in range(b1 - b0):
            b = td.update(bh[i:i + 1, :4], bh[i:i + 1, 4], 256, 256) if bh[i, 4] >= 0 else td.update(None, None, 256, 256)
            out[b0 + i] = normalize_box(b, 256, 256)
        check(lib().og_memcpy_h2d(ptr(boxes[b0:]), ptr(out[b0:b1]), (b1 - b0) * 16), "h2d")
        m.segment_dev(gray[b0:], b1 - b0, 256, 256, area[b0:], boxes_dev=boxes[b0:])
    m.sync()

run()
t0 = time.perf_counter(); n = 3
for _ in range(n):
    run()
el = time.perf_counter() - t0
t1 = time.perf_counter()
for _ in range(n):
    for b0 in range(0, F, 256):
        check(lib().og_yolo_detect_u8_dev(y._h, ptr(bgr[b0:]), min(256, F - b0), 256, 256, 0.25, ptr(best[b0:]), None), "yolo")
    check(lib().og_yolo_sync(y._h), "sync")
ely = time.perf_counter() - t1
print(json.dumps({"precision": int(os.environ.get("OG_PRECISION", "0")), "pipeline": "YOLO+UNet (gated), 256x256, 1xMI355X", "frames": F, "fps": round(n * F / el, 1),
                  "yolo_only_fps": round(n * F / ely, 1), "areas_head": area[:8].tolist()}))
