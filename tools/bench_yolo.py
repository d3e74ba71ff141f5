"""Detector-only throughput (frames resident in HBM), for rocprofv3 --stats."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openglottal_amd import synth
from openglottal_amd._lib import lib, ptr, check
from openglottal_amd.yolo import YoloV8Detector
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
y = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
bgr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (F, 256, 256, 3), dtype=np.uint8)).cuda()
best = torch.empty((F, 5), dtype=torch.float32, device="cuda")
def run():
    for b0 in range(0, F, C):
        check(lib().og_yolo_detect_u8_dev(y._h, ptr(bgr[b0:]), min(C, F - b0), 256, 256, 0.25, ptr(best[b0:]), None), "yolo")
    check(lib().og_yolo_sync(y._h), "sync")
run()
t0 = time.perf_counter()
for _ in range(5):
    run()
print(json.dumps({"yolo_fps": round(5 * F / (time.perf_counter() - t0), 1), "frames_per_launch": C}))
