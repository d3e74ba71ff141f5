"""Compare the detector's named activations of frame 0 between a large batch (occupancy kernels) and a small one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector
d = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
f = np.random.RandomState(11).randint(0, 256, (192, 256, 256, 3), dtype=np.uint8)
names = ["model.0", "model.1", "model.2", "model.3", "model.4", "model.5", "model.6", "model.7", "model.8", "model.9", "model.12", "model.15",
         "model.16", "model.18", "model.19", "model.21", "box0", "cls0", "box1", "cls1", "box2", "cls2"]
d.detect_batch(f[:2]); small = {n: d.activation(n, 1) for n in names}
d.detect_batch(f); big = {n: d.activation(n, 1) for n in names}
for n in names:
    x, y = big[n], small[n]
    bad = np.argwhere(x != y)
    print(f"{n:10s} {str(x.shape):18s} differing {len(bad):7d}  max |diff| {float(np.abs(x - y).max()):.4g}")
    if len(bad):
        print("   first:", bad[:5].tolist(), " channels:", np.unique(bad[:, 1])[:16].tolist(), " rows:", np.unique(bad[:, 2])[:12].tolist(), " cols:", np.unique(bad[:, 3])[:12].tolist())
        break
