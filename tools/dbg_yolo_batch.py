import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector
d = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
f = np.random.RandomState(11).randint(0, 256, (192, 256, 256, 3), dtype=np.uint8)
_, big = d.detect_batch(f, 0.25, want_pred=True)
_, big2 = d.detect_batch(f, 0.25, want_pred=True)
print("big run repeatable:", np.array_equal(big, big2), "differing", int((big != big2).sum()))
bad = {}
for lo in range(0, 192, 2):
    _, p = d.detect_batch(f[lo:lo + 2], 0.25, want_pred=True)
    w = np.argwhere(big[lo:lo + 2] != p)
    if len(w):
        bad[lo] = w
print("frames with mismatches:", {k: len(v) for k, v in bad.items()})
for k, v in list(bad.items())[:3]:
    print(k, v[:12].tolist())
