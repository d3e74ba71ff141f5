"""One-frame-per-call detector latency over the latency-path knobs (C-ABI host entry point, pageable frame in, boxes out)."""
import os, sys, time, json, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector
d = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
fr = np.random.RandomState(0).randint(0, 256, (200, 256, 256, 3), dtype=np.uint8)
def run(**kw):
    for k, v in kw.items(): d.set_option(k, v)
    for i in range(30): d.detect_batch(fr[i:i + 1], 0.25)
    t0 = time.perf_counter()
    for i in range(200): d.detect_batch(fr[i:i + 1], 0.25)
    return round((time.perf_counter() - t0) / 200 * 1e3, 3)
base = dict(latency_batch=1, head_fused=1, latency_nt1=1, splitk_max=8, splitk_min_steps=3, splitk_slots=1, splitk_div=2)
print("base", base, run(**base), flush=True)
for k, vals in [("latency_nt1", [0]), ("splitk_min_steps", [9, 1]), ("splitk_max", [2, 4, 16, 32]), ("splitk_div", [1, 4]), ("splitk_slots", [2]),
                ("head_fused", [0]), ("latency_batch", [0])]:
    for v in vals:
        kw = dict(base); kw[k] = v
        print(k, v, run(**kw), flush=True)
