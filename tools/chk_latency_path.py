import sys, numpy as np
sys.path.insert(0, "/root/repo")
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector
d = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
fr = np.random.RandomState(0).randint(0, 256, (6, 256, 256, 3), dtype=np.uint8)
d.set_option("latency_batch", 0)
ref = [d.detect_batch(fr[i:i+1], 0.25, True) for i in range(6)]
d.set_option("latency_batch", 1)
for rep in range(2):
    got = [d.detect_batch(fr[i:i+1], 0.25, True) for i in range(6)]
    for (b0, p0), (b1, p1) in zip(ref, got):
        print("best", np.abs(b0 - b1).max(), "pred", np.abs(p0 - p1).max(), "pred scale", np.abs(p0).max())
for shape in [(96, 160), (320, 256), (32, 32)]:
    f = np.random.RandomState(1).randint(0, 256, (1,) + shape + (3,), dtype=np.uint8)
    d.set_option("latency_batch", 0); b0, p0 = d.detect_batch(f, 0.25, True)
    d.set_option("latency_batch", 1); b1, p1 = d.detect_batch(f, 0.25, True)
    print(shape, np.abs(b0 - b1).max(), np.abs(p0 - p1).max())
