"""Per-frame latency of the reference-style call pattern: TemporalDetector.detect(frame) and unet_segment_frame(frame), one frame per call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import unet_segment_frame, bgr_to_gray
from openglottal_amd.yolo import YoloV8Detector
d = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
td = og.TemporalDetector(d, conf=0.25)
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=5, head_scale=3.0, head_bias=-2.5)); m.to("cuda:0").eval()
fr = np.random.RandomState(0).randint(0, 256, (200, 256, 256, 3), dtype=np.uint8)
gray = [bgr_to_gray(f) for f in fr]
for f in fr[:20]: td.detect(f)
t0 = time.perf_counter()
for f in fr: td.detect(f)
t_det = (time.perf_counter() - t0) / len(fr)
for g in gray[:20]: unet_segment_frame(g, m, "cuda:0")
t0 = time.perf_counter()
for g in gray: unet_segment_frame(g, m, "cuda:0")
t_seg = (time.perf_counter() - t0) / len(gray)
print(json.dumps({"detect_ms_per_frame": round(t_det * 1e3, 3), "unet_segment_frame_ms": round(t_seg * 1e3, 3),
                  "per_frame_loop_fps": round(1.0 / (t_det + t_seg), 1)}))
