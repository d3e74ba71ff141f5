"""U-Net at one frame per call (utils.unet_segment_frame, the reference's per-frame pattern): wall time per call; under
rocprofv3 the per-kernel split (tools/trace_one_frame.py with the head kernel as the frame marker)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import unet_segment_frame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
for a in sys.argv[2:]:
    k, v = a.split("=")
    m.set_option(k, int(v))
gray = synth.bulk_gray_frames(n)
for g in gray[:20]: unet_segment_frame(g, m, "cuda:0")
t0 = time.perf_counter()
for g in gray: unet_segment_frame(g, m, "cuda:0")
t = (time.perf_counter() - t0) / n
print(json.dumps({"unet_segment_frame_ms": round(t * 1e3, 3), "options": sys.argv[2:]}))
