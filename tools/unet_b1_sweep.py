"""unet_segment_frame (one frame per call, host in/out) over the split-K knobs, both precisions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import unet_segment_frame
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
gray = synth.bulk_gray_frames(200)
def run(**kw):
    for k, v in kw.items(): m.set_option(k, v)
    for g in gray[:30]: unet_segment_frame(g, m, "cuda:0")
    t0 = time.perf_counter()
    for g in gray: unet_segment_frame(g, m, "cuda:0")
    return round((time.perf_counter() - t0) / len(gray) * 1e3, 3)
for prec in (0, 1):
    base = dict(precision=prec, splitk_nt1=1, splitk_min_steps=9, splitk_slots=1, splitk_div=2)
    print("precision", prec, "base", run(**base), flush=True)
    for k, vals in [("splitk_nt1", [0]), ("splitk_min_steps", [3]), ("splitk_slots", [2]), ("splitk_div", [1, 4])]:
        for v in vals:
            kw = dict(base); kw[k] = v
            print("  ", k, v, run(**kw), flush=True)
    kw = dict(base); kw.update(splitk_min_steps=3, splitk_slots=2)
    print("   min_steps=3 slots=2", run(**kw), flush=True)
