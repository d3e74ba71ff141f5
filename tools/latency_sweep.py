"""Sweep of the split-K knobs at one frame per kernel chain (three lanes), both precisions."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(256)).cuda(); area = torch.zeros(256, dtype=torch.int32, device="cuda")
m.set_chunk(1); m.set_option("lanes", int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for prec in (0, 1):
    m.set_option("precision", prec)
    for slots, div, ms in itertools.product((1, 2, 3, 4), (2, 4, 8), (3, 9)):
        m.set_option("splitk_slots", slots); m.set_option("splitk_div", div); m.set_option("splitk_min_steps", ms)
        m.segment_dev(fr, 256, 256, 256, area); m.sync()
        best = 0
        for _ in range(3):
            t0 = time.perf_counter(); m.segment_dev(fr, 256, 256, 256, area); m.sync(); best = max(best, 256 / (time.perf_counter() - t0))
        print(f"precision {prec} slots {slots} div {div} min_steps {ms}: {best:7.0f} fps", flush=True)
