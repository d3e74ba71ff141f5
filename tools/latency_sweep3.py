"""One frame per kernel chain, three lanes (bench.py's latency_mode), over the split-K knobs; f32 path."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(256)).cuda(); area = torch.zeros(256, dtype=torch.int32, device="cuda")
m.set_chunk(1)
def run(**kw):
    for k, v in kw.items(): m.set_option(k, v)
    m.segment_dev(fr, 256, 256, 256, area); m.sync()
    best = 0
    for _ in range(4):
        t0 = time.perf_counter(); m.segment_dev(fr, 256, 256, 256, area); m.sync(); best = max(best, 256 / (time.perf_counter() - t0))
    return round(best)
base = dict(precision=0, lanes=3, splitk_nt1=1, splitk_min_steps=9, splitk_slots=1, splitk_div=2, occ_min_pct=0)
print("base", base, run(**base), flush=True)
for k, vals in [("splitk_nt1", [0]), ("splitk_min_steps", [3]), ("splitk_slots", [2]), ("splitk_div", [1, 3, 4]), ("occ_min_pct", [25, 50, 100]), ("lanes", [2])]:
    for v in vals:
        kw = dict(base); kw[k] = v
        print("  ", k, v, run(**kw), flush=True)
for combo in [dict(splitk_min_steps=3, splitk_div=1), dict(splitk_min_steps=3, splitk_div=4), dict(splitk_div=4, occ_min_pct=50), dict(splitk_div=1, occ_min_pct=50)]:
    kw = dict(base); kw.update(combo)
    print("  ", combo, run(**kw), flush=True)
