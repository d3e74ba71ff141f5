"""Sweep of occ_min_pct / lanes / prio at 1-4 frames per chain, both precisions."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(256)).cuda(); area = torch.zeros(256, dtype=torch.int32, device="cuda")
for prec in (0, 1):
    m.set_option("precision", prec)
    for occ, lanes in itertools.product((0, 25, 50, 100, 200), (2, 3)):
        m.set_option("occ_min_pct", occ); m.set_option("lanes", lanes)
        row = []
        for chunk in (1, 2, 4):
            m.set_chunk(chunk)
            m.segment_dev(fr, 256, 256, 256, area); m.sync()
            best = 0
            for _ in range(3):
                t0 = time.perf_counter(); m.segment_dev(fr, 256, 256, 256, area); m.sync(); best = max(best, 256 / (time.perf_counter() - t0))
            row.append(f"{chunk}:{best:.0f}")
        print(f"precision {prec} occ_min_pct {occ:3d} lanes {lanes}:", " ".join(row), flush=True)
