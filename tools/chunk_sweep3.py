"""frames/s over micro-batch sizes for two split-K settings (default lanes rule), both precisions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(512)).cuda(); area = torch.zeros(512, dtype=torch.int32, device="cuda")
for prec in (0, 1):
    m.set_option("precision", prec)
    for slots, div in ((2, 4), (1, 2)):
        m.set_option("splitk_slots", slots); m.set_option("splitk_div", div)
        row = []
        for chunk in (1, 2, 4, 8, 16, 32):
            m.set_chunk(chunk); n = 256 if chunk < 8 else 512
            m.segment_dev(fr, n, 256, 256, area); m.sync()
            best = 0
            for _ in range(3):
                t0 = time.perf_counter(); m.segment_dev(fr, n, 256, 256, area); m.sync(); best = max(best, n / (time.perf_counter() - t0))
            row.append(f"{chunk}:{best:.0f}")
        print(f"precision {prec} slots {slots} div {div}:", " ".join(row), flush=True)
