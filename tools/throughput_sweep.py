import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
N = 768
fr = torch.from_numpy(synth.bulk_gray_frames(N)).cuda(); area = torch.zeros(N, dtype=torch.int32, device="cuda")
for chunk, lanes in itertools.product((32, 48, 64, 96, 128), (2, 3)):
    m.set_chunk(chunk); m.set_option("lanes", lanes)
    m.segment_dev(fr, N, 256, 256, area); m.sync()
    best = 0
    for _ in range(4):
        t0 = time.perf_counter(); m.segment_dev(fr, N, 256, 256, area); m.sync(); best = max(best, N / (time.perf_counter() - t0))
    print(f"chunk {chunk:3d} lanes {lanes}: {best:7.0f} fps", flush=True)
