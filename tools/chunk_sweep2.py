"""Throughput vs frames-per-launch for a given kernel configuration (options as k=v args)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
for kv in sys.argv[1:]:
    k, v = kv.split("="); m.set_option(k, int(v))
F = 576
frames = torch.from_numpy(synth.bulk_gray_frames(F)).cuda()
area = torch.zeros(F, dtype=torch.int32, device="cuda")
for chunk in (16, 24, 32, 48, 64, 96, 144, 192):
    m.set_chunk(chunk)
    m.segment_dev(frames, F, 256, 256, area); m.sync()
    best = 0
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(2):
            m.segment_dev(frames, F, 256, 256, area)
        m.sync()
        best = max(best, 2 * F / (time.perf_counter() - t0))
    print(f"{' '.join(sys.argv[1:])} chunk={chunk:3d} fps={best:9.1f}", flush=True)
