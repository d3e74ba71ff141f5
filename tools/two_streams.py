"""Experiment: do two independent kernel chains (two handles = two streams + two arenas) overlap each other's
launch tails?  Combined frames/s of 2 handles vs 1 handle on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
ms = []
for _ in range(3):
    m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); ms.append(m)
F = 512
frames = [torch.from_numpy(synth.bulk_gray_frames(F, seed=s)).cuda() for s in range(3)]
areas = [torch.zeros(F, dtype=torch.int32, device="cuda") for _ in range(3)]
for chunk in (32, 64):
    for n in (1, 2, 3):
        for m in ms[:n]:
            m.set_chunk(chunk)
        best = 0
        for rep in range(4):
            t0 = time.perf_counter()
            for _ in range(2):
                for i in range(n):
                    ms[i].segment_dev(frames[i], F, 256, 256, areas[i])
            for i in range(n):
                ms[i].sync()
            best = max(best, 2 * n * F / (time.perf_counter() - t0))
        print(f"chunk={chunk} handles={n}: {best:8.1f} fps", flush=True)
