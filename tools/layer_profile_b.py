"""Per-layer profile at a given micro-batch (one lane).  usage: layer_profile_b.py <frames_per_launch>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(chunk); m.set_option("dual", 0)
frames = torch.from_numpy(synth.bulk_gray_frames(64)).cuda()
m.profile(frames, chunk, 256, 256, reps=3)
prof = m.profile(frames, chunk, 256, 256, reps=20)
tot = sum(p["ms"] for p in prof)
print(f"chain {tot*1e3:.1f} us for {chunk} frame(s) = {chunk/tot*1e3:.0f} frames/s on one lane")
for p in prof:
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
    print(f"{p['layer']:38s} {p['kernel']:28s} {p['ms']*1e3:8.1f} us {tf:7.1f} TF/s")
