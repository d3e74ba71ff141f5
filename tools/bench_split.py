"""A/B of the split-precision kernels: frames/s and per-kernel ms at 64 frames per launch for option sets given as
NAME=VALUE[,NAME=VALUE...] arguments (each argument = one configuration)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
m.set_chunk(64); m.set_option("precision", 1)
fr = torch.from_numpy(synth.bulk_gray_frames(512)).cuda(); area = torch.zeros(512, dtype=torch.int32, device="cuda")
for cfg in (sys.argv[1:] or ["tile_h=0"]):
    for kv in cfg.split(","):
        k, v = kv.split("=")
        if k == "chunk": m.set_chunk(int(v))
        else: m.set_option(k, int(v))
    for _ in range(2): m.segment_dev(fr, 512, 256, 256, area); m.sync()
    t0 = time.perf_counter()
    for _ in range(10): m.segment_dev(fr, 512, 256, 256, area)
    m.sync(); el = time.perf_counter() - t0
    per = {}
    for p in m.profile(fr, 64, 256, 256, reps=5): per[p["kernel"]] = round(per.get(p["kernel"], 0) + p["ms"], 4)
    print(cfg, "frames/s %.0f" % (5120 / el), per, "sum %.3f" % sum(per.values()), flush=True)
