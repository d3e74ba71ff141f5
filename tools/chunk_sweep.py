"""Throughput vs frames-per-launch (micro-batch) on one GPU; prints one line per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth

feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
frames = torch.from_numpy(synth.bulk_gray_frames(F)).cuda()
area = torch.zeros(F, dtype=torch.int32, device="cuda")
for graphs in (True, False):
    m.set_graphs(graphs)
    for chunk in (1, 2, 4, 8, 16, 32, 64):
        m.set_chunk(chunk)
        m.segment_dev(frames, F, 256, 256, area); m.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            m.segment_dev(frames, F, 256, 256, area)
        m.sync()
        el = time.perf_counter() - t0
        print(f"graphs={int(graphs)} chunk={chunk:3d} fps={3 * F / el:9.1f}  ms/frame={1e3 * el / (3 * F):.3f}", flush=True)
