"""Determinism soak of the Winograd chain: the same 256 frames N times (two lanes, graphs), every result compared bit for bit
with the first; then interleaved with direct-form and one-frame calls on the same handle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
m.set_chunk(64); m.set_option("dual", 1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
fr = torch.from_numpy(synth.bulk_gray_frames(256)).cuda()
area = torch.zeros(256, dtype=torch.int32, device="cuda"); mask = torch.zeros((256, 256, 256), dtype=torch.uint8, device="cuda")
logits = torch.zeros((256, 256, 256), dtype=torch.float32, device="cuda")
m.segment_dev(fr, 256, 256, 256, area, mask_dev=mask, logits_dev=logits); m.sync()
a0, m0, l0 = area.clone(), mask.clone(), logits.clone()
bad = 0
for i in range(n):
    if i % 3 == 1:
        m.set_option("wino", 0); m.segment_dev(fr, 64, 256, 256, area[:64]); m.set_option("wino", 1)
    if i % 3 == 2:
        m.set_chunk(1); m.segment_dev(fr, 8, 256, 256, area[:8]); m.set_chunk(64)
    area.zero_(); mask.zero_(); logits.zero_()
    m.segment_dev(fr, 256, 256, 256, area, mask_dev=mask, logits_dev=logits); m.sync()
    ok = torch.equal(area, a0) and torch.equal(mask, m0) and torch.equal(logits, l0)
    bad += not ok
    if not ok: print("mismatch at", i, int((logits != l0).sum()), flush=True)
print("runs", n, "mismatching", bad)
