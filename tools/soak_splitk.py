"""Soak of the fused split-K reduce: many batch-1 chains on 3 lanes must equal the separate-reduce-kernel results bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)); m.to("cuda:0").eval()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
fr = torch.from_numpy(synth.bulk_gray_frames(F, seed=5)).cuda()
def run(fused, chunk):
    m.set_chunk(chunk); m.set_option("splitk_fused", fused)
    area = torch.zeros(F, dtype=torch.int32, device="cuda"); logits = torch.zeros((F, 256, 256), dtype=torch.float32, device="cuda")
    m.segment_dev(fr, F, 256, 256, area, logits_dev=logits); m.sync()
    return area.cpu().numpy(), logits
bad = 0
for chunk in (1, 2, 3):
    a0, l0 = run(0, chunk)
    for rep in range(3):
        a1, l1 = run(1, chunk)
        eq = bool(torch.equal(l0, l1)) and np.array_equal(a0, a1)
        bad += not eq
        print(f"chunk {chunk} rep {rep}: fused == separate: {eq}", flush=True)
print("SOAK", "FAILED" if bad else "OK", F, "frames")
sys.exit(1 if bad else 0)
