"""Debug helper for k_conv_wino_ps: eager (no graphs), one lane, step by step with prints."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth

feats = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,64,128,256").split(","))
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sd = synth.make_unet_state_dict(feats, seed=3, head_scale=2.0, head_bias=-0.3)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
m.set_graphs(False); m.set_option("dual", 0)
fr = synth.random_gray_frames(B, H, H, seed=5)
m.set_chunk(B)
m.set_option("wino_ps", 0)
_, a0, l0 = m.segment(fr, want_mask=False, want_logits=True)
print("reference (k_conv_wino) done", a0.tolist(), flush=True)
m.set_option("wino_ps", ps)
import torch
print([p["kernel"] for p in m.profile(torch.from_numpy(fr).cuda(), B, H, H, reps=1)], flush=True)
_, a1, l1 = m.segment(fr, want_mask=False, want_logits=True)
print("ps done", a1.tolist(), "identical:", np.array_equal(l0, l1), "max|d|", float(np.abs(l0 - l1).max()), flush=True)
