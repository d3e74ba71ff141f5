import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from oracle import unet_oracle as O
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(16)
fr = synth.random_gray_frames(32, 512, 512, seed=4)
masks, areas, logits = m.segment(fr, want_logits=True)
t0 = time.perf_counter(); masks, areas, logits = m.segment(fr, want_logits=True); dt = time.perf_counter() - t0
ref_mask, ref_logits = O.segment_frames(sd, fr[:2], backend="torch")
scale = max(1.0, np.abs(ref_logits).max())
print("512x512: max|dlogit|", float(np.abs(logits[:2] - ref_logits).max()), "tol", 5e-5 * scale, "flipped", int(((masks[:2] > 0) != (ref_mask > 0)).sum()),
      "fps(host buffers, with masks+logits)", round(32 / dt, 1), "= 256x256-equivalents/s", round(4 * 32 / dt))
