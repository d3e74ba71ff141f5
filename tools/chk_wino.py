"""Winograd form (option "wino") against the direct kernels: logits / areas on seeded frames, per-layer timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)); m.to("cuda:0").eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gray = synth.bulk_gray_frames(n)
m.set_chunk(64)
_, a0, l0 = m.segment(gray, want_mask=False, want_logits=True)
m.set_option("wino", 1)
_, a1, l1 = m.segment(gray, want_mask=False, want_logits=True)
_, a2, l2 = m.segment(gray, want_mask=False, want_logits=True)
print("max |dlogit| wino vs direct", float(np.abs(l1 - l0).max()), "logit scale", float(np.abs(l0).max()), "deterministic", bool(np.array_equal(l1, l2)))
print("areas differing", int((a0 != a1).sum()), "of", n, "max", int(np.abs(a0.astype(int) - a1.astype(int)).max()))
fr = torch.from_numpy(synth.bulk_gray_frames(512)).cuda(); area = torch.zeros(512, dtype=torch.int32, device="cuda")
for w in (0, 1):
    m.set_option("wino", w)
    m.segment_dev(fr, 512, 256, 256, area); m.sync()
    best = 0
    for _ in range(3):
        t0 = time.perf_counter(); m.segment_dev(fr, 512, 256, 256, area); m.sync(); best = max(best, 512 / (time.perf_counter() - t0))
    print("wino", w, "frames/s", round(best))
    for r in m.profile(fr[:64], 64, 256, 256):
        print("    %-12s %-28s %7.3f ms %7.1f TFLOP/s" % (r["layer"], r["kernel"], r["ms"], r["flops"] / r["ms"] / 1e9))
