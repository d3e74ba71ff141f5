"""Per-launch table of one kernel chain (HIP events around every launch): ms, TFLOP/s, % of f32 MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
frames = torch.from_numpy(synth.bulk_gray_frames(B)).cuda()
m.profile(frames, B, 256, 256, reps=2)
prof = m.profile(frames, B, 256, 256, reps=10)
tot = sum(p["ms"] for p in prof)
print(f"B={B} chain {tot:.3f} ms  -> {B / tot * 1e3:.0f} fps (eager, event-bracketed)")
for p in prof:
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
    print(f"{p['layer']:28s} {p['kernel']:20s} {p['ms']:8.4f} ms {100 * p['ms'] / tot:5.1f}%  {tf:7.1f} TF/s  {100 * tf / 157.3:5.1f}% of peak")
