"""Per-launch table of one kernel chain (HIP events around every launch, one lane, eager): ms, share of the chain, algorithmic
TFLOP/s (SURVEY 8(d)'s direct-form count) and the fraction of the f32 MFMA peak the launch EXECUTES (a Winograd launch issues
16/36 of the direct form's multiplies).  usage: layer_profile.py [frames per launch = 16] [NAME=VALUE option ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    m.set_option(k, int(v))
frames = torch.from_numpy(synth.bulk_gray_frames(B)).cuda()
m.profile(frames, B, 256, 256, reps=2)
prof = m.profile(frames, B, 256, 256, reps=10)
tot = sum(p["ms"] for p in prof)
ex = lambda p: p["flops"] / (2.25 if p["kernel"].startswith("k_conv_wino") else 1.0)
print(f"B={B} options={sys.argv[2:]} chain {tot:.3f} ms -> {B / tot * 1e3:.0f} frames/s (eager, event-bracketed, one lane); "
      f"chain executed {sum(ex(p) for p in prof) / tot / 1e9:.1f} TFLOP/s = {sum(ex(p) for p in prof) / tot / 1e9 / 157.3:.3f} of the f32 MFMA peak")
print(f"{'layer':36s} {'kernel':28s} {'ms':>8s} {'share':>6s} {'alg TF/s':>9s} {'executed frac':>13s}")
for p in prof:
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
    print(f"{p['layer']:36s} {p['kernel']:28s} {p['ms']:8.4f} {100 * p['ms'] / tot:5.1f}% {tf:9.1f} {ex(p) / (p['ms'] * 1e-3) / 1e12 / 157.3:13.3f}")
