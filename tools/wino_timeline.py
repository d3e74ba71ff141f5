"""In-kernel timeline of k_conv_wino at a chip-filling micro-batch: per launch, medians over the first 511 workgroups of the
phases entry -> first DMAs issued -> landed -> main loop starts -> main loop done -> output transform done -> epilogue issued ->
stores acknowledged, in shader-clock cycles (s_memtime)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd._lib import lib, ptr, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
OPTS = [a.split("=") for a in sys.argv[2:]]
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
for k, v in OPTS:
    m.set_option(k, int(v))
fr = torch.from_numpy(synth.bulk_gray_frames(B)).cuda()
prof = m.profile(fr, B, 256, 256, reps=2)
m.clock_probe(fr, B, 256, 256)
buf = np.zeros((512, 8), np.uint64)
print(f"{'layer':28s} {'kernel':16s} {'to 1st DMA':>10s} {'DMA wait':>9s} {'1st xform':>9s} {'loop':>8s} {'out xform':>9s} {'epilogue':>9s} {'store ack':>9s} {'total':>8s}  (cycles, median of 510 workgroups)")
for i, p in enumerate(prof):
    if not p["kernel"].startswith("k_conv_wino<"):
        continue
    check(lib().og_unet_clock_probe_raw(m._h, i, ptr(buf)), "raw")
    v = buf[:510].astype(np.float64)
    v = v[(v[:, 0] > 0) & (v[:, 7] > v[:, 0])]
    d = np.diff(v, axis=1)
    med = np.median(d, axis=0)
    print(f"{p['layer']:28s} {p['kernel']:16s} {med[0]:10.0f} {med[1]:9.0f} {med[2]:9.0f} {med[3]:8.0f} {med[4]:9.0f} {med[5]:9.0f} {med[6]:9.0f} {np.median(v[:,7]-v[:,0]):8.0f}")
