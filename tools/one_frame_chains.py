"""N one-frame chains on ONE lane (hipGraph replay, frames resident), for rocprofv3 --kernel-trace --stats: the per-kernel durations of
the one-frame chain (k_conv_wino_w / _wp, k_convt_w, ...) behind profiles/r04_layer_profile_one_frame_per_chain.txt."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(256)).cuda(); area = torch.zeros(256, dtype=torch.int32, device="cuda")
m.set_chunk(1); m.set_option("lanes", 1)
for a in sys.argv[2:]:      # graphs=0 | any option name=value
    k, v = a.split("=")
    if k == "graphs":
        m.set_graphs(bool(int(v)))
    else:
        m.set_option(k, int(v))
m.segment_dev(fr, 256, 256, 256, area); m.sync()
t0 = time.perf_counter()
for _ in range(max(1, N // 256)):
    m.segment_dev(fr, 256, 256, 256, area)
m.sync()
el = time.perf_counter() - t0
n = max(1, N // 256) * 256
print(f"{n} one-frame chains on one lane: {1e6 * el / n:.1f} us per frame -> {n / el:.0f} frames/s")
