"""One frame per kernel chain in the canonical form: frames/s by lanes (resident frames, hipGraph replay) with and without the
position-split launches, the opt-in split-K path beside it, and the per-call time of `unet_segment_frame` (host array in,
mask out) -- the reference's call pattern (utils.py:218-241)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import unet_segment_frame
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
N = 256
fr_h = synth.bulk_gray_frames(N)
fr = torch.from_numpy(fr_h).cuda(); area = torch.zeros(N, dtype=torch.int32, device="cuda")
for name, opts in [("canonical, position-split launches", {"wino": 1, "splitk": 0, "wino_ps": 1}),
                   ("canonical, k_conv_wino only", {"wino": 1, "splitk": 0, "wino_ps": 0}),
                   ("opt-in split-K on the direct kernels", {"wino": 0, "splitk": 1, "wino_ps": 0})]:
    for k, v in opts.items():
        m.set_option(k, v)
    for lanes in (1, 2, 3):
        m.set_option("lanes", lanes); m.set_chunk(1)
        m.segment_dev(fr, N, 256, 256, area); m.sync()
        t0 = time.perf_counter(); m.segment_dev(fr, N, 256, 256, area); m.sync(); t2 = time.perf_counter()
        print(f"{name:40s} lanes {lanes}: {1e6*(t2-t0)/N:7.1f} us/frame -> {N/(t2-t0):7.0f} frames/s", flush=True)
    m.set_option("lanes", 0)
    for g in fr_h[:20]: unet_segment_frame(g, m, "cuda:0")
    t0 = time.perf_counter()
    for g in fr_h[:200]: unet_segment_frame(g, m, "cuda:0")
    print(f"{name:40s} unet_segment_frame: {1e3*(time.perf_counter()-t0)/200:.3f} ms per call", flush=True)
