"""One frame per kernel chain (BASELINE configs[1]; the reference's call pattern, utils.py:235-237) in the canonical form:
frames/s by lanes (resident frames, hipGraph replay) and the per-call time of `unet_segment_frame` (host array in, mask out),
for each way of scheduling the under-filled Winograd launches.  usage: latency_r04.py [NAME=VALUE option ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import unet_segment_frame
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
for kv in sys.argv[1:]:
    k, v = kv.split("="); m.set_option(k, int(v))
N = 256
fr_h = synth.bulk_gray_frames(N)
fr = torch.from_numpy(fr_h).cuda(); area = torch.zeros(N, dtype=torch.int32, device="cuda")
ref = None
for name, opts in [("wave-split + position-split (default)", {"wino_w": 1, "wino_ps": 1}),
                   ("position-split launches only (round 3)", {"wino_w": 0, "wino_ps": 1}),
                   ("wave-split forced on every layer, WB 1", {"wino_w": 2, "wino_ps": 1}),
                   ("k_conv_wino only", {"wino_w": 0, "wino_ps": 0})]:
    for k, v in opts.items():
        m.set_option(k, v)
    for lanes in (1, 2, 3):
        m.set_option("lanes", lanes); m.set_chunk(1)
        m.segment_dev(fr, N, 256, 256, area); m.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); m.segment_dev(fr, N, 256, 256, area); m.sync(); best = min(best, time.perf_counter() - t0)
        a = area.cpu().numpy().copy()
        ref = a if ref is None else ref
        assert np.array_equal(a, ref), name
        print(f"{name:42s} lanes {lanes}: {1e6*best/N:7.1f} us/frame -> {N/best:7.0f} frames/s", flush=True)
    m.set_option("lanes", 0)
    for g in fr_h[:20]: unet_segment_frame(g, m, "cuda:0")
    t0 = time.perf_counter()
    for g in fr_h[:200]: unet_segment_frame(g, m, "cuda:0")
    print(f"{name:42s} unet_segment_frame: {1e3*(time.perf_counter()-t0)/200:.3f} ms per call", flush=True)
