"""unet_segment_frame per call (host arrays in and out), hipGraph replay of the chain's body on / off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import unet_segment_frame
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = synth.bulk_gray_frames(64)
for a in sys.argv[1:]:
    k, v = a.split("=")
    m.set_option(k, int(v))
for graphs in (1, 0, 1, 0):
    m.set_graphs(bool(graphs))
    for i in range(32):
        unet_segment_frame(fr[i], m)
    ts = []
    for rep in range(8):
        t0 = time.perf_counter()
        for i in range(64):
            unet_segment_frame(fr[i], m)
        ts.append((time.perf_counter() - t0) / 64)
    print(f"graphs {graphs}: unet_segment_frame {1e3 * min(ts):.4f} ms per call (best of 8 x 64), median {1e3 * float(np.median(ts)):.4f}", flush=True)
