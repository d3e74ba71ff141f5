"""PCIe-inclusive rate of the host-buffer entry point (og_unet_segment_u8: numpy frames in, areas out)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=1, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
m.set_chunk(64)
fr = synth.bulk_gray_frames(1024)
out = {}
for want_mask in (False, True):
    m.segment(fr[:128], want_mask=want_mask)
    t0 = time.perf_counter()
    for _ in range(3):
        m.segment(fr, want_mask=want_mask)
    out["areas_only" if not want_mask else "areas_and_masks"] = round(3 * len(fr) / (time.perf_counter() - t0), 1)
print(json.dumps({"host_buffer_frames_per_s": out, "frames": len(fr)}))
