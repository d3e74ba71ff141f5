"""Soak of the per-frame host call (zero-copy, eager launches): N calls of unet_segment_frame / segment(area) / segment_stream(BGR) on
one to four frames, interleaved with 64-frame launches, against the 64-frame result -- every mask and area array_equal."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray, unet_segment_frame
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)); m.to("cuda:0").eval()
m.set_chunk(64)
bgr = np.stack([synth.bench_frame_bgr(i) for i in range(192)])
gray = bgr_to_gray(bgr)
ref_mask, ref_area, _ = m.segment(gray)
rs = np.random.RandomState(5)
bad = 0
t0 = time.time()
for it in range(N):
    i = int(rs.randint(0, 188))
    kind = it % 5
    if kind == 0:
        ok = np.array_equal(unet_segment_frame(gray[i], m), ref_mask[i])
    elif kind == 1:
        nb = int(rs.randint(1, 5))
        mk, ar, _ = m.segment(gray[i:i + nb])
        ok = np.array_equal(mk, ref_mask[i:i + nb]) and np.array_equal(ar, ref_area[i:i + nb])
    elif kind == 2:
        nb = int(rs.randint(1, 5))
        mk, ar = m.segment_stream(bgr[i:i + nb], want_mask=True)
        ok = np.array_equal(mk, ref_mask[i:i + nb]) and np.array_equal(ar, ref_area[i:i + nb])
    elif kind == 3:
        mk, ar = m.segment_stream([bgr[i]], want_mask=False)
        ok = ar[0] == ref_area[i]
    else:
        if it % 50 == 4:
            _, ar = m.segment_stream(bgr)
            ok = np.array_equal(ar, ref_area)
        else:
            ok = np.array_equal(unet_segment_frame(gray[i], m), ref_mask[i])
    bad += not ok
print(f"soak: {N} per-frame / few-frame host calls (zero-copy, eager launches) interleaved with 192-frame streams, mismatching calls: {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
