"""Winograd form against the REFERENCE fixture (tests/golden/unet_full128.npz): sampled-logit error and mask flips, both forms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "unet_full128.npz"))
feats = tuple(int(f) for f in g["features"])
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=int(g["seed"]), head_scale=float(g["head_scale"]), head_bias=float(g["head_bias"]))); m.to("cuda:0").eval()
m.set_chunk(64)
frames, gt = synth.full128_frames()
nz = {(int(f), int(p)): float(v) for f, p, v in zip(g["near_zero_frame"], g["near_zero_pixel"], g["near_zero_logit"])}
for w in (0, 1):
    m.set_option("wino", w)
    masks, areas, logits = m.segment(frames, want_logits=True)
    err = max(float(np.abs(logits[i].ravel()[g["sample_idx"]] - g["logits_samples"][i]).max()) for i in range(128))
    flips = 0; worst = 0.0; bad = 0
    for i in range(128):
        ref = np.unpackbits(g["masks_packed"][i])[:65536].reshape(256, 256) > 0
        fl = np.flatnonzero(((masks[i] > 0) != ref).ravel())
        flips += len(fl)
        for p in fl:
            v = abs(nz.get((i, int(p)), 1.0)); worst = max(worst, v); bad += v > 5e-5
    print(f"wino {w}: max |logit - reference| on the samples {err:.3e}; flipped pixels {flips} (outside the 5e-5 band: {bad}, largest reference |logit| among them {worst:.2e}); area mismatches {(areas.astype(int) != g['areas'].astype(int)).sum()}")
