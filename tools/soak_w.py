"""Soak of the one-frame chain of round 4 (k_conv_wino_w: LDS exchange between the waves; k_conv_wino_wp: four workgroups per tile,
exchange through the split-K workspace, arrival counters reset by the last arriver, under hipGraph replay on one to three lanes;
k_convt_w): N one-, two- and three-frame chains against the 64-frames-per-chain result (k_conv_wino), logits bit for bit, R rounds;
every fourth pass forces k_conv_wino_wp onto every layer its workspace allows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)); m.to("cuda:0").eval()
N = 192
fr = np.stack([bgr_to_gray(synth.bench_frame_bgr(i)) for i in range(N)])
m.set_chunk(64)
_, a0, l0 = m.segment(fr, want_mask=False, want_logits=True)
bad = passes = 0
t0 = time.time()
for r in range(R):
    for chunk, lanes, force in ((1, 3, 1), (2, 3, 1), (3, 2, 1), (1, 1, 1), (1, 2, 4), (1, 1, 4)):
        m.set_chunk(chunk); m.set_option("lanes", lanes); m.set_option("wino_w", force)
        _, a, l = m.segment(fr, want_mask=False, want_logits=True)
        passes += 1
        d = int((l != l0).sum())
        if d or not np.array_equal(a, a0):
            bad += 1
            print(f"round {r} chunk {chunk} lanes {lanes} wino_w {force}: {d} logits differ, {int((a != a0).sum())} areas differ", flush=True)
print(f"soak: {R} rounds x 6 configurations x {N} frames = {passes * N} frames through one- to three-frame chains, mismatching passes: {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
