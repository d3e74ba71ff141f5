"""Debug helper: the ragged-tail scenario of tests/test_gpu_invariance.py under option variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth

graphs, dual, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=3, head_scale=2.0, head_bias=-0.3)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
m.set_graphs(bool(graphs)); m.set_option("dual", dual)
fr = synth.random_gray_frames(n, 256, 256, seed=5)
m.set_chunk(64)
m.set_option("wino_ps", 0)
mk0, a0, l0 = m.segment(fr, want_logits=True)
print("reference done", flush=True)
m.set_option("wino_ps", 1)
for rep in range(2):
    mk1, a1, l1 = m.segment(fr, want_logits=True)
    print("ps done rep", rep, "identical:", np.array_equal(l0, l1), "max|d|", float(np.abs(l0 - l1).max()), flush=True)
