"""Audit helper: WHERE does a build without the store wait state (-DOG_STORE_NOP=0) go wrong?  One U-Net level, taps of every layer vs the CPU
oracle; for the first wrong tap, histogram the wrong elements by (row in tile, column in tile, channel)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from oracle import unet_oracle as O
feats = (32, 64)
sd = synth.make_unet_state_dict(feats, seed=3, head_scale=2.0, head_bias=-0.3)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fr = synth.random_gray_frames(B, 64, 64, seed=2)
x = (fr.astype("float32") / 255.0)[:, None]
taps = {}
O.forward_numpy(sd, x[:2], taps)
m.set_chunk(B)
m.set_option("keep_taps", 1)
m.set_option("fuse_first", 0); m.set_option("fuse_head", 0)
_, a, l = m.segment(fr, want_mask=False, want_logits=True)
for name in ["downs.0.a", "downs.0.b", "pool0", "downs.1.a", "downs.1.b", "pool1", "bottleneck.a", "bottleneck.b", "ups.0", "ups.1.a", "ups.1.b", "ups.2", "ups.3.a", "ups.3.b"]:
    got = m.activation(name, 2); ref = taps[name]
    bad = np.abs(got - ref) > 1e-4 * max(1, np.abs(ref).max())
    print(f"{name:14s} shape {got.shape} wrong {int(bad.sum())} of {bad.size}")
    if bad.any():
        b, c, y, x_ = np.nonzero(bad)
        print("   by row%8 :", np.bincount(y % 8, minlength=8).tolist())
        print("   by row%16:", np.bincount(y % 16, minlength=16).tolist())
        print("   by col%16:", np.bincount(x_ % 16, minlength=16).tolist())
        print("   by ch%32 :", np.bincount(c % 32, minlength=32).tolist())
        print("   by ch//32:", np.bincount(c // 32).tolist(), " by frame:", np.bincount(b).tolist())
        i = 0
        print("   first wrong:", (b[i], c[i], y[i], x_[i]), "got", got[b[i], c[i], y[i], x_[i]], "ref", ref[b[i], c[i], y[i], x_[i]])
        # is the wrong value some OTHER element of the reference tensor (misplaced) ?
        v = got[b[i], c[i], y[i], x_[i]]
        w = np.argwhere(np.abs(ref[b[i]] - v) < 1e-6)
        print("   value found in reference at (c,y,x):", w[:6].tolist())
        break
