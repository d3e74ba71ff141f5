"""Audit helper (profiles/r02_epilogue_fence_audit.md): one quick correctness probe of the library named by OPENGLOTTAL_HIP_LIB (8 full-width frames vs the
reference fixture, the same call twice, a 64-frame chunk twice).  Used by tools/epilogue_fence_audit.sh probe."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import openglottal_amd as og
from openglottal_amd import synth
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "unet_full.npz"))
sd = synth.make_unet_state_dict(tuple(g["features"]), seed=int(g["seed"]), head_scale=float(g["head_scale"]), head_bias=float(g["head_bias"]))
m = og.UNet(1, 1, tuple(int(f) for f in g["features"])); m.load_state_dict(sd); m.to("cuda:0").eval()
fr = np.concatenate([synth.random_gray_frames(4, seed=7), synth.glottis_frames(1, 4, seed=99)[0]])
_, a, l = m.segment(fr, want_mask=False, want_logits=True)
err = np.abs(l.reshape(8, -1)[:, g["sample_idx"]] - g["logits_samples"]).max()
_, a2, l2 = m.segment(fr, want_mask=False, want_logits=True)
big = np.concatenate([fr] * 8)
m.set_chunk(64)
_, ab, lb = m.segment(big, want_mask=False, want_logits=True)
_, ab2, lb2 = m.segment(big, want_mask=False, want_logits=True)
print(os.path.basename(os.environ.get("OPENGLOTTAL_HIP_LIB", "default")), "max|dlogit| vs reference %.3g" % err,
      "8-frame repeat:", np.array_equal(l, l2), "64-frame repeat:", np.array_equal(lb, lb2),
      "64-frame differing:", int((lb != lb2).sum()), "big==small:", np.array_equal(lb[:8], l))
