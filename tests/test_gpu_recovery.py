"""-m gpu: a call that fails half-way leaves the handle usable.

The fused reduces of the one-frame chain (k_conv_wino_wp, k_conv_wino_ps, the opt-in split-K) rely on arrival counters that are zero
at launch; only a tile's last arriver re-zeroes them.  A launch that fails in the middle of a chain must not leave counters behind:
the next call would return stale activations with rc 0.  The test hook "inject_fault" n makes the n-th conv launch fail (OG_EHIP)
after scribbling over the counters, as a launch that died half-way would; every error path has to restore them.  Nothing throws or
aborts across the ABI (include/openglottal_hip.h); the reference's own error convention is an exception from the model call
(openglottal/utils.py:236-237)."""
import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd._lib import OpenGlottalHipError

pytestmark = pytest.mark.gpu

FEATS = (32, 64, 128, 256)


def test_next_call_after_a_failed_launch_is_bit_identical():
    sd = synth.make_unet_state_dict(FEATS, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)
    m = og.UNet(1, 1, FEATS)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    fr = synth.bulk_gray_frames(12)
    for lanes, entry in [(1, "host"), (3, "host"), (1, "stream")]:
        m.set_option("lanes", lanes)
        m.set_chunk(1)                      # one frame per chain: the position-row-split launches and their counters
        run = (lambda: m.segment(fr, want_logits=True)) if entry == "host" else (lambda: m.segment_stream(np.repeat(fr[..., None], 3, axis=-1), want_mask=True))
        ref = run()
        for n in (1, 7, 12, 20):
            m.set_option("inject_fault", n)
            with pytest.raises(OpenGlottalHipError, match="injected"):
                run()
            again = run()
            for a, b in zip(ref, again):
                assert np.array_equal(a, b), (lanes, entry, n)
    m.set_option("lanes", 0)
    with pytest.raises(OpenGlottalHipError):      # the f32 parity entry point too
        m.set_option("inject_fault", 3)
        m(np.zeros((1, 1, 256, 256), np.float32))
    x = np.random.RandomState(0).rand(1, 1, 256, 256).astype(np.float32)
    y0 = m(x)
    assert np.array_equal(y0, m(x))
