"""-m gpu: k_conv_wino_w (one frame per kernel chain: the 16 Winograd positions of a tile over the four waves of ONE workgroup,
V transformed in registers, accumulators exchanged through LDS) against k_conv_wino / k_conv_wino_ps -- bit for bit.

The reference calls the network one frame at a time (`openglottal/utils.py:235-237`); a frame's logits must not depend on
which of the Winograd kernels ran its layers (choosing among kernels with the same per-output sums by micro-batch
size is a scheduling choice).  Everything is compared with `array_equal`, logits included, on the workload where a changed
summation order shows: the full-width seeded net on the seeded noise stream (dozens of |logit| <= 1e-5 pixels per hundred frames).
"""
import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray

pytestmark = pytest.mark.gpu

FEATS = (32, 64, 128, 256)


def _kernels(m, fr, B, H, W):
    import torch

    return [p["kernel"] for p in m.profile(torch.from_numpy(fr[:B]).to("cuda:0"), B, H, W, reps=1)]


@pytest.fixture(scope="module")
def base():
    sd = synth.make_unet_state_dict(FEATS, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)   # bench.py's net
    m = og.UNet(1, 1, FEATS)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    m.set_chunk(64)
    fr = np.stack([bgr_to_gray(synth.bench_frame_bgr(i)) for i in range(70)])
    m.set_option("wino_w", 0)
    masks, areas, logits = m.segment(fr, want_logits=True)      # 64-frame launches of k_conv_wino + a 6-frame tail
    m.set_option("wino_w", 1)
    assert int((np.abs(logits) <= 1e-5).sum()) >= 5
    return m, fr, masks, areas, logits


def test_every_winograd_layer_on_the_wave_split_kernel_is_bit_identical(base):
    """Forced onto EVERY Winograd layer (k_conv_wino_w WB 1 and WB 2; k_conv_wino_wp, position rows on four workgroups, wherever
    its workspace allows), one to five frames per chain and a chip-filling one, repeated (arrival order varies)."""
    m, fr, masks, areas, logits = base
    try:
        for force, tag in [(2, ",1>"), (3, ",2>"), (4, "p<")]:
            m.set_option("wino_w", force)
            names = _kernels(m, fr, 1, 256, 256)
            assert sum(k.startswith("k_conv_wino_w") for k in names) == 17, names
            if force == 4:      # (its workspace holds 1 024 tiles: the 256 x 256 layers fall back to k_conv_wino_w<1> from 3 frames per launch on)
                assert sum(k.startswith("k_conv_wino_wp<") for k in names) == 17, names
            else:
                assert all(k.endswith(tag) for k in names if k.startswith("k_conv_wino_w")), names
            for chunk, n in [(1, 10), (2, 10), (5, 10), (64, 70)]:
                m.set_chunk(chunk)
                for rep in range(2):
                    mk, ar, lg = m.segment(fr[:n], want_logits=True)
                    assert np.array_equal(lg, logits[:n]), (force, chunk, rep, float(np.abs(lg - logits[:n]).max()))
                    assert np.array_equal(ar, areas[:n]) and np.array_equal(mk, masks[:n]), (force, chunk, rep)
    finally:
        m.set_option("wino_w", 1)
        m.set_chunk(64)


def test_automatic_choice_at_one_frame_per_chain(base):
    """The default ("wino_w" 1): which kernel a layer takes may depend on the micro-batch size, its bits may not.  One frame per
    chain takes the wave-split kernel on the shallow layers; per-frame calls through the reference's entry point too."""
    m, fr, masks, areas, logits = base
    names = _kernels(m, fr, 1, 256, 256)
    assert sum(k.startswith("k_conv_wino_w") for k in names) >= 6, names
    assert not any("splitK" in k for k in names), names
    names64 = _kernels(m, fr, 64, 256, 256)
    assert not any(k.startswith(("k_conv_wino_w", "k_conv_wino_ps<")) for k in names64), names64
    try:
        for chunk, lanes in [(1, 1), (1, 3), (3, 2), (16, 0)]:
            m.set_chunk(chunk)
            m.set_option("lanes", lanes)
            mk, ar, lg = m.segment(fr[:40], want_logits=True)
            assert np.array_equal(lg, logits[:40]) and np.array_equal(ar, areas[:40]) and np.array_equal(mk, masks[:40]), (chunk, lanes)
        for i in (0, 7, 69):
            assert np.array_equal(og.unet_segment_frame(fr[i], m, "cuda:0"), masks[i]), i
    finally:
        m.set_chunk(64)
        m.set_option("lanes", 0)


@pytest.mark.parametrize("feats,shape", [((32, 64), (128, 256)), ((32, 64), (96, 160)), ((64, 128), (48, 64)), ((40, 80), (64, 64)),
                                         ((32, 64, 128), (64, 32)), ((32, 64), (256, 128)), ((4, 8, 16, 32), (256, 256))])
def test_wave_split_kernel_on_other_shapes(feats, shape):
    """Other widths, depths and frame shapes (maps of 1 to 16 tiles across, padded channel slots, layers whose map does not tile by
    8 x 16 fall back): forced WB 1 / WB 2 and the automatic choice against k_conv_wino at one, two and five frames per chain,
    logits bit for bit; and the canonical result against the oracle."""
    import oracle
    from oracle import unet_oracle as O

    H, W = shape
    sd = synth.make_unet_state_dict(feats, seed=H * 7 + W + len(feats), head_scale=2.0, head_bias=-0.4)
    m = og.UNet(1, 1, feats)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    fr = synth.random_gray_frames(5, H, W, seed=H + W)
    m.set_option("wino_w", 0)
    m.set_option("wino_ps", 0)
    m.set_chunk(5)
    _, a0, l0 = m.segment(fr, want_mask=False, want_logits=True)
    ref_mask, ref_logits = O.segment_frames(sd, fr[:2], backend="torch")
    assert np.abs(l0[:2] - ref_logits).max() <= oracle.reference_band() * max(1.0, float(np.abs(ref_logits).max()))
    m.set_option("wino_ps", 1)
    seen = 0
    for force in (1, 2, 3, 4):
        m.set_option("wino_w", force)
        for chunk in (1, 2, 5):
            m.set_chunk(chunk)
            _, a1, l1 = m.segment(fr, want_mask=False, want_logits=True)
            assert np.array_equal(l1, l0) and np.array_equal(a1, a0), (force, chunk, float(np.abs(l1 - l0).max()))
        seen += sum(k.startswith("k_conv_wino_w") for k in _kernels(m, fr, 1, H, W))
    assert seen > 0     # the wave-split kernel really ran on this shape


def test_transposed_convs_on_wave_tiles_are_bit_identical(base):
    """k_convt_w (one frame per chain: ConvTranspose2d on 32-pixel x 32-column wave tiles, operands straight to registers) against
    the direct kernels: the same k order per output and the same epilogue, so logits bit for bit -- at one to five frames per chain,
    and on a small-width net whose transposed convs have padded channel slots."""
    m, fr, masks, areas, logits = base
    assert sum(k.startswith("k_convt_w<") for k in _kernels(m, fr, 1, 256, 256)) >= 3      # (the 128 -> 256 one fills the chip by itself)
    assert sum(k.startswith("k_convt_w<") for k in _kernels(m, fr, 64, 256, 256)) == 0
    try:
        for on in (0, 1):
            m.set_option("convt_w", on)
            for chunk in (1, 2, 5):
                m.set_chunk(chunk)
                mk, ar, lg = m.segment(fr[:10], want_logits=True)
                assert np.array_equal(lg, logits[:10]) and np.array_equal(ar, areas[:10]) and np.array_equal(mk, masks[:10]), (on, chunk)
    finally:
        m.set_option("convt_w", 1)
        m.set_chunk(64)
    for feats, (H, W) in [((4, 8, 16, 32), (64, 96)), ((40, 80), (32, 32)), ((32, 64, 128), (96, 48))]:
        sd = synth.make_unet_state_dict(feats, seed=H + 3 * W, head_scale=2.0, head_bias=-0.4)
        s = og.UNet(1, 1, feats)
        s.load_state_dict(sd)
        s.to("cuda:0").eval()
        f = synth.random_gray_frames(3, H, W, seed=W)
        s.set_chunk(1)
        s.set_option("convt_w", 0)
        _, a0, l0 = s.segment(f, want_mask=False, want_logits=True)
        s.set_option("convt_w", 1)
        _, a1, l1 = s.segment(f, want_mask=False, want_logits=True)
        assert np.array_equal(l0, l1) and np.array_equal(a0, a1), (feats, H, W, float(np.abs(l0 - l1).max()))
        assert any(k.startswith("k_convt_w<") for k in _kernels(s, f, 1, H, W)), (feats, H, W)
