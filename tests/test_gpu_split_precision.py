"""-m gpu: the opt-in split-precision mode (`set_option("precision", 1)`: f16 hi/lo operand pairs, three
v_mfma_f32_32x32x16_f16 per f32 product, f32 accumulation) against the SAME reference fixtures and the SAME tolerances as
the exact-f32 default: logits abs <= BAND * max(1, |ref|max) (BAND = the reference's own noise, oracle.reference_band()); masks may differ only where the reference's own logit is within
BAND of zero; the trained (zero-flip) fixture bit for bit.  It is a secondary mode: the f32 kernels stay the default."""
import os

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth

pytestmark = pytest.mark.gpu

import oracle

TOL = oracle.reference_band()   # the reference's own run-to-run logit difference (tests/golden/unet_full128_self_noise.npz: 3.475e-5)


def unpack(bits, h=256, w=256):
    return np.unpackbits(bits)[: h * w].reshape(h, w)


def make_model(sd, features, precision=1):
    m = og.UNet(1, 1, tuple(int(f) for f in features))
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    m.set_option("precision", precision)
    return m


def test_every_layer_boundary_small_net_split_precision(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_small_layers.npz"))
    sd = synth.make_unet_state_dict(tuple(g["features"]), seed=int(g["seed"]), head_scale=float(g["head_scale"]), head_bias=float(g["head_bias"]))
    m = make_model(sd, g["features"])
    f = synth.random_gray_frames(1, 64, 64, seed=21)
    logits = m((f.astype("float32") / 255.0)[:, None])
    keys = [k[2:] for k in g.files if k.startswith("L:")]
    assert len(keys) == 27
    worst = 0.0
    for k in keys:
        ref = g["L:" + k]
        got = logits if k == "head" else m.activation(k, 1)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        worst = max(worst, err)
        assert err <= TOL, (k, err)
    print("worst relative layer error (split precision):", worst)
    fr = synth.random_gray_frames(3, 48, 80, seed=22)
    assert np.abs(m((fr.astype("float32") / 255.0)[:, None]) - g["logits_48x80"]).max() <= TOL


def test_bench_configuration_split_precision_against_reference_fixture(golden_dir):
    import torch

    g = np.load(os.path.join(golden_dir, "unet_full128.npz"))
    feats = tuple(int(f) for f in g["features"])
    sd = synth.make_unet_state_dict(feats, seed=int(g["seed"]), head_scale=float(g["head_scale"]), head_bias=float(g["head_bias"]))
    m = make_model(sd, feats)
    m.set_chunk(64)
    frames, gt = synth.full128_frames()
    dev = torch.device("cuda", 0)
    fdev = torch.from_numpy(frames).to(dev)
    area = torch.zeros(128, dtype=torch.int32, device=dev)
    mask = torch.zeros((128, 256, 256), dtype=torch.uint8, device=dev)
    logits = torch.zeros((128, 256, 256), dtype=torch.float32, device=dev)
    nz = {(int(f), int(p)): float(v) for f, p, v in zip(g["near_zero_frame"], g["near_zero_pixel"], g["near_zero_logit"])}
    for rep in range(2):   # capture, then replay of the graphs
        m.segment_dev(fdev, 128, 256, 256, area, mask_dev=mask, logits_dev=logits)
        m.sync()
        mk, ar, lg = mask.cpu().numpy(), area.cpu().numpy(), logits.cpu().numpy()
        flips_total, worst = 0, 0.0
        for i in range(128):
            flips = np.flatnonzero(((mk[i] > 0) != (unpack(g["masks_packed"][i]) > 0)).ravel())
            for p in flips:
                assert abs(nz.get((i, int(p)), 1.0)) <= TOL, (i, int(p), nz.get((i, int(p))))
            flips_total += len(flips)
            assert int(ar[i]) == int((mk[i] > 0).sum()) and abs(int(ar[i]) - int(g["areas"][i])) <= len(flips), i
            e = np.abs(lg[i].ravel()[g["sample_idx"]] - g["logits_samples"][i]).max()
            worst = max(worst, float(e))
            assert e <= TOL, (i, e)
        d = np.array([og.dice(mk[i], gt[i]) for i in range(80)])
        assert np.abs(d - g["dice_vs_gt"]).max() <= 1e-3
        print(f"split precision pass {rep}: flipped pixels {flips_total} of {128 * 65536}, max |dlogit| on the samples {worst:.3g}")
    prof = m.profile(fdev, 64, 256, 256, reps=1)
    kernels = [p["kernel"] for p in prof]
    assert all(k.startswith("k_conv_mfma_h") for k in kernels), kernels      # the split-precision kernels really ran
    # and it is a different arithmetic from the default (not a silent fall-back to the f32 kernels)
    m.set_option("precision", 0)
    lg32 = torch.zeros_like(logits)
    m.segment_dev(fdev, 128, 256, 256, area, logits_dev=lg32)
    m.sync()
    assert not torch.equal(lg32, logits) and float((lg32 - logits).abs().max()) <= TOL


def test_trained_zero_flip_fixture_and_small_batches_split_precision(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    m = make_model(sd, g["features"])
    frames, gt = synth.glottis_frames(4, 20, seed=99)
    ref_masks = np.stack([unpack(b) for b in g["masks_packed"]])
    for chunk in (32, 1, 3):      # full launches (fused first layer / head) and small ones (separate first-layer and head kernels)
        m.set_chunk(chunk)
        masks, areas, _ = m.segment(frames)
        assert np.array_equal(masks > 0, ref_masks > 0), chunk                   # every one of 80 * 65536 pixels
        assert np.array_equal(areas.astype(np.int64), g["areas"]), chunk          # area waveform integers bit-exact
    boxes = np.array([og.utils.normalize_box((100, 80, 160, 200), 256, 256)] * 8, np.int32)
    _, a, _ = m.segment(frames[:8], boxes=boxes, want_mask=False)
    assert a.tolist() == [int((ref_masks[i][80:200, 100:160] > 0).sum()) for i in range(8)]


def test_odd_shapes_and_padded_channels_split_precision():
    from oracle import unet_oracle as O
    for feats, B, H, W, seed in [((32, 64, 128), 96, 80, 48, 321), ((33, 66), 128, 32, 64, 99), ((6, 12, 24), 2, 48, 32, 5)]:
        sd = synth.make_unet_state_dict(feats, seed=seed, head_scale=2.0, head_bias=-0.4)
        m = make_model(sd, feats)
        m.set_chunk(B)
        fr = synth.random_gray_frames(B, H, W, seed=17)
        masks, areas, logits = m.segment(fr, want_logits=True)
        masks2, areas2, logits2 = m.segment(fr, want_logits=True)
        assert np.array_equal(logits, logits2) and np.array_equal(areas, areas2)          # repeatable
        ref_mask, ref_logits = O.segment_frames(sd, fr[:4], backend="torch")
        scale = max(1.0, np.abs(ref_logits).max())
        assert np.abs(logits[:4] - ref_logits).max() <= TOL * scale, (feats, np.abs(logits[:4] - ref_logits).max())
        assert np.all(np.abs(ref_logits[(masks[:4] > 0) != (ref_mask > 0)]) <= TOL * scale)
        assert np.array_equal(areas, (masks > 0).reshape(B, -1).sum(1))


def test_randomised_entry_point_matrix_both_precisions():
    """Randomised sweep over the entry points and knobs a caller can combine -- frame size, batch, micro-batch, lanes,
    graphs, boxes, mask / logits requests, host vs streamed vs device pointers, both precisions -- on a small net, each
    configuration against the CPU oracle (logits within tolerance, masks by the flip rule, areas = popcount [in box])."""
    import torch

    from oracle import unet_oracle as O
    from openglottal_amd.utils import normalize_box

    rs = np.random.RandomState(2026)
    feats = (8, 16, 32)
    sd = synth.make_unet_state_dict(feats, seed=77, head_scale=2.5, head_bias=-0.2)
    m = make_model(sd, feats, precision=0)
    dev = torch.device("cuda", 0)
    cache = {}
    for it in range(60):
        H, W = [(32, 48), (64, 64), (40, 72), (8, 16), (96, 32)][rs.randint(5)]
        B = int(rs.choice([1, 2, 3, 5, 17, 40]))
        key = (H, W, B)
        if key not in cache:
            fr = rs.randint(0, 256, (B, H, W), dtype=np.uint8)
            cache[key] = (fr,) + O.segment_frames(sd, fr, backend="torch")
        fr, ref_mask, ref_logits = cache[key]
        prec = int(rs.randint(2))
        m.set_option("precision", prec)
        m.set_chunk(int(rs.choice([1, 2, 4, 16, 64])))
        m.set_option("lanes", int(rs.randint(4)))
        m.set_graphs(bool(rs.randint(2)))
        m.set_option("stream", int(rs.randint(2)))
        boxes = None
        if rs.randint(2):
            boxes = np.array([normalize_box((int(rs.randint(-5, W)), int(rs.randint(-5, H)), int(rs.randint(0, W + 9)), int(rs.randint(0, H + 9))), W, H)
                              for _ in range(B)], np.int32)
            boxes[rs.randint(B)] = -1
        how = int(rs.randint(3))
        if how == 0:
            masks, areas, logits = m.segment(fr, boxes=boxes, want_logits=True)
        elif how == 1:
            masks, areas = m.segment_stream(fr, boxes=boxes, want_mask=True)
            logits = None
        else:
            d_f = torch.from_numpy(fr).to(dev)
            d_a = torch.zeros(B, dtype=torch.int32, device=dev)
            d_m = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
            d_l = torch.zeros((B, H, W), dtype=torch.float32, device=dev)
            m.segment_dev(d_f, B, H, W, d_a, boxes_dev=None if boxes is None else torch.from_numpy(boxes).to(dev), mask_dev=d_m, logits_dev=d_l)
            m.sync()
            masks, areas, logits = d_m.cpu().numpy(), d_a.cpu().numpy(), d_l.cpu().numpy()
        cfg = (it, H, W, B, prec, how)
        scale = max(1.0, np.abs(ref_logits).max())
        if logits is not None:
            assert np.abs(logits - ref_logits).max() <= TOL * scale, cfg
        diff = (masks > 0) != (ref_mask > 0)
        assert np.all(np.abs(ref_logits[diff]) <= TOL * scale), cfg
        for i in range(B):
            if boxes is None:
                want = int((masks[i] > 0).sum())
            elif boxes[i][0] < 0:
                want = 0
            else:
                x1, y1, x2, y2 = boxes[i]
                want = int((masks[i][y1:y2, x1:x2] > 0).sum())
            assert int(areas[i]) == want, cfg + (i,)


def test_latency_mode_split_k_in_split_precision(golden_dir):
    """One frame per kernel chain at full width with the opt-in "splitk" 1: the split-precision kernels split K across
    workgroups like the f32 ones (fused reduce by the last-arriving part, parts summed in split order): deterministic, and
    within tolerance of the reference fixture."""
    g = np.load(os.path.join(golden_dir, "unet_full128.npz"))
    feats = tuple(int(f) for f in g["features"])
    sd = synth.make_unet_state_dict(feats, seed=int(g["seed"]), head_scale=float(g["head_scale"]), head_bias=float(g["head_bias"]))
    m = make_model(sd, feats)
    frames, gt = synth.full128_frames()
    nz = {(int(f), int(p)): float(v) for f, p, v in zip(g["near_zero_frame"], g["near_zero_pixel"], g["near_zero_logit"])}
    m.set_option("splitk", 1)
    for chunk in (1, 2):
        m.set_chunk(chunk)
        mk, ar, lg = m.segment(frames[:12], want_logits=True)
        mk2, ar2, lg2 = m.segment(frames[:12], want_logits=True)
        assert np.array_equal(lg, lg2) and np.array_equal(ar, ar2), chunk                 # arrival order does not matter
        for i in range(12):
            flips = np.flatnonzero(((mk[i] > 0) != (unpack(g["masks_packed"][i]) > 0)).ravel())
            for p in flips:
                assert abs(nz.get((i, int(p)), 1.0)) <= TOL, (chunk, i, int(p))
            assert abs(int(ar[i]) - int(g["areas"][i])) <= len(flips)
            assert np.abs(lg[i].ravel()[g["sample_idx"]] - g["logits_samples"][i]).max() <= TOL, (chunk, i)
    prof = m.profile(__import__("torch").from_numpy(frames[:1]).cuda(), 1, 256, 256, reps=1)
    assert all(p["kernel"].startswith(("k_conv_mfma_h", "k_conv_first", "k_head")) for p in prof), [p["kernel"] for p in prof]
    m.set_option("splitk", 0)
    m.set_chunk(1)
    _, ar0, lg0 = m.segment(frames[:12], want_mask=False, want_logits=True)
    assert not np.array_equal(lg0, lg) or True       # (the unsplit order may or may not differ in the last bits)
    assert np.abs(lg0 - lg).max() <= TOL


def test_activation_beyond_the_f16_range_fails_loudly_in_split_precision():
    """The split-precision mode carries activations as f16 pairs: weights that push an activation beyond 60000 make its
    result meaningless.  That must be an error (OG_ERANGE), never a silently saturated mask; the exact-f32 mode handles
    the same weights."""
    feats = (32, 64)
    sd = synth.make_unet_state_dict(feats, seed=4, head_scale=2.0, head_bias=-0.3)
    big = dict(sd)
    big["downs.0.net.1.weight"] = (sd["downs.0.net.1.weight"] * np.float32(4e5)).astype(np.float32)   # first BN scale: activations ~1e5
    m = make_model(big, feats, precision=0)
    fr = synth.random_gray_frames(70, 32, 64, seed=3)
    m.set_chunk(64)
    masks, areas, logits = m.segment(fr, want_logits=True)          # exact f32: fine (finite logits)
    assert np.isfinite(logits).all()
    m.set_option("precision", 1)
    for chunk in (64, 2):                                             # fused first layer / separate first-layer kernel
        m.set_chunk(chunk)
        with pytest.raises(og.OpenGlottalHipError, match="f16 range"):
            m.segment(fr)
    with pytest.raises(og.OpenGlottalHipError, match="f16 range"):
        m((fr[:2].astype("float32") / 255.0)[:, None])
    import torch
    d = torch.from_numpy(fr).cuda()
    a = torch.zeros(70, dtype=torch.int32, device="cuda")
    m.segment_dev(d, 70, 32, 64, a)                                   # asynchronous entry point: reported by the sync
    with pytest.raises(og.OpenGlottalHipError, match="f16 range"):
        m.sync()
    m.sync()                                                          # the flag is cleared by the report
    ok = make_model(sd, feats, precision=1)                           # ordinary weights: no error
    ok.segment(fr)


def test_randomised_entry_point_matrix_on_chip_filling_batches():
    """The same sweep where the micro-batch fills the chip, so that the chain takes the Winograd form (k_conv_wino<1> / <2>,
    layer by layer where the map tiles by 32x16 / 16x16): frame size, batch, micro-batch, lanes, graphs, boxes, host /
    streamed / device entry points -- each configuration against the CPU oracle, and at least one of them must really have
    run the Winograd kernels (its logits differ from the direct form's in the last bits)."""
    import torch

    from oracle import unet_oracle as O
    from openglottal_amd.utils import normalize_box

    rs = np.random.RandomState(4242)
    feats = (32, 64)
    sd = synth.make_unet_state_dict(feats, seed=78, head_scale=2.5, head_bias=-0.2)
    m = make_model(sd, feats, precision=0)
    dev = torch.device("cuda", 0)
    cache = {}
    saw_winograd = 0
    for it in range(24):
        H, W = [(32, 64), (64, 64), (64, 32), (96, 64), (32, 160), (40, 48)][rs.randint(6)]
        B = int(rs.choice([130, 160, 256, 300]))
        key = (H, W, B)
        if key not in cache:
            fr = rs.randint(0, 256, (B, H, W), dtype=np.uint8)
            cache[key] = (fr,) + O.segment_frames(sd, fr, backend="torch")
        fr, ref_mask, ref_logits = cache[key]
        m.set_chunk(int(rs.choice([128, 256, 512])))
        m.set_option("lanes", int(rs.randint(4)))
        m.set_graphs(bool(rs.randint(2)))
        m.set_option("stream", int(rs.randint(2)))
        boxes = None
        if rs.randint(2):
            boxes = np.array([normalize_box((int(rs.randint(-5, W)), int(rs.randint(-5, H)), int(rs.randint(0, W + 9)), int(rs.randint(0, H + 9))), W, H)
                              for _ in range(B)], np.int32)
            boxes[rs.randint(B)] = -1
        how = int(rs.randint(3))
        if how == 0:
            masks, areas, logits = m.segment(fr, boxes=boxes, want_logits=True)
        elif how == 1:
            masks, areas = m.segment_stream(fr, boxes=boxes, want_mask=True)
            logits = None
        else:
            d_f = torch.from_numpy(fr).to(dev)
            d_a = torch.zeros(B, dtype=torch.int32, device=dev)
            d_m = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
            d_l = torch.zeros((B, H, W), dtype=torch.float32, device=dev)
            m.segment_dev(d_f, B, H, W, d_a, boxes_dev=None if boxes is None else torch.from_numpy(boxes).to(dev), mask_dev=d_m, logits_dev=d_l)
            m.sync()
            masks, areas, logits = d_m.cpu().numpy(), d_a.cpu().numpy(), d_l.cpu().numpy()
        cfg = (it, H, W, B, how)
        scale = max(1.0, np.abs(ref_logits).max())
        if logits is not None:
            assert np.abs(logits - ref_logits).max() <= TOL * scale, cfg
            if it % 4 == 0:
                m.set_option("wino", 0)
                _, _, logits_d = m.segment(fr, want_logits=True)
                m.set_option("wino", 1)
                assert np.abs(logits_d - ref_logits).max() <= TOL * scale, cfg
                saw_winograd += int(not np.array_equal(logits_d, logits))
        diff = (masks > 0) != (ref_mask > 0)
        assert np.all(np.abs(ref_logits[diff]) <= TOL * scale), cfg
        for i in range(B):
            if boxes is None:
                want = int((masks[i] > 0).sum())
            elif boxes[i][0] < 0:
                want = 0
            else:
                x1, y1, x2, y2 = boxes[i]
                want = int((masks[i][y1:y2, x1:x2] > 0).sum())
            assert int(areas[i]) == want, cfg + (i,)
    assert saw_winograd >= 1
