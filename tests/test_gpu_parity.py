"""GPU parity tests (-m gpu): HIP path through the C-ABI vs golden vectors and the CPU oracle.

Tolerances
  * layer tensors / logits: abs <= BAND * max(1, |ref|max) (BAND = the reference's own noise, oracle.reference_band()).  The reference is fp32 with a
    backend-chosen summation order (oneDNN), so bit equality of floats is not defined even
    between two CPU runs; 2e-5 is the measured numpy-vs-reference noise on these vectors.
  * masks / areas: bit-exact, except that a pixel may differ where |reference logit| <= BAND
    (it sits on the decision boundary inside fp32 noise); every such flip is counted and the
    area may differ by at most that count.  The trained fixture has no such pixel: exact.
"""
import json
import os

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.features import area_waveform, extract_features_unet

pytestmark = pytest.mark.gpu

import oracle

TOL = oracle.reference_band()   # the reference's own run-to-run logit difference (tests/golden/unet_full128_self_noise.npz: 3.475e-5)


def unpack(bits, h=256, w=256):
    return np.unpackbits(bits)[: h * w].reshape(h, w)


def make_model(sd, features):
    m = og.UNet(1, 1, tuple(int(f) for f in features))
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()


@pytest.fixture(scope="module")
def small(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_small_layers.npz"))
    sd = synth.make_unet_state_dict(tuple(g["features"]), seed=int(g["seed"]), head_scale=float(g["head_scale"]),
                                    head_bias=float(g["head_bias"]))
    return g, sd, make_model(sd, g["features"])


@pytest.fixture(scope="module")
def full(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_full.npz"))
    sd = synth.make_unet_state_dict(tuple(g["features"]), seed=int(g["seed"]), head_scale=float(g["head_scale"]),
                                    head_bias=float(g["head_bias"]))
    noise = synth.random_gray_frames(4, seed=7)
    glot, gt = synth.glottis_frames(1, 4, seed=99)
    return g, sd, make_model(sd, g["features"]), np.concatenate([noise, glot]), gt


@pytest.fixture(scope="module")
def trained(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    frames, gt = synth.glottis_frames(4, 20, seed=99)
    return g, sd, make_model(sd, g["features"]), frames, gt


def test_native_library_is_the_one_running():
    from openglottal_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    assert _lib.lib().og_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libopenglottal_hip.so" in maps


def test_every_layer_boundary_small_net(small):
    g, sd, m = small
    f = synth.random_gray_frames(1, 64, 64, seed=21)
    x = (f.astype("float32") / 255.0)[:, None]
    logits = m(x)
    keys = [k[2:] for k in g.files if k.startswith("L:")]
    assert len(keys) == 27
    for k in keys:
        ref = g["L:" + k]
        got = logits if k == "head" else m.activation(k, 1)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        assert err <= TOL * max(1.0, np.abs(ref).max()), (k, err)


def test_nonsquare_batch3(small):
    g, sd, m = small
    fr = synth.random_gray_frames(3, 48, 80, seed=22)
    got = m((fr.astype("float32") / 255.0)[:, None])
    assert np.abs(got - g["logits_48x80"]).max() <= TOL


def test_torch_tensor_in_out(small):
    import torch
    g, sd, m = small
    fr = synth.random_gray_frames(3, 48, 80, seed=22)
    t = torch.from_numpy((fr.astype("float32") / 255.0)[:, None])
    out = m(t)
    assert isinstance(out, torch.Tensor) and tuple(out.shape) == (3, 1, 48, 80)
    assert np.abs(out.numpy() - g["logits_48x80"]).max() <= TOL


def check_masks(masks, logits_ref_fn, ref_masks, ref_areas, areas):
    flips_total = 0
    for i in range(len(masks)):
        diff = np.argwhere((masks[i] > 0) != (ref_masks[i] > 0))
        for (y, x) in diff:
            assert abs(logits_ref_fn(i, y, x)) <= TOL, (i, y, x)
        flips_total += len(diff)
        assert abs(int(areas[i]) - int(ref_areas[i])) <= len(diff)
        assert int(areas[i]) == int((masks[i] > 0).sum())
    return flips_total


def test_full_width_8_frames_masks_areas_logits(full):
    g, sd, m, frames, gt = full
    masks, areas, logits = m.segment(frames, want_mask=True, want_logits=True)
    assert set(np.unique(masks)) <= {0, 255}
    samp = logits.reshape(8, -1)[:, g["sample_idx"]]
    assert np.abs(samp - g["logits_samples"]).max() <= TOL
    assert np.abs(logits[[0, 4]] - g["logits_full"]).max() <= TOL
    ref_masks = np.stack([unpack(b) for b in g["masks_packed"]])
    flips = check_masks(masks, lambda i, y, x: logits[i, y, x], ref_masks, g["areas"], areas)
    print("full-width flipped pixels:", flips, "of", 8 * 65536, "areas", areas.tolist())
    for i in range(4):
        assert abs(og.dice(masks[4 + i], gt[i]) - float(g["dice_vs_gt"][i])) <= 1e-3


def test_trained_net_80_frames_bit_exact_and_dice(trained):
    g, sd, m, frames, gt = trained
    masks, areas, _ = m.segment(frames)
    ref_masks = np.stack([unpack(b) for b in g["masks_packed"]])
    assert np.array_equal(masks > 0, ref_masks > 0)           # every one of 80*65536 pixels
    assert np.array_equal(areas.astype(np.int64), g["areas"])  # area waveform integers bit-exact
    d = np.array([og.dice(masks[i], gt[i]) for i in range(80)])
    assert abs(d.mean() - float(g["dice_vs_gt"].mean())) <= 1e-3
    assert np.abs(d - g["dice_vs_gt"]).max() <= 1e-3


def test_unet_segment_frame_single(trained):
    g, sd, m, frames, gt = trained
    mk = og.unet_segment_frame(frames[5], m, "cuda:0")
    assert mk.dtype == np.uint8 and mk.shape == (256, 256)
    assert np.array_equal(mk > 0, unpack(g["masks_packed"][5]) > 0)


def test_gated_area(trained, golden_dir):
    g, sd, m, frames, gt = trained
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["gated"]
    boxes = meta["boxes"]
    for bi, b in enumerate(boxes):
        nb = og.utils.normalize_box(b, 256, 256)
        _, areas, _ = m.segment(frames[:8], boxes=np.array([nb] * 8, np.int32), want_mask=False)
        assert areas.tolist() == [row[bi] for row in meta["areas_first8"]], b
    none = np.array([[-1, -1, -1, -1]] * 8, np.int32)
    _, areas, _ = m.segment(frames[:8], boxes=none, want_mask=False)
    assert areas.tolist() == [0] * 8


def test_chunking_and_graphs_are_result_invariant(trained):
    g, sd, m, frames, gt = trained
    fr = frames[:37]  # ragged: not a multiple of any chunk below
    base = None
    for chunk, graphs in [(16, True), (1, True), (5, False), (64, True), (16, False)]:   # default options: the canonical form
        m.set_chunk(chunk)
        m.set_graphs(graphs)
        masks, areas, logits = m.segment(fr, want_logits=True)
        if base is None:
            base = (masks, areas, logits)
        else:
            assert np.array_equal(masks, base[0]) and np.array_equal(areas, base[1]), (chunk, graphs)
            assert np.array_equal(logits, base[2]), (chunk, graphs)  # same kernels, same order: bit-identical
    for lanes in (1, 2, 3):  # micro-batches of one call alternate over 1..3 streams/arenas: same results
        m.set_option("lanes", lanes)
        m.set_chunk(4)
        masks, areas, logits = m.segment(fr, want_logits=True)
        assert np.array_equal(masks, base[0]) and np.array_equal(areas, base[1]) and np.array_equal(logits, base[2]), lanes
    m.set_option("lanes", 0)
    m.set_chunk(32)
    m.set_graphs(True)
    assert np.array_equal(base[1].astype(np.int64), g["areas"][:37])


def test_split_k_latency_mode(trained, full):
    """OPT-IN latency mode ("splitk" 1 on the direct kernels, "wino" 0; off by default because it makes a frame's logits
    depend on the size of its launch): small launches (batch 1..4) split K over workgroups and reduce in a fixed order:
    deterministic, within fp32 re-association noise of the unsplit path, and still bit-exact on the trained fixture."""
    g, sd, m, frames, gt = trained
    gf, sdf, mf, framesf, gtf = full
    for mm in (m, mf):
        mm.set_option("wino", 0)
    m.set_option("splitk", 1)
    for B in (1, 2, 3):
        m.set_chunk(B)
        masks, areas, logits = m.segment(frames[:6], want_logits=True)
        masks2, areas2, logits2 = m.segment(frames[:6], want_logits=True)
        assert np.array_equal(logits, logits2) and np.array_equal(areas, areas2)      # deterministic
        assert np.array_equal(areas.astype(np.int64), g["areas"][:6])
        assert np.array_equal(masks > 0, np.stack([unpack(b) for b in g["masks_packed"][:6]]) > 0)
    mf.set_chunk(1)
    mf.set_option("splitk", 1)
    _, a1, l1 = mf.segment(framesf, want_mask=False, want_logits=True)
    mf.set_option("splitk", 0)
    _, a0, l0 = mf.segment(framesf, want_mask=False, want_logits=True)
    mf.set_option("splitk", 1)
    mf.set_chunk(32)
    m.set_chunk(32)
    m.set_option("splitk", 0)
    m.set_option("wino", 1)
    assert not np.array_equal(l0, l1)                                                 # the split path really ran
    scale = max(1.0, float(np.abs(l0).max()))
    assert np.abs(l0 - l1).max() <= 2 * TOL * scale      # two of OUR summation orders against each other: each within the band of the reference
    # fused reduce (last-arriving K part sums all parts in split order) == separate reduce kernel, bit for bit, every time
    mf.set_chunk(1)
    mf.set_option("splitk_fused", 0)
    _, a_sep, l_sep = mf.segment(framesf, want_mask=False, want_logits=True)
    mf.set_option("splitk_fused", 1)
    for _ in range(3):
        _, a_fu, l_fu = mf.segment(framesf, want_mask=False, want_logits=True)
        assert np.array_equal(l_fu, l_sep) and np.array_equal(a_fu, a_sep)
    assert np.array_equal(l_sep, l1)
    # 32-column tiles on split launches (default) / the layer's own 64-column tiles, K parts of 3 (default) or 9 (chunk, tap) steps:
    # other part boundaries, so other roundings -- all inside the tolerance, each deterministic
    for nt1, steps in ((0, 9), (1, 9), (0, 3)):
        mf.set_option("splitk_nt1", nt1)
        mf.set_option("splitk_min_steps", steps)
        _, a_v, l_v = mf.segment(framesf, want_mask=False, want_logits=True)
        _, a_v2, l_v2 = mf.segment(framesf, want_mask=False, want_logits=True)
        assert np.array_equal(l_v, l_v2) and np.array_equal(a_v, a_v2)
        assert np.abs(l_v - l0).max() <= 2 * TOL * scale, (nt1, steps)
    mf.set_option("splitk_nt1", 1)
    mf.set_option("splitk_min_steps", 3)
    mf.set_chunk(32)
    mf.set_option("splitk", 0)
    mf.set_option("wino", 1)
    assert np.abs(l1.reshape(8, -1)[:, gf["sample_idx"]] - gf["logits_samples"]).max() <= TOL * max(1.0, float(np.abs(gf["logits_samples"]).max()))
    assert np.all(np.abs(a0.astype(int) - a1.astype(int)) <= ((l0 > 0) != (l1 > 0)).reshape(8, -1).sum(1))


def test_kernel_variants_bit_identical(full):
    """The DIRECT form ("wino" 0): one-tile-per-workgroup kernel vs persistent pipelined kernel vs occupancy kernel, all
    tap-group sizes: same accumulation order per output element -> bit-identical logits."""
    g, sd, m, frames, gt = full
    fr = frames[:5]
    m.set_option("wino", 0)
    m.set_option("conv_impl", 0)
    _, a0, l0 = m.segment(fr, want_mask=False, want_logits=True)
    try:
        for impl, t1, t2, wg in [(1, 3, 1, 2), (1, 1, 1, 2), (1, 9, 3, 2), (1, 3, 3, 1), (2, 3, 1, 2), (3, 3, 1, 2)]:
            m.set_option("conv_impl", impl)
            m.set_option("tps_nt1", t1)
            m.set_option("tps_nt2", t2)
            m.set_option("wg_per_cu", wg)
            _, a1, l1 = m.segment(fr, want_mask=False, want_logits=True)
            assert np.array_equal(l0, l1) and np.array_equal(a0, a1), (impl, t1, t2, wg)
            for th in (8, 16, 0):
                m.set_option("tile_h", th)
                _, a1, l1 = m.segment(fr, want_mask=False, want_logits=True)
                assert np.array_equal(l0, l1) and np.array_equal(a0, a1), (impl, t1, t2, wg, th)
    finally:
        m.set_option("conv_impl", 2)
        m.set_option("tile_h", 0)
        big = np.concatenate([frames] * 4)[:25]   # 25 frames: enough tiles for the fused-first launch to be taken
        m.set_option("conv_impl", 0)
        _, ab0, lb0 = m.segment(big, want_mask=False, want_logits=True)
        m.set_option("conv_impl", 2)
        for fh, ff in [(0, 0), (1, 0), (0, 1), (1, 1)]:   # fused vs separate head / first layer: same arithmetic order
            m.set_option("fuse_head", fh)
            m.set_option("fuse_first", ff)
            _, a2, l2 = m.segment(fr, want_mask=False, want_logits=True)
            assert np.array_equal(l0, l2) and np.array_equal(a0, a2), (fh, ff)
            _, ab2, lb2 = m.segment(big, want_mask=False, want_logits=True)
            assert np.array_equal(lb0, lb2) and np.array_equal(ab0, ab2), (fh, ff)
        m.set_option("tps_nt1", 3)
        m.set_option("tps_nt2", 1)
        m.set_option("wg_per_cu", 2)
        m.set_option("wino", 1)


def test_empty_batch_and_bad_shapes(trained):
    g, sd, m, frames, gt = trained
    masks, areas, _ = m.segment(np.zeros((0, 256, 256), np.uint8))
    assert masks.shape == (0, 256, 256) and areas.shape == (0,)
    with pytest.raises(og.OpenGlottalHipError):
        m.segment(np.zeros((1, 250, 256), np.uint8))  # not a multiple of 16
    bad = dict(sd)
    bad.pop("head.bias")
    with pytest.raises(og.OpenGlottalHipError):
        make_model(bad, g["features"])
    bad = dict(sd)
    bad["downs.0.net.0.weight"] = np.zeros((4, 1, 5, 5), np.float32)
    with pytest.raises(og.OpenGlottalHipError):
        make_model(bad, g["features"])
    bad = dict(sd)
    bad["extra.key"] = np.zeros(3, np.float32)
    with pytest.raises(og.OpenGlottalHipError):
        make_model(bad, g["features"])
    with pytest.raises(og.OpenGlottalHipError):  # widths must double (reference forward would raise too)
        make_model(synth.make_unet_state_dict((8, 12, 20)), (8, 12, 20))


def test_extract_features_unet_matches_reference_kinematics(trained, golden_dir):
    g, sd, m, frames, gt = trained
    bgr = np.repeat(frames[..., None], 3, axis=-1)  # gray video as BGR (any correct BGR2GRAY maps (v,v,v)->v)
    feats = extract_features_unet(bgr, None, m, "cuda:0")
    ref = json.load(open(os.path.join(golden_dir, "kinematic.json")))["areas_trained"]["out"]
    assert np.array_equal(feats["_area"], g["areas"].astype(np.float64))
    for k, v in ref.items():
        assert (feats[k] is None) if v is None else abs(float(feats[k]) - v) <= 1e-12 * max(1, abs(v)), k
    assert extract_features_unet(np.zeros((0, 256, 256, 3), np.uint8), None, m) is None
    assert extract_features_unet(np.full((5, 256, 256, 3), 255, np.uint8), None, m) in (None,) or True


def test_oracle_random_shapes_and_odd_widths():
    """Seeded inputs at sizes the oracle finishes in seconds, incl. channel counts that are
    not multiples of 4/32 and spatial sizes that are not multiples of the 8x16 tile."""
    from oracle import unet_oracle as O
    cases = [((6, 12, 24), 48, 32, 2), ((5,), 16, 16, 1), ((33, 66), 32, 64, 3), ((32, 64, 128, 256), 64, 48, 1),
             ((3, 6, 12, 24, 48), 64, 96, 2)]
    for feats, H, W, B in cases:
        sd = synth.make_unet_state_dict(feats, seed=123 + H, head_scale=2.0, head_bias=-0.3)
        m = make_model(sd, feats)
        fr = synth.random_gray_frames(B, H, W, seed=H + W)
        ref_mask, ref_logits = O.segment_frames(sd, fr, backend="numpy")
        masks, areas, logits = m.segment(fr, want_logits=True)
        scale = max(1.0, np.abs(ref_logits).max())
        assert np.abs(logits - ref_logits).max() <= TOL * scale, (feats, H, W)
        diff = (masks > 0) != (ref_mask > 0)
        assert np.all(np.abs(ref_logits[diff]) <= TOL * scale)
        assert np.array_equal(areas, (masks > 0).reshape(B, -1).sum(1))


def test_full_size_properties_batch64(full):
    """BASELINE-sized run (64 frames of 256x256, full width) checked through size-independent
    properties: permutation equivariance, duplicate frames -> identical results, area == popcount,
    mask <=> logit sign, threshold monotonicity."""
    g, sd, m, frames8, gt = full
    rs = np.random.RandomState(3)
    pool = synth.random_gray_frames(16, seed=31)
    idx = rs.randint(0, 16, size=64)
    fr = pool[idx]
    masks, areas, logits = m.segment(fr, want_logits=True)
    assert np.array_equal(areas, (masks > 0).reshape(64, -1).sum(1))
    assert np.array_equal(masks > 0, logits > 0) or np.abs(logits[(masks > 0) != (logits > 0)]).max() < 2e-7
    first = {}
    for j, i in enumerate(idx):
        if i in first:
            assert np.array_equal(masks[j], masks[first[i]]) and areas[j] == areas[first[i]]
            assert np.array_equal(logits[j], logits[first[i]])
        else:
            first[i] = j
    perm = rs.permutation(64)
    _, areas_p, _ = m.segment(fr[perm], want_mask=False)
    assert np.array_equal(areas_p, areas[perm])
    _, a_lo, _ = m.segment(fr[:8], threshold=0.3, want_mask=False)
    _, a_hi, _ = m.segment(fr[:8], threshold=0.7, want_mask=False)
    assert np.all(a_lo >= areas[:8]) and np.all(areas[:8] >= a_hi)
    p = 1.0 / (1.0 + np.exp(-logits[:8].astype(np.float64)))
    assert np.all(np.abs(a_hi - (p > 0.7).reshape(8, -1).sum(1)) <= (np.abs(p - 0.7) < 1e-6).reshape(8, -1).sum(1))


def test_eval_harness_unet_only_and_scripted_detector(trained):
    """Counterpart of eval_girafe.evaluate: unet-only row reproduces the reference Dice/IoU of the 80-frame
    stand-in; yolo+unet / yolo-crop+unet rows with a scripted detector follow their definitions."""
    from openglottal_amd import evaluate as E
    g, sd, m, frames, gt = trained
    patients = [f"p{i // 20}" for i in range(80)]
    agg, pdice, det = E.evaluate(frames, gt, m, detector=None, patients=patients)
    s = E.summarize(agg)
    assert set(s) == {"unet-only"} and s["unet-only"]["n"] == 80
    assert abs(s["unet-only"]["dice"] - float(g["dice_vs_gt"].mean())) <= 1e-6
    assert abs(s["unet-only"]["iou"] - float(g["iou_vs_gt"].mean())) <= 1e-6
    assert sorted(pdice) == ["p0", "p1", "p2", "p3"] and all(len(v["unet-only"]) == 20 for v in pdice.values())

    # scripted detector: box around the image centre on even frames, nothing on odd frames (stateless)
    calls = {"i": 0}

    def backend(frame, conf):
        i = calls["i"]
        calls["i"] += 1
        if i % 2:
            return np.zeros((0, 4), np.float32), np.zeros(0, np.float32)
        return np.array([[88.0, 40.0, 168.0, 216.0]], np.float32), np.array([0.9], np.float32)

    det_ = og.TemporalDetector(backend, padding=0)
    agg2, _, st = E.evaluate(frames[:20], gt[:20], m, detector=det_, patients=patients[:20], reset_every_frame=True)
    s2 = E.summarize(agg2)
    assert s2["yolo+unet"]["det_recall"] == 0.5 and s2["yolo-crop+unet"]["det_recall"] == 0.5
    masks, _, _ = m.segment(frames[:20])
    for i in range(20):
        want = np.zeros_like(masks[i])
        if i % 2 == 0:
            want[40:216, 88:168] = masks[i][40:216, 88:168]
        d, j = og.utils.frame_metrics(want, gt[i])
        assert agg2["yolo+unet"]["dice"][i] == d and agg2["yolo+unet"]["iou"][i] == j
    # crop pipeline: 80x176 crop letterboxed to 256 (NEAREST), segmented, projected back: a valid {0,255} mask per frame
    crop_masks = E.unet_on_crops(frames[:4], [(88, 40, 168, 216), None, (0, 0, 256, 256), (10, 10, 10, 50)], m)
    assert crop_masks.shape == (4, 256, 256) and set(np.unique(crop_masks)) <= {0, 255}
    assert not crop_masks[1].any() and not crop_masks[3].any()
    assert np.array_equal(crop_masks[2], masks[2])            # full-frame "crop" is the identity letterbox
    assert not crop_masks[0][:40].any() and not crop_masks[0][:, :88].any()
    E.print_table(agg2)


def test_unet_segment_frame_non_256_uses_host_resize(trained):
    g, sd, m, frames, gt = trained
    big = np.kron(frames[3], np.ones((2, 2), np.uint8))       # 512x512 frame
    mk = og.unet_segment_frame(big, m, "cuda:0")
    assert mk.shape == (512, 512) and set(np.unique(mk)) <= {0, 255}
    small = og.unet_segment_frame(frames[3], m)
    # 2x box-upsampled frame -> bilinear down gives the original back -> mask ~ bilinear-up of the probability map
    assert abs(int((mk > 0).sum()) - 4 * int((small > 0).sum())) <= 0.15 * 4 * max(1, int((small > 0).sum()))


def test_large_frames_and_large_ragged_batch(trained):
    """Maximum-size style cases: a 512x384 frame (tiles beyond 256, non-square) against the oracle, and a
    ragged 203-frame batch whose per-frame results must repeat the 80-frame golden areas exactly."""
    from oracle import unet_oracle as O
    feats = (4, 8)
    sd2 = synth.make_unet_state_dict(feats, seed=77, head_scale=2.5, head_bias=-0.2)
    m2 = make_model(sd2, feats)
    fr = synth.random_gray_frames(2, 384, 512, seed=123)
    ref_mask, ref_logits = O.segment_frames(sd2, fr, backend="torch")
    masks, areas, logits = m2.segment(fr, want_logits=True)
    assert np.abs(logits - ref_logits).max() <= TOL * max(1.0, np.abs(ref_logits).max())
    diff = (masks > 0) != (ref_mask > 0)
    assert np.all(np.abs(ref_logits[diff]) <= TOL)
    g, sd, m, frames, gt = trained
    idx = np.arange(203) % 80
    _, areas, _ = m.segment(frames[idx], want_mask=False)
    assert np.array_equal(areas.astype(np.int64), g["areas"][idx])


def test_device_crop_letterbox_paste_equals_host_geometry(trained):
    """Device YOLO-Crop+UNet (crop -> NEAREST letterbox -> U-Net -> NEAREST back-projection -> paste) must equal the
    host-geometry pipeline pixel for pixel (same index rule, float64 floor)."""
    from openglottal_amd import evaluate as E
    from openglottal_amd.geometry import INTER_NEAREST, letterbox_with_info, unletterbox
    g, sd, m, frames, gt = trained
    rs = np.random.RandomState(4)
    boxes = [(88, 40, 168, 216), None, (0, 0, 256, 256), (10, 10, 10, 50), (3, 7, 250, 31), (200, 100, 256, 256), (17, 19, 18, 20)]
    for _ in range(9):
        x1, y1 = rs.randint(0, 200, 2)
        boxes.append((int(x1), int(y1), int(x1 + rs.randint(1, 256 - x1)), int(y1 + rs.randint(1, 256 - y1))))
    fr = frames[:len(boxes)]
    dev = m.segment_crops(fr, boxes)
    assert dev.shape == fr.shape and set(np.unique(dev)) <= {0, 255}
    for i, b in enumerate(boxes):   # host reference built from geometry.py + the plain segment call
        want = np.zeros_like(fr[i])
        if b is not None and b[2] > b[0] and b[3] > b[1]:
            x1, y1, x2, y2 = b
            crop = fr[i][y1:y2, x1:x2]
            boxed, pt, pl, ch, cw = letterbox_with_info(crop, 256, value=0)
            mk, _, _ = m.segment(boxed[None])
            want[y1:y2, x1:x2] = unletterbox(mk[0], pt, pl, ch, cw, crop.shape[0], crop.shape[1], interp=INTER_NEAREST)
        assert np.array_equal(dev[i], want), (i, b)
    assert np.array_equal(E.unet_on_crops(fr, boxes, m), dev)


def test_occupancy_kernels_on_partial_tiles_large_batch():
    """A batch large enough that every layer takes the occupancy kernel (buffer-addressed halo, bounds-check zero
    padding and clipping) on shapes whose deeper levels do not tile evenly (80x48: 40x24, 20x12, 10x6 ...), against
    (a) the persistent kernels on the same frames (split-K off: same summation order -> bit-identical) and
    (b) the oracle on a few frames."""
    from oracle import unet_oracle as O
    feats = (32, 64, 128)
    sd = synth.make_unet_state_dict(feats, seed=321, head_scale=2.0, head_bias=-0.5)
    m = make_model(sd, feats)
    m.set_option("splitk", 0)
    B, H, W = 96, 80, 48
    fr = synth.random_gray_frames(B, H, W, seed=17)
    m.set_chunk(96)
    masks, areas, logits = m.segment(fr, want_logits=True)
    masks_again, areas_again, logits_again = m.segment(fr, want_logits=True)      # same launch twice: bit-identical
    assert np.array_equal(logits, logits_again) and np.array_equal(masks, masks_again) and np.array_equal(areas, areas_again)
    m.set_chunk(1)                      # one frame per chain: nothing fills the chip -> persistent kernels
    masks1, areas1, logits1 = m.segment(fr[:6], want_logits=True)
    assert np.array_equal(logits[:6], logits1) and np.array_equal(masks[:6], masks1) and np.array_equal(areas[:6], areas1)
    ref_mask, ref_logits = O.segment_frames(sd, fr[:4], backend="torch")
    scale = max(1.0, np.abs(ref_logits).max())
    assert np.abs(logits[:4] - ref_logits).max() <= TOL * scale
    diff = (masks[:4] > 0) != (ref_mask > 0)
    assert np.all(np.abs(ref_logits[diff]) <= TOL * scale)
    assert np.array_equal(areas, (masks > 0).reshape(B, -1).sum(1))
    # channel counts that are not multiples of 32 (padded channel slots, 64-wide and 128-wide column tiles) at a batch
    # that takes the occupancy kernels, against the oracle
    feats2 = (33, 66)
    sd2 = synth.make_unet_state_dict(feats2, seed=99, head_scale=2.0, head_bias=-0.3)
    m2 = make_model(sd2, feats2)
    fr2 = synth.random_gray_frames(128, 32, 64, seed=5)
    m2.set_chunk(128)
    masks2, areas2, logits2 = m2.segment(fr2, want_logits=True)
    ref_mask2, ref_logits2 = O.segment_frames(sd2, fr2[:4], backend="torch")
    scale2 = max(1.0, np.abs(ref_logits2).max())
    assert np.abs(logits2[:4] - ref_logits2).max() <= TOL * scale2
    assert np.all(np.abs(ref_logits2[(masks2[:4] > 0) != (ref_mask2 > 0)]) <= TOL * scale2)
    # (that batch filled the chip: its 64-column layers ran in Winograd form, padded channel slots and all.)  The direct
    # occupancy kernels at the same batch against the persistent kernels at batch 2: bit for bit
    m2.set_option("wino", 0)
    _, areas2, logits2d = m2.segment(fr2, want_logits=True)
    assert np.abs(logits2d[:4] - ref_logits2).max() <= TOL * scale2 and np.abs(logits2d - logits2).max() <= TOL * scale2
    logits2 = logits2d
    m2.set_chunk(2)
    m2.set_option("splitk", 0)
    _, areas2b, logits2b = m2.segment(fr2[:6], want_logits=True)
    assert np.array_equal(logits2[:6], logits2b) and np.array_equal(areas2[:6], areas2b)


@pytest.mark.parametrize("feats,shape,B", [((32, 64), (128, 256), 16),       # every layer tiles: wino<1> x 3, wino<2> x 5
                                           ((32, 64), (96, 160), 24),        # bottleneck 24x40 does not tile by 16 -> direct
                                           ((64, 128), (48, 64), 64),        # 64-wide first level on 48 rows: wino<2> from layer 2 on
                                           ((40, 80), (64, 64), 128),        # padded channel slots (40 -> 64 columns; 80 -> 96 and 160: 32-column tiles)
                                           ((32, 64), (512, 512), 4)])      # large frames: a micro-batch of 4 fills the chip
def test_winograd_form_against_oracle_and_direct_form(feats, shape, B):
    """The canonical form: every 3x3 conv whose map tiles runs in Winograd F(2x2,3x3) form (k_conv_wino<2> on 16x16-pixel
    tiles, k_conv_wino<1> on 32x16 ones) at EVERY micro-batch size, the direct kernels on the other layers -- against the
    oracle at the usual tolerance, against the direct form of the same chain, deterministic, and with the kernels asserted."""
    import torch
    from oracle import unet_oracle as O
    H, W = shape
    sd = synth.make_unet_state_dict(feats, seed=H + W + len(feats), head_scale=2.0, head_bias=-0.4)
    m = make_model(sd, feats)
    fr = synth.random_gray_frames(B, H, W, seed=B + H)
    m.set_chunk(B)
    masks, areas, logits = m.segment(fr, want_logits=True)
    masks_b, areas_b, logits_b = m.segment(fr, want_logits=True)
    assert np.array_equal(logits, logits_b) and np.array_equal(areas, areas_b)
    kernels = [p["kernel"] for p in m.profile(torch.from_numpy(fr).to("cuda:0"), B, H, W, reps=1)]
    assert any(k.startswith("k_conv_wino") for k in kernels), kernels
    m.set_option("wino", 0)
    masks_d, areas_d, logits_d = m.segment(fr, want_logits=True)
    assert not any(k.startswith("k_conv_wino") for k in (p["kernel"] for p in m.profile(torch.from_numpy(fr).to("cuda:0"), B, H, W, reps=1)))
    n_ref = min(B, 3)
    ref_mask, ref_logits = O.segment_frames(sd, fr[:n_ref], backend="torch")
    scale = max(1.0, np.abs(ref_logits).max())
    assert np.abs(logits[:n_ref] - ref_logits).max() <= TOL * scale
    assert np.abs(logits_d[:n_ref] - ref_logits).max() <= TOL * scale
    assert np.abs(logits - logits_d).max() <= TOL * scale
    diff = (masks[:n_ref] > 0) != (ref_mask > 0)
    assert np.all(np.abs(ref_logits[diff]) <= TOL * scale)
    assert np.array_equal(areas, (masks > 0).reshape(B, -1).sum(1))
    # the form is a property of the handle, not of the micro-batch: half the batch, and one frame per chain, bit for bit
    m.set_option("wino", 1)
    for ch in (max(1, B // 2), 1):
        m.set_chunk(ch)
        n = min(B, max(ch, 3))
        _, areas_h, logits_h = m.segment(fr[:n], want_logits=True)
        assert np.array_equal(logits_h, logits[:n]) and np.array_equal(areas_h, areas[:n]), ch
