"""CPU tests of the letterbox / resize geometry (host side; OpenCV semantics restated, parity unpinned)."""
import numpy as np

from openglottal_amd import geometry as G


def test_nearest_index_rule_and_identity():
    img = np.arange(6 * 8, dtype=np.uint8).reshape(6, 8)
    assert np.array_equal(G.resize_nearest(img, 8, 6), img)
    up = G.resize_nearest(img, 16, 12)
    assert np.array_equal(up[::2, ::2], img) and np.array_equal(up[1::2, 1::2], img)   # floor(dst*0.5)
    down = G.resize_nearest(img, 4, 3)
    assert np.array_equal(down, img[::2, ::2])


def test_linear_identity_constant_and_monotone():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (20, 30), dtype=np.uint8)
    assert np.array_equal(G.resize_linear(img, 30, 20), img)
    const = np.full((17, 23), 200, np.uint8)
    assert np.all(G.resize_linear(const, 50, 40) == 200)            # fixed-point weights sum to 2^11 exactly
    ramp = np.tile(np.arange(0, 256, 8, dtype=np.uint8), (4, 1))
    out = G.resize_linear(ramp, 64, 4)
    assert np.all(np.diff(out[0].astype(int)) >= 0) and out.min() == 0 and out.max() == 248
    f = rs.rand(9, 11).astype(np.float32)
    o = G.resize_linear(f, 22, 18)
    assert o.dtype == np.float32 and o.min() >= f.min() - 1e-6 and o.max() <= f.max() + 1e-6
    bgr = rs.randint(0, 256, (10, 14, 3), dtype=np.uint8)
    o3 = G.resize_linear(bgr, 28, 20)
    for c in range(3):
        assert np.array_equal(o3[..., c], G.resize_linear(bgr[..., c], 28, 20))


def test_letterbox_roundtrip_geometry():
    rs = np.random.RandomState(1)
    for (h, w) in [(256, 256), (512, 256), (128, 512), (208, 352), (57, 91)]:
        m = (rs.rand(h, w) > 0.5).astype(np.uint8) * 255
        boxed, pt, pl, ch, cw = G.letterbox_with_info(m, 256)
        assert boxed.shape == (256, 256)
        assert max(ch, cw) == 256 and abs(ch / cw - h / w) < 0.02
        assert pt == (256 - ch) // 2 and pl == (256 - cw) // 2
        assert np.all(boxed[:pt] == 0) and np.all(boxed[:, :pl] == 0)
        same = G.letterbox_apply_geometry(m, 256, pt, pl, ch, cw)
        assert np.array_equal(same, boxed)
        back = G.unletterbox(boxed, pt, pl, ch, cw, h, w)
        assert back.shape == (h, w)
        if (h, w) == (256, 256):
            assert np.array_equal(back, m)
    img3 = rs.randint(0, 256, (100, 200, 3), dtype=np.uint8)
    b3 = G.letterbox(img3, 256, value=114)
    assert b3.shape == (256, 256, 3) and np.all(b3[0] == 114)
