"""-m gpu: device-side frame ingest and the gated recount on resident masks.

`og_bgr2gray_dev` must equal the host restatement `utils.bgr_to_gray` bit for bit (both restate OpenCV's published
u8 BGR2GRAY; OpenCV itself is absent: parity unpinned, exact for R = G = B by construction).  `og_mask_area_dev` must
equal numpy's `mask[y1:y2, x1:x2] > 0` count (features.py:244-245), incl. the reference-generated gated areas.
"""
import json
import os

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray, normalize_box

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def trained(golden_dir):
    g = np.load(os.path.join(golden_dir, "unet_trained_small.npz"))
    sd = {k[2:]: g[k] for k in g.files if k.startswith("W:")}
    m = og.UNet(1, 1, tuple(int(f) for f in g["features"]))
    m.load_state_dict(sd)
    frames, gt = synth.glottis_frames(4, 20, seed=99)
    return g, m.to("cuda:0").eval(), frames


def test_bgr2gray_dev_equals_host_restatement(trained):
    import torch

    g, m, frames = trained
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(12)
    for (B, H, W) in [(5, 256, 256), (3, 208, 352), (1, 1, 7), (2, 33, 65)]:
        bgr = rs.randint(0, 256, (B, H, W, 3), dtype=np.uint8)
        bgr[0, 0, :3] = [[0, 0, 0], [255, 255, 255], [255, 0, 0]][:min(3, W)]      # extremes
        d_bgr = torch.from_numpy(bgr).to(dev)
        d_gray = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
        m.bgr2gray_dev(d_bgr, B, H, W, d_gray)
        m.sync()
        assert np.array_equal(d_gray.cpu().numpy(), bgr_to_gray(bgr)), (B, H, W)
    # every (b,g,r) with b=g=r maps to itself; all 256^2 (g, r) pairs at b = 77 against the host formula
    gg, rr = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    bgr = np.stack([np.full_like(gg, 77), gg, rr], axis=-1)[None]
    d_gray = torch.zeros((1, 256, 256), dtype=torch.uint8, device=dev)
    m.bgr2gray_dev(torch.from_numpy(bgr).to(dev), 1, 256, 256, d_gray)
    m.sync()
    assert np.array_equal(d_gray.cpu().numpy(), bgr_to_gray(bgr))
    v = np.arange(256, dtype=np.uint8)
    same = np.stack([v, v, v], axis=-1)[None, None]
    d_g = torch.zeros((1, 1, 256), dtype=torch.uint8, device=dev)
    m.bgr2gray_dev(torch.from_numpy(same).to(dev), 1, 1, 256, d_g)
    m.sync()
    assert np.array_equal(d_g.cpu().numpy().ravel(), v)


def test_mask_area_dev_equals_numpy_slices_and_golden_gated(trained, golden_dir):
    import torch

    g, m, frames = trained
    dev = torch.device("cuda", 0)
    masks, areas, _ = m.segment(frames[:8])
    d_masks = torch.from_numpy(masks).to(dev)
    meta = json.load(open(os.path.join(golden_dir, "meta.json")))["gated"]
    for bi, b in enumerate(meta["boxes"]):       # reference-generated: features.py:244-245 on the reference's masks
        bx = torch.tensor([normalize_box(b, 256, 256)] * 8, dtype=torch.int32, device=dev)
        out = torch.full((8,), -7, dtype=torch.int32, device=dev)
        m.mask_area_dev(d_masks, 8, 256, 256, bx, out)
        m.sync()
        assert out.cpu().tolist() == [row[bi] for row in meta["areas_first8"]], b
    rs = np.random.RandomState(9)
    boxes = []
    for i in range(8):
        x1, y1 = rs.randint(-20, 250, 2)
        boxes.append((int(x1), int(y1), int(x1 + rs.randint(-5, 300)), int(y1 + rs.randint(-5, 300))))
    boxes[3] = None
    nb = np.array([normalize_box(b, 256, 256) for b in boxes], np.int32)
    out = torch.zeros(8, dtype=torch.int32, device=dev)
    m.mask_area_dev(d_masks, 8, 256, 256, torch.from_numpy(nb).to(dev), out)
    m.sync()
    # python slice semantics on the raw boxes (negative indices wrap, as mask[y1:y2, x1:x2] does in features.py:244-245)
    want = [0 if b is None else int((masks[i][b[1]:b[3], b[0]:b[2]] > 0).sum()) for i, b in enumerate(boxes)]
    assert out.cpu().tolist() == want
    # no boxes: the plain full-frame count
    out2 = torch.zeros(8, dtype=torch.int32, device=dev)
    m.mask_area_dev(d_masks, 8, 256, 256, None, out2)
    m.sync()
    assert out2.cpu().tolist() == areas.tolist()
    # non-square masks
    mk = (rs.rand(3, 40, 72) > 0.6).astype(np.uint8) * 255
    bx = np.array([[5, 3, 60, 33], [0, 0, 72, 40], [-1, -1, -1, -1]], np.int32)
    out3 = torch.zeros(3, dtype=torch.int32, device=dev)
    m.mask_area_dev(torch.from_numpy(mk).to(dev), 3, 40, 72, torch.from_numpy(bx).to(dev), out3)
    m.sync()
    assert out3.cpu().tolist() == [int((mk[0][3:33, 5:60] > 0).sum()), int((mk[1] > 0).sum()), 0]
