"""Deterministic synthetic weights and frames.

No trained weights and no GIRAFE/BAGLS data ship with the reference snapshot
(`/root/reference/.MISSING_LARGE_BLOBS`), and there is no network.  Everything
the tests, the golden-vector generator and ``bench.py`` feed to the U-Net is
therefore synthesised here from ``numpy.random.RandomState`` streams, which
are bit-stable across machines and numpy versions, so the GPU box regenerates
exactly the tensors the fixtures under ``tests/golden/`` were captured with
(no 31 MB weight file has to travel).

State-dict layout follows the reference module tree
(`openglottal/models/unet.py:50-72`): ``downs.{i}.net.{0,1,3,4}``,
``bottleneck.net.*``, ``ups.{0,2,4,6}`` (ConvTranspose2d, with bias),
``ups.{1,3,5,7}.net.*`` and ``head``.
"""

from __future__ import annotations

import numpy as np

DEFAULT_FEATURES = (32, 64, 128, 256)


def _conv_w(rs: np.random.RandomState, co: int, ci: int, k: int) -> np.ndarray:
    # He-uniform keeps activation magnitudes O(1) through 18 conv+ReLU layers,
    # so no layer degenerates to all-zero (which would hide indexing bugs).
    bound = np.sqrt(6.0 / (ci * k * k))
    return rs.uniform(-bound, bound, size=(co, ci, k, k)).astype(np.float32)


def _bn(rs: np.random.RandomState, c: int, prefix: str, sd: dict) -> None:
    # Non-trivial affine + running stats: default BN (γ=1, β=0, μ=0, σ²=1) is
    # identity-ish and would hide scale/shift folding mistakes.
    sd[prefix + ".weight"] = rs.uniform(0.8, 1.2, size=c).astype(np.float32)
    sd[prefix + ".bias"] = rs.uniform(-0.1, 0.1, size=c).astype(np.float32)
    sd[prefix + ".running_mean"] = rs.uniform(-0.1, 0.1, size=c).astype(np.float32)
    sd[prefix + ".running_var"] = rs.uniform(0.5, 1.5, size=c).astype(np.float32)
    sd[prefix + ".num_batches_tracked"] = np.array(100, dtype=np.int64)


def _double_conv(rs, prefix: str, ci: int, co: int, sd: dict) -> None:
    sd[prefix + ".net.0.weight"] = _conv_w(rs, co, ci, 3)
    _bn(rs, co, prefix + ".net.1", sd)
    sd[prefix + ".net.3.weight"] = _conv_w(rs, co, co, 3)
    _bn(rs, co, prefix + ".net.4", sd)


def make_unet_state_dict(
    features=DEFAULT_FEATURES,
    in_ch: int = 1,
    out_ch: int = 1,
    seed: int = 20260227,
    head_scale: float = 1.0,
    head_bias: float | None = None,
) -> dict[str, np.ndarray]:
    """Seeded numpy state_dict with the reference's key names and shapes.

    ``head_scale``/``head_bias`` let a caller re-apply a head calibration that
    was measured once with the reference model (see ``tests/golden``): random
    weights otherwise put every logit on one side of zero and every mask is
    all-0 or all-255.
    """
    rs = np.random.RandomState(seed)
    sd: dict[str, np.ndarray] = {}
    ch = in_ch
    for i, f in enumerate(features):
        _double_conv(rs, f"downs.{i}", ch, f, sd)
        ch = f
    _double_conv(rs, "bottleneck", ch, ch * 2, sd)
    for j, f in enumerate(reversed(features)):
        bound = np.sqrt(3.0 / (f * 2))
        sd[f"ups.{2 * j}.weight"] = rs.uniform(-bound, bound, size=(f * 2, f, 2, 2)).astype(np.float32)
        sd[f"ups.{2 * j}.bias"] = rs.uniform(-0.05, 0.05, size=f).astype(np.float32)
        _double_conv(rs, f"ups.{2 * j + 1}", f * 2, f, sd)
    bound = np.sqrt(3.0 / features[0])
    hw = rs.uniform(-bound, bound, size=(out_ch, features[0], 1, 1)).astype(np.float32)
    hb = rs.uniform(-0.05, 0.05, size=out_ch).astype(np.float32)
    sd["head.weight"] = (hw * np.float32(head_scale)).astype(np.float32)
    sd["head.bias"] = hb if head_bias is None else np.full(out_ch, head_bias, dtype=np.float32)
    return sd


def state_dict_to_torch(sd: dict[str, np.ndarray]):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


# ── Frames ───────────────────────────────────────────────────────────────────


def bench_frame_bgr(i: int, h: int = 256, w: int = 256) -> np.ndarray:
    """Seeded counterpart of `scripts/benchmark_video_speed.py:69`."""
    return np.random.RandomState(1234 + i).randint(0, 256, (h, w, 3), dtype=np.uint8)


def random_gray_frames(n: int, h: int = 256, w: int = 256, seed: int = 7) -> np.ndarray:
    """``[n,h,w]`` uint8 noise frames (one RandomState stream; fast)."""
    return np.random.RandomState(seed).randint(0, 256, (n, h, w), dtype=np.uint8)


def bulk_gray_frames(n: int, h: int = 256, w: int = 256, seed: int = 1234) -> np.ndarray:
    """Large synthetic grayscale 'video' for throughput runs (PCG64, ~1 GB/s)."""
    return np.random.default_rng(seed).integers(0, 256, size=(n, h, w), dtype=np.uint8)


def glottis_frames(
    n_patients: int = 4, frames_per_patient: int = 20, h: int = 256, w: int = 256, seed: int = 99
) -> tuple[np.ndarray, np.ndarray]:
    """Structured stand-in for the 80-frame GIRAFE test split.

    Bright textured background with a dark rotated ellipse whose opening
    oscillates per "patient".  Returns ``(gray [N,h,w] u8, gt [N,h,w] u8 {0,255})``.
    """
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    frames, gts = [], []
    for _ in range(n_patients):
        bg = rs.uniform(120, 220)
        dark = rs.uniform(20, 60)
        cx0, cy0 = w / 2 + rs.uniform(-20, 20), h / 2 + rs.uniform(-20, 20)
        ang = rs.uniform(-0.4, 0.4)
        a_max, b_len = rs.uniform(6, 12), rs.uniform(25, 60)
        period = rs.uniform(16, 30)
        phase = rs.uniform(0, 2 * np.pi)
        for t in range(frames_per_patient):
            a = max(0.0, a_max * 0.5 * (1 + np.sin(2 * np.pi * t / period + phase)) - 1.0)
            cx, cy = cx0 + rs.uniform(-1, 1), cy0 + rs.uniform(-1, 1)
            xr = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
            yr = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
            inside = (xr / max(a, 1e-6)) ** 2 + (yr / b_len) ** 2 <= 1.0 if a > 0 else np.zeros((h, w), bool)
            img = bg + rs.normal(0, 10, size=(h, w))
            img = np.where(inside, dark + rs.normal(0, 5, size=(h, w)), img)
            frames.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
            gts.append(inside.astype(np.uint8) * 255)
    return np.stack(frames), np.stack(gts)


def degraded_glottis_frames(n_patients: int = 2, n_frames: int = 12, seed: int = 4242) -> tuple[np.ndarray, np.ndarray]:
    """Glottis frames a trained net is UNSURE about: contrast reduced to 25-75 % and Gaussian noise of sigma 12-32 added.
    The full-width trained fixture (tests/golden/unet_trained_full.npz) is evaluated on them as well: Dice anywhere between 0
    and 1 and logit margins down to 1e-3, i.e. the boundary-pixel situation of a real recording rather than the clean
    synthetic frames' margins of 0.1 and more.  Returns (frames u8 [N,256,256], gt u8)."""
    fr, gt = glottis_frames(n_patients, n_frames, seed=seed)
    rs = np.random.RandomState(seed + 1)
    out = np.empty_like(fr)
    for i, f in enumerate(fr):
        x = f.astype(np.float32)
        m = x.mean()
        c = 0.25 + 0.5 * rs.rand()
        x = (x - m) * c + m + rs.normal(0.0, 12.0 + 20.0 * rs.rand(), x.shape)
        out[i] = np.clip(np.rint(x), 0, 255).astype(np.uint8)
    return out, gt


def detuned_weights(sd: dict[str, np.ndarray], amplitude: float = 0.8, seed: int = 31337) -> dict[str, np.ndarray]:
    """A DE-TUNED copy of the trained fixture's net (tests/golden/unet_trained_full.npz), derived without storing a byte: every
    convolution-kernel element is multiplied by 1 + amplitude * u, u uniform in (-1, 1) from a seeded generator, in float64, rounded
    once to float32.  Two things come with it.  (1) FULL f32 MANTISSAS: the stored kernels are float16-representable, so the Winograd
    weight transform G g G^T is (nearly) exact in f32 for them, which it is not for a real float32 checkpoint
    (scripts/train_unet.py:204-208).  (2) A net that is UNSURE: the trained net's logits jump by several units from one pixel to the
    next (a few pixels below 1e-2 in a hundred frames); the de-tuned one hovers near zero over whole regions -- thousands of pixels
    with |logit| < 1e-2, hundreds below 1e-3, a dozen inside the reference's own noise band -- which is where another summation
    order flips mask pixels and changes area integers.  Regenerated identically by the fixture's generator (which feeds it to the
    reference, tests/golden/gen_golden.py) and by the tests (which feed it to the oracle and to this library)."""
    out = {}
    for i, k in enumerate(sorted(sd)):
        v = sd[k]
        if v.ndim >= 2 and v.dtype != np.int64:
            u = np.random.RandomState(seed + i).uniform(-1.0, 1.0, size=v.shape)
            out[k] = (v.astype(np.float64) * (1.0 + u * amplitude)).astype(np.float32)
        else:
            out[k] = v.astype(np.float32) if v.dtype == np.float16 else v
    return out


def full128_frames() -> tuple[np.ndarray, np.ndarray]:
    """The 128 frames of the bench-configuration fixture (tests/golden/unet_full128.npz): the 80-frame structured
    GIRAFE stand-in (4 "patients" x 20 frames, ``glottis_frames(4, 20, seed=99)``) followed by frames 0..47 of the
    throughput stream (``bench_frame_bgr(i)`` = ``RandomState(1234+i)``) through BGR->gray.
    Returns ``(gray [128,256,256] u8, gt [80,256,256] u8)``."""
    from .utils import bgr_to_gray

    glot, gt = glottis_frames(4, 20, seed=99)
    stream = np.stack([bgr_to_gray(bench_frame_bgr(i)) for i in range(48)])
    return np.concatenate([glot, stream], axis=0), gt


BAGLS_SIZES_WH = ((256, 256), (512, 256), (512, 128), (352, 208))   # "256×256, 512×256, 512×128, 352×208" (scripts/eval_bagls.py:3)


def bagls_standin(n: int = 3500, seed: int = 2020, bgr: bool = True):
    """Stand-in for the BAGLS test split (3 500 frames of mixed sizes, each with a binary GT mask; the data set itself is
    absent): dark rotated ellipse on a bright textured background as ``glottis_frames``, sizes drawn from
    ``BAGLS_SIZES_WH`` (every fourth frame transposed to portrait), every 9th frame without a glottis (empty GT).
    Returns ``(frames list of [H,W,3] (or [H,W]) u8, gts list of [H,W] u8 {0,255})``."""
    rs = np.random.RandomState(seed)
    frames, gts = [], []
    for i in range(n):
        w, h = BAGLS_SIZES_WH[rs.randint(len(BAGLS_SIZES_WH))]
        if i % 4 == 3:
            w, h = h, w
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        bg, dark = rs.uniform(120, 220), rs.uniform(20, 60)
        cx, cy = w / 2 + rs.uniform(-0.15, 0.15) * w, h / 2 + rs.uniform(-0.15, 0.15) * h
        ang = rs.uniform(-0.5, 0.5)
        a, b_len = rs.uniform(0.01, 0.05) * min(h, w) + 1.0, rs.uniform(0.1, 0.3) * min(h, w)
        xr = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        yr = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        inside = ((xr / a) ** 2 + (yr / b_len) ** 2 <= 1.0) if i % 9 != 8 else np.zeros((h, w), bool)
        img = np.where(inside, dark + rs.normal(0, 5, size=(h, w)), bg + rs.normal(0, 10, size=(h, w)))
        g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if bgr:
            g = np.clip(g[..., None].astype(np.int32) + rs.randint(-6, 7, size=(h, w, 3)), 0, 255).astype(np.uint8)
        frames.append(g)
        gts.append(inside.astype(np.uint8) * 255)
    return frames, gts


# ── YOLOv8 detector weights (ultralytics state_dict keys) ────────────────────


def make_yolov8_state_dict(seed: int = 7, nc: int = 1, width: float = 0.25, depth: float = 0.33,
                           max_channels: int = 1024, cls_bias: float = 0.0, fused: bool = False) -> dict[str, np.ndarray]:
    """Random-init YOLOv8 (default: scale ``n``) under ultralytics' key names.

    The reference's detector weights (`weights/openglottal_yolo.pt`) are missing from the snapshot
    and are an ultralytics pickle anyway; this produces a flat ``{key: ndarray}`` of the same
    architecture (yolov8.yaml: widths ×``width`` capped at ``max_channels``, C2f repeats ×``depth``)
    so the detector path can be exercised and timed.  ``fused=True`` emits conv.weight/conv.bias
    (what ``model.fuse()`` leaves) instead of BatchNorm tensors.
    """
    rs = np.random.RandomState(seed)
    sd: dict[str, np.ndarray] = {}

    def ch(c):
        return int(np.ceil(min(c, max_channels) * width / 8) * 8)

    def rep(n):
        return max(round(n * depth), 1)

    def conv(p, c1, c2, k):
        bound = np.sqrt(6.0 / (c1 * k * k))
        sd[p + ".conv.weight"] = rs.uniform(-bound, bound, (c2, c1, k, k)).astype(np.float32)
        if fused:
            sd[p + ".conv.bias"] = rs.uniform(-0.1, 0.1, c2).astype(np.float32)
            return
        sd[p + ".bn.weight"] = rs.uniform(0.8, 1.2, c2).astype(np.float32)
        sd[p + ".bn.bias"] = rs.uniform(-0.1, 0.1, c2).astype(np.float32)
        sd[p + ".bn.running_mean"] = rs.uniform(-0.1, 0.1, c2).astype(np.float32)
        sd[p + ".bn.running_var"] = rs.uniform(0.5, 1.5, c2).astype(np.float32)
        sd[p + ".bn.num_batches_tracked"] = np.array(10, dtype=np.int64)

    def c2f(p, c1, c2, n):
        c = c2 // 2
        conv(p + ".cv1", c1, 2 * c, 1)
        conv(p + ".cv2", (2 + n) * c, c2, 1)
        for j in range(n):
            conv(f"{p}.m.{j}.cv1", c, c, 3)
            conv(f"{p}.m.{j}.cv2", c, c, 3)

    c64, c128, c256, c512, c1024 = ch(64), ch(128), ch(256), ch(512), ch(1024)
    conv("model.0", 3, c64, 3)
    conv("model.1", c64, c128, 3)
    c2f("model.2", c128, c128, rep(3))
    conv("model.3", c128, c256, 3)
    c2f("model.4", c256, c256, rep(6))
    conv("model.5", c256, c512, 3)
    c2f("model.6", c512, c512, rep(6))
    conv("model.7", c512, c1024, 3)
    c2f("model.8", c1024, c1024, rep(3))
    conv("model.9.cv1", c1024, c1024 // 2, 1)
    conv("model.9.cv2", c1024 // 2 * 4, c1024, 1)
    c2f("model.12", c1024 + c512, c512, rep(3))
    c2f("model.15", c512 + c256, c256, rep(3))
    conv("model.16", c256, c256, 3)
    c2f("model.18", c256 + c512, c512, rep(3))
    conv("model.19", c512, c512, 3)
    c2f("model.21", c512 + c1024, c1024, rep(3))
    feats = (c256, c512, c1024)
    reg_max = 16
    cb = max(16, feats[0] // 4, reg_max * 4)
    cc = max(feats[0], min(nc, 100))
    for l, f in enumerate(feats):
        conv(f"model.22.cv2.{l}.0", f, cb, 3)
        conv(f"model.22.cv2.{l}.1", cb, cb, 3)
        sd[f"model.22.cv2.{l}.2.weight"] = rs.uniform(-0.1, 0.1, (4 * reg_max, cb, 1, 1)).astype(np.float32)
        sd[f"model.22.cv2.{l}.2.bias"] = rs.uniform(0.5, 1.5, 4 * reg_max).astype(np.float32)
        conv(f"model.22.cv3.{l}.0", f, cc, 3)
        conv(f"model.22.cv3.{l}.1", cc, cc, 3)
        sd[f"model.22.cv3.{l}.2.weight"] = rs.uniform(-0.3, 0.3, (nc, cc, 1, 1)).astype(np.float32)
        sd[f"model.22.cv3.{l}.2.bias"] = np.full(nc, cls_bias, dtype=np.float32)
    sd["model.22.dfl.conv.weight"] = np.arange(reg_max, dtype=np.float32).reshape(1, reg_max, 1, 1)
    return sd
