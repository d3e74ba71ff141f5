"""-m gpu: a frame's mask / area / logits are a function of THE FRAME ONLY.

In the reference a frame's mask does not depend on its neighbours (`openglottal/features.py:234-238`: one
`unet_segment_frame` call per frame, no cross-frame state).  Here frames share kernel launches, so the property has to be
built: the arithmetic form of every layer (Winograd F(2x2,3x3) where the map tiles, the direct kernel elsewhere; no
split-K) is decided from the handle's options and (H, W) alone -- never from the micro-batch size, the lane, the shard or
the entry point (csrc/og_api.hip `pick_chain_form`, `launch_conv: use_wino`).

Tested where it can fail: the FULL-WIDTH (32,64,128,256) seeded net of bench.py on the seeded NOISE stream of bench.py
(frame i = RandomState(1234+i)), whose logits come within 1e-5 of zero on hundreds of pixels -- any change of summation
order flips some of them (round 2: 25 of 512 areas differed between the two forms).  Everything below is compared with
`array_equal`, logits included.  SURVEY 8(e): gathered waveform == 1-GPU waveform bit for bit, ragged N included.
"""
import os
import socket

import numpy as np
import pytest

import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.utils import bgr_to_gray

pytestmark = pytest.mark.gpu

FEATS = (32, 64, 128, 256)
N = 203   # ragged over 2 and 3 ranks, over chunk 64 (11 left) and chunk 48 (11 left)


def _model(chunk=64):
    sd = synth.make_unet_state_dict(FEATS, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)   # bench.py's net
    m = og.UNet(1, 1, FEATS)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    m.set_chunk(chunk)
    return m


def _stream(n=N, lo=0):
    return np.stack([bgr_to_gray(synth.bench_frame_bgr(lo + i)) for i in range(n)])


@pytest.fixture(scope="module")
def base():
    m = _model(64)
    fr = _stream()
    masks, areas, logits = m.segment(fr, want_logits=True)
    near = int((np.abs(logits) <= 1e-5).sum())
    assert near >= 20, near      # the workload really has pixels that any change of summation order would flip
    return m, fr, masks, areas, logits


def test_micro_batch_size_does_not_change_a_frame(base):
    """(i) chunk 64 vs 48 vs 200 (one launch for almost everything) vs 1 (one frame per chain), graphs on and off."""
    m, fr, masks, areas, logits = base
    try:
        for chunk, graphs, n in [(48, True, N), (200, True, N), (7, False, 40), (1, True, 24)]:
            m.set_chunk(chunk)
            m.set_graphs(graphs)
            mk, ar, lg = m.segment(fr[:n], want_logits=True)
            assert np.array_equal(ar, areas[:n]), (chunk, int((ar != areas[:n]).sum()))
            assert np.array_equal(mk, masks[:n]) and np.array_equal(lg, logits[:n]), chunk
    finally:
        m.set_chunk(64)
        m.set_graphs(True)


def test_position_in_the_video_does_not_change_a_frame(base):
    """(iii) C4's tail: 10 000 = 156 x 64 + 16 -- the last 16 frames of a video run in a ragged micro-batch.  Frames 128..143
    as the 16-frame tail of a 144-frame call, as the head of their own call, in the middle of another, and one by one."""
    m, fr, masks, areas, logits = base
    tail = slice(128, 144)
    _, ar_a, lg_a = m.segment(fr[:144], want_mask=False, want_logits=True)          # 64 + 64 + 16
    _, ar_b, lg_b = m.segment(fr[tail], want_mask=False, want_logits=True)          # a 16-frame call
    _, ar_c, lg_c = m.segment(fr[100:N], want_mask=False, want_logits=True)         # frames 128.. at offset 28 of a 64-frame chunk
    assert np.array_equal(ar_a[tail], areas[tail]) and np.array_equal(lg_a[tail], logits[tail])
    assert np.array_equal(ar_b, areas[tail]) and np.array_equal(lg_b, logits[tail])
    assert np.array_equal(ar_c[28:44], areas[tail]) and np.array_equal(lg_c[28:44], logits[tail])
    for i in (128, 135, 143):
        mk1 = og.unet_segment_frame(fr[i], m, "cuda:0")                              # the reference's per-frame call
        assert np.array_equal(mk1, masks[i]), i


def test_entry_point_does_not_change_a_frame(base):
    """Host batch (og_unet_segment_u8), streamed BGR (og_unet_stream_u8 + BGR->gray on the device), device pointers
    (og_unet_segment_u8_dev), the frame-loop wrapper (area_waveform), one-shot staging ("stream" 0), 1 to 3 lanes."""
    import torch

    from openglottal_amd.features import area_waveform

    m, fr, masks, areas, logits = base
    n = 150
    bgr = np.stack([synth.bench_frame_bgr(i) for i in range(n)])
    mk_s, ar_s = m.segment_stream(bgr, want_mask=True)
    assert np.array_equal(ar_s, areas[:n]) and np.array_equal(mk_s, masks[:n])
    assert np.array_equal(area_waveform(bgr, None, m).astype(np.int64), areas[:n].astype(np.int64))
    assert np.array_equal(area_waveform(list(bgr), None, m).astype(np.int64), areas[:n].astype(np.int64))
    dev = torch.device("cuda", 0)
    d_a = torch.zeros(n, dtype=torch.int32, device=dev)
    d_l = torch.zeros((n, 256, 256), dtype=torch.float32, device=dev)
    m.segment_dev(torch.from_numpy(fr[:n]).to(dev), n, 256, 256, d_a, logits_dev=d_l)
    m.sync()
    assert np.array_equal(d_a.cpu().numpy(), areas[:n]) and np.array_equal(d_l.cpu().numpy(), logits[:n])
    try:
        m.set_option("stream", 0)
        _, ar0, lg0 = m.segment(fr[:n], want_mask=False, want_logits=True)
        assert np.array_equal(ar0, areas[:n]) and np.array_equal(lg0, logits[:n])
        m.set_option("stream", 1)
        for lanes in (1, 2, 3):
            m.set_option("lanes", lanes)
            m.set_chunk(16)
            _, arl, lgl = m.segment(fr[:70], want_mask=False, want_logits=True)
            assert np.array_equal(arl, areas[:70]) and np.array_equal(lgl, logits[:70]), lanes
    finally:
        m.set_option("stream", 1)
        m.set_option("lanes", 0)
        m.set_chunk(64)


def test_same_kernels_at_every_micro_batch_size(base):
    """The chain of a one-frame launch has the FORM of a 64-frame launch, layer by layer: k_conv_first, 14 x k_conv_wino<2>,
    3 x k_conv_wino<1>, 4 transposed convs on a direct kernel without split-K (which direct variant -- occupancy or
    persistent -- is a scheduling choice: they sum in the same order, test_kernel_variants_bit_identical)."""
    import torch

    m, fr, *_ = base
    d = torch.from_numpy(fr[:64]).to("cuda:0")

    def form(kernel):
        assert "splitK" not in kernel, kernel
        if kernel.startswith(("k_conv_wino_ps<", "k_conv_wino_w<", "k_conv_wino_wp<")):     # position-split / wave-split launch of the same form (same sums)
            return "k_conv_wino<" + kernel.split("<")[1].split(",")[0].rstrip(">") + ">"
        return kernel if kernel.startswith(("k_conv_wino", "k_conv_first")) else "direct:" + kernel.split("<")[1].split(",")[1]   # MODE

    k64 = [(p["layer"], form(p["kernel"])) for p in m.profile(d, 64, 256, 256, reps=1)]
    for B in (1, 5, 16):
        kB = [(p["layer"], form(p["kernel"])) for p in m.profile(d, B, 256, 256, reps=1)]
        assert kB == k64, (B, kB)
    names = [k for _, k in k64]
    assert names.count("k_conv_wino<2>") == 14 and names.count("k_conv_wino<1>") == 3, names
    assert not any("splitK" in k for k in names), names


def test_position_split_launches_are_bit_identical(base):
    """k_conv_wino_ps (under-filled launches: a tile's 16 Winograd positions on 16 / PN workgroups, last arriver reduces)
    against k_conv_wino: every PN, one to three frames per chain, repeated (arrival order varies), logits bit for bit."""
    import torch

    m, fr, masks, areas, logits = base
    n = 12
    try:
        m.set_option("wino_w", 0)     # (round 4's default hands these launches to k_conv_wino_w / _wp: tests/test_gpu_wino_w.py)
        for ps, tag in [(0, None), (1, "auto"), (2, ",4>"), (3, ",2>"), (4, ",1>")]:
            m.set_option("wino_ps", ps)
            for chunk in (1, 3):
                m.set_chunk(chunk)
                for rep in range(2):
                    mk, ar, lg = m.segment(fr[:n], want_logits=True)
                    assert np.array_equal(lg, logits[:n]), (ps, chunk, rep, float(np.abs(lg - logits[:n]).max()))
                    assert np.array_equal(ar, areas[:n]) and np.array_equal(mk, masks[:n]), (ps, chunk, rep)
            names = [p["kernel"] for p in m.profile(torch.from_numpy(fr[:1]).to("cuda:0"), 1, 256, 256, reps=1)]
            n_ps = sum(k.startswith("k_conv_wino_ps") for k in names)
            if ps == 0:
                assert n_ps == 0, names
            elif tag == "auto":
                assert n_ps >= 10, names          # every 3x3 layer behind the first two levels is under-filled at one frame
            else:
                assert n_ps >= 10 and all(k.endswith(tag) for k in names if k.startswith("k_conv_wino_ps")), names
    finally:
        m.set_option("wino_ps", 1)
        m.set_option("wino_w", 1)
        m.set_chunk(64)


@pytest.mark.parametrize("feats,shape", [((32, 64), (128, 256)), ((32, 64), (96, 160)), ((64, 128), (48, 64)), ((40, 80), (64, 64)),
                                         ((32, 64, 128), (64, 32)), ((32, 64), (256, 128))])
def test_position_split_on_other_shapes(feats, shape):
    """The position-split launches on other widths, depths and frame shapes (maps of 1 to 10 tiles across, padded channel slots,
    32- and 64-column layers, layers that fall back to the direct kernel): every forced PN and the automatic choice against
    k_conv_wino, at one, two and five frames per chain -- logits bit for bit; and the canonical result against the oracle."""
    import torch
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
    seen = 0
    for ps in (1, 2, 3, 4):
        m.set_option("wino_ps", ps)
        for chunk in (1, 2, 5):
            m.set_chunk(chunk)
            _, a1, l1 = m.segment(fr, want_mask=False, want_logits=True)
            assert np.array_equal(l1, l0) and np.array_equal(a1, a0), (ps, chunk, float(np.abs(l1 - l0).max()))
        seen += sum(p["kernel"].startswith("k_conv_wino_ps") for p in m.profile(torch.from_numpy(fr[:1]).to("cuda:0"), 1, H, W, reps=1))
    assert seen > 0     # the split kernels really ran on this shape


def _rank(rank, world, port, q):
    import torch.distributed as dist

    from openglottal_amd.dist import sharded_area_waveform

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _model(64)
    wave = sharded_area_waveform(_stream(), m, rank, world)
    q.put((rank, wave.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_does_not_change_a_frame(base, world):
    """(ii) `sharded_area_waveform` at world 2 and 3 (ranks are processes sharing the one GPU of the box, gloo collectives;
    one rank per GPU over RCCL on a node) == world 1 == the single-process waveform, N = 203 (ragged everywhere)."""
    import torch.multiprocessing as mp

    from openglottal_amd.dist import sharded_area_waveform

    m, fr, masks, areas, logits = base
    assert np.array_equal(sharded_area_waveform(fr, m, 0, 1), areas)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(120)
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, wave in res:
        assert wave == areas.astype(np.int64).tolist(), (world, rank)
