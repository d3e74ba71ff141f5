"""A/B of one kernel option in ONE process, interleaved rounds; also checks results are bit-identical.
usage: ab_option.py <chunk> <option> <v0> <v1> [<v2> ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth

chunk, opt, vals = int(sys.argv[1]), sys.argv[2], [int(v) for v in sys.argv[3:]]
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(chunk)
F = 512
frames = torch.from_numpy(synth.bulk_gray_frames(F)).cuda()
area = torch.zeros(F, dtype=torch.int32, device="cuda")
logits = torch.zeros((F, 256, 256), dtype=torch.float32, device="cuda")
res = {v: [] for v in vals}
ref = None
for v in vals:
    m.set_option(opt, v)
    m.segment_dev(frames, 64, 256, 256, area, logits_dev=logits); m.sync()
    cur = (area[:64].clone(), logits[:64].clone())
    if ref is None:
        ref = cur
    else:
        print(f"{opt}={v}: areas equal {bool((cur[0] == ref[0]).all())}, logits bit-identical {bool(torch.equal(cur[1], ref[1]))}")
for rnd in range(5):
    for v in vals:
        m.set_option(opt, v)
        m.segment_dev(frames, F, 256, 256, area); m.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            m.segment_dev(frames, F, 256, 256, area)
        m.sync()
        res[v].append(3 * F / (time.perf_counter() - t0))
for v in vals:
    r = res[v]
    print(f"{opt}={v:<3d} median {np.median(r):8.1f} fps  max {max(r):8.1f}  min {min(r):8.1f}", flush=True)
for v in vals:
    m.set_option(opt, v); m.set_option("dual", 0)
    m.profile(frames, chunk, 256, 256, reps=2)
    prof = m.profile(frames, chunk, 256, 256, reps=8)
    tot = sum(p["ms"] for p in prof)
    print(f"--- {opt}={v}: chain {tot:.3f} ms")
    for p in prof:
        tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
        print(f"{p['layer']:38s} {p['kernel']:28s} {p['ms']:8.4f} ms {tf:7.1f} TF/s {100 * tf / 157.3:5.1f}%")
    m.set_option("dual", 1)
