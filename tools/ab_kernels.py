"""A/B of kernel variants in ONE process, interleaved rounds (cdna guide §5.4 rule 24)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 32
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(chunk)
F = 256
frames = torch.from_numpy(synth.bulk_gray_frames(F)).cuda()
area = torch.zeros(F, dtype=torch.int32, device="cuda")
variants = {"gen1 (one tile per WG)": dict(conv_impl=0, tile_h=8, prio_mode=0, convt_occ=0),
            "gen2 th8": dict(conv_impl=1, tile_h=8, tps_nt1=3, tps_nt2=1, wg_per_cu=2, prio_mode=0),
            "gen2 th8 prio2 (default)": dict(conv_impl=1, tile_h=8, tps_nt1=3, tps_nt2=1, wg_per_cu=2, prio_mode=2),
            "gen2 th8 1wg/cu": dict(conv_impl=1, tile_h=8, tps_nt1=3, tps_nt2=1, wg_per_cu=1, prio_mode=0),
            "occ3 (single halo buf, 3 wg/cu)": dict(conv_impl=2, tile_h=8, prio_mode=0),
            "occ3 + convT occ": dict(conv_impl=2, tile_h=8, prio_mode=0, convt_occ=1),
            "occ th16 + convT occ": dict(conv_impl=2, tile_h=16, prio_mode=0, convt_occ=1),
            "auto (default)": dict(conv_impl=2, tile_h=0, prio_mode=2, convt_occ=1)}
res = {k: [] for k in variants}
for rnd in range(5):
    for name, opts in variants.items():
        for k, v in opts.items():
            m.set_option(k, v)
        m.segment_dev(frames, F, 256, 256, area); m.sync()
        t0 = time.perf_counter()
        for _ in range(2):
            m.segment_dev(frames, F, 256, 256, area)
        m.sync()
        res[name].append(2 * F / (time.perf_counter() - t0))
for name, v in res.items():
    print(f"{name:34s} median {np.median(v):8.1f} fps  max {max(v):8.1f}  min {min(v):8.1f}", flush=True)
if len(sys.argv) > 2:
    for name in sys.argv[2:]:
        for k, v in variants[name].items():
            m.set_option(k, v)
        m.profile(frames, chunk, 256, 256, reps=2)
        prof = m.profile(frames, chunk, 256, 256, reps=8)
        tot = sum(p["ms"] for p in prof)
        print(f"--- {name}: chain {tot:.3f} ms")
        for p in prof:
            tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
            print(f"{p['layer']:26s} {p['kernel']:24s} {p['ms']:8.4f} ms {tf:7.1f} TF/s {100 * tf / 157.3:5.1f}%")
